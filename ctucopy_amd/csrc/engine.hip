// MI355X (gfx950) engine: kernels + C ABI (include/ctu_engine.h).
//
// Data layout in HBM
//   pcm   : one packed int16 arena; utterance i starts at sample_off[i] (multiple of 8 samples)
//   rows  : float32 [total_frames][D] in writer order (c1..cN, c0[, E]), utterance i at row_off[i]
//   tiles : int4 {utt, first frame, valid frames, -}; a tile is <= 64 consecutive frames of ONE utterance
//
// Front-end kernel (one 512-thread workgroup walks tiles; 2 workgroups per CU, 80 KiB LDS each)
//   phase 1  16 lanes per frame, 4 frames per wave pass: int16 -> pre-emphasis * Hamming -> DC removal
//            -> 512-pt real FFT as a 256-pt complex FFT (two in-register radix-16 stages, one LDS
//            transpose) -> untangle + |.|^2 -> P tile in LDS  [64 frames][257 bins]
//   (NR)     extended spectral subtraction: one lane per bin walks the tile's frames in order
//   phase 2  one lane per frame, bands split over the 8 waves: banded filter bank with wave-uniform
//            (scalar) weights -> ^0.33 / log -> DCT-II+lifter (or cosine iDFT + Levinson-Durbin + a->c)
//            -> rows
// MFMA is deliberately not used: the bank is banded (2 non-zeros per bin), the FFT is not a dense
// contraction at this size, and f32 MFMA runs at the VALU rate anyway.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ctu_engine.h"
#include "design.h"
#include "opts.h"


namespace {

constexpr int TILE = 64;       // frames per tile (= lanes of the per-frame phase)
constexpr int WG = 512;        // threads per workgroup (8 waves)
constexpr int NWAVE = WG / 64;
constexpr int PSTRIDE = 260;   // floats per P-tile row: 257 bins padded so rows stay 16-byte aligned (b128 reads in phase 2)
constexpr int LDS_2WG = 80 * 1024;  // two workgroups per CU fit when a workgroup's LDS stays at or under this
constexpr int MAX_LP = 16;     // Levinson order limit of the in-register recursion
constexpr int MAXC = 24;       // most coefficients accumulated per frame in phase 2 (cepstra incl. c0, or LP lags)
constexpr int PCM_ALIGN = 8;   // utterance starts are multiples of this many samples
constexpr int PCM_HEAD = 8;    // samples of padding before the first utterance (x[-2..-1] of frame 0 is loaded)
constexpr int PCM_TAIL = 64;   // padding after the last one (loads run to the end of the 32-sample row)

// Per-lane constant record, one per l16 = lane & 15, streamed from L1 every pass instead of pinning
// 70+ VGPRs:  [0,32) Hamming pairs (w[32j+2l], w[32j+2l+1]) j=0..15 | [32,64) 1/0 "sample is inside the
// window" pairs for DC removal | [64,96) inter-stage twiddles W256^(l*k1), k1=1..15 (+pad) |
// [96,112) W512^(l+16*k2), k2=0..7
constexpr int LC_WIN = 0, LC_MASK = 32, LC_TW = 64, LC_UT = 96, LANEC = 112;
// The records are copied into LDS per workgroup: PCM streaming keeps evicting them from L1 and a miss costs
// ~1k cycles.  Row stride 116 floats makes the 16 lanes' ds_read_b128 conflict-free (116 mod 64 = 52).
constexpr int LTW_STRIDE = 116, LTW_FLOATS = 16 * LTW_STRIDE;

// Kernel variants by feature tail.  BANDS covers spec / logspec / the log-mel scratch of TRAP (runtime flags
// band_log, band_to_scratch); LP covers lpc and lpa (runtime flag lp_is_lpa).
enum FeatMode { FEAT_BANDS = 0, FEAT_DCTC = 2, FEAT_LP = 3 };

struct KParams {
    const int16_t *pcm;
    float *rows;
    float *logmel;              // [total_frames][B] scratch (TRAP only)
    float2 *xri;                // [total_frames][K] complex spectrum before NR (VAD cepdist-lpc only)
    float *pnr;                 // [total_frames][K] spectrum after NR (VAD cepdist-lpc) or [total_frames] energy (VAD energy)
    int vad_export;             // 0 none, 1 spectra for the Burg-cepstral criterion, 2 frame energy criterion
    int band_log, band_to_scratch, lp_is_lpa;
    const struct TileRec *tiles;
    const int *wg_first;        // [grid] first tile of each workgroup's chain (-1 = none)
    const float *lanec;         // [16][LANEC]
    const float *ftab;          // image of the LDS tables (tab_floats), then the lifter at lift_off
    const int *itab;            // slot_chunk[NS+1] | row_slot[nfea]
    // LDS tables (float index): chunk weights float4 [NC][8] at 0 | cell {first bin, band index or -1} (int2)
    // [NS][8] at ck_off | per-cell coefficient rows [NS][8][CW] at cf_off
    int tab_floats, ck_off, cf_off, NS, CW;
    int ncoef_out;              // DCTC: coefficients written per row (table rows are in output order)
    int e_mode, e_slot, K, window;  // -fea_E: 0 none, 1 spectrum (nr->E), 2 log R[0], 3 band energy, 4 raw frame energy
    int wshift, B, nfea, D, ncep, lporder;
    int lift_off;
    float preem, inv_window;
    int remove_dc, fb_power, fb_inld, lifter_on, nr_exten;
    float nr_p, nr_a;
    unsigned long long *stamps;  // [grid][NWAVE][16] (CTU_STAMP builds)
    int skip_phase2;  // signal output (row N3): spectra are exported, nothing is projected
    int dbg;  // diagnostic ablation (CTU_DEBUG_MODE): 1 = phase 1 only, 2 = phase 2 only; 0 in production
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// Radix-4 butterfly, forward transform (W4 = -i).
__device__ __forceinline__ void bfly4(float2 &p0, float2 &p1, float2 &p2, float2 &p3) {
    const float2 s02 = make_float2(p0.x + p2.x, p0.y + p2.y), d02 = make_float2(p0.x - p2.x, p0.y - p2.y);
    const float2 s13 = make_float2(p1.x + p3.x, p1.y + p3.y), d13 = make_float2(p1.x - p3.x, p1.y - p3.y);
    p0 = make_float2(s02.x + s13.x, s02.y + s13.y);
    p2 = make_float2(s02.x - s13.x, s02.y - s13.y);
    p1 = make_float2(d02.x + d13.y, d02.y - d13.x);  // d02 - i*d13
    p3 = make_float2(d02.x - d13.y, d02.y + d13.x);  // d02 + i*d13
}

// In-register 16-point DFT, natural order in and out: x[n] -> X[k] = sum_n x[n] W16^(nk).
// n = 4a+b, k = c+4d:  X[c+4d] = sum_b W4^(bd) * W16^(bc) * sum_a x[4a+b] W4^(ac).
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
    for (int b = 0; b < 4; b++) bfly4(v[b], v[4 + b], v[8 + b], v[12 + b]);  // v[4c+b] = y_b[c]
    // twiddles W16^(b*c), b,c in 1..3
    v[4 * 1 + 1] = cmul(v[4 * 1 + 1], make_float2(C1, -S1));                                   // W^1
    v[4 * 1 + 2] = make_float2((v[4 * 1 + 2].x + v[4 * 1 + 2].y) * R2, (v[4 * 1 + 2].y - v[4 * 1 + 2].x) * R2);  // W^2
    v[4 * 1 + 3] = cmul(v[4 * 1 + 3], make_float2(S1, -C1));                                   // W^3
    v[4 * 2 + 1] = make_float2((v[4 * 2 + 1].x + v[4 * 2 + 1].y) * R2, (v[4 * 2 + 1].y - v[4 * 2 + 1].x) * R2);  // W^2
    v[4 * 2 + 2] = make_float2(v[4 * 2 + 2].y, -v[4 * 2 + 2].x);                                // W^4 = -i
    v[4 * 2 + 3] = make_float2((v[4 * 2 + 3].y - v[4 * 2 + 3].x) * R2, -(v[4 * 2 + 3].x + v[4 * 2 + 3].y) * R2);  // W^6
    v[4 * 3 + 1] = cmul(v[4 * 3 + 1], make_float2(S1, -C1));                                   // W^3
    v[4 * 3 + 2] = make_float2((v[4 * 3 + 2].y - v[4 * 3 + 2].x) * R2, -(v[4 * 3 + 2].x + v[4 * 3 + 2].y) * R2);  // W^6
    v[4 * 3 + 3] = cmul(v[4 * 3 + 3], make_float2(-C1, S1));                                   // W^9
#pragma unroll
    for (int c = 0; c < 4; c++) bfly4(v[4 * c + 0], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);  // v[4c+d] = X[c+4d]
    // reorder to natural: X[k] sits at v[4*(k&3) + (k>>2)]
    float2 t[16];
#pragma unroll
    for (int k = 0; k < 16; k++) t[k] = v[4 * (k & 3) + (k >> 2)];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = t[k];
}

// Wave-uniform tables are read through constant-address-space pointers so that they become scalar
// loads (s_load_dword*) into SGPRs instead of per-lane VMEM loads.
typedef __attribute__((address_space(4))) const float cf32;
typedef __attribute__((address_space(4))) const int ci32;
typedef __attribute__((address_space(4))) const int64_t ci64;
__device__ __forceinline__ cf32 *as_const(const float *p) { return (cf32 *)p; }
__device__ __forceinline__ ci32 *as_const(const int *p) { return (ci32 *)p; }
__device__ __forceinline__ ci64 *as_const(const int64_t *p) { return (ci64 *)p; }

// Sum over the 16 lanes of a DPP row, result in every lane: four row-rotate adds on the VALU
// (no LDS round trips, unlike __shfl_xor which lowers to ds_bpermute).
__device__ __forceinline__ float row16_allreduce_add(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x122 /* row_ror:2 */, 0xf, 0xf, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
    return x;
}

// Sum over 8 consecutive lanes (a frame's band groups), result in all 8: xor-1, xor-2 inside quads, then the
// mirrored half row brings in the other quad.
__device__ __forceinline__ float lanes8_allreduce_add(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141 /* row_half_mirror */, 0xf, 0xf, false));
    return x;
}

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

struct __attribute__((aligned(4))) pcm4 {  // four consecutive int16 samples, 4-byte aligned
    uint32_t lo, hi;
};

// Phase-2 helpers with a compile-time coefficient count (16 or MAXC) so that nothing branches per coefficient.
template <int NCW>
__device__ __forceinline__ void cell_accumulate(float (&c)[NCW], const float4 *cf, float y) {
    float4 k4[NCW / 4];
#pragma unroll
    for (int i = 0; i < NCW / 4; i++) k4[i] = cf[i];  // all loads first, then the FMAs
#pragma unroll
    for (int i = 0; i < NCW / 4; i++) {
        c[4 * i + 0] += k4[i].x * y;
        c[4 * i + 1] += k4[i].y * y;
        c[4 * i + 2] += k4[i].z * y;
        c[4 * i + 3] += k4[i].w * y;
    }
}
template <int NCW>
__device__ __forceinline__ void cells_reduce(float (&c)[NCW]) {
#pragma unroll
    for (int i = 0; i < NCW; i++) c[i] = lanes8_allreduce_add(c[i]);
}

// Tile record (32 bytes, read with one scalar load): where the tile's first frame starts in the PCM
// arena, where its first output row goes, how many of its 64 frame slots are real, the frame index of
// slot 0 inside its utterance, and the next tile this workgroup walks (-1 = done).
struct TileRec {
    int64_t sbase, rbase;
    int nvalid, t0, next, pad;
};

__device__ __forceinline__ TileRec load_rec(const TileRec *tiles, int tile) {
    ci32 *w = as_const(reinterpret_cast<const int *>(tiles)) + 8 * tile;
    TileRec r;
    r.sbase = ((int64_t)w[1] << 32) | (uint32_t)w[0];
    r.rbase = ((int64_t)w[3] << 32) | (uint32_t)w[2];
    r.nvalid = w[4];
    r.t0 = w[5];
    r.next = w[6];
    r.pad = 0;
    return r;
}

// NZ = number of 32-sample rows that can hold non-zero input (ceil(window/32)); rows >= NZ are
// literal zeros so the compiler prunes the first butterflies.
#ifndef CTU_LB
#define CTU_LB 4        // waves per SIMD the register allocator must leave room for (2 workgroups x 8 waves / 4 SIMDs)
#endif
#ifndef CTU_STAMP
#define CTU_STAMP 0     // diagnostic build: per-wave s_memtime sums per code segment (never in production)
#endif
#if CTU_STAMP
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long now_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        st_acc[i] += now_ - st_prev;                                                               \
        st_prev = now_;                                                                            \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#ifndef CTU_B64A
#define CTU_B64A 0      // experiment: one-pass float2 transpose in the first pass (raises register pressure: spills)
#endif
#ifndef CTU_LDSDMA
#define CTU_LDSDMA 0    // experiment: second pass's PCM by LDS-DMA during the first pass (no gain, costs LDS cycles)
#endif
// MODE 0: 512-point real FFT, one frame per 16-lane group, NZ = rows of 32 samples, two passes of 4 frames.
// MODE 1: 256-point real FFT, TWO frames per 16-lane group packed as re/im of the same 256-point complex FFT
//         (no twiddles in the untangle), NZ = rows of 16 samples, one pass of 8 frames.
// VX:     also export what the VAD kernels need (kept out of the default instantiation: it costs registers).
// NC:     coefficients accumulated per frame in phase 2 (16 or MAXC): a compile-time width keeps eight accumulators
//         and a code path out of the common instantiation (9 -> 2 spilled VGPRs, +5 %).
// GEN:    false = the plain chain (DC removal on, power spectrum, no -fea_E, no exten, no intensity-loudness law, no
//         diagnostics): the option flags below become constants, which frees 30 SGPRs and the last spills (+4 %).
template <int NZ, int FEAT, int MODE, bool VX, int NC, bool GEN>
__global__ __launch_bounds__(WG, CTU_LB) void frontend_kernel(const KParams p) {
    const int o_e_mode = GEN ? p.e_mode : 0, o_dbg = GEN ? p.dbg : 0;
    const bool o_fb_inld = GEN ? p.fb_inld != 0 : false, o_nr_exten = GEN ? p.nr_exten != 0 : false;
    const bool o_fb_power = GEN ? p.fb_power != 0 : true, o_remove_dc = GEN ? p.remove_dc != 0 : true;
    const bool o_skip_phase2 = GEN ? p.skip_phase2 != 0 : false;
    extern __shared__ __align__(16) float lds[];
    float *Pt = lds;                       // [TILE][PSTRIDE]
    float *ltab = lds + TILE * PSTRIDE;    // phase-2 tables (layout: KParams)
    float *ltw = ltab + p.tab_floats;      // [16][LTW_STRIDE] per-lane constant records

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15;   // n2 in stage 1, k1 in stage 2
    const int fg = lane >> 4;    // frame slot within the wave pass
    const int partner = ((lane & 48) | ((16 - l16) & 15)) << 2;  // byte address for ds_bpermute
    ci32 *slot_chunk = as_const(p.itab), *row_slot = slot_chunk + p.NS + 1;
    cf32 *ftab = as_const(p.ftab);
    for (int i = tid; i < p.tab_floats; i += WG) ltab[i] = p.ftab[i];
    for (int i = tid; i < 16 * LANEC; i += WG) ltw[(i / LANEC) * LTW_STRIDE + (i % LANEC)] = p.lanec[i];
    // Phase 2 reads whole 4-bin chunks, so bins a frame never writes (row padding 257..259; everything above bin
    // 128 in the 256-point mode) are read under zero weights: start the tile finite.  Only finite values (spectra,
    // transpose scratch) are ever written afterwards.
    for (int i = tid; i < TILE * PSTRIDE; i += WG) Pt[i] = 0.f;
    __syncthreads();
    const float4 *lc = reinterpret_cast<const float4 *>(ltw + l16 * LTW_STRIDE);  // this lane's constant record
    const float4 *ltw4 = lc + (LC_TW >> 2);                                       // [0,8) stage twiddles, [8,12) untangle

#if CTU_STAMP
    unsigned long long st_acc[16] = {0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
    int tile = as_const(p.wg_first)[blockIdx.x];
    if (tile < 0) return;
    TileRec rec = load_rec(p.tiles, tile);

    // samples x[i0-2 .. i0+1] of row j of frame slot f sit at pcm + sbase + f*wshift + 32 j + 2 l16 - 2
    auto pcm_ptr = [&](const TileRec &r, int it) {
        const int f = wave * 8 + it * 4 + fg;
        const int fc = f < r.nvalid ? f : r.nvalid - 1;  // clamp: duplicates are computed but never stored
        return p.pcm + r.sbase + (int64_t)fc * p.wshift + 2 * l16 - 2;
    };
    // exten NR state: thread = bin
    float navg = 0.95f, yavg = 0.05f;

    while (true) {
        const int nvalid = rec.nvalid;
        const int64_t rbase = rec.rbase;
        const int next = rec.next;
        TileRec nrec = rec;
        if (next >= 0) nrec = load_rec(p.tiles, next);
        // this wave owns frame slots [8*wave, 8*wave+8) of the tile and the P rows of the same numbers
        const int nv = min(max(nvalid - wave * 8, 0), 8);

        // ================= phase 1: frames -> power spectrum rows =================
        // Pass A (frame slots 0-3 of the wave) loads its PCM from global memory and meanwhile has the PCM of
        // pass B (slots 4-7) copied by LDS-DMA into the wave's rows 4-7, which nobody needs before pass B
        // writes its spectra there.  The transpose scratch is rows 0-3 in pass A and rows 4-7 in pass B.
        constexpr bool DMA = CTU_LDSDMA && (NZ <= 15) && MODE == 0;  // a frame's 32*NZ+8 samples must fit 64 lanes x 8 samples
        if (o_dbg != 2 && nv > 0) {
            const int npass = (MODE == 0 && nv > 4) ? 2 : 1;
            // the pass body is instantiated twice (it = 0, 1) so that the choice of transpose is made at compile time
            auto pass = [&](auto IT) {
                constexpr int it = decltype(IT)::value;
                const int f = wave * 8 + it * 4 + fg;  // frame slot in tile
                const bool file_start = (l16 == 0) && (rec.t0 + (f < nvalid ? f : nvalid - 1) == 0);
                float *scratch = Pt + (wave * 8 + (DMA ? 4 * it : 4)) * PSTRIDE;
                STAMP(0);  // loop overhead / previous tail
                float2 v[16];
                if constexpr (MODE == 1) {
                    // frames A = slot 2*fg, B = A+1 of this wave's 8; sample n = 16 j + l16 of each goes to re / im
                    const int fa = wave * 8 + 2 * fg, fb_ = fa + 1;
                    const int ca = fa < nvalid ? fa : nvalid - 1, cb_ = fb_ < nvalid ? fb_ : nvalid - 1;
                    const int16_t *xa = p.pcm + rec.sbase + (int64_t)ca * p.wshift + l16;
                    const int16_t *xb = p.pcm + rec.sbase + (int64_t)cb_ * p.wshift + l16;
                    const bool start_a = (l16 == 0) && (rec.t0 + ca == 0), start_b = (l16 == 0) && (rec.t0 + cb_ == 0);
                    float dca = 0.f, dcb = 0.f;
#pragma unroll
                    for (int j = 0; j < NZ; j++) {
                        const float4 w4 = lc[(LC_WIN + 2 * j) >> 2];
                        const float w = (j & 1) ? w4.z : w4.x;  // 0 beyond the window
                        float pa = (float)xa[16 * j - 1], pb = (float)xb[16 * j - 1];
                        const float a0 = (float)xa[16 * j], b0 = (float)xb[16 * j];
                        if (j == 0) {
                            pa = start_a ? 0.f : pa;
                            pb = start_b ? 0.f : pb;
                        }
                        const float ya = w * (a0 - p.preem * pa), yb = w * (b0 - p.preem * pb);
                        v[j] = make_float2(ya, yb);
                        dca += ya;
                        dcb += yb;
                    }
#pragma unroll
                    for (int j = NZ; j < 16; j++) v[j] = make_float2(0.f, 0.f);
                    STAMP(1);
                    if (o_remove_dc) {
                        const float ma = row16_allreduce_add(dca) * p.inv_window, mb = row16_allreduce_add(dcb) * p.inv_window;
#pragma unroll
                        for (int j = 0; j < NZ; j++) {
                            const float4 mk = lc[(LC_MASK + 2 * j) >> 2];
                            const float mm = (j & 1) ? mk.z : mk.x;
                            v[j].x -= ma * mm;
                            v[j].y -= mb * mm;
                        }
                    }
                } else {
                float dc = 0.f;
                pcm4 q[NZ];
                if (!DMA || it == 0) {
                    const int16_t *x = pcm_ptr(rec, it);
#pragma unroll
                    for (int j = 0; j < NZ; j++) q[j] = *reinterpret_cast<const pcm4 *>(x + 32 * j);
                    if (DMA && npass == 2) {
                        const int cl = lane < (32 * NZ + 8) / 8 ? lane : (32 * NZ + 8) / 8 - 1;  // 8-sample chunks of a frame
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int fb = wave * 8 + 4 + k;
                            const int16_t *src = p.pcm + rec.sbase + (int64_t)(fb < nvalid ? fb : nvalid - 1) * p.wshift - 8 + 8 * cl;
                            __builtin_amdgcn_global_load_lds((gvoid_t *)src, (lvoid_t *)(Pt + (wave * 8 + 4) * PSTRIDE + 256 * k), 16, 0, 0);
                        }
                    }
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA issued in pass A has landed
                    const uint32_t *lp = reinterpret_cast<const uint32_t *>(Pt + (wave * 8 + 4) * PSTRIDE + 256 * fg) + 3 + l16;
#pragma unroll
                    for (int j = 0; j < NZ; j++) {
                        q[j].lo = lp[16 * j];
                        q[j].hi = lp[16 * j + 1];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int j = 0; j < NZ; j++) {
                    const float4 w4 = lc[(LC_WIN + 2 * j) >> 2];  // two rows of window pairs per float4
                    const float w0 = (j & 1) ? w4.z : w4.x, w1 = (j & 1) ? w4.w : w4.y;
                    float xm = (float)(int16_t)(q[j].lo >> 16);
                    const float x0 = (float)(int16_t)(q[j].hi & 0xffffu);
                    const float x1 = (float)(int16_t)(q[j].hi >> 16);
                    if (j == 0) xm = file_start ? 0.f : xm;  // first sample of the file: history is 0
                    const float y0 = w0 * (x0 - p.preem * xm);
                    const float y1 = w1 * (x1 - p.preem * x0);  // w is 0 beyond the window
                    v[j] = make_float2(y0, y1);
                    dc += y0 + y1;
                }
#pragma unroll
                for (int j = NZ; j < 16; j++) v[j] = make_float2(0.f, 0.f);
                STAMP(1);  // PCM + window loads, convert, pre-emphasis, window
                if (o_remove_dc) {
                    // mean of the windowed frame over `window` samples (src/io/in.cc:375-382)
                    const float m = row16_allreduce_add(dc) * p.inv_window;
                    if (NZ == 16) {  // generic instantiation: any window <= 512, per-sample masks
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            const float4 mk = lc[(LC_MASK + 2 * j) >> 2];
                            v[j].x -= m * ((j & 1) ? mk.z : mk.x);
                            v[j].y -= m * ((j & 1) ? mk.w : mk.y);
                        }
                    } else {  // exact instantiation: rows < NZ-1 are fully inside the window
                        const float4 mk = lc[(LC_MASK + 2 * (NZ - 1)) >> 2];
#pragma unroll
                        for (int j = 0; j < NZ - 1; j++) {
                            v[j].x -= m;
                            v[j].y -= m;
                        }
                        v[NZ - 1].x -= m * (((NZ - 1) & 1) ? mk.z : mk.x);
                        v[NZ - 1].y -= m * (((NZ - 1) & 1) ? mk.w : mk.y);
                    }
                }

                }
                STAMP(2);  // DC removal
                // ---- stage 1: DFT16 over n1 (registers), lane = n2; then twiddle W256^(n2*k1)
                dft16(v);
                __builtin_amdgcn_sched_barrier(0);  // twiddles are L1 hits: fetch them just in time, not across the DFT
#pragma unroll
                for (int h = 0; h < 8; h++) {
                    const float4 tw = ltw4[h];  // (k1 = 2h+1, k1 = 2h+2)
                    v[2 * h + 1] = cmul(v[2 * h + 1], make_float2(tw.x, tw.y));
                    if (2 * h + 2 < 16) v[2 * h + 2] = cmul(v[2 * h + 2], make_float2(tw.z, tw.w));
                }
                __builtin_amdgcn_sched_barrier(0);
                STAMP(3);  // DFT16 #1 + twiddles
                // ---- transpose [k1][n2] -> lane k1 holds all n2, through LDS.
                //      element (k1,n2) of frame slot fg at  fg*256 + ((k1^(fg&1))<<4) + (n2^k1): conflict-free
                //      both ways.  Pass A has all 8 rows of the wave free: one pass of float2 (b64).  Pass B has only
                //      rows 4-7 (rows 0-3 already hold pass A's spectra): re then im (b32).
                const int sw = fg * 256;
                const int par = fg & 1;
                __builtin_amdgcn_wave_barrier();
                if (CTU_B64A && !DMA && it == 0) {
                    float2 *sc2 = reinterpret_cast<float2 *>(Pt + wave * 8 * PSTRIDE);
#pragma unroll
                    for (int k1 = 0; k1 < 16; k1++) sc2[sw + ((k1 ^ par) << 4) + (l16 ^ k1)] = v[k1];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int n2 = 0; n2 < 16; n2++) v[n2] = sc2[sw + ((l16 ^ par) << 4) + (n2 ^ l16)];
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                } else {
#pragma unroll
                    for (int k1 = 0; k1 < 16; k1++) scratch[sw + ((k1 ^ par) << 4) + (l16 ^ k1)] = v[k1].x;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    float re[16];
#pragma unroll
                    for (int n2 = 0; n2 < 16; n2++) re[n2] = scratch[sw + ((l16 ^ par) << 4) + (n2 ^ l16)];
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k1 = 0; k1 < 16; k1++) scratch[sw + ((k1 ^ par) << 4) + (l16 ^ k1)] = v[k1].y;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int n2 = 0; n2 < 16; n2++) v[n2] = make_float2(re[n2], scratch[sw + ((l16 ^ par) << 4) + (n2 ^ l16)]);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }

                STAMP(4);  // LDS transpose
                // ---- stage 2: DFT16 over n2, lane = k1: v[k2] = Z[k1 + 16 k2]
                dft16(v);
                STAMP(5);  // DFT16 #2

                if constexpr (MODE == 1) {
                    // two real frames in one complex FFT: XA[k] = (Z[k] + conj Z[256-k])/2, XB[k] = (Z[k] - conj Z[256-k])/2i;
                    // bins 0..128 of both; the mirror bin comes from lane (16-k1)%16 as in MODE 0
                    float *pa = Pt + (wave * 8 + 2 * fg) * PSTRIDE, *pb = pa + PSTRIDE;
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) {
                        if ((k2 & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                        float br = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(v[15 - k2].x)));
                        float bi = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(v[15 - k2].y)));
                        if (l16 == 0) {
                            br = v[(16 - k2) & 15].x;
                            bi = v[(16 - k2) & 15].y;
                        }
                        const float ar = v[k2].x, ai = v[k2].y;
                        const float sr = ar + br, si = ai - bi, dr = ar - br, di = ai + bi;
                        const int k = l16 + 16 * k2;
                        pa[k] = 0.25f * (sr * sr + si * si);
                        pb[k] = 0.25f * (dr * dr + di * di);
                        if (VX && p.vad_export == 1) {  // XA = s/2, XB = (d)/(2i) = (di - i dr)/2
                            const int fa = wave * 8 + 2 * fg;
                            if (fa < nvalid) p.xri[(rbase + fa) * 129 + k] = make_float2(0.5f * sr, 0.5f * si);
                            if (fa + 1 < nvalid) p.xri[(rbase + fa + 1) * 129 + k] = make_float2(0.5f * di, -0.5f * dr);
                        }
                    }
                    if (l16 == 0) {
                        pa[128] = v[8].x * v[8].x;
                        pb[128] = v[8].y * v[8].y;
                        if (o_remove_dc) pa[0] = pb[0] = 1e-10f;
                        if (VX && p.vad_export == 1) {
                            const int fa = wave * 8 + 2 * fg;
                            if (fa < nvalid) p.xri[(rbase + fa) * 129 + 128] = make_float2(v[8].x, 0.f);
                            if (fa + 1 < nvalid) p.xri[(rbase + fa + 1) * 129 + 128] = make_float2(v[8].y, 0.f);
                        }
                    }
                } else {
                // ---- untangle the packed real FFT and take |.|^2.  Lane k1 handles its bins k2=0..7,
                //      each together with its mirror bin 256-k held by lane (16-k1)%16 in register 15-k2
                //      (register (16-k2)%16 for k1 = 0).
                float *prow = Pt + f * PSTRIDE;
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) {
                    if ((k2 & 3) == 0) __builtin_amdgcn_sched_barrier(0);  // two batches: bounds the registers in flight
                    const float4 u4q = ltw4[8 + (k2 >> 1)];
                    float br = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(v[15 - k2].x)));
                    float bi = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(v[15 - k2].y)));
                    if (l16 == 0) {
                        br = v[(16 - k2) & 15].x;
                        bi = v[(16 - k2) & 15].y;
                    }
                    const float wr = (k2 & 1) ? u4q.z : u4q.x, wi = (k2 & 1) ? u4q.w : u4q.y;
                    const float ar = v[k2].x, ai = v[k2].y;
                    const float sr = ar + br, si = ai - bi, dr = ar - br, di = ai + bi;
                    const float tr = wr * di + wi * dr;
                    const float ti = wi * di - wr * dr;
                    const float ur = sr + tr, ui = si + ti, vr = sr - tr, vi = si - ti;
                    const float pk = 0.25f * (ur * ur + ui * ui);
                    const float pm = 0.25f * (vr * vr + vi * vi);
                    const int k = l16 + 16 * k2;
                    prow[k] = pk;
                    prow[256 - k] = pm;
                    if (VX && p.vad_export == 1 && f < nvalid) {  // X[k] = u/2, X[256-k] = conj(v)/2
                        float2 *xo = p.xri + (rbase + f) * 257;
                        xo[k] = make_float2(0.5f * ur, 0.5f * ui);
                        xo[256 - k] = make_float2(0.5f * vr, -0.5f * vi);
                    }
                }
                if (l16 == 0) {  // bin 128 is its own mirror: X[128] = conj(Z[128]); bin 0 floor (src/io/in.cc:390)
                    prow[128] = v[8].x * v[8].x + v[8].y * v[8].y;
                    if (o_remove_dc) prow[0] = 1e-10f;
                    if (VX && p.vad_export == 1 && f < nvalid) p.xri[(rbase + f) * 257 + 128] = make_float2(v[8].x, -v[8].y);
                }
                }
                STAMP(6);  // untangle + P writes
            };
            pass(std::integral_constant<int, 0>{});
            if (npass == 2) pass(std::integral_constant<int, 1>{});
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        if (!o_fb_power && nv > 0) {  // magnitude instead of power (src/io/in.cc:415-417); off the default path
            for (int e = lane; e < nv * p.K; e += 64) {
                const int f = e / p.K, k = e - f * p.K;
                float *q_ = Pt + (wave * 8 + f) * PSTRIDE + k;
                *q_ = sqrtf(*q_);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        // ================= extended spectral subtraction (src/nr/nr.cc:86-140) =================
        // The only cross-wave step: one lane per bin walks the tile's frames in order (workgroup barriers).
        if (o_nr_exten) {
            __syncthreads();
            if (rec.t0 == 0) {  // new file: Navg = 0.95, Yavg = 0.05
                navg = 0.95f;
                yavg = 0.05f;
            }
            if (tid < p.K) {
                const float pp = p.nr_p, qq = 1.0f - p.nr_p;
                for (int f = 0; f < nvalid; f++) {
                    const float X = Pt[f * PSTRIDE + tid];
                    // H = Navg / (Navg^a + Yavg^a)^(1/a); the output X - H X is formed as X (1 - H) with 1 - H written
                    // without cancellation (fp32 here, double in the reference; double was measured: no accuracy gain,
                    // -15 % on the main path through register allocation)
                    // (v_rcp_f32 / v_rsq_f32, ~1 ulp, instead of IEEE division and square root: this loop is a serial chain)
                    float H, omH;
                    if (p.nr_a == 1.0f) {
                        const float ir = __builtin_amdgcn_rcpf(navg + yavg);
                        H = navg * ir;
                        omH = yavg * ir;
                    } else if (p.nr_a == 2.0f) {
                        const float r2 = navg * navg + yavg * yavg;
                        const float ir = __builtin_amdgcn_rsqf(r2);
                        const float r = r2 * ir;
                        H = navg * ir;
                        omH = (yavg * yavg) * __builtin_amdgcn_rcpf(r * (r + navg));
                    } else {
                        H = navg / powf(powf(navg, p.nr_a) + powf(yavg, p.nr_a), 1.0f / p.nr_a);
                        omH = 1.0f - H;
                    }
                    const float N = H * X;
                    navg = pp * navg + qq * N;
                    yavg = fabsf(X - navg);
                    Pt[f * PSTRIDE + tid] = X * omH;
                }
            }
            __syncthreads();
        }
        if (VX && p.vad_export && nv > 0) {  // the VAD looks at in->_Xsabs after NR (src/io/batch.cc:230-240, src/vad/vad.cc:96-107,227-230)
            if (p.vad_export == 1) {
                for (int e = lane; e < nv * p.K; e += 64) {
                    const int f = e / p.K, k = e - f * p.K;
                    p.pnr[(rbase + wave * 8 + f) * p.K + k] = Pt[(wave * 8 + f) * PSTRIDE + k];
                }
            } else {
                const int f8e = lane >> 3, ge = lane & 7;
                float es = 0.f;
                for (int k = ge; k < p.K; k += 8) {
                    const float x = Pt[(wave * 8 + f8e) * PSTRIDE + k];
                    es += x * x;
                }
                es = lanes8_allreduce_add(es);
                if (ge == 0 && f8e < nv) p.pnr[rbase + wave * 8 + f8e] = es;
            }
        }
        STAMP(7);  // hand-over to phase 2 (incl. NR)

        // ================= phase 2 (wave-local): lane = (frame, band group) =================
        // The wave's 8 frames x 8 band groups.  Bands are dealt to (slot, group) cells by the host so that the
        // 8 bands of a slot have similar widths; every group walks the same number of 4-bin chunks per slot.
        if (o_dbg != 1 && !o_skip_phase2 && nv > 0) {
            const int f8 = lane >> 3, g = lane & 7;
            const int fslot = wave * 8 + f8;
            const bool fvalid = f8 < nv;
            const float *prow2 = Pt + fslot * PSTRIDE;
            float c[NC];
#pragma unroll
            for (int i = 0; i < NC; i++) c[i] = 0.f;
            float esum = 0.f;
            if (o_e_mode == 4) {  // raw energy: sum of x[i]^2, i = 1..window-1 (src/io/in.cc:353-361); rare, read from HBM
                const int16_t *xr = p.pcm + rec.sbase + (int64_t)(f8 < nv ? fslot : wave * 8) * p.wshift;
                for (int i = 1 + g; i < p.window; i += 8) {
                    const float x = (float)xr[i];
                    esum += x * x;
                }
            }
            if (o_e_mode == 1) {  // E = log(2 (X0^2/2 + sum X_i^2 + X_{K-1}^2/2)) on the post-NR vector (src/nr/nr.cc:36-45)
                for (int k = g; k < p.K; k += 8) {
                    const float x = prow2[k];
                    esum += ((k == 0 || k == p.K - 1) ? 0.5f : 1.0f) * x * x;
                }
            }
            for (int sl = 0; sl < p.NS; sl++) {
                const int cb = slot_chunk[sl], ce = slot_chunk[sl + 1];
                // {first bin of the cell's chunk run, band index or -1}: the only per-lane indirection of the slot
                const int kstart = __float_as_int(ltab[p.ck_off + (sl * 8 + g) * 2]);
                const int bidx = __float_as_int(ltab[p.ck_off + (sl * 8 + g) * 2 + 1]);
                const float4 *pq = reinterpret_cast<const float4 *>(prow2 + kstart);  // kstart is a multiple of 4
                const float4 *wq = reinterpret_cast<const float4 *>(ltab) + cb * 8 + g;
                float acc = 0.f, acc1 = 0.f;
                const int nch = ce - cb;
                int ch = 0;
                for (; ch + 4 <= nch; ch += 4) {  // 4 chunks per group: 12 LDS reads in flight, then 16 FMAs
                    float4 w4[4], p4[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) w4[u] = wq[(ch + u) * 8];
#pragma unroll
                    for (int u = 0; u < 4; u++) p4[u] = pq[ch + u];
#pragma unroll
                    for (int u = 0; u < 4; u += 2) {
                        acc += w4[u].x * p4[u].x;
                        acc1 += w4[u + 1].x * p4[u + 1].x;
                        acc += w4[u].y * p4[u].y;
                        acc1 += w4[u + 1].y * p4[u + 1].y;
                        acc += w4[u].z * p4[u].z;
                        acc1 += w4[u + 1].z * p4[u + 1].z;
                        acc += w4[u].w * p4[u].w;
                        acc1 += w4[u + 1].w * p4[u + 1].w;
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // DS reads
                    __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);  // VALU
                }
                for (; ch < nch; ch++) {
                    const float4 w4 = wq[ch * 8];
                    const float4 p4 = pq[ch];
                    acc += w4.x * p4.x;
                    acc1 += w4.y * p4.y;
                    acc += w4.z * p4.z;
                    acc1 += w4.w * p4.w;
                }
                acc += acc1;
                float y = acc;
                if (o_fb_inld) y = __builtin_amdgcn_exp2f(0.33f * __builtin_amdgcn_logf(y));  // pow(Y, 0.33), src/fea/fb.cc:81-83
                // v_log_f32 (log2, ~1 ulp) * ln 2: band energies of int16 speech are far from the denormal range
                if (FEAT == FEAT_DCTC || (FEAT == FEAT_BANDS && p.band_log)) y = __builtin_amdgcn_logf(y) * 0.69314718056f;
                if (o_e_mode == 3 && bidx >= 0)  // band energy of the FB output (src/fea/fea_impl.cc:44-50,68-74)
                    esum += ((bidx == 0 || bidx == p.B - 1) ? 0.5f : 1.0f) * acc * acc;
                if (FEAT == FEAT_BANDS) {
                    float *dst = p.band_to_scratch ? p.logmel : p.rows;
                    const int out_w = p.band_to_scratch ? p.B : p.D;
                    if (bidx >= 0 && fvalid) dst[(rbase + fslot) * out_w + bidx] = y;
                } else {
                    if (FEAT == FEAT_LP && !o_fb_inld) y *= y;  // src/fea/fea_impl.cc:165-169
                    y = bidx >= 0 ? y : 0.f;  // idle cell: its log(0) must not meet the zero coefficients
                    const float4 *cf = reinterpret_cast<const float4 *>(ltab + p.cf_off + (sl * 8 + g) * (NC + 4));  // +4: bank spread
                    cell_accumulate<NC>(c, cf, y);
                }
            }
            STAMP(8);  // filter bank + per-band accumulation
            if (o_e_mode && !(FEAT == FEAT_BANDS && p.band_to_scratch)) {
                float e = 0.f;
                if (o_e_mode == 1 || o_e_mode == 3) e = __logf(2.0f * lanes8_allreduce_add(esum));
                else if (o_e_mode == 4) e = __logf(lanes8_allreduce_add(esum));
                if (o_e_mode != 2 && fvalid && g == 0) p.rows[(rbase + fslot) * p.D + p.e_slot] = e;
            }
            if (FEAT == FEAT_DCTC || FEAT == FEAT_LP) {
                cells_reduce<NC>(c);
                float *orow = p.rows + (rbase + fslot) * p.D;
                if (FEAT == FEAT_DCTC) {
                    // c[r] = value of output slot r = sum_b dct[i(r)][b] * logY[b]  (norm, lifter and the writer's
                    // c1..cN,c0 order are folded into the table on the host); lane g stores slots g, g+8, g+16
#pragma unroll
                    for (int h = 0; h < NC / 8; h++) {
                        if (h * 8 < p.ncoef_out) {
                            float val = c[h * 8];
#pragma unroll
                            for (int j = 1; j < 8; j++) val = (g == j) ? c[h * 8 + j] : val;
                            if (fvalid && h * 8 + g < p.ncoef_out) orow[h * 8 + g] = val;
                        }
                    }
                } else {
                    // c[k] = R[k], the autocorrelation by cosine iDFT (src/fea/fea_impl.cc:181-198); every lane of the
                    // frame runs Levinson-Durbin in double (src/fea/fea_impl.cc:200-222; the reference's aa[] copy is
                    // replaced by the in-place symmetric update, same operations), then a -> c (251-284)
                    // fp32: with the cube-root (or squared) band energies the autocorrelation matrix is well
                    // conditioned; measured deviation from a double recursion ~1e-6 (tests/test_gpu_parity.py::test_c3_plp)
                    const int P_ = p.lporder;
                    float a[MAX_LP + 1], cc[MAX_LP + 1];
                    const float r0 = c[0];
                    if (o_e_mode == 2 && fvalid && g == 0) orow[p.e_slot] = __builtin_amdgcn_logf(r0) * 0.69314718056f;  // E = ln R[0] (src/fea/fea_impl.cc:177)
                    float rc = -c[1] / r0;
                    float err = r0 * (1 - rc * rc);
                    a[0] = 1;
                    a[1] = rc;
#pragma unroll
                    for (int ik = 2; ik <= MAX_LP; ik++) {
                        if (ik <= P_ && ik < NC) {  // the host picks NC > lporder
                            float dm = c[ik < NC ? ik : NC - 1];
#pragma unroll
                            for (int n = 1; n < ik; n++) dm += a[n] * c[ik - n];
                            rc = -dm / err;
#pragma unroll
                            for (int n = 1; n <= ik / 2; n++) {
                                const float lo = a[n], hi = a[ik - n];
                                a[n] = lo + rc * hi;
                                if (n != ik - n) a[ik - n] = hi + rc * lo;
                            }
                            a[ik] = rc;
                            err *= (1 - rc * rc);
                        }
                    }
                    if (p.lp_is_lpa) {
#pragma unroll
                        for (int i = 1; i <= MAX_LP; i++)
                            if (i <= P_ && fvalid && g == (i & 7)) orow[i - 1] = a[i];
                    } else {
                        cc[0] = __builtin_amdgcn_logf(err) * 0.69314718056f;
#pragma unroll
                        for (int n = 1; n <= MAX_LP; n++) {
                            if (n <= p.ncep) {
                                float sum = 0;
#pragma unroll
                                for (int k = 1; k < n; k++)
                                    if (k <= P_) sum += (float)(n - k) * cc[n - k] * a[k];
                                cc[n] = (n <= P_ ? -a[n] : 0.0f) - sum / (float)n;
                            }
                        }
#pragma unroll
                        for (int n = 0; n <= MAX_LP; n++) {
                            if (n <= p.ncep) {
                                float val = cc[n];
                                if (n >= 1 && p.lifter_on) val *= ftab[p.lift_off + n - 1];
                                const int slot = row_slot[n];
                                if (slot >= 0 && fvalid && g == (n & 7)) orow[slot] = val;
                            }
                        }
                    }
                }
            }
            STAMP(10);  // reduction, tail, row store
        }
        if (o_nr_exten) __syncthreads();  // the bin-wise NR pass of the next tile reads every wave's rows
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (next < 0) break;
        rec = nrec;
    }
#if CTU_STAMP
    if (lane == 0 && p.stamps)
        for (int i = 0; i < 16; i++) p.stamps[(blockIdx.x * NWAVE + wave) * 16 + i] = st_acc[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// VAD module (src/vad/vad.cc, src/vad/vad.h, src/vdet/Burg.h).  Decisions are discontinuous, so this side path
// computes in double.  Kernel A is frame-parallel (HC2R of the post-NR spectrum with the original phase, Burg
// lattice, a -> c); kernel B is one thread per utterance and replays the sequential part: cepstral distance to
// the adaptive background, threshold recurrences, background update, majority ("median") filter.
// ------------------------------------------------------------------------------------------------
struct VadParams {
    int K, wfft, window, ncoef;  // ncoef = vad_lpc_coefs (cepdist lpc) or feature vector length (cepdist fea)
    int cri;                     // 0 energy, 1 cepdist-lpc, 2 cepdist-fea
    int thr;                     // 0 absolute, 1 perc, 2 adapt, 3 dyn
    int energy_db, cep_init, filter_order;
    double cep_p, abs_thr, perc_thr, adapt_q, adapt_za, dyn_perc, dyn_min, qmaxinc, qmaxdec, qmindec, qmininc;
    int perc_init, adapt_init, dyn_init;
    int D, ncep, c0_slot;        // cepdist-fea: where the internal vector sits in a written row
};

// One wave per frame (4 frames per 256-thread workgroup): the frame's samples live in registers, strided over the
// lanes (sample j = lane + 64 q), reductions are wave shuffles, no workgroup barrier inside the lattice.
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

#ifndef CTU_VAD_REAL
#define CTU_VAD_REAL double  // arithmetic of the HC2R + Burg kernel (float was measured: see DESIGN.md)
#endif
typedef CTU_VAD_REAL vreal;
__device__ __forceinline__ vreal wave_sum_r(vreal x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// Row rotation by DPP for 32- and 64-bit values (v_mov_b32_dpp per half), and a wave all-reduce built on it: four
// rotate-and-add steps inside each row of 16 lanes, then the four row sums through v_readlane.  A shuffle-based
// butterfly (ds_bpermute) costs an LDS round trip per step; the lattice below runs two reductions per order.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_read(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ double lane_read(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
template <class T>
__device__ __forceinline__ T wave_sum_fast(T x) {
    x += dpp_mov<0x128>(x);  // row_ror:8
    x += dpp_mov<0x124>(x);  // row_ror:4
    x += dpp_mov<0x122>(x);  // row_ror:2
    x += dpp_mov<0x121>(x);  // row_ror:1
    return (lane_read(x, 0) + lane_read(x, 16)) + (lane_read(x, 32) + lane_read(x, 48));
}

struct vreal2 { vreal x, y; };
__device__ __forceinline__ vreal2 vcmul(vreal2 a, vreal2 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

template <int Q, int NCMAX>  // samples per lane: window <= 64*Q; cepstral coefficients: ncoef <= NCMAX
__global__ __launch_bounds__(256) void vad_burg_kernel(const float2 *__restrict__ xri, const float *__restrict__ pnr,
                                                       double *__restrict__ ci_out, VadParams vp, int64_t total_frames) {
    extern __shared__ __align__(16) unsigned char burg_lds[];
    vreal2 *root = reinterpret_cast<vreal2 *>(burg_lds);  // [512] e^{+2 pi i m / 512}; then per wave 2 x [wfft/2 + 4] ping-pong
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = vp.K, n = vp.wfft, W = vp.window, nc = vp.ncoef;
    for (int m = tid; m < 512; m += 256) {
        double sd, cd;
        sincospi((double)m / 256.0, &sd, &cd);
        root[m] = {(vreal)cd, (vreal)sd};
    }
    __syncthreads();  // the only workgroup barrier; everything below is wave-local (persistent waves walk the frames)
    const int M = n / 2, rs = 512 / n, mr = 512 / M;
    vreal2 *A = root + 512 + (size_t)wave * 2 * (M + 4), *Bf = A + (M + 4);
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int64_t fr = (int64_t)blockIdx.x * 4 + wave; fr < total_frames; fr += (int64_t)gridDim.x * 4) {
    for (int k = lane; k < K; k += 64) {  // halfcomplex input: Xa cos(phi), Xa sin(phi)   (src/vad/vad.cc:227-230)
        const float2 x = xri[fr * K + k];
        const vreal xa = pnr[fr * K + k];
        vreal c = 1.0, s_ = 0.0;  // Xph[0] = 0 (src/io/in.cc:398)
        if (k > 0) {
            // direction of a float spectrum value: float arithmetic (v_rsq + one Newton step) is as exact as its input
            const float mag2 = x.x * x.x + x.y * x.y;
            if (mag2 > 0.f) {
                float inv = __builtin_amdgcn_rsqf(mag2);
                inv = inv * (1.5f - 0.5f * mag2 * inv * inv);
                c = (vreal)(x.x * inv);
                s_ = (vreal)(x.y * inv);
            } else {  // c_ph(0, 0) = -pi/2 (src/io/in.cc:191-193); the last bin is 0 or pi by the sign of re (:399)
                c = (k == K - 1) ? 1.0 : 0.0;
                s_ = (k == K - 1) ? 0.0 : -1.0;
            }
            if (k == K - 1) s_ = 0.0;
        }
        A[k] = {xa * c, (k == 0 || k == K - 1) ? (vreal)0.0 : xa * s_};  // FFTW's halfcomplex format has no imaginary DC / Nyquist
    }
    wave_sync();
    // FFTW_HC2R, unnormalised: x_j = sum over the Hermitian extension of X_k e^{+2 pi i jk/n}.  Packed half-size form:
    // Z[k] = (X[k] + X*[M-k]) + i e^{+2 pi i k/n} (X[k] - X*[M-k]),  z = IDFT_M(Z),  x[2m] = Re z[m], x[2m+1] = Im z[m]
    for (int k = lane; k < M; k += 64) {
        const vreal2 xa = A[k], xb = A[M - k];
        const vreal2 sm = {xa.x + xb.x, xa.y - xb.y}, df = {xa.x - xb.x, xa.y + xb.y};
        const vreal2 w = root[k * rs];
        Bf[k] = {sm.x - (w.x * df.y + w.y * df.x), sm.y + (w.x * df.x - w.y * df.y)};
    }
    wave_sync();
    vreal2 *src = Bf, *dst = A;
    int Ns = 1;
    while (Ns * 4 <= M) {  // radix-4 Stockham passes
        const int q4 = M / 4;
        for (int j = lane; j < q4; j += 64) {
            const int kk = j % Ns, tstep = kk * (M / (4 * Ns)) * mr;
            const vreal2 v0 = src[j];
            const vreal2 v1 = vcmul(src[j + q4], root[tstep & 511]);
            const vreal2 v2 = vcmul(src[j + 2 * q4], root[(2 * tstep) & 511]);
            const vreal2 v3 = vcmul(src[j + 3 * q4], root[(3 * tstep) & 511]);
            const vreal2 s02 = {v0.x + v2.x, v0.y + v2.y}, d02 = {v0.x - v2.x, v0.y - v2.y};
            const vreal2 s13 = {v1.x + v3.x, v1.y + v3.y}, d13 = {v1.x - v3.x, v1.y - v3.y};
            const int base = (j / Ns) * Ns * 4 + kk;
            dst[base] = {s02.x + s13.x, s02.y + s13.y};
            dst[base + Ns] = {d02.x - d13.y, d02.y + d13.x};
            dst[base + 2 * Ns] = {s02.x - s13.x, s02.y - s13.y};
            dst[base + 3 * Ns] = {d02.x + d13.y, d02.y - d13.x};
        }
        wave_sync();
        vreal2 *t_ = src; src = dst; dst = t_;
        Ns *= 4;
    }
    if (Ns < M) {  // radix-2 tail (M = 128)
        const int h = M / 2;
        for (int j = lane; j < h; j += 64) {
            const int kk = j % Ns;
            const vreal2 v0 = src[j], v1 = vcmul(src[j + h], root[(kk * (M / (2 * Ns)) * mr) & 511]);
            const int base = (j / Ns) * Ns * 2 + kk;
            dst[base] = {v0.x + v1.x, v0.y + v1.y};
            dst[base + Ns] = {v0.x - v1.x, v0.y - v1.y};
        }
        wave_sync();
        vreal2 *t_ = src; src = dst; dst = t_;
    }
    vreal ef[Q], eb[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int j = lane + 64 * q;
        const vreal2 z = src[(j >> 1) & (M - 1)];
        ef[q] = (j & 1) ? z.y : z.x;
    }
    wave_sync();  // the buffers are free for the next frame once every lane has its samples
    vreal part = 0.0;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        if (lane + 64 * q >= W) ef[q] = 0.0;  // only the first `window` samples go to Burg (src/vad/vad.cc:233)
        eb[q] = ef[q];
        part += ef[q] * ef[q];
    }
    vreal alpha = wave_sum_fast(part) / (vreal)W;
    // Burg lattice (src/vdet/Burg.h:49-95).  Prediction coefficients live one per lane (lane i = a[i]); the order
    // update a'[i] = a[i] + rc a[ik-i] is one cross-lane read.
    vreal acoef = lane == 0 ? (vreal)1.0 : (vreal)0.0;
    for (int ik = 1; ik < nc; ik++) {
        // eb[i-1]: the previous sample sits in the previous lane (or lane 63 of the previous q)
        vreal ebm[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const vreal up = dpp_mov<0x138>(eb[q]);  // wave_shr:1
            const vreal wrap = q > 0 ? lane_read(eb[q > 0 ? q - 1 : 0], 63) : (vreal)0.0;
            ebm[q] = lane == 0 ? wrap : up;
        }
        vreal num = 0.0, den = 0.0;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int i = lane + 64 * q;
            if (i >= ik && i < W) {
                den += ef[q] * ef[q] + ebm[q] * ebm[q];
                num += ef[q] * ebm[q];
            }
        }
        num = wave_sum_fast(num);
        den = wave_sum_fast(den);
        const vreal rc = -(2.0 * num) / den;
        alpha *= 1.0 - rc * rc;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int i = lane + 64 * q;
            if (i >= 1 && i < W) {  // both updates use the old values
                const vreal nef = ef[q] + rc * ebm[q], neb = ebm[q] + rc * ef[q];
                ef[q] = nef;
                eb[q] = neb;
            }
        }
        const vreal other = __shfl(acoef, (ik - lane) & 63, 64);
        acoef = (lane >= 1 && lane < ik) ? acoef + rc * other : (lane == ik ? rc : acoef);
    }
    // Burg2Cepstrum (src/vdet/Burg.h:141-152) in registers, the same in every lane; lane m keeps c[m] for the store
    {
        vreal av[NCMAX], cc[NCMAX];
#pragma unroll
        for (int i = 0; i < NCMAX; i++) av[i] = lane_read(acoef, i);
        cc[0] = (vreal)log((double)alpha);
        vreal mine = cc[0];
#pragma unroll
        for (int m = 1; m < NCMAX; m++) {
            vreal sum = 0.0;
#pragma unroll
            for (int k = 1; k < m; k++) sum += (vreal)(m - k) * cc[m - k] * av[k];
            cc[m] = -av[m] - sum * (vreal)(1.0 / m);
            mine = lane == m ? cc[m] : mine;
        }
        if (lane < nc) ci_out[fr * nc + lane] = (double)mine;
    }
    }  // frames
}

// One wave per utterance (src/vad/vad.cc:220-294 distance + background, :329-625 thresholds, vad.h:126-175 filter).
// The recurrences are sequential in t; the wave stages 64 frames of criterion inputs in LDS with coalesced loads,
// then every lane replays them (same values in all lanes, lane 0 writes).
__global__ __launch_bounds__(64) void vad_decide_kernel(const double *__restrict__ ci_all, const float *__restrict__ cri_energy,
                                                         const float *__restrict__ rows, const int64_t *__restrict__ row_off,
                                                         int n_utt, uint8_t *__restrict__ vad_out, VadParams vp) {
    __shared__ double stage[64 * 32];
    const int u = blockIdx.x, lane = threadIdx.x;
    if (u >= n_utt) return;
    const int64_t r0 = row_off[u];
    const int T = (int)(row_off[u + 1] - r0);
    const int nc = vp.cri == 0 ? 1 : vp.ncoef, order = vp.filter_order, h = (order - 1) / 2;
    // majority filter (src/vad/vad.h:126-175): the last `order` (<= 31) raw decisions as bits, with a running count
    unsigned long long hist = 0;
    int hidx = 0, nout = 0, nsum = 0;
    auto push = [&](int v) {
        const int old = (int)((hist >> hidx) & 1ull);
        hist = (hist & ~(1ull << hidx)) | ((unsigned long long)v << hidx);
        nsum += v - old;
        hidx = (hidx + 1 == order) ? 0 : hidx + 1;
    };
    double crimin = 0, crimax = 0, crimean = 0, crimean2 = 0, crivar = 0, dmin = 0, dmax = 0;
    int adapt_vad = 0;
    double c0r = 0.0;  // background cepstrum, coefficient `lane` (src/vad/vad.cc:220-294)
    for (int tb = 0; tb < T; tb += 64) {
        const int nt = min(64, T - tb);
        __syncthreads();
        if (vp.cri == 0) {
            if (lane < nt) stage[lane] = cri_energy[r0 + tb + lane];
        } else if (vp.cri == 1) {
            for (int e = lane; e < nt * nc; e += 64) stage[e] = ci_all[(r0 + tb) * nc + e];
        } else {  // internal vector order: c0 first, then c1..cN (src/fea/fea_impl.cc:104-131)
            for (int e = lane; e < nt * nc; e += 64) {
                const int f = e / nc, i = e - f * nc;
                const float *row = rows + (r0 + tb + f) * vp.D;
                stage[e] = i == 0 ? (vp.c0_slot >= 0 ? (double)row[vp.c0_slot] : 0.0) : (double)row[i - 1];
            }
        }
        __syncthreads();
        for (int tt = 0; tt < nt; tt++) {
            const int t = tb + tt;
            double cri, cil = 0.0;
            if (vp.cri == 0) {
                double en = stage[tt];
                if (vp.energy_db) en = 10.0 * log10(2.2250738585072014e-308 + en);
                cri = en;
            } else {
                cil = lane < nc ? stage[tt * nc + lane] : 0.0;
                if (t == 0) {
                    c0r = cil;
                    cri = 0.0;
                } else {
                    if (t == 1) c0r = (c0r + cil) / 2.0;
                    const double dl = (lane >= 1 && lane < nc) ? cil - c0r : 0.0;  // c0 itself is not part of the distance
                    cri = 4.3429 * sqrt(2 * wave_sum_fast(dl * dl));
                }
            }
            int vad0;
            if (vp.thr == 0) vad0 = cri >= vp.abs_thr;
            else if (vp.thr == 1) {
                if (t == 0 || (double)t < (double)vp.perc_init) crimin = crimax = cri;
                else {
                    crimin = cri < crimin ? cri : crimin;
                    crimax = cri > crimax ? cri : crimax;
                }
                vad0 = cri >= crimin + (vp.perc_thr / 100.0) * (crimax - crimin);
            } else if (vp.thr == 2) {
                if (t == 0) {
                    crimean = cri;
                    crimean2 = cri * cri;
                    crivar = 0.0;
                    adapt_vad = 0;
                } else {
                    const double thr = crimean + vp.adapt_za * sqrt(crivar);
                    if (cri < thr || t <= vp.adapt_init) {
                        crimean = vp.adapt_q * crimean + (1.0 - vp.adapt_q) * cri;
                        crimean2 = vp.adapt_q * crimean2 + (1.0 - vp.adapt_q) * cri * cri;
                        crivar = crimean2 - crimean * crimean;
                        adapt_vad = 0;
                    } else adapt_vad = 1;
                }
                vad0 = adapt_vad;
            } else {
                const int init = vp.dyn_init > 1 ? vp.dyn_init : 1;
                if (t < init) {
                    dmax = dmin = cri;
                    vad0 = 0;
                } else if (t == init) {
                    dmax = (cri > dmax ? cri : dmax) + vp.dyn_min / 10.0;
                    dmin = (cri < dmin ? cri : dmin) - vp.dyn_min / 10.0;
                    vad0 = 0;
                } else {
                    dmax = dmax < cri ? vp.qmaxinc * dmax + (1.0 - vp.qmaxinc) * cri : vp.qmaxdec * dmax + (1.0 - vp.qmaxdec) * cri;
                    dmin = dmin > cri ? vp.qmindec * dmin + (1.0 - vp.qmindec) * cri : vp.qmininc * dmin + (1.0 - vp.qmininc) * cri;
                    const double dyn = dmax - dmin;
                    vad0 = (cri > dmin + (vp.dyn_perc / 100.0) * dyn) && (dyn > vp.dyn_min);
                }
            }
            if (vp.cri != 0 && !(vad0 && t > vp.cep_init))  // background update (src/vad/vad.cc:288-294)
                c0r = vp.cep_p * c0r + (1.0 - vp.cep_p) * cil;
            push(vad0);
            if (t >= h) {
                if (lane == 0) vad_out[r0 + nout] = ((double)nsum / (double)order >= 0.5) ? '1' : '0';
                nout++;
            }
        }
    }
    for (int k = 0; k < h && nout < T; k++) {  // flush: zeros pushed (src/vad/vad.h:156-175)
        push(0);
        if (lane == 0) vad_out[r0 + nout] = ((double)nsum / (double)order >= 0.5) ? '1' : '0';
        nout++;
    }
}

// TRAP-DCT (src/fea/fea_trap.cc:53-127): out[t][b*ndct+k] = sum_j G[k][j] * logmel[clamp(t-half+j)][b]
// with mean removal, Hamming and REDFT10 folded into G on the host.  Unlike the banded filter bank this IS a dense
// contraction (ndct x traplen per band and frame), so it runs on the matrix cores: v_mfma_f32_16x16x4_f32 (exact
// fp32 FMA chain), D[k][t] += G[k][4s..4s+3] * X[4s..4s+3][t] with the Toeplitz operand X[j][t] = x[t+j-half][b]
// read straight from an LDS tile of log-mel frames.  Rows of G sum to zero, so each column is offset by its centre
// value first (keeps the fp32 accumulation small).
// One workgroup = 64 output frames of one utterance (4 waves x 16 frames), all bands.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NRB, int NSM>  // row blocks of 16 DCT coefficients (ndct <= 16*NRB); NSM >= ceil(traplen/4) tap groups
__global__ __launch_bounds__(256) void trapdct_mfma_kernel(const float *__restrict__ logmel, float *__restrict__ rows,
                                                           const float *__restrict__ G, const int4 *__restrict__ utt_info,
                                                           const int *__restrict__ chunk_tab, int B, int traplen, int ndct, int D) {
    extern __shared__ float tile[];  // [64 + 4*nsteps][Bs]
    const int u = chunk_tab[blockIdx.x * 2], tc = chunk_tab[blockIdx.x * 2 + 1];
    const int4 ui = utt_info[u];
    const int64_t r0 = ((int64_t)ui.y << 32) | (uint32_t)ui.x;
    const int T = ui.z;
    const int half = (traplen - 1) / 2, nsteps = (traplen + 3) / 4;
    const int Bs = B | 1, nfr = 64 + 4 * (NSM <= 32 ? NSM : nsteps);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // log-mel tile with the first / last frame replicated beyond the utterance (src/fea/fea_trap.cc:64-70,111-127)
    for (int e = tid; e < nfr * B; e += 256) {
        const int f = e / B, b = e - f * B;
        int t = tc - half + f;
        t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
        tile[f * Bs + b] = logmel[(r0 + t) * B + b];
    }
    // this lane's slice of G: A[i = lane&15][k = lane>>4] of every 16x4 block
    const int ai = lane & 15, ak = lane >> 4;
    float areg[NRB][NSM];
#pragma unroll
    for (int rb = 0; rb < NRB; rb++)
#pragma unroll
        for (int s_ = 0; s_ < NSM; s_++) {
            const int k = rb * 16 + ai, j = 4 * s_ + ak;
            areg[rb][s_] = (s_ < nsteps && k < ndct && j < traplen) ? G[k * traplen + j] : 0.f;
        }
    __syncthreads();
    const int tl = wave * 16 + (lane & 15);  // local output frame of this lane's column
    const int t_out = tc + tl;
    for (int b = 0; b < B; b++) {
        const float xc = tile[(tl + half) * Bs + b];
        f32x4 acc[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; rb++) acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < NSM; s_++) {
            if (NSM <= 32 || s_ < nsteps) {  // exact instantiations run unguarded (A is zero beyond traplen)
                const float bv = tile[(tl + 4 * s_ + ak) * Bs + b] - xc;
#pragma unroll
                for (int rb = 0; rb < NRB; rb++) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[rb][s_], bv, acc[rb], 0, 0, 0);
            }
        }
        if (t_out < T && ndct == 16 && NRB == 1 && (D & 3) == 0) {
            // C/D layout: this lane holds coefficients 4*(lane>>4)..+3 of frame column lane&15: one 16-byte store
            *reinterpret_cast<f32x4 *>(rows + (r0 + t_out) * D + b * 16 + ak * 4) = acc[0];
        } else if (t_out < T) {
            float *o = rows + (r0 + t_out) * D + b * ndct;
#pragma unroll
            for (int rb = 0; rb < NRB; rb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = rb * 16 + ak * 4 + r;  // C/D layout: row = (lane>>4)*4 + reg, col = lane&15
                    if (k < ndct) o[k] = acc[rb][r];
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local std::string g_create_error;

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    void upload(const std::vector<T> &h) {
        release();
        n = h.size();
        if (!n) return;
        HIP_TRY(hipMalloc(&p, n * sizeof(T)));
        HIP_TRY(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    }
    void alloc(size_t count) {
        release();
        n = count;
        if (n) HIP_TRY(hipMalloc(&p, n * sizeof(T)));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Row N1: the deltaFEA chain (src/fea/fea_delta.cc, wired by src/io/batch.cc:122-130,172-192,251-291) and the
// writers' block layout (src/io/out.cc:188-201), as one pass over the base rows of a 64-frame chunk.
//
// The reference streams frames through a ring per stage; what that ring computes is (tests/test_oracle_delta.py
// holds the oracle's replay of the ring against exactly these formulas):
//   stage k, window w:   y[t] = sum_{i=1..w} i * (x[min(t+i,T-1)] - x[max(t-i,0)]) / (2 sum i^2),
//                        except that a stage with w == 1 emits y[T-1] = 0 (its flush writes the last frame twice);
//   x of stage k+1 is y of stage k (clamping applies to the frame index of y, not to a virtual y beyond the edge);
//   row[t] = [x | y1 | y2 | y3] in the base block order (c1..cN, c0), then E of frame min(t + sum w, T-1) - the
//            writers read E through a pointer, so it belongs to the newest frame fed in;
//   -fea_trap (stack): X[i*L+j] = fvec_i of frame clamp(t-w+j) with L = 2w+1, fvec order (c0, c1..cN); the first
//            row uses frames (0 x w, 1, 1, 2..w) and, for w == 1, the last row uses frame T-1 three times; rows 0 and
//            T-w..T-1 then get X[0..fea_c) overwritten by the centre frame's fvec (fea_delta.cc:88-90,196-198).
// HBM-bound: reads Dbase floats (+ halo) and writes D floats per frame, both fully coalesced (a chunk's rows are
// contiguous); LDS holds the levels of the chain for 64 + 2*halo frames.
struct PostParams {
    int fea_c, Dbase, D, order, stack, has_e;
    int w[3];
    float inv_den[3];
};

__global__ __launch_bounds__(256) void post_kernel(const float *__restrict__ base, float *__restrict__ rows,
                                                   const int4 *__restrict__ utt_info, const int *__restrict__ chunks,
                                                   const int n_chunks, const PostParams pp) {
    extern __shared__ float psm[];
    constexpr int PF = 12;  // prefetch registers per thread; the host keeps R * Dbase <= 256 * PF
    const int fc = pp.fea_c, Db = pp.Dbase, D = pp.D;
    const int H = pp.stack ? pp.w[0] : pp.w[0] + (pp.order > 1 ? pp.w[1] : 0) + (pp.order > 2 ? pp.w[2] : 0);
    const int R = 64 + 2 * H;
    float *x0 = psm;                        // [R][Db]   base rows (E column included)
    float *lv = psm + (size_t)R * Db;       // levels 1..order: [R][fc] each
    // e / d for 0 <= e < 2^16, 1 <= d < 2^10 through the float reciprocal: (e + 0.5) / d stays at least 0.5/d away from
    // an integer, far more than the rounding error of the product, so the truncation is exact
    auto fdiv = [](int e, float inv) { return (int)(((float)e + 0.5f) * inv); };
    const float invD = 1.0f / (float)D, invfc = 1.0f / (float)fc;
    // element e = threadIdx.x + 256 q of a [frames][D] (or [frames][fc]) image: (frame, column) advance by a fixed
    // (quotient, remainder) per step, so the loops below carry them instead of dividing
    const int tt_first = fdiv(threadIdx.x, invD), k_first = threadIdx.x - tt_first * D;
    const int dqD = fdiv(256, invD), drD = 256 - dqD * D;
    const int ff_first = fdiv(threadIdx.x, invfc), cc_first = threadIdx.x - ff_first * fc;
    const int dqF = fdiv(256, invfc), drF = 256 - dqF * fc;

    struct Meta { long long ro; int T, t0; };
    auto meta = [&](int c) {
        const int u = chunks[2 * c];
        const int4 ui = utt_info[u];
        Meta m;
        m.ro = (long long)(unsigned)ui.x | ((long long)ui.y << 32);
        m.T = ui.z;
        m.t0 = chunks[2 * c + 1];
        return m;
    };
    // the contiguous run of base rows a chunk touches, into registers (the loads stay in flight while the previous
    // chunk is being computed and written: a workgroup walks chunks blockIdx.x, +gridDim.x, ...)
    auto issue = [&](const Meta &m, float (&r)[PF]) {
        const int flo = max(m.t0 - H, 0), fhi = min(m.t0 + 63 + H, m.T - 1);
        const float *src = base + (m.ro + flo) * Db;
        const int n = (fhi - flo + 1) * Db;
#pragma unroll
        for (int q = 0; q < PF; q++) {
            const int e = threadIdx.x + 256 * q;
            r[q] = e < n ? src[e] : 0.f;
        }
    };

    int c = blockIdx.x;
    if (c >= n_chunks) return;
    Meta m = meta(c);
    float r[PF];
    issue(m, r);
    while (true) {
        const int t0 = m.t0, T = m.T, tlo = t0 - H;
        const long long ro = m.ro;
        const int nout = min(64, T - t0);
        {
            const int flo = max(tlo, 0), fhi = min(t0 + 63 + H, T - 1);
            float *dst = x0 + (size_t)(flo - tlo) * Db;
            const int n = (fhi - flo + 1) * Db;
#pragma unroll
            for (int q = 0; q < PF; q++) {
                const int e = threadIdx.x + 256 * q;
                if (e < n) dst[e] = r[q];
            }
        }
        __syncthreads();
        const int cn = c + gridDim.x;
        const bool more = cn < n_chunks;
        Meta mn = m;
        if (more) {
            mn = meta(cn);
            issue(mn, r);
        }
        auto rowof = [&](int f) { return min(max(f, 0), T - 1) - tlo; };
        if (pp.stack) {
            const int w = pp.w[0], L = 2 * w + 1, xs = fc * L;
            const float invL = 1.0f / (float)L;
            int tt = tt_first, k = k_first;
            for (int e = threadIdx.x; e < nout * D; e += 256, tt += dqD, k += drD) {
                if (k >= D) { k -= D; tt++; }
                const int t = t0 + tt;
                float v;
                if (k == xs) v = x0[(size_t)rowof(t + w) * Db + fc];  // E
                else {
                    int i, f;
                    if (k < fc && (t == 0 || t >= T - w)) { i = k; f = t; }
                    else {
                        i = fdiv(k, invL);
                        const int j = k - i * L;
                        if (t == 0) f = j < w ? 0 : max(1, j - w);
                        else if (w == 1 && t == T - 1) f = T - 1;
                        else f = t - w + j;
                    }
                    v = x0[(size_t)rowof(f) * Db + (i == 0 ? fc - 1 : i - 1)];
                }
                rows[(ro + t0) * D + e] = v;
            }
        } else {
            int hk = H;
            const float *prev = x0;
            int pstride = Db;
            for (int k = 0; k < pp.order; k++) {
                const int w = pp.w[k];
                hk -= w;  // halo this level still needs for the stages after it
                float *cur = lv + (size_t)k * R * fc;
                const int flo = max(t0 - hk, 0), fhi = min(t0 + 63 + hk, T - 1);
                const int n = (fhi - flo + 1) * fc;
                int ff = ff_first, cc = cc_first;
                for (int e = threadIdx.x; e < n; e += 256, ff += dqF, cc += drF) {
                    if (cc >= fc) { cc -= fc; ff++; }
                    const int f = flo + ff;
                    float acc = 0.f;
                    for (int i = 1; i <= w; i++)
                        acc += (float)i * (prev[(size_t)rowof(f + i) * pstride + cc] - prev[(size_t)rowof(f - i) * pstride + cc]);
                    acc *= pp.inv_den[k];
                    if (w == 1 && f == T - 1) acc = 0.f;
                    cur[(size_t)(f - tlo) * fc + cc] = acc;
                }
                __syncthreads();
                prev = cur;
                pstride = fc;
            }
            const int xs = fc * (pp.order + 1);
            int tt = tt_first, k = k_first;
            for (int e = threadIdx.x; e < nout * D; e += 256, tt += dqD, k += drD) {
                if (k >= D) { k -= D; tt++; }
                const int t = t0 + tt;
                float v;
                if (k == xs) v = x0[(size_t)rowof(t + H) * Db + fc];  // E
                else {
                    const int j = (k >= fc) + (k >= 2 * fc) + (k >= 3 * fc), cc = k - j * fc;
                    v = j == 0 ? x0[(size_t)(t - tlo) * Db + cc] : lv[((size_t)(j - 1) * R + (t - tlo)) * fc + cc];
                }
                rows[(ro + t0) * D + e] = v;
            }
        }
        if (!more) break;
        __syncthreads();  // every read of this chunk's LDS image is done before the next one is written
        c = cn;
        m = mn;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Row N2, CMS part: cms_POST (src/fea/post_impl.cc:159-240) - a running cepstral mean subtracted from the first
// fea_ncepcoefs+1 entries of the vector that is about to be written, state reset per file (src/io/batch.cc:388-392).
// In row terms: columns [0, ncols) of block 0.  Source = the front end's base rows (block 0 of a delta row is the
// base row), destination = the final rows; `copy_rest` also carries the remaining base columns (E) when no delta
// pass wrote them.  The reference keeps the mean in `float`; the arithmetic below rounds where it rounds.
//   exp:    m = fl32(fl32(m z) + F (1 - z));  out = F - m                       (sequential in t; one lane per column)
//   block:  t >= L-1: m = fl32(sum over ring slots x = 0..L-1 of F[newest frame == x mod L]) / L;  out = F - m
//           t <  L-1: out = F                                                   (64-frame chunks, LDS tile with L-1 halo)
struct CmsParams {
    int ncols, Dbase, D, copy_rest, L;
    float z, omz;
};

__global__ __launch_bounds__(64) void cms_exp_kernel(const float *__restrict__ base, float *__restrict__ rows,
                                                     const int4 *__restrict__ utt_info, int n_utt, const CmsParams cp) {
    const int u = blockIdx.x * 2 + (threadIdx.x >> 5), c = threadIdx.x & 31;
    if (u >= n_utt) return;
    const int4 ui = utt_info[u];
    const long long ro = (long long)(unsigned)ui.x | ((long long)ui.y << 32);
    const int T = ui.z;
    const float *src = base + ro * cp.Dbase;
    float *dst = rows + ro * cp.D;
    if (c < cp.ncols) {
        float m = 0.f;
        for (int t0 = 0; t0 < T; t0 += 8) {
            float f[8];
#pragma unroll
            for (int i = 0; i < 8; i++) f[i] = t0 + i < T ? src[(size_t)(t0 + i) * cp.Dbase + c] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (t0 + i < T) {
                    m = __fmaf_rn(f[i], cp.omz, __fmul_rn(m, cp.z));
                    dst[(size_t)(t0 + i) * cp.D + c] = f[i] - m;
                }
            }
        }
    } else if (cp.copy_rest && c < cp.Dbase) {
        for (int t = 0; t < T; t++) dst[(size_t)t * cp.D + c] = src[(size_t)t * cp.Dbase + c];
    }
}

__global__ __launch_bounds__(256) void cms_block_kernel(const float *__restrict__ base, float *__restrict__ rows,
                                                        const int4 *__restrict__ utt_info, const int *__restrict__ chunks,
                                                        const CmsParams cp) {
    extern __shared__ float csm[];  // [64 + L - 1][ncols]
    const int u = chunks[2 * blockIdx.x], t0 = chunks[2 * blockIdx.x + 1];
    const int4 ui = utt_info[u];
    const long long ro = (long long)(unsigned)ui.x | ((long long)ui.y << 32);
    const int T = ui.z, L = cp.L, nc = cp.ncols;
    const int nout = min(64, T - t0);
    const int flo = max(t0 - (L - 1), 0);
    const int nrow = t0 + nout - flo;
    for (int e = threadIdx.x; e < nrow * nc; e += 256) {
        const int r = e / nc, c = e - r * nc;
        csm[e] = base[(ro + flo + r) * cp.Dbase + c];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nout * nc; e += 256) {
        const int tt = e / nc, c = e - tt * nc, t = t0 + tt;
        const float f = csm[(t - flo) * nc + c];
        float m = 0.f;
        if (t >= L - 1) {
            // Ring slot x holds the newest frame congruent to x mod L, and the reference adds slots 0..L-1 in that
            // order: first the frames of the current ring cycle, t - t%L .. t, then the tail of the previous cycle,
            // t-L+1 .. t - t%L - 1.  Same order here, so the float sum rounds the same way.
            const int tm = t % L;
            const float *q = csm + (t - tm - flo) * nc + c;
            for (int i = 0; i <= tm; i++) m += q[i * nc];
            q = csm + (t - L + 1 - flo) * nc + c;
            const int n2 = L - 1 - tm;
            for (int i = 0; i < n2; i++) m += q[i * nc];
            m = m / (float)L;
        }
        rows[(ro + t) * cp.D + c] = f - m;
    }
    if (cp.copy_rest)
        for (int e = threadIdx.x; e < nout * (cp.Dbase - nc); e += 256) {
            const int tt = e / (cp.Dbase - nc), c = nc + e - tt * (cp.Dbase - nc);
            rows[(ro + t0 + tt) * cp.D + c] = base[(ro + t0 + tt) * cp.Dbase + c];
        }
}

// ---------------------------------------------------------------------------------------------------------------
// Row N2, CMVN part (src/fea/post_impl.cc:51-118).  One workgroup per 64-frame chunk; thread = statistic slot.
// HBM-bound: every row is read once per pass (coalesced: consecutive slots are consecutive columns but for the rotated
// c0 entries), partial sums in double, one fp64 atomic per (chunk, slot).
__global__ __launch_bounds__(256) void cmvn_accumulate_kernel(const float *__restrict__ rows, const int4 *__restrict__ utt_info,
                                                              const int *__restrict__ chunks, const int *__restrict__ spk_of_utt,
                                                              const int *__restrict__ col_of_slot, const double *__restrict__ mean,
                                                              double *__restrict__ acc, int cols, int D) {
    __shared__ double part[4][128];
    const int u = chunks[2 * blockIdx.x], t0 = chunks[2 * blockIdx.x + 1];
    const int4 ui = utt_info[u];
    const long long ro = (long long)(unsigned)ui.x | ((long long)ui.y << 32);
    const int n = min(64, ui.z - t0), spk = spk_of_utt[u];
    // thread = (row group rg of four, statistic slot): the 64 rows of a chunk are summed in four interleaved groups,
    // combined through LDS, then one fp64 atomic per (chunk, slot)
    const int rg = threadIdx.x >> 6, kl = threadIdx.x & 63;
    for (int k0 = 0; k0 < cols; k0 += 64) {
        const int k = k0 + kl;
        double sum = 0.0;
        if (k < cols) {
            const float *src = rows + (ro + t0) * D + col_of_slot[k];
            if (mean) {
                const double m = mean[(size_t)spk * cols + k];
                for (int t = rg; t < n; t += 4) {
                    const double dlt = (double)src[(size_t)t * D] - m;
                    sum += dlt * dlt;
                }
            } else {
                for (int t = rg; t < n; t += 4) sum += (double)src[(size_t)t * D];
            }
        }
        part[rg][kl] = sum;
        __syncthreads();
        if (rg == 0 && k < cols) atomicAdd(&acc[(size_t)spk * (cols + 1) + k], (part[0][kl] + part[1][kl]) + (part[2][kl] + part[3][kl]));
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(&acc[(size_t)spk * (cols + 1) + cols], (double)n);
}

__global__ __launch_bounds__(256) void cmvn_apply_kernel(float *__restrict__ rows, const int4 *__restrict__ utt_info,
                                                         const int *__restrict__ chunks, const int *__restrict__ spk_of_utt,
                                                         const int *__restrict__ slot_of_col, const double *__restrict__ mean,
                                                         const double *__restrict__ var, int cols, int D) {
    const int u = chunks[2 * blockIdx.x], t0 = chunks[2 * blockIdx.x + 1];
    const int4 ui = utt_info[u];
    const long long ro = (long long)(unsigned)ui.x | ((long long)ui.y << 32);
    const int n = min(64, ui.z - t0), spk = spk_of_utt[u];
    float *dst = rows + (ro + t0) * D;
    for (int e = threadIdx.x; e < n * D; e += 256) {
        const int t = e / D, c = e - t * D;
        const int k = slot_of_col[c];
        if (k >= 0) {
            const double m = mean[(size_t)spk * cols + k], v = var[(size_t)spk * cols + k];
            dst[e] = (float)(((double)dst[e] - m) / v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Row N3: sigOUT (src/io/out.cc:346-451) - enhanced speech from the post-NR magnitudes and the ORIGINAL phases.
//   synth_kernel   one wave per frame: X[k] = |Y[k]|/N * X0[k]/|X0[k]| (DC and Nyquist as positive reals, as the
//                  reference stores them before its sign fix-up, out.cc:416-419), Hermitian -> real by the packed
//                  half-size inverse FFT  Z[k] = (X[k] + X*[M-k]) + i e^{+2 pi i k/N} (X[k] - X*[M-k]),  z = IDFT_M(Z),
//                  y[2n] = Re z[n], y[2n+1] = Im z[n]  (radix-4 Stockham passes in LDS, a radix-2 tail when M = 128);
//                  the first `window` samples of y go to a per-frame scratch row.
//   ola_kernel     one thread per output sample: sum of the frames that cover it, in frame order as the ring of the
//                  reference accumulates them, floor(x / correction), +-32767 clip (out.cc:436-451); an utterance of
//                  T frames yields T*wshift + (window - wshift) samples (the tail is what close() writes).
// HBM-bound through the spectra scratch (12 B/bin in, 4 B/sample out); fusing the inverse transform into the front
// end is the obvious next step once this path matters.
struct SynthParams {
    int K, wfft, window, wshift;
    float inv_n;
    double corr;
};

__global__ __launch_bounds__(256) void synth_kernel(const float2 *__restrict__ xri, const float *__restrict__ pnr,
                                                    float *__restrict__ ybuf, long long total_frames, const SynthParams sp) {
    __shared__ float2 root[512];          // e^{+2 pi i m / 512}
    __shared__ float2 bufs[4][2][260];
    for (int m = threadIdx.x; m < 512; m += 256) {
        float sn, cs;
        sincospif((float)m / 256.0f, &sn, &cs);
        root[m] = make_float2(cs, sn);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int M = sp.wfft / 2;            // 256 or 128
    const int rs = 512 / sp.wfft;         // stride of the N-th roots in the table
    float2 *A = bufs[wave][0], *Bf = bufs[wave][1];
    for (long long f = (long long)blockIdx.x * 4 + wave; f < total_frames; f += (long long)gridDim.x * 4) {
        const float2 *xr = xri + f * sp.K;
        const float *pn = pnr + f * sp.K;
        for (int k = lane; k <= M; k += 64) {
            float2 v;
            if (k == 0 || k == M) v = make_float2(pn[k] * sp.inv_n, 0.f);
            else {
                const float2 x0 = xr[k];
                const float mag2 = x0.x * x0.x + x0.y * x0.y;
                const float sc = mag2 > 0.f ? pn[k] * sp.inv_n * rsqrtf(mag2) : 0.f;
                v = make_float2(x0.x * sc, x0.y * sc);
            }
            A[k] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int k = lane; k < M; k += 64) {
            const float2 a = A[k], b = A[M - k];
            const float2 sm = make_float2(a.x + b.x, a.y - b.y);      // X[k] + conj(X[M-k])
            const float2 df = make_float2(a.x - b.x, a.y + b.y);      // X[k] - conj(X[M-k])
            const float2 w = root[k * rs];
            // i * w * df
            const float2 t = make_float2(-(w.x * df.y + w.y * df.x), w.x * df.x - w.y * df.y);
            Bf[k] = make_float2(sm.x + t.x, sm.y + t.y);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float2 *src = Bf, *dst = A;
        int Ns = 1;
        const int mr = 512 / M;  // stride of the M-th roots in the table
        while (Ns * 4 <= M) {
            const int q4 = M / 4;
            for (int j = lane; j < q4; j += 64) {
                const int kk = j % Ns;
                const int tstep = kk * (M / (4 * Ns)) * mr;  // index of e^{2 pi i kk / (4 Ns)} in the table
                const float2 v0 = src[j];
                float2 v1 = src[j + q4], v2 = src[j + 2 * q4], v3 = src[j + 3 * q4];
                v1 = cmul(v1, root[(tstep) & 511]);
                v2 = cmul(v2, root[(2 * tstep) & 511]);
                v3 = cmul(v3, root[(3 * tstep) & 511]);
                // inverse radix-4 butterfly (W4 = +i)
                const float2 s02 = make_float2(v0.x + v2.x, v0.y + v2.y), d02 = make_float2(v0.x - v2.x, v0.y - v2.y);
                const float2 s13 = make_float2(v1.x + v3.x, v1.y + v3.y), d13 = make_float2(v1.x - v3.x, v1.y - v3.y);
                const int base = (j / Ns) * Ns * 4 + kk;
                dst[base] = make_float2(s02.x + s13.x, s02.y + s13.y);
                dst[base + Ns] = make_float2(d02.x - d13.y, d02.y + d13.x);      // d02 + i d13
                dst[base + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
                dst[base + 3 * Ns] = make_float2(d02.x + d13.y, d02.y - d13.x);  // d02 - i d13
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float2 *tmp = src; src = dst; dst = tmp;
            Ns *= 4;
        }
        if (Ns < M) {  // one radix-2 pass (M = 128)
            const int h = M / 2;
            for (int j = lane; j < h; j += 64) {
                const int kk = j % Ns;
                const float2 v0 = src[j];
                const float2 v1 = cmul(src[j + h], root[(kk * (M / (2 * Ns)) * mr) & 511]);
                const int base = (j / Ns) * Ns * 2 + kk;
                dst[base] = make_float2(v0.x + v1.x, v0.y + v1.y);
                dst[base + Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float2 *tmp = src; src = dst; dst = tmp;
        }
        float2 *yo = reinterpret_cast<float2 *>(ybuf + f * sp.window);  // window is even (checked on the host)
        for (int n = lane; 2 * n < sp.window; n += 64) yo[n] = src[n];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void ola_kernel(const float *__restrict__ ybuf, int16_t *__restrict__ out,
                                                  const int4 *__restrict__ utt_info, const long long *__restrict__ sample_off,
                                                  int n_utt, const SynthParams sp) {
    const int u = blockIdx.y;
    if (u >= n_utt) return;
    const int4 ui = utt_info[u];
    const long long ro = (long long)(unsigned)ui.x | ((long long)ui.y << 32);
    const int T = ui.z, w = sp.window, s = sp.wshift;
    const long long nout = (long long)T * s + (w - s);
    int16_t *o = out + sample_off[u];
    for (long long n = (long long)blockIdx.x * 256 + threadIdx.x; n < nout; n += (long long)gridDim.x * 256) {
        long long t0 = (n - w + s) / s;   // ceil((n - w + 1) / s) for n - w + 1 > 0
        if (n - w + 1 <= 0) t0 = 0;
        long long t1 = n / s;
        if (t1 > T - 1) t1 = T - 1;
        double acc = 0.0;
        for (long long t = t0; t <= t1; t++) acc += (double)ybuf[(ro + t) * w + (n - t * s)];
        const int value = (int)floor(acc / sp.corr);
        o[n] = fabsf((float)value) > 32767.f ? (value < 0 ? -32767 : 32767) : (int16_t)value;
    }
}

struct ctu_engine {
    std::unique_ptr<ctu::Design> design;
    int device = 0;
    int n_cu = 256;
    std::string err;
    int feat = FEAT_DCTC;
    int nz = 16;
    int mode = 0;  // 0: 512-point FFT, 1: 256-point FFT (two frames per complex transform)
    DevBuf<float> lanec, ftab, trapG;
    DevBuf<int> itab;
    int lift_off = 0, tab_floats = 0, ck_off = 0, cf_off = 0, NS = 0, CW = 4, ncoef_out = 0;
    size_t lds_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    bool in_signal_call = false;
    // CMVN (row N2): statistic slot <-> row column maps, and per-call scratch
    std::vector<int> col_of_slot, slot_of_col;
    DevBuf<int> d_col_of_slot, d_slot_of_col, d_spk;
    DevBuf<double> d_stat_a, d_stat_b;
    DevBuf<unsigned long long> stamps;
    bool do_vad = false;
    VadParams vp;
};

struct ctu_plan {
    ctu_engine *eng = nullptr;
    int n_utt = 0;
    std::vector<int64_t> nsamples, sample_off, row_off, frames;
    int64_t total_samples = 0, total_frames = 0;
    int n_tiles = 0;
    int grid = 0;               // workgroups of the front-end launch (tile chains are built for it)
    DevBuf<TileRec> tiles;
    DevBuf<int> wg_first;
    DevBuf<float2> xri;         // VAD scratch
    DevBuf<float> pnr;
    DevBuf<double> vad_ci;
    DevBuf<int64_t> d_row_off;
    // scratch between the kernels of one run; owned by the plan, so plans can run concurrently on different streams
    DevBuf<float> logmel;      // TRAP: log-mel rows [total_frames][B]
    DevBuf<float> base_rows;   // front-end rows ahead of the delta / stacking / CMS passes [total_frames][Dbase]
    DevBuf<float> ybuf;        // signal output: time-domain frames ahead of the overlap-add [total_frames][window]
    std::vector<int64_t> out_samples;   // signal output: samples written per utterance
    DevBuf<long long> d_sample_off;
    // TRAP
    DevBuf<int4> utt_info;
    DevBuf<int> trap_chunks;
    int n_trap_chunks = 0;
};

namespace {

void set_error(ctu_engine *e, const std::string &m) { e->err = m; }

// reasons a valid ctucopy configuration is outside the accelerated path
std::string unsupported_reason(const ctu::Design &d) {
    const ctu::Opts &o = d.o;
    if (d.signal_out) {  // row N3: IN -> NR -> sigOUT
        if (o.format_in == "htk") return "HTK feature input with signal output";
        if (o.fea_kind == "td-iir-mfcc") return "fea_kind outside the spectral path";
        if (o.dither != 0.) return "-dither != 0 makes outputs depend on file order (src/io/in.cc:205,454)";
        if (o.remove_dc1) return "-remove_dc1 mutates the sample history across frames (src/io/in.cc:343-350)";
        if (o.nr_mode != "none" && o.nr_mode != "exten") return "nr_mode hwss/fwss/2fwss seed their noise estimate from the previous file (src/nr/nr.cc:212-221)";
        if (o.rasta) return "-nr_rasta";
        if (o.do_vad()) return "VAD together with signal output";
        if (d.wfft != 512 && d.wfft != 256) return "FFT size other than 512 or 256";
        if (d.wfft == 512 && d.wshift % 2) return "odd frame shift with the 512-point transform (frame starts must be 4-byte aligned)";
        if (d.window % 2) return "odd window length with signal output";
        if (d.window < 32) return "window shorter than 32 samples";
        return "";
    }
    if (o.format_in == "htk") return "HTK feature input (-format_in htk) bypasses the spectral path";
    if (o.fea_kind == "td-iir-mfcc" || o.fea_kind == "none") return "fea_kind outside the spectral feature path";
    if (o.dither != 0.) return "-dither != 0 makes outputs depend on file order (src/io/in.cc:205,454)";
    if (o.remove_dc1) return "-remove_dc1 mutates the sample history across frames (src/io/in.cc:343-350)";
    if (o.nr_mode != "none" && o.nr_mode != "exten") return "nr_mode hwss/fwss/2fwss seed their noise estimate from the previous file (src/nr/nr.cc:212-221)";
    if (o.nr_when_afterFB) return "-nr_when afterFB";
    if (o.rasta) return "-nr_rasta";
    if (d.post_order > 0) {
        if (d.kind != ctu::FeaKind::Dctc && d.kind != ctu::FeaKind::Lpc) return "delta / stacking on non-cepstral kinds (the reference sizes the chain as fea_ncepcoefs+1, src/fea/fea_delta.cc:22-28)";
        if (!o.fea_c0) return "delta / stacking without -fea_c0 (the reference's writers leave slots of the row unwritten, src/io/out.cc:190-201)";
        if (o.do_vad()) return "VAD together with delta / stacking (the detector would run on delayed and on flushed frames)";
        int wsum = 0;
        for (int j = 0; j < d.post_order; j++) {
            if (d.post_w[j] > 16) return "delta / stacking window above 16 frames";
            wsum += d.post_w[j];
        }
        if (wsum > 24) return "delta windows adding up to more than 24 frames (LDS tile of the chain)";
    }
    if (o.stat_cmvn || o.apply_cmvn) {
        if (d.kind != ctu::FeaKind::Dctc && d.kind != ctu::FeaKind::Lpc) return "CMVN on non-cepstral kinds";
        if (!o.fea_c0) return "CMVN without -fea_c0 (c0 is part of the statistics but not of the written row)";
        if (d.post_stack) return "CMVN on stacked vectors";
        if (d.cms) return "CMVN together with CMS (the reference warns and lets CMVN win, src/io/opts.cc:262-264)";
        if (o.do_vad()) return "VAD together with CMVN";
    }
    if (d.cms) {
        if (d.kind != ctu::FeaKind::Dctc && d.kind != ctu::FeaKind::Lpc) return "CMS on non-cepstral kinds (the reference walks fea_ncepcoefs+1 entries whatever the vector holds, src/fea/post_impl.cc:203-240)";
        if (d.post_stack) return "CMS on stacked vectors";
        if (o.do_vad()) return "VAD together with CMS";
        if (d.cms == 2 && (o.length_b < 1 || o.length_b > 512)) return "block CMS window outside 1..512 frames";
        if (d.cms_cols > 32) return "more than 32 CMS columns";
        if (d.cms == 2 && (size_t)(64 + o.length_b - 1) * d.cms_cols * sizeof(float) > 64 * 1024) return "block CMS tile above 64 KiB of LDS";
    }
    if (o.fea_E && d.kind == ctu::FeaKind::TrapDct) return "-fea_E with trapdct (the energy lags the features by 50 frames in the reference)";
    if (o.do_vad()) {
        if (d.kind == ctu::FeaKind::TrapDct) return "VAD together with trapdct";
        if (o.vad_cri_mode != "energy" && o.vad_cri_mode != "cepdist") return "";  // rejected with the reference's text at create
        if (o.vad_cri_mode == "cepdist") {
            if (o.vad_cepdist_mode == "in") return "-vad_cepdist_mode in (HTK feature input)";
            if (o.vad_cepdist_mode == "fea" && d.kind != ctu::FeaKind::Dctc && d.kind != ctu::FeaKind::Lpc) return "-vad_cepdist_mode fea on non-cepstral features";
            const int nc = o.vad_cepdist_mode == "lpc" ? o.vad_lpc_coefs : d.nfea;
            if (nc > 32 || nc < 2) return "more than 32 (or fewer than 2) VAD cepstral coefficients";
        }
        if (o.vad_filter_order > 31) return "VAD filter order above 31";
    }
    if (d.wfft != 512 && d.wfft != 256) return "FFT size other than 512 or 256";
    if (d.wfft == 512 && d.wshift % 2) return "odd frame shift with the 512-point transform (frame starts must be 4-byte aligned)";
    if (d.window < 32) return "window shorter than 32 samples";
    if (d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa) {
        if (o.fea_lporder > MAX_LP || o.fea_ncepcoefs > MAX_LP) return "LP order / cepstral order above the in-register limit";
    }
    if (d.kind == ctu::FeaKind::Dctc && d.nfea > MAXC) return "more cepstral coefficients than the kernel accumulates";
    if (d.B > 512) return "more than 512 filter bank channels";
    if (d.kind == ctu::FeaKind::TrapDct && o.fea_trapdct_ndct > 32) return "more than 32 TRAP DCT coefficients";
    if (d.kind == ctu::FeaKind::TrapDct && o.fea_trapdct_traplen > 255) return "TRAP longer than 255 frames";
    if (d.kind == ctu::FeaKind::TrapDct && (size_t)(64 + 256) * (d.B | 1) * 4 > 64 * 1024) return "too many bands for the TRAP tile";
    return "";
}

// Host-side image of the phase-2 LDS tables (see KParams for the layout).
struct Phase2Tables {
    std::vector<float> ft;   // LDS image followed by the lifter
    std::vector<int> it;     // slot_chunk[NS+1] | row_slot[nfea]
    std::vector<int> cells, slot_chunk;
    int lift_off = 0, tab_floats = 0, ck_off = 0, cf_off = 0, NS = 0, CW = 4, ncoef_out = 0;
};

void build_phase2(const ctu::Design &d, Phase2Tables &t) {
    // ---- phase-2 tables.  Bands are dealt to (slot, group) cells: sorted by width, eight per slot, so that the
    // eight lanes of a frame walk bands of similar width in lock step.  A band is cut into 4-bin chunks whose
    // bin range stays inside [0,K); weights outside the band's own [first,last] are zero.
    const int B = d.B, K = d.K;
    std::vector<int> order(B);
    for (int b = 0; b < B; b++) order[b] = b;
    auto width = [&](int b) { return d.fb_last[b] - d.fb_first[b] + 1; };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return width(x) > width(y); });
    const int NS = (B + 7) / 8;
    int ncoef = 0;
    const std::vector<double> *coef_tab = nullptr;
    if (d.kind == ctu::FeaKind::Dctc) { coef_tab = &d.dct; ncoef = d.nfea; }
    else if (d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa) { coef_tab = &d.idft; ncoef = d.o.fea_lporder + 1; }
    if (ncoef > MAXC) throw std::runtime_error("more cepstral / LP coefficients than the kernel accumulates");
    const int CW = ncoef <= 16 ? 16 : MAXC;  // the kernel has straight-line code for these two widths
    std::vector<int> slot_chunk(NS + 1, 0);
    std::vector<float> cw;                  // chunk weights [NC][8][4]
    std::vector<int> cell(NS * 8 * 2, 0);   // {first bin of the chunk run, band index or -1}
    const int CWS = CW + 4;  // row stride: 20 or 28 floats = 5 or 7 16-byte units, distinct mod 16 over the 8 groups
    std::vector<float> cf((size_t)NS * 8 * CWS, 0.f);
    // DCTC: coefficient row r of a cell is the value written to output slot r (c1..cN, then c0)
    std::vector<int> coef_of_slot;
    if (d.kind == ctu::FeaKind::Dctc) {
        coef_of_slot.assign(d.nfea, -1);
        int nout = 0;
        for (int i = 0; i < d.nfea; i++)
            if (d.row_slot[i] >= 0) {
                coef_of_slot[d.row_slot[i]] = i;
                nout = std::max(nout, d.row_slot[i] + 1);
            }
        t.ncoef_out = nout;
    }
    auto chunks_of = [&](int b) {  // chunks of 4 bins from the aligned start of the band to its last bin
        const int k0 = d.fb_first[b] & ~3;
        return (d.fb_last[b] - k0) / 4 + 1;
    };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return chunks_of(x) > chunks_of(y); });
    for (int sl = 0; sl < NS; sl++) {
        int nch = 0;
        for (int g = 0; g < 8 && sl * 8 + g < B; g++) nch = std::max(nch, chunks_of(order[sl * 8 + g]));
        if (4 * nch > PSTRIDE) throw std::runtime_error("filter band wider than the spectrum");
        slot_chunk[sl] = (int)cw.size() / 32;
        // Phase 2 reads P with one ds_read_b128 per lane; the 16-byte unit a lane touches is
        // (frame + kstart/4 + chunk) mod 16 (rows are 65 units apart).  A b128 wave access is served in four groups of
        // 16 lanes = {frame f: groups 0-3, f+1: groups 4-7, f+2: groups 4-7, f+3: groups 0-3}; collisions depend only
        // on a_g = kstart_g/4.  Search the assignment of this slot's bands to groups (and up to `slack` leading
        // zero chunks) for the fewest colliding lane pairs.
        std::vector<int> perm(8), best_perm(8), lead(8, 0), best_lead(8, 0);
        for (int g = 0; g < 8; g++) perm[g] = best_perm[g] = sl * 8 + g < B ? order[sl * 8 + g] : -1;
        auto a_of = [&](int b, int ld) { return b < 0 ? -1000 : (std::min(d.fb_first[b] & ~3, PSTRIDE - 4 * nch) / 4 - ld); };
        auto cost = [&](const std::vector<int> &pm, const std::vector<int> &ld) {
            int c = 0, a[8];
            for (int g = 0; g < 8; g++) a[g] = a_of(pm[g], ld[g]);
            const int fo[4] = {0, 1, 2, 3}, lo[4] = {1, 0, 0, 1};  // frame offset, uses groups 0-3 (1) or 4-7 (0)
            int unit[16], n = 0;
            for (int q = 0; q < 4; q++)
                for (int g = lo[q] ? 0 : 4; g < (lo[q] ? 4 : 8); g++) unit[n++] = a[g] < -500 ? -1 - n : ((a[g] + fo[q]) % 16 + 16) % 16;
            for (int i = 0; i < 16; i++)
                for (int j = i + 1; j < 16; j++) c += (unit[i] >= 0 && unit[i] == unit[j]);
            return c;
        };
        {
            int best = cost(perm, lead);
            uint32_t rng = 12345u + sl;
            for (int it = 0; it < 4000 && best > 0; it++) {
                std::vector<int> pm = best_perm, ld = best_lead;
                rng = rng * 1664525u + 1013904223u;
                const int i = (rng >> 8) % 8, j = (rng >> 16) % 8;
                std::swap(pm[i], pm[j]);
                std::swap(ld[i], ld[j]);
                rng = rng * 1664525u + 1013904223u;
                const int g = (rng >> 8) % 8;
                if (pm[g] >= 0) {
                    // leading zero chunks are allowed while the run still starts at >= 0 and ends beyond the last bin
                    const int k0 = std::min(d.fb_first[pm[g]] & ~3, PSTRIDE - 4 * nch);
                    const int mx = std::min({3, k0 / 4, (k0 + 4 * nch - (d.fb_last[pm[g]] + 1)) / 4});
                    ld[g] = mx > 0 ? (int)((rng >> 20) % (mx + 1)) : 0;
                }
                const int c = cost(pm, ld);
                if (c <= best) {
                    best = c;
                    best_perm = pm;
                    best_lead = ld;
                }
            }
        }
        std::vector<int> kstart(8, 0);
        for (int g = 0; g < 8; g++) {
            cell[(sl * 8 + g) * 2 + 1] = -1;
            if (best_perm[g] < 0) continue;
            const int b = best_perm[g];
            // aligned start; the run of nch chunks must end inside the row's PSTRIDE floats (bins >= K get weight 0
            // but are read, so the kernel keeps them finite: it zeroes the row padding once per workgroup)
            kstart[g] = std::min(d.fb_first[b] & ~3, PSTRIDE - 4 * nch) - 4 * best_lead[g];
            cell[(sl * 8 + g) * 2] = kstart[g];
            cell[(sl * 8 + g) * 2 + 1] = b;
            for (int i = 0; i < ncoef; i++) {
                const int src = (d.kind == ctu::FeaKind::Dctc) ? coef_of_slot[i] : i;
                if (src >= 0) cf[((size_t)sl * 8 + g) * CWS + i] = (float)(*coef_tab)[(size_t)src * B + b];
            }
        }
        for (int ch = 0; ch < nch; ch++)
            for (int g = 0; g < 8; g++)
                for (int i = 0; i < 4; i++) {
                    float w = 0.f;
                    if (best_perm[g] >= 0) {
                        const int b = best_perm[g], k = kstart[g] + 4 * ch + i;
                        if (k >= d.fb_first[b] && k <= d.fb_last[b]) w = (float)d.fb[b][k];
                    }
                    cw.push_back(w);
                }
    }
    slot_chunk[NS] = (int)cw.size() / 32;
    std::vector<float> ft(cw);
    t.cells = cell;
    t.slot_chunk = slot_chunk;
    auto push_ints = [&](const std::vector<int> &v) {
        for (int x : v) {
            float f;
            std::memcpy(&f, &x, 4);
            ft.push_back(f);
        }
        while (ft.size() & 3) ft.push_back(0.f);
    };
    t.ck_off = (int)ft.size();
    push_ints(cell);
    t.cf_off = (int)ft.size();
    ft.insert(ft.end(), cf.begin(), cf.end());
    while (ft.size() & 3) ft.push_back(0.f);
    t.tab_floats = (int)ft.size();
    t.NS = NS;
    t.CW = CW;
    t.lift_off = (int)ft.size();
    for (double v : d.lifter) ft.push_back((float)v);
    ft.push_back(0.f);
    t.ft = ft;
    std::vector<int> it(slot_chunk);
    it.insert(it.end(), d.row_slot.begin(), d.row_slot.end());
    t.it = it;
}

// Rebuilds every band's dense weight row from the chunk tables and returns the largest deviation from the
// float-rounded filter bank (0 when the tables are consistent).
double check_phase2(const ctu::Design &d, const Phase2Tables &t) {
    double worst = 0;
    std::vector<int> seen(d.B, 0);
    for (int sl = 0; sl < t.NS; sl++)
        for (int g = 0; g < 8; g++) {
            const int kstart = t.cells[(sl * 8 + g) * 2], b = t.cells[(sl * 8 + g) * 2 + 1];
            if (b < 0) continue;
            seen[b]++;
            std::vector<double> row(PSTRIDE, 0.0);
            for (int ch = t.slot_chunk[sl]; ch < t.slot_chunk[sl + 1]; ch++)
                for (int i = 0; i < 4; i++) {
                    const int k = kstart + 4 * (ch - t.slot_chunk[sl]) + i;
                    if (k < 0 || k >= PSTRIDE) return 1e30;
                    row[k] += t.ft[(size_t)(ch * 8 + g) * 4 + i];
                }
            for (int k = 0; k < PSTRIDE; k++) {
                const double want = (k < d.K && k >= d.fb_first[b] && k <= d.fb_last[b]) ? (double)(float)d.fb[b][k] : 0.0;
                if (std::fabs(row[k] - want) > 0 && getenv("CTU_P2_DEBUG")) fprintf(stderr, "sl %d g %d band %d k %d kstart %d first %d last %d row %g want %g nch %d\n", sl, g, b, k, kstart, d.fb_first[b], d.fb_last[b], row[k], want, t.slot_chunk[sl+1]-t.slot_chunk[sl]);
                worst = std::max(worst, std::fabs(row[k] - want));
            }
        }
    for (int b = 0; b < d.B; b++)
        if (seen[b] != 1) return 1e30;
    return worst;
}

void build_tables(ctu_engine *e) {
    const ctu::Design &d = *e->design;
    const double pi = 3.14159265358979323846;
    // ---- per-lane constant records (see LC_* above)
    const bool mode1 = d.wfft == 256;
    e->mode = mode1 ? 1 : 0;
    std::vector<float> lc(16 * LANEC, 0.f);
    for (int l = 0; l < 16; l++) {
        float *r = lc.data() + l * LANEC;
        for (int j = 0; j < 16; j++)
            for (int h = 0; h < 2; h++) {
                // MODE 0: lane l holds samples 32j+2l, +1 of row j; MODE 1: sample 16j+l (second slot unused)
                const int i = mode1 ? (h ? d.window : 16 * j + l) : 32 * j + 2 * l + h;
                r[LC_WIN + 2 * j + h] = i < d.window ? (float)d.hamming[i] : 0.f;
                r[LC_MASK + 2 * j + h] = i < d.window ? 1.f : 0.f;
            }
        for (int k1 = 1; k1 < 16; k1++) {
            const double a = -2 * pi * (double)(k1 * l) / 256.0;
            r[LC_TW + 2 * (k1 - 1)] = (float)std::cos(a);
            r[LC_TW + 2 * (k1 - 1) + 1] = (float)std::sin(a);
        }
        for (int k2 = 0; k2 < 8; k2++) {
            const double a = -2 * pi * (double)(l + 16 * k2) / 512.0;
            r[LC_UT + 2 * k2] = (float)std::cos(a);
            r[LC_UT + 2 * k2 + 1] = (float)std::sin(a);
        }
    }
    e->lanec.upload(lc);
    if (d.signal_out) {  // nothing is projected: empty table area
        e->ncoef_out = 0; e->ck_off = 0; e->cf_off = 0; e->tab_floats = 0; e->NS = 0; e->CW = 16; e->lift_off = 0;
        e->ftab.upload(std::vector<float>(4, 0.f));
        e->itab.upload(std::vector<int>(4, 0));
        e->lds_bytes = ((size_t)TILE * PSTRIDE + LTW_FLOATS) * sizeof(float);
        e->feat = FEAT_BANDS;
        e->nz = e->mode ? (d.window + 15) / 16 : (d.window + 31) / 32;
        return;
    }
    Phase2Tables t;
    build_phase2(d, t);
    if (check_phase2(d, t) != 0.0) throw std::runtime_error("internal: phase-2 chunk tables do not reproduce the filter bank");
    e->ncoef_out = t.ncoef_out;
    e->ck_off = t.ck_off;
    e->cf_off = t.cf_off;
    e->tab_floats = t.tab_floats;
    e->NS = t.NS;
    e->CW = t.CW;
    e->lift_off = t.lift_off;
    e->ftab.upload(t.ft);
    e->itab.upload(t.it);
    e->lds_bytes = ((size_t)TILE * PSTRIDE + e->tab_floats + LTW_FLOATS) * sizeof(float);
    if (e->lds_bytes > 160 * 1024) throw std::runtime_error("configuration needs more than 160 KiB of LDS");
    if (d.kind == ctu::FeaKind::TrapDct) {
        std::vector<float> g(d.trap.begin(), d.trap.end());
        e->trapG.upload(g);
    }
    switch (d.kind) {
        case ctu::FeaKind::Spec:
        case ctu::FeaKind::LogSpec:
        case ctu::FeaKind::TrapDct: e->feat = FEAT_BANDS; break;
        case ctu::FeaKind::Dctc: e->feat = FEAT_DCTC; break;
        case ctu::FeaKind::Lpc:
        case ctu::FeaKind::Lpa: e->feat = FEAT_LP; break;
        case ctu::FeaKind::None: e->feat = FEAT_BANDS; break;  // not reached: the signal path returns above
    }
    e->nz = e->mode ? (d.window + 15) / 16 : (d.window + 31) / 32;  // rows of samples per lane that can be non-zero
}

template <int NZ, int MODE, bool VX, bool GEN>
void launch_nz(int feat, dim3 grid, hipStream_t s, const KParams &kp, size_t shm) {
#define LAUNCH(F, NCW)                                                                                 \
    {                                                                                                  \
        static bool attr_set = false;                                                                  \
        if (!attr_set) {                                                                               \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&frontend_kernel<NZ, F, MODE, VX, NCW, GEN>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));     \
            attr_set = true;                                                                           \
        }                                                                                              \
        hipLaunchKernelGGL((frontend_kernel<NZ, F, MODE, VX, NCW, GEN>), grid, dim3(WG), shm, s, kp);   \
    }
    const bool wide = kp.CW != 16;  // coefficient rows of MAXC entries (more than 16 cepstra / LP lags)
    if (feat == FEAT_BANDS) LAUNCH(FEAT_BANDS, 16)
    else if (feat == FEAT_DCTC && !wide) LAUNCH(FEAT_DCTC, 16)
    else if (feat == FEAT_DCTC) LAUNCH(FEAT_DCTC, MAXC)
    else if (!wide) LAUNCH(FEAT_LP, 16)
    else LAUNCH(FEAT_LP, MAXC)
#undef LAUNCH
}

template <int NZ, int MODE>
void launch_vx(bool vx, int feat, dim3 grid, hipStream_t s, const KParams &kp, size_t shm) {
    // the plain chain gets its own instantiation (see GEN); everything else, and every VAD-export run, is generic
    const bool plain = !vx && kp.e_mode == 0 && !kp.fb_inld && !kp.nr_exten && kp.fb_power && kp.remove_dc && !kp.dbg && !kp.skip_phase2;
    if (vx) launch_nz<NZ, MODE, true, true>(feat, grid, s, kp, shm);
    else if (plain) launch_nz<NZ, MODE, false, false>(feat, grid, s, kp, shm);
    else launch_nz<NZ, MODE, false, true>(feat, grid, s, kp, shm);
}

std::vector<std::string> to_args(int argc, const char *const *argv) {
    std::vector<std::string> a;
    for (int i = 0; i < argc; i++) a.emplace_back(argv[i] ? argv[i] : "");
    return a;
}

void fill_dims(const ctu::Design &d, ctu_dims *out) {
    out->fs = d.o.fs;
    out->window = d.window;
    out->wshift = d.wshift;
    out->wfft = d.wfft;
    out->nbins = d.K;
    out->nbands = d.B;
    out->row_floats = d.D;
    out->htk_kind = d.htk_kind;
    out->htk_period = d.period;
    out->has_vad = d.o.do_vad() ? 1 : 0;
    out->swap_out = d.o.swap_out ? 1 : 0;
    out->pcm_align = PCM_ALIGN;
    out->signal_out = d.signal_out ? 1 : 0;
}

}  // namespace

extern "C" {

const char *ctu_create_error(void) { return g_create_error.c_str(); }

int ctu_config_dims(int argc, const char *const *argv, ctu_dims *out) {
    try {
        ctu::Opts o = ctu::Opts::from_args(to_args(argc, argv));
        ctu::Design d(o);
        fill_dims(d, out);
        return CTU_OK;
    } catch (const std::exception &ex) {
        g_create_error = ex.what();
        return CTU_ERR_OPTS;
    }
}

int64_t ctu_config_table(int argc, const char *const *argv, const char *name, double *out, int64_t cap) {
    try {
        ctu::Opts o = ctu::Opts::from_args(to_args(argc, argv));
        ctu::Design d(o);
        std::vector<double> v;
        const std::string n = name ? name : "";
        if (n == "hamming") v = d.hamming;
        else if (n == "fbank") for (const auto &row : d.fb) v.insert(v.end(), row.begin(), row.end());
        else if (n == "fb_first") v.assign(d.fb_first.begin(), d.fb_first.end());
        else if (n == "fb_last") v.assign(d.fb_last.begin(), d.fb_last.end());
        else if (n == "dct") v = d.dct;
        else if (n == "idft") v = d.idft;
        else if (n == "trap") v = d.trap;
        else if (n == "lifter") v = d.lifter;
        else if (n == "phase2_check") {
            Phase2Tables t;
            build_phase2(d, t);
            v = {check_phase2(d, t), (double)t.slot_chunk.back(), (double)t.NS};
        }
        else {
            g_create_error = "unknown table name";
            return CTU_ERR_INPUT;
        }
        for (int64_t i = 0; i < (int64_t)v.size() && i < cap; i++) out[i] = v[i];
        return (int64_t)v.size();
    } catch (const std::exception &ex) {
        g_create_error = ex.what();
        return CTU_ERR_OPTS;
    }
}

int ctu_engine_create(int argc, const char *const *argv, int device, ctu_engine **out) {
    if (!out) return CTU_ERR_INPUT;
    *out = nullptr;
    std::unique_ptr<ctu_engine> e(new ctu_engine);
    try {
        ctu::Opts o = ctu::Opts::from_args(to_args(argc, argv));
        e->design.reset(new ctu::Design(o));
    } catch (const std::exception &ex) {
        g_create_error = ex.what();
        return CTU_ERR_OPTS;
    }
    {
        const ctu::Opts &o = e->design->o;
        if (o.do_vad()) {  // constructor checks of src/vad/vad.cc:639-660, :149-195 and src/vad/vad.h:96-99
            const char *bad = nullptr;
            if (o.vad_cri_mode != "energy" && o.vad_cri_mode != "cepdist") bad = "VAD: unknown vad_cri_mode!";
            else if (o.vad_thr_mode != "absolute" && o.vad_thr_mode != "perc" && o.vad_thr_mode != "adapt" && o.vad_thr_mode != "dyn") bad = "VAD: unknown vad_thr_mode!";
            else if (o.vad_cri_mode == "cepdist" && o.vad_cepdist_mode != "lpc" && o.vad_cepdist_mode != "fea" && o.vad_cepdist_mode != "in") bad = "VADcri_cepdist: unknown vad_cepdist_mode!";
            else if (o.vad_cri_mode == "cepdist" && o.vad_cepdist_mode == "lpc" && !o.phase_needed) bad = "VADcri_cepdist: cannot perform iFFT!";
            else if (o.vad_filter_order < 1 || o.vad_filter_order % 2 == 0) bad = "medianFilter: filter order must be positive, odd number!";
            if (bad) {
                g_create_error = bad;
                return CTU_ERR_OPTS;
            }
        }
    }
    const std::string why = unsupported_reason(*e->design);
    if (!why.empty()) {
        g_create_error = "ENGINE: configuration not on the accelerated path: " + why;
        return CTU_ERR_UNSUPPORTED;
    }
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device (the engine has no CPU fallback)");
        if (device < 0 || device >= ndev) throw std::runtime_error("HIP device ordinal out of range");
        HIP_TRY(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        e->device = device;
        e->n_cu = prop.multiProcessorCount;
        build_tables(e.get());
        {
            const ctu::Opts &o = e->design->o;
            const ctu::Design &d = *e->design;
            e->do_vad = o.do_vad();
            VadParams &vp = e->vp;
            std::memset(&vp, 0, sizeof vp);
            vp.K = d.K; vp.wfft = d.wfft; vp.window = d.window;
            vp.cri = o.vad_cri_mode == "energy" ? 0 : (o.vad_cepdist_mode == "lpc" ? 1 : 2);
            vp.ncoef = vp.cri == 1 ? o.vad_lpc_coefs : d.nfea;
            vp.thr = o.vad_thr_mode == "absolute" ? 0 : o.vad_thr_mode == "perc" ? 1 : o.vad_thr_mode == "adapt" ? 2 : 3;
            vp.energy_db = o.vad_energy_db; vp.cep_init = o.vad_cepdist_init; vp.filter_order = o.vad_filter_order;
            vp.cep_p = o.vad_cepdist_p; vp.abs_thr = o.vad_absolute_thr; vp.perc_thr = o.vad_perc_thr;
            vp.adapt_q = o.vad_adapt_q; vp.adapt_za = o.vad_adapt_za; vp.dyn_perc = o.vad_dyn_perc; vp.dyn_min = o.vad_dyn_min;
            vp.qmaxinc = o.vad_dyn_qmaxinc; vp.qmaxdec = o.vad_dyn_qmaxdec; vp.qmindec = o.vad_dyn_qmindec; vp.qmininc = o.vad_dyn_qmininc;
            vp.perc_init = o.vad_perc_init; vp.adapt_init = o.vad_adapt_init; vp.dyn_init = o.vad_dyn_init;
            vp.D = d.D; vp.ncep = o.fea_ncepcoefs; vp.c0_slot = d.row_slot.empty() ? -1 : d.row_slot[0];
        }
        HIP_TRY(hipEventCreate(&e->ev0));
        HIP_TRY(hipEventCreate(&e->ev1));
    } catch (const std::exception &ex) {
        g_create_error = std::string("ENGINE: ") + ex.what();
        return CTU_ERR_DEVICE;
    }
    *out = e.release();
    return CTU_OK;
}

void ctu_engine_destroy(ctu_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    delete e;
}

const char *ctu_last_error(const ctu_engine *e) { return e ? e->err.c_str() : "null engine"; }

int ctu_engine_dims(const ctu_engine *e, ctu_dims *out) {
    if (!e || !out) return CTU_ERR_INPUT;
    fill_dims(*e->design, out);
    return CTU_OK;
}

int64_t ctu_num_frames(const ctu_engine *e, int64_t n) {
    const int pre = e->design->window - e->design->wshift;
    if (n < pre) return -1;
    return (n - pre) / e->design->wshift;
}

int ctu_plan_create(ctu_engine *e, const int64_t *utt_nsamples, int32_t n_utt, ctu_plan **out) {
    if (!e || !out || n_utt < 0 || (n_utt && !utt_nsamples)) return CTU_ERR_INPUT;
    *out = nullptr;
    std::unique_ptr<ctu_plan> pl(new ctu_plan);
    pl->eng = e;
    pl->n_utt = n_utt;
    pl->nsamples.assign(utt_nsamples, utt_nsamples + n_utt);
    pl->sample_off.resize(n_utt + 1);
    pl->row_off.resize(n_utt + 1);
    pl->frames.resize(n_utt);
    const ctu::Design &d = *e->design;
    int64_t so = PCM_HEAD, ro = 0;
    std::vector<TileRec> tiles;
    std::vector<int> uts(n_utt + 1, 0);
    std::vector<int4> uinfo(n_utt);
    std::vector<int> chunks;
    const int trap_chunk = 64;  // output frames per TRAP workgroup
    for (int i = 0; i < n_utt; i++) {
        const int64_t T = ctu_num_frames(e, utt_nsamples[i]);
        if (T < 0) {
            set_error(e, "IO: Signal shorter than one frame!");  // src/io/in.cc:277
            return CTU_ERR_INPUT;
        }
        if (d.post_order > 0 && T > 0) {
            int wmax = 0;
            for (int j = 0; j < d.post_order; j++) wmax = std::max(wmax, d.post_w[j]);
            // With exactly window+1 frames a stage never takes its steady-state branch, so its flush starts from ring
            // slot 0 instead of the oldest slot and the emitted rows mix frames (src/fea/fea_delta.cc:118,183-186);
            // with fewer it reads slots that were never written.  Neither is a feature worth reproducing.
            if (T < wmax + 2) {
                set_error(e, "ENGINE: delta / stacking on fewer than window+2 frames is ill-defined in the reference (src/fea/fea_delta.cc:74-130,178-206)");
                return CTU_ERR_INPUT;
            }
        }
        if (d.kind == ctu::FeaKind::TrapDct && T > 0 && T < (d.o.fea_trapdct_traplen + 1) / 2) {
            set_error(e, "ENGINE: trapdct on fewer than (traplen+1)/2 frames is undefined in the reference (src/fea/fea_trap.cc:64-70)");
            return CTU_ERR_INPUT;
        }
        pl->sample_off[i] = so;
        pl->row_off[i] = ro;
        pl->frames[i] = T;
        uts[i] = (int)tiles.size();
        for (int64_t t0 = 0; t0 < T; t0 += TILE) {
            TileRec r;
            r.sbase = so + t0 * d.wshift;
            r.rbase = ro + t0;
            r.nvalid = (int)std::min<int64_t>(TILE, T - t0);
            r.t0 = (int)t0;
            r.next = -1;
            r.pad = 0;
            tiles.push_back(r);
        }
        uinfo[i] = make_int4((int)(ro & 0xffffffff), (int)(ro >> 32), (int)T, 0);
        for (int64_t tc = 0; tc < T; tc += trap_chunk) {
            chunks.push_back(i);
            chunks.push_back((int)tc);
        }
        so += (utt_nsamples[i] + PCM_ALIGN - 1) / PCM_ALIGN * PCM_ALIGN;
        ro += T;
    }
    uts[n_utt] = (int)tiles.size();
    pl->sample_off[n_utt] = so;
    pl->row_off[n_utt] = ro;
    pl->total_samples = so + PCM_TAIL;  // loads run to the end of the last 32-sample row of a frame
    pl->total_frames = ro;
    pl->n_tiles = (int)tiles.size();
    // Each workgroup walks a chain of tiles.  Stateless chains stride over the tile list; with a
    // per-utterance recurrence (exten) a workgroup takes whole utterances, tile after tile.
    std::vector<int> wg_first;
    const int max_wg = e->n_cu * ((CTU_LB >= 4 && e->lds_bytes <= (size_t)LDS_2WG) ? 2 : 1);
    if (d.o.nr_mode == "exten") {
        std::vector<int> live;  // utterances that have at least one frame
        for (int i = 0; i < n_utt; i++)
            if (uts[i + 1] > uts[i]) live.push_back(i);
        const int G = std::max(1, std::min<int>((int)live.size(), max_wg));
        wg_first.assign(G, -1);
        for (size_t k = 0; k < live.size(); k++) {
            const int u = live[k];
            for (int t = uts[u]; t + 1 < uts[u + 1]; t++) tiles[t].next = t + 1;
            tiles[uts[u + 1] - 1].next = (k + G < live.size()) ? uts[live[k + G]] : -1;
            if ((int)k < G) wg_first[k] = uts[u];
        }
        pl->grid = G;
    } else {
        const int G = std::max(1, std::min(pl->n_tiles, max_wg));
        wg_first.assign(G, -1);
        for (int t = 0; t < pl->n_tiles; t++) tiles[t].next = (t + G < pl->n_tiles) ? t + G : -1;
        for (int g = 0; g < G && g < pl->n_tiles; g++) wg_first[g] = g;
        pl->grid = G;
    }
    try {
        HIP_TRY(hipSetDevice(e->device));
        pl->tiles.upload(tiles);
        pl->wg_first.upload(wg_first);
        if (e->do_vad) {
            pl->d_row_off.upload(pl->row_off);
            if (e->vp.cri == 1) {
                pl->xri.alloc((size_t)ro * d.K);
                pl->pnr.alloc((size_t)ro * d.K);
                pl->vad_ci.alloc((size_t)ro * e->vp.ncoef);
            } else if (e->vp.cri == 0) pl->pnr.alloc((size_t)ro);
        }
        if (d.signal_out) {
            pl->xri.alloc((size_t)ro * d.K);
            pl->pnr.alloc((size_t)ro * d.K);
            pl->utt_info.upload(uinfo);
            std::vector<long long> so64(pl->sample_off.begin(), pl->sample_off.end());
            pl->d_sample_off.upload(so64);
            pl->out_samples.resize(n_utt);
            for (int i = 0; i < n_utt; i++) pl->out_samples[i] = pl->frames[i] * d.wshift + (d.window - d.wshift);
            pl->ybuf.alloc((size_t)ro * d.window);
        }
        if (d.kind == ctu::FeaKind::TrapDct || d.post_order > 0 || d.cms || d.o.stat_cmvn || d.o.apply_cmvn) {
            pl->utt_info.upload(uinfo);
            pl->trap_chunks.upload(chunks);
            pl->n_trap_chunks = (int)chunks.size() / 2;
        }
        if (d.kind == ctu::FeaKind::TrapDct) pl->logmel.alloc((size_t)ro * d.B);
        if (d.post_order > 0 || d.cms) pl->base_rows.alloc((size_t)ro * d.Dbase);
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    *out = pl.release();
    return CTU_OK;
}

void ctu_plan_destroy(ctu_plan *p) { delete p; }
const int64_t *ctu_plan_sample_offsets(const ctu_plan *p) { return p->sample_off.data(); }
const int64_t *ctu_plan_row_offsets(const ctu_plan *p) { return p->row_off.data(); }
int64_t ctu_plan_total_samples(const ctu_plan *p) { return p->total_samples; }
int64_t ctu_plan_total_frames(const ctu_plan *p) { return p->total_frames; }

int ctu_engine_run(ctu_engine *e, const ctu_plan *pl, const int16_t *d_pcm, float *d_rows, uint8_t *d_vad, void *stream) {
    if (!e || !pl || pl->eng != e) return CTU_ERR_INPUT;
    if (pl->n_tiles == 0) return CTU_OK;
    const ctu::Design &d = *e->design;
    const bool signal = d.signal_out;
    if (signal && !e->in_signal_call) {
        set_error(e, "ENGINE: this configuration writes speech (-format_out raw|wave): use ctu_engine_run_signal");
        return CTU_ERR_INPUT;
    }
    if (!d_pcm || (!signal && !d_rows) || (e->do_vad && !d_vad)) {
        set_error(e, "ENGINE: null device buffer");
        return CTU_ERR_INPUT;
    }
    hipStream_t s = (hipStream_t)stream;
    try {
        HIP_TRY(hipSetDevice(e->device));
        KParams kp;
        std::memset(&kp, 0, sizeof kp);
        kp.pcm = d_pcm;
        kp.rows = (d.post_order > 0 || d.cms) ? pl->base_rows.p : d_rows;
        kp.logmel = pl->logmel.p;
        kp.xri = pl->xri.p;
        kp.pnr = pl->pnr.p;
        kp.band_log = d.kind != ctu::FeaKind::Spec;
        kp.band_to_scratch = d.kind == ctu::FeaKind::TrapDct;
        kp.lp_is_lpa = d.kind == ctu::FeaKind::Lpa;
        kp.vad_export = signal ? 1 : (!e->do_vad ? 0 : (e->vp.cri == 1 ? 1 : (e->vp.cri == 0 ? 2 : 0)));
        kp.skip_phase2 = signal ? 1 : 0;
        kp.tiles = pl->tiles.p;
        kp.wg_first = pl->wg_first.p;
        kp.lanec = e->lanec.p;
        kp.ftab = e->ftab.p;
        kp.itab = e->itab.p;
        kp.tab_floats = e->tab_floats;
        kp.ck_off = e->ck_off;
        kp.cf_off = e->cf_off;
        kp.NS = e->NS;
        kp.CW = e->CW;
        kp.ncoef_out = e->ncoef_out;
        kp.K = d.K;
        kp.window = d.window;
        kp.e_slot = d.e_slot;
        kp.e_mode = 0;
        if (d.o.fea_E) {  // energy routing of src/io/batch.cc:98-119
            if (d.o.fea_rawenergy) kp.e_mode = 4;
            else if (d.kind == ctu::FeaKind::Dctc) kp.e_mode = 1;
            else if (d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa) kp.e_mode = 2;
            else kp.e_mode = 3;
        }
        if (signal) kp.e_mode = 0;
        kp.wshift = d.wshift;
        kp.B = d.B;
        kp.nfea = d.nfea;
        kp.D = d.Dbase;
        kp.ncep = d.o.fea_ncepcoefs;
        kp.lporder = d.o.fea_lporder;
        kp.lift_off = e->lift_off;
        kp.preem = d.o.preem;
        kp.inv_window = 1.0f / (float)d.window;
        kp.remove_dc = d.o.remove_dc;
        kp.fb_power = d.o.fb_power;
        kp.fb_inld = d.o.fb_inld;
        kp.lifter_on = d.o.fea_lifter > 1;
        kp.nr_exten = d.o.nr_mode == "exten";
        kp.nr_p = (float)d.o.nr_p;
        kp.nr_a = (float)d.o.nr_a;
        kp.dbg = getenv("CTU_DEBUG_MODE") ? atoi(getenv("CTU_DEBUG_MODE")) : 0;
        const int grid = pl->grid;
#if CTU_STAMP
        if (e->stamps.n < (size_t)grid * NWAVE * 16) e->stamps.alloc((size_t)grid * NWAVE * 16);
        HIP_TRY(hipMemsetAsync(e->stamps.p, 0, e->stamps.n * 8, s));
        kp.stamps = e->stamps.p;
#endif
        HIP_TRY(hipEventRecord(e->ev0, s));
        switch (e->nz) {
            case 13: e->mode ? launch_vx<13, 1>(kp.vad_export, e->feat, dim3(grid), s, kp, e->lds_bytes) : launch_vx<13, 0>(kp.vad_export, e->feat, dim3(grid), s, kp, e->lds_bytes); break;
            default: e->mode ? launch_vx<16, 1>(kp.vad_export, e->feat, dim3(grid), s, kp, e->lds_bytes) : launch_vx<16, 0>(kp.vad_export, e->feat, dim3(grid), s, kp, e->lds_bytes); break;
        }
        HIP_TRY(hipEventRecord(e->ev1, s));
        e->timed = true;
        HIP_TRY(hipGetLastError());
#if CTU_STAMP
        if (const char *sf = getenv("CTU_STAMP_FILE")) {
            HIP_TRY(hipStreamSynchronize(s));
            std::vector<unsigned long long> h(e->stamps.n);
            HIP_TRY(hipMemcpy(h.data(), e->stamps.p, h.size() * 8, hipMemcpyDeviceToHost));
            double sum[16] = {0};
            for (size_t w = 0; w < (size_t)grid * NWAVE; w++)
                for (int i = 0; i < 16; i++) sum[i] += (double)h[w * 16 + i];
            if (FILE *f = fopen(sf, "w")) {
                double tot = 0;
                for (int i = 0; i < 16; i++) tot += sum[i];
                for (int i = 0; i < 16; i++) fprintf(f, "seg %2d  mean cycles per wave %12.0f  share %.3f\n", i, sum[i] / (grid * NWAVE), sum[i] / tot);
                fclose(f);
            }
        }
#endif
        if (e->do_vad) {
            if (e->vp.cri == 1) {
                const dim3 g((unsigned)std::min<int64_t>((pl->total_frames + 3) / 4, (int64_t)e->n_cu * 4));
                const size_t bshm = (512 + (size_t)4 * 2 * (d.wfft / 2 + 4)) * 2 * sizeof(vreal);
#define BURG_LAUNCH(Q, NC) hipLaunchKernelGGL((vad_burg_kernel<Q, NC>), g, dim3(256), bshm, s, pl->xri.p, pl->pnr.p, pl->vad_ci.p, e->vp, pl->total_frames)
                if (d.window <= 256) {
                    if (e->vp.ncoef <= 16) BURG_LAUNCH(4, 16);
                    else BURG_LAUNCH(4, 32);
                } else {
                    if (e->vp.ncoef <= 16) BURG_LAUNCH(8, 16);
                    else BURG_LAUNCH(8, 32);
                }
#undef BURG_LAUNCH
            }
            hipLaunchKernelGGL(vad_decide_kernel, dim3(pl->n_utt), dim3(64), 0, s, pl->vad_ci.p, pl->pnr.p, d_rows,
                               pl->d_row_off.p, pl->n_utt, d_vad, e->vp);
            HIP_TRY(hipGetLastError());
        }
        if (d.kind == ctu::FeaKind::TrapDct) {
            const int tl = d.o.fea_trapdct_traplen, nd = d.o.fea_trapdct_ndct;
            const int ns = (tl + 3) / 4;
            const size_t shm = (size_t)(64 + 4 * (ns <= 26 ? 26 : ns)) * (d.B | 1) * sizeof(float);
#define TRAP_LAUNCH(NRB, NSM)                                                                                          \
    hipLaunchKernelGGL((trapdct_mfma_kernel<NRB, NSM>), dim3(pl->n_trap_chunks), dim3(256), shm, s, pl->logmel.p, d_rows, \
                       e->trapG.p, pl->utt_info.p, pl->trap_chunks.p, d.B, tl, nd, d.D)
            if (nd <= 16 && ns <= 26) TRAP_LAUNCH(1, 26);
            else if (nd <= 16) TRAP_LAUNCH(1, 64);
            else if (ns <= 26) TRAP_LAUNCH(2, 26);
            else TRAP_LAUNCH(2, 64);
#undef TRAP_LAUNCH
            HIP_TRY(hipGetLastError());
        }
        if (d.post_order > 0) {
            PostParams pp;
            std::memset(&pp, 0, sizeof pp);
            pp.fea_c = d.o.fea_ncepcoefs + 1;
            pp.Dbase = d.Dbase;
            pp.D = d.D;
            pp.order = d.post_order;
            pp.stack = d.post_stack ? 1 : 0;
            pp.has_e = d.o.fea_E ? 1 : 0;
            int H = 0;
            for (int j = 0; j < d.post_order; j++) {
                pp.w[j] = d.post_w[j];
                int den = 0;
                for (int i = 1; i <= d.post_w[j]; i++) den += i * i;
                pp.inv_den[j] = (float)(1.0 / (2.0 * den));
                H += d.post_w[j];
            }
            const int R = 64 + 2 * H;
            const size_t shm = ((size_t)R * d.Dbase + (size_t)(d.post_stack ? 0 : d.post_order) * R * pp.fea_c) * sizeof(float);
            if ((size_t)R * d.Dbase > 256 * 12) throw std::runtime_error("delta tile larger than the prefetch registers");
            const int pgrid = std::min(pl->n_trap_chunks, e->n_cu * 8);
            hipLaunchKernelGGL(post_kernel, dim3(pgrid), dim3(256), shm, s, pl->base_rows.p, d_rows,
                               pl->utt_info.p, pl->trap_chunks.p, pl->n_trap_chunks, pp);
            HIP_TRY(hipGetLastError());
        }
        if (d.cms) {
            CmsParams cp;
            std::memset(&cp, 0, sizeof cp);
            cp.ncols = d.cms_cols;
            cp.Dbase = d.Dbase;
            cp.D = d.D;
            cp.copy_rest = d.post_order > 0 ? 0 : 1;
            cp.L = d.o.length_b;
            cp.z = d.o.fea_Z_exp;
            cp.omz = 1 - d.o.fea_Z_exp;
            if (d.cms == 1)
                hipLaunchKernelGGL(cms_exp_kernel, dim3((pl->n_utt + 1) / 2), dim3(64), 0, s, pl->base_rows.p, d_rows,
                                   pl->utt_info.p, pl->n_utt, cp);
            else
                hipLaunchKernelGGL(cms_block_kernel, dim3(pl->n_trap_chunks), dim3(256),
                                   (size_t)(64 + cp.L - 1) * cp.ncols * sizeof(float), s, pl->base_rows.p, d_rows,
                                   pl->utt_info.p, pl->trap_chunks.p, cp);
            HIP_TRY(hipGetLastError());
        }
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    return CTU_OK;
}

int ctu_engine_run_host(ctu_engine *e, const ctu_plan *pl, const int16_t *h_pcm, float *h_rows, uint8_t *h_vad,
                        int64_t *rows_per_utt) {
    if (!e || !pl || pl->eng != e) return CTU_ERR_INPUT;
    const ctu::Design &d = *e->design;
    if (rows_per_utt)
        for (int i = 0; i < pl->n_utt; i++) rows_per_utt[i] = pl->frames[i];
    if (pl->total_frames == 0) return CTU_OK;
    try {
        HIP_TRY(hipSetDevice(e->device));
        DevBuf<int16_t> pcm;
        DevBuf<float> rows;
        DevBuf<uint8_t> vad;
        pcm.alloc((size_t)pl->total_samples);
        rows.alloc((size_t)pl->total_frames * d.D);
        if (e->do_vad) vad.alloc((size_t)pl->total_frames);
        HIP_TRY(hipMemcpy(pcm.p, h_pcm, (size_t)pl->total_samples * 2, hipMemcpyHostToDevice));
        int rc = ctu_engine_run(e, pl, pcm.p, rows.p, vad.p, nullptr);
        if (rc != CTU_OK) return rc;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(h_rows, rows.p, (size_t)pl->total_frames * d.D * 4, hipMemcpyDeviceToHost));
        if (e->do_vad) {
            std::vector<uint8_t> v((size_t)pl->total_frames);
            HIP_TRY(hipMemcpy(v.data(), vad.p, v.size(), hipMemcpyDeviceToHost));
            if (h_vad) std::memcpy(h_vad, v.data(), v.size());
            if (d.o.vad_apply_mode == "drop") {
                // rows of non-speech frames are dropped (src/io/batch.cc:237-238): compact each utterance in place
                for (int i = 0; i < pl->n_utt; i++) {
                    const int64_t r0 = pl->row_off[i], T = pl->frames[i];
                    int64_t keep = 0;
                    for (int64_t t = 0; t < T; t++)
                        if (v[r0 + t] == '1') {
                            if (keep != t) std::memmove(h_rows + (r0 + keep) * d.D, h_rows + (r0 + t) * d.D, (size_t)d.D * 4);
                            keep++;
                        }
                    if (rows_per_utt) rows_per_utt[i] = keep;
                }
            }
        }
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    return CTU_OK;
}

static void cmvn_maps(ctu_engine *e) {
    if (!e->col_of_slot.empty()) return;
    const ctu::Design &d = *e->design;
    const int fc = d.o.fea_ncepcoefs + 1, X = fc * (d.post_order + 1);
    e->col_of_slot.assign(X, 0);
    e->slot_of_col.assign(d.D, -1);
    for (int k = 0; k < X; k++) {
        const int i = (k + 1) % X;               // internal vector entry held by slot k (post_impl.cc:56-62)
        const int j = i / fc, ii = i % fc;       // block, entry within the block (0 = c0)
        const int col = fc * j + (ii == 0 ? fc - 1 : ii - 1);  // writer order: c1..cN, c0 (out.cc:188-201)
        e->col_of_slot[k] = col;
        e->slot_of_col[col] = k;
    }
    e->d_col_of_slot.upload(e->col_of_slot);
    e->d_slot_of_col.upload(e->slot_of_col);
}

static int cmvn_check(ctu_engine *e, const ctu_plan *pl, const int32_t *spk_of_utt, int32_t n_spk) {
    if (!e || !pl || pl->eng != e) return CTU_ERR_INPUT;
    const ctu::Design &d = *e->design;
    if (!(d.o.stat_cmvn || d.o.apply_cmvn)) {
        set_error(e, "ENGINE: engine was not created with -stat_cmvn / -apply_cmvn");
        return CTU_ERR_INPUT;
    }
    if (n_spk < 1 || (pl->n_utt && !spk_of_utt)) {
        set_error(e, "ENGINE: speaker table missing");
        return CTU_ERR_INPUT;
    }
    for (int i = 0; i < pl->n_utt; i++)
        if (spk_of_utt[i] < 0 || spk_of_utt[i] >= n_spk) {
            set_error(e, "ENGINE: speaker index out of range");
            return CTU_ERR_INPUT;
        }
    return CTU_OK;
}

int ctu_cmvn_cols(const ctu_engine *e) {
    if (!e) return -1;
    const ctu::Design &d = *e->design;
    return (d.o.fea_ncepcoefs + 1) * (d.post_order + 1);
}

int ctu_cmvn_accumulate(ctu_engine *e, const ctu_plan *pl, const float *d_rows, const int32_t *spk_of_utt, int32_t n_spk,
                        const double *mean, double *acc, void *stream) {
    int rc = cmvn_check(e, pl, spk_of_utt, n_spk);
    if (rc != CTU_OK) return rc;
    if (!acc || (pl->total_frames && !d_rows)) return CTU_ERR_INPUT;
    if (pl->total_frames == 0) return CTU_OK;
    hipStream_t s = (hipStream_t)stream;
    try {
        HIP_TRY(hipSetDevice(e->device));
        cmvn_maps(e);
        const int cols = ctu_cmvn_cols(e);
        std::vector<int> spk(spk_of_utt, spk_of_utt + pl->n_utt);
        e->d_spk.upload(spk);
        std::vector<double> zero((size_t)n_spk * (cols + 1), 0.0);
        e->d_stat_a.upload(zero);
        if (mean) e->d_stat_b.upload(std::vector<double>(mean, mean + (size_t)n_spk * cols));
        hipLaunchKernelGGL(cmvn_accumulate_kernel, dim3(pl->n_trap_chunks), dim3(256), 0, s, d_rows, pl->utt_info.p,
                           pl->trap_chunks.p, e->d_spk.p, e->d_col_of_slot.p, mean ? e->d_stat_b.p : nullptr, e->d_stat_a.p,
                           cols, e->design->D);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(zero.data(), e->d_stat_a.p, zero.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < zero.size(); i++) acc[i] += zero[i];
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    return CTU_OK;
}

int ctu_cmvn_apply(ctu_engine *e, const ctu_plan *pl, float *d_rows, const int32_t *spk_of_utt, int32_t n_spk,
                   const double *mean, const double *var, void *stream) {
    int rc = cmvn_check(e, pl, spk_of_utt, n_spk);
    if (rc != CTU_OK) return rc;
    if (!mean || !var || (pl->total_frames && !d_rows)) return CTU_ERR_INPUT;
    if (pl->total_frames == 0) return CTU_OK;
    hipStream_t s = (hipStream_t)stream;
    try {
        HIP_TRY(hipSetDevice(e->device));
        cmvn_maps(e);
        const int cols = ctu_cmvn_cols(e);
        std::vector<int> spk(spk_of_utt, spk_of_utt + pl->n_utt);
        e->d_spk.upload(spk);
        e->d_stat_a.upload(std::vector<double>(mean, mean + (size_t)n_spk * cols));
        e->d_stat_b.upload(std::vector<double>(var, var + (size_t)n_spk * cols));
        hipLaunchKernelGGL(cmvn_apply_kernel, dim3(pl->n_trap_chunks), dim3(256), 0, s, d_rows, pl->utt_info.p,
                           pl->trap_chunks.p, e->d_spk.p, e->d_slot_of_col.p, e->d_stat_a.p, e->d_stat_b.p, cols, e->design->D);
        HIP_TRY(hipGetLastError());
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    return CTU_OK;
}

const int64_t *ctu_plan_out_samples(const ctu_plan *p) { return p->out_samples.empty() ? nullptr : p->out_samples.data(); }

int ctu_engine_run_signal(ctu_engine *e, const ctu_plan *pl, const int16_t *d_pcm, int16_t *d_out, void *stream) {
    if (!e || !pl || pl->eng != e) return CTU_ERR_INPUT;
    const ctu::Design &d = *e->design;
    if (!d.signal_out) {
        set_error(e, "ENGINE: ctu_engine_run_signal needs -format_out raw|wave");
        return CTU_ERR_INPUT;
    }
    if (pl->n_utt == 0) return CTU_OK;
    if (!d_pcm || !d_out) {
        set_error(e, "ENGINE: null device buffer");
        return CTU_ERR_INPUT;
    }
    hipStream_t s = (hipStream_t)stream;
    e->in_signal_call = true;
    const int rc = ctu_engine_run(e, pl, d_pcm, nullptr, nullptr, stream);  // spectra before / after NR into the plan's scratch
    e->in_signal_call = false;
    if (rc != CTU_OK) return rc;
    try {
        HIP_TRY(hipSetDevice(e->device));
        SynthParams sp;
        sp.K = d.K; sp.wfft = d.wfft; sp.window = d.window; sp.wshift = d.wshift;
        sp.inv_n = 1.0f / (float)d.wfft;
        sp.corr = d.ola_corr;
        if (pl->total_frames > 0) {
            const int g = (int)std::min<int64_t>((pl->total_frames + 3) / 4, (int64_t)e->n_cu * 8);
            hipLaunchKernelGGL(synth_kernel, dim3(g), dim3(256), 0, s, pl->xri.p, pl->pnr.p, pl->ybuf.p, (long long)pl->total_frames, sp);
            HIP_TRY(hipGetLastError());
        }
        int64_t longest = 0;
        for (int64_t n : pl->out_samples) longest = std::max(longest, n);
        const int gx = (int)std::max<int64_t>(1, std::min<int64_t>((longest + 255) / 256, 64));
        hipLaunchKernelGGL(ola_kernel, dim3(gx, pl->n_utt), dim3(256), 0, s, pl->ybuf.p, d_out, pl->utt_info.p, pl->d_sample_off.p,
                           pl->n_utt, sp);
        HIP_TRY(hipGetLastError());
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    return CTU_OK;
}

int ctu_engine_run_signal_host(ctu_engine *e, const ctu_plan *pl, const int16_t *h_pcm, int16_t *h_out) {
    if (!e || !pl || pl->eng != e) return CTU_ERR_INPUT;
    try {
        HIP_TRY(hipSetDevice(e->device));
        DevBuf<int16_t> pcm, out;
        pcm.alloc((size_t)pl->total_samples);
        out.alloc((size_t)pl->total_samples);
        HIP_TRY(hipMemcpy(pcm.p, h_pcm, (size_t)pl->total_samples * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(out.p, 0, (size_t)pl->total_samples * 2));
        const int rc = ctu_engine_run_signal(e, pl, pcm.p, out.p, nullptr);
        if (rc != CTU_OK) return rc;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(h_out, out.p, (size_t)pl->total_samples * 2, hipMemcpyDeviceToHost));
        return CTU_OK;
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
}

int ctu_cmvn_accumulate_host(ctu_engine *e, const ctu_plan *pl, const float *h_rows, const int32_t *spk_of_utt, int32_t n_spk,
                             const double *mean, double *acc) {
    if (!e || !pl || pl->eng != e) return CTU_ERR_INPUT;
    if (pl->total_frames == 0) return CTU_OK;
    try {
        HIP_TRY(hipSetDevice(e->device));
        DevBuf<float> rows;
        rows.alloc((size_t)pl->total_frames * e->design->D);
        HIP_TRY(hipMemcpy(rows.p, h_rows, rows.n * sizeof(float), hipMemcpyHostToDevice));
        return ctu_cmvn_accumulate(e, pl, rows.p, spk_of_utt, n_spk, mean, acc, nullptr);
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
}

int ctu_cmvn_apply_host(ctu_engine *e, const ctu_plan *pl, float *h_rows, const int32_t *spk_of_utt, int32_t n_spk,
                        const double *mean, const double *var) {
    if (!e || !pl || pl->eng != e) return CTU_ERR_INPUT;
    if (pl->total_frames == 0) return CTU_OK;
    try {
        HIP_TRY(hipSetDevice(e->device));
        DevBuf<float> rows;
        rows.alloc((size_t)pl->total_frames * e->design->D);
        HIP_TRY(hipMemcpy(rows.p, h_rows, rows.n * sizeof(float), hipMemcpyHostToDevice));
        const int rc = ctu_cmvn_apply(e, pl, rows.p, spk_of_utt, n_spk, mean, var, nullptr);
        if (rc != CTU_OK) return rc;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(h_rows, rows.p, rows.n * sizeof(float), hipMemcpyDeviceToHost));
        return CTU_OK;
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
}

float ctu_engine_last_kernel_ms(ctu_engine *e) {
    if (!e || !e->timed) return -1.f;
    float ms = -1.f;
    if (hipEventSynchronize(e->ev1) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, e->ev0, e->ev1) != hipSuccess) return -1.f;
    return ms;
}

}  // extern "C"
