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
constexpr int PSTRIDE = 257;   // floats per P-tile row (odd: lane-per-frame column reads are conflict-free)
constexpr int AUX_ROWS = 62;   // [AUX_ROWS][64] floats of per-band / per-coefficient / output staging
constexpr int LDS_FLOATS = TILE * PSTRIDE + AUX_ROWS * 64;
static_assert(LDS_FLOATS * 4 <= 80 * 1024, "two workgroups per CU need <= 80 KiB each");
constexpr int MAX_LP = 16;     // Levinson order limit of the in-register recursion
constexpr int PCM_ALIGN = 8;   // utterance starts are multiples of this many samples
constexpr int PCM_HEAD = 8;    // samples of padding before the first utterance (x[-2..-1] of frame 0 is loaded)
constexpr int PCM_TAIL = 64;   // padding after the last one (loads run to the end of the 32-sample row)

// Per-lane constant record, one per l16 = lane & 15, streamed from L1 every pass instead of pinning
// 70+ VGPRs:  [0,32) Hamming pairs (w[32j+2l], w[32j+2l+1]) j=0..15 | [32,64) 1/0 "sample is inside the
// window" pairs for DC removal | [64,96) inter-stage twiddles W256^(l*k1), k1=1..15 (+pad) |
// [96,112) W512^(l+16*k2), k2=0..7
constexpr int LC_WIN = 0, LC_MASK = 32, LC_TW = 64, LC_UT = 96, LANEC = 112;

enum FeatMode { FEAT_SPEC = 0, FEAT_LOGSPEC = 1, FEAT_DCTC = 2, FEAT_LPC = 3, FEAT_LPA = 4, FEAT_LOGMEL_SCRATCH = 5 };

struct KParams {
    const int16_t *pcm;
    float *rows;
    float *logmel;              // [total_frames][B] scratch (TRAP only)
    const int4 *tiles;
    const int64_t *sample_off;  // per utterance
    const int64_t *row_off;     // per utterance
    const int *utt_tile_start;  // [n_utt+1] (by_utt)
    const float *lanec;         // [16][LANEC]
    const float *ftab;          // band weights at 0, then dct/idft table at dct_off, lifter at lift_off
    const int *itab;            // band_first[B] | band_len[B] | band_off[B] | grp_start[NWAVE+1] | row_slot[nfea]
    int n_tiles, n_utt, by_utt;
    int wshift, B, nfea, D, ncep, lporder;
    int dct_off, lift_off;
    float preem, inv_window;
    int remove_dc, fb_power, fb_inld, lifter_on, nr_exten;
    float nr_p, nr_a;
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

struct __attribute__((aligned(4))) pcm4 {  // four consecutive int16 samples, 4-byte aligned
    uint32_t lo, hi;
};

// NZ = number of 32-sample rows that can hold non-zero input (ceil(window/32)); rows >= NZ are
// literal zeros so the compiler prunes the first butterflies.
template <int NZ, int FEAT>
__global__ __launch_bounds__(WG, 4) void frontend_kernel(const KParams p) {
    extern __shared__ __align__(16) float lds[];
    float *Pt = lds;                       // [TILE][PSTRIDE]
    float *aux = lds + TILE * PSTRIDE;     // [AUX_ROWS][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15;   // n2 in stage 1, k1 in stage 2
    const int fg = lane >> 4;    // frame slot within the wave pass
    const float4 *lc = reinterpret_cast<const float4 *>(p.lanec + l16 * LANEC);
    const int partner = ((lane & 48) | ((16 - l16) & 15)) << 2;  // byte address for ds_bpermute
    const int B = p.B;
    const int *band_first = p.itab, *band_len = p.itab + B, *band_off = p.itab + 2 * B;
    const int *grp_start = p.itab + 3 * B, *row_slot = p.itab + 3 * B + NWAVE + 1;

    // exten NR state: thread = bin
    float navg = 0.95f, yavg = 0.05f;
    int cur_utt = -1;

    const int n_outer = p.by_utt ? p.n_utt : p.n_tiles;
    for (int outer = blockIdx.x; outer < n_outer; outer += gridDim.x) {
        int tile_lo = outer, tile_hi = outer + 1;
        if (p.by_utt) {
            tile_lo = p.utt_tile_start[outer];
            tile_hi = p.utt_tile_start[outer + 1];
        }
        for (int tile = tile_lo; tile < tile_hi; tile++) {
            const int4 td = p.tiles[tile];
            const int utt = td.x, t0 = td.y, nvalid = td.z;
            const int64_t sbase = p.sample_off[utt];
            const int64_t rbase = p.row_off[utt] + t0;

            // ================= phase 1: frames -> power spectrum rows =================
            float *scratch = Pt + (wave * 8 + 4) * PSTRIDE;  // this wave's last 4 rows double as transpose scratch
#pragma unroll 1
            for (int it = 0; it < 2; it++) {
                const int f = wave * 8 + it * 4 + fg;               // frame slot in tile
                const int fc = f < nvalid ? f : nvalid - 1;         // clamp (duplicates are never stored)
                const int t = t0 + fc;
                // samples x[i0-2 .. i0+1] of row j sit at x + 32 j + 2 l16 - 2 (4-byte aligned)
                const int16_t *x = p.pcm + sbase + (int64_t)t * p.wshift + 2 * l16 - 2;

                float2 v[16];
                float dc = 0.f;
                {
                    pcm4 q[NZ];
#pragma unroll
                    for (int j = 0; j < NZ; j++) q[j] = *reinterpret_cast<const pcm4 *>(x + 32 * j);
#pragma unroll
                    for (int j = 0; j < NZ; j++) {
                        const float4 w4 = lc[(LC_WIN + 2 * j) >> 2];  // two rows of window pairs per float4
                        const float w0 = (j & 1) ? w4.z : w4.x, w1 = (j & 1) ? w4.w : w4.y;
                        float xm = (float)(int16_t)(q[j].lo >> 16);
                        const float x0 = (float)(int16_t)(q[j].hi & 0xffffu);
                        const float x1 = (float)(int16_t)(q[j].hi >> 16);
                        if (j == 0) xm = (l16 == 0 && t == 0) ? 0.f : xm;  // first sample of the file: history is 0
                        const float y0 = w0 * (x0 - p.preem * xm);
                        const float y1 = w1 * (x1 - p.preem * x0);         // w is 0 beyond the window
                        v[j] = make_float2(y0, y1);
                        dc += y0 + y1;
                    }
                }
#pragma unroll
                for (int j = NZ; j < 16; j++) v[j] = make_float2(0.f, 0.f);
                if (p.remove_dc) {
                    // mean of the windowed frame over `window` samples (src/io/in.cc:375-382)
                    dc += __shfl_xor(dc, 8, 64);
                    dc += __shfl_xor(dc, 4, 64);
                    dc += __shfl_xor(dc, 2, 64);
                    dc += __shfl_xor(dc, 1, 64);
                    const float m = dc * p.inv_window;
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

                // ---- stage 1: DFT16 over n1 (registers), lane = n2; then twiddle W256^(n2*k1)
                __builtin_amdgcn_sched_barrier(0);  // keep the twiddle loads below the PCM/window block
                float4 tw[8];
#pragma unroll
                for (int h = 0; h < 8; h++) tw[h] = lc[(LC_TW >> 2) + h];  // (k1 = 2h+1, k1 = 2h+2)
                dft16(v);
#pragma unroll
                for (int h = 0; h < 8; h++) {
                    v[2 * h + 1] = cmul(v[2 * h + 1], make_float2(tw[h].x, tw[h].y));
                    if (2 * h + 2 < 16) v[2 * h + 2] = cmul(v[2 * h + 2], make_float2(tw[h].z, tw[h].w));
                }
                __builtin_amdgcn_sched_barrier(0);

                // ---- transpose [k1][n2] -> lane k1 holds all n2, through LDS, re then im
                //      element (k1,n2) of frame slot fg at  fg*256 + ((k1^(fg&1))<<4) + (n2^k1)
                const int sw = fg * 256;
                const int par = fg & 1;
                __builtin_amdgcn_wave_barrier();
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

                // ---- stage 2: DFT16 over n2, lane = k1: v[k2] = Z[k1 + 16 k2]
                __builtin_amdgcn_sched_barrier(0);
                float4 u4[4];
#pragma unroll
                for (int h = 0; h < 4; h++) u4[h] = lc[(LC_UT >> 2) + h];
                dft16(v);

                // ---- untangle the packed real FFT and take |.|^2.  Lane k1 handles its bins k2=0..7,
                //      each together with its mirror bin 256-k held by lane (16-k1)%16 in register 15-k2
                //      (register (16-k2)%16 for k1 = 0).
                float *prow = Pt + f * PSTRIDE;
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) {
                    float br = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(v[15 - k2].x)));
                    float bi = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(v[15 - k2].y)));
                    if (l16 == 0) {
                        br = v[(16 - k2) & 15].x;
                        bi = v[(16 - k2) & 15].y;
                    }
                    const float wr = (k2 & 1) ? u4[k2 >> 1].z : u4[k2 >> 1].x, wi = (k2 & 1) ? u4[k2 >> 1].w : u4[k2 >> 1].y;
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
                }
                if (l16 == 0) {  // bin 128 is its own mirror: X[128] = conj(Z[128]); bin 0 floor (src/io/in.cc:390)
                    prow[128] = v[8].x * v[8].x + v[8].y * v[8].y;
                    if (p.remove_dc) prow[0] = 1e-10f;
                }
            }
            __syncthreads();

            if (!p.fb_power) {  // magnitude instead of power (src/io/in.cc:415-417); off the default path
                for (int e = tid; e < nvalid * 257; e += WG) {
                    const int f = e / 257, k = e - f * 257;
                    Pt[f * PSTRIDE + k] = sqrtf(Pt[f * PSTRIDE + k]);
                }
                __syncthreads();
            }

            // ================= extended spectral subtraction (src/nr/nr.cc:86-140) =================
            if (p.nr_exten) {
                if (utt != cur_utt) {  // new file: Navg = 0.95, Yavg = 0.05
                    navg = 0.95f;
                    yavg = 0.05f;
                    cur_utt = utt;
                }
                if (tid < 257) {
                    const float pp = p.nr_p, qq = 1.0f - p.nr_p;
                    for (int f = 0; f < nvalid; f++) {
                        const float X = Pt[f * PSTRIDE + tid];
                        float H;
                        if (p.nr_a == 1.0f) H = navg / (navg + yavg);
                        else if (p.nr_a == 2.0f) H = navg / sqrtf(navg * navg + yavg * yavg);
                        else H = navg / powf(powf(navg, p.nr_a) + powf(yavg, p.nr_a), 1.0f / p.nr_a);
                        const float N = H * X;
                        navg = pp * navg + qq * N;
                        yavg = fabsf(X - navg);
                        Pt[f * PSTRIDE + tid] = X - N;
                    }
                }
                __syncthreads();
            }

            // ================= phase 2: lane = frame =================
            const float *prow = Pt + lane * PSTRIDE;
            {
                const int b0 = grp_start[wave], b1 = grp_start[wave + 1];
                for (int b = b0; b < b1; b++) {
                    const int kf = band_first[b], len = band_len[b];
                    const float *w = p.ftab + band_off[b];
                    float acc = 0.f;
                    for (int i = 0; i < len; i += 4) {
                        acc += w[i + 0] * prow[kf + i + 0];
                        acc += w[i + 1] * prow[kf + i + 1];
                        acc += w[i + 2] * prow[kf + i + 2];
                        acc += w[i + 3] * prow[kf + i + 3];
                    }
                    if (p.fb_inld) acc = __powf(acc, 0.33f);  // src/fea/fb.cc:81-83
                    if (FEAT == FEAT_LOGSPEC || FEAT == FEAT_DCTC || FEAT == FEAT_LOGMEL_SCRATCH) acc = __logf(acc);
                    aux[b * 64 + lane] = acc;
                }
            }
            __syncthreads();

            float *stage = aux + B * 64;  // [64][D] output rows (after the R rows for LPC)
            int out_w = p.D;
            float *dst = p.rows;
            if (FEAT == FEAT_SPEC || FEAT == FEAT_LOGSPEC || FEAT == FEAT_LOGMEL_SCRATCH) {
                // rows are the band values themselves: coalesced transpose-copy out of aux
                if (FEAT == FEAT_LOGMEL_SCRATCH) {
                    dst = p.logmel;
                    out_w = B;
                }
                for (int e = tid; e < nvalid * B; e += WG) {
                    const int f = e / B, b = e - f * B;
                    dst[(rbase + f) * out_w + b] = aux[b * 64 + f];
                }
            } else {
                if (FEAT == FEAT_DCTC) {
                    // c_i = sum_b dct[i][b] * logY[b]   (norm and lifter folded into the table)
                    for (int i = wave; i < p.nfea; i += NWAVE) {
                        const float *d = p.ftab + p.dct_off + i * B;
                        float c = 0.f;
                        for (int b = 0; b < B; b++) c += d[b] * aux[b * 64 + lane];
                        const int slot = row_slot[i];
                        if (slot >= 0) stage[lane * p.D + slot] = c;
                    }
                } else {  // LPC / LPA
                    // autocorrelation by cosine iDFT, k spread over the waves (src/fea/fea_impl.cc:181-198)
                    float *R = aux + B * 64;
                    stage = R + (p.lporder + 1) * 64;
                    for (int k = wave; k <= p.lporder; k += NWAVE) {
                        const float *d = p.ftab + p.dct_off + k * B;
                        float r = 0.f;
                        for (int b = 0; b < B; b++) {
                            float y = aux[b * 64 + lane];
                            if (!p.fb_inld) y *= y;  // src/fea/fea_impl.cc:165-169
                            r += d[b] * y;
                        }
                        R[k * 64 + lane] = r;
                    }
                    __syncthreads();
                    if (wave == 0) {
                        // Levinson-Durbin in double (src/fea/fea_impl.cc:200-222; the reference's aa[] copy is
                        // replaced by the in-place symmetric update, same operations), then a -> c (251-284)
                        const int P_ = p.lporder;
                        double a[MAX_LP + 1], c[MAX_LP + 1];
                        const double r0 = R[lane];
                        double rc = -(double)R[64 + lane] / r0;
                        double err = r0 * (1 - rc * rc);
                        a[0] = 1;
                        a[1] = rc;
#pragma unroll
                        for (int ik = 2; ik <= MAX_LP; ik++) {
                            if (ik <= P_) {
                                double dm = R[ik * 64 + lane];
#pragma unroll
                                for (int n = 1; n < ik; n++) dm += a[n] * (double)R[(ik - n) * 64 + lane];
                                rc = -dm / err;
#pragma unroll
                                for (int n = 1; n <= ik / 2; n++) {
                                    const double lo = a[n], hi = a[ik - n];
                                    a[n] = lo + rc * hi;
                                    if (n != ik - n) a[ik - n] = hi + rc * lo;
                                }
                                a[ik] = rc;
                                err *= (1 - rc * rc);
                            }
                        }
                        if (FEAT == FEAT_LPA) {
#pragma unroll
                            for (int i = 1; i <= MAX_LP; i++)
                                if (i <= P_) stage[lane * p.D + (i - 1)] = (float)a[i];
                        } else {
                            c[0] = log(err);
#pragma unroll
                            for (int n = 1; n <= MAX_LP; n++) {
                                if (n <= p.ncep) {
                                    double sum = 0;
#pragma unroll
                                    for (int k = 1; k < n; k++)
                                        if (k <= P_) sum += (n - k) * c[n - k] * a[k];
                                    c[n] = (n <= P_ ? -a[n] : 0.0) - sum / n;
                                }
                            }
#pragma unroll
                            for (int n = 0; n <= MAX_LP; n++) {
                                if (n <= p.ncep) {
                                    double val = c[n];
                                    if (n >= 1 && p.lifter_on) val *= (double)p.ftab[p.lift_off + n - 1];
                                    const int slot = row_slot[n];
                                    if (slot >= 0) stage[lane * p.D + slot] = (float)val;
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                // coalesced store of the tile's [nvalid][D] block
                float *o = p.rows + rbase * p.D;
                for (int e = tid; e < nvalid * p.D; e += WG) o[e] = stage[e];
            }
            __syncthreads();  // P tile / aux are reused by the next tile
        }
    }
}

// TRAP-DCT (src/fea/fea_trap.cc:53-127): out[t][b*ndct+k] = sum_j G[k][j] * logmel[clamp(t-half+j)][b]
// with mean removal, Hamming and REDFT10 folded into G on the host (rows of G sum to zero, so the centre
// frame's value is subtracted first to keep the fp32 accumulation small).  One thread per (t, b).
__global__ void trapdct_kernel(const float *__restrict__ logmel, float *__restrict__ rows, const float *__restrict__ G,
                               const int4 *__restrict__ utt_info /* {row_off lo, row_off hi, T, -} */, int n_utt, int B,
                               int traplen, int ndct, int D, const int *__restrict__ utt_of_chunk, int chunk) {
    const int u = utt_of_chunk[blockIdx.x * 2];
    const int tc = utt_of_chunk[blockIdx.x * 2 + 1];
    const int4 ui = utt_info[u];
    const int64_t r0 = ((int64_t)ui.y << 32) | (uint32_t)ui.x;
    const int T = ui.z;
    const int half = (traplen - 1) / 2;
    for (int e = threadIdx.x; e < chunk * B; e += blockDim.x) {
        const int t = tc + e / B, b = e % B;
        if (t >= T) continue;
        float acc[32];
#pragma unroll
        for (int k = 0; k < 32; k++) acc[k] = 0.f;
        const float xc = logmel[(r0 + t) * B + b];
        for (int j = 0; j < traplen; j++) {
            int s = t - half + j;
            s = s < 0 ? 0 : (s > T - 1 ? T - 1 : s);
            const float x = logmel[(r0 + s) * B + b] - xc;
#pragma unroll
            for (int k = 0; k < 32; k++)
                if (k < ndct) acc[k] += G[k * traplen + j] * x;
        }
        float *o = rows + (r0 + t) * D + b * ndct;
#pragma unroll
        for (int k = 0; k < 32; k++)
            if (k < ndct) o[k] = acc[k];
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

struct ctu_engine {
    std::unique_ptr<ctu::Design> design;
    int device = 0;
    int n_cu = 256;
    std::string err;
    int feat = FEAT_DCTC;
    int nz = 16;
    DevBuf<float> lanec, ftab, trapG;
    DevBuf<int> itab;
    int dct_off = 0, lift_off = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    DevBuf<float> logmel;  // TRAP scratch, sized by the largest plan seen
};

struct ctu_plan {
    ctu_engine *eng = nullptr;
    int n_utt = 0;
    std::vector<int64_t> nsamples, sample_off, row_off, frames;
    int64_t total_samples = 0, total_frames = 0;
    int n_tiles = 0;
    DevBuf<int4> tiles;
    DevBuf<int64_t> d_sample_off, d_row_off;
    DevBuf<int> utt_tile_start;
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
    if (o.format_in == "htk") return "HTK feature input (-format_in htk) bypasses the spectral path";
    if (o.fea_kind == "td-iir-mfcc" || o.fea_kind == "none") return "fea_kind outside the spectral feature path";
    if (o.dither != 0.) return "-dither != 0 makes outputs depend on file order (src/io/in.cc:205,454)";
    if (o.remove_dc1) return "-remove_dc1 mutates the sample history across frames (src/io/in.cc:343-350)";
    if (o.nr_mode != "none" && o.nr_mode != "exten") return "nr_mode hwss/fwss/2fwss seed their noise estimate from the previous file (src/nr/nr.cc:212-221)";
    if (o.nr_when_afterFB) return "-nr_when afterFB";
    if (o.rasta) return "-nr_rasta";
    if (o.fea_delta || o.fea_trap) return "delta / stacked features (next row N1)";
    if (o.stat_cmvn || o.apply_cmvn || o.fea_Z_exp > 0 || o.fea_Z_block > 0) return "CMVN / CMS (next row N2)";
    if (o.fea_E) return "-fea_E on";
    if (o.do_vad()) return "VAD module";
    if (d.wfft != 512) return "FFT size other than 512";
    if (d.wshift % 2) return "odd frame shift";
    if (d.window < 32) return "window shorter than 32 samples";
    if (d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa) {
        if (d.B + o.fea_lporder + 1 + d.D > AUX_ROWS) return "filter bank + LP order too large for the LDS staging";
        if (o.fea_lporder > MAX_LP || o.fea_ncepcoefs > MAX_LP) return "LP order / cepstral order above the in-register limit";
    } else if (d.kind == ctu::FeaKind::Dctc) {
        if (d.B + d.D > AUX_ROWS) return "filter bank + cepstral order too large for the LDS staging";
    } else if (d.B > AUX_ROWS) return "more filter bank channels than the LDS staging holds";
    if (d.kind == ctu::FeaKind::TrapDct && o.fea_trapdct_ndct > 32) return "more than 32 TRAP DCT coefficients";
    return "";
}

void build_tables(ctu_engine *e) {
    const ctu::Design &d = *e->design;
    const double pi = 3.14159265358979323846;
    // ---- per-lane constant records (see LC_* above)
    std::vector<float> lc(16 * LANEC, 0.f);
    for (int l = 0; l < 16; l++) {
        float *r = lc.data() + l * LANEC;
        for (int j = 0; j < 16; j++)
            for (int h = 0; h < 2; h++) {
                const int i = 32 * j + 2 * l + h;
                r[LC_WIN + 2 * j + h] = i < d.window ? (float)d.hamming[i] : 0.f;
            }
        for (int j = 0; j < 16; j++)
            for (int h = 0; h < 2; h++) r[LC_MASK + 2 * j + h] = (32 * j + 2 * l + h) < d.window ? 1.f : 0.f;
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
    // ---- banded filter bank: per band a run of weights padded to a multiple of 4 bins inside [0,K)
    std::vector<int> bf(d.B), bl(d.B), bo(d.B);
    std::vector<float> ft;
    for (int b = 0; b < d.B; b++) {
        int first = d.fb_first[b], len = d.fb_last[b] - d.fb_first[b] + 1;
        int plen = (len + 3) & ~3;
        if (first + plen > d.K) first = d.K - plen;
        if (first < 0) throw std::runtime_error("filter band wider than the spectrum");
        bf[b] = first;
        bl[b] = plen;
        bo[b] = (int)ft.size();
        for (int i = 0; i < plen; i++) {
            const int k = first + i;
            ft.push_back((k >= d.fb_first[b] && k <= d.fb_last[b]) ? (float)d.fb[b][k] : 0.f);
        }
    }
    // contiguous split of the bands over the 8 waves, balanced by (padded) weight count
    std::vector<int> gs(NWAVE + 1, d.B);
    {
        int total = 0;
        for (int b = 0; b < d.B; b++) total += bl[b] + 8;
        int acc = 0, g = 0;
        gs[0] = 0;
        for (int b = 0; b < d.B; b++) {
            const int cost = bl[b] + 8;
            while (g + 1 < NWAVE && acc + cost / 2 > (int64_t)total * (g + 1) / NWAVE) gs[++g] = b;
            acc += cost;
        }
        while (g + 1 <= NWAVE) gs[++g] = d.B;
    }
    e->dct_off = (int)ft.size();
    if (d.kind == ctu::FeaKind::Dctc) for (double v : d.dct) ft.push_back((float)v);
    else if (d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa) for (double v : d.idft) ft.push_back((float)v);
    e->lift_off = (int)ft.size();
    for (double v : d.lifter) ft.push_back((float)v);
    ft.push_back(0.f);
    e->ftab.upload(ft);
    std::vector<int> it;
    it.insert(it.end(), bf.begin(), bf.end());
    it.insert(it.end(), bl.begin(), bl.end());
    it.insert(it.end(), bo.begin(), bo.end());
    it.insert(it.end(), gs.begin(), gs.end());
    it.insert(it.end(), d.row_slot.begin(), d.row_slot.end());
    e->itab.upload(it);
    if (d.kind == ctu::FeaKind::TrapDct) {
        std::vector<float> g(d.trap.begin(), d.trap.end());
        e->trapG.upload(g);
    }
    switch (d.kind) {
        case ctu::FeaKind::Spec: e->feat = FEAT_SPEC; break;
        case ctu::FeaKind::LogSpec: e->feat = FEAT_LOGSPEC; break;
        case ctu::FeaKind::Dctc: e->feat = FEAT_DCTC; break;
        case ctu::FeaKind::Lpc: e->feat = FEAT_LPC; break;
        case ctu::FeaKind::Lpa: e->feat = FEAT_LPA; break;
        case ctu::FeaKind::TrapDct: e->feat = FEAT_LOGMEL_SCRATCH; break;
    }
    e->nz = (d.window + 31) / 32;
}

template <int NZ>
void launch_nz(int feat, dim3 grid, hipStream_t s, const KParams &kp) {
    const size_t shm = LDS_FLOATS * sizeof(float);
#define LAUNCH(F)                                                                                      \
    case F: {                                                                                          \
        static bool attr_set = false;                                                                  \
        if (!attr_set) {                                                                               \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&frontend_kernel<NZ, F>),       \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));        \
            attr_set = true;                                                                           \
        }                                                                                              \
        hipLaunchKernelGGL((frontend_kernel<NZ, F>), grid, dim3(WG), shm, s, kp);                      \
        break;                                                                                         \
    }
    switch (feat) {
        LAUNCH(FEAT_SPEC)
        LAUNCH(FEAT_LOGSPEC)
        LAUNCH(FEAT_DCTC)
        LAUNCH(FEAT_LPC)
        LAUNCH(FEAT_LPA)
        LAUNCH(FEAT_LOGMEL_SCRATCH)
    }
#undef LAUNCH
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
    std::vector<int4> tiles;
    std::vector<int> uts(n_utt + 1, 0);
    std::vector<int4> uinfo(n_utt);
    std::vector<int> chunks;
    const int trap_chunk = 8;
    for (int i = 0; i < n_utt; i++) {
        const int64_t T = ctu_num_frames(e, utt_nsamples[i]);
        if (T < 0) {
            set_error(e, "IO: Signal shorter than one frame!");  // src/io/in.cc:277
            return CTU_ERR_INPUT;
        }
        if (d.kind == ctu::FeaKind::TrapDct && T > 0 && T < (d.o.fea_trapdct_traplen + 1) / 2) {
            set_error(e, "ENGINE: trapdct on fewer than (traplen+1)/2 frames is undefined in the reference (src/fea/fea_trap.cc:64-70)");
            return CTU_ERR_INPUT;
        }
        pl->sample_off[i] = so;
        pl->row_off[i] = ro;
        pl->frames[i] = T;
        uts[i] = (int)tiles.size();
        for (int64_t t0 = 0; t0 < T; t0 += TILE) tiles.push_back(make_int4(i, (int)t0, (int)std::min<int64_t>(TILE, T - t0), 0));
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
    try {
        HIP_TRY(hipSetDevice(e->device));
        pl->tiles.upload(tiles);
        pl->d_sample_off.upload(pl->sample_off);
        pl->d_row_off.upload(pl->row_off);
        pl->utt_tile_start.upload(uts);
        if (d.kind == ctu::FeaKind::TrapDct) {
            pl->utt_info.upload(uinfo);
            pl->trap_chunks.upload(chunks);
            pl->n_trap_chunks = (int)chunks.size() / 2;
            if (e->logmel.n < (size_t)ro * d.B) e->logmel.alloc((size_t)ro * d.B);
        }
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
    (void)d_vad;
    if (pl->n_tiles == 0) return CTU_OK;
    if (!d_pcm || !d_rows) {
        set_error(e, "ENGINE: null device buffer");
        return CTU_ERR_INPUT;
    }
    const ctu::Design &d = *e->design;
    hipStream_t s = (hipStream_t)stream;
    try {
        HIP_TRY(hipSetDevice(e->device));
        KParams kp;
        std::memset(&kp, 0, sizeof kp);
        kp.pcm = d_pcm;
        kp.rows = d_rows;
        kp.logmel = e->logmel.p;
        kp.tiles = pl->tiles.p;
        kp.sample_off = pl->d_sample_off.p;
        kp.row_off = pl->d_row_off.p;
        kp.n_tiles = pl->n_tiles;
        kp.lanec = e->lanec.p;
        kp.ftab = e->ftab.p;
        kp.itab = e->itab.p;
        kp.wshift = d.wshift;
        kp.B = d.B;
        kp.nfea = d.nfea;
        kp.D = d.D;
        kp.ncep = d.o.fea_ncepcoefs;
        kp.lporder = d.o.fea_lporder;
        kp.dct_off = e->dct_off;
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
        kp.by_utt = kp.nr_exten;
        kp.utt_tile_start = pl->utt_tile_start.p;
        kp.n_utt = pl->n_utt;
        const int n_outer = kp.by_utt ? pl->n_utt : pl->n_tiles;
        const int grid = std::max(1, std::min(n_outer, e->n_cu * 2));
        HIP_TRY(hipEventRecord(e->ev0, s));
        switch (e->nz) {
            case 13: launch_nz<13>(e->feat, dim3(grid), s, kp); break;
            default: launch_nz<16>(e->feat, dim3(grid), s, kp); break;
        }
        HIP_TRY(hipEventRecord(e->ev1, s));
        e->timed = true;
        HIP_TRY(hipGetLastError());
        if (d.kind == ctu::FeaKind::TrapDct) {
            hipLaunchKernelGGL(trapdct_kernel, dim3(pl->n_trap_chunks), dim3(256), 0, s, e->logmel.p, d_rows, e->trapG.p,
                               pl->utt_info.p, pl->n_utt, d.B, d.o.fea_trapdct_traplen, d.o.fea_trapdct_ndct, d.D,
                               pl->trap_chunks.p, 8);
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
        pcm.alloc((size_t)pl->total_samples);
        rows.alloc((size_t)pl->total_frames * d.D);
        HIP_TRY(hipMemset(pcm.p, 0, (size_t)pl->total_samples * 2));
        HIP_TRY(hipMemcpy(pcm.p, h_pcm, (size_t)pl->total_samples * 2, hipMemcpyHostToDevice));
        int rc = ctu_engine_run(e, pl, pcm.p, rows.p, nullptr, nullptr);
        if (rc != CTU_OK) return rc;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(h_rows, rows.p, (size_t)pl->total_frames * d.D * 4, hipMemcpyDeviceToHost));
        (void)h_vad;
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    return CTU_OK;
}

float ctu_engine_last_kernel_ms(ctu_engine *e) {
    if (!e || !e->timed) return -1.f;
    float ms = -1.f;
    if (hipEventSynchronize(e->ev1) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, e->ev0, e->ev1) != hipSuccess) return -1.f;
    return ms;
}

}  // extern "C"
