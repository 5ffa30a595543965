// Constants, kernel parameter block and device helpers shared by the kernels.
// Included by engine.hip (one translation unit: the kernels and their host launchers share types).
#pragma once

namespace {

constexpr int TILE = 64;       // frames per tile (= lanes of the per-frame phase)
constexpr int WG = 512;        // threads per workgroup (8 waves)
constexpr int NWAVE = WG / 64;
constexpr int PSTRIDE = 260;   // floats per P-tile row: 257 bins padded so rows stay 16-byte aligned (b128 reads in phase 2)
constexpr int LDS_2WG = 80 * 1024;  // two workgroups per CU fit when a workgroup's LDS stays at or under this
constexpr int MAX_LP = 23;     // LP order / cepstral order limit: the front end accumulates MAXC = 24 lags (R[0..23]) per frame, lp_tail_kernel keeps one frame's recursion in a lane's registers
constexpr int GEN_PLAIN = 0, GEN_INLD = 1, GEN_EXTEN = 2, GEN_FULL = 3, GEN_DC1 = 4;  // front-end option specialisations (frontend_kernel.h)
constexpr int MAXC = 24;       // most coefficients accumulated per frame in phase 2 (cepstra incl. c0, or LP lags)
constexpr int PCM_ALIGN = 8;   // utterance starts are multiples of this many samples
constexpr int PCM_HEAD = 8;    // samples of padding before the first utterance (x[-2..-1] of frame 0 is loaded)
constexpr int PCM_TAIL = 512;  // padding after the last one: the generic instantiation loads 16 rows of 32 samples whatever the window

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
enum FeatMode { FEAT_BANDS = 0, FEAT_DCTC = 2, FEAT_LP = 3, FEAT_LPD = 4 };  // LPD: LP analysis with the autocorrelation and the recursions in double

// VAD module parameters (src/vad/vad.cc, src/vad/vad.h; used by vad_kernels.h and by the fused path of the front end)
struct VadParams {
    int K, wfft, window, ncoef;  // ncoef = vad_lpc_coefs (cepdist lpc) or feature vector length (cepdist fea)
    int cri;                     // 0 energy, 1 cepdist-lpc, 2 cepdist-fea
    int thr;                     // 0 absolute, 1 perc, 2 adapt, 3 dyn
    int energy_db, cep_init, filter_order;
    double cep_p, abs_thr, perc_thr, adapt_q, adapt_za, dyn_perc, dyn_min, qmaxinc, qmaxdec, qmindec, qmininc;
    int perc_init, adapt_init, dyn_init;
    int D, ncep, c0_slot;        // cepdist-fea: where the internal vector sits in a written row
    int delay;                   // delta / stacking ahead of the writer: the detector is called when a delayed vector comes out, on the
                                 // criterion of the newest input frame min(call + delay, T - 1) (src/io/batch.cc:172-192,230-241,251-291)
    int e_slot, e_delay;         // -fea_E: the writer reads the energy through a pointer when the median filter releases a
                                 // vector, so row j carries the energy of row min(j + e_delay, T - 1); e_slot < 0: no such column
};

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
    // [NS + 1][8] records of four ints at ck_off | per-cell coefficient rows [NS][8][CW] at cf_off
    int tab_floats, ck_off, cf_off, NS, CW;
    int cfd_off;                // FEAT_LPD: per-cell coefficient rows in double, [NS][8][CW] doubles (8-byte aligned)
    int ncoef_out;              // DCTC: coefficients written per row (table rows are in output order)
    int e_mode, e_slot, K, window;  // -fea_E: 0 none, 1 spectrum (nr->E), 2 log R[0], 3 band energy, 4 raw frame energy
    int wshift, B, nfea, D, ncep, lporder;
    int lift_off;
    float preem, inv_window;
    int remove_dc, fb_power, fb_inld, lifter_on, nr_exten;
    int kstride;                // FFT sizes below 256 run as the 256-point mode: their bin k is bin k * kstride of it
    int remove_dc1, dc1_J;      // -remove_dc1: offsets of the frames (dc1, one per frame) and floor(window / wshift) <= 8
    const float *dc1;
    float nr_p, nr_a;
    unsigned long long *stamps;  // [grid][NWAVE][16] (CTU_STAMP builds)
    int skip_phase2;  // signal output (row N3): spectra are exported, nothing is projected
    int nr_after_fb;  // -nr_when afterFB: exten runs on the band energies inside phase 2 (GEN_FULL)
    int per_wave;     // chains per wave (wg_first has grid * NWAVE entries) instead of per workgroup
    int am_off;       // MD instantiations: A operands of the DCT MFMAs [2 * NS][64] in the LDS tables
    double nr_p_d;    // exten smoothing constant in double
    double inv_window_d;
    double *vad_ci;   // VF instantiations: [total_frames][vad_nc] Burg cepstra for the decision replay
    int vad_nc;       // cepstral coefficients of the Burg criterion (vad_lpc_coefs)
    // SS instantiations (hwss / fwss / 2fwss): mode 1 / 2 / 3, oversubtraction b, initial noise-only frames, the detector's q,
    // the noise seeds of the utterances (the previous file's last vector) and where each utterance leaves its own,
    // the utterance of every tile, and the detector's Hann window [208] in the LDS tables
    int ss_mode, ss_init, han_off;
    int ss_nc;                  // cepstral coefficients of the *ss modes' detector (-fea_ncepcoefs, src/nr/nr.cc:263-276): 2 .. SS_NC
    float nr_b;
    double ss_q;
    const float *ss_seed;
    float *ss_last;
    const int *tile_utt;
    const unsigned char *ss_dirty;  // [n_utt] or null: utterances to (re)compute in this pass of the seed iteration
    unsigned char *ss_vbits;        // [total_frames]: the detector's decision of every frame, written by the first pass
    int ss_cached;                  // later passes: take the decisions from ss_vbits (they do not depend on the noise seed)
    float *ybuf;      // SY instantiations: time-domain frames [total_frames][window] ahead of the overlap-add
    float syn_scale;  // SY: 1 / wfft (sigOUT's amplitude factor, src/io/out.cc:416-422)
    float *vad_cf;    // VF instantiations: Burg cepstra of every frame [total_frames][VFC_STRIDE] for vad_lanes_kernel
    void *lp_r;       // LP kinds: autocorrelation lags of every frame [total_frames][lp_stride] (float; double for FEAT_LPD), finished by lp_tail_kernel
    int lp_stride;
    int dbg;  // diagnostic ablation (CTU_DEBUG_MODE): 1 = phase 1 only, 2 = phase 2 only; 0 in production
};

#ifndef CTU_VF_A2C
#define CTU_VF_A2C 1  // fused Burg-cepstral criterion: 1 = the front end stops at the lattice and vad_a2c_kernel finishes the cepstra (vad_fused.h); 0: the tail stays inside the front end (A/B)
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));  // MFMA accumulator
typedef int int32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
// A pointer that is the same in every lane, pinned to SGPRs: stores through it take the scalar-base + 32-bit lane-offset form, so the
// lane part is one VGPR of cheap arithmetic instead of a 64-bit product per lane (and nothing the register allocator has to spill).
template <class T>
__device__ __forceinline__ __attribute__((address_space(1))) T *uniform_ptr(T *p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (__attribute__((address_space(1))) T *)(((unsigned long long)hi << 32) | lo);
}

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

// Two-lane packed arithmetic (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32): the two passes of a step run in lock step, so
// pass A's value and pass B's value of every quantity share a 64-bit register pair and every add / multiply / FMA of the
// transform serves both.  Element-wise only (no swizzles between the halves: the compiler would spend moves on them);
// scalars - twiddles, window weights - are broadcast by op_sel.  A packed instruction issues in ~1.7x the time of a plain
// one (tools/probes/valu_rate.hip) for twice the work.
typedef float v2f __attribute__((ext_vector_type(2)));
struct cx2 {
    v2f x, y;  // real parts of (pass A, pass B), imaginary parts of (pass A, pass B)
};
__device__ __forceinline__ cx2 cmul_s(cx2 a, float wr, float wi) {  // both passes times the same twiddle
    return cx2{a.x * wr - a.y * wi, a.x * wi + a.y * wr};
}
__device__ __forceinline__ void bfly4(cx2 &p0, cx2 &p1, cx2 &p2, cx2 &p3) {
    const cx2 s02{p0.x + p2.x, p0.y + p2.y}, d02{p0.x - p2.x, p0.y - p2.y};
    const cx2 s13{p1.x + p3.x, p1.y + p3.y}, d13{p1.x - p3.x, p1.y - p3.y};
    p0 = cx2{s02.x + s13.x, s02.y + s13.y};
    p2 = cx2{s02.x - s13.x, s02.y - s13.y};
    p1 = cx2{d02.x + d13.y, d02.y - d13.x};  // d02 - i*d13
    p3 = cx2{d02.x - d13.y, d02.y + d13.x};  // d02 + i*d13
}
__device__ __forceinline__ void dft16(cx2 (&v)[16]) {  // as dft16(float2[16]) above, statement for statement
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
    for (int b = 0; b < 4; b++) bfly4(v[b], v[4 + b], v[8 + b], v[12 + b]);
    v[5] = cmul_s(v[5], C1, -S1);
    v[6] = cx2{(v[6].x + v[6].y) * R2, (v[6].y - v[6].x) * R2};
    v[7] = cmul_s(v[7], S1, -C1);
    v[9] = cx2{(v[9].x + v[9].y) * R2, (v[9].y - v[9].x) * R2};
    v[10] = cx2{v[10].y, -v[10].x};
    v[11] = cx2{(v[11].y - v[11].x) * R2, -(v[11].x + v[11].y) * R2};
    v[13] = cmul_s(v[13], S1, -C1);
    v[14] = cx2{(v[14].y - v[14].x) * R2, -(v[14].x + v[14].y) * R2};
    v[15] = cmul_s(v[15], -C1, S1);
#pragma unroll
    for (int c = 0; c < 4; c++) bfly4(v[4 * c + 0], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    cx2 t[16];
#pragma unroll
    for (int k = 0; k < 16; k++) t[k] = v[4 * (k & 3) + (k >> 2)];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = t[k];
}

// LDS stores of one dword per lane at  M0 + offset + 4 * lane  (ds_write_addtid_b32): linear in the lane number, which is
// what the transposes write (16 k1-rows of [frame slot][n2], 260 bytes apart).  M0 is compiler-reserved (it may hold an
// offset of the compiler's own, e.g. for register spills) and an asm statement's write to it is invisible to the compiler:
// the statement saves M0, writes the 16 rows and restores it.  One wait state between the SALU write of M0 and an LDS
// add-TID instruction (ISA manual, required software nops).
#define CTU_ADDTID16(V, C, BASE)                                                                                                     \
    do {                                                                                                                             \
        uint32_t keep_;                                                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %17\n\ts_nop 0\n\t"                                                         \
                     "ds_write_addtid_b32 %1 offset:0\n\tds_write_addtid_b32 %2 offset:260\n\tds_write_addtid_b32 %3 offset:520\n\t"  \
                     "ds_write_addtid_b32 %4 offset:780\n\tds_write_addtid_b32 %5 offset:1040\n\tds_write_addtid_b32 %6 offset:1300\n\t" \
                     "ds_write_addtid_b32 %7 offset:1560\n\tds_write_addtid_b32 %8 offset:1820\n\tds_write_addtid_b32 %9 offset:2080\n\t" \
                     "ds_write_addtid_b32 %10 offset:2340\n\tds_write_addtid_b32 %11 offset:2600\n\tds_write_addtid_b32 %12 offset:2860\n\t" \
                     "ds_write_addtid_b32 %13 offset:3120\n\tds_write_addtid_b32 %14 offset:3380\n\tds_write_addtid_b32 %15 offset:3640\n\t" \
                     "ds_write_addtid_b32 %16 offset:3900\n\ts_mov_b32 m0, %0"                                                        \
                     : "=&s"(keep_)                                                                                                  \
                     : "v"(V[0].C), "v"(V[1].C), "v"(V[2].C), "v"(V[3].C), "v"(V[4].C), "v"(V[5].C), "v"(V[6].C), "v"(V[7].C),      \
                       "v"(V[8].C), "v"(V[9].C), "v"(V[10].C), "v"(V[11].C), "v"(V[12].C), "v"(V[13].C), "v"(V[14].C),              \
                       "v"(V[15].C), "s"(BASE)                                                                                      \
                     : "memory");                                                                                                    \
    } while (0)

// The same for sixteen arbitrary float expressions E(i), i = 0..15 (the halves of packed pairs), rows 516 bytes apart
// starting OFF bytes behind BASE: the layout of the packed transpose (frontend_kernel.h, CTU_PK) - row k1 of pass A at
// dword 129 k1, of pass B at 129 k1 + 64, so that one ds_read2_b32 (offset0 = n2, offset1 = 64 + n2) returns the
// (pass A, pass B) pair of an element into one register pair, and lane (k1, fg) reading dword 129 k1 + 16 fg + n2 meets
// bank (k1 + 16 fg + n2) mod 32: distinct over each half wave.  16 x 129 dwords fit the wave's eight P rows (2080).
#define CTU_ADDTID16_P(E, BASE, OFF)                                                                                                 \
    do {                                                                                                                             \
        uint32_t keep_;                                                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %17\n\ts_nop 0\n\t"                                                         \
                     "ds_write_addtid_b32 %1 offset:%c18\n\tds_write_addtid_b32 %2 offset:516+%c18\n\tds_write_addtid_b32 %3 offset:1032+%c18\n\t"  \
                     "ds_write_addtid_b32 %4 offset:1548+%c18\n\tds_write_addtid_b32 %5 offset:2064+%c18\n\tds_write_addtid_b32 %6 offset:2580+%c18\n\t" \
                     "ds_write_addtid_b32 %7 offset:3096+%c18\n\tds_write_addtid_b32 %8 offset:3612+%c18\n\tds_write_addtid_b32 %9 offset:4128+%c18\n\t" \
                     "ds_write_addtid_b32 %10 offset:4644+%c18\n\tds_write_addtid_b32 %11 offset:5160+%c18\n\tds_write_addtid_b32 %12 offset:5676+%c18\n\t" \
                     "ds_write_addtid_b32 %13 offset:6192+%c18\n\tds_write_addtid_b32 %14 offset:6708+%c18\n\tds_write_addtid_b32 %15 offset:7224+%c18\n\t" \
                     "ds_write_addtid_b32 %16 offset:7740+%c18\n\ts_mov_b32 m0, %0"                                                        \
                     : "=&s"(keep_)                                                                                                  \
                     : "v"(E(0)), "v"(E(1)), "v"(E(2)), "v"(E(3)), "v"(E(4)), "v"(E(5)), "v"(E(6)), "v"(E(7)), "v"(E(8)), "v"(E(9)),   \
                       "v"(E(10)), "v"(E(11)), "v"(E(12)), "v"(E(13)), "v"(E(14)), "v"(E(15)), "s"(BASE), "n"(OFF)                    \
                     : "memory");                                                                                                    \
    } while (0)

// The sixteen (pass A, pass B) pairs of one component back from that layout: ds_read2_b32 with offset0 = n2, offset1 = 64 + n2
// fills a register pair with the two passes' values (the compiler's own pairing of ds_read_b32s goes by adjacent addresses,
// which would pair n2 with n2 + 1 and cost a move per value).  Inline assembly is invisible to the compiler's wait-count
// insertion: the statement waits for its own reads.
#define CTU_READ2_PAIRS16(R, ADDR)                                                                                                   \
    asm volatile("ds_read2_b32 %0, %16 offset0:0 offset1:64\n\tds_read2_b32 %1, %16 offset0:1 offset1:65\n\t"                        \
                 "ds_read2_b32 %2, %16 offset0:2 offset1:66\n\tds_read2_b32 %3, %16 offset0:3 offset1:67\n\t"                        \
                 "ds_read2_b32 %4, %16 offset0:4 offset1:68\n\tds_read2_b32 %5, %16 offset0:5 offset1:69\n\t"                        \
                 "ds_read2_b32 %6, %16 offset0:6 offset1:70\n\tds_read2_b32 %7, %16 offset0:7 offset1:71\n\t"                        \
                 "ds_read2_b32 %8, %16 offset0:8 offset1:72\n\tds_read2_b32 %9, %16 offset0:9 offset1:73\n\t"                        \
                 "ds_read2_b32 %10, %16 offset0:10 offset1:74\n\tds_read2_b32 %11, %16 offset0:11 offset1:75\n\t"                    \
                 "ds_read2_b32 %12, %16 offset0:12 offset1:76\n\tds_read2_b32 %13, %16 offset0:13 offset1:77\n\t"                    \
                 "ds_read2_b32 %14, %16 offset0:14 offset1:78\n\tds_read2_b32 %15, %16 offset0:15 offset1:79\n\t"                    \
                 "s_waitcnt lgkmcnt(0)"                                                                                              \
                 : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3]), "=&v"(R[4]), "=&v"(R[5]), "=&v"(R[6]), "=&v"(R[7]), "=&v"(R[8]), \
                   "=&v"(R[9]), "=&v"(R[10]), "=&v"(R[11]), "=&v"(R[12]), "=&v"(R[13]), "=&v"(R[14]), "=&v"(R[15])                       \
                 : "v"(ADDR)                                                                                                         \
                 : "memory")

// Transpose of 16 x 16 complex values between "lane" and "register" inside each 16-lane group of a wave, through an LDS
// scratch of 16 x 65 dwords (re, then im): element (k1, n2) of group fg sits at dword 65 k1 + 16 fg + n2 = 65 k1 + lane.
// The stores are linear in the lane (ds_write_addtid_b32, no address register, 2 LDS cycles); lane k1 reads 65 k1 + 16 fg +
// n2, n2 = 0..15, with immediate offsets: banks (k1 + 16 fg + n2) mod 32, distinct over each half wave.
//   sbase = LDS byte address of the scratch (wave-uniform), rd = scratch + 65 * (lane & 15) + 16 * (lane >> 4).
__device__ __forceinline__ void wave_transpose16(float2 (&v)[16], uint32_t sbase, const float *rd) {
    sbase = __builtin_amdgcn_readfirstlane(sbase);  // wave-uniform by construction; this makes it provably so ("s" operand)
    __builtin_amdgcn_wave_barrier();
    CTU_ADDTID16(v, x, sbase);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float re[16];
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) re[n2] = rd[n2];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    CTU_ADDTID16(v, y, sbase);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) v[n2] = make_float2(re[n2], rd[n2]);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Two such transposes at once (two independent sets of 16 x 16 values, scratch areas sa / sb): the stores of both go out
// before the reads of either, so the LDS round trips overlap.
__device__ __forceinline__ void wave_transpose16_dual(float2 (&va)[16], float2 (&vb)[16], uint32_t sa, uint32_t sb, const float *rda,
                                                      const float *rdb) {
    sa = __builtin_amdgcn_readfirstlane(sa);
    sb = __builtin_amdgcn_readfirstlane(sb);
    __builtin_amdgcn_wave_barrier();
    CTU_ADDTID16(va, x, sa);
    CTU_ADDTID16(vb, x, sb);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float rea[16], reb[16];
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) rea[n2] = rda[n2];
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) reb[n2] = rdb[n2];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#if defined(CTU_ABL) && (CTU_ABL & 4)  // diagnostic: no imaginary transpose (wrong results)
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) {
        va[n2] = make_float2(rea[n2], va[n2].y);
        vb[n2] = make_float2(reb[n2], vb[n2].y);
    }
    return;
#endif
    CTU_ADDTID16(va, y, sa);
    CTU_ADDTID16(vb, y, sb);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) va[n2] = make_float2(rea[n2], rda[n2]);
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) vb[n2] = make_float2(reb[n2], rdb[n2]);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
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

// Row rotation by DPP for 32- and 64-bit values (v_mov_b32_dpp per half), and a wave all-reduce built on it: four
// rotate-and-add steps inside each row of 16 lanes, then the four row sums through v_readlane.  A shuffle-based
// butterfly (ds_bpermute) costs an LDS round trip per step; the lattice below runs two reductions per order.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
// The mirror-bin partner of the untangle steps: lane l of a 16-lane row reads lane (16 - l) & 15.  CTU_DPPMIRROR = 1:
// row_mirror then row_ror:1 on the VALU (lane l <- mirror lane l-1 = original lane 16-l), no LDS crossbar cycles;
// 0: ds_bpermute (`partner` = byte address of the source lane).
#ifndef CTU_DPPMIRROR
#define CTU_DPPMIRROR 0
#endif
__device__ __forceinline__ float mirror_fetch(float x, int partner) {
#if CTU_DPPMIRROR
    (void)partner;
    return dpp_mov<0x121>(dpp_mov<0x140>(x));
#else
    return __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(x)));
#endif
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// the same 8-lane sum for doubles (energy columns: squares of squares leave the float range on silent frames)
__device__ __forceinline__ double lanes8_allreduce_add(double x) {
    x += dpp_mov<0xB1>(x);
    x += dpp_mov<0x4E>(x);
    x += dpp_mov<0x141>(x);
    return x;
}


struct __attribute__((packed, aligned(2))) pcm2 {  // two consecutive int16 samples at any sample address
    uint32_t v;
};
struct __attribute__((packed, aligned(2))) pcm4 {  // four consecutive int16 samples at any sample address: one global_load_dwordx2 either
    uint32_t lo, hi;                                // way (global memory takes unaligned accesses), so odd frame shifts cost nothing
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
__device__ __forceinline__ void cell_accumulate(double (&c)[NCW], const double *cf, double y) {
#pragma unroll
    for (int i = 0; i < NCW; i++) c[i] += cf[i] * y;
}
template <int NCW>
__device__ __forceinline__ void cells_reduce(double (&c)[NCW]) {
#pragma unroll
    for (int i = 0; i < NCW; i++) c[i] = lanes8_allreduce_add(c[i]);
}
template <int NCW>
__device__ __forceinline__ void cells_reduce(float (&c)[NCW]) {
#pragma unroll
    for (int i = 0; i < NCW; i++) c[i] = lanes8_allreduce_add(c[i]);
}

// Tile record (32 bytes, read with one scalar load): where the tile's first frame starts in the PCM
// arena, where its first output row goes, how many of its 64 frame slots are real, the frame index of
// slot 0 inside its utterance, the next tile this workgroup / wave walks (-1 = done), the utterance's frame count.
struct TileRec {
    int64_t sbase, rbase;
    int nvalid, t0, next, T;  // T = frames of the tile's utterance
};

__device__ __forceinline__ TileRec load_rec(const TileRec *tiles, int tile) {
    ci32 *w = as_const(reinterpret_cast<const int *>(tiles)) + 8 * tile;
    TileRec r;
    r.sbase = ((int64_t)w[1] << 32) | (uint32_t)w[0];
    r.rbase = ((int64_t)w[3] << 32) | (uint32_t)w[2];
    r.nvalid = w[4];
    r.t0 = w[5];
    r.next = w[6];
    r.T = w[7];
    return r;
}

}  // namespace
