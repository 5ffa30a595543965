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
constexpr int AUX_ROWS = 62;   // [AUX_ROWS][64] floats of per-band / per-coefficient staging
constexpr int LDS_FLOATS = TILE * PSTRIDE + AUX_ROWS * 64;
static_assert(LDS_FLOATS * 4 <= 80 * 1024, "two workgroups per CU need <= 80 KiB each");
constexpr int MAX_LP = 20;     // Levinson order limit of the in-register recursion
constexpr int PCM_ALIGN = 8;

enum FeatMode { FEAT_SPEC = 0, FEAT_LOGSPEC = 1, FEAT_DCTC = 2, FEAT_LPC = 3, FEAT_LPA = 4, FEAT_LOGMEL_SCRATCH = 5 };

struct KParams {
    const int16_t *pcm;
    float *rows;
    float *logmel;            // [total_frames][B] scratch (TRAP only)
    const int4 *tiles;
    const int64_t *sample_off;  // per utterance
    const int64_t *row_off;     // per utterance
    int n_tiles;
    // tables
    const float *win;         // [512] zero padded
    const float2 *tw1;        // [16][16]  W256^(n2*k1), index k1*16+n2
    const float2 *tw2;        // [129]     W512^k
    const int *band_first;    // [B] first bin (padded range)
    const int *band_len;      // [B] multiple of 4
    const int *band_off;      // [B] offset into fbw
    const float *fbw;         // packed band weights
    const int *grp_start;     // [NWAVE+1] bands of wave g: [grp_start[g], grp_start[g+1])
    const float *dct;         // [nfea][B]  (dctc)   or idft [(p+1)][B] (lpc)
    const float *lifter;      // [ncep]
    const int *row_slot;      // [nfea]
    // scalars
    int window, wshift, B, nfea, D, ncep, lporder;
    float preem;
    int remove_dc, fb_power, fb_inld, lifter_on;
    int nr_exten;
    float nr_p, nr_a;
    int by_utt;               // tiles are walked utterance by utterance (sequential state)
    const int *utt_tile_start;  // [n_utt+1] (by_utt)
    int n_utt;
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

    // ---- per-lane constants (live across tiles)
    float w0[NZ], w1[NZ];
#pragma unroll
    for (int j = 0; j < NZ; j++) {
        w0[j] = p.win[32 * j + 2 * l16];
        w1[j] = p.win[32 * j + 2 * l16 + 1];
    }
    float2 tw[16];
#pragma unroll
    for (int k1 = 1; k1 < 16; k1++) tw[k1] = p.tw1[k1 * 16 + l16];
    float2 ut[8];
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) ut[k2] = p.tw2[l16 + 16 * k2];
    const int partner = (lane & 48) | ((16 - l16) & 15);
    const float inv_window = 1.0f / (float)p.window;

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
                const int16_t *x = p.pcm + sbase + (int64_t)t * p.wshift;

                float2 v[16];
                float dc = 0.f;
#pragma unroll
                for (int j = 0; j < NZ; j++) {
                    const int i0 = 32 * j + 2 * l16;  // sample index of the even sample
                    float y0 = 0.f, y1 = 0.f;
                    if (i0 < p.window) {
                        const uint32_t pr = *reinterpret_cast<const uint32_t *>(x + i0);
                        const float x0 = (float)(int16_t)(pr & 0xffffu);
                        const float x1 = (float)(int16_t)(pr >> 16);
                        float xm = 0.f;
                        if (i0 > 0 || t > 0) xm = (float)x[i0 - 1];
                        y0 = w0[j] * (x0 - p.preem * xm);
                        y1 = w1[j] * (x1 - p.preem * x0);  // w1 is 0 beyond the window
                    }
                    v[j] = make_float2(y0, y1);
                    dc += y0 + y1;
                }
#pragma unroll
                for (int j = NZ; j < 16; j++) v[j] = make_float2(0.f, 0.f);
                if (p.remove_dc) {
                    // mean of the windowed frame over `window` samples (src/io/in.cc:375-382)
                    dc += __shfl_xor(dc, 8, 64);
                    dc += __shfl_xor(dc, 4, 64);
                    dc += __shfl_xor(dc, 2, 64);
                    dc += __shfl_xor(dc, 1, 64);
                    const float m = dc * inv_window;
#pragma unroll
                    for (int j = 0; j < NZ; j++) {
                        const int i0 = 32 * j + 2 * l16;
                        if (i0 < p.window) v[j].x -= m;
                        if (i0 + 1 < p.window) v[j].y -= m;
                    }
                }

                // ---- stage 1: DFT16 over n1 (registers), lane = n2; then twiddle W256^(n2*k1)
                dft16(v);
#pragma unroll
                for (int k1 = 1; k1 < 16; k1++) v[k1] = cmul(v[k1], tw[k1]);

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
                dft16(v);

                // ---- untangle the packed real FFT and take |.|^2.  Lane k1 handles its bins k2=0..7,
                //      each together with its mirror bin 256-k held by lane (16-k1)%16 in register 15-k2
                //      (register (16-k2)%16 for k1 = 0).
                float *prow = Pt + f * PSTRIDE;
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) {
                    float br = __shfl(v[15 - k2].x, partner, 64);
                    float bi = __shfl(v[15 - k2].y, partner, 64);
                    if (l16 == 0) {
                        br = v[(16 - k2) & 15].x;
                        bi = v[(16 - k2) & 15].y;
                    }
                    const float ar = v[k2].x, ai = v[k2].y;
                    const float sr = ar + br, si = ai - bi, dr = ar - br, di = ai + bi;
                    const float tr = ut[k2].x * di + ut[k2].y * dr;
                    const float ti = ut[k2].y * di - ut[k2].x * dr;
                    const float ur = sr + tr, ui = si + ti, vr = sr - tr, vi = si - ti;
                    float pk = 0.25f * (ur * ur + ui * ui);
                    float pm = 0.25f * (vr * vr + vi * vi);
                    const int k = l16 + 16 * k2;
                    if (k == 0 && p.remove_dc) pk = 1e-10f;  // src/io/in.cc:390
                    if (!p.fb_power) {
                        pk = sqrtf(pk);
                        pm = sqrtf(pm);
                    }
                    prow[k] = pk;
                    prow[256 - k] = pm;
                }
                if (l16 == 0) {  // bin 128 is its own mirror: X[128] = conj(Z[128])
                    float p128 = v[8].x * v[8].x + v[8].y * v[8].y;
                    if (!p.fb_power) p128 = sqrtf(p128);
                    prow[128] = p128;
                }
            }
            __syncthreads();

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
                const int b0 = p.grp_start[wave], b1 = p.grp_start[wave + 1];
                for (int b = b0; b < b1; b++) {
                    const int kf = p.band_first[b], len = p.band_len[b];
                    const float *w = p.fbw + p.band_off[b];
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

            if (FEAT == FEAT_SPEC || FEAT == FEAT_LOGSPEC || FEAT == FEAT_LOGMEL_SCRATCH) {
                float *dst = (FEAT == FEAT_LOGMEL_SCRATCH) ? p.logmel : p.rows;
                const int width = (FEAT == FEAT_LOGMEL_SCRATCH) ? p.B : p.D;
                // coalesced copy-out: consecutive threads write consecutive floats of the [nvalid][B] block
                for (int e = tid; e < nvalid * p.B; e += WG) {
                    const int f = e / p.B, b = e - f * p.B;
                    dst[(rbase + f) * width + b] = aux[b * 64 + f];
                }
            } else if (FEAT == FEAT_DCTC) {
                // c_i = sum_b dct[i][b] * logY[b]   (norm and lifter folded into the table)
                for (int i = wave; i < p.nfea; i += NWAVE) {
                    const float *d = p.dct + i * p.B;
                    float c = 0.f;
                    for (int b = 0; b < p.B; b++) c += d[b] * aux[b * 64 + lane];
                    const int slot = p.row_slot[i];
                    if (slot >= 0 && lane < nvalid) p.rows[(rbase + lane) * p.D + slot] = c;
                }
            } else {  // LPC / LPA
                // autocorrelation by cosine iDFT, k spread over the waves (src/fea/fea_impl.cc:181-198)
                float *R = aux + p.B * 64;
                for (int k = wave; k <= p.lporder; k += NWAVE) {
                    const float *d = p.dct + k * p.B;
                    float r = 0.f;
                    for (int b = 0; b < p.B; b++) {
                        float y = aux[b * 64 + lane];
                        if (!p.fb_inld) y *= y;  // src/fea/fea_impl.cc:165-169
                        r += d[b] * y;
                    }
                    R[k * 64 + lane] = r;
                }
                __syncthreads();
                if (wave == 0) {
                    // Levinson-Durbin in double (src/fea/fea_impl.cc:200-222), then a -> c (251-284)
                    const int P_ = p.lporder;
                    double a[MAX_LP + 1], aa[MAX_LP + 1], c[MAX_LP + 1];
                    const double r0 = R[lane];
                    double rc = -(double)R[64 + lane] / r0;
                    double err = r0 * (1 - rc * rc);
                    a[0] = aa[0] = 1;
                    a[1] = aa[1] = rc;
#pragma unroll
                    for (int ik = 2; ik <= MAX_LP; ik++) {
                        if (ik <= P_) {
                            double dm = R[ik * 64 + lane];
#pragma unroll
                            for (int n = 1; n < ik; n++) dm += aa[n] * (double)R[(ik - n) * 64 + lane];
                            rc = -dm / err;
                            a[ik] = rc;
#pragma unroll
                            for (int n = 1; n < ik; n++) a[n] = aa[n] + rc * aa[ik - n];
#pragma unroll
                            for (int n = 1; n <= ik; n++) aa[n] = a[n];
                            err *= (1 - rc * rc);
                        }
                    }
                    if (FEAT == FEAT_LPA) {
#pragma unroll
                        for (int i = 1; i <= MAX_LP; i++)
                            if (i <= P_ && lane < nvalid) p.rows[(rbase + lane) * p.D + (i - 1)] = (float)a[i];
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
                                if (n >= 1 && p.lifter_on) val *= (double)p.lifter[n - 1];
                                const int slot = p.row_slot[n];
                                if (slot >= 0 && lane < nvalid) p.rows[(rbase + lane) * p.D + slot] = (float)val;
                            }
                        }
                    }
                }
            }
            __syncthreads();  // P tile / aux are reused by the next tile
        }
    }
}

// TRAP-DCT (src/fea/fea_trap.cc:53-127): out[t][b*ndct+k] = sum_j G[k][j] * logmel[clamp(t-half+j)][b]
// with mean removal, Hamming and REDFT10 folded into G on the host.  One thread per (t, b).
__global__ void trapdct_kernel(const float *__restrict__ logmel, float *__restrict__ rows, const float *__restrict__ G,
                               const int4 *__restrict__ utt_info /* {row_off lo, row_off hi, T, -} */, int n_utt, int B,
                               int traplen, int ndct, int D, const int *__restrict__ utt_of_chunk, int chunk) {
    // grid.x = chunk of frames, threads = chunk*B laid out band-fastest
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
        for (int j = 0; j < traplen; j++) {
            int s = t - half + j;
            s = s < 0 ? 0 : (s > T - 1 ? T - 1 : s);
            const float x = logmel[(r0 + s) * B + b];
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
    DevBuf<float> win, fbw, dct, lifter, trapG;
    DevBuf<float2> tw1, tw2;
    DevBuf<int> band_first, band_len, band_off, grp_start, row_slot;
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
        if (d.B + o.fea_lporder + 1 > AUX_ROWS) return "filter bank + LP order too large for the LDS staging";
        if (o.fea_lporder > MAX_LP || o.fea_ncepcoefs > MAX_LP) return "LP order / cepstral order above the in-register limit";
    } else if (d.B > AUX_ROWS) return "more filter bank channels than the LDS staging holds";
    if (d.kind == ctu::FeaKind::TrapDct && o.fea_trapdct_ndct > 32) return "more than 32 TRAP DCT coefficients";
    return "";
}

void build_tables(ctu_engine *e) {
    const ctu::Design &d = *e->design;
    std::vector<float> win(512, 0.f);
    for (int i = 0; i < d.window; i++) win[i] = (float)d.hamming[i];
    e->win.upload(win);
    const double pi = 3.14159265358979323846;
    std::vector<float2> tw1(256), tw2(129);
    for (int k1 = 0; k1 < 16; k1++)
        for (int n2 = 0; n2 < 16; n2++) {
            const double a = -2 * pi * (double)(k1 * n2) / 256.0;
            tw1[k1 * 16 + n2] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int k = 0; k <= 128; k++) {
        const double a = -2 * pi * (double)k / 512.0;
        tw2[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    e->tw1.upload(tw1);
    e->tw2.upload(tw2);
    // banded filter bank: per band a run of weights padded to a multiple of 4 bins that stays inside [0,K)
    std::vector<int> bf(d.B), bl(d.B), bo(d.B);
    std::vector<float> w;
    for (int b = 0; b < d.B; b++) {
        int first = d.fb_first[b], len = d.fb_last[b] - d.fb_first[b] + 1;
        int plen = (len + 3) & ~3;
        if (first + plen > d.K) first = d.K - plen;
        if (first < 0) throw std::runtime_error("filter band wider than the spectrum");
        bf[b] = first;
        bl[b] = plen;
        bo[b] = (int)w.size();
        for (int i = 0; i < plen; i++) {
            const int k = first + i;
            w.push_back((k >= d.fb_first[b] && k <= d.fb_last[b]) ? (float)d.fb[b][k] : 0.f);
        }
    }
    e->band_first.upload(bf);
    e->band_len.upload(bl);
    e->band_off.upload(bo);
    e->fbw.upload(w);
    // contiguous split of the bands over the 8 waves, balanced by (padded) weight count
    std::vector<int> gs(NWAVE + 1, d.B);
    {
        int total = 0;
        for (int b = 0; b < d.B; b++) total += bl[b] + 8;
        int acc = 0, g = 0;
        gs[0] = 0;
        for (int b = 0; b < d.B; b++) {
            // start a new group when this band would overshoot the ideal boundary by more than half
            const int cost = bl[b] + 8;
            while (g + 1 < NWAVE && acc + cost / 2 > (int64_t)total * (g + 1) / NWAVE) gs[++g] = b;
            acc += cost;
        }
        while (g + 1 <= NWAVE) gs[++g] = d.B;
    }
    e->grp_start.upload(gs);
    std::vector<float> tab;
    if (d.kind == ctu::FeaKind::Dctc) tab.assign(d.dct.begin(), d.dct.end());
    else if (d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa) tab.assign(d.idft.begin(), d.idft.end());
    else tab.assign(1, 0.f);
    e->dct.upload(tab);
    std::vector<float> lf(d.lifter.begin(), d.lifter.end());
    if (lf.empty()) lf.push_back(1.f);
    e->lifter.upload(lf);
    e->row_slot.upload(d.row_slot);
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
    int64_t so = 0, ro = 0;
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
    pl->total_samples = so + PCM_ALIGN;  // tail pad: the last dword of an odd window may straddle the end
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
        kp.win = e->win.p;
        kp.tw1 = e->tw1.p;
        kp.tw2 = e->tw2.p;
        kp.band_first = e->band_first.p;
        kp.band_len = e->band_len.p;
        kp.band_off = e->band_off.p;
        kp.fbw = e->fbw.p;
        kp.grp_start = e->grp_start.p;
        kp.dct = e->dct.p;
        kp.lifter = e->lifter.p;
        kp.row_slot = e->row_slot.p;
        kp.window = d.window;
        kp.wshift = d.wshift;
        kp.B = d.B;
        kp.nfea = d.nfea;
        kp.D = d.D;
        kp.ncep = d.o.fea_ncepcoefs;
        kp.lporder = d.o.fea_lporder;
        kp.preem = d.o.preem;
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
        HIP_TRY(hipMemcpy(pcm.p, h_pcm, (size_t)(pl->total_samples - PCM_ALIGN) * 2, hipMemcpyHostToDevice));
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
