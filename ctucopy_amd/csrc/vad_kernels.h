// VAD module: Burg-cepstral criterion per frame, decision replay per utterance.
// Included by engine.hip (one translation unit: the kernels and their host launchers share types).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------------
// VAD module (src/vad/vad.cc, src/vad/vad.h, src/vdet/Burg.h).  Kernel A (vad_burg_kernel) is frame-parallel: HC2R of
// the post-NR spectrum with the original phase, Burg lattice, a -> c, in CTU_VAD_REAL arithmetic.  Kernel B
// (vad_decide_kernel) is one wave per utterance and replays the sequential, discontinuous part in double: cepstral
// distance to the adaptive background, threshold recurrences, background update, majority ("median") filter.
// ------------------------------------------------------------------------------------------------

// vad_burg_kernel: one wave per frame (4 waves per workgroup, persistent): the frame's samples live in registers,
// strided over the lanes (sample j = lane + 64 q); reductions are DPP row rotations + v_readlane (wave_sum_fast), no
// workgroup barrier inside the lattice.
#ifndef CTU_VAD_REAL
#define CTU_VAD_REAL float   // arithmetic of the HC2R + Burg kernel.  Its inputs (the front end's spectra) are float; on the
                             // reference recordings float and double give the same decisions, frame for frame, as the oracle
                             // (626 of 1186 on CS3 @ 8 kHz, the count the compiled reference wrote); double costs 1.6x
#endif
typedef CTU_VAD_REAL vreal;

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

// The sequential part of the VAD, one frame at a time, every lane of a wave in step (the state is wave-uniform except
// c0r, the background cepstrum, which lane i keeps for coefficient i): cepstral distance to the adaptive background
// (src/vad/vad.cc:220-294), the four threshold recurrences (:329-625), the majority filter (src/vad/vad.h:126-175).
struct VadRun {
    double crimin, crimax, crimean, crimean2, crivar, dmin, dmax;
    double c0r;  // background cepstrum, coefficient `lane`
    int adapt_vad;
    unsigned long long hist;  // the last `order` (<= 31) raw decisions as bits, with a running count
    int hidx, nout, nsum;
};

__device__ __forceinline__ void vad_run_reset(VadRun &r) {
    r.crimin = r.crimax = r.crimean = r.crimean2 = r.crivar = r.dmin = r.dmax = 0.0;
    r.c0r = 0.0;
    r.adapt_vad = 0;
    r.hist = 0;
    r.hidx = r.nout = r.nsum = 0;
}

__device__ __forceinline__ void vad_push(VadRun &r, int v, int order) {
    const int old = (int)((r.hist >> r.hidx) & 1ull);
    r.hist = (r.hist & ~(1ull << r.hidx)) | ((unsigned long long)v << r.hidx);
    r.nsum += v - old;
    r.hidx = (r.hidx + 1 == order) ? 0 : r.hidx + 1;
}

// Frame t of an utterance.  en: the energy criterion's value (cri 0); cil: coefficient `lane` of the frame's cepstrum
// (0 beyond nc).  out: the utterance's VAD bytes (lane 0 writes; decisions leave (order-1)/2 frames late).
__device__ __forceinline__ void vad_frame(VadRun &r, const VadParams &vp, int t, double en, double cil, int lane, uint8_t *out) {
    const int nc = vp.cri == 0 ? 1 : vp.ncoef, order = vp.filter_order, h = (order - 1) / 2;
    double cri;
    if (vp.cri == 0) {
        if (vp.energy_db) en = 10.0 * log10(2.2250738585072014e-308 + en);
        cri = en;
    } else {
        if (t == 0) {
            r.c0r = cil;
            cri = 0.0;
        } else {
            if (t == 1) r.c0r = (r.c0r + cil) / 2.0;
            const double dl = (lane >= 1 && lane < nc) ? cil - r.c0r : 0.0;  // c0 itself is not part of the distance
            cri = 4.3429 * sqrt(2 * wave_sum_fast(dl * dl));
        }
    }
    int vad0;
    if (vp.thr == 0) vad0 = cri >= vp.abs_thr;
    else if (vp.thr == 1) {
        if (t == 0 || (double)t < (double)vp.perc_init) r.crimin = r.crimax = cri;
        else {
            r.crimin = cri < r.crimin ? cri : r.crimin;
            r.crimax = cri > r.crimax ? cri : r.crimax;
        }
        vad0 = cri >= r.crimin + (vp.perc_thr / 100.0) * (r.crimax - r.crimin);
    } else if (vp.thr == 2) {
        if (t == 0) {
            r.crimean = cri;
            r.crimean2 = cri * cri;
            r.crivar = 0.0;
            r.adapt_vad = 0;
        } else {
            const double thr = r.crimean + vp.adapt_za * sqrt(r.crivar);
            if (cri < thr || t <= vp.adapt_init) {
                r.crimean = vp.adapt_q * r.crimean + (1.0 - vp.adapt_q) * cri;
                r.crimean2 = vp.adapt_q * r.crimean2 + (1.0 - vp.adapt_q) * cri * cri;
                r.crivar = r.crimean2 - r.crimean * r.crimean;
                r.adapt_vad = 0;
            } else r.adapt_vad = 1;
        }
        vad0 = r.adapt_vad;
    } else {
        const int init = vp.dyn_init > 1 ? vp.dyn_init : 1;
        if (t < init) {
            r.dmax = r.dmin = cri;
            vad0 = 0;
        } else if (t == init) {
            r.dmax = (cri > r.dmax ? cri : r.dmax) + vp.dyn_min / 10.0;
            r.dmin = (cri < r.dmin ? cri : r.dmin) - vp.dyn_min / 10.0;
            vad0 = 0;
        } else {
            r.dmax = r.dmax < cri ? vp.qmaxinc * r.dmax + (1.0 - vp.qmaxinc) * cri : vp.qmaxdec * r.dmax + (1.0 - vp.qmaxdec) * cri;
            r.dmin = r.dmin > cri ? vp.qmindec * r.dmin + (1.0 - vp.qmindec) * cri : vp.qmininc * r.dmin + (1.0 - vp.qmininc) * cri;
            const double dyn = r.dmax - r.dmin;
            vad0 = (cri > r.dmin + (vp.dyn_perc / 100.0) * dyn) && (dyn > vp.dyn_min);
        }
    }
    if (vp.cri != 0 && !(vad0 && t > vp.cep_init))  // background update (src/vad/vad.cc:288-294)
        r.c0r = vp.cep_p * r.c0r + (1.0 - vp.cep_p) * cil;
    vad_push(r, vad0, order);
    if (t >= h) {
        if (lane == 0) out[r.nout] = (2 * r.nsum >= order) ? '1' : '0';
        r.nout++;
    }
}

// (The majority test (double)nsum / order >= 0.5 of src/vad/vad.h:139-150 is taken as 2 nsum >= order: the same for integers.)
// End of an utterance of T frames: zeros are pushed until every frame has its byte (src/vad/vad.h:156-175).
__device__ __forceinline__ void vad_flush(VadRun &r, const VadParams &vp, int T, int lane, uint8_t *out) {
    const int order = vp.filter_order, h = (order - 1) / 2;
    for (int k = 0; k < h && r.nout < T; k++) {
        vad_push(r, 0, order);
        if (lane == 0) out[r.nout] = (2 * r.nsum >= order) ? '1' : '0';
        r.nout++;
    }
}

// The same recurrences for the fused Burg-cepstral path (vad_fused.h) with the utterances spread over the LANES: the front end
// leaves the cepstra of every frame in a scratch row (VFC_STRIDE floats) and a wave of this kernel walks 16 utterances at once, four
// lanes each - a lane keeps four coefficients of the background cepstrum and adds four terms to the distance (two quad_perm adds
// bring the four partial sums together, the same value in the four lanes); the scalar state is replicated in the quad and evolves
// identically.  Inside the front end the replay ran eight strictly sequential frames per wave step with all 64 lanes doing one
// utterance's scalar work - a quarter of that kernel (profiles/r02_c4_vf_stamps.txt).  The launch lasts as long as its longest
// utterance; everything but the distance's summation order is vad_frame / vad_flush statement for statement.
constexpr int VFC_STRIDE = 16;  // floats per frame in the cepstra scratch: the fused path's 14 coefficients, 64-byte rows

template <int NCL, int THR>  // cepstral coefficients (<= 16); threshold mode (vp.thr) at compile time: the launch lasts as long as its longest
                              // utterance's chain of frames, one wave per SIMD - every instruction of a frame's step is on the critical path
__global__ __launch_bounds__(64) void vad_lanes_kernel(const float *__restrict__ cf, const int *__restrict__ order, int n_live,
                                                        const int64_t *__restrict__ row_off, uint8_t *__restrict__ vad_out, VadParams vp) {
    static_assert(NCL <= 16, "four lanes x four coefficients");
    const int lane = threadIdx.x, pq = lane & 3;
    const int gidx = blockIdx.x * 16 + (lane >> 2);
    const bool live = gidx < n_live;
    const int u = live ? order[gidx] : 0;
    const int64_t r0 = live ? row_off[u] : 0;
    const int T = live ? (int)(row_off[u + 1] - r0) : 0;
    int Tmax = T;
#pragma unroll
    for (int off = 32; off >= 4; off >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, off, 64));
    const int order_f = vp.filter_order, h = (order_f - 1) / 2;
    double crimin = 0, crimax = 0, crimean = 0, crimean2 = 0, crivar = 0, dmin = 0, dmax = 0;
    double c0[4] = {0.0, 0.0, 0.0, 0.0};   // coefficients 4 pq .. 4 pq + 3 of the background cepstrum
    int adapt_vad = 0, hidx = 0, nout = 0, nsum = 0;
    unsigned long long hist = 0;
    uint8_t *out = vad_out + r0;
    auto push = [&](int v) {
        const int old = (int)((hist >> hidx) & 1ull);
        hist = (hist & ~(1ull << hidx)) | ((unsigned long long)v << hidx);
        nsum += v - old;
        hidx = (hidx + 1 == order_f) ? 0 : hidx + 1;
    };
    constexpr int AHEAD = 4;  // frames whose cepstra are in flight while the current ones are worked on (few waves per SIMD: no other cover)
    float4 q[AHEAD];
    auto fetch = [&](int slot, int t) {
        q[slot] = (t < T) ? reinterpret_cast<const float4 *>(cf + (r0 + t) * VFC_STRIDE)[pq] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
#pragma unroll
    for (int a = 0; a < AHEAD; a++) fetch(a, a);
    for (int tb = 0; tb < Tmax; tb += AHEAD) {
#pragma unroll
        for (int a = 0; a < AHEAD; a++) {
            const int t = tb + a;
            const double ci[4] = {(double)q[a].x, (double)q[a].y, (double)q[a].z, (double)q[a].w};
            fetch(a, t + AHEAD);
            // the quad's lanes run the same scalar code on the same values: a lane beyond its utterance's end just idles in step
            double cri;
            if (t == 0) {
#pragma unroll
                for (int i = 0; i < 4; i++) c0[i] = ci[i];
                cri = 0.0;
            } else {
                if (t == 1) {
#pragma unroll
                    for (int i = 0; i < 4; i++) c0[i] = (c0[i] + ci[i]) / 2.0;
                }
                double sum = 0.0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int k = 4 * pq + i;
                    // c0 itself is not part of the distance; vad_a2c_kernel stores zeros there and in the row's padding
                    const double dl = (CTU_VF_A2C || (k >= 1 && k < NCL)) ? ci[i] - c0[i] : 0.0;
                    sum += dl * dl;
                }
                sum += dpp_mov<0xB1>(sum);  // quad_perm [1,0,3,2]
                sum += dpp_mov<0x4E>(sum);  // quad_perm [2,3,0,1]
                cri = 4.3429 * sqrt(2 * sum);
            }
            {   // a quad beyond its utterance's end keeps stepping on zeros: its detector state is spent, only the filter and the stores below are guarded
                int vad0;
                if (THR == 0) vad0 = cri >= vp.abs_thr;
                else if (THR == 1) {
                    if (t == 0 || (double)t < (double)vp.perc_init) crimin = crimax = cri;
                    else {
                        crimin = cri < crimin ? cri : crimin;
                        crimax = cri > crimax ? cri : crimax;
                    }
                    vad0 = cri >= crimin + (vp.perc_thr / 100.0) * (crimax - crimin);
                } else if (THR == 2) {
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
                if (!(vad0 && t > vp.cep_init)) {  // background update (src/vad/vad.cc:288-294)
#pragma unroll
                    for (int i = 0; i < 4; i++) c0[i] = vp.cep_p * c0[i] + (1.0 - vp.cep_p) * ci[i];
                }
                if (t < T) {
                    push(vad0);
                    if (t >= h) {
                        if (pq == 0) out[nout] = (2 * nsum >= order_f) ? '1' : '0';
                        nout++;
                    }
                }
            }
        }
    }
    for (int k = 0; k < h && nout < T; k++) {  // end of the utterance: zeros until every frame has its byte (src/vad/vad.h:156-175)
        push(0);
        if (pq == 0) out[nout] = (2 * nsum >= order_f) ? '1' : '0';
        nout++;
    }
}

// One wave per utterance: stages 64 frames of criterion inputs in LDS with coalesced loads, then replays them.
__global__ __launch_bounds__(64) void vad_decide_kernel(const double *__restrict__ ci_all, const float *__restrict__ cri_energy,
                                                         float *__restrict__ rows, const int64_t *__restrict__ row_off,
                                                         int n_utt, uint8_t *__restrict__ vad_out, VadParams vp) {
    __shared__ double stage[64 * 32];
    const int u = blockIdx.x, lane = threadIdx.x;
    if (u >= n_utt) return;
    const int64_t r0 = row_off[u];
    const int T = (int)(row_off[u + 1] - r0);
    const int nc = vp.cri == 0 ? 1 : vp.ncoef;
    VadRun run;
    vad_run_reset(run);
    for (int tb = 0; tb < T; tb += 64) {
        const int nt = min(64, T - tb);
        __syncthreads();
        if (vp.cri == 0) {
            if (lane < nt) stage[lane] = cri_energy[r0 + min(tb + lane + vp.delay, T - 1)];
        } else if (vp.cri == 1) {
            for (int e = lane; e < nt * nc; e += 64) {
                const int f = e / nc, i = e - f * nc;
                stage[e] = ci_all[(r0 + min(tb + f + vp.delay, T - 1)) * nc + i];
            }
        } else {  // internal vector order: c0 first, then c1..cN (src/fea/fea_impl.cc:104-131).  Behind a delta / stacking chain the `fea`
                  // mode reads the vector OUT sees at this call - the chain's output, whose first block is the frame that comes out - so,
                  // unlike the two criteria above, it is not ahead of the row by the chain's delay
            for (int e = lane; e < nt * nc; e += 64) {
                const int f = e / nc, i = e - f * nc;
                const float *row = rows + (r0 + tb + f) * vp.D;
                // -fea_trap rows are the internal vector as it stands (the writers fall back to the straight copy, src/io/out.cc:182)
                stage[e] = vp.c0_slot == -2 ? (double)row[i] : (i == 0 ? (vp.c0_slot >= 0 ? (double)row[vp.c0_slot] : 0.0) : (double)row[i - 1]);
            }
        }
        __syncthreads();
        for (int tt = 0; tt < nt; tt++) {
            const double en = vp.cri == 0 ? stage[tt] : 0.0;
            const double cil = (vp.cri != 0 && lane < nc) ? stage[tt * nc + lane] : 0.0;
            vad_frame(run, vp, tb + tt, en, cil, lane, vad_out + r0);
        }
    }
    vad_flush(run, vp, T, lane, vad_out + r0);
    if (vp.e_slot >= 0 && vp.e_delay > 0) {  // energy column: forward shift in place, 64 rows at a time (reads run ahead of the writes)
        for (int tb = 0; tb < T; tb += 64) {
            const int j = tb + lane;
            float ev = 0.f;
            if (j < T) ev = rows[(r0 + min(j + vp.e_delay, T - 1)) * vp.D + vp.e_slot];
            __syncthreads();
            if (j < T) rows[(r0 + j) * vp.D + vp.e_slot] = ev;
            __syncthreads();
        }
    }
}

// Decisions of the files the majority filter never got `ready` on (src/vad/vad.h:126-136: no more frames than (order-1)/2): the
// reference writes nothing for them - their bytes become NUL.
__global__ void vad_short_files_kernel(unsigned char *vad, const int64_t *row_off, int n_utt, int delay) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_utt) return;
    const int64_t r0 = row_off[i], T = row_off[i + 1] - r0;
    if (T <= delay)
        for (int64_t t = 0; t < T; t++) vad[r0 + t] = 0;
}

// The reference's majority filter out of phase with its own ring (ctu_plan_set_vad_ring, engine.hip): row r of the output := the row
// src[r] of the rows as computed (-1: zeros), every column but the energy's.
__global__ void vad_ring_gather_kernel(const float *__restrict__ in, float *__restrict__ out, const int *__restrict__ src, int64_t total_rows, int D, int e_slot) {
    const int64_t n = total_rows * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D;
        const int c = (int)(i - r * D);
        if (c == e_slot) continue;
        const int sr = src[r];
        out[i] = sr >= 0 ? in[(int64_t)sr * D + c] : 0.f;
    }
}

}  // namespace
