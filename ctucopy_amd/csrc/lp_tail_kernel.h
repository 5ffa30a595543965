// LP tail: autocorrelation lags -> Levinson-Durbin -> prediction coefficients (lpa) or their cepstrum (lpc) -> row.
// Included by engine.hip (one translation unit: the kernels and their host launchers share types).
//
// The front end leaves R[0..p] of every frame in a scratch row (frontend_kernel.h).  The recursions of
// src/fea/fea_impl.cc:200-222 (LevDurb; the reference's aa[] copy is replaced by the in-place symmetric update, same
// operations) and :251-284 (a -> c) are short and strictly sequential per frame, so they run here with ONE FRAME PER LANE:
// inside the front end the eight lanes of a frame would each repeat them, for eight frames per wave step.
//   T = float: band energies compressed by the intensity-loudness law (-fb_inld, PLP): the autocorrelation matrix is well
//       conditioned, measured deviation from a double recursion ~1e-6 (tests/test_gpu_parity.py::test_c3_plp);
//   T = double: LP analysis on uncompressed band energies - the lags are nearly equal and fp32 recursions lose the result.
#pragma once

namespace {

struct LpTailParams {
    const void *lags;     // [total_frames][stride] of T
    float *rows;          // [total_frames][D]
    const float *lifter;  // [ncep] (lifter_on)
    const int *row_slot;  // [ncep + 1]: output slot of c0, c1 .. (-1: not written)
    int64_t total_frames;
    int stride, D, lporder, ncep, is_lpa, lifter_on, e_mode, e_slot;
    unsigned inv_stride, inv_D;  // floor(2^32 / n) + 1: e / n = __umulhi(e, inv) for the e < 2^13 of a 256-frame block
};

template <class T, int LPO>  // LPO: order = number of cepstra fixed at compile time (12: the PLP preset), 0 = run-time up to MAX_LP
__global__ __launch_bounds__(256) void lp_tail_kernel(const LpTailParams p) {
    constexpr bool DBL = sizeof(T) == 8;
    constexpr int PM = LPO ? LPO : MAX_LP;
    const int P_ = LPO ? LPO : p.lporder, ncep_ = LPO ? LPO : p.ncep;
    ci32 *row_slot = as_const(p.row_slot);
    cf32 *lifter = as_const(p.lifter);
    // 256 frames per block pass.  Lags and rows are short records (13 values): a lane reading its own record straight from
    // HBM touches a cache line per value, so both go through LDS - coalesced in, one record per lane (odd strides: no bank
    // conflicts; even ones are padded by one), coalesced out.  Rows are read first: the front end may have written an energy
    // column there already (every -fea_E routing but ln R[0]).
    extern __shared__ __align__(16) unsigned char lp_lds[];
    const int ls = p.stride | 1, rs = p.D | 1;
    T *lag_s = reinterpret_cast<T *>(lp_lds);
    float *row_s = reinterpret_cast<float *>(lag_s + 256 * ls);
    const int tid = threadIdx.x;
    for (int64_t f0 = (int64_t)blockIdx.x * 256; f0 < p.total_frames; f0 += (int64_t)gridDim.x * 256) {
        const int nf = (int)min((int64_t)256, p.total_frames - f0);
        const T *lg = reinterpret_cast<const T *>(p.lags) + f0 * p.stride;
        float *rg = p.rows + f0 * p.D;
        __syncthreads();
        for (int e = tid; e < nf * p.stride; e += 256) {
            const int q = (int)__umulhi((unsigned)e, p.inv_stride);
            lag_s[q * ls + (e - q * p.stride)] = lg[e];
        }
        // the front end has written an energy column already with every -fea_E routing but ln R[0] (e_mode 2): keep it
        const bool keep = p.e_mode != 0 && p.e_mode != 2;
        for (int e = tid; e < nf * p.D; e += 256) {
            const int q = (int)__umulhi((unsigned)e, p.inv_D);
            row_s[q * rs + (e - q * p.D)] = keep ? rg[e] : 0.f;
        }
        __syncthreads();
        if (tid < nf) {
        const T *r = lag_s + tid * ls;
        float *orow = row_s + tid * rs;
        T c[PM + 1];
#pragma unroll
        for (int k = 0; k <= PM; k++) c[k] = k <= P_ ? r[k] : (T)0;
        T a[PM + 1], cc[PM + 1];
        const T r0 = c[0];
        if (p.e_mode == 2) orow[p.e_slot] = DBL ? (float)log((double)r0) : __builtin_amdgcn_logf((float)r0) * 0.69314718056f;  // E = ln R[0] (src/fea/fea_impl.cc:177)
        T rc = -c[1] / r0;
        T err = r0 * (1 - rc * rc);
        a[0] = 1;
        a[1] = rc;
#pragma unroll
        for (int ik = 2; ik <= PM; ik++) {
            if (ik <= P_) {
                T dm = c[ik];
#pragma unroll
                for (int n = 1; n < ik; n++) dm += a[n] * c[ik - n];
                rc = -dm / err;
#pragma unroll
                for (int n = 1; n <= ik / 2; n++) {
                    const T lo = a[n], hi = a[ik - n];
                    a[n] = lo + rc * hi;
                    if (n != ik - n) a[ik - n] = hi + rc * lo;
                }
                a[ik] = rc;
                err *= (1 - rc * rc);
            }
        }
        if (p.is_lpa) {
#pragma unroll
            for (int i = 1; i <= PM; i++)
                if (i <= P_) orow[i - 1] = (float)a[i];
        } else {
            cc[0] = DBL ? (T)log((double)err) : (T)(__builtin_amdgcn_logf((float)err) * 0.69314718056f);
#pragma unroll
            for (int n = 1; n <= PM; n++) {
                if (n <= ncep_) {
                    T sum = 0;
#pragma unroll
                    for (int k = 1; k < n; k++)
                        if (k <= P_) sum += (T)(n - k) * cc[n - k] * a[k];
                    cc[n] = (n <= P_ ? -a[n] : (T)0) - sum / (T)n;
                }
            }
#pragma unroll
            for (int n = 0; n <= PM; n++) {
                if (n <= ncep_) {
                    float val = (float)cc[n];
                    if (n >= 1 && p.lifter_on) val = (float)(cc[n] * (T)lifter[n - 1]);
                    const int slot = row_slot[n];
                    if (slot >= 0) orow[slot] = val;
                }
            }
        }
        }
        __syncthreads();
        for (int e = tid; e < nf * p.D; e += 256) {
            const int q = (int)__umulhi((unsigned)e, p.inv_D);
            rg[e] = row_s[q * rs + (e - q * p.D)];
        }
    }
}

}  // namespace
