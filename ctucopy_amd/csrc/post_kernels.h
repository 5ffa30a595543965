// Rows N1 / N2: delta chain and stacking, CMS, per-speaker CMVN over resident rows.
// Included by engine.hip (one translation unit: the kernels and their host launchers share types).
#pragma once

namespace {

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

template <bool STD>  // STD: the MFCC_0_D_A layout (13 -> 39 floats, windows 2 / 2) with every size a constant
__global__ __launch_bounds__(256) void post_kernel(const float *__restrict__ base, float *__restrict__ rows,
                                                   const int4 *__restrict__ utt_info, const int *__restrict__ chunks,
                                                   const int n_chunks, const PostParams pp) {
    extern __shared__ float psm[];
    constexpr int PF = STD ? 4 : 12;  // prefetch registers per thread; the host keeps R * Dbase <= 256 * PF
    const int fc = STD ? 13 : pp.fea_c, Db = STD ? 13 : pp.Dbase, D = STD ? 39 : pp.D;
    const int order_ = STD ? 2 : pp.order;
    const bool stack_ = STD ? false : pp.stack != 0;
    const int wv[3] = {STD ? 2 : pp.w[0], STD ? 2 : pp.w[1], STD ? 0 : pp.w[2]};
    const int H = stack_ ? wv[0] : wv[0] + (order_ > 1 ? wv[1] : 0) + (order_ > 2 ? wv[2] : 0);
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
        if (stack_) {
            const int w = wv[0], L = 2 * w + 1, xs = fc * L;
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
            for (int k = 0; k < order_; k++) {
                const int w = wv[k];
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
            const int xs = fc * (order_ + 1);
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

}  // namespace
