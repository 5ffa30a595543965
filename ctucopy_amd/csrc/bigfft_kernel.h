// FFT sizes of 1024 to 4096 points (windows of 513 to 4096 samples: 22.05 / 32 / 44.1 / 48 kHz input, or long windows;
// src/io/opts.cc:277-280 allows any power of two).  The 16-lane register transforms of frontend_kernel.h are built for 512
// and 256 points and for P rows of 260 floats; these sizes take the plain road: one workgroup per frame, the packed real
// FFT as a Stockham radix-2 autosort through two LDS buffers, the power spectrum in LDS, four lanes per band for the
// filter bank, one lane per output coefficient.  The plain chain only (no noise reduction, no VAD): pre-emphasis,
// window, mean removal, |.|^2 or |.|, any filter bank, ^0.33, log, DCT / band outputs / LP analysis, the energy column.
// Every barrier here hands over LDS data only (lds_barrier, trap_kernel.h): __syncthreads() would also drain the vector-memory
// counter, i.e. wait for the samples fetched a frame ahead and for the previous frame's row stores at every pass.
// Included by engine.hip.
#pragma once

namespace {

struct BigParams {
    const int16_t *pcm;
    float *rows, *logmel;
    const TileRec *tiles;
    int n_tiles;
    const float *win;        // [window] Hamming
    const float2 *tw;        // [wfft/2] W_wfft^m = (cos, -sin)(2 pi m / wfft)
    const float *fbw;        // the bands' non-zero runs one after the other (fb_total floats)
    const int *fb_range;     // [B][3] first / last non-zero bin, offset of the run in fbw
    int fb_total;
    const int *seg;          // wave1k_kernel: the bank cut into <= 64 segments, [64][4] {band, first bin, bins, offset into fbw} | [B][2] {first segment, segments}
    const float *coef;       // dctc: [ncoef_out][B] rows in writer order (norm and lifter folded in); lp: [lporder+1][B] as doubles below
    const double *coef_d;    // lp: cosine iDFT rows in double [lporder+1][B]
    const float *lifter;     // [ncep]
    const int *row_slot;     // lp: output slot of cepstrum n (or -1)
    int wfft, K, window, wshift, B, D, ncoef_out, feat, e_mode, e_slot;
    int remove_dc, fb_power, fb_inld, band_log, band_to_scratch, lp_is_lpa, lporder, ncep, lifter_on;
    float preem;
    // wave1k_kernel with -nr_mode exten (src/nr/nr.cc:86-140): the recurrence runs along an utterance, so a wave walks a chain of whole
    // utterances (the plan's per-wave chains: chain_first[n_chains], TileRec::next) instead of striding over the tile list
    int nr_exten, n_chains;
    const int *chain_first;
    float nr_p, nr_a;
    float2 *xri;             // speech output (-format_out raw|wave): the frame's complex spectrum [K] and, in pnr, the magnitudes the NR left [K]
    float *pnr;              // go to the plan's scratch for bigsynth_kernel; nothing is projected.  NULL on the feature path
    const float *dc1;        // -remove_dc1 (bigfft_kernel): the frames' offsets (decode_kernels.h), or NULL; dc1_J = floor(window / wshift) <= 8
    int dc1_J;
    float *vad_en;           // wave1k_kernel: the VAD's energy criterion per frame (sum of squares of the vector the NR left, src/vad/vad.cc:96-107), or NULL
};

__device__ __forceinline__ double block_sum(double v, double *red) {  // 256 threads; red: 4 doubles of LDS
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    lds_barrier();
    return red[0] + red[1] + red[2] + red[3];
}

// EXTEN: -nr_mode exten (src/nr/nr.cc:86-140) - the recurrence runs along an utterance, so a WORKGROUP walks one of the plan's chains of
// whole utterances (chain_first / TileRec::next, as wave1k_kernel's waves do) with Navg / Yavg of its bins k = tid + 256 i in registers.
template <int NIT, bool EXTEN = false>  // NIT = wfft / 256: 4, 8 or 16 samples per lane
__global__ __launch_bounds__(256) void bigfft_kernel(const BigParams p) {
    extern __shared__ __align__(16) float smem[];
    const int N = p.wfft, Nc = N >> 1, K = p.K;
    float2 *A = reinterpret_cast<float2 *>(smem);      // [Nc]
    float2 *Bf = A + Nc;                               // [Nc]
    float *P = reinterpret_cast<float *>(Bf + Nc);     // [K] (+3 padding)
    float *Y = P + ((K + 3) & ~3);                     // [64] band values
    double *red = reinterpret_cast<double *>(Y + 64);  // [4]
    // tables, staged once per workgroup: twiddles [Nc], window [window rounded up to 4], filter-bank runs [fb_total]
    float2 *ltw = reinterpret_cast<float2 *>(red + 4);
    float *lwin = reinterpret_cast<float *>(ltw + Nc);
    float *lfb = lwin + ((p.window + 3) & ~3);
    // coefficient rows of the tail (a loop over global memory would pay an L2 round trip per band): dctc [ncoef_out][B]
    // floats, lp [lporder+1][B] doubles; band ranges [B][3]; band logarithms [64]
    double *lcoef_d = reinterpret_cast<double *>(lfb + ((p.fb_total + 3) & ~3));
    const int ncd = (p.feat == FEAT_LP) ? (p.lporder + 1) * p.B : 0, ncf = (p.feat == FEAT_DCTC) ? p.ncoef_out * p.B : 0;
    float *lcoef = reinterpret_cast<float *>(lcoef_d + ncd);
    int *lrange = reinterpret_cast<int *>(lcoef + ((ncf + 3) & ~3));
    float *Ylog = reinterpret_cast<float *>(lrange + ((3 * p.B + 3) & ~3));
    const int tid = threadIdx.x;
    for (int i = tid; i < Nc; i += 256) ltw[i] = p.tw[i];
    for (int i = tid; i < p.window; i += 256) lwin[i] = p.win[i];
    for (int i = tid; i < p.fb_total; i += 256) lfb[i] = p.fbw[i];
    for (int i = tid; i < ncd; i += 256) lcoef_d[i] = p.coef_d[i];
    for (int i = tid; i < ncf; i += 256) lcoef[i] = p.coef[i];
    for (int i = tid; i < 3 * p.B; i += 256) lrange[i] = p.fb_range[i];
    lds_barrier();
    constexpr int NB = NIT / 2 + 1;  // bins per thread: K = 128 NIT + 1
    float navg[EXTEN ? NB : 1], yavg[EXTEN ? NB : 1];
#pragma unroll
    for (int r = 0; r < (EXTEN ? NB : 1); r++) {
        navg[r] = 0.95f;
        yavg[r] = 0.05f;
    }
    int tile = EXTEN ? ((int)blockIdx.x < p.n_chains ? p.chain_first[blockIdx.x] : -1) : ((int)blockIdx.x < p.n_tiles ? (int)blockIdx.x : -1);
    while (tile >= 0) {
        const TileRec rec = load_rec(p.tiles, tile);
        tile = EXTEN ? rec.next : (tile + (int)gridDim.x < p.n_tiles ? tile + (int)gridDim.x : -1);
        // the samples of a frame are fetched one frame ahead (registers): the loads fly under the previous frame's passes
        int16_t cur[NIT], prv[NIT];
        auto fetch = [&](int f) {
            const int16_t *x = p.pcm + rec.sbase + (int64_t)f * p.wshift;
            const bool file_start = rec.t0 + f == 0;
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int i = tid + 256 * it;
                cur[it] = i < p.window ? x[i] : (int16_t)0;
                prv[it] = (i < p.window && !(i == 0 && file_start)) ? x[i - 1] : (int16_t)0;
            }
        };
        fetch(0);
        for (int f = 0; f < rec.nvalid; f++) {
            // ---- pre-emphasis x window (src/io/in.cc:364-372), packed as z[n] = y[2n] + i y[2n+1]
            float *yb = reinterpret_cast<float *>(A);
            float partf = 0.f;  // a lane's 4..16 values in float (as the register kernels do), the block sum in double
            double raw = 0.0;
            // -remove_dc1 (src/io/in.cc:343-350, decode_kernels.h): the sample at position i of frame t is read as
            // x - o_t - sum_{j >= 1, i <= window-1-j*wshift} o_{t-j}; the sample ahead of the frame (position -1) without the o_t term
            float ov[9];
#pragma unroll
            for (int j = 0; j <= 8; j++) ov[j] = (p.dc1 && j <= p.dc1_J && rec.t0 + f - j >= 0) ? p.dc1[rec.rbase + f - j] : 0.f;
            auto dc1_cum = [&](int i, bool with_own) {
                float c = with_own ? ov[0] : 0.f;
#pragma unroll
                for (int j = 1; j <= 8; j++) c += (j <= p.dc1_J && i <= p.window - 1 - j * p.wshift) ? ov[j] : 0.f;
                return c;
            };
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int i = tid + 256 * it;
                float y = 0.f;
                if (i < p.window) {
                    float xi = (float)cur[it], xp = (float)prv[it];
                    if (p.dc1) {
                        xi -= dc1_cum(i, true);
                        xp -= dc1_cum(i - 1, i >= 1);
                        if (i == 0 && rec.t0 + f == 0) xp = 0.f;  // first sample of the file: history is 0
                    }
                    y = lwin[i] * (xi - p.preem * xp);
                    partf += y;
                    if (p.e_mode == 4 && i >= 1) raw += (double)xi * (double)xi;
                }
                yb[i] = y;
            }
            const double part = (double)partf;
            if (f + 1 < rec.nvalid) fetch(f + 1);
            if (p.remove_dc) {  // src/io/in.cc:375-382
                const float m = (float)(block_sum(part, red) / (double)p.window);
                for (int i = tid; i < p.window; i += 256) yb[i] -= m;
            }
            double e_raw = 0.0;
            if (p.e_mode == 4) e_raw = block_sum(raw, red);
            lds_barrier();
            // ---- Nc-point complex FFT, Stockham radix-2: log2(Nc) passes between the two buffers
            float2 *src = A, *dst = Bf;
            for (int Ns = 1; Ns < Nc; Ns <<= 1) {
                const int tstep = Nc / Ns;  // W_{2Ns}^k = W_wfft^(k * Nc / Ns)
                for (int j = tid; j < (Nc >> 1); j += 256) {
                    const int k = j & (Ns - 1);
                    const float2 w = ltw[k * tstep];
                    const float2 a = src[j], b0 = src[j + (Nc >> 1)];
                    const float2 b = make_float2(b0.x * w.x - b0.y * w.y, b0.x * w.y + b0.y * w.x);
                    const int j0 = ((j - k) << 1) + k;
                    dst[j0] = make_float2(a.x + b.x, a.y + b.y);
                    dst[j0 + Ns] = make_float2(a.x - b.x, a.y - b.y);
                }
                lds_barrier();
                float2 *t_ = src;
                src = dst;
                dst = t_;
            }
            // ---- untangle the packed transform, |.|^2 (src/io/in.cc:388-394), bins 0..Nc
            for (int k = tid; k <= Nc; k += 256) {
                float pw;
                float2 Xk;  // the bin itself (speech output keeps its direction)
                if (k == 0) {
                    const float v = src[0].x + src[0].y;
                    pw = p.remove_dc ? 1e-10f : v * v;
                    Xk = make_float2(v, 0.f);
                } else if (k == Nc) {
                    const float v = src[0].x - src[0].y;
                    pw = v * v;
                    Xk = make_float2(v, 0.f);
                } else {
                    const float2 a = src[k], c = src[Nc - k], w = ltw[k];
                    const float sr = a.x + c.x, si = a.y - c.y, dr = a.x - c.x, di = a.y + c.y;
                    // X = (s - i W d) / 2 with W = (w.x, w.y): -i W d = (w.x di + w.y dr, w.y di - w.x dr) ... conj convention of tw = (cos, -sin)
                    const float tr = w.x * di + w.y * dr, ti = w.y * di - w.x * dr;
                    const float ur = sr + tr, ui = si + ti;
                    pw = 0.25f * (ur * ur + ui * ui);
                    Xk = make_float2(0.5f * ur, 0.5f * ui);
                }
                P[k] = p.fb_power ? pw : sqrtf(pw);  // src/io/in.cc:415-417
                if (p.xri) p.xri[(rec.rbase + f) * K + k] = Xk;
            }
            lds_barrier();
            if constexpr (EXTEN) {
                // extended spectral subtraction, frontend_kernel's float recurrence (see wave1k_kernel.h): thread t owns bins t + 256 r
                if (rec.t0 + f == 0) {
#pragma unroll
                    for (int r = 0; r < NB; r++) {
                        navg[r] = 0.95f;
                        yavg[r] = 0.05f;
                    }
                }
                const float pp = p.nr_p, qq = 1.0f - p.nr_p;
#pragma unroll
                for (int r = 0; r < NB; r++) {
                    const int k = tid + 256 * r;
                    const float X = k < K ? P[k] : 1.f;
                    float H, omH;
                    if (p.nr_a == 1.0f) {
                        const float ir = __builtin_amdgcn_rcpf(navg[r] + yavg[r]);
                        H = navg[r] * ir;
                        omH = yavg[r] * ir;
                    } else if (p.nr_a == 2.0f) {
                        const float r2 = navg[r] * navg[r] + yavg[r] * yavg[r];
                        const float ir = __builtin_amdgcn_rsqf(r2);
                        const float rr = r2 * ir;
                        H = navg[r] * ir;
                        omH = (yavg[r] * yavg[r]) * __builtin_amdgcn_rcpf(rr * (rr + navg[r]));
                    } else {
                        H = navg[r] / powf(powf(navg[r], p.nr_a) + powf(yavg[r], p.nr_a), 1.0f / p.nr_a);
                        omH = 1.0f - H;
                    }
                    const float N = H * X;
                    navg[r] = pp * navg[r] + qq * N;
                    yavg[r] = fabsf(X - navg[r]);
                    if (k < K) P[k] = X * omH;
                }
                lds_barrier();
            }
            if (p.xri) {  // speech output: what the NR left of the magnitudes; the inverse transform is bigsynth_kernel's
                for (int k = tid; k < K; k += 256) p.pnr[(rec.rbase + f) * K + k] = P[k];
                lds_barrier();  // P and the FFT buffers are rewritten by the next frame
                continue;
            }
            if (p.vad_en) {  // the VAD's energy criterion on the vector the NR left (frontend_kernel's vad_export == 2)
                double s = 0.0;
                for (int k = tid; k < K; k += 256) s += (double)(P[k] * P[k]);
                const double tot = block_sum(s, red);
                if (tid == 0) p.vad_en[rec.rbase + f] = (float)tot;
            }
            double e_spec = 0.0;
            if (p.e_mode == 1) {  // E = log(2 (X0^2/2 + sum X_i^2 + X_{K-1}^2/2)) (src/nr/nr.cc:36-45)
                double s = 0.0;
                for (int k = tid; k < K; k += 256) s += ((k == 0 || k == K - 1) ? 0.5 : 1.0) * (double)P[k] * (double)P[k];
                e_spec = block_sum(s, red);
            }
            // ---- filter bank: lanes 4b..4b+3 share band b (src/fea/fb.cc:60-83)
            {
                const int b = tid >> 2, q = tid & 3;
                float acc = 0.f;
                if (b < p.B) {
                    const int k0 = lrange[3 * b], k1 = lrange[3 * b + 1];
                    const float *w = lfb + lrange[3 * b + 2] - k0;
                    // four independent partial sums: the reads of a step are in flight together
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                    int k = k0 + q;
                    for (; k + 12 <= k1; k += 16) {
                        a0 += w[k] * P[k];
                        a1 += w[k + 4] * P[k + 4];
                        a2 += w[k + 8] * P[k + 8];
                        a3 += w[k + 12] * P[k + 12];
                    }
                    for (; k <= k1; k += 4) a0 += w[k] * P[k];
                    acc = (a0 + a1) + (a2 + a3);
                }
                acc += __shfl_xor(acc, 1);
                acc += __shfl_xor(acc, 2);
                if (p.fb_inld) acc = __builtin_amdgcn_exp2f(0.33f * __builtin_amdgcn_logf(acc));
                if (q == 0 && b < 64) {
                    Y[b] = b < p.B ? acc : 0.f;
                    Ylog[b] = b < p.B ? __builtin_amdgcn_logf(acc) * 0.69314718056f : 0.f;
                }
            }
            lds_barrier();
            const int64_t row = rec.rbase + f;
            double e_band = 0.0;
            if (p.e_mode == 3 && tid == 0) {  // band energy of the FB output (src/fea/fea_impl.cc:44-50,68-74)
                for (int b = 0; b < p.B; b++) e_band += ((b == 0 || b == p.B - 1) ? 0.5 : 1.0) * (double)Y[b] * (double)Y[b];
            }
            if (p.feat == FEAT_BANDS) {
                float *dst_ = p.band_to_scratch ? p.logmel : p.rows;
                const int out_w = p.band_to_scratch ? p.B : p.D;
                if (tid < p.B) dst_[row * out_w + tid] = p.band_log ? Ylog[tid] : Y[tid];
            } else if (p.feat == FEAT_DCTC) {
                // 16 lanes per coefficient (ncoef_out <= 16 here or the remaining rows take a second round): lane j sums bands
                // j, j+16, ..., then four shuffle steps
                for (int r0_ = 0; r0_ < p.ncoef_out; r0_ += 16) {
                    const int r = r0_ + (tid >> 4), j = tid & 15;
                    float acc = 0.f;
                    if (r < p.ncoef_out) {
                        const float *c = lcoef + r * p.B;
                        for (int b = j; b < p.B; b += 16) acc += c[b] * Ylog[b];
                    }
                    acc += __shfl_xor(acc, 1);
                    acc += __shfl_xor(acc, 2);
                    acc += __shfl_xor(acc, 4);
                    acc += __shfl_xor(acc, 8);
                    if (j == 0 && r < p.ncoef_out && p.row_slot[r] >= 0) p.rows[row * p.D + r] = acc;
                }
            } else if (tid == 0) {
                // LP analysis in double by one lane (src/fea/fea_impl.cc:163-222, 251-284): R by cosine iDFT, Levinson-Durbin, a -> c
                double R[MAX_LP + 1], a[MAX_LP + 1], aa[MAX_LP + 1], cc[MAX_LP + 1];
                const int P_ = p.lporder;
                for (int k = 0; k <= P_; k++) {
                    double r = 0.0;
                    for (int b = 0; b < p.B; b++) {
                        const double y = p.fb_inld ? (double)Y[b] : (double)Y[b] * (double)Y[b];
                        r += lcoef_d[k * p.B + b] * y;
                    }
                    R[k] = r;
                }
                float *orow = p.rows + row * p.D;
                if (p.e_mode == 2) orow[p.e_slot] = (float)log(R[0]);
                double rc = -R[1] / R[0], err = R[0] * (1 - rc * rc);
                a[0] = aa[0] = 1;
                a[1] = aa[1] = rc;
                for (int ik = 2; ik <= P_; ik++) {
                    double dm = R[ik];
                    for (int n = 1; n < ik; n++) dm += aa[n] * R[ik - n];
                    rc = -dm / err;
                    a[ik] = rc;
                    for (int n = 1; n < ik; n++) a[n] = aa[n] + rc * aa[ik - n];
                    for (int n = 1; n <= ik; n++) aa[n] = a[n];
                    err *= (1 - rc * rc);
                }
                if (p.lp_is_lpa) {
                    for (int i = 1; i <= P_; i++) orow[i - 1] = (float)a[i];
                } else {
                    cc[0] = log(err);
                    for (int n = 1; n <= p.ncep; n++) {
                        double sum = 0;
                        for (int k = 1; k < n; k++)
                            if (k <= P_) sum += (double)(n - k) * cc[n - k] * a[k];
                        cc[n] = (n <= P_ ? -a[n] : 0.0) - sum / (double)n;
                    }
                    for (int n = 0; n <= p.ncep; n++) {
                        double val = cc[n];
                        if (n >= 1 && p.lifter_on) val *= (double)p.lifter[n - 1];
                        const int slot = p.row_slot[n];
                        if (slot >= 0) orow[slot] = (float)val;
                    }
                }
            }
            if (p.e_mode && p.e_mode != 2 && tid == 0 && !(p.feat == FEAT_BANDS && p.band_to_scratch)) {
                double e = 0.0;
                if (p.e_mode == 1) e = log(2.0 * e_spec);
                else if (p.e_mode == 3) e = log(2.0 * e_band);
                else if (p.e_mode == 4) e = log(e_raw);
                p.rows[row * p.D + p.e_slot] = (float)e;
            }
            lds_barrier();  // P, Y and the FFT buffers are rewritten by the next frame
        }
    }
}

// Speech output on 1024 .. 4096-point frames (sigOUT, src/io/out.cc:405-434; signal_kernels.h has the 256 / 512-point form): a workgroup
// per frame.  Every bin keeps its direction and takes the magnitude the NR left, times 1/N (DC and Nyquist as positive reals: the
// reference stores them before its sign fix-up, out.cc:416-419); Hermitian -> real by the packed half-size inverse transform
//   Z[k] = (X[k] + X*[M-k]) + i e^{+2 pi i k / N} (X[k] - X*[M-k]),  z = IDFT_M(Z),  y[2n] = Re z[n], y[2n+1] = Im z[n]
// (bigfft_kernel's Stockham radix-2 passes with the conjugate twiddles); the first `window` samples go to the frame's scratch row,
// which ola_kernel overlaps and adds.
template <int NIT>
__global__ __launch_bounds__(256) void bigsynth_kernel(const float2 *__restrict__ xri, const float *__restrict__ pnr, float *__restrict__ ybuf,
                                                       long long total_frames, int wfft, int window, float inv_n, const float2 *__restrict__ tw) {
    extern __shared__ __align__(16) float smem[];
    const int Nc = wfft >> 1, K = Nc + 1, tid = threadIdx.x;
    float2 *A = reinterpret_cast<float2 *>(smem);  // [Nc + 1]
    float2 *Bf = A + Nc + 4;                       // [Nc]
    float2 *ltw = Bf + Nc;                         // [Nc] (cos, -sin)(2 pi m / wfft)
    for (int i = tid; i < Nc; i += 256) ltw[i] = tw[i];
    lds_barrier();
    for (long long f = blockIdx.x; f < total_frames; f += gridDim.x) {
        const float2 *xr = xri + f * K;
        const float *pn = pnr + f * K;
        for (int k = tid; k <= Nc; k += 256) {
            float2 v;
            if (k == 0 || k == Nc) v = make_float2(pn[k] * inv_n, 0.f);
            else {
                const float2 x0 = xr[k];
                const float mag2 = x0.x * x0.x + x0.y * x0.y;
                const float sc = mag2 > 0.f ? pn[k] * inv_n * rsqrtf(mag2) : 0.f;
                v = make_float2(x0.x * sc, x0.y * sc);
            }
            A[k] = v;
        }
        lds_barrier();
        for (int k = tid; k < Nc; k += 256) {
            const float2 a = A[k], b = A[Nc - k];
            const float2 sm = make_float2(a.x + b.x, a.y - b.y);  // X[k] + conj(X[M-k])
            const float2 df = make_float2(a.x - b.x, a.y + b.y);  // X[k] - conj(X[M-k])
            const float2 w = make_float2(ltw[k].x, -ltw[k].y);    // e^{+2 pi i k / N}
            const float2 t = make_float2(-(w.x * df.y + w.y * df.x), w.x * df.x - w.y * df.y);  // i w df
            Bf[k] = make_float2(sm.x + t.x, sm.y + t.y);
        }
        lds_barrier();
        float2 *src = Bf, *dst = A;
        for (int Ns = 1; Ns < Nc; Ns <<= 1) {
            const int tstep = Nc / Ns;  // conj of W_{2Ns}^k = W_wfft^(k * Nc / Ns)
            for (int j = tid; j < (Nc >> 1); j += 256) {
                const int k = j & (Ns - 1);
                const float2 w = make_float2(ltw[k * tstep].x, -ltw[k * tstep].y);
                const float2 a = src[j], b0 = src[j + (Nc >> 1)];
                const float2 b = make_float2(b0.x * w.x - b0.y * w.y, b0.x * w.y + b0.y * w.x);
                const int j0 = ((j - k) << 1) + k;
                dst[j0] = make_float2(a.x + b.x, a.y + b.y);
                dst[j0 + Ns] = make_float2(a.x - b.x, a.y - b.y);
            }
            lds_barrier();
            float2 *t_ = src;
            src = dst;
            dst = t_;
        }
        float *yo = ybuf + f * window;  // any window: 25 ms at 44.1 kHz are 1103 samples, rows of odd length are not 8-byte aligned
        for (int n = tid; n < window; n += 256) yo[n] = (n & 1) ? src[n >> 1].y : src[n >> 1].x;
        lds_barrier();
    }
}

}  // namespace
