// 1024-point frames (windows of 513 to 1024 samples - every window of 33 to 64 ms at 16 kHz, src/io/opts.cc:277-280) with the
// transform in registers: ONE WAVE PER FRAME, no workgroup barrier anywhere.  The packed real FFT is a 512-point complex
// FFT = 8 x 8 x 8: each lane holds 8 complex values, three in-register 8-point DFTs with two transposes through the wave's
// own LDS plane in between (re, then im; conflict-free: row stride 72 dwords), then the untangle with the mirror bin fetched across
// the wave by ds_bpermute.  The rest of the chain is bigfft_kernel's (the plain chain: pre-emphasis, window, mean removal,
// |.|^2 or |.|, any filter bank of up to 64 bands, ^0.33, log, DCT / band outputs / LP lags, the energy column), at wave
// granularity; LP kinds leave their autocorrelation lags to lp_tail_kernel.  Included by engine.hip.
//
// Index algebra (n = n0 + 8 n1 + 64 n2 in, k = k2 + 8 k1 + 64 k0 out, W = e^{-2 pi i / 512}):
//   A[n0,n1,k2] = sum_n2 z[n] W8^(n2 k2)                         lane = n0 + 8 n1, registers over n2 -> k2
//   B[n0,k1,k2] = sum_n1 A W64^(n1 k2) W8^(n1 k1)                lane = n0 + 8 k2, registers over n1 -> k1
//   Z[k]        = sum_n0 B W^(n0 (k2 + 8 k1)) W8^(n0 k0)         lane = k2 + 8 k1, registers over n0 -> k0
// so lane L ends with bins L + 64 k0: consecutive lanes hold consecutive bins.
#pragma once

namespace {

// In-register 8-point DFT, natural order in and out (forward, W8 = e^{-2 pi i / 8}).
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    constexpr float R2 = 0.70710678118654752f;
    // radix-2 decimation in frequency, three levels; outputs come out bit-reversed and are renamed at the end
    float2 a[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = make_float2(v[i].x + v[i + 4].x, v[i].y + v[i + 4].y);
        a[i + 4] = make_float2(v[i].x - v[i + 4].x, v[i].y - v[i + 4].y);
    }
    // twiddles W8^i on the lower half
    a[5] = make_float2((a[5].x + a[5].y) * R2, (a[5].y - a[5].x) * R2);    // W8^1 = (1 - i) / sqrt 2
    a[6] = make_float2(a[6].y, -a[6].x);                                    // W8^2 = -i
    a[7] = make_float2((a[7].y - a[7].x) * R2, -(a[7].x + a[7].y) * R2);   // W8^3 = (-1 - i) / sqrt 2
    float2 b[8];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int o = 4 * h;
        b[o + 0] = make_float2(a[o + 0].x + a[o + 2].x, a[o + 0].y + a[o + 2].y);
        b[o + 1] = make_float2(a[o + 1].x + a[o + 3].x, a[o + 1].y + a[o + 3].y);
        b[o + 2] = make_float2(a[o + 0].x - a[o + 2].x, a[o + 0].y - a[o + 2].y);
        const float2 d = make_float2(a[o + 1].x - a[o + 3].x, a[o + 1].y - a[o + 3].y);
        b[o + 3] = make_float2(d.y, -d.x);  // times W4^1 = -i
    }
    // last level: pairs (0,1), (2,3), (4,5), (6,7) -> X[0], X[4], X[2], X[6], X[1], X[5], X[3], X[7]
    v[0] = make_float2(b[0].x + b[1].x, b[0].y + b[1].y);
    v[4] = make_float2(b[0].x - b[1].x, b[0].y - b[1].y);
    v[2] = make_float2(b[2].x + b[3].x, b[2].y + b[3].y);
    v[6] = make_float2(b[2].x - b[3].x, b[2].y - b[3].y);
    v[1] = make_float2(b[4].x + b[5].x, b[4].y + b[5].y);
    v[5] = make_float2(b[4].x - b[5].x, b[4].y - b[5].y);
    v[3] = make_float2(b[6].x + b[7].x, b[6].y + b[7].y);
    v[7] = make_float2(b[6].x - b[7].x, b[6].y - b[7].y);
}

constexpr int W1K_TS = 72;                       // dwords between the rows of the transpose plane (first transpose)
constexpr int W1K_TS2 = 68;                      // ... of the second transpose
constexpr int W1K_PLANE = 8 * W1K_TS;            // the plane: re and im of a transpose go through it one after the other; the power
                                                 // spectrum P[513 (+3)] takes its place once the transform is done
constexpr int W1K_WAVE_FLOATS = W1K_PLANE + 64 + 64 + 64;  // plane / P | Y[64] | Ylog[64] | partial band sums [64]
constexpr int W1K_TW_FLOATS = 2 * (512 + 64 + 512);        // W1024^m for m < 512 | t1[8][8] | t2[8][64], float2 each
#ifndef CTU_W1K_WAVES
#define CTU_W1K_WAVES 4  // measured (profiles/r03_ab_wave1k_occupancy.txt, 1.78 M frames): 4 waves x 5 workgroups per CU (96 VGPRs) 2.86 ms,
#define CTU_W1K_LB 5     // 8 x 3 (80 VGPRs, 35 spilled) 3.09 ms, 4 x 4 (120 VGPRs, none spilled) 3.05 ms; round 3's first version (163 VGPRs, 3 per SIMD) 4.10 ms
#endif
constexpr int W1K_WAVES = CTU_W1K_WAVES;         // waves per workgroup

// A frame is one wave's serial chain of ~40 LDS round trips: what hides them is other waves, so the kernel is built for five
// per SIMD - twiddles come from one shared W1024 table instead of 48 registers per lane, samples are not fetched a frame ahead.
// EXTEN: -nr_mode exten (the recurrence's eighteen state registers and the chain walk) as an instantiation of its own, one wave per SIMD
// fewer: the plain kernel keeps its registers.
template <bool EXTEN>
__global__ __launch_bounds__(64 * W1K_WAVES, EXTEN ? CTU_W1K_LB - 1 : CTU_W1K_LB) void wave1k_kernel(const BigParams p, void *lp_r, int lp_stride) {
    extern __shared__ __align__(16) float smem[];
    constexpr int Nc = 512;
    const int K = p.K;  // 513
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // per-wave areas, then the tables shared by the workgroup's four waves
    float *wbase = smem + wave * W1K_WAVE_FLOATS;
    float *xpl = wbase;
    float *P = wbase;
    float *Y = wbase + W1K_PLANE, *Ylog = Y + 64, *part_s = Ylog + 64;
    float *tab = smem + W1K_WAVES * W1K_WAVE_FLOATS;
    float2 *lwin2 = reinterpret_cast<float2 *>(tab);           // [512] (w[2n], w[2n+1]), zero beyond the window
    float2 *ltw = reinterpret_cast<float2 *>(tab + 1024);      // [512] W1024^m = e^{-2 pi i m / 1024}, m < 512 (the untangle's W1024^k)
    // the stages' twiddles laid out as they are read: t1[r][n1] (eight distinct values per read, eight neighbouring entries) and
    // t2[r][lane] - out of the shared W1024 table they sat 128 n1 r bytes apart, all on one bank (SQ_LDS_BANK_CONFLICT: 247 of the
    // kernel's 651 LDS cycles per frame, profiles/r03_pmc_configs.txt)
    float2 *lt1 = ltw + 512, *lt2 = lt1 + 64;
    float *lfb = tab + 1024 + W1K_TW_FLOATS;
    double *lcoef_d = reinterpret_cast<double *>(lfb + ((p.fb_total + 3) & ~3) + (((p.fb_total + 3) & ~3) & 1));
    const int ncd = (p.feat == FEAT_LP) ? (p.lporder + 1) * p.B : 0, ncf = (p.feat == FEAT_DCTC) ? p.ncoef_out * p.B : 0;
    float *lcoef = reinterpret_cast<float *>(lcoef_d + ncd);
    int *lrange = reinterpret_cast<int *>(lcoef + ((ncf + 3) & ~3));  // [B][3] bands' bin ranges, then the bank's segment table
    int *lseg = lrange + ((3 * p.B + 3) & ~3);                        // [64][4] {band, first bin, bins, weight offset} | [B][2] {first lane, lanes}
    constexpr int NT_ = 64 * W1K_WAVES;
    for (int i = tid; i < 1024; i += NT_) tab[i] = i < p.window ? p.win[i] : 0.f;
    auto w1024 = [&](int m) {  // p.tw holds m < 512: the other half is its negative
        const float2 t = p.tw[m & 511];
        return m & 512 ? make_float2(-t.x, -t.y) : t;
    };
    for (int i = tid; i < 512; i += NT_) ltw[i] = p.tw[i];
    // after step 1, lane (n0 = d0, n1 = d1), register k2 = r: W64^(n1 k2) = W1024^(16 n1 r); after step 2, lane (n0 = d0, k2 = d1),
    // register k1 = r: W512^(n0 (k2 + 8 k1)) = W1024^(2 n0 k2 + 16 n0 r)
    for (int i = tid; i < 64; i += NT_) lt1[i] = w1024((16 * (i & 7) * (i >> 3)) & 1023);
    for (int i = tid; i < 512; i += NT_) {
        const int l_ = i & 63, r_ = i >> 6;
        lt2[i] = w1024((2 * (l_ & 7) * (l_ >> 3) + 16 * (l_ & 7) * r_) & 1023);
    }
    for (int i = tid; i < p.fb_total; i += NT_) lfb[i] = p.fbw[i];
    for (int i = tid; i < ncd; i += NT_) lcoef_d[i] = p.coef_d[i];
    for (int i = tid; i < ncf; i += NT_) lcoef[i] = p.coef[i];
    for (int i = tid; i < 3 * p.B; i += NT_) lrange[i] = p.fb_range[i];
    for (int i = tid; i < 256 + 2 * p.B; i += NT_) lseg[i] = p.seg[i];
    __syncthreads();  // the only workgroup barrier: tables are in place

    const int d0 = lane & 7, d1 = lane >> 3;  // low / high digit of the lane number
    const int mirror = ((64 - lane) & 63) << 2;
    // output slots of the DCT rows 4 (lane / 16) + r this lane stores (-1: not written): read once, not per frame
    int slot_of_row[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row_ = 4 * (lane >> 4) + r;
        slot_of_row[r] = (p.feat == FEAT_DCTC && row_ < p.ncoef_out) ? p.row_slot[row_] : -1;
    }

    const int gw = blockIdx.x * W1K_WAVES + wave, nw = gridDim.x * W1K_WAVES;
    // exten state (src/nr/nr.cc:86-93): lane = bin (bins lane + 64 r, r < 8, and bin 512 with lane 0), carried along the wave's utterance
    float navg[EXTEN ? 9 : 1], yavg[EXTEN ? 9 : 1];
#pragma unroll
    for (int r = 0; r < (EXTEN ? 9 : 1); r++) {
        navg[r] = 0.95f;
        yavg[r] = 0.05f;
    }
    int tile = EXTEN ? (gw < p.n_chains ? p.chain_first[gw] : -1) : (gw < p.n_tiles ? gw : -1);
    while (tile >= 0) {
        const TileRec rec = load_rec(p.tiles, tile);
        tile = EXTEN ? rec.next : (tile + nw < p.n_tiles ? tile + nw : -1);
        for (int f = 0; f < rec.nvalid; f++) {
            pcm4 q[8];  // samples x[i-2 .. i+1], i = 2 lane + 128 n2
            {
                const int16_t *x = p.pcm + rec.sbase + (int64_t)f * p.wshift + 2 * lane - 2;
#pragma unroll
                for (int j = 0; j < 8; j++) q[j] = *reinterpret_cast<const pcm4 *>(x + 128 * j);
            }
            const bool file_start = lane == 0 && rec.t0 + f == 0;
            // ---- pre-emphasis x window (src/io/in.cc:364-372), z[n] = y[2n] + i y[2n+1], n = lane + 64 n2
            float2 v[8];
            float part = 0.f;
            double raw = 0.0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float2 w = lwin2[lane + 64 * j];
                float xm = (float)(int16_t)(q[j].lo >> 16);
                const float x0 = (float)(int16_t)(q[j].hi & 0xffffu), x1 = (float)(int16_t)(q[j].hi >> 16);
                if (j == 0) xm = file_start ? 0.f : xm;
                const float y0 = w.x * (x0 - p.preem * xm), y1 = w.y * (x1 - p.preem * x0);
                v[j] = make_float2(y0, y1);
                part += y0 + y1;
                if (p.e_mode == 4) {  // raw energy: sum of x[i]^2, i = 1 .. window-1 (src/io/in.cc:353-361)
                    const int i0 = 2 * lane + 128 * j;
                    if (i0 >= 1 && i0 < p.window) raw += (double)x0 * (double)x0;
                    if (i0 + 1 < p.window) raw += (double)x1 * (double)x1;
                }
            }
            if (p.remove_dc) {  // src/io/in.cc:375-382: the mean over `window` samples leaves the samples inside the window
                const float m = (float)(wave_sum_fast((double)part) / (double)p.window);  // a lane's 16 values in float, the wave's sum in double (as bigfft_kernel)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int i0 = 2 * lane + 128 * j;
                    v[j].x -= i0 < p.window ? m : 0.f;
                    v[j].y -= i0 + 1 < p.window ? m : 0.f;
                }
            }
            // ---- 512-point complex FFT
            auto exchange = [&](int wbase_, int wstep, int rstride) {  // one transpose: every lane writes v[r] at wbase_ + wstep r and reads lane + rstride r
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 8; r++) xpl[wbase_ + wstep * r] = v[r].x;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 8; r++) v[r].x = xpl[lane + rstride * r];
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 8; r++) xpl[wbase_ + wstep * r] = v[r].y;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 8; r++) v[r].y = xpl[lane + rstride * r];
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            };
            dft8(v);
#pragma unroll
            for (int r = 1; r < 8; r++) v[r] = cmul(v[r], lt1[8 * r + d1]);
            exchange(d0 + W1K_TS * d1, 8, W1K_TS);      // element (n0, n1, k2 = r) -> row n1, column n0 + 8 k2; then lane = n0 + 8 k2, register n1
            dft8(v);
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = cmul(v[r], lt2[64 * r + lane]);
            // rows 68 dwords apart here: the row index is the lane's LOW digit, and 72 d0 = 8 d0 (mod 32) puts d0 and d0 + 4 of a half wave on
            // one bank (a two-way conflict on all sixteen writes: 32 of the kernel's 162 conflict cycles per frame); 68 d0 = 4 d0 keeps the 32
            // lanes of a half wave on 32 banks
            exchange(d1 + W1K_TS2 * d0, 8, W1K_TS2);    // element (n0, k1 = r, k2) -> row n0, column k2 + 8 k1; then lane = k2 + 8 k1, register n0
            dft8(v);  // v[r] = Z[lane + 64 r]
            // ---- untangle the packed transform, |.|^2 (src/io/in.cc:388-394): bin k = lane + 64 r with Z[512 - k] from lane
            //      (64 - lane) % 64, register 7 - r (lane 0: register (8 - r) % 8, Z[512] = Z[0])
#pragma unroll
            for (int r = 0; r < 8; r++) {
                float cr = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(v[7 - r].x)));
                float ci = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(v[7 - r].y)));
                if (lane == 0) {
                    cr = v[(8 - r) & 7].x;
                    ci = v[(8 - r) & 7].y;
                }
                const float2 a = v[r], w = ltw[lane + 64 * r];
                const float sr = a.x + cr, si = a.y - ci, dr = a.x - cr, di = a.y + ci;
                const float tr = w.x * di + w.y * dr, ti = w.y * di - w.x * dr;
                const float ur = sr + tr, ui = si + ti;
                float pw = 0.25f * (ur * ur + ui * ui);
                if (lane == 0 && r == 0) {
                    const float s0 = a.x + a.y, s1 = a.x - a.y;
                    pw = p.remove_dc ? 1e-10f : s0 * s0;
                    P[Nc] = p.fb_power ? s1 * s1 : fabsf(s1);
                }
                P[lane + 64 * r] = p.fb_power ? pw : sqrtf(pw);  // src/io/in.cc:415-417
            }
            if (lane < 3) P[K + lane] = 0.f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (EXTEN) {
                // extended spectral subtraction (src/nr/nr.cc:95-140), frontend_kernel's float recurrence: H = Navg / (Navg^a + Yavg^a)^(1/a),
                // N = H X, Navg = p Navg + (1 - p) N, Yavg = |X - Navg|, X -= N written as X (1 - H) without the cancellation
                if (rec.t0 + f == 0) {
#pragma unroll
                    for (int r = 0; r < 9; r++) {
                        navg[r] = 0.95f;
                        yavg[r] = 0.05f;
                    }
                }
                const float pp = p.nr_p, qq = 1.0f - p.nr_p;
#pragma unroll
                for (int r = 0; r < 9; r++) {
                    const int k = lane + 64 * r;
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
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (p.vad_en) {  // the VAD's energy criterion on the vector the NR left (frontend_kernel's vad_export == 2)
                float es = 0.f;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const float x = P[lane + 64 * r];
                    es += x * x;
                }
                if (lane == 0) es += P[Nc] * P[Nc];
                const float tot = (float)wave_sum_fast((double)es);
                if (lane == 0) p.vad_en[rec.rbase + f] = tot;
            }
            double e_spec = 0.0;
            if (p.e_mode == 1) {  // E = log(2 (X0^2/2 + sum X_i^2 + X_{K-1}^2/2)) (src/nr/nr.cc:36-45)
                double s = 0.0;
                for (int k = lane; k < K; k += 64) s += ((k == 0 || k == K - 1) ? 0.5 : 1.0) * (double)P[k] * (double)P[k];
                e_spec = wave_sum_fast(s);
            }
            // ---- filter bank (src/fea/fb.cc:60-83).  The bands are cut into at most 64 segments of about equal length (host
            //      table): every lane sums one segment, then lane b adds up band b's partial sums in segment order.
            {
                const int sk = lseg[4 * lane + 1], sn = lseg[4 * lane + 2];
                const float *w = lfb + lseg[4 * lane + 3];
                const float *pk = P + sk;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                int i = 0;
                for (; i + 3 < sn; i += 4) {
                    a0 += w[i] * pk[i];
                    a1 += w[i + 1] * pk[i + 1];
                    a2 += w[i + 2] * pk[i + 2];
                    a3 += w[i + 3] * pk[i + 3];
                }
                for (; i < sn; i++) a0 += w[i] * pk[i];
                part_s[lane] = (a0 + a1) + (a2 + a3);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane < p.B) {
                    const int l0 = lseg[256 + 2 * lane], nl = lseg[256 + 2 * lane + 1];
                    float acc = 0.f;
                    for (int j = 0; j < nl; j++) acc += part_s[l0 + j];
                    if (p.fb_inld) acc = __builtin_amdgcn_exp2f(0.33f * __builtin_amdgcn_logf(acc));
                    Y[lane] = acc;
                    Ylog[lane] = __builtin_amdgcn_logf(acc) * 0.69314718056f;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int64_t row = rec.rbase + f;
            double e_band = 0.0;
            if (p.e_mode == 3) {  // band energy of the FB output (src/fea/fea_impl.cc:44-50,68-74)
                double s = 0.0;
                if (lane < p.B) s = ((lane == 0 || lane == p.B - 1) ? 0.5 : 1.0) * (double)Y[lane] * (double)Y[lane];
                e_band = wave_sum_fast(s);
            }
            if (p.feat == FEAT_BANDS) {
                float *dst_ = p.band_to_scratch ? p.logmel : p.rows;
                const int out_w = p.band_to_scratch ? p.B : p.D;
                if (lane < p.B) dst_[row * out_w + lane] = p.band_log ? Ylog[lane] : Y[lane];
            } else if (p.feat == FEAT_DCTC) {
                if (p.ncoef_out <= 16 && p.B <= 64) {
                    // the DCT as a chain of v_mfma_f32_16x16x4_f32 (exact fp32 FMAs): A[m][kk] = row m of the folded table at band
                    // 4 s + kk (lane = m + 16 kk), B[kk][n] = that band's logarithm in every column n: all 16 columns of D carry the
                    // frame's coefficients, lane 16 q stores rows 4 q .. 4 q + 3
                    const int m = lane & 15, kk = lane >> 4;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    for (int s4 = 0; s4 < p.B; s4 += 4) {
                        const int b = s4 + kk;
                        const float av = (m < p.ncoef_out && b < p.B) ? lcoef[m * p.B + b] : 0.f;
                        const float bv = b < p.B ? Ylog[b] : 0.f;
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
                    }
                    if (m == 0) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (slot_of_row[r] >= 0) p.rows[row * p.D + 4 * kk + r] = acc[r];
                    }
                } else {
                    // more than 16 coefficients: 16 lanes per coefficient, four coefficients per round
                    for (int r0_ = 0; r0_ < p.ncoef_out; r0_ += 4) {
                        const int r = r0_ + (lane >> 4), j = lane & 15;
                        float acc = 0.f;
                        if (r < p.ncoef_out) {
                            const float *c = lcoef + r * p.B;
                            for (int b = j; b < p.B; b += 16) acc += c[b] * Ylog[b];
                        }
                        acc = row16_allreduce_add(acc);
                        if (j == 0 && r < p.ncoef_out && p.row_slot[r] >= 0) p.rows[row * p.D + r] = acc;
                    }
                }
            } else {
                // LP kinds: R[k] by cosine iDFT of the band energies in double (src/fea/fea_impl.cc:163-198), lane k takes lag k;
                // Levinson-Durbin and a -> c are lp_tail_kernel's (one frame per lane)
                if (lane <= p.lporder) {
                    double r = 0.0;
                    for (int b = 0; b < p.B; b++) {
                        const double y = p.fb_inld ? (double)Y[b] : (double)Y[b] * (double)Y[b];
                        r += lcoef_d[lane * p.B + b] * y;
                    }
                    reinterpret_cast<double *>(lp_r)[row * lp_stride + lane] = r;
                }
            }
            double e_raw = 0.0;
            if (p.e_mode == 4) e_raw = wave_sum_fast(raw);
            if (p.e_mode && p.e_mode != 2 && lane == 0 && !(p.feat == FEAT_BANDS && p.band_to_scratch)) {
                double e = 0.0;
                if (p.e_mode == 1) e = log(2.0 * e_spec);
                else if (p.e_mode == 3) e = log(2.0 * e_band);
                else if (p.e_mode == 4) e = log(e_raw);
                p.rows[row * p.D + p.e_slot] = (float)e;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // P, Y and the planes are rewritten by the next frame
        }
    }
}

}  // namespace
