// Row N3: speech-enhancement output (inverse transform, overlap-add).
// Included by engine.hip (one translation unit: the kernels and their host launchers share types).
#pragma once

namespace {

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
    __shared__ float2 bufs[8][2][260];  // two frames per wave (32 lanes each): the four dependent LDS passes of a frame are latency-bound
    for (int m = threadIdx.x; m < 512; m += 256) {
        float sn, cs;
        sincospif((float)m / 256.0f, &sn, &cs);
        root[m] = make_float2(cs, sn);
    }
    __syncthreads();
    const int lane = threadIdx.x & 31, half = (threadIdx.x >> 5) & 1, wave = threadIdx.x >> 6;
    const int M = sp.wfft / 2;            // 256 or 128
    const int rs = 512 / sp.wfft;         // stride of the N-th roots in the table
    float2 *A = bufs[wave * 2 + half][0], *Bf = bufs[wave * 2 + half][1];
    for (long long f0 = ((long long)blockIdx.x * 4 + wave) * 2; f0 < total_frames; f0 += (long long)gridDim.x * 8) {
        const bool live = f0 + half < total_frames;   // an odd tail frame: the idle half recomputes it, stores nothing
        const long long f = live ? f0 + half : f0;
        const float2 *xr = xri + f * sp.K;
        const float *pn = pnr + f * sp.K;
        for (int k = lane; k <= M; k += 32) {
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
        for (int k = lane; k < M; k += 32) {
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
        int Ns = 1, lg = 0;      // Ns = 1 << lg: sub-transform length so far (powers of two: masks and shifts, no division)
        const int mr = 512 / M;  // stride of the M-th roots in the table
        while (Ns * 4 <= M) {
            const int q4 = M / 4;
            for (int j = lane; j < q4; j += 32) {
                const int kk = j & (Ns - 1);
                const int tstep = (kk * (q4 * mr)) >> lg;    // index of e^{2 pi i kk / (4 Ns)} in the table
                const float2 v0 = src[j];
                float2 v1 = src[j + q4], v2 = src[j + 2 * q4], v3 = src[j + 3 * q4];
                v1 = cmul(v1, root[(tstep) & 511]);
                v2 = cmul(v2, root[(2 * tstep) & 511]);
                v3 = cmul(v3, root[(3 * tstep) & 511]);
                // inverse radix-4 butterfly (W4 = +i)
                const float2 s02 = make_float2(v0.x + v2.x, v0.y + v2.y), d02 = make_float2(v0.x - v2.x, v0.y - v2.y);
                const float2 s13 = make_float2(v1.x + v3.x, v1.y + v3.y), d13 = make_float2(v1.x - v3.x, v1.y - v3.y);
                const int base = ((j >> lg) << (lg + 2)) + kk;
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
            lg += 2;
        }
        if (Ns < M) {  // one radix-2 pass (M = 128)
            const int h = M / 2;
            for (int j = lane; j < h; j += 32) {
                const int kk = j & (Ns - 1);
                const float2 v0 = src[j];
                const float2 v1 = cmul(src[j + h], root[((kk * (h * mr)) >> lg) & 511]);
                const int base = ((j >> lg) << (lg + 1)) + kk;
                dst[base] = make_float2(v0.x + v1.x, v0.y + v1.y);
                dst[base + Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float2 *tmp = src; src = dst; dst = tmp;
        }
        float2 *yo = reinterpret_cast<float2 *>(ybuf + f * sp.window);  // window is even (checked on the host)
        if (live)
            for (int n = lane; 2 * n < sp.window; n += 32) yo[n] = src[n];
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
    // four samples per thread and step: their loads are independent, which is what keeps enough bytes in flight
    const int stride = gridDim.x * 256;
    for (long long n0 = (long long)blockIdx.x * 256 + threadIdx.x; n0 < nout; n0 += 4LL * stride) {
        double acc[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n = (int)(n0 + (long long)q * stride);  // an utterance stays far below 2^31 samples
            acc[q] = 0.0;
            if (n < nout) {
                int t0 = n - w + 1 <= 0 ? 0 : (n - w + s) / s;  // ceil((n - w + 1) / s)
                int t1 = n / s;
                if (t1 > T - 1) t1 = T - 1;
                for (int t = t0; t <= t1; t++) acc[q] += (double)ybuf[(ro + t) * w + (n - t * s)];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const long long n = n0 + (long long)q * stride;
            if (n < nout) {
                const int value = (int)floor(acc[q] / sp.corr);
                o[n] = fabsf((float)value) > 32767.f ? (value < 0 ? -32767 : 32767) : (int16_t)value;
            }
        }
    }
}

}  // namespace
