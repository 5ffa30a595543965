// G.711 expansion on the device (row N4: the a-law / mu-law decoders of src/io/in.cc:481-560 on the fast path).
// Included by engine.hip.
#pragma once

namespace {

// One code -> one int16 sample, exactly as the reference computes it (src/io/amulaw.h:20-53): chord / step -> magnitude,
// then the "2x amplification" in 16-bit wrap-around arithmetic.  Eight codes per lane: 8-byte loads, 16-byte stores.
__device__ __forceinline__ int g711_expand(int code, bool alaw) {
    const int a = (int)(int8_t)code;  // the reference works on a (signed) char
    const int sgn = (~(a >> 7)) & 1;
    int mag;
    if (!alaw) {
        const int chord = (~(a >> 4)) & 7, step = (~a) & 0xf;
        mag = (((2 * step) + 33) << chord) - 33;
    } else {
        int chord = ((a ^ 0x55) >> 4) & 7;
        const int step = (a ^ 0x55) & 0xf;
        mag = (step << 1) + 1;
        if (chord > 0) mag += 32;
        else chord = 1;
        mag <<= chord;
    }
    int out = ((1 - 2 * sgn) * mag) & 0xffff;
    out = (out << 2) & 0xffff;
    return out;  // low 16 bits = the sample
}

__global__ __launch_bounds__(256) void g711_kernel(const uint8_t *__restrict__ codes, int16_t *__restrict__ pcm, int64_t n, int alaw) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n && ((uintptr_t)(codes + i) & 7) == 0 && ((uintptr_t)(pcm + i) & 15) == 0) {
            const uint2 c = *reinterpret_cast<const uint2 *>(codes + i);
            uint4 o;
            o.x = g711_expand(c.x & 255, alaw) | (g711_expand((c.x >> 8) & 255, alaw) << 16);
            o.y = g711_expand((c.x >> 16) & 255, alaw) | (g711_expand(c.x >> 24, alaw) << 16);
            o.z = g711_expand(c.y & 255, alaw) | (g711_expand((c.y >> 8) & 255, alaw) << 16);
            o.w = g711_expand((c.y >> 16) & 255, alaw) | (g711_expand(c.y >> 24, alaw) << 16);
            *reinterpret_cast<uint4 *>(pcm + i) = o;
        } else {
            for (int64_t k = i; k < n && k < i + 8; k++) pcm[k] = (int16_t)g711_expand(codes[k], alaw);
        }
    }
}

// -remove_dc1 (src/io/in.cc:343-350): before anything else a frame subtracts the mean of the sample buffer from the
// buffer itself, so a sample carries the offsets of every frame it has been part of.  With o_t the offset of frame t and
// m_t the mean of the frame's untouched samples,
//     o_t = m_t - sum_{j >= 1, j*wshift < window} (window - j*wshift) / window * o_{t-j}          (o_u = 0 for u < 0)
// and the sample at position i of frame t is read as  x - o_t - sum_{j >= 1, i <= window-1-j*wshift} o_{t-j}  (the front
// end applies that; the sample ahead of the frame, position -1, without the o_t term).
// dc1_means: one wave per frame chunk, m_t in double (integer sums: exact).  dc1_offsets: one lane per utterance walks
// the recurrence in double and stores the offsets as floats.
__global__ __launch_bounds__(256) void dc1_means_kernel(const int16_t *__restrict__ pcm, const int4 *__restrict__ utt_info,
                                                        const long long *__restrict__ sample_off, double *__restrict__ mean, int n_utt,
                                                        int window, int wshift) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blockIdx.y;
    const int4 ui = utt_info[u];
    const int64_t r0 = ((int64_t)ui.y << 32) | (uint32_t)ui.x;
    const int T = ui.z;
    const int16_t *x = pcm + sample_off[u];
    for (int t = blockIdx.x * 4 + wave; t < T; t += gridDim.x * 4) {
        int sum = 0;  // |sum| <= 32768 * window: fits for windows up to 65535 samples
        for (int i = lane; i < window; i += 64) sum += x[(int64_t)t * wshift + i];
        for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) mean[r0 + t] = (double)sum / (double)window;
    }
}

__global__ __launch_bounds__(64) void dc1_offsets_kernel(const double *__restrict__ mean, const int4 *__restrict__ utt_info, float *__restrict__ off,
                                                         int n_utt, int window, int wshift) {
    const int u = blockIdx.x * 64 + threadIdx.x;
    if (u >= n_utt) return;
    const int4 ui = utt_info[u];
    const int64_t r0 = ((int64_t)ui.y << 32) | (uint32_t)ui.x;
    const int T = ui.z;
    constexpr int JM = 8;  // the engine refuses window / wshift > 8
    double hist[JM] = {0, 0, 0, 0, 0, 0, 0, 0};  // o_{t-1} .. o_{t-8}
    double cj[JM];
#pragma unroll
    for (int j = 1; j <= JM; j++) cj[j - 1] = (j * wshift < window) ? (double)(window - j * wshift) / (double)window : 0.0;
    for (int t = 0; t < T; t++) {
        double o = mean[r0 + t];
#pragma unroll
        for (int j = 0; j < JM; j++) o -= cj[j] * hist[j];
#pragma unroll
        for (int j = JM - 1; j > 0; j--) hist[j] = hist[j - 1];
        hist[0] = o;
        off[r0 + t] = (float)o;
    }
}

}  // namespace
