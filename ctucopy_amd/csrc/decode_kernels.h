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

}  // namespace
