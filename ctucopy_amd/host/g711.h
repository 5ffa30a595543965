// G.711 a-law / mu-law expansion of the `ctucopy` executable's -format_in alaw|mulaw decoders.
// Kept in a header of its own so that tests can compile it stand-alone against the reference's table
// (oracle/_ref/libref_amulaw.so, tests/test_host_decoders.py).
#pragma once
#include <cstdint>

// G.711 expansion exactly as the reference computes it (src/io/amulaw.h:20-53): chord/step -> magnitude,
// then "2x amplification" in 16-bit wrap-around arithmetic.
inline int16_t g711_to_linear(uint8_t code, bool alaw) {
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
    if (out & 0x8000) out -= 65536;
    return (int16_t)out;
}

