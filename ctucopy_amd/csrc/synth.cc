// Deterministic synthetic input sets (include/ctu_synth.h).  Integer arithmetic only: the numpy
// definition in ctucopy_amd/synth.py yields the same samples bit for bit.  Host code, no device.
#include "../../include/ctu_synth.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

namespace {

constexpr uint64_t GAMMA = 0x9E3779B97F4A7C15ull;
constexpr int64_t SEED0 = 20260101;
constexpr int64_t COEF[6] = {0, 32768, 16384, 10923, 8192, 6554};  // 32768 / h
// round(2^20 * 3000 * 10^(-snr/20) / 59825.9), snr = 5..15 dB: speech rms ~3000 over the rms of (white + pink) raw noise
constexpr int64_t SNR_MUL[11] = {29569, 26353, 23487, 20933, 18657, 16628, 14819, 13208, 11772, 10491, 9350};

inline uint64_t mix(uint64_t z) {  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t param(int64_t base, int k) { return mix((uint64_t)base + (uint64_t)(k + 1) * GAMMA); }

inline int64_t psin(int64_t p) {  // parabolic sine of a 32-bit phase, +-32768 peak
    int64_t q = p >> 16;
    const bool hi = q >= 32768;
    if (hi) q -= 32768;
    const int64_t y = (q * (32768 - q)) >> 13;
    return hi ? -y : y;
}

int fs_of(int set) { return set == CTU_SET_NOISY ? 8000 : 16000; }

int64_t length_of(int set, int64_t index, int mini) {
    const int64_t fs = fs_of(set);
    const int64_t lo = mini ? fs * 6 / 10 : fs * 3, hi = mini ? fs * 2 : fs * 15;
    return lo + (int64_t)(param(SEED0 + index, 0) % (uint64_t)(hi - lo + 1));
}

int64_t fill(int set, int64_t index, int mini, int16_t *out, int64_t cap) {
    const int64_t fs = fs_of(set), base = SEED0 + index;
    const int64_t n = std::min(length_of(set, index, mini), cap);
    const int H = 3 + (int)(param(base, 1) % 3);
    const int64_t inc_mid = ((int64_t)170 << 32) / fs, inc_dev = ((int64_t)80 << 32) / fs;
    const int64_t ginc = ((int64_t)10 << 32) / (fs * (13 + (int64_t)(param(base, 2) % 21)));
    const int64_t g0 = (int64_t)(param(base, 3) & 0xffffffffull), ph0 = (int64_t)(param(base, 4) & 0xffffffffull);
    const int64_t a0 = (int64_t)(param(base, 5) & 0xffffffffull), ainc = ((int64_t)4 << 32) / fs;
    int64_t off[6] = {0};
    for (int h = 1; h <= H; h++) off[h] = (int64_t)(param(base, 8 + h) & 0xffffffffull);
    const uint64_t nkey = param(base, 100);
    uint64_t pkey[6];
    for (int r = 0; r < 6; r++) pkey[r] = param(base, 101 + r);
    const int64_t snr_mul = SNR_MUL[param(base, 6) % 11];
    // gate segments of the noisy set: off for 0.5 s, then on / off runs of 0.3-1.5 s
    int64_t seg_end = fs / 2;
    int seg_on = 0, seg_j = 0;
    uint64_t acc = (uint64_t)ph0;  // phase accumulator (wraps mod 2^64; the low 32 bits are the phase)
    for (int64_t i = 0; i < n; i++) {
        const int64_t gl = (g0 + i * ginc) & 0xffffffffll;
        const int64_t t = gl >= ((int64_t)1 << 31) ? gl - ((int64_t)1 << 32) : gl;
        const int64_t tri = (t < 0 ? -t : t) - ((int64_t)1 << 30);
        acc += (uint64_t)(inc_mid + ((inc_dev * tri) >> 30));
        const int64_t ph = (int64_t)(acc & 0xffffffffull);
        int64_t v = 0;
        for (int h = 1; h <= H; h++) v += (psin((h * ph + off[h]) & 0xffffffffll) * COEF[h]) >> 15;
        const int64_t am = 19661 + ((13107 * psin((a0 + i * ainc) & 0xffffffffll)) >> 15);
        const int64_t speech = (((v * 5200) >> 15) * am) >> 15;
        const uint64_t z = mix(nkey + (uint64_t)(i + 1) * GAMMA);
        const int64_t g = (int64_t)((z & 0xffff) + ((z >> 16) & 0xffff) + ((z >> 32) & 0xffff) + (z >> 48)) - 131070;
        int64_t x;
        if (set == CTU_SET_SPEECH) {
            x = speech + ((g * 130) >> 14);
        } else {
            int64_t pink = 0;
            for (int r = 0; r < 6; r++) pink += (int64_t)(mix(pkey[r] + (uint64_t)((i >> r) + 1) * GAMMA) & 0xffff) - 32768;
            while (i >= seg_end) {
                const int64_t d = fs * 3 / 10 + (int64_t)(param(base, 200 + seg_j) % (uint64_t)(fs * 12 / 10 + 1));
                seg_end += d;
                seg_on ^= 1;
                seg_j++;
            }
            x = (seg_on ? speech : 0) + (((g + pink) * snr_mul) >> 20);
        }
        out[i] = (int16_t)std::max<int64_t>(-32768, std::min<int64_t>(32767, x));
    }
    return n;
}

}  // namespace

extern "C" {

int64_t ctu_synth_length(int32_t set, int64_t index, int32_t mini) { return length_of(set, index, mini); }

int64_t ctu_synth_fill(int32_t set, int64_t index, int32_t mini, int16_t *out, int64_t cap) {
    if (!out || cap < 0) return -1;
    return fill(set, index, mini, out, cap);
}

int64_t ctu_synth_fill_arena(int32_t set, const int64_t *indices, int32_t mini, int32_t n_utt, const int64_t *sample_off,
                             int16_t *out, int32_t n_threads) {
    if (!indices || !sample_off || !out || n_utt < 0) return -1;
    int nt = n_threads > 0 ? n_threads : std::min(64, (int)std::thread::hardware_concurrency());  // 0: one per hardware thread, at most 64
    nt = std::max(1, std::min(nt, std::max(1, (int)n_utt)));
    std::atomic<int> next(0);
    std::atomic<int64_t> total(0);
    auto work = [&] {
        int64_t mine = 0;
        for (int i = next.fetch_add(1); i < n_utt; i = next.fetch_add(1))
            mine += fill(set, indices[i], mini, out + sample_off[i], length_of(set, indices[i], mini));
        total += mine;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    return total.load();
}

}  // extern "C"
