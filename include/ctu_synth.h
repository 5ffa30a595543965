/*
 * ctu_synth.h -- deterministic synthetic input sets of the benchmark and of the parity fixtures.
 *
 * Not part of the reference's interface: the reference ships no generator (its examples read recordings).  The sets
 * are defined by integer arithmetic only (splitmix64 streams, 32-bit phase accumulators, a parabolic sine), so that
 * this C implementation and the numpy one in ctucopy_amd/synth.py produce the same int16 samples bit for bit
 * (tests/test_synth.py).  No device is touched.
 *
 *   CTU_SET_SPEECH  S-MFCC / S-PLP / S-TRAP: 16 kHz, 3-5 harmonics of an f0 gliding 90-250 Hz, 4 Hz amplitude
 *                   modulation, white noise sigma ~300 LSB, peak ~ +-12000; never digitally silent.
 *   CTU_SET_NOISY   S-NOISY: 8 kHz, the same source gated on/off in 0.3-1.5 s bursts, first 0.5 s noise only,
 *                   white + pink noise at 5-15 dB SNR.
 *
 * Utterance `index` of a set uses seed 20260101 + index.  `mini` != 0 selects the short lengths of the committed
 * 16-utterance fixtures (0.6-2.0 s) instead of the benchmark's 3-15 s.
 */
#ifndef CTU_SYNTH_H
#define CTU_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CTU_SET_SPEECH = 0, CTU_SET_NOISY = 1 };

/* samples of utterance `index` (a function of the seed alone) */
int64_t ctu_synth_length(int32_t set, int64_t index, int32_t mini);

/* writes the utterance (ctu_synth_length samples; fewer if cap is smaller) and returns the number written */
int64_t ctu_synth_fill(int32_t set, int64_t index, int32_t mini, int16_t *out, int64_t cap);

/* fills a packed arena: utterance indices[i] goes to out + sample_off[i]; n_threads host threads (0 = all cores) */
int64_t ctu_synth_fill_arena(int32_t set, const int64_t *indices, int32_t mini, int32_t n_utt, const int64_t *sample_off,
                             int16_t *out, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
