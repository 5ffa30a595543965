"""Shared test helpers: configurations of BASELINE.json and fixture loading."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# egs/conf/01_mfcc_0_16k_2510_30.ctuconf of the reference, written out as a command line (C1).
C1 = ("-fs 16000 -format_in raw -format_out htk -endian_in little -endian_out little -w 25 -s 10 -preem 0.97 "
      "-fb_scale mel -fb_shape triang -fb_power on -fb_definition 30filters -nr_mode none -fb_eqld off "
      "-fb_inld off -fea_kind dctc -fea_ncepcoefs 12 -fea_c0 on -fea_E off -fea_lifter 22 -fea_rawenergy off").split()
# SURVEY.md Appendix B command lines (C2..C5)
C2 = "-fs 16000 -format_in raw -format_out htk -preset mfcc -preem 0.97".split()
C3 = "-fs 16000 -format_in raw -format_out htk -preset plpc".split()
C4 = ("-fs 8000 -format_in raw -format_out htk -preset mfcc -nr_mode exten -nr_a 2 -vad burg -vad_out_mode vad "
      "-vad_cri_mode cepdist -vad_cepdist_mode lpc -vad_thr_mode adapt").split()
C4_NOVAD = "-fs 8000 -format_in raw -format_out htk -preset mfcc -nr_mode exten -nr_a 2".split()
C5 = ("-fs 16000 -format_in raw -format_out htk -preset mfcc -fb_definition 23filters "
      "-fea_kind trapdct,101,16").split()


def sig(name):
    """Bundled reference signals (egs/sig of the reference: 16 kHz int16 LE raw)."""
    return np.fromfile(os.path.join(GOLDEN, "SA000CB1." + name), dtype="<i2")


def synth_utt(seed, nsamples, fs=16000, noise=300.0):
    """Small deterministic speech-like utterance for tests (harmonics of a gliding f0 + AM + white noise)."""
    rng = np.random.default_rng(seed)
    t = np.arange(nsamples) / fs
    f0 = 90 + 160 * (0.5 + 0.5 * np.sin(2 * np.pi * 0.31 * t + rng.uniform(0, 6.28)))
    ph = 2 * np.pi * np.cumsum(f0) / fs
    x = np.zeros(nsamples)
    for h in range(1, 5):
        x += np.sin(h * ph + rng.uniform(0, 6.28)) / h
    x *= 6000 * (0.6 + 0.4 * np.sin(2 * np.pi * 4 * t))
    x += rng.normal(0, noise, nsamples)
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)
