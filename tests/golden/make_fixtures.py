#!/usr/bin/env python3
"""Regenerates the committed 16-utterance miniatures of the synthetic sets and the oracle's outputs on them.

  python tests/golden/make_fixtures.py

Inputs: ctucopy_amd/synth.py (S-MFCC / S-PLP / S-TRAP share the 16 kHz speech set; S-NOISY is the 8 kHz one), seed
20260101 + index, `mini` lengths (0.6-2.0 s).  Expected outputs: the CPU oracle (oracle/ctu_oracle.c) on BASELINE.json's
configurations C2..C5 (tests/util.py).  The reference itself cannot be built in this image (FFTW3 is missing, see
DESIGN.md section 2), so these are outputs of the restatement - fixtures that pin the oracle against drift and give the
GPU tests fixed targets - not reference-produced vectors.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ctucopy_amd import synth  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from tests.util import C2, C3, C4, C5  # noqa: E402

N = 16


def main():
    sets = {"smfcc": synth.SET_SPEECH, "snoisy": synth.SET_NOISY}
    pcm = {}
    for name, sid in sets.items():
        utts = [synth.utterance(sid, i, mini=True) for i in range(N)]
        pcm[name] = utts
        np.savez_compressed(os.path.join(HERE, f"{name}_mini_pcm.npz"), **{f"u{i:02d}": u for i, u in enumerate(utts)})
    out = {}
    for tag, cfg, src, n in (("c2", C2, "smfcc", N), ("c3", C3, "smfcc", N), ("c4", C4, "snoisy", N), ("c5", C5, "smfcc", 2)):
        orc = Oracle(cfg)
        for i in range(n):
            if orc.dims.do_vad:
                rows, vad = orc.process(pcm[src][i], want_vad=True)
                out[f"{tag}_vad_{i:02d}"] = vad
            else:
                rows = orc.process(pcm[src][i])
            out[f"{tag}_rows_{i:02d}"] = rows
    np.savez_compressed(os.path.join(HERE, "mini_expected.npz"), **out)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
