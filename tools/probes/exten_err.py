#!/usr/bin/env python3
"""Error distribution of the GPU rows against the oracle on the S-NOISY / S-MFCC miniatures (diagnostic)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctucopy_amd import Engine, synth
from oracle.oracle import Oracle
from tests.util import C2, C4, C4_NOVAD

CFGS = {"C4_novad": (C4_NOVAD, synth.SET_NOISY), "C4": (C4, synth.SET_NOISY),
        "C4_novad_preem": (C4_NOVAD + ["-preem", "0.97"], synth.SET_NOISY),
        "C2_exten": (C2 + ["-nr_mode", "exten", "-nr_a", "2"], synth.SET_SPEECH), "C2": (C2, synth.SET_SPEECH)}
for name in sys.argv[1:] or list(CFGS):
    cfg, set_id = CFGS[name]
    utts = [synth.utterance_c(set_id, i, True) for i in range(16)]
    eng, orc = Engine(cfg), Oracle(cfg)
    got = eng.extract(utts, want_vad=True)
    errs, agree, total = [], 0, 0
    for i, u in enumerate(utts):
        ref = orc.process(u, want_vad=True) if eng.dims.has_vad else (orc.process(u), None)
        e = np.abs(got[0][i] - ref[0]) / np.maximum(np.abs(ref[0]), 1.0)
        errs.append(e.ravel())
        if ref[1] is not None:
            agree += int((got[1][i] == ref[1]).sum()); total += ref[1].size
    e = np.concatenate(errs)
    print(f"{name}: entries {e.size} max {e.max():.3e} p99.9 {np.quantile(e, 0.999):.3e} p99 {np.quantile(e, 0.99):.3e} "
          f"median {np.median(e):.3e} frac>1e-4 {float((e > 1e-4).mean()):.2e}" + (f" vad agree {agree}/{total}" if total else ""), flush=True)
