#!/usr/bin/env python3
"""Diagnostic: where do the 16 kHz *ss rows leave the oracle's (first frame beyond 1e-3, error profile)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctucopy_amd import Engine, synth
from oracle.oracle import Oracle
from tests.util import C2, sig, synth_utt
utts = [synth.utterance_c(synth.SET_SPEECH, i, True) for i in (1, 4, 6)] + [sig("CS0")[:40000], synth_utt(18, 240), sig("CS3")[:30000], synth_utt(21, 240 + 160 * 5 + 7)]
for extra in (["-nr_mode", "2fwss"], ["-nr_mode", "2fwss", "-fea_kind", "spec"], ["-nr_mode", "fwss", "-fea_kind", "spec"], ["-nr_mode", "hwss", "-fea_kind", "spec"],
              ["-nr_mode", "fwss", "-nr_a", "2", "-nr_b", "1.5", "-fea_kind", "logspec"]):
    cfg = C2 + ["-vad", "burg"] + extra
    got, orc = Engine(cfg).extract(utts), Oracle(cfg)
    for i, (u, g) in enumerate(zip(utts, got)):
        ref = orc.process(u)
        if not ref.size:
            continue
        e = np.abs(g - ref) / np.maximum(np.abs(ref), 1.0)
        rn = np.abs(g - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1.0)
        bad = np.nonzero(e.max(axis=1) > 1e-3)[0]
        print(" ".join(extra), f"utt {i} frames {ref.shape[0]} el {np.nanmax(e):.2e} rn {np.nanmax(rn):.2e} first bad {bad[:6].tolist()} nbad {bad.size} nan {int(np.isnan(g).sum())}/{int(np.isnan(ref).sum())}", flush=True)
