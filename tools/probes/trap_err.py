"""TRAP-DCT (C5): error of the device rows against the oracle on fixture and synthetic utterances (which split is built: see CTU_TRAP_F16)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctucopy_amd import Engine, synth
from oracle.oracle import Oracle
from tests.util import C5, sig
eng, orc = Engine(C5), Oracle(C5)
utts = [synth.utterance_c(synth.SET_SPEECH, i) for i in range(4)] + [sig("CS0"), sig("CS3")]
got = eng.extract(utts)
for u, g in zip(utts, got):
    r = orc.process(u)
    e = np.abs(g - r) / np.maximum(np.abs(r), 1.0)
    print("frames %5d  max rel err %.2e  mean %.2e  max |ref| %.1f" % (r.shape[0], e.max(), e.mean(), np.abs(r).max()))
