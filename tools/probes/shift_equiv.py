"""Shift equivariance probe: rows of an utterance with one hop dropped at the front against the original's rows (bitwise)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctucopy_amd import Engine, synth
from tests.util import C2
eng = Engine(C2)
u = synth.utterance_c(synth.SET_SPEECH, 17)
import sys as _s
H = int(_s.argv[1]) if len(_s.argv) > 1 else 1
full = eng.extract([u])[0]
sh = eng.extract([u[160 * H:].copy()])[0]
d = sh[1:] != full[1 + H:]
print("rows", full.shape, sh.shape, "differing elements", int(d.sum()), "rows with a difference", int(d.any(axis=1).sum()))
r = np.nonzero(d.any(axis=1))[0]
print("first differing rows (index in shifted[1:])", r[:20], "mod 8:", (r[:40] % 8))
print("max abs diff", float(np.abs(sh[1:] - full[1 + H:]).max()))
