import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctucopy_amd import Engine, synth
from oracle.oracle import Oracle
from tests.util import sig, synth_utt
SS8 = "-fs 8000 -format_in raw -format_out htk -preset mfcc -preem 0.97 -vad burg".split()
cfg = SS8 + sys.argv[1:]
utts = [synth.utterance_c(synth.SET_NOISY, i, True) for i in (2, 5, 9)] + [sig("CS3")[:30000]]
got = Engine(cfg).extract(utts)
orc = Oracle(cfg)
np.set_printoptions(precision=4, suppress=True, linewidth=220)
for i, (u, g) in enumerate(zip(utts, got)):
    r = orc.process(u)
    e = np.abs(g - r) / np.maximum(np.abs(r), 1)
    fe = e.max(axis=1)
    bad = np.where(fe > 1e-3)[0]
    print(f"utt {i} frames {g.shape[0]} max {e.max():.3e} median-frame {np.median(fe):.2e} first bad frames {bad[:12]} n_bad {bad.size}")
    if bad.size:
        t = bad[0]
        print("  gpu", g[t]); print("  ref", r[t])
