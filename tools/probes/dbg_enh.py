import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctucopy_amd import Engine
from oracle.oracle import Oracle
from tests.util import sig, synth_utt
cfg = sys.argv[1:]
utts = [sig("CS0")[:30000], synth_utt(79, 16000 * 2 + 5)]
eng, orc = Engine(cfg), Oracle(cfg)
for u, g in zip(utts, eng.enhance(utts)):
    ref = orc.enhance(u)
    d = np.abs(g.astype(int) - ref.astype(int))
    bad = np.where(d > 2)[0]
    print("n", g.size, "max", d.max(), "mean", d.mean(), "bad count", bad.size, "first bad", bad[:10], "last bad", bad[-10:], "hop", eng.dims.wshift, "win", eng.dims.window)
