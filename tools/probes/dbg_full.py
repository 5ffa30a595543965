import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctucopy_amd import Engine
from oracle.oracle import Oracle
from tests.util import C2, synth_utt
cfg = C2 + sys.argv[1:]
u = synth_utt(21, 20000)
g = Engine(cfg).extract([u])[0]
r = Oracle(cfg).process(u)
e = np.abs(g - r) / np.maximum(np.abs(r), 1)
print("shape", g.shape, "max err", e.max())
print("bad frames", np.where(e.max(axis=1) > 1e-3)[0][:40], "count", int((e.max(axis=1) > 1e-3).sum()))
print("bad cols", np.where(e.max(axis=0) > 1e-3)[0])
np.set_printoptions(precision=4, suppress=True, linewidth=200)
print(g[:3]); print(r[:3])
