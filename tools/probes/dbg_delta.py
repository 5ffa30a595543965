import sys, numpy as np
sys.path.insert(0, '.')
from ctucopy_amd import Engine
from oracle.oracle import Oracle
from tests.util import C2, synth_utt, sig
cfg = C2 + ["-fea_delta", "d_a"]
frames = [3, 5, 9, 63, 64, 65, 66, 127, 128, 129, 130, 0, 517]
utts = [synth_utt(300 + i, 240 + 160 * f + (i % 5)) for i, f in enumerate(frames)] + [sig("CS0")]
eng = Engine(cfg); got = eng.extract(utts); orc = Oracle(cfg)
for u, g, f in zip(utts, got, frames + [594]):
    ref = orc.process(u)
    if not ref.size: continue
    err = np.abs(g - ref) / np.maximum(np.abs(ref), 1)
    bad = np.argwhere(err > 1e-4)
    print(f, g.shape, err.max(), "bad rows", sorted(set(bad[:, 0]))[:12], "bad cols", sorted(set(bad[:, 1]))[:12])
