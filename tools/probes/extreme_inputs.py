"""Inputs at the edges of int16: digital silence (the reference's unguarded log gives -inf / NaN rows, src/fea/fea_impl.cc:109), a
full-scale square wave, full-scale noise, a single impulse.  Finite values against the oracle, non-finite ones by pattern.
python tools/probes/extreme_inputs.py    (GPU box)"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ctucopy_amd import Engine
from oracle.oracle import Oracle
from tests.util import C2, C3, C4_NOVAD, C5
rng = np.random.default_rng(5)
n = 16000
sigs = {"silence": np.zeros(n, np.int16), "square full scale": np.where((np.arange(n) // 40) % 2, 32767, -32768).astype(np.int16),
        "noise full scale": rng.integers(-32768, 32768, n).astype(np.int16), "impulse": np.eye(1, n, 5000, dtype=np.int16)[0] * 32767,
        "silence then tone": np.concatenate([np.zeros(n // 2, np.int16), (8000 * np.sin(np.arange(n // 2) * 0.3)).astype(np.int16)])}
for cname, cfg in (("C2", C2), ("C3", C3), ("C4 exten", C4_NOVAD), ("C5", C5)):
    eng, orc = Engine(cfg), Oracle(cfg)
    for sname, x in sigs.items():
        if "C4" in cname: x = x[::2].copy()
        with np.errstate(all="ignore"):
            g, r = eng.extract([x])[0], orc.process(x)
            fin = np.isfinite(r) & np.isfinite(g)
            same_pattern = bool(np.array_equal(np.isnan(g), np.isnan(r)) and np.array_equal(np.isposinf(g), np.isposinf(r)) and np.array_equal(np.isneginf(g), np.isneginf(r)))
            err = float((np.abs(g - r) / np.maximum(np.abs(r), 1.0))[fin].max()) if fin.any() else 0.0
        print("%-9s %-18s rows %4d  finite %6.2f %%  worst rel err on finite %.3g  non-finite pattern %s" % (cname, sname, g.shape[0], 100.0 * np.isfinite(r).mean(), err, "same" if same_pattern else "DIFFERS"))
# round 4: the fused Burg criterion takes its reflection coefficient by v_rcp_f32 + one correction - on digital silence the lattice's
# denominator is 0 either way (0 * inf = NaN as 0 / 0): the decisions must still be the oracle's
from tests.util import C4
eng, orc = Engine(C4), Oracle(C4)
for sname, x in sigs.items():
    x = x[::2].copy()
    with np.errstate(all="ignore"):
        (g,), (v,) = eng.extract([x], want_vad=True)
        r, rv = orc.process(x, want_vad=True)
    print("%-9s %-18s rows %4d  vad bytes %4d differing %d" % ("C4 + VAD", sname, g.shape[0], v.size, int((v != rv).sum()) if v.size == rv.size else -1))
