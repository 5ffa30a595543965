"""One very long utterance (20 minutes) beside a short one: tile chains of thousands of tiles, a per-wave chain that walks 120 000
frames (exten, fused VAD), TRAP and delta contexts far from the edges.  Rows and decisions against the oracle.
python tools/probes/long_utterance.py      (GPU box)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ctucopy_amd import Engine
from oracle.oracle import Oracle
from tests.util import C2, C3, C4, C5, synth_utt
ONLY = sys.argv[1] if len(sys.argv) > 1 else None
for name, cfg, fs in (("C2", C2, 16000), ("C3", C3, 16000), ("C4+VAD", C4, 8000), ("C5", C5, 16000), ("C2 d_a + cms", C2 + ["-fea_delta", "d_a", "-fea_Z_exp", "0.98"], 16000)):
    if ONLY and ONLY not in name: continue
    minutes = 20
    long_u = np.concatenate([synth_utt(900 + k, fs * 60, fs=fs) for k in range(minutes)])
    utts = [long_u, synth_utt(55, fs * 2 + 77, fs=fs)]
    eng, orc = Engine(cfg), Oracle(cfg)
    t0 = time.time()
    if eng.dims.has_vad:
        got, vad = eng.extract(utts, want_vad=True)
    else:
        got, vad = eng.extract(utts), None
    t1 = time.time()
    worst = 0.0
    flips = 0
    for i, (u, g) in enumerate(zip(utts, got)):
        r = orc.process(u, want_vad=True) if vad is not None else orc.process(u)
        ref, rv = (r if vad is not None else (r, None))
        assert g.shape == ref.shape, (name, g.shape, ref.shape)
        worst = max(worst, float((np.abs(g - ref) / np.maximum(np.abs(ref), 1.0)).max()))
        if vad is not None:
            dv = np.asarray(vad[i]) != np.asarray(rv)
            flips += int(dv.sum())
            if dv.any():
                print("   differing decisions at frames", np.flatnonzero(dv).tolist()[:40])
    print("%-14s frames %d + %d  worst rel err %.3g%s  (engine %.2f s incl. upload)" % (name, got[0].shape[0], got[1].shape[0], worst,
          "" if vad is None else "  VAD bytes differing: %d" % flips, t1 - t0), flush=True)
