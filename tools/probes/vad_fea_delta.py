"""-vad_cepdist_mode fea behind a delta / stacking chain (and CMS): decisions and rows against the oracle.  python tools/probes/vad_fea_delta.py"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ctucopy_amd import Engine, CtuError
from oracle.oracle import Oracle
from tests.util import C2, C3, sig, synth_utt
vadf = "-vad_out_mode vad -vad_cri_mode cepdist -vad_cepdist_mode fea -vad_thr_mode adapt".split()
utts = [sig("CS0"), synth_utt(41, 50000), synth_utt(42, 240 + 160 * 9)]
for extra in (["-fea_delta", "d_a"], ["-fea_delta", "d"], ["-fea_trap", "3"], ["-fea_delta", "d_a_t", "-fea_E", "on"], ["-fea_delta", "d_a", "-fea_Z_exp", "0.98"],
              ["-fea_delta", "d_a", "-vad_filter_order", "5"], ["-fea_delta", "d_a", "-vad_thr_mode", "dyn"]):
    for base in (C2, C3):
        cfg = base + vadf + extra
        try:
            eng, orc = Engine(cfg), Oracle(cfg)
        except Exception as e:
            print(" ".join(extra), "REFUSED/ERR", str(e)[:100]); continue
        got, vads = eng.extract(utts, want_vad=True)
        flips = 0; worst = 0; n = 0
        for u, g, v in zip(utts, got, vads):
            r, rv = orc.process(u, want_vad=True)
            assert g.shape == r.shape and len(v) == len(rv), (g.shape, r.shape)
            flips += int((np.asarray(v) != np.asarray(rv)).sum()); n += len(rv)
            worst = max(worst, float((np.abs(g - r) / np.maximum(np.abs(r), 1.0)).max()))
        print("%-6s %-40s decisions differing %d of %d (ones %d), rows worst %.3g" % ("C2" if base is C2 else "C3", " ".join(extra), flips, n, sum(int((np.asarray(x) == ord("1")).sum()) for x in vads), worst))
