"""The reference's list behaviour of the VAD majority filter (ctu_plan_set_vad_ring) on random lists: 300 utterances of 0..150 frames per
configuration, rows and decisions against the oracle's list mode.  python tools/probes/vad_list_fuzz.py   (GPU box)"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ctucopy_amd import Engine
from oracle.oracle import Oracle
from tests.util import C2, C4, synth_utt
import tests.test_gpu_parity as T
for name, cfg, fs, order in (("C4 fused", C4, 8000, 3), ("C2 energy order 3", C2 + "-vad_out_mode vad -vad_cri_mode energy -vad_thr_mode adapt".split(), 16000, 3),
                             ("C2 energy order 7", C2 + "-vad_out_mode vad -vad_cri_mode energy -vad_thr_mode perc -vad_filter_order 7".split(), 16000, 7),
                             ("C2 d_a drop", C2 + "-fea_delta d_a -vad_out_mode vad -vad_apply_mode drop -vad_cri_mode energy -vad_thr_mode perc".split(), 16000, 3)):
    rng = np.random.default_rng(order + fs)
    hop, pre = fs // 100, fs * 25 // 1000 - fs // 100
    lo = 6 if "-fea_delta" in cfg else (order // 2 + 1 if order > 3 else 0)   # order >= 5: no file shorter than the delay (the corner the ABI leaves out)
    frames = [int(x) for x in rng.integers(lo, 150, 300)]
    utts = [synth_utt(7000 + i, pre + hop * f + (hop // 2 if f == 0 else int(rng.integers(0, hop))), fs=fs) for i, f in enumerate(frames)]
    eng, orc = Engine(cfg), Oracle(cfg)
    got, vads = eng.extract(utts, want_vad=True, as_list_of_one_process=order)
    ref = orc.process_list(utts, want_vad=True)
    bad = 0; worst = 0.0; shifted = 0
    alone, _ = eng.extract(utts, want_vad=True)
    for i, (g, v, (r, rv)) in enumerate(zip(got, vads, ref)):
        if g.shape != r.shape or not np.array_equal(np.asarray(v), np.asarray(rv)):
            bad += 1; continue
        if r.size:
            z = ~r.any(axis=1)
            if not np.array_equal(~g.any(axis=1), z): bad += 1; continue
            if (~z).any(): worst = max(worst, float((np.abs(g[~z] - r[~z]) / np.maximum(np.abs(r[~z]), 1.0)).max()))
            shifted += not np.array_equal(g, alone[i])
    print("%-20s utterances %d, mismatching %d, rows differing from the in-phase run in %d, worst rel err %.3g" % (name, len(utts), bad, shifted, worst))
