"""-nr_when afterFB together with the VAD module: decisions and rows against the oracle.  python tools/probes/afterfb_vad.py  (GPU box)"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ctucopy_amd import Engine, CtuError
from oracle.oracle import Oracle, OracleError
from tests.util import C2, sig, synth_utt
from ctucopy_amd import synth
for fs in (16000, 8000):
    base = f"-fs {fs} -format_in raw -format_out htk -preset mfcc -preem 0.97 -nr_mode exten -nr_when afterFB -vad_out_mode vad".split()
    utts = [synth_utt(61, fs * 2, fs=fs), synth_utt(62, fs // 100 * 70 + fs // 40, fs=fs), synth.utterance_c(synth.SET_NOISY if fs == 8000 else synth.SET_SPEECH, 6, True)]
    for extra in ("-vad_cri_mode energy -vad_thr_mode adapt", "-vad_cri_mode energy -vad_thr_mode dyn", "-vad burg -vad_cri_mode cepdist -vad_cepdist_mode lpc -vad_thr_mode adapt",
                  "-vad_cri_mode cepdist -vad_cepdist_mode fea -vad_thr_mode adapt", "-vad_cri_mode energy -vad_thr_mode perc -vad_apply_mode drop", "-vad_cri_mode energy -vad_thr_mode adapt -fea_delta d_a"):
        cfg = base + extra.split()
        try:
            orc = Oracle(cfg)
        except OracleError as e:
            print(fs, extra, "ORACLE REFUSES", e); continue
        try:
            eng = Engine(cfg)
        except CtuError as e:
            print(fs, extra, "ENGINE REFUSES", str(e)[:120]); continue
        got, vads = eng.extract(utts, want_vad=True)
        flips = n = 0; worst = 0.0; ok = True
        for u, g, v in zip(utts, got, vads):
            r, rv = orc.process(u, want_vad=True)
            if g.shape != r.shape or len(v) != len(rv): ok = False; print("   shape", g.shape, r.shape, len(v), len(rv)); continue
            flips += int((np.asarray(v) != np.asarray(rv)).sum()); n += len(rv)
            if r.size: worst = max(worst, float((np.abs(g - r) / np.maximum(np.abs(r), 1.0)).max()))
        print(fs, "%-80s decisions differing %d of %d, rows worst %.3g %s" % (extra, flips, n, worst, "" if ok else "SHAPES DIFFER"))
