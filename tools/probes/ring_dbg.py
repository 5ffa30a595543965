import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ctucopy_amd import Engine
from oracle.oracle import Oracle
from tests.util import C2, synth_utt
for cfg, order in ((C2 + "-fea_delta d_a -fea_E on -vad_out_mode vad -vad_cri_mode energy -vad_thr_mode dyn".split(), 3),
                   (C2 + "-vad_out_mode vad -vad_apply_mode drop -vad_cri_mode energy -vad_thr_mode adapt -vad_filter_order 5".split(), 5)):
    frames = [17, 9, 31, 8, 1 if order == 3 else 12, 25, 0, 14, 40]
    utts = [synth_utt(400 + i, 240 + 160 * T + (80 if T == 0 else 3 * i)) for i, T in enumerate(frames)]
    eng, orc = Engine(cfg), Oracle(cfg)
    print("ring", eng.vad_ring_of_list([len(u) for u in utts], order))
    got, vads = eng.extract(utts, want_vad=True, as_list_of_one_process=order)
    alone, va = eng.extract(utts, want_vad=True)
    ref = orc.process_list(utts, want_vad=True)
    ral = [Oracle(cfg).process(u, want_vad=True) for u in utts]
    for i in range(len(utts)):
        print(i, frames[i], "got", got[i].shape, "alone", alone[i].shape, "ref", ref[i][0].shape, "ref alone", ral[i][0].shape,
              "got==alone", np.array_equal(got[i], alone[i]), "ref==refalone", np.array_equal(ref[i][0], ral[i][0]), bytes(vads[i])[:8], bytes(ref[i][1])[:8])
