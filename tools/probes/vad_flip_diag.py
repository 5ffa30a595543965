import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ctucopy_amd import Engine
from oracle.oracle import Oracle
from tests.util import C4, synth_utt
fs = 8000
eng, orc = Engine(C4), Oracle(C4)
segs = [synth_utt(900 + k, fs * 60, fs=fs) for k in range(20)]
cases = {"minute 19 alone": segs[19], "minutes 18-19": np.concatenate(segs[18:]), "minutes 10-19": np.concatenate(segs[10:]), "all 20": np.concatenate(segs),
         "all 20, first 8 samples dropped": np.concatenate(segs)[8:], "all 20, first 80 dropped (one hop)": np.concatenate(segs)[80:]}
for name, u in cases.items():
    got, vads = eng.extract([u], want_vad=True)
    ref, rv = orc.process(u, want_vad=True)
    d = np.asarray(vads[0]) != np.asarray(rv)
    print("%-36s frames %6d  differing %3d at %s   rows worst %.3g" % (name, d.size, int(d.sum()), (np.flatnonzero(d)[:6] - d.size).tolist(), float((np.abs(got[0]-ref)/np.maximum(np.abs(ref),1)).max())))
