import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ctucopy_amd import Engine
from oracle.oracle import Oracle
from tests.util import C2, synth_utt, sig
from ctucopy_amd import synth
utts = [synth.utterance_c(synth.SET_SPEECH, i, True) for i in (1, 4)] + [sig("CS0")[:30000], synth_utt(21, 240 + 160 * 50 + 7)]
for extra in (["-w","20","-s","10"], ["-w","20","-s","10","-fea_kind","spec"], ["-w","24","-s","10"], ["-w","25","-s","10","-fea_E","on"], ["-w","17","-s","10"], ["-w","20","-s","10","-vad","file=/tmp/v.bin"]):
    cfg = C2 + ["-vad","burg","-nr_mode","fwss"] + extra
    if "file=/tmp/v.bin" in extra[-1]:
        open("/tmp/v.bin","wb").write(bytes(np.random.default_rng(1).choice(np.array([0,1],np.uint8), 100000)))
    try:
        eng, orc = Engine(cfg), Oracle(cfg)
    except Exception as e:
        print(extra, "ERR", e); continue
    got = eng.extract(utts)
    out = []
    for u, g in zip(utts, got):
        ref = orc.process(u)
        e = np.abs(g-ref)/np.maximum(np.abs(ref),1.0)
        bad = int((e.max(axis=1) > 1e-3).sum()) if ref.size else 0
        first = int(np.argmax(e.max(axis=1) > 1e-3)) if bad else -1
        out.append("%.2g(%d/%d@%d)" % (e.max() if ref.size else 0, bad, ref.shape[0], first))
    print(extra, eng.kernel_name(), out)
