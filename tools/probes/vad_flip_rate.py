"""How often does the fp32 lattice of the fused Burg-cepstral VAD decide differently from the float64 oracle at scale?
N utterances of S-NOISY (C4) - the committed fixtures (16 utterances, the recordings) are identical byte for byte.
python tools/probes/vad_flip_rate.py [N]    (GPU box)"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ctucopy_amd import Engine, synth
from oracle.oracle import Oracle
from tests.util import C4
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
idx = list(range(N))
utts = [np.asarray(synth.utterance_c(synth.SET_NOISY, k)) for k in idx]
eng, orc = Engine(C4), Oracle(C4)
got, vads = eng.extract(utts, want_vad=True)
frames = flips = files = 0
runs = []
for u, v in zip(utts, vads):
    _, rv = orc.process(u, want_vad=True)
    d = np.asarray(v) != np.asarray(rv)
    frames += d.size
    flips += int(d.sum())
    files += bool(d.any())
    if d.any():
        # lengths of the runs of differing decisions (a cascade would show as long runs)
        e = np.flatnonzero(np.diff(np.concatenate([[0], d.astype(np.int8), [0]])))
        runs += list((e[1::2] - e[::2]).tolist())
print("utterances %d, frames %d: decisions differing from the oracle %d (%.4f %%) in %d utterances; runs of differing frames: %s" %
      (N, frames, flips, 100.0 * flips / max(frames, 1), files, sorted(runs, reverse=True)[:12]))
