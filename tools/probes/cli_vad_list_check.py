"""After tools/cli_e2e.py --set S-NOISY-VAD --utts N --dir D: a random sample of the HTK / VAD files the command line wrote against the oracle,
each file started where the list's earlier files leave the VAD's majority filter (frame counts only).  python tools/probes/cli_vad_list_check.py D N"""
import os, sys, struct, ctypes, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ctucopy_amd import synth, engine as ceng
from oracle.oracle import Oracle
D, N = sys.argv[1], int(sys.argv[2])
cfg = "-fs 8000 -format_in raw -format_out htk -preset mfcc -nr_mode exten -nr_a 2 -vad burg -vad_out_mode vad -vad_cri_mode cepdist -vad_cepdist_mode lpc -vad_thr_mode adapt".split()
L = ceng.load_library()
ns = synth.lengths(synth.SET_NOISY, list(range(N)))
hidx, hi, hs = [], ctypes.c_int32(0), ctypes.c_int32(0)
for n in ns:
    hidx.append(hi.value)
    L.ctu_vad_ring_step(3, (int(n) - 120) // 80, ctypes.byref(hi), ctypes.byref(hs))
orc = Oracle(cfg)
rng = np.random.default_rng(1)
worst, bad, shifted = 0.0, 0, 0
sample = sorted(rng.choice(N, 40, replace=False).tolist())
for i in sample:
    u = np.fromfile(os.path.join(D, "in", "u%05d.raw" % i), dtype="<i2")
    orc.set_vad_ring(hidx[i])
    ref, rv = orc.process(u, want_vad=True, first_in_process=False)
    raw = open(os.path.join(D, "out", "u%05d.htk" % i), "rb").read()
    n = struct.unpack("<I", raw[:4])[0]
    got = np.frombuffer(raw[12:], dtype="<f4").reshape(n, -1)
    v = open(os.path.join(D, "out", "u%05d.vad" % i), "rb").read()
    ok = n == ref.shape[0] and v == bytes(rv) and np.array_equal(~got.any(axis=1), ~ref.any(axis=1))
    if not ok:
        bad += 1
        continue
    z = ~ref.any(axis=1)
    worst = max(worst, float((np.abs(got[~z] - ref[~z]) / np.maximum(np.abs(ref[~z]), 1.0)).max()))
    shifted += hidx[i] != 0
print("files checked %d (of %d in the list), mismatching %d, out of phase %d, worst rel err %.3g" % (len(sample), N, bad, shifted, worst))
