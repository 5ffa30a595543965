"""Random chains behind hwss / fwss / 2fwss (round 4: the *ss modes ahead of energy columns, -fb_inld, magnitude spectra, LP kinds,
delta / stacking / CMS) and exten at 1024 points, each against the oracle on a short list of utterances.  Refusals must carry a reason.
python tools/probes/fuzz_ss.py [first_seed] [n]          (GPU box; ~2 s per configuration)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ctucopy_amd import Engine, CtuError, synth
from oracle.oracle import Oracle, OracleError
from tests.util import C2, synth_utt, sig

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad, soft, ran, refused = [], [], 0, 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    fs = int(rng.choice([8000, 16000]))
    base = f"-fs {fs} -format_in raw -format_out htk -preset mfcc -preem 0.97".split()
    w1k = fs == 16000 and rng.random() < 0.25
    cfg = list(base)
    if w1k:
        cfg += ["-w", "40", "-s", "10", "-nr_mode", "exten", "-nr_a", str(rng.choice(["1", "2", "1.5"])), "-nr_p", str(rng.choice(["0.95", "0.9"]))]
    else:
        cfg += ["-vad", "burg", "-nr_mode", str(rng.choice(["hwss", "fwss", "2fwss"]))]
        if rng.random() < 0.3: cfg += ["-nr_a", "2"]
        if rng.random() < 0.3: cfg += ["-nr_b", str(rng.choice(["0.8", "1.5"]))]
    kind = str(rng.choice(["dctc", "dctc", "spec", "logspec", "lpc", "lpa"]))
    if "hwss" in cfg and kind in ("logspec", "dctc", "lpc", "lpa"): kind = "spec"   # half-wave rectification leaves zeros: no logarithms
    cfg += ["-fea_kind", kind]
    if kind in ("lpc", "lpa"): cfg += ["-fea_lporder", str(int(rng.integers(6, 15)))]
    if rng.random() < 0.4: cfg += ["-fb_inld", "on", "-fb_eqld", "on"]
    if rng.random() < 0.25 and "hwss" not in cfg: cfg += ["-fb_power", "off"]
    if rng.random() < 0.5: cfg += ["-fea_E", "on"]
    if rng.random() < 0.2: cfg += ["-fea_rawenergy", "on"]
    if kind in ("dctc", "lpc"):
        r = rng.random()
        if r < 0.3: cfg += ["-fea_delta", str(rng.choice(["d", "d_a", "d_a_t"]))]
        elif r < 0.4: cfg += ["-fea_trap", "5"]
        if rng.random() < 0.3: cfg += [str(rng.choice(["-fea_Z_exp", "-fea_Z_block"])), "0.97" if "-fea_Z_exp" in cfg else "40"]
    step = 16000 // fs
    utts = [sig("CS0")[::step][:24000 // step].copy(), synth_utt(900 + seed, 30000 // step, fs=fs), sig("CS3")[::step][4000 // step:30000 // step].copy(),
            synth_utt(950 + seed, 14000 // step, fs=fs)]
    try:
        eng = Engine(cfg)
    except CtuError as e:
        refused += 1
        if not str(e): bad.append((seed, "refusal without a reason", " ".join(cfg)))
        continue
    try:
        orc = Oracle(cfg)
    except OracleError as e:
        bad.append((seed, "engine accepts what the oracle refuses: " + str(e)[:200], " ".join(cfg)))
        continue
    try:
        got = eng.extract(utts)
    except CtuError as e:
        bad.append((seed, "run: " + str(e)[:200], " ".join(cfg)))
        continue
    worst = 0.0
    for u, g in zip(utts, got):
        ref = orc.process(u)
        if g.shape != ref.shape:
            bad.append((seed, f"shape {g.shape} vs {ref.shape}", " ".join(cfg)))
            break
        if ref.size:
            fin = np.isfinite(ref)
            if not np.array_equal(np.isfinite(g), fin):
                bad.append((seed, "finite pattern differs", " ".join(cfg)))
                break
            e = (np.abs(g - ref)[fin] / np.maximum(np.abs(ref)[fin], 1.0)).max() if fin.any() else 0.0
            rn = float((np.abs(np.where(fin, g - ref, 0)).max(axis=1) / np.maximum(np.abs(np.where(fin, ref, 0)).max(axis=1), 1.0)).max())
            worst = max(worst, float(e))
            if e > 1e-3 or rn > 1e-4:
                # the subtraction modes' conditioning class on real recordings (tests/test_gpu_parity.py: a frame whose dominant bins meet the
                # noise estimate to four digits): reported, and counted as a mismatch only when gross or frequent
                ef = (np.abs(np.where(fin, g - ref, 0)) / np.maximum(np.abs(np.where(fin, ref, 1)), 1.0)).max(axis=1)
                outl = int((ef > 1e-3).sum())
                if e > 5e-2 or outl > max(2, ref.shape[0] // 50):
                    bad.append((seed, f"err {e:.3g} rown {rn:.3g} frames out {outl}/{ref.shape[0]}", " ".join(cfg)))
                    break
                soft.append((seed, f"err {e:.3g} rown {rn:.3g} frames out {outl}/{ref.shape[0]}"))
    ran += 1
    print(seed, eng.kernel_name()[:60], "worst %.2g" % worst, flush=True)
print("ran %d, refused %d, mismatches %d, frames in the conditioning class %d" % (ran, refused, len(bad), len(soft)))
for b in soft:
    print("  soft", b)
for b in bad:
    print(b)
