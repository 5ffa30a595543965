"""Worst errors of the seeded random configuration sweep, exten cases listed separately (diagnostic)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctucopy_amd import Engine, CtuError
from oracle.oracle import Oracle, OracleError
from tests.util import sig, synth_utt
res = []
for seed, fs in ((7, 16000), (11, 16000), (13, 8000)):
    rng = np.random.default_rng(seed)
    utts = [sig("CS0")[:24000], synth_utt(55, 20000, fs=fs)]
    for _ in range(40):
        scale = str(rng.choice(["mel", "bark", "lin", "expolog"])); shape = str(rng.choice(["triang", "rect", "trapez"]))
        kind = str(rng.choice(["dctc", "logspec", "spec", "lpc", "lpa"])); ncep = int(rng.integers(4, 17))
        lpo = ncep if kind == "lpa" else int(rng.integers(ncep, 17))
        cfg = ["-fs", str(fs), "-format_in", "raw", "-format_out", "htk", "-w", str(rng.choice([20, 25, 32])), "-s", str(rng.choice([8, 10, 16])),
               "-preem", str(rng.choice([0, 0.95, 0.97])), "-fb_scale", scale, "-fb_shape", shape, "-fb_definition", f"{int(rng.integers(8, 33))}filters",
               "-fb_norm", str(rng.choice(["on", "off"])), "-fb_eqld", str(rng.choice(["on", "off"])), "-fb_inld", str(rng.choice(["on", "off"])),
               "-fb_power", str(rng.choice(["on", "off"])), "-nr_mode", str(rng.choice(["none", "none", "exten"])),
               "-fea_kind", kind, "-fea_ncepcoefs", str(ncep), "-fea_lporder", str(lpo), "-fea_c0", str(rng.choice(["on", "off"])),
               "-fea_E", str(rng.choice(["on", "off"])), "-fea_lifter", str(int(rng.choice([0, 22]))), "-remove_dc", str(rng.choice(["on", "off"]))]
        try:
            orc = Oracle(cfg); eng = Engine(cfg)
        except (OracleError, CtuError):
            continue
        worst = rown = 0.0
        for u, g in zip(utts, eng.extract(utts)):
            ref = orc.process(u)
            worst = max(worst, float((np.abs(g - ref) / np.maximum(np.abs(ref), 1.0)).max()))
            rown = max(rown, float((np.abs(g - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1.0)).max()))
        res.append((worst, rown, "exten" in cfg, " ".join(cfg)))
res.sort(reverse=True)
for w, r, ex, c in res[:12]:
    print(f"{w:.2e} rown {r:.2e} {'EXTEN' if ex else '     '} {c}")
print("n", len(res), "exten>1e-4:", sum(1 for w, r, ex, c in res if ex and w > 1e-4), "plain>1e-4:", sum(1 for w, r, ex, c in res if not ex and w > 1e-4))
