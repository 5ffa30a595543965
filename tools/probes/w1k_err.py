#!/usr/bin/env python3
"""Diagnostic: worst errors of the 1024-point random sweep (tests/test_gpu_parity.py::test_random_configurations_on_1024_points)
for wave1k_kernel and, with CTU_WAVE1K=0 in the environment, for bigfft_kernel."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctucopy_amd import Engine, CtuError
from oracle.oracle import Oracle, OracleError
from tests.util import sig, synth_utt
res = []
for seed in (17, 19):
    rng = np.random.default_rng(seed)
    utts = [sig("CS0")[:30000], synth_utt(56, 26000)]
    for _ in range(24):
        kind = str(rng.choice(["dctc", "logspec", "spec", "lpc", "lpa"]))
        ncep = int(rng.integers(4, 20))
        lpo = ncep if kind == "lpa" else int(rng.integers(ncep, 21))
        nb = int(rng.integers(max(lpo + 1, 8), 41))
        cfg = ["-fs", "16000", "-format_in", "raw", "-format_out", "htk", "-w", str(rng.choice([33, 40, 40.0625, 50, 64])),
               "-s", str(rng.choice([10, 10.0625, 16, 20])), "-preem", str(rng.choice([0.95, 0.97])),
               "-fb_scale", str(rng.choice(["mel", "bark", "lin", "expolog"])), "-fb_shape", str(rng.choice(["triang", "rect", "trapez"])),
               "-fb_definition", f"{nb}filters", "-fb_norm", str(rng.choice(["on", "off"])), "-fb_eqld", str(rng.choice(["on", "off"])),
               "-fb_inld", str(rng.choice(["on", "off"])), "-fea_kind", kind, "-fea_ncepcoefs", str(ncep), "-fea_lporder", str(lpo),
               "-fea_c0", str(rng.choice(["on", "off"])), "-fea_E", str(rng.choice(["on", "off"])), "-fea_lifter", str(int(rng.choice([0, 22])))]
        try:
            orc = Oracle(cfg); eng = Engine(cfg)
        except (OracleError, CtuError):
            continue
        per = []
        for u, g in zip(utts, eng.extract(utts)):
            ref = orc.process(u)
            per.append(float((np.abs(g - ref) / np.maximum(np.abs(ref), 1.0)).max()))
        res.append((max(per), per, eng.kernel_name(), " ".join(cfg[6:])))
res.sort(reverse=True)
for w, per, k, c in res[:8]:
    print(f"{w:.2e} (CS0 {per[0]:.2e}, synthetic {per[1]:.2e}) {k} {c}")
print("n", len(res), "over 1e-4:", sum(1 for r in res if r[0] > 1e-4))
