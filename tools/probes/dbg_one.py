import sys, numpy as np
sys.path.insert(0, '.')
from ctucopy_amd import Engine
from oracle.oracle import Oracle, OracleError
from ctucopy_amd import CtuError
from tests.util import sig, synth_utt
rng = np.random.default_rng(13); fs = 8000
utts = [sig("CS0")[:24000], synth_utt(55, 20000, fs=fs)]
for it in range(40):
    scale = str(rng.choice(["mel", "bark", "lin", "expolog"])); shape = str(rng.choice(["triang", "rect", "trapez"]))
    kind = str(rng.choice(["dctc", "logspec", "spec", "lpc", "lpa"])); ncep = int(rng.integers(4, 17))
    lpo = ncep if kind == "lpa" else int(rng.integers(ncep, 17))
    cfg = ["-fs", str(fs), "-format_in", "raw", "-format_out", "htk", "-w", str(rng.choice([20, 25, 32])), "-s", str(rng.choice([8, 10, 16])),
           "-preem", str(rng.choice([0, 0.95, 0.97])), "-fb_scale", scale, "-fb_shape", shape, "-fb_definition", f"{int(rng.integers(8, 33))}filters",
           "-fb_norm", str(rng.choice(["on", "off"])), "-fb_eqld", str(rng.choice(["on", "off"])), "-fb_inld", str(rng.choice(["on", "off"])),
           "-fb_power", str(rng.choice(["on", "off"])), "-nr_mode", str(rng.choice(["none", "none", "exten"])),
           "-fea_kind", kind, "-fea_ncepcoefs", str(ncep), "-fea_lporder", str(lpo), "-fea_c0", str(rng.choice(["on", "off"])),
           "-fea_E", str(rng.choice(["on", "off"])), "-fea_lifter", str(int(rng.choice([0, 22]))), "-remove_dc", str(rng.choice(["on", "off"]))]
    try: orc = Oracle(cfg)
    except OracleError: continue
    try: eng = Engine(cfg)
    except CtuError: continue
    for ui, (u, g) in enumerate(zip(utts, eng.extract(utts))):
        ref = orc.process(u)
        err = np.abs(g - ref) / np.maximum(np.abs(ref), 1)
        if err.max() > 3e-4:
            print(" ".join(cfg[6:]))
            i, j = np.unravel_index(err.argmax(), err.shape)
            print("utt", ui, "frame", i, "col", j, "of", g.shape, "gpu", g[i, j], "ref", ref[i, j])
            print("row gpu", g[i]); print("row ref", ref[i])
            bad = np.argwhere(err > 1e-4); print("bad frames", np.unique(bad[:, 0])[:20], "bad cols", np.unique(bad[:, 1]))
