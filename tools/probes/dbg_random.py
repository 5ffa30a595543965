import sys, numpy as np
sys.path.insert(0, '.')
from ctucopy_amd import Engine, CtuError
from oracle.oracle import Oracle, OracleError
from tests.util import sig, synth_utt
rng = np.random.default_rng(7)
utts = [sig("CS0")[:24000], synth_utt(55, 20000)]
for it in range(40):
    scale = str(rng.choice(["mel", "bark", "lin", "expolog"]))
    shape = str(rng.choice(["triang", "rect", "trapez"]))
    kind = str(rng.choice(["dctc", "logspec", "spec", "lpc", "lpa"]))
    ncep = int(rng.integers(4, 17))
    lpo = ncep if kind == "lpa" else int(rng.integers(ncep, 17))
    cfg = ["-fs", "16000", "-format_in", "raw", "-format_out", "htk", "-w", str(rng.choice([20, 25, 32])), "-s", str(rng.choice([8, 10, 16])),
           "-preem", str(rng.choice([0, 0.95, 0.97])), "-fb_scale", scale, "-fb_shape", shape, "-fb_definition", f"{int(rng.integers(8, 33))}filters",
           "-fb_norm", str(rng.choice(["on", "off"])), "-fb_eqld", str(rng.choice(["on", "off"])), "-fb_inld", str(rng.choice(["on", "off"])),
           "-fb_power", str(rng.choice(["on", "off"])), "-nr_mode", str(rng.choice(["none", "none", "exten"])),
           "-fea_kind", kind, "-fea_ncepcoefs", str(ncep), "-fea_lporder", str(lpo), "-fea_c0", str(rng.choice(["on", "off"])),
           "-fea_E", str(rng.choice(["on", "off"])), "-fea_lifter", str(int(rng.choice([0, 22]))), "-remove_dc", str(rng.choice(["on", "off"]))]
    try:
        orc = Oracle(cfg)
    except OracleError as e:
        print(it, "oracle refuses:", e); continue
    try:
        eng = Engine(cfg)
    except CtuError as e:
        print(it, "engine refuses:", str(e)[:90]); continue
    worst = 0
    for u, g in zip(utts, eng.extract(utts)):
        ref = orc.process(u)
        bad = not np.isfinite(g).all()
        err = float((np.abs(g - ref) / np.maximum(np.abs(ref), 1)).max()) if not bad else float('nan')
        worst = max(worst, err) if not bad else float('nan')
        if bad: print("   ref finite:", np.isfinite(ref).all(), "nan cols", np.unique(np.argwhere(~np.isfinite(g))[:,1])[:20])
    print(it, "err %.2e" % worst, " ".join(cfg[6:]))
