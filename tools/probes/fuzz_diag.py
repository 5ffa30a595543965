import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ctucopy_amd import Engine
from oracle.oracle import Oracle
from tests.util import sig, synth_utt
for cfg, fs in (("-fs 8000 -format_in raw -format_out htk -w 25 -s 8 -preem 0.95 -fb_scale bark -fb_shape rect -fb_definition 30filters -fb_norm on -fb_eqld off -fb_inld on -fb_power on -nr_mode none -fea_kind spec -fea_ncepcoefs 15 -fea_lporder 16 -fea_c0 off -fea_E on -fea_lifter 0 -remove_dc on", 8000),
                ("-fs 16000 -format_in raw -format_out htk -w 50.0 -s 10.0 -preem 0.97 -fb_scale lin -fb_shape rect -fb_definition 25filters -fb_norm off -fb_eqld on -fb_inld off -fea_kind dctc -fea_ncepcoefs 16 -fea_lporder 17 -fea_c0 on -fea_E on -fea_lifter 22", 16000)):
    cfg = cfg.split()
    utts = [sig("CS0")[:30000], synth_utt(55, 26000, fs=fs)]
    eng, orc = Engine(cfg), Oracle(cfg)
    for u, g in zip(utts, eng.extract(utts)):
        ref = orc.process(u)
        err = np.abs(g - ref) / np.maximum(np.abs(ref), 1.0)
        t, c = np.unravel_index(np.argmax(err), err.shape)
        print(" ".join(cfg[8:20]), "| worst %.3g at frame %d col %d of %d: got %.7g ref %.7g | per-col max:" % (err.max(), t, c, g.shape[1], g[t, c], ref[t, c]), np.array2string(err.max(0), precision=1, max_line_width=250))
