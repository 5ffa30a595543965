import sys, numpy as np
sys.path.insert(0, '.')
from ctucopy_amd import Engine, CtuError
from oracle.oracle import Oracle
from tests.util import sig, synth_utt
base = "-fs 16000 -format_in raw -format_out htk".split()
cfgs = {
 "10": "-w 32 -s 16 -preem 0.95 -fb_scale bark -fb_shape triang -fb_definition 19filters -fb_norm off -fb_eqld off -fb_inld on -fb_power on -nr_mode none -fea_kind logspec -fea_ncepcoefs 13 -fea_lporder 13 -fea_c0 off -fea_E on -fea_lifter 0 -remove_dc off",
 "12": "-w 32 -s 10 -preem 0.95 -fb_scale lin -fb_shape triang -fb_definition 27filters -fb_norm off -fb_eqld off -fb_inld off -fb_power on -nr_mode none -fea_kind lpc -fea_ncepcoefs 15 -fea_lporder 15 -fea_c0 off -fea_E on -fea_lifter 0 -remove_dc off",
 "23": "-w 20 -s 8 -preem 0.97 -fb_scale mel -fb_shape triang -fb_definition 19filters -fb_norm off -fb_eqld off -fb_inld off -fb_power on -nr_mode none -fea_kind lpc -fea_ncepcoefs 11 -fea_lporder 13 -fea_c0 off -fea_E on -fea_lifter 22 -remove_dc on",
 "29": "-w 25 -s 16 -preem 0.0 -fb_scale lin -fb_shape triang -fb_definition 12filters -fb_norm on -fb_eqld on -fb_inld off -fb_power off -nr_mode none -fea_kind logspec -fea_ncepcoefs 13 -fea_lporder 16 -fea_c0 off -fea_E on -fea_lifter 22 -remove_dc on",
 "30": "-w 20 -s 16 -preem 0.0 -fb_scale lin -fb_shape triang -fb_definition 26filters -fb_norm off -fb_eqld on -fb_inld off -fb_power off -nr_mode none -fea_kind lpc -fea_ncepcoefs 15 -fea_lporder 16 -fea_c0 on -fea_E on -fea_lifter 0 -remove_dc off",
 "36": "-w 32 -s 8 -preem 0.0 -fb_scale mel -fb_shape triang -fb_definition 16filters -fb_norm off -fb_eqld off -fb_inld off -fb_power on -nr_mode exten -fea_kind lpc -fea_ncepcoefs 16 -fea_lporder 16 -fea_c0 on -fea_E on -fea_lifter 0 -remove_dc on",
 "33": "-w 20 -s 16 -preem 0.95 -fb_scale mel -fb_shape triang -fb_definition 32filters -fb_norm off -fb_eqld off -fb_inld off -fb_power off -nr_mode none -fea_kind lpc -fea_ncepcoefs 9 -fea_lporder 14 -fea_c0 off -fea_E off -fea_lifter 0 -remove_dc on",
}
u = sig("CS0")[:24000]
for k, c in cfgs.items():
    cfg = base + c.split()
    g = Engine(cfg).extract([u])[0]; r = Oracle(cfg).process(u)
    err = np.abs(g - r) / np.maximum(np.abs(r), 1)
    with np.errstate(invalid='ignore'):
        colmax = np.nanmax(np.where(np.isfinite(err), err, np.nan), 0) if np.isfinite(err).any() else None
    print(k, g.shape, "nan cols", np.unique(np.argwhere(~np.isfinite(g))[:, 1]), "worst cols", np.argsort(-np.nan_to_num(err.max(0), nan=9e9))[:4], "max err", np.nanmax(err))
    print("   ref row0 tail", r[0, -3:], "gpu row0 tail", g[0, -3:], " ref row5", r[5, :3], "gpu", g[5, :3])
