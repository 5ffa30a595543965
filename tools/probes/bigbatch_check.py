#!/usr/bin/env python3
"""Large batches (thousands of utterances: several utterances per chain / wave, every tile path) against the oracle on a random
sample of utterances, for the configurations the unit tests only run on a handful of files."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ctucopy_amd import Engine, synth
from oracle.oracle import Oracle
from tests.util import C2, C3, C4, C4_NOVAD, C5

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
CFGS = {"C3": C3, "C2_d_a_cms": C2 + ["-fea_delta", "d_a", "-fea_Z_block", "500"], "C2_trap9": C2 + ["-fea_trap", "9"],
        "C2_vad_energy": C2 + "-vad_out_mode vad -vad_cri_mode energy -vad_thr_mode adapt".split(),
        "C2_vad16_exten": C2 + "-nr_mode exten -nr_a 2 -vad burg -vad_out_mode vad -vad_cri_mode cepdist -vad_thr_mode adapt".split(),
        "C4_novad": C4_NOVAD, "fft1024_plp": C3 + ["-w", "40", "-s", "10"], "lp_noinld": C2 + "-fb_inld off -fea_kind lpc -fea_lporder 12 -fea_ncepcoefs 12".split(),
        "C2_dc1": C2 + ["-remove_dc1", "on"], "fwss8": "-fs 8000 -format_in raw -format_out htk -preset mfcc -preem 0.97 -vad burg -nr_mode fwss".split(),
        "exten_raw": "-fs 16000 -format_in raw -format_out raw -preset exten".split(),
        # round 4: exten along chains of utterances on 1024-point frames (more utterances than waves), the VAD's energy criterion there,
        # fwss ahead of an energy column and a delta chain
        "fft1024_exten": C2 + ["-w", "40", "-s", "10", "-nr_mode", "exten"],
        "fft1024_vad": C2 + "-w 40 -s 10 -nr_mode exten -vad_out_mode vad -vad_cri_mode energy -vad_thr_mode adapt".split(),
        "fwss8_E_d_a": "-fs 8000 -format_in raw -format_out htk -preset mfcc -preem 0.97 -vad burg -nr_mode fwss -fea_E on -fea_delta d_a".split()}
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
for name, cfg in CFGS.items():
    if len(sys.argv) > 2 and name not in sys.argv[2:]:
        continue
    eng = Engine(cfg)
    set_id = synth.SET_NOISY if eng.dims.fs == 8000 else synth.SET_SPEECH
    chained = "fwss" in cfg
    n = 600 if chained else N                      # the oracle walks a chained list file by file
    idx = np.arange(n)
    plan = eng.plan(synth.lengths(set_id, idx, chained))
    host = synth.fill_arena(set_id, idx, plan.sample_off, plan.total_samples, mini=chained, threads=16)
    pcm = torch.from_numpy(host).to(dev)
    orc = Oracle(cfg)
    pick = np.arange(n) if chained else rng.choice(n, 8, replace=False)
    if eng.dims.signal_out:
        out = eng.enhance_device(plan, pcm).cpu().numpy()
        worst = 0
        for k in pick:
            u = host[plan.sample_off[k]:plan.sample_off[k] + plan.nsamples[k]]
            ref = orc.enhance(u)
            got = out[plan.sample_off[k]:plan.sample_off[k] + ref.size]
            worst = max(worst, int(np.abs(got.astype(int) - ref.astype(int)).max()))
        print(f"{name}: {n} utterances, worst |LSB| {worst}", flush=True)
        continue
    rows = torch.empty((plan.total_frames, eng.dims.row_floats), dtype=torch.float32, device=dev)
    vad = torch.empty(plan.total_frames, dtype=torch.uint8, device=dev) if eng.dims.has_vad else None
    eng.run_device(plan, pcm, rows, vad=vad)
    torch.cuda.synchronize()
    worst = rown = 0.0
    flips = 0
    for k in pick:
        u = host[plan.sample_off[k]:plan.sample_off[k] + plan.nsamples[k]]
        ref = orc.process(u, want_vad=vad is not None)
        rr = ref[0] if vad is not None else ref
        got = rows[plan.row_off[k]:plan.row_off[k + 1]].cpu().numpy()
        if rr.size:
            worst = max(worst, float((np.abs(got - rr) / np.maximum(np.abs(rr), 1.0)).max()))
            rown = max(rown, float((np.abs(got - rr).max(axis=1) / np.maximum(np.abs(rr).max(axis=1), 1.0)).max()))
        if vad is not None:
            flips += int((vad[plan.row_off[k]:plan.row_off[k + 1]].cpu().numpy() != ref[1]).sum())
    print(f"{name}: {n} utterances, {plan.total_frames} frames, kernel {eng.kernel_name()}: worst {worst:.2e} row-norm {rown:.2e}" +
          (f" vad bytes differing {flips}" if vad is not None else "") + f" finite {bool(torch.isfinite(rows).all().item())}", flush=True)
