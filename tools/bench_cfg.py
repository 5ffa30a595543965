#!/usr/bin/env python3
"""Times BASELINE.json configs other than the headline one (diagnostic; bench.py stays the contract)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctucopy_amd import Engine, shard, synth
from tests.util import C2, C3, C4, C4_NOVAD, C5

CFGS = {"C2": C2, "C3": C3, "C4": C4, "C4_novad": C4_NOVAD, "C5": C5, "C2_d_a": C2 + ["-fea_delta", "d_a"],
        "C2_trap9": C2 + ["-fea_trap", "9"], "C2_cms_exp": C2 + ["-fea_Z_exp", "2000"], "C2_cms_block": C2 + ["-fea_Z_block", "2000"],
        "C2_d_a_cms": C2 + ["-fea_delta", "d_a", "-fea_Z_block", "2000"],
        "exten_raw": "-fs 16000 -format_in raw -format_out raw -preset exten".split(),
        "C2_vad16": C2 + "-vad burg -vad_out_mode vad -vad_cri_mode cepdist -vad_cepdist_mode lpc -vad_thr_mode adapt".split(),
        "C2_fwss16": C2 + "-vad burg -nr_mode fwss".split(),
        "C4_nc2": C4 + ["-vad_lpc_coefs", "2"], "C4_nc8": C4 + ["-vad_lpc_coefs", "8"],
        "fft1024": C2 + ["-w", "40", "-s", "10"], "fft1024_exten": C2 + ["-w", "40", "-s", "10", "-nr_mode", "exten"],
        "C2_fwss16_E_d_a": C2 + "-vad burg -nr_mode fwss -fea_E on -fea_delta d_a".split(), "fft128": "-fs 8000 -format_in raw -format_out htk -preset mfcc -preem 0.97 -w 16 -s 8".split(),
        "lp_noinld": C2 + "-fb_inld off -fea_kind lpc -fea_lporder 12 -fea_ncepcoefs 12".split(), "C2_dc1": C2 + ["-remove_dc1", "on"],
        "C2_d_a_cmvn": C2 + ["-fea_delta", "d_a", "-stat_cmvn", "x.stat", "-apply_cmvn", "x.stat"]}
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="C3")
ap.add_argument("--utts", type=int, default=2000)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
if a.cfg == "C4_10k":  # configs[3] at the size of the benchmark's list: 10 000 utterances of S-NOISY
    a.cfg, a.utts = "C4", 10000
cfg = CFGS[a.cfg]
eng = Engine(cfg)
fs = eng.dims.fs
set_id = synth.SET_NOISY if fs == 8000 else synth.SET_SPEECH   # S-NOISY for the 8 kHz configurations, S-MFCC otherwise
idx = np.arange(a.utts)
plan = eng.plan(synth.lengths(set_id, idx))
dev = torch.device("cuda", 0)
pcm = torch.from_numpy(synth.fill_arena(set_id, idx, plan.sample_off, plan.total_samples)).to(dev)
if eng.dims.signal_out:
    out = torch.zeros(plan.total_samples, dtype=torch.int16, device=dev)
    for _ in range(2):
        eng.enhance_device(plan, pcm, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.enhance_device(plan, pcm, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"cfg": a.cfg, "frames": plan.total_frames, "ms_per_step": dt * 1e3, "frames_per_s": plan.total_frames / dt,
                      "front_kernel_ms": eng.last_kernel_ms(), "samples_per_s": plan.total_samples / dt}))
    sys.exit(0)
rows = torch.empty((plan.total_frames, eng.dims.row_floats), dtype=torch.float32, device=dev)
vad = torch.empty(plan.total_frames, dtype=torch.uint8, device=dev) if eng.dims.has_vad else None
for _ in range(2):
    eng.run_device(plan, pcm, rows, vad=vad)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    eng.run_device(plan, pcm, rows, vad=vad)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
bpf = 2 * eng.dims.wshift + 4 * eng.dims.row_floats + (1 if eng.dims.has_vad else 0)
if "-stat_cmvn" in cfg:  # the three CMVN passes over the resident rows (64 synthetic speakers)
    spk = (np.arange(plan.n_utt) % 64).astype(np.int32)
    def passes():
        acc = eng.cmvn_accumulate(plan, rows, spk, 64)
        mean = acc[:, :-1] / acc[:, -1:]
        acc2 = eng.cmvn_accumulate(plan, rows, spk, 64, mean=mean)
        var = acc2[:, :-1] / (acc2[:, -1:] - 1)
        eng.cmvn_apply(plan, rows, spk, 64, mean, var * 0 + 1)   # unit "variance": repeated runs stay finite
        torch.cuda.synchronize()
    passes()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        passes()
    print(json.dumps({"cfg": a.cfg, "cmvn_three_passes_ms": (time.perf_counter() - t0) / a.steps * 1e3, "frames": plan.total_frames}))
print(json.dumps({"cfg": a.cfg, "frames": plan.total_frames, "ms_per_step": dt * 1e3, "frames_per_s": plan.total_frames / dt,
                  "front_kernel_ms": eng.last_kernel_ms(), "bytes_per_frame": bpf,
                  "hbm_frac": plan.total_frames * bpf / dt / 8e12}))
