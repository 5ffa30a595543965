#!/usr/bin/env python3
"""Headline benchmark: MFCC-13 frames/s on synthetic 16 kHz / 25 ms / 10 ms streams (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the hot path (ctu_engine_run through the C ABI) over the rank's resident batch:
10 000 synthetic utterances of 3-15 s (~9 M frames, ~2.9 GB int16 PCM) already in HBM.  Utterances are
independent, so ranks shard them with no collective on the data path ("weak" scaling: every rank owns a
full 10k-utterance batch); RCCL is only used for the timing barrier / max-over-ranks.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     algorithmic HBM bytes (2*wshift + 4*D = 372 B/frame) / measured kernel time vs 8 TB/s
  cpu_baseline the CPU oracle (a port of the reference's algorithm, oracle/) timed on this host's cores
               over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = "-fs 16000 -format_in raw -format_out htk -preset mfcc -preem 0.97".split()
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3


def synth_arena(total_samples, seed, device):
    """Speech-like stream generated on the GPU: 4 harmonics of an f0 gliding 90-250 Hz, 4 Hz AM,
    white noise sigma~300 LSB from an integer hash of the sample index (never digitally silent)."""
    import torch
    out = torch.empty(total_samples, dtype=torch.int16, device=device)
    chunk = 1 << 24
    two_pi = 2.0 * np.pi
    for s in range(0, total_samples, chunk):
        n = min(chunk, total_samples - s)
        idx = torch.arange(s, s + n, device=device, dtype=torch.int64)
        t = idx.to(torch.float64) / 16000.0
        # f0(t) = 170 + 80 sin(2 pi 0.3 t)  ->  closed-form phase (no cumsum)
        ph = two_pi * (170.0 * t - (80.0 / (two_pi * 0.3)) * torch.cos(two_pi * 0.3 * t))
        ph = torch.remainder(ph, two_pi).to(torch.float32)
        x = torch.sin(ph) + 0.5 * torch.sin(2 * ph + 1.0) + 0.33 * torch.sin(3 * ph + 2.0) + 0.25 * torch.sin(4 * ph + 0.5)
        am = 0.6 + 0.4 * torch.sin((two_pi * 4.0) * torch.remainder(t, 1.0)).to(torch.float32)
        x = 6000.0 * am * x
        # integer hash -> 4 uniform bytes -> approximately Gaussian noise
        h = (idx + seed * 7919) * 2654435761 % 4294967296
        h = (h ^ (h >> 15)) * 2246822519 % 4294967296
        h = (h ^ (h >> 13)) * 3266489917 % 4294967296
        h = h ^ (h >> 16)
        u = ((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + ((h >> 24) & 255)).to(torch.float32)
        x = x + (u - 510.0) * (300.0 / 147.8)
        out[s:s + n] = torch.clamp(torch.round(x), -32768, 32767).to(torch.int16)
    return out


def cpu_baseline(pcm_host, lens, offs, max_utts):
    """Oracle (port of the reference algorithm, float64) on the host cores over the first max_utts utterances."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import Oracle
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    n = min(max_utts, len(lens))
    oracles = [Oracle(CFG) for _ in range(cores)]

    def work(w):
        o, frames = oracles[w], 0
        for i in range(w, n, cores):
            frames += o.process(pcm_host[offs[i]:offs[i] + lens[i]]).shape[0]
        return frames

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL inside the C call
        frames = sum(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"first {n} utterances ({frames} frames) of the same workload, float64 C oracle, "
                      f"{cores} threads, {dt:.2f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=10000, help="utterances per GPU")
    ap.add_argument("--cpu-utts", type=int, default=4096, help="utterances in the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ctucopy_amd import Engine, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    eng = Engine(CFG, device=local)
    lens = shard.rank_shard(rank, args.utts)
    plan = eng.plan(lens)
    pcm = synth_arena(plan.total_samples, seed=rank, device=dev)
    rows = torch.empty((plan.total_frames, eng.dims.row_floats), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        eng.run_device(plan, pcm, rows, stream=stream)
    sync_all()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run_device(plan, pcm, rows, stream=stream)
    sync_all()
    dt = time.perf_counter() - t0
    # per-launch kernel time from HIP events recorded by the library on this stream (separate short loop,
    # outside the timed region, because reading an event blocks the host)
    for _ in range(min(args.steps, 10)):
        eng.run_device(plan, pcm, rows, stream=stream)
        kernel_ms.append(eng.last_kernel_ms())
    dt, total_frames = shard.reduce_timing(dt, plan.total_frames, device=dev)

    if rank == 0:
        d = eng.dims
        bytes_per_frame = 2 * d.wshift + 4 * d.row_floats
        k_ms = float(np.median(kernel_ms))
        achieved = plan.total_frames * bytes_per_frame / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC passes (FETCH_SIZE doubled per the gfx950 correction, plus
        # WRITE_SIZE), scaled per frame; PMC counters cannot be collected inside this process
        traffic_gb = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            traffic_gb = json.load(open(tf))["hbm_bytes_per_frame"] * plan.total_frames  # bytes per launch
        result = {
            "metric": "frames/sec (16 kHz, 25 ms/10 ms, MFCC-13)",
            "value": total_frames * args.steps / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"S-MFCC: {args.utts} synthetic utterances/GPU, 3-15 s, 16 kHz int16, 512-pt FFT, "
                                   "26 mel -> MFCC-13 (-preset mfcc -preem 0.97), device-resident",
                       "frames_per_gpu": plan.total_frames, "pcm_bytes_per_gpu": plan.total_samples * 2,
                       "parallelism": f"utterance shard x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_gb,
                         "kernel": "frontend_kernel<13, DCTC>", "kernel_ms": k_ms,
                         "bytes_per_frame": bytes_per_frame},
        }
        if not args.no_cpu and world >= 1:
            n = min(args.cpu_utts, plan.n_utt)
            end = int(plan.sample_off[n])
            host = pcm[:end].cpu().numpy()
            result["cpu_baseline"] = cpu_baseline(host, lens, plan.sample_off, n)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
