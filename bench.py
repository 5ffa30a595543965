#!/usr/bin/env python3
"""Headline benchmark: MFCC-13 frames/s on synthetic 16 kHz / 25 ms / 10 ms streams (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

Workload: ONE seeded list of N x 10 000 utterances of the S-MFCC set (ctucopy_amd/synth.py: 3-15 s, 16 kHz int16, seed
20260101 + index; ~9 M frames and ~2.9 GB of PCM per GPU), partitioned over the N ranks by longest-processing-time
(ctucopy_amd/shard.py) - utterances are independent, so there is no collective on the data path; RCCL only
carries the timing barrier and the max-over-ranks.  Per-GPU work stays fixed as N grows ("weak" scaling); the per-rank
frame counts in the JSON line show the balance the partition reached.

A step = one pass of the hot path (ctu_engine_run through the C ABI) over the rank's resident batch.  After the timed
region a sample of the rows it produced is checked against the CPU oracle (and every row for finiteness).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     algorithmic HBM bytes (2*wshift + 4*D = 372 B/frame) / measured kernel time vs 8 TB/s, and beside it the
               fractions of the VALU issue rate and of the LDS rate the same launch used (instruction / LDS-cycle counts
               per frame from the committed PMC passes, profiles/traffic.json, only if they are for this kernel)
  cpu_baseline the CPU oracle (a float64 C port of the reference's algorithm, oracle/) on this host: one process per
               core over an LPT split of a bounded sample of the same list, and a single core beside it.
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
KERNEL = "frontend_kernel<13, DCTC, MODE 0, plain, MD>"
# issue / LDS ceilings of the chip (MI355X_MICROARCH.md): 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles
# and SIMD; one LDS-array cycle per cycle and CU; 2.4 GHz
VALU_PEAK_WAVE_INSTR_S = 256 * 4 * 2.4e9 / 2
LDS_PEAK_CYCLES_S = 256 * 2.4e9


def _cpu_worker(args):
    """One process of the CPU baseline: the oracle over this worker's utterances until the deadline."""
    indices, deadline = args
    from ctucopy_amd import synth
    from oracle.oracle import Oracle
    orc = Oracle(CFG)
    frames = 0
    t0 = time.perf_counter()
    for i in indices:
        frames += orc.process(synth.utterance_c(synth.SET_SPEECH, int(i))).shape[0]
        if time.perf_counter() - t0 > deadline:
            break
    return frames, time.perf_counter() - t0


def cpu_baseline(n_sample, seconds):
    """the reference path is single-threaded, so the host runs one process per core on an LPT split of
    the list; reported as all-core and 1-core frames/s with the core count and CPU model."""
    import multiprocessing as mp
    from ctucopy_amd import shard, synth
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    idx = np.arange(n_sample)
    lens = synth.lengths(synth.SET_SPEECH, idx)
    parts = shard.lpt_shard((lens - 240) // 160, cores)
    f1, t1 = _cpu_worker((idx[:64], min(seconds, 6.0)))          # one core
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(p, seconds) for p in parts])
    wall = time.perf_counter() - t0
    frames = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": frames / busy, "unit": "frames/s", "cores": cores, "kind": "port",
            "one_core": f1 / t1, "cpu_model": model,
            "sample": f"S-MFCC utterances 0..{n_sample - 1} LPT-split over {cores} processes (one per core), each running the "
                      f"float64 C oracle (-O2, port of the reference's double arithmetic) for <= {seconds:.0f} s: {frames} frames, "
                      f"slowest worker {busy:.2f} s (pool wall {wall:.2f} s incl. process start); one_core = a single process, {f1} frames"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=10000, help="utterances per GPU (the list holds gpus x utts)")
    ap.add_argument("--cpu-utts", type=int, default=4096, help="utterances in the CPU baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")

    import torch  # before the engine library: one HIP runtime per process (torch's), whichever is loaded first wins
    import torch.distributed as dist
    from ctucopy_amd import Engine, shard, synth

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(args.cpu_utts, args.cpu_seconds)  # before this process touches the GPU (it starts processes)

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    eng = Engine(CFG, device=local)
    # the whole list, identical on every rank; this rank's share by LPT
    n_list = args.utts * world
    all_idx = np.arange(n_list)
    all_len = synth.lengths(synth.SET_SPEECH, all_idx)
    all_frames = (all_len - (eng.dims.window - eng.dims.wshift)) // eng.dims.wshift
    mine = shard.lpt_shard(all_frames, world)[rank]
    plan = eng.plan(all_len[mine])
    host = synth.fill_arena(synth.SET_SPEECH, mine, plan.sample_off, plan.total_samples)
    pcm = torch.from_numpy(host).to(dev)
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
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run_device(plan, pcm, rows, stream=stream)
    sync_all()
    dt = time.perf_counter() - t0
    # per-launch kernel time from HIP events recorded by the library on this stream (separate short loop,
    # outside the timed region, because reading an event blocks the host)
    kernel_ms = []
    for _ in range(min(args.steps, 10)):
        eng.run_device(plan, pcm, rows, stream=stream)
        kernel_ms.append(eng.last_kernel_ms())
    dt, total_frames = shard.reduce_timing(dt, plan.total_frames, device=dev)
    per_rank = [plan.total_frames]
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = plan.total_frames
        dist.all_reduce(t)
        per_rank = [int(x) for x in t.tolist()]

    # ---- what was timed is checked: every row finite, a sample of utterances against the CPU oracle
    finite = bool(torch.isfinite(rows).all().item())
    checked, worst = 0, 0.0
    if rank == 0:
        from oracle.oracle import Oracle
        orc = Oracle(CFG)
        pick = np.random.default_rng(1).choice(plan.n_utt, size=min(4, plan.n_utt), replace=False)
        for k in pick:
            u = host[plan.sample_off[k]:plan.sample_off[k] + plan.nsamples[k]]
            ref = orc.process(u)
            got = rows[plan.row_off[k]:plan.row_off[k + 1]].cpu().numpy()
            worst = max(worst, float((np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)).max()))
            checked += ref.shape[0]
        if not finite or worst > 1e-4:
            raise SystemExit(f"bench validation failed: finite={finite}, worst error {worst:.3e} over {checked} frames")

    if rank == 0:
        d = eng.dims
        bytes_per_frame = 2 * d.wshift + 4 * d.row_floats
        k_ms = float(np.median(kernel_ms))
        fps_kernel = plan.total_frames / (k_ms * 1e-3)
        achieved = fps_kernel * bytes_per_frame / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "kernel": KERNEL, "kernel_ms": k_ms, "bytes_per_frame": bytes_per_frame}
        # counters cannot be collected inside this process: per-frame figures of the committed PMC passes, used only when
        # they were taken on this kernel (provenance travels with the numbers)
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            pm = json.load(open(tf))
            if pm.get("kernel") == KERNEL:
                roof["traffic"] = pm["hbm_bytes_per_frame"] * plan.total_frames
                roof["traffic_source"] = pm.get("source")
                roof["valu_frac"] = fps_kernel * pm["valu_instr_per_frame"] / VALU_PEAK_WAVE_INSTR_S
                roof["lds_frac"] = fps_kernel * pm["lds_cycles_per_frame"] / LDS_PEAK_CYCLES_S
                roof["limiter"] = pm.get("limiter")
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
            "config": {"workload": f"S-MFCC: one seeded list of {n_list} synthetic utterances ({args.utts}/GPU), 3-15 s, 16 kHz int16, "
                                   "512-pt FFT, 26 mel -> MFCC-13 (-preset mfcc -preem 0.97), device-resident, LPT-sharded",
                       "frames_per_rank": per_rank, "pcm_bytes_per_gpu": plan.total_samples * 2,
                       "parallelism": f"utterance shard x{world} (LPT), no collective"},
            "roofline": roof,
            "validated": {"rows_finite": finite, "oracle_frames_checked": checked, "worst_rel_err": worst, "tol": 1e-4},
        }
        if cpu is not None:
            result["cpu_baseline"] = cpu
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
