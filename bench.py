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
               usable core, each over its own pre-generated utterances of the same list, all started together, and a
               single core beside it (N=1 only; the processes are started before this one loads torch or the HIP library)
  e2e          the same batch from page-locked host memory through ctu_engine_run_host: upload + kernels + download
               (N=1 only; never `value`)
  configs      BASELINE.json configs[2..4] (PLP-12, exten + Burg-cepstral VAD at 8 kHz, TRAP-DCT) on their synthetic sets:
               ms per pass, frames/s, fraction of the HBM roofline at their own bytes per frame, each checked against the
               oracle on a sample (N=1 only; outside the timed region)
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
# issue / LDS ceilings of the chip (MI355X_MICROARCH.md): 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles
# and SIMD; one LDS-array cycle per cycle and CU; 2.4 GHz
VALU_PEAK_WAVE_INSTR_S = 256 * 4 * 2.4e9 / 2
LDS_PEAK_CYCLES_S = 256 * 2.4e9
SET_SPEECH, SET_NOISY = 0, 1


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_loop(orc, utts, seconds):
    """The oracle over `utts`, round and round, for `seconds`: (frames, wall seconds, CPU seconds of this process)."""
    t0, c0 = time.perf_counter(), time.process_time()
    frames = k = 0
    while True:
        frames += orc.process(utts[k % len(utts)]).shape[0]
        k += 1
        if time.perf_counter() - t0 >= seconds:
            break
    return frames, time.perf_counter() - t0, time.process_time() - c0


def _cpu_worker(indices, seconds, barrier, q):
    """One process of the CPU baseline.  Everything that is not the reference's per-frame work - imports, the oracle's
    tables, generating this worker's utterances - happens before the start barrier; the clock covers the oracle only."""
    try:
        from ctucopy_amd import synth
        from oracle.oracle import Oracle
        orc = Oracle(CFG)
        utts = [synth.utterance_c(SET_SPEECH, int(i)) for i in indices]
        orc.process(utts[0][:8000])
        try:
            barrier.wait(timeout=180)
        except Exception:
            pass  # a worker that died before the barrier must not hang the others: they run unsynchronised instead
        q.put(_cpu_loop(orc, utts, seconds))
    except Exception as e:  # noqa: BLE001
        q.put(("error", repr(e)))


def _usable_cores():
    """Cores this process may use: the affinity mask, cut down by a cgroup CPU quota if there is one."""
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    eff = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return aff, quota, eff


def cpu_baseline(seconds, per_worker, max_procs):
    """the reference path is single-threaded, so the host runs one process per core on its own share of
    the list; reported as all-core and 1-core frames/s with the core count and CPU model.  Runs before this process
    imports torch or loads the HIP-linked engine library (the children are spawned, and none of them loads it either)."""
    import multiprocessing as mp
    from ctucopy_amd import synth
    from oracle.oracle import Oracle
    aff, quota, cores = _usable_cores()
    cores = max(1, min(cores, max_procs))
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # one core, alone on the machine
    orc = Oracle(CFG)
    mine = [synth.utterance_c(SET_SPEECH, i) for i in range(per_worker)]
    orc.process(mine[0][:8000])
    f1, t1, c1 = _cpu_loop(orc, mine, min(seconds, 4.0))
    del orc, mine
    one_core = f1 / t1
    # all cores: worker w owns utterances w, w + cores, ... of the list's first cores * per_worker entries
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(cores), ctx.Queue()
    t0 = time.perf_counter()
    procs = [ctx.Process(target=_cpu_worker, args=(list(range(w, cores * per_worker, cores)), seconds, barrier, q), daemon=True)
             for w in range(cores)]
    for p in procs:
        p.start()
    res = []
    for _ in procs:
        try:
            res.append(q.get(timeout=seconds + 300))
        except Exception:
            break
    for p in procs:
        p.join(timeout=30)
    wall = time.perf_counter() - t0
    errors = [r for r in res if r and r[0] == "error"]
    ok = [r for r in res if r and r[0] != "error"]
    frames = sum(r[0] for r in ok)
    busy = max((r[1] for r in ok), default=float("nan"))
    cpu_s = sum(r[2] for r in ok)
    value = frames / busy if ok else float("nan")
    util = cpu_s / (busy * len(ok)) if ok else float("nan")
    out = {"value": value, "unit": "frames/s", "cores": len(ok), "kind": "port", "one_core": one_core,
           "per_core": value / max(len(ok), 1), "cpu_seconds_per_wall_second": cpu_s / busy if ok else float("nan"),
           "cpu_model": model, "affinity_cores": aff, "cgroup_quota_cores": quota,
           "sample": f"S-MFCC utterances 0..{cores * per_worker - 1}, {per_worker} per process, {len(ok)} processes (one per usable core) started "
                     f"together after generating their PCM and building the oracle's tables; each runs the float64 C oracle (-O2, a port of "
                     f"the reference's double arithmetic) over its utterances repeatedly for {seconds:.0f} s: {frames} frames, slowest "
                     f"process {busy:.2f} s, {util:.2f} CPU-seconds per process-second (launch to last result {wall:.1f} s); one_core = a "
                     f"single process alone on the machine, {f1} frames in {t1:.2f} s"}
    if errors:
        out["worker_errors"] = errors[:3]
    if ok and value < 0.3 * len(ok) * one_core:
        out["note"] = (f"all-core rate is {value / (len(ok) * one_core):.2f} of processes x one_core: "
                       + ("the processes got less CPU time than wall time (quota or oversubscription)" if util < 0.8 else
                          "the processes were busy the whole time: shared caches / memory bandwidth / SMT siblings / clock"))
    return out


# ------------------------------------------------------------------------------------------------ GPU legs
def _validate(rows, plan, host, orc, n_pick, seed, vad=None):
    """A sample of utterances against the CPU oracle: (frames checked, worst |a-b| / max(|b|, 1), VAD bytes that differ)."""
    pick = np.random.default_rng(seed).choice(plan.n_utt, size=min(n_pick, plan.n_utt), replace=False)
    checked, worst, flips = 0, 0.0, 0
    for k in pick:
        u = host[plan.sample_off[k]:plan.sample_off[k] + plan.nsamples[k]]
        ref = orc.process(u, want_vad=vad is not None)
        ref_rows = ref[0] if vad is not None else ref
        got = rows[plan.row_off[k]:plan.row_off[k + 1]].cpu().numpy()
        worst = max(worst, float((np.abs(got - ref_rows) / np.maximum(np.abs(ref_rows), 1.0)).max()))
        if vad is not None:
            flips += int((vad[plan.row_off[k]:plan.row_off[k + 1]].cpu().numpy() != ref[1]).sum())
        checked += ref_rows.shape[0]
    return checked, worst, flips


def other_configs(torch, dev, n_utt, steps, pcm16, host16, idx16):
    """BASELINE.json configs[2..4] on their synthetic sets, one pass each timed with events on the launch stream."""
    from ctucopy_amd import Engine, synth
    from oracle.oracle import Oracle
    from tests.util import C3, C4, C5
    out = {}
    for name, cfg, set_id, tol in (("C3_plp12", C3, SET_SPEECH, 1e-4), ("C4_exten_burgvad_8k", C4, SET_NOISY, 1e-4),
                                   ("C5_trapdct_23x16", C5, SET_SPEECH, 1e-4)):
        try:
            eng = Engine(cfg, device=dev.index)
            if set_id == SET_SPEECH:
                idx, plan, pcm, host = idx16, eng.plan(synth.lengths(set_id, idx16)), pcm16, host16
                assert plan.total_samples == pcm16.numel()
            else:
                idx = np.arange(n_utt)
                plan = eng.plan(synth.lengths(set_id, idx))
                host = synth.fill_arena(set_id, idx, plan.sample_off, plan.total_samples, threads=max(1, min(32, _usable_cores()[2])))
                pcm = torch.from_numpy(host).to(dev)
            d = eng.dims
            rows = torch.empty((plan.total_frames, d.row_floats), dtype=torch.float32, device=dev)
            vad = torch.empty(plan.total_frames, dtype=torch.uint8, device=dev) if d.has_vad else None
            for _ in range(2):
                eng.run_device(plan, pcm, rows, vad=vad)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(steps):
                eng.run_device(plan, pcm, rows, vad=vad)
            ev1.record()
            torch.cuda.synchronize(dev)
            ms = ev0.elapsed_time(ev1) / steps
            bpf = 2 * d.wshift + 4 * d.row_floats + (1 if d.has_vad else 0)
            fps = plan.total_frames / (ms * 1e-3)
            checked, worst, flips = _validate(rows, plan, host, Oracle(cfg), 3, 2, vad=vad)
            rec = {"workload": f"{plan.n_utt} utterances of {'S-NOISY (8 kHz)' if set_id == SET_NOISY else 'S-MFCC (16 kHz)'}: " + " ".join(cfg),
                   "frames": plan.total_frames, "ms_per_pass": ms, "frames_per_s": fps, "bytes_per_frame": bpf,
                   "hbm_frac": fps * bpf / (HBM_PEAK_GBS * 1e9), "front_kernel_ms": eng.last_kernel_ms(),
                   "validated": {"oracle_frames_checked": checked, "worst_rel_err": worst, "tol": tol,
                                 "rows_finite": bool(torch.isfinite(rows).all().item())}}
            if vad is not None:
                rec["validated"]["vad_bytes_differing"] = flips
            if worst > tol or flips or not rec["validated"]["rows_finite"]:
                rec["error"] = "validation failed"
            out[name] = rec
            del rows, vad, eng, plan
            if set_id != SET_SPEECH:
                del pcm, host
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001  - the headline line must still be printed
            out[name] = {"error": repr(e)}
    return out


def end_to_end(eng, plan, host, runs=3):
    """Page-locked host arena -> ctu_engine_run_host -> page-locked rows: upload, kernels and download."""
    from ctucopy_amd.engine import host_alloc
    pinned = host_alloc((plan.total_samples,), np.int16)
    pinned[:] = host
    rows = host_alloc((plan.total_frames, eng.dims.row_floats), np.float32)
    eng.run_host(plan, pinned, rows_out=rows)
    t0 = time.perf_counter()
    for _ in range(runs):
        eng.run_host(plan, pinned, rows_out=rows)
    dt = (time.perf_counter() - t0) / runs
    nbytes = plan.total_samples * 2 + plan.total_frames * eng.dims.row_floats * 4
    return {"value": plan.total_frames / dt, "unit": "frames/s", "ms_per_pass": dt * 1e3, "link_GBs": nbytes / dt / 1e9,
            "rows_finite": bool(np.isfinite(rows).all()),
            "what": "ctu_engine_run_host from page-locked buffers: H2D of the int16 arena, the kernels, D2H of the float32 rows, "
                    "utterance ranges on two streams (file decode and the HTK writer are host work outside the C ABI)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=10000, help="utterances per GPU (the list holds gpus x utts)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--cpu-utts", type=int, default=8, help="utterances each CPU-baseline process owns")
    ap.add_argument("--cpu-procs", type=int, default=512, help="upper bound on CPU-baseline processes")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the e2e and configs legs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")

    # The CPU leg first: its processes are spawned while this one has neither torch nor the HIP runtime in it
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        assert "torch" not in sys.modules
        cpu = cpu_baseline(args.cpu_seconds, args.cpu_utts, args.cpu_procs)

    import torch  # before the engine library: one HIP runtime per process (torch's), whichever is loaded first wins
    import torch.distributed as dist
    from ctucopy_amd import Engine, shard, synth

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
    # generator threads: this rank's share of the usable cores (eight ranks of a node each starting one thread per hardware
    # thread would be thousands of tasks at once)
    gen_threads = max(1, min(32, _usable_cores()[2] // max(world, 1)))
    host = synth.fill_arena(synth.SET_SPEECH, mine, plan.sample_off, plan.total_samples, threads=gen_threads)
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
        checked, worst, _ = _validate(rows, plan, host, Oracle(CFG), 24, 1)
        if not finite or worst > 1e-4:
            raise SystemExit(f"bench validation failed: finite={finite}, worst error {worst:.3e} over {checked} frames")

    if rank == 0:
        d = eng.dims
        kname = eng.kernel_name()
        bytes_per_frame = 2 * d.wshift + 4 * d.row_floats
        k_ms = float(np.median(kernel_ms))
        fps_kernel = plan.total_frames / (k_ms * 1e-3)
        achieved = fps_kernel * bytes_per_frame / 1e9
        # `achieved` / `peak` / `frac` are the contract's numbers: algorithmic bytes against the HBM roofline (`roofline_of`).  `bound`
        # says what limits the kernel: at ~37 flop/B this chain sits above the fp32 ridge - VALU issue and the LDS array, never HBM
        # (`limiter` and the two fractions below come with the counters)
        roof = {"bound": "valu+lds", "roofline_of": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": kname, "kernel_ms": k_ms, "bytes_per_frame": bytes_per_frame}
        # counters cannot be collected inside this process: per-frame figures of the committed PMC passes, used only when
        # they were taken on this kernel (provenance travels with the numbers)
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            pm = json.load(open(tf))
            if pm.get("kernel") == kname:
                roof["traffic"] = pm["hbm_bytes_per_frame"] * plan.total_frames
                roof["traffic_source"] = pm.get("source")
                # issue slots: a packed v_pk_*_f32 counts as the 1.9 plain instructions it costs inside this kernel, the conversion /
                # select / DPP forms as 1.65 (profiles/r04_frontend_cost_model.txt); nominal peaks at 2.4 GHz
                slots = pm.get("valu_issue_slots_per_frame", pm["valu_instr_per_frame"])
                roof["valu_frac"] = fps_kernel * slots / VALU_PEAK_WAVE_INSTR_S
                roof["lds_frac"] = fps_kernel * pm["lds_cycles_per_frame"] / LDS_PEAK_CYCLES_S
                # The ceilings at the clock the chip holds under this kernel (GRBM_GUI_ACTIVE / 8 / kernel time of the PMC passes): a
                # 512-point fp32 transform per frame on the vector ALUs cannot come near the HBM roofline - 0.70 of it would need
                # <= 55 issue slots per frame, the radix-16 butterflies alone are 84 - so the fraction of min(HBM, VALU) stands beside `frac`
                clk = pm.get("clock_ghz")
                if clk:
                    valu_ceiling = 256 * 4 * clk * 1e9 / 2.0 / slots          # one wave-instruction per 2 cycles and SIMD
                    hbm_ceiling = HBM_PEAK_GBS * 1e9 / bytes_per_frame
                    roof["clock_ghz"] = clk
                    roof["valu_ceiling_frames_s"] = valu_ceiling
                    roof["lds_ceiling_frames_s"] = 256 * clk * 1e9 / pm["lds_cycles_per_frame"]
                    roof["hbm_ceiling_frames_s"] = hbm_ceiling
                    roof["frac_of_min_hbm_valu"] = fps_kernel / min(hbm_ceiling, valu_ceiling)
                if "mfma_busy_cycles_per_frame" in pm:  # matrix-pipe cycles summed over SIMDs: 1024 pipes at 2.4 GHz
                    roof["mfma_frac"] = fps_kernel * pm["mfma_busy_cycles_per_frame"] / (256 * 4 * 2.4e9)
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
        if world == 1 and not args.no_extra:
            try:
                result["e2e"] = end_to_end(eng, plan, host)
            except Exception as e:  # noqa: BLE001
                result["e2e"] = {"error": repr(e)}
            del rows
            result["configs"] = other_configs(torch, dev, args.utts, max(3, min(args.steps, 10)), pcm, host, mine)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
