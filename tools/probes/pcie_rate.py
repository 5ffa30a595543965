"""PCIe-inclusive rate of ctu_engine_run_host (H2D + kernel + D2H, pageable host memory) for DESIGN.md section 8."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from ctucopy_amd import Engine, shard
from bench import CFG
eng = Engine(CFG)
lens = shard.rank_shard(0, 2000)
plan = eng.plan(lens)
arena = (np.random.default_rng(0).integers(-3000, 3000, plan.total_samples)).astype(np.int16)
eng.run_host(plan, arena)
t0 = time.perf_counter()
for _ in range(3):
    eng.run_host(plan, arena)
dt = (time.perf_counter() - t0) / 3
print("frames", plan.total_frames, "ms", dt * 1e3, "frames/s %.3g" % (plan.total_frames / dt), "GB/s in+out %.2f" % ((plan.total_samples * 2 + plan.total_frames * 52) / dt / 1e9))
