"""PCIe-inclusive rate of ctu_engine_run_host (H2D + kernel + D2H) for DESIGN.md section 8: pageable and pinned buffers."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctucopy_amd import Engine, synth
from ctucopy_amd.engine import host_alloc
from bench import CFG
eng = Engine(CFG)
idx = np.arange(int(sys.argv[1]) if len(sys.argv) > 1 else 4000)
plan = eng.plan(synth.lengths(synth.SET_SPEECH, idx))
pageable = synth.fill_arena(synth.SET_SPEECH, idx, plan.sample_off, plan.total_samples)
pinned = host_alloc((plan.total_samples,), np.int16)
pinned[:] = pageable
rows_pinned = host_alloc((plan.total_frames, eng.dims.row_floats), np.float32)
for name, arena, rows in (("pageable", pageable, None), ("pinned", pinned, rows_pinned)):
    eng.run_host(plan, arena, rows_out=rows)
    t0 = time.perf_counter()
    for _ in range(3):
        out = eng.run_host(plan, arena, rows_out=rows)
    dt = (time.perf_counter() - t0) / 3
    print(name, "frames", plan.total_frames, "ms %.1f" % (dt * 1e3), "frames/s %.3g" % (plan.total_frames / dt),
          "GB/s in+out %.1f" % ((plan.total_samples * 2 + plan.total_frames * 52) / dt / 1e9), "finite", bool(np.isfinite(out).all()))
