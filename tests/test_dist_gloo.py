"""N>1 path on CPU: two gloo ranks shard utterances independently and only reduce timing/frames."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ctucopy_amd import shard


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = shard.rank_shard(rank, 100)
    frames = int(((lens - 240) // 160).sum())
    dt = 0.5 + 0.25 * rank                      # rank 1 is the slow one
    dist.barrier()
    tmax, fsum = shard.reduce_timing(dt, frames)
    q.put((rank, lens[:3].tolist(), frames, tmax, fsum))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_weak_scaling_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, l0, f0, t0, s0), (r1, l1, f1, t1, s1) = res
    assert l0 != l1                              # ranks own different utterances
    assert t0 == t1 == 0.75                      # max over ranks
    assert s0 == s1 == f0 + f1                   # whole-job frames
    assert f0 == int(((shard.rank_shard(0, 100) - 240) // 160).sum())


def test_split_list_covers_everything_once():
    for n in (0, 1, 7, 8, 9, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard.split_list(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_lengths_are_deterministic_and_in_range():
    a, b = shard.utterance_lengths(1000, 5), shard.utterance_lengths(1000, 5)
    assert np.array_equal(a, b) and a.min() >= 48000 and a.max() <= 240000
    assert not np.array_equal(a, shard.utterance_lengths(1000, 6))
    assert torch.is_tensor(torch.zeros(1))
