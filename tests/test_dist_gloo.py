"""N>1 path on CPU: two gloo ranks shard utterances independently and only reduce timing/frames."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ctucopy_amd import shard


def _worker(rank, world, port, q):
    """What bench.py does per rank, minus the GPU: ONE seeded list, identical on every rank, partitioned by
    longest-processing-time; the rank keeps its own part; only the timing and the frame count are reduced."""
    from ctucopy_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_list = 150 * world
    lens = synth.lengths(synth.SET_SPEECH, np.arange(n_list))
    frames_all = (lens - 240) // 160
    mine = shard.lpt_shard(frames_all, world)[rank]
    frames = int(frames_all[mine].sum())
    dt = 0.5 + 0.25 * rank                      # rank 1 is the slow one
    dist.barrier()
    tmax, fsum = shard.reduce_timing(dt, frames)
    q.put((rank, mine.tolist(), frames, tmax, fsum, int(frames_all.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_one_list_and_reduce_timing():
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
    (r0, m0, f0, t0, s0, tot0), (r1, m1, f1, t1, s1, tot1) = res
    assert sorted(m0 + m1) == list(range(300)) and not set(m0) & set(m1)   # the list is covered once, ranks own different utterances
    assert t0 == t1 == 0.75                      # max over ranks
    assert s0 == s1 == f0 + f1 == tot0 == tot1   # whole-job frames
    assert abs(f0 - f1) <= 1500                  # LPT: the loads differ by at most one utterance (<= 15 s = 1498 frames)


def test_split_list_covers_everything_once():
    for n in (0, 1, 7, 8, 9, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard.split_list(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_lpt_shard_balances_and_covers_the_list():
    rng = np.random.default_rng(3)
    frames = rng.integers(300, 1500, size=5000)
    for world in (1, 2, 3, 8):
        parts = shard.lpt_shard(frames, world)
        allidx = np.concatenate(parts)
        assert len(parts) == world and np.array_equal(np.sort(allidx), np.arange(frames.size))
        assert all(np.all(np.diff(p) > 0) for p in parts)                  # list order inside a rank
        loads = np.array([frames[p].sum() for p in parts])
        assert loads.max() - loads.min() <= frames.max()                   # within one utterance
    assert [p.tolist() for p in shard.lpt_shard([5, 1, 4], 2)] == [[0], [1, 2]]


def test_lengths_are_deterministic_and_in_range():
    a, b = shard.utterance_lengths(1000, 5), shard.utterance_lengths(1000, 5)
    assert np.array_equal(a, b) and a.min() >= 48000 and a.max() <= 240000
    assert not np.array_equal(a, shard.utterance_lengths(1000, 6))
    assert torch.is_tensor(torch.zeros(1))


# ---- per-speaker CMVN: the one exchange step (row N2).  Each rank holds half of the utterances; speakers span ranks.
def _cmvn_worker(rank, world, port, q):
    from oracle.oracle import cmvn_slot_columns, cmvn_stats
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    rows = [rng.standard_normal((50 + 10 * i, 13)).astype(np.float32) * (1 + i) + i for i in range(8)]
    spk = np.array([0, 1, 2, 0, 1, 2, 0, 1], dtype=np.int32)
    cols = cmvn_slot_columns(12, 1)
    lo, hi = shard.split_list(len(rows), world, rank)
    # what Engine.cmvn_accumulate returns for this rank's shard: [n_spk, cols+1] sums and counts
    def partial(mean):
        acc = np.zeros((3, 14))
        for r, s_ in zip(rows[lo:hi], spk[lo:hi]):
            x = r[:, cols].astype(np.float64)
            acc[s_, :13] += x.sum(0) if mean is None else ((x - mean[s_]) ** 2).sum(0)
            acc[s_, 13] += r.shape[0]
        return acc
    a = shard.allreduce_stats(partial(None))
    mean = a[:, :13] / a[:, 13:]
    b = shard.allreduce_stats(partial(mean))
    var = b[:, :13] / (b[:, 13:] - 1)
    ref_mean, ref_var, ref_count = cmvn_stats(rows, spk, 3, cols)
    q.put((rank, float(np.abs(mean - ref_mean).max()), float(np.abs(var / ref_var - 1).max()), a[:, 13].tolist(), ref_count.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_cmvn_statistics_match_the_single_process_ones():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_cmvn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, dm, dv, counts, ref_counts in res:
        assert dm < 1e-12 and dv < 1e-12 and counts == ref_counts
