"""Utterance sharding across ranks and the cross-rank timing reduction used by bench.py.

Utterances are independent on the hot path, so N ranks = N independent shards and NO collective touches the data;
torch.distributed (RCCL on GPUs, gloo in the CPU tests) only carries the timing barrier, max(time), sum(frames).
The one real exchange step of the tool is per-speaker CMVN (row N2): a speaker's utterances may sit on several ranks,
so the [n_spk, cols+1] partial sums of each statistics pass are all-reduced (allreduce_stats).
"""
import numpy as np


def utterance_lengths(n_utt, seed, lo=48000, hi=240000):
    """Deterministic per-shard utterance lengths in samples (3-15 s at 16 kHz): splitmix64 of the index."""
    i = np.arange(n_utt, dtype=np.uint64) + np.uint64(seed) * np.uint64(1000003)
    with np.errstate(over="ignore"):
        z = i + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (lo + (z % np.uint64(hi - lo + 1))).astype(np.int64)


def rank_shard(rank, utts_per_rank, base_seed=20260101):
    """Weak scaling: every rank owns its own `utts_per_rank` utterances (seeded by rank)."""
    return utterance_lengths(utts_per_rank, seed=base_seed + rank)


def split_list(n_items, world, rank):
    """Strong-scaling split of a fixed list (the CLI's case): contiguous, sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def lpt_shard(frames, world):
    """Longest-processing-time partition of a list over `world` ranks: utterances sorted by frame count,
    each handed to the rank with the least frames so far.  Returns one index array per rank (ascending list order inside
    a rank, so outputs are written in list order); the rank loads differ by at most one utterance's frames."""
    import heapq
    frames = np.asarray(frames, dtype=np.int64)
    order = np.argsort(-frames, kind="stable")
    heap = [(0, r) for r in range(world)]
    parts = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        parts[r].append(int(i))
        heapq.heappush(heap, (load + int(frames[i]), r))
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def reduce_timing(dt_seconds, frames, device=None):
    """(max over ranks of dt, sum over ranks of frames); identity when torch.distributed is not initialised."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(dt_seconds), float(frames)
    t = torch.tensor([float(dt_seconds)], dtype=torch.float64, device=device)
    f = torch.tensor([float(frames)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(t.item()), float(f.item())


def allreduce_stats(acc, device=None):
    """Sum the per-speaker partial statistics of Engine.cmvn_accumulate over ranks, in place (float64, [n_spk, cols+1]).

    One small all-reduce per statistics pass (RCCL when `device` is a GPU, gloo on CPU); identity without a process
    group.  Every rank must hold the same speaker table (the list's order of first appearance)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return acc
    t = torch.from_numpy(acc)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    acc[...] = t.cpu().numpy()
    return acc
