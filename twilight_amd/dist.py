"""Sharding of one guide-tree level over ranks (one process per GPU).

The pairs of a level are independent units (reference alignment-cpu.cpp:46; the reference's GPU host code deals
batches to devices with an atomic counter, hip/alignment-gpu.hip.cpp:239-254), so ranks never exchange DP data:
each rank aligns its shard and only the report scalars (and, for a caller that wants them on one rank, the paths)
travel.  torch.distributed is plumbing here: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import heapq

import numpy as np


def pair_costs(lens: np.ndarray) -> np.ndarray:
    """Cost proxy of a pair: R+Q anti-diagonals times a band that is roughly constant for fixed scoring."""
    lens = np.asarray(lens, dtype=np.int64)
    return lens[:, 0] + lens[:, 1]


def lpt_shards(costs, world: int):
    """Longest-processing-time-first dealing of items to `world` ranks.  Deterministic; returns a list of index arrays
    (each in descending-cost order, which is also the launch order the kernel's work queue wants)."""
    costs = np.asarray(costs)
    order = np.argsort(-costs, kind="stable")
    heap = [(0, r) for r in range(world)]
    heapq.heapify(heap)
    shards = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        shards[r].append(int(i))
        heapq.heappush(heap, (load + int(costs[i]), r))
    return [np.asarray(s, dtype=np.int64) for s in shards]


def take(batch, idx):
    """Sub-batch of a LevelBatch-like object."""
    from .synth import LevelBatch

    idx = np.asarray(idx, dtype=np.int64)
    return LevelBatch(P=batch.P, seq_len=batch.seq_len, freq=batch.freq[idx], gap_open=batch.gap_open[idx],
                      gap_extend=batch.gap_extend[idx], len=batch.len[idx], num=batch.num[idx])


def reduce_report(cells: float, seconds: float, device=None):
    """(sum of cells over ranks, max of seconds over ranks): the whole-job rate is their quotient."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(cells), float(seconds)
    t_c = torch.tensor([float(cells)], dtype=torch.float64, device=device)
    t_s = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t_c, op=dist.ReduceOp.SUM)
    dist.all_reduce(t_s, op=dist.ReduceOp.MAX)
    return float(t_c.item()), float(t_s.item())


def align_level_sharded(align_fn, batch, rank: int, world: int, gather_to_rank0: bool = True):
    """Align one level on `world` ranks.  `align_fn(sub_batch) -> (aln, aln_len, err)` is the per-rank aligner
    (twl.align_batch on a GPU rank).  Returns this rank's (indices, aln, aln_len, err); with gather_to_rank0, rank 0
    additionally gets the full-level arrays in original pair order (other ranks get None)."""
    import torch.distributed as dist

    shards = lpt_shards(pair_costs(batch.len), world)
    mine = shards[rank]
    aln, n, err = align_fn(take(batch, mine)) if len(mine) else (np.zeros((0, 2 * batch.seq_len), np.int8), np.zeros(0, np.int32), np.zeros(0, np.int16))
    full = None
    if gather_to_rank0 and world > 1:
        parts = [None] * world if rank == 0 else None
        dist.gather_object((mine, aln, n, err), parts, dst=0)
        if rank == 0:
            N = batch.n_pairs
            f_aln = np.zeros((N, 2 * batch.seq_len), np.int8)
            f_n = np.zeros(N, np.int32)
            f_err = np.zeros(N, np.int16)
            for idx, a, ln, e in parts:
                f_aln[idx] = a
                f_n[idx] = ln
                f_err[idx] = e
            full = (f_aln, f_n, f_err)
    elif gather_to_rank0:
        f_aln = np.zeros((batch.n_pairs, 2 * batch.seq_len), np.int8)
        f_n = np.zeros(batch.n_pairs, np.int32)
        f_err = np.zeros(batch.n_pairs, np.int16)
        f_aln[mine] = aln
        f_n[mine] = n
        f_err[mine] = err
        full = (f_aln, f_n, f_err)
    return (mine, aln, n, err), full
