"""Several processes (one per GPU) aligning one family together.

The pairs of a level are independent units (reference alignment-cpu.cpp:46; the reference's GPU host code deals
batches to devices with an atomic counter, hip/alignment-gpu.hip.cpp:239-254).  In a sharded run (include/twl_msa.h,
twl_msa_shard) every process holds a replica of the sequences, runs the same host flow, aligns the pairs dealt to its
rank, and once per level the paths are all-gathered: `make_exchange` below is that all-gather over torch.distributed --
backend "nccl" (= RCCL over xGMI) moves the blocks GPU to GPU, "gloo" serves the CPU tests.  torch.distributed is plumbing
here; the dealing and the block format live in the C++ host library (twilight_amd/csrc/host/align_gpu.cpp).
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


class _DevBuf:
    """A device pointer of the C++ side as an object torch.as_tensor understands (CUDA array interface, zero copy)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2, "strides": None}


def make_device_exchange(device):
    """The all-gather of DEVICE blocks a sharded twl_msa run calls once per level (include/twl_msa.h, twl_msa_shard_device):
    exchange(send_dev_ptr, bytes_per_rank, recv_dev_ptr) -> 0, recv = [world][bytes_per_rank] in the HBM of `device`.  Backend nccl
    (= RCCL over xGMI): ONE all_gather_into_tensor on the library's own buffers, nothing touches the host.  Backend gloo (tests: two
    processes sharing one GPU, which RCCL refuses): the blocks bounce through host tensors inside this function."""
    import torch
    import torch.distributed as dist

    def exchange(send_ptr, nbytes, recv_ptr):
        world = dist.get_world_size()
        with torch.cuda.device(device):
            send = torch.as_tensor(_DevBuf(send_ptr, nbytes), device=device)
            recv = torch.as_tensor(_DevBuf(recv_ptr, nbytes * world), device=device)
            if dist.get_backend() == "nccl":
                dist.all_gather_into_tensor(recv, send)
            else:
                h = send.cpu()
                parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, h)
                recv.copy_(torch.cat(parts))
            torch.cuda.synchronize(device)          # the library's stream is not torch's: the collective is complete when we return
        return 0

    return exchange


def make_exchange(device=None):
    """The all-gather a sharded twl_msa run calls once per level: exchange(send_ptr, bytes_per_rank, recv_ptr) -> 0.
    send/recv are host buffers of the C++ side (recv = [world][bytes_per_rank]).  With a CUDA/HIP `device` (backend nccl) the
    blocks travel GPU to GPU over RCCL/xGMI; with device None (backend gloo) they stay on the host."""
    import ctypes

    import torch
    import torch.distributed as dist

    def exchange(send_ptr, nbytes, recv_ptr):
        world = dist.get_world_size()
        send = torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(send_ptr), dtype=torch.uint8)
        recv = torch.frombuffer((ctypes.c_uint8 * (nbytes * world)).from_address(recv_ptr), dtype=torch.uint8)
        if device is not None:
            d_send = send.to(device, non_blocking=False)
            d_recv = torch.empty(nbytes * world, dtype=torch.uint8, device=device)
            dist.all_gather_into_tensor(d_recv, d_send)
            recv.copy_(d_recv)
        else:
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, send)
            recv.copy_(torch.cat(parts))
        return 0

    return exchange
