"""N>1 path on CPU: 2 ranks over gloo shard a level by pairs, align their shards, and the union equals the whole level."""
import os
import socket

import numpy as np
import pytest

from twilight_amd import dist as tdist
from twilight_amd import synth


def test_lpt_shards_are_a_balanced_partition():
    rng = np.random.default_rng(0)
    costs = rng.integers(100, 20000, size=301)
    for world in (1, 2, 3, 8):
        shards = tdist.lpt_shards(costs, world)
        allidx = np.sort(np.concatenate(shards))
        assert np.array_equal(allidx, np.arange(costs.size))
        loads = np.array([costs[s].sum() for s in shards])
        assert loads.max() - loads.min() <= costs.max()          # LPT bound
        for s in shards:
            assert np.all(np.diff(costs[s]) <= 0)                # each shard stays in descending-cost (launch) order
    assert tdist.lpt_shards(np.array([5, 4]), 4)[2].size == 0   # fewer pairs than ranks: idle ranks get nothing


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist

    import oracle_lib as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    M = synth.nucleotide_matrix()
    batch = synth.make_level_batch(9, 300, members=((1, 4), (1, 4)), seed=77, length_jitter=0.4)   # same batch on every rank
    p = O.make_params(M)

    def align_fn(sub):                     # the CPU checker stands in for the per-rank GPU aligner in this CPU test
        a, n, e, st = O.align_batch(p, sub, threads=1)
        align_fn.cells += st.cells
        return a, n, e

    align_fn.cells = 0
    import time

    t0 = time.perf_counter()
    (mine, aln, n, err), full = tdist.align_level_sharded(align_fn, batch, rank, world)
    cells, secs = tdist.reduce_report(align_fn.cells, time.perf_counter() - t0)
    if rank == 0:
        np.savez(os.path.join(out_dir, "full.npz"), aln=full[0], n=full[1], err=full[2], cells=cells, secs=secs)
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_ranks_gloo_equal_single_process(tmp_path):
    import torch.multiprocessing as mp

    import oracle_lib as O

    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    got = np.load(tmp_path / "full.npz")
    batch = synth.make_level_batch(9, 300, members=((1, 4), (1, 4)), seed=77, length_jitter=0.4)
    a, n, e, st = O.align_batch(O.make_params(synth.nucleotide_matrix()), batch, threads=2)
    assert np.array_equal(got["n"], n) and np.array_equal(got["err"], e)
    for i in range(batch.n_pairs):
        assert np.array_equal(got["aln"][i, : n[i]], a[i, : n[i]])
    assert int(got["cells"]) == st.cells            # SUM over ranks of band cells == whole level
    assert float(got["secs"]) > 0                   # MAX over ranks of wall time
