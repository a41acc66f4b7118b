"""N > 1 path on CPU: 2 processes over gloo align ONE family together through the product's host library.

What runs is the real orchestration of twilight_amd/csrc/host (twl_msa_open / twl_msa_shard / twl_msa_align: deal the pairs of
every level to the ranks, align the own share, all-gather the paths through twilight_amd.dist.make_exchange, commit all of them),
in the CPU-check build oracle/libtwl_host_cpucheck.so, where the GPU library's twl_align_batch is answered by the oracle
(oracle/twl_align_cpu_shim.cpp) -- this box has no GPU.  Expected: both ranks write the MSA the single-process CPU checker writes.
"""
import hashlib
import os
import socket
import subprocess

import numpy as np
import pytest

from twilight_amd import dist as tdist
from twilight_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPUCHECK = os.path.join(ROOT, "oracle", "libtwl_host_cpucheck.so")


def test_lpt_shards_are_a_balanced_partition():
    rng = np.random.default_rng(0)
    costs = rng.integers(100, 20000, size=301)
    for world in (1, 2, 3, 8):
        shards = tdist.lpt_shards(costs, world)
        allidx = np.sort(np.concatenate(shards))
        assert np.array_equal(allidx, np.arange(costs.size))
        loads = np.array([costs[s].sum() for s in shards])
        assert loads.max() - loads.min() <= costs.max()          # LPT bound
    assert tdist.lpt_shards(np.array([5, 4]), 4)[2].size == 0   # fewer pairs than ranks: idle ranks get nothing


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _family(tmp, leaves, length, P, seed, **kw):
    nwk, seqs = synth.make_family(leaves, length, P=P, seed=seed, **kw)
    t, f = os.path.join(tmp, "t.nwk"), os.path.join(tmp, "s.fa")
    open(t, "w").write(nwk + "\n")
    with open(f, "w") as fh:
        for name, s in seqs:
            fh.write(f">{name}\n{s}\n")
    return t, f


def _worker(rank, world, port, tree, fasta, out_dir, typ):
    import sys

    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from twilight_amd import dist as tdist
    from twilight_amd import msa

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = os.path.join(out_dir, f"rank{rank}.aln")
    m = msa.Msa(["-t", tree, "-i", fasta, "-o", out, "--type", typ, "--host-staged"], lib_path=CPUCHECK)
    m.shard(rank, world, tdist.make_exchange(None))
    m.align()
    tot, levels = m.report()
    m.write()
    m.close()
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array([tot.band_cells, tot.pairs, tot.n_levels] + [lv.pairs for lv in levels], dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


def _md5(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("typ,leaves,length", [("n", 40, 400), ("p", 24, 150)])
def test_two_ranks_gloo_align_one_family_like_the_cpu_checker(tmp_path, typ, leaves, length):
    import torch.multiprocessing as mp

    assert os.path.exists(CPUCHECK), "run `make -C oracle` (build() does)"
    tmp = str(tmp_path)
    tree, fasta = _family(tmp, leaves, length, 6 if typ == "n" else 22, seed=4242, sub=0.08, indel=0.01)
    # single process, the CPU checker (host mirror + oracle DP)
    ref = os.path.join(tmp, "ref.aln")
    r = subprocess.run([os.path.join(ROOT, "oracle", "e2e_oracle"), "-t", tree, "-i", fasta, "-o", ref, "--type", typ, "--threads", "2"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("E2E")][-1]
    ref_cells = int(line.split("band_cells=")[1].split()[0])
    ref_levels = [int(x) for x in line.split("pairs_per_level=")[1].split()[0].split("/")]
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, tree, fasta, tmp, typ), nprocs=2, join=True, start_method="spawn")
    for rank in range(2):
        assert _md5(os.path.join(tmp, f"rank{rank}.aln")) == _md5(ref)          # every rank holds the whole, identical MSA
        rep = np.load(os.path.join(tmp, f"rank{rank}.npy"))
        assert int(rep[0]) == ref_cells                                            # band cells: SUM over the ranks of each level
        assert list(rep[3:]) == ref_levels                                         # same level batches as the single process


@pytest.mark.timeout(300)
def test_single_process_cpucheck_library_equals_cpu_checker(tmp_path):
    import sys

    sys.path.insert(0, ROOT)
    from twilight_amd import msa

    tmp = str(tmp_path)
    tree, fasta = _family(tmp, 30, 300, 6, seed=99, sub=0.05, indel=0.01)
    ref = os.path.join(tmp, "ref.aln")
    r = subprocess.run([os.path.join(ROOT, "oracle", "e2e_oracle"), "-t", tree, "-i", fasta, "-o", ref, "--threads", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = os.path.join(tmp, "lib.aln")
    with msa.Msa(["-t", tree, "-i", fasta, "-o", out, "--host-staged"], lib_path=CPUCHECK) as m:
        m.align()
        m.write()
        tot, levels = m.report()
        assert tot.n_levels == len(levels) and tot.aln_len > 0
    assert _md5(out) == _md5(ref)


@pytest.mark.timeout(300)
def test_a_handle_with_lowered_thresholds_leaves_the_next_one_alone(tmp_path):
    """--test-cal-profile-th / --test-update-seq-th are per RUN (round 5: as process globals, like the reference's two constants, a handle that lowered them changed
    the cached-profile and compressed-group branches -- and with them the band cells and the rows -- of every later handle of the process)."""
    import sys

    sys.path.insert(0, ROOT)
    from twilight_amd import msa

    tmp = str(tmp_path)
    tree, fasta = _family(tmp, 50, 300, 6, seed=17, sub=0.08, indel=0.03)
    ref = os.path.join(tmp, "ref.aln")
    r = subprocess.run([os.path.join(ROOT, "oracle", "e2e_oracle"), "-t", tree, "-i", fasta, "-o", ref, "--threads", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    ref_cells = int([l for l in r.stdout.splitlines() if l.startswith("E2E")][-1].split("band_cells=")[1].split()[0])
    low = os.path.join(tmp, "low.aln")
    with msa.Msa(["-t", tree, "-i", fasta, "-o", low, "--host-staged", "--test-cal-profile-th", "2", "--test-update-seq-th", "3"], lib_path=CPUCHECK) as m:
        m.align()
        m.write()
        low_cells = m.report()[0].band_cells
    out = os.path.join(tmp, "lib.aln")
    with msa.Msa(["-t", tree, "-i", fasta, "-o", out, "--host-staged"], lib_path=CPUCHECK) as m:
        m.align()
        m.write()
        cells = m.report()[0].band_cells
    assert cells == ref_cells and _md5(out) == _md5(ref)
    assert low_cells > 0
