"""N > 1 on the hardware there is: TWO processes share ONE MI355X (both on cuda:0), deal the pairs of every level between them, align
their shares with the device-resident level kernel and exchange the paths (gloo between the processes: RCCL refuses two ranks on one
device).  Expected: every rank writes the MSA of the independent replay's fixture (tests/golden/e2e_variants.json) and the band cells
of the two ranks add up to the fixture's.  Families with gappy-column runs, deferred pairs, cached profiles and compressed groups."""
import hashlib
import json
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from variants import VARIANTS, write_family  # noqa: E402

FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_variants.json")))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tree, fasta, out_dir, typ, flags, env, device_exchange, try_native=False):
    sys.path.insert(0, ROOT)
    os.environ.update(env)                      # (the thresholds are read when the host library loads)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from twilight_amd import dist as tdist
    from twilight_amd import msa

    out = os.path.join(out_dir, f"rank{rank}.aln")
    m = msa.Msa(["-t", tree, "-i", fasta, "-o", out, "--type", typ, "--gpu-index", "0"] + list(flags))
    if try_native:
        # the library's own communicator cannot be made here (RCCL refuses two ranks on one device): twl_msa_shard_rccl must say so on every rank
        # -- an error code, not the end of the process -- and leave the handle usable for a caller's own collective
        ids = [msa.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        try:
            m.shard_rccl(rank, world, ids[0])
            raise SystemExit("twl_msa_shard_rccl succeeded with two ranks on one device")
        except msa.MsaError as e:
            assert "twl_comm_init failed" in str(e), str(e)
    if device_exchange:
        # device blocks for every level of both passes (the deferred pass runs on the resident rows too); the host callback stays registered for the host-staged kernel
        m.shard(rank, world, tdist.make_exchange(None), exchange_device=tdist.make_device_exchange(torch.device("cuda:0")))
    else:
        m.shard(rank, world, tdist.make_exchange(None))
    m.upload()
    m.align()
    tot, levels = m.report()
    m.write()
    m.close()
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array([tot.band_cells, tot.pairs, tot.n_levels] + [lv.pairs for lv in levels], dtype=np.int64))
    np.save(os.path.join(out_dir, f"rank{rank}_x.npy"), np.array([lv.exchange_ms for lv in levels], dtype=np.float64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_processes_without_subtree_ownership(built, tmp_path):
    """--test-no-ownership: every level dealt and exchanged (the design of round 3) -- the same MSA, and an exchange on every level."""
    import torch.multiprocessing as mp

    name = "nuc_deferrals_cache_compress"
    _, fam, ins, flags, env = [v for v in VARIANTS if v[0] == name][0]
    d = str(tmp_path)
    t, f, typ = write_family(d, fam, ins)
    mp.start_processes(_worker, args=(2, _free_port(), t, f, d, typ, list(flags) + ["--test-no-ownership"], env, True), nprocs=2, join=True, start_method="spawn")
    fx = FIX[name]
    for rank in range(2):
        assert hashlib.md5(open(os.path.join(d, f"rank{rank}.aln"), "rb").read()).hexdigest() == fx["md5"], f"rank {rank}"
        assert all(v > 0 for v in np.load(os.path.join(d, f"rank{rank}_x.npy")))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("device_exchange", [False, True])
@pytest.mark.parametrize("name", ["nuc_default", "nuc_r0.7", "nuc_deferrals_cache_compress", "nuc_length_deviation_filter", "prot_cache_and_compress",
                                  "nuc_10500_leaves_more_than_10000_on_one_side", "nuc_xdrop_failure_retried_in_deferred_pass"])
def test_two_processes_on_one_gpu_write_the_fixture_msa(built, tmp_path, name, device_exchange):
    import torch.multiprocessing as mp

    _, fam, ins, flags, env = [v for v in VARIANTS if v[0] == name][0]
    d = str(tmp_path)
    t, f, typ = write_family(d, fam, ins)
    mp.start_processes(_worker, args=(2, _free_port(), t, f, d, typ, flags, env, device_exchange), nprocs=2, join=True, start_method="spawn")
    fx = FIX[name]
    for rank in range(2):
        assert hashlib.md5(open(os.path.join(d, f"rank{rank}.aln"), "rb").read()).hexdigest() == fx["md5"], f"rank {rank}"
        rep = np.load(os.path.join(d, f"rank{rank}.npy"))
        assert int(rep[0]) == fx["band_cells"], f"rank {rank}: band cells (sum over the ranks of every level)"
        assert list(rep[3:3 + len(fx["pairs_per_level"])]) == fx["pairs_per_level"]
        # subtree ownership (align_owned.cpp): the levels below the cut are aligned by their owners alone -- no exchange there, ONE where the subtrees meet,
        # then one per level; both passes of the run
        x = np.load(os.path.join(d, f"rank{rank}_x.npy"))
        n_main = len(fx["pairs_per_level"])
        quiet = [k for k in range(n_main) if x[k] == 0.0]
        assert quiet == list(range(len(quiet))) and len(quiet) < n_main, (name, x[:n_main].tolist())      # (the level of the cut carries the subtree exchange)
        assert all(v > 0 for v in x[len(quiet):]), (name, x.tolist())
        if sum(fx["pairs_per_level"]) >= 39:
            assert len(quiet) >= 1, (name, x[:n_main].tolist())                                           # (8 subtrees per rank fit under a cut above the leaf level)


@pytest.mark.timeout(900)
def test_no_communicator_is_an_error_code_and_the_handle_stays_usable(built, tmp_path):
    """twl_msa_shard_rccl where RCCL cannot make a communicator: both ranks get an error (bench.py then agrees on torch.distributed's callbacks), shard the
    same handles through callbacks and write the fixture's MSA."""
    import torch.multiprocessing as mp

    name = "nuc_default"
    _, fam, ins, flags, env = [v for v in VARIANTS if v[0] == name][0]
    d = str(tmp_path)
    t, f, typ = write_family(d, fam, ins)
    mp.start_processes(_worker, args=(2, _free_port(), t, f, d, typ, flags, env, True, True), nprocs=2, join=True, start_method="spawn")
    for rank in range(2):
        assert hashlib.md5(open(os.path.join(d, f"rank{rank}.aln"), "rb").read()).hexdigest() == FIX[name]["md5"], f"rank {rank}"


@pytest.mark.timeout(900)
def test_bench_with_gpus_2_as_a_plain_command(built, tmp_path):
    """`python bench.py --gpus 2` without a launcher: the script starts torch.distributed.run itself as a child (before it touches a GPU) and relays rank 0's
    ONE JSON line.  TWL_BENCH_ONE_GPU puts both ranks on this box's one device (gloo between them: a code-path run, never a measurement)."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, TWL_BENCH_ONE_GPU="1", TWL_BENCH_DIR=str(tmp_path / "fam"))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--leaves", "400", "--length", "1200",
                        "--no-cpu", "--no-peak", "--no-e2e", "--no-survey8d"], capture_output=True, text=True, env=env, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0
    assert "n_ranks_seen_by_rccl" in line and line["n_ranks_seen_by_rccl"] is None      # (gloo carried this run's exchange)
    assert line["config"]["msa_md5"]


def _two_devices():
    import torch

    return torch.cuda.device_count() >= 2


@pytest.mark.timeout(900)
@pytest.mark.parametrize("no_ownership", [False, True])
def test_cli_on_two_gpus_writes_the_single_gpu_msa(built, tmp_path, no_ownership):
    """twilight-mi355x --gpu-index 0,1: one forked process per GPU, the library's own RCCL communicator (twl_comm_*), subtree ownership below the cut and one all-gather per
    level above it.  Needs a box with two devices (the multi-rank RCCL path had no hardware test); skipped on the one-GPU pool."""
    if not _two_devices():
        pytest.skip("needs two GPUs")
    name = "nuc_deferrals_cache_compress"
    _, fam, ins, flags, env = [v for v in VARIANTS if v[0] == name][0]
    d = str(tmp_path)
    t, f, typ = write_family(d, fam, ins)
    out = os.path.join(d, "two.aln")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import subprocess

    r = subprocess.run([os.path.join(root, "twilight_amd", "twilight-mi355x"), "-t", t, "-i", f, "-o", out, "--type", typ, "--gpu-index", "0,1"] + list(flags) +
                       (["--test-no-ownership"] if no_ownership else []), capture_output=True, text=True, env=dict(os.environ, **env), timeout=800)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == FIX[name]["md5"]


@pytest.mark.timeout(900)
def test_bench_on_two_gpus_reports_two_ranks_in_the_communicator(built, tmp_path):
    if not _two_devices():
        pytest.skip("needs two GPUs")
    import json
    import subprocess
    import sys

    env = dict(os.environ, TWL_BENCH_DIR=str(tmp_path / "fam"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TWL_BENCH_ONE_GPU"):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--leaves", "600", "--length", "1500",
                        "--no-cpu", "--no-peak", "--no-e2e", "--no-survey8d"], capture_output=True, text=True, env=env, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert line["n_gpus"] == 2 and line["n_ranks_seen_by_rccl"] == 2 and line["config"]["msa_md5"]
