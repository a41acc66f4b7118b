#!/usr/bin/env python3
"""bench.py -- TALCO-XDrop level-batch throughput on MI355X (BASELINE.json metric: DP cells/s).

One "step" = one pass of the hot path (twl_align_batch_device: column packing + the DP/traceback
kernel) over one synthetic guide-tree-level batch that is already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs P] [--length L]

For N > 1 the driver launches one process per GPU with torch.distributed.run; the pairs of a level
are independent, so each rank aligns its own shard (no data-path collective) and rank 0 reports the
whole-job rate: all ranks' band cells / max-over-ranks wall time ("weak" scaling: per-GPU work fixed).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from twilight_amd import synth  # noqa: E402

B_CELL_NUC = 64          # algorithmic operand bytes per band cell, P=6: 2*P*4 + 4*4
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01", "final_pmc_summary.json")


def measured_traffic_per_launch(cells_per_launch):
    """HBM bytes per DP-kernel launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, per the
    gfx950 correction in MI355X_MICROARCH.md), scaled by band cells to this launch size.  None if no profile is committed."""
    try:
        with open(PMC_SUMMARY) as f:
            h = json.load(f)["hbm_per_launch"]
        return h["traffic_bytes_per_cell"] * cells_per_launch
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=2048, help="sibling pairs per GPU per step")
    ap.add_argument("--length", type=int, default=10000, help="ancestor length in columns (10 kbp)")
    ap.add_argument("--pool", type=int, default=64, help="distinct synthetic pairs generated per rank (replicated to --pairs)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs for the CPU baseline leg (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the wall-clock-to-final-MSA leg (N=1 only: product CLI on a generated 10k x 10 kbp family)")
    return ap.parse_args()


def effective_cpus():
    """CPUs this process may really use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(batch, idx, gpu_paths, gpu_lens, target_seconds=15.0):
    """Oracle ("port" of the reference CPU path, OpenMP over pairs like tbb::parallel_for at alignment-cpu.cpp:46) timed on
    the host cores this process may use.  A short calibration sizes the sample to about `target_seconds` of CPU work; the
    sample is the first k pairs of the very batch the GPU aligned, and the paths are compared while we are at it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    def sub(ix):
        return synth.LevelBatch(P=batch.P, seq_len=batch.seq_len, freq=batch.freq[ix], gap_open=batch.gap_open[ix],
                                gap_extend=batch.gap_extend[ix], len=batch.len[ix], num=batch.num[ix])

    threads = effective_cpus()
    p = O.make_params(synth.nucleotide_matrix())
    cal = idx[: max(threads, 8)]
    t0 = time.perf_counter()
    _, _, _, st = O.align_batch(p, sub(cal), threads=threads)
    rate = st.cells / (time.perf_counter() - t0)
    per_pair = st.cells / len(cal)
    k = int(min(len(idx), max(len(cal), target_seconds * rate / per_pair)))
    t0 = time.perf_counter()
    aln, n, err, st = O.align_batch(p, sub(idx[:k]), threads=threads)
    dt = time.perf_counter() - t0
    parity = bool(np.array_equal(n, gpu_lens[:k]) and all(np.array_equal(aln[i, : n[i]], gpu_paths[i][: n[i]]) for i in range(k)))
    return {"value": st.cells / dt, "unit": "cells/s", "cores": threads, "kind": "port",
            "sample": f"first {k} pairs of the same batch ({st.cells} band cells) in {dt:.1f} s, OpenMP over pairs, {threads} threads",
            "gpu_paths_equal_on_sample": parity}


def wallclock_to_msa(leaves=10000, length=10000):
    """Second half of BASELINE.json's metric: wall-clock from FASTA + guide tree to the final MSA for the RNASim-shaped 10k x 10 kbp family
    (BASELINE config 3), product CLI `twilight-mi355x` on this GPU, measured around the whole process.  Never takes the bench line down."""
    import subprocess
    import tempfile

    try:
        exe = os.path.join(ROOT, "twilight_amd", "twilight-mi355x")
        if not os.path.exists(exe):
            return {"value": None, "unit": "s", "note": "twilight-mi355x not built (run __graft_entry__.build())"}
        with tempfile.TemporaryDirectory(prefix="twl_bench_e2e_") as d:
            outj = os.path.join(d, "e2e.json")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_bench.py"), "--leaves", str(leaves), "--length", str(length), "--out", outj],
                               capture_output=True, text=True, timeout=600)
            if r.returncode != 0 or not os.path.exists(outj):
                return {"value": None, "unit": "s", "note": "e2e run failed: " + (r.stdout + r.stderr)[-300:]}
            e = json.load(open(outj))
        rec = {}
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "r01", "e2e_wallclock_10000x10k.json")))
        except OSError:
            pass
        same_family = bool(rec) and rec.get("leaves") == leaves and rec.get("length") == length
        return {"value": e["gpu"]["wall_s"], "unit": "s", "higher_is_better": False,
                "config": f"synthetic RNASim-shaped family, {leaves} sequences x {length} bp, full progressive alignment over the guide tree ({e['gpu']['levels']} levels), "
                          f"gappy-column removal on, device-resident level path, 1 GPU; time of the whole process (read FASTA + tree ... write MSA)",
                "aln_len": e["aln_len"], "summary": e["gpu"]["summary"], "msa_md5": e["gpu"]["md5"],
                "recorded_cpu_checker_s": rec.get("cpu", {}).get("wall_s") if same_family else None,
                "msa_equals_recorded_cpu_checker_msa": (e["gpu"]["md5"] == rec.get("cpu", {}).get("md5")) if same_family else None,
                "note": "the CPU leg (oracle/e2e_oracle, 16 threads, ~107 s) is recorded in profiles/r01/e2e_wallclock_10000x10k.json, not re-run here"}
    except Exception as ex:  # noqa: BLE001
        return {"value": None, "unit": "s", "note": f"e2e leg failed: {ex}"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    # wall-clock-to-MSA leg first: a child process, started before this process touches the GPU
    e2e = wallclock_to_msa() if (world == 1 and not args.no_e2e) else None

    # ---- synthetic workload, built on the host before the GPU is touched ----
    pool_n = min(args.pool, args.pairs)
    batch = synth.make_level_batch(pool_n, args.length, members=((1, 8), (1, 8)), seed=20260501 + 3 + 1000 * rank)
    idx = np.arange(args.pairs) % pool_n

    import torch
    import torch.distributed as dist

    import twilight_amd as twl

    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    twl.init([local_rank])

    tidx = torch.from_numpy(idx).to(dev)
    freq = torch.from_numpy(batch.freq).to(dev)[tidx].contiguous()
    gop = torch.from_numpy(batch.gap_open).to(dev)[tidx].contiguous()
    gex = torch.from_numpy(batch.gap_extend).to(dev)[tidx].contiguous()
    ln = torch.from_numpy(batch.len).to(dev)[tidx].contiguous()
    nm = torch.from_numpy(batch.num).to(dev)[tidx].contiguous()
    n, sl = args.pairs, batch.seq_len
    aln = torch.zeros((n, 2 * sl), dtype=torch.int8, device=dev)
    alen = torch.zeros(n, dtype=torch.int32, device=dev)
    err = torch.zeros(n, dtype=torch.int16, device=dev)
    params = twl.make_params(synth.nucleotide_matrix())          # CLI defaults: 18/-8/-4, gap -50/-5, xdrop 5000, marker 1024, flen 4096

    def step():
        twl.align_batch_device(params, n, sl, freq.data_ptr(), gop.data_ptr(), gex.data_ptr(), ln.data_ptr(), nm.data_ptr(),
                               aln.data_ptr(), alen.data_ptr(), err.data_ptr(), device=local_rank)
        return twl.get_stats(local_rank)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    cells = 0
    kernel_ms = 0.0
    pack_ms = 0.0
    launches = 0
    relaunched = 0
    for _ in range(args.steps):
        st = step()
        cells += st.band_cells
        kernel_ms += st.kernel_ms
        pack_ms += st.pack_ms
        launches += st.n_launches
        relaunched += st.n_relaunched
    fence()
    dt = time.perf_counter() - t0

    n_bad = int((err != 0).sum().item())
    from twilight_amd.dist import reduce_report

    cells_all, dt_max = reduce_report(cells, dt, device=dev)      # SUM of cells, MAX of seconds over ranks

    if rank == 0:
        value = cells_all / dt_max
        achieved = (cells * B_CELL_NUC) / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0     # GB/s, this rank's DP kernel
        out = {
            "metric": "DP band cells/s, TALCO-XDrop level-batch alignment (bit-exact vs reference CPU path)",
            "value": value,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / max(1, args.steps),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"RNASim-shaped 10k seqs x 10 kbp: one guide-tree level batch, {args.pairs} sibling pairs/GPU "
                            f"of ~{args.length}-column profiles (1-8 member seqs per side, weighted counts, PSGP gap penalties); "
                            f"{pool_n} distinct pairs per rank replicated; scoring 18/-8/-4, gap -50/-5, xdrop 5000, marker 1024, flen 4096",
                "pairs_per_gpu": args.pairs, "seq_len": sl, "P": 6,
                "band_cells_per_step_per_gpu": cells // max(1, args.steps),
                "deferred_pairs": n_bad, "relaunched_pairs": relaunched,
                "window_rows": st.window, "persistent_workgroups": st.grid,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic_per_launch(cells // max(1, launches)),
                "traffic_source": "profiles/r01/final_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 2 --warmup 1 --no-cpu --pairs 1024`: bytes per band cell, times the cells of this launch)",
                "kernel": "twl::talco_kernel<6, 8, 2, false, true, true, 4, 2>", "kernel_ms_per_launch": kernel_ms / max(1, launches),
                "algorithmic_bytes_per_cell": B_CELL_NUC, "cells_per_launch": cells // max(1, launches),
                "note": "achieved = band cells x 64 B / DP-kernel time (HIP events on the library stream); the path is "
                        "VALU/latency-bound, real HBM traffic is far below this (DESIGN.md section 5)",
            },
        }
        if world == 1 and not args.no_cpu:
            kmax = args.cpu_sample or min(args.pairs, 4096)
            try:
                out["cpu_baseline"] = cpu_baseline(batch, idx[:kmax], aln[:kmax].cpu().numpy(), alen[:kmax].cpu().numpy())
            except Exception as e:  # the checker must never take the bench line down
                out["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        if e2e is not None:
            out["wallclock_to_msa"] = e2e
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    twl.shutdown()


if __name__ == "__main__":
    main()
