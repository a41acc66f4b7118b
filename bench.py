#!/usr/bin/env python3
"""bench.py -- TALCO-XDrop progressive alignment on MI355X (BASELINE.json metric: DP cells/s + wall-clock to final MSA).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config rnasim10k|rnasim100k|protein5k|rnasim1k_band512]

Default workload = the configuration the metric is quoted on: RNASim-shaped 10 000 sequences x 10 kbp, FULL progressive
alignment over the guide tree (BASELINE.json configs[2]).  One "step" = one complete pass of the hot path over that family:
every guide-tree level's batch of sibling pairs through the product path (libtwl_host -> libtwl_align: profile kernels, the
TALCO-XDrop DP kernel, write-back kernels), with the sequences already resident in HBM when the timed region starts
(twl_msa_upload is outside it).  Each step runs on a fresh handle, all opened and uploaded before the clock starts.

    value        = band cells of K passes / wall seconds of the timed region (whole job, max over ranks)
    dp_kernel    = the same cells / summed DP-kernel time (HIP events inside the library): the rate the roofline fraction uses
    levels       = per guide-tree level {pairs, cells, kernel_ms, level_ms} of the last pass (reference progressive.cpp:178-189 prints the same)
    peak_level   = one wide synthetic level (2048 pairs x 10 kbp profiles) through twl_align_batch_device: the kernel's best case

For N > 1 the driver launches one process per GPU (torch.distributed.run); the ranks align the SAME family together: every rank
holds a replica, aligns the pairs of each level that the longest-first deal gives to it, and the paths are all-gathered once per
level over RCCL (twilight_amd/dist.py) -- "strong" scaling: total work fixed.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from twilight_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector peak (256 CUs x 128 lanes x 2 flop x 2.4 GHz); SURVEY.md 8d(ii) prices F_cell against it
F_CELL = {6: 120.0, 22: 1460.0}  # SURVEY.md 8d(ii) / BASELINE.md 3: fp32 operations of one band cell (column score + recurrence), nucleotide / protein
# algorithmic operand bytes per band cell (SURVEY.md 8d / BASELINE.md 3): 2 profile columns + 4 gap penalties = 2*P*4 + 16
CONFIGS = {
    "rnasim10k": dict(kind="family", leaves=10000, length=10000, type="n", P=6, bcell=64, sub=0.015, indel=0.001,
                      name="RNASim-shaped 10k seqs x 10 kbp, full progressive over the guide tree, gappy-column removal on (BASELINE configs[2])"),
    "rnasim100k": dict(kind="family", leaves=100000, length=1600, type="n", P=6, bcell=64, sub=0.015, indel=0.001,
                       name="RNASim-shaped 100k seqs x 1.6 kbp, full progressive (BASELINE configs[3])"),
    "protein5k": dict(kind="family", leaves=5000, length=2000, type="p", P=22, bcell=192, sub=0.015, indel=0.001,
                      name="protein 5k seqs x 2 kaa, 5xBLOSUM62, full progressive (BASELINE configs[4])"),
    "rnasim1k_band512": dict(kind="level", pairs=320, length=1600, type="n", P=6, bcell=64, flen=512, xdrop=4000,
                             name="RNASim-shaped 1k seqs x 1.6 kbp: the leaf level (320 sibling pairs) as ONE batch, fLen 512 / xdrop 4000 (BASELINE configs[1])"),
}
PROF = "r06"
ISA = os.path.join(ROOT, "profiles", PROF, "isa_block_step.json")      # static instruction counts / VALU units of a block step, from the disassembly (tools/isa_block_step.py --all)


def isa_key(isa, kernel_name):
    """The entry of profiles/r06/isa_block_step.json a kernel name of the library belongs to.  Plain launches are spelt as the profiler spells them; a tile-parallel level is
    named by its families ("talco_lean_kernel<6, 4, 3, 2, 4, false, false, 2 / 1> + <6, 16, 1, ...>"): its cells are computed by the TILE jobs (MT 1) of the first family."""
    import re
    m = re.search(r"talco_lean_kernel<(\d+, \d+, \d+, \d+, \d+, false, false, )([^>]*)>", kernel_name)
    if not m: return None
    pre, rest = m.group(1), m.group(2)
    want = pre + ("1" if "/" in rest else rest)
    keys = sorted(isa["kernels"], key=len, reverse=True)
    return next((k for k in keys if k == "talco_lean_kernel<" + want + ">"), None) or next((k for k in keys if k.startswith("talco_lean_kernel<" + pre) and k.endswith((", 1>" if "/" in rest else ", 0>"))), None)


def pmc_path(config, workload):
    """Counters of the bench command of this configuration, taken on the same kernel sources (tools/final_profiles.sh)."""
    return os.path.join(ROOT, "profiles", PROF, "bench_pmc_summary.json" if (config == "rnasim10k" and workload == "calibrated") else f"{config}_pmc_summary.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="rnasim10k", choices=sorted(CONFIGS))
    ap.add_argument("--leaves", type=int, default=0, help="override the family size (development)")
    ap.add_argument("--length", type=int, default=0, help="override the sequence length (development)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-peak", action="store_true", help="skip the single-level peak leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the wall-clock-to-final-MSA leg (product CLI as a child process)")
    ap.add_argument("--no-survey8d", action="store_true", help="skip the one-pass sub-record on the family of SURVEY.md 8d as written (default configuration only)")
    ap.add_argument("--keep", default="", help="directory for the generated family (kept)")
    ap.add_argument("--workload", default="calibrated", choices=["calibrated", "survey8d"],
                    help="family parameters: calibrated on the reference's RNASim sample (default), or SURVEY.md 8d as written (per-branch substitution U(0.03, 0.10), "
                         "indel 0.005/site, seed 20260501 + config index)")
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def write_family(cfg, d):
    sys.setrecursionlimit(1000000)
    if cfg.get("workload") == "survey8d":
        nwk, seqs = synth.make_family(cfg["leaves"], cfg["length"], P=cfg["P"], seed=20260501 + cfg["index"], indel=0.005, sub_range=(0.03, 0.10))
    else:
        nwk, seqs = synth.make_family(cfg["leaves"], cfg["length"], P=cfg["P"], seed=20260501, sub=cfg["sub"], indel=cfg["indel"])
    tree, fasta = os.path.join(d, "t.nwk"), os.path.join(d, "s.fa")
    with open(tree, "w") as f:
        f.write(nwk + "\n")
    with open(fasta, "w") as f:
        for name, s in seqs:
            f.write(f">{name}\n{s}\n")
    return tree, fasta


def cpu_baseline(batch, matrix, pk, gpu_paths, gpu_lens, target_seconds=12.0):
    """The oracle ("port" of the reference CPU path; OpenMP over pairs like tbb::parallel_for at alignment-cpu.cpp:46) timed on this
    box's host cores, all usable cores and one: a bounded sample of wide-level pairs of the workload's shape, the very pairs the GPU
    aligned in the peak leg, and the paths are compared while we are at it.  The reference's own code, with its vector<vector<float>>
    layout, 14 allocations per tile and AVX2 masked loads, is restated in oracle/talco_faithful.cpp (held to the checker bit for bit by
    tests/test_oracle_cpu.py) and timed here as well, on a share of the sample: `reference_layout`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    def sub(ix):
        return synth.LevelBatch(P=batch.P, seq_len=batch.seq_len, freq=batch.freq[ix], gap_open=batch.gap_open[ix],
                                gap_extend=batch.gap_extend[ix], len=batch.len[ix], num=batch.num[ix])

    threads = effective_cpus()
    p = O.make_params(matrix, **pk)
    n = batch.n_pairs                      # distinct pairs; the sample cycles through them (the GPU leg replicated them the same way)
    cal = np.arange(min(n, max(threads, 4)))
    t0 = time.perf_counter()
    _, _, _, st = O.align_batch(p, sub(cal), threads=threads)
    rate = st.cells / (time.perf_counter() - t0)
    per_pair = st.cells / len(cal)
    k = int(max(len(cal), min(4096, target_seconds * rate / per_pair)))
    ix = np.arange(k) % n
    t0 = time.perf_counter()
    aln, ln, err, st = O.align_batch(p, sub(ix), threads=threads)
    dt = time.perf_counter() - t0
    parity = bool(all(ln[i] == gpu_lens[ix[i]] and np.array_equal(aln[i, : ln[i]], gpu_paths[ix[i]][: ln[i]]) for i in range(k)))
    k1 = max(1, int(round(3.0 * rate / threads / per_pair)))
    t0 = time.perf_counter()
    _, _, _, st1 = O.align_batch(p, sub(np.arange(k1) % n), threads=1)
    dt1 = time.perf_counter() - t0
    # the reference's layout and allocation pattern (SURVEY 8d "faithful" mode), measured on this box: all threads, then one
    kf = max(len(cal), k // 3)
    t0 = time.perf_counter()
    fa, fl, fe, fcells = O.align_batch_faithful(p, sub(np.arange(kf) % n), threads=threads)
    dtf = time.perf_counter() - t0
    same = bool(all(fl[i] == ln[i] and np.array_equal(fa[i, : fl[i]], aln[i, : ln[i]]) for i in range(min(kf, k))))
    t0 = time.perf_counter()
    _, _, _, fcells1 = O.align_batch_faithful(p, sub(np.arange(k1) % n), threads=1)
    dtf1 = time.perf_counter() - t0
    flat = {"value": st.cells / dt, "one_thread_value": st1.cells / dt1, "unit": "cells/s", "cores": threads,
            "sample": f"{k} pairs of ~{batch.seq_len}-column profiles of this workload's shape ({st.cells} band cells) in {dt:.1f} s on {threads} threads "
                      f"(OpenMP over pairs); 1 thread: {k1} pairs in {dt1:.1f} s",
            "note": "flat-array restatement of the reference CPU path (oracle/talco_oracle.c)"}
    layout = {"value": fcells / dtf, "one_thread_value": fcells1 / dtf1, "unit": "cells/s", "cores": threads,
              "sample": f"{kf} of those pairs in {dtf:.1f} s on {threads} threads; 1 thread: {k1} pairs in {dtf1:.1f} s", "paths_equal_to_port": same,
              "note": "oracle/talco_faithful.cpp: the same algorithm with the reference's data layout and allocation pattern (vector<vector<float>> "
                      "profiles, 14 new[] per tile, AVX2 masked loads; TALCO-XDrop.cpp:279-312,378-395), measured on this box"}
    # the stated baseline is the FASTER of the two restatements (the reference's layout with its AVX2 loads beats the flat port on
    # this box, and a baseline must not understate the reference); the other one stays beside it
    best, which = (layout, "reference_layout") if layout["value"] >= flat["value"] else (flat, "flat_port")
    return {"value": best["value"], "unit": "cells/s", "cores": threads, "kind": "port", "which": which, "cpu_model": cpu_model(),
            "one_thread_value": best["one_thread_value"], "sample": best["sample"],
            "note": "the faster of two CPU restatements of the reference path, both timed here: `flat_port` (oracle/talco_oracle.c, the checker) and `reference_layout` "
                    "(oracle/talco_faithful.cpp, the reference's own data layout with AVX2 masked loads); the reference itself cannot be built in this image",
            "flat_port": flat, "reference_layout": layout,
            "gpu_paths_equal_on_sample": parity}


def peak_level(twl, dev, local_rank, cfg, pairs=2048, pool=512, reps=3, warm=1, fence=None):
    """One wide level through twl_align_batch_device (HBM in, HBM out): what the DP kernel does when the GPU is full.
    `warm` untimed launches, then `reps` timed ones; with `fence` (the bench's barrier + synchronize) the wall clock of exactly those `reps`
    launches -- inputs resident in HBM, results left in HBM -- is returned as out["timed_s"]."""
    import torch

    P = cfg["P"]
    prot = P == 22
    length = cfg["length"]
    if cfg["kind"] == "level":
        pairs = cfg["pairs"]
    pool = min(pool, pairs)
    batch = synth.make_level_batch(pool, length, members=((1, 8), (1, 8)), seed=20260501 + 3, P=P, sub=(0.15 if prot else 0.06))
    idx = np.arange(pairs) % pool
    tidx = torch.from_numpy(idx).to(dev)
    freq = torch.from_numpy(batch.freq).to(dev)[tidx].contiguous()
    gop = torch.from_numpy(batch.gap_open).to(dev)[tidx].contiguous()
    gex = torch.from_numpy(batch.gap_extend).to(dev)[tidx].contiguous()
    ln = torch.from_numpy(batch.len).to(dev)[tidx].contiguous()
    nm = torch.from_numpy(batch.num).to(dev)[tidx].contiguous()
    sl = batch.seq_len
    aln = torch.zeros((pairs, 2 * sl), dtype=torch.int8, device=dev)
    alen = torch.zeros(pairs, dtype=torch.int32, device=dev)
    err = torch.zeros(pairs, dtype=torch.int16, device=dev)
    matrix = synth.protein_matrix() if prot else synth.nucleotide_matrix()
    pk = {k: cfg[k] for k in ("flen", "xdrop") if k in cfg}
    params = twl.make_params(matrix, **pk)
    torch.cuda.synchronize()        # the inputs above were produced on torch's stream; the library runs on its own
    cells = kms = 0.0
    nominal = 0
    timed_s = None
    for r in range(warm + reps):
        if r == warm and fence is not None:
            fence()
            t0 = time.perf_counter()
        twl.align_batch_device(params, pairs, sl, freq.data_ptr(), gop.data_ptr(), gex.data_ptr(), ln.data_ptr(), nm.data_ptr(),
                               aln.data_ptr(), alen.data_ptr(), err.data_ptr(), device=local_rank)
        st = twl.get_stats(local_rank)
        if r >= warm:
            cells += st.band_cells
            kms += st.kernel_ms
            nominal = int(st.nominal_cells)
    if fence is not None:
        fence()
        timed_s = time.perf_counter() - t0
    out = {"pairs": pairs, "seq_len": sl, "band_cells_per_launch": int(cells // reps), "nominal_cells_per_launch": nominal, "kernel_ms_per_launch": kms / reps,
           "cells_per_s": cells / (kms * 1e-3), "frac_of_hbm_roofline": cells * cfg["bcell"] / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "deferred_pairs": int((err != 0).sum().item()), "relaunched_pairs": int(st.n_relaunched), "window_rows": int(st.window),
           "persistent_workgroups": int(st.grid), "timed_s": timed_s, "kernel": st.kernel.decode(),
           "workload": f"{pairs} sibling pairs of ~{length}-column profiles (1-8 member sequences per side, weighted counts, PSGP gap penalties), "
                       f"{pool} distinct pairs replicated"}
    k = min(pairs, pool)
    return out, batch, matrix, pk, aln[:k].cpu().numpy(), alen[:k].cpu().numpy()


def wallclock_to_msa(tree, fasta, typ, d):
    """Second half of the metric: wall-clock from FASTA + guide tree to the final MSA, product CLI as a child process (whole process:
    start-up, read, align, write)."""
    exe = os.path.join(ROOT, "twilight_amd", "twilight-mi355x")
    out = os.path.join(d, "cli.aln")
    t0 = time.perf_counter()
    r = subprocess.run([exe, "-t", tree, "-i", fasta, "-o", out, "--type", typ, "-v"], capture_output=True, text=True, timeout=1200)
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        return {"value": None, "unit": "s", "note": "CLI failed: " + (r.stdout + r.stderr)[-300:]}
    tail = [l for l in r.stderr.splitlines() if l.startswith("Wrote")][-1]
    md5 = hashlib.md5(open(out, "rb").read()).hexdigest()
    os.remove(out)
    return {"value": wall, "unit": "s", "higher_is_better": False, "summary": tail, "msa_md5": md5,
            "note": "twilight-mi355x on the same family files, 1 GPU, time of the whole process (HIP start-up, read FASTA + tree, align, write MSA)"}


def survey8d_record(cfg, local_rank, base):
    """One pass over the family of SURVEY.md 8d exactly as written (per-branch substitution U(0.03, 0.10), indel 0.005/site): the regime in
    which most pairs outgrow the fast window, tiles stop converging and 96 % of the sequences come back through the deferred pass.  The
    headline family is calibrated on the reference's RNASim sample; this record keeps the other one in the driver's line."""
    from twilight_amd import msa
    c8 = dict(cfg)
    c8["workload"] = "survey8d"
    d = os.path.join(base, "s8d")
    os.makedirs(d, exist_ok=True)
    t0 = time.perf_counter()
    tree, fasta = write_family(c8, d)
    gen = time.perf_counter() - t0
    # (round 6: an untimed pass on a handle of its own first -- until then the record timed the first touch of every kernel of the wide re-run chain)
    warm = msa.Msa(["-t", tree, "-i", fasta, "-o", os.path.join(d, "w.aln"), "--type", c8["type"], "--gpu-index", str(local_rank)])
    warm.upload()
    warm.align()
    warm.close()
    m = msa.Msa(["-t", tree, "-i", fasta, "-o", os.path.join(d, "o.aln"), "--type", c8["type"], "--gpu-index", str(local_rank)])
    m.upload()
    t0 = time.perf_counter()
    m.align()
    dt = time.perf_counter() - t0
    tot, levels = m.report()
    out = os.path.join(d, "o.aln")
    m.write(out)
    md5 = hashlib.md5(open(out, "rb").read()).hexdigest()
    m.close()
    for f in (out, tree, fasta):
        try:
            os.remove(f)
        except OSError:
            pass
    try:
        os.rmdir(d)
    except OSError:
        pass
    deferred = sum(1 for lv in levels if int(lv.task) == 1)
    return {"cells_per_s": tot.band_cells / dt, "s_per_pass": dt, "band_cells": int(tot.band_cells), "frac_of_hbm_roofline": tot.band_cells * cfg["bcell"] / dt / 1e9 / HBM_PEAK_GBS,
            "dp_kernel_ms": tot.kernel_ms, "levels_main_pass": int(tot.n_levels) - deferred, "deferred_profiles": deferred, "pairs_rerun_in_wider_window": int(tot.relaunched),
            "aln_len": int(tot.aln_len), "msa_md5": md5, "generate_s": gen,
            "note": "one timed pass, after one untimed pass on a handle of its own, over 10 000 x 10 kbp generated with SURVEY.md 8d's parameters, seed 20260501 + 2; the "
                    "CPU checker's MSA for this family has md5 11284078... (tests/golden/e2e_synthetic_expected.json)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # A plain `python bench.py --gpus N`: one process per GPU is started HERE, as a child, before this process has imported torch or touched a
        # GPU (never a re-exec); rank 0's JSON line is relayed, the child's exit code is ours.
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if lines:
            print(lines[-1], flush=True)
        sys.exit(r.returncode if (r.returncode != 0 or lines) else 1)
    # the host library reports on C-level stdout as TWILIGHT does; this process's stdout carries the ONE JSON line only
    json_fd = os.dup(1)
    os.dup2(2, 1)
    cfg = dict(CONFIGS[args.config])
    cfg["workload"] = args.workload
    cfg["index"] = {"rnasim1k_band512": 1, "rnasim10k": 2, "rnasim100k": 3, "protein5k": 4}[args.config]      # BASELINE.json configs[] index
    if args.leaves:
        cfg["leaves"] = args.leaves
    if args.length:
        cfg["length"] = args.length
    family = cfg["kind"] == "family"

    # ---- the workload, built on the host before the GPU is touched: rank 0 writes the family, every rank reads the same files ----
    base = args.keep or os.environ.get("TWL_BENCH_DIR") or os.path.join(tempfile.gettempdir(), f"twl_bench_{os.environ.get('MASTER_PORT', 'solo')}_{os.getppid() if world > 1 else os.getpid()}")
    tree = fasta = None
    gen_s = 0.0
    if family:
        os.makedirs(base, exist_ok=True)
        tree, fasta = os.path.join(base, "t.nwk"), os.path.join(base, "s.fa")
        if rank == 0:
            t0 = time.perf_counter()
            tree, fasta = write_family(cfg, base)
            gen_s = time.perf_counter() - t0
            open(os.path.join(base, "ready"), "w").write("1")
    e2e = None
    if family and world == 1 and not args.no_e2e:
        try:    # a child process, started before this process touches the GPU
            e2e = wallclock_to_msa(tree, fasta, cfg["type"], base)
        except Exception as ex:  # noqa: BLE001
            e2e = {"value": None, "unit": "s", "note": f"e2e leg failed: {ex}"}

    import torch
    import torch.distributed as dist

    import twilight_amd as twl
    from twilight_amd import dist as tdist
    from twilight_amd import msa

    # development: TWL_BENCH_ONE_GPU=1 runs every rank on cuda:0 with gloo between the processes (RCCL refuses two ranks on one device):
    # the N > 1 code path of this script on a one-GPU box; never a measurement
    one_gpu = bool(os.environ.get("TWL_BENCH_ONE_GPU"))
    if one_gpu:
        local_rank = 0
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    if world > 1 or os.environ.get("TWL_BENCH_FORCE_SHARD"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist.barrier()          # rank 0 has written the family
    twl.init([local_rank])
    if os.environ.get("TWL_BENCH_MT_TAIL_PCT"):          # development: how full a last round of the throughput kernel must be to stay there
        twl.set_knob(twl.knobs.KNOB_MT_TAIL_PCT, int(os.environ["TWL_BENCH_MT_TAIL_PCT"]))
    if os.environ.get("TWL_BENCH_THR_SMALL"):            # development: the 512-row throughput window never (1) / on every throughput level (2)
        twl.set_knob(twl.knobs.KNOB_THR_SMALL, int(os.environ["TWL_BENCH_THR_SMALL"]))
    if os.environ.get("TWL_BENCH_MT_LEAD"):              # development: anti-diagonals a scout starts ahead of its tile boundary (default 320)
        twl.set_knob(twl.knobs.KNOB_MT_LEAD, int(os.environ["TWL_BENCH_MT_LEAD"]))
    if os.environ.get("TWL_BENCH_MT_ANCHOR"):            # development: 0 = every scout from the straight line with the long lead
        twl.set_knob(twl.knobs.KNOB_MT_ANCHOR, int(os.environ["TWL_BENCH_MT_ANCHOR"]))
    if os.environ.get("TWL_BENCH_MT_LEAD2"):             # development: the lead of an anchored scout (default 96)
        twl.set_knob(twl.knobs.KNOB_MT_LEAD2, int(os.environ["TWL_BENCH_MT_LEAD2"]))
    if os.environ.get("TWL_BENCH_LEAF_STEP"):            # development: 0 = leaf x leaf levels on the general step
        twl.set_knob(twl.knobs.KNOB_LEAF_STEP, int(os.environ["TWL_BENCH_LEAF_STEP"]))
    if os.environ.get("TWL_BENCH_SCOUT_XDROP_PCT"):      # development: what a narrower band of the pair scouts costs and saves
        twl.set_knob(twl.knobs.KNOB_SCOUT_XDROP_PCT, int(os.environ["TWL_BENCH_SCOUT_XDROP_PCT"]))
    num_cu = torch.cuda.get_device_properties(dev).multi_processor_count

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    peak = pk_batch = None
    if family:
        # ---- K + W fresh handles, opened and made resident in HBM before the clock starts ----
        force_shard = bool(os.environ.get("TWL_BENCH_FORCE_SHARD"))      # development: a 1-rank world still goes through the RCCL all-gather
        # the per-level all-gather: the library's own RCCL communicator (one per handle: ncclCommInitRank on ids rank 0 makes and torch.distributed's
        # store carries); the development switch TWL_BENCH_ONE_GPU (two ranks on one device, which RCCL refuses) keeps the gloo callbacks
        sharded = world > 1 or force_shard
        native = sharded and (not one_gpu or bool(os.environ.get("TWL_BENCH_TRY_NATIVE"))) and not os.environ.get("TWL_BENCH_CALLBACK_EXCHANGE")      # (TRY_NATIVE on one GPU: RCCL refuses two ranks on a device -- the way out below, exercised)
        exchange = tdist.make_exchange(None if one_gpu else dev) if (sharded and not native) else None
        exchange_dev = tdist.make_device_exchange(dev) if (sharded and not native) else None
        native_note = None
        handles = []
        t0 = time.perf_counter()
        for i in range(args.warmup + args.steps):
            m = msa.Msa(["-t", tree, "-i", fasta, "-o", os.path.join(base, f"out_r{rank}_{i}.aln"), "--type", cfg["type"], "--gpu-index", str(local_rank)] + (["-v"] if os.environ.get("TWL_BENCH_VERBOSE") else []))
            if native:
                ids = [msa.rccl_unique_id() if rank == 0 else None]
                if world > 1:
                    dist.broadcast_object_list(ids, src=0)
                ok, why = 1, ""
                try:
                    m.shard_rccl(rank, world, ids[0])   # (twl.init above brought the device up; one communicator per process: handles are aligned one after the other)
                except Exception as e:                  # no communicator on this rank: every rank must take the same way out
                    ok, why = 0, str(e)
                if world > 1:
                    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    ok = int(flag.item())
                if not ok:
                    # the library's own communicator could not be made: the same all-gather through torch.distributed's RCCL (device blocks, callbacks)
                    native = False
                    native_note = f"twl_msa_shard_rccl failed ({why or 'on another rank'}): per-level all-gather through torch.distributed callbacks"
                    sys.stderr.write(f"[bench rank {rank}] {native_note}\n")
                    exchange = tdist.make_exchange(None if one_gpu else dev)
                    exchange_dev = tdist.make_device_exchange(dev)
                    m.shard(rank, world, exchange, exchange_device=exchange_dev)      # (replaces the handle's shard description, also where the communicator came up)
            elif exchange is not None:
                m.shard(rank, world, exchange, exchange_device=exchange_dev)
            m.upload()
            handles.append(m)
        open_s = time.perf_counter() - t0
        for m in handles[: args.warmup]:
            m.align()
        fence()
        t0 = time.perf_counter()
        for m in handles[args.warmup:]:
            m.align()
        fence()
        dt = time.perf_counter() - t0
        cells = 0
        nominal_cells = 0
        kernel_ms = exch_ms = 0.0
        for m in handles[args.warmup:]:
            tot, levels = m.report()
            cells += tot.band_cells
            nominal_cells += int(tot.nominal_cells)
            kernel_ms += tot.kernel_ms
            exch_ms += tot.exchange_ms
        md5 = None
        if rank == 0:
            out = os.path.join(base, "bench.aln")
            handles[-1].write(out)
            md5 = hashlib.md5(open(out, "rb").read()).hexdigest()
            os.remove(out)
        for m in handles:
            m.close()
    else:
        # single-level configuration: the step is one launch of the level through twl_align_batch_device
        # (W warm-up launches, then exactly K timed ones between two fences: the batch is generated and uploaded before the clock starts, the paths stay in HBM)
        peak, pk_batch, matrix, pk, gp, gl = peak_level(twl, dev, local_rank, cfg, reps=max(1, args.steps), warm=max(1, args.warmup), fence=fence)
        dt = peak["timed_s"]
        cells = peak["band_cells_per_launch"] * args.steps
        nominal_cells = peak["nominal_cells_per_launch"] * args.steps
        kernel_ms = peak["kernel_ms_per_launch"] * args.steps
        exch_ms = 0.0
        levels, tot, md5, open_s = [], None, None, 0.0

    # ranks the collective's communicator holds: the library's own (twl_comm_*) or torch.distributed's RCCL; None when gloo carried the exchange (one-GPU development run)
    rccl_world = 0
    if family and sharded:
        if native:
            rccl_world = world
        elif dist.is_initialized() and dist.get_backend() == "nccl":
            rccl_world = dist.get_world_size()
    dt_max = tdist.reduce_report(0.0, dt, device=(None if one_gpu else dev))[1] if world > 1 else dt      # MAX of seconds over ranks (cells are whole-job already)

    if rank == 0:
        steps = max(1, args.steps)
        bcell = cfg["bcell"]
        achieved = cells * bcell / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0          # GB/s of algorithmic operand bytes
        out = {
            "metric": "DP band cells/s, TALCO-XDrop progressive profile-profile alignment (bit-exact vs reference CPU path)",
            "value": cells / dt_max,
            "unit": "cells/s",
            "n_gpus": world,
            "n_ranks_seen_by_rccl": (int(rccl_world) if rccl_world else None),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / steps,
            "higher_is_better": True,
            "scaling": "strong" if family else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": cfg["name"] + ((f"; synthetic family: random root evolved down a random binary tree, per-branch substitution {cfg['sub']} x U(0.5,1.5), "
                                            f"indel rate {cfg['indel']}/site (calibrated on the reference's RNASim sample: band avg 310-420, max ~640 at 10 kbp; milder than "
                                            f"SURVEY 8d's 0.03-0.10 / 0.005, which --workload survey8d runs)" if args.workload == "calibrated" else
                                            f"; synthetic family exactly as SURVEY.md 8d: per-branch substitution U(0.03, 0.10), indel 0.005/site, geometric lengths of mean 3, seed 20260501 + {cfg['index']}")
                                           + "; scoring = CLI defaults (18/-8/-4 or 5xBLOSUM62, gap -50/-5, xdrop 5000, marker 1024, fLen 4096)" if family else ""),
                "family_parameters": args.workload,
                "name": args.config,
            },
        }
        if family:
            out["config"].update({"n_sequences": int(tot.n_sequences), "seq_length": cfg["length"], "levels": int(tot.n_levels), "pairs": int(tot.pairs),
                                  "band_cells_per_pass": int(cells // steps), "aln_len": int(tot.aln_len), "pairs_rerun_in_wider_window": int(tot.relaunched),
                                  "msa_md5": md5, "generate_s": gen_s, "open_and_upload_s_per_handle": open_s / max(1, args.warmup + args.steps),
                                  "parallelism": (f"{world} rank(s), one GPU each: subtrees below a cut of the guide tree owned by one rank (no exchange there), one exchange at the cut, "
                                                  f"above it the pairs of each level dealt to the ranks and the final paths all-gathered HBM to HBM over RCCL (one collective per level)") if world > 1 else "1 GPU"})
            if sharded:
                out["config"]["collective"] = native_note or ("the library's own RCCL communicator (twl_msa_shard_rccl)" if native else "torch.distributed callbacks")
            out["dp_kernel"] = {"cells_per_s": cells / (kernel_ms * 1e-3), "kernel_ms_per_pass": kernel_ms / steps, "exchange_ms_per_pass": exch_ms / steps,
                                "share_of_step_time": (kernel_ms / steps) / (dt_max * 1e3 / steps),
                                "note": "all DP launches of a pass (HIP events on the library's stream; with several ranks the slowest rank of each level)"}
            out["levels"] = [{"pairs": int(lv.pairs), "cells": int(lv.band_cells), "kernel_ms": round(lv.kernel_ms, 3), "level_ms": round(lv.level_ms, 3),
                              **({"tiles_predicted": int(lv.mt_tiles_predicted), "tiles_inline": int(lv.mt_tiles_inline)} if int(lv.speculative) == 3 else {})} for lv in levels]
            tp = sum(int(lv.mt_tiles_predicted) for lv in levels)
            ti = sum(int(lv.mt_tiles_inline) for lv in levels)
            out["tile_parallel"] = {"levels": sum(1 for lv in levels if int(lv.speculative) == 3), "tiles_predicted": tp, "tiles_inline": ti,
                                    "hit_rate": tp / max(1, tp + ti),
                                    "note": "levels with few pairs: every tile of every pair runs at once from a predicted start cell; a tile whose true start differs is recomputed in line (last report of the timed passes)"}
            by = {}
            for lv in levels:
                v = by.setdefault(lv.kernel.decode() or "?", [0, 0.0, 0])      # the kernel name comes from the library (twl_stats.kernel)
                v[0] += 1; v[1] += lv.kernel_ms; v[2] += lv.band_cells
            kernels = [{"kernel": k, "launches": v[0], "avg_ms": v[1] / v[0], "cells_per_launch": v[2] // v[0],
                        "frac": (v[2] * bcell / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS) if v[1] > 0 else 0.0} for k, v in by.items()]
            dom = max(kernels, key=lambda x: x["launches"] * x["avg_ms"]) if kernels else None
        else:
            out["config"].update(peak)
            kernels, dom = [], None
        traffic = issue = None
        pmc_ok, pmc_note, pmc_tb_share = None, None, None
        try:
            PMC = pmc_path(args.config, args.workload)
            pmc = json.load(open(PMC))
            lib_hash = (twl.version().split("src ")[-1] if hasattr(twl, "version") else "")
            pmc_ok = (pmc.get("source_hash") == lib_hash) and args.workload == "calibrated"
            if not pmc_ok:
                pmc_note = (f"counters withheld: profiles/r06/<config>_pmc_summary.json was taken on kernel sources {pmc.get('source_hash')} for the default configuration, "
                            f"this run is {lib_hash} / {args.config} / {args.workload}")
            else:
                pmc_note = ("profiles/r06/<config>_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on the same kernel sources, bytes per pass; "
                            "FETCH_SIZE doubled per MI355X_MICROARCH.md)")
                traffic = pmc["hbm_per_pass"]["traffic_bytes"]
                pmc_tb_share = pmc["hbm_per_pass"]["write_bytes"] / pmc["hbm_per_pass"]["traffic_bytes"]
                # per kernel of the profiled run: VALU issue slots used (a wave's VALU instruction holds its SIMD-32 for 2 cycles), scalar instructions per
                # cycle and CU (the scalar unit retires ~1), wave-cycles spent waiting, and the HBM rate the counters saw
                issue = {"source": "profiles/r06/<config>_pmc_summary.json (tools/summarize_pmc.py: rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAIT_ANY / SQ_WAVE_CYCLES / "
                                   "FETCH_SIZE / WRITE_SIZE passes and the kernel trace of this command)",
                         "ceiling": "issue_frac_of_measured_ceiling = instructions of every kind issued per ns and CU / 3.97 (the measured ceiling); valu_issue_frac = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x 2.4 GHz x kernel time), the figure VERDICT round 4 asked for (a vector instruction holds its SIMD at least 2 cycles; compares, selects and integer forms hold it ~4.5)",
                         "kernels": [{"kernel": k.replace("void twl::", "").replace("(twl::NArgs)", ""), **{f: round(float(v[f]), 4) for f in
                                      ("issue_frac_of_measured_ceiling", "insts_per_ns_per_cu", "valu_issue_frac", "salu_per_cycle_per_cu", "wait_frac_of_wave_cycles", "active_inst_frac_of_wave_cycles", "hbm_gb_per_s", "seconds_in_run", "dispatches") if f in v}}
                                     for k, v in sorted(pmc.get("issue_per_kernel", {}).items(), key=lambda kv: -kv[1].get("seconds_in_run", 0.0))]}
        except Exception as ex:  # noqa: BLE001
            pmc_note = f"no counters: {ex}"
        # compulsory HBM traffic of a pass: every pair reads its two packed profiles once and writes its path once ((R+Q)(4P+8) + (R+Q) bytes)
        lib_ver = twl.version() if hasattr(twl, "version") else ""
        # ---- the roofline of this path: INSTRUCTION ISSUE ----
        # A CU issues at most ~3.97 instructions per ns over its four SIMDs whatever their kind (tools/micro/issue_rates.hip, profiles/r02/issue_rates.log:
        # 16 waves of interleaved vector and scalar instructions; vector alone 3.88, scalar alone 2.27) -- measured, so no clock enters.  One 64-row block on
        # one anti-diagonal costs a wave a fixed number of instructions, counted in the disassembly (profiles/r06/isa_block_step.json + the listings next to
        # it).  peak = the cells/s at which every issue slot of the chip would carry a block-step instruction (of the dominant kernel's step) and every lane a band cell;
        # achieved = band cells of a pass / DP-kernel time of a pass (HIP events of the library around every DP launch, live in this run; the rocprofv3 trace of the
        # same command gives the same time: sum of the talco_* rows of profiles/r06/bench_kernel_stats.csv / passes);
        # frac = achieved / peak <= 1.  What separates them: lanes of a block outside the band (~16 %), the per-diagonal bookkeeping every wave runs
        # (~60 instructions: barrier, band update), idle issue slots while a workgroup's waves wait for each other.
        isa, isa_note, peak_cells, step_ins, ceiling = None, None, None, None, None
        step_units, valu_cap, peak_issue = None, None, None
        try:
            isa = json.load(open(ISA))
            ceiling = isa["issue_ceiling"]["instr_per_ns_per_cu"] * 1e9 * num_cu
            valu_cap = isa["valu_ceiling"]["units_per_ns_per_cu"] * 1e9 * num_cu
            if isa.get("source_hash") != (lib_ver.split("src ")[-1] if lib_ver else ""):
                isa_note = f"static counts are of kernel sources {isa.get('source_hash')}, this library is {lib_ver}"
            dk = dom["kernel"] if dom else (peak or {}).get("kernel")
            if dk:
                key = isa_key(isa, dk)
                if key is None:
                    isa_note = (isa_note + "; " if isa_note else "") + f"no static count for {dk} in profiles/r06/isa_block_step.json: frac not formed"
                if key:
                    # (one-letter query rows with single-sequence references have neither gap letters nor a denominator: the leaf level; everything else pays both)
                    which = "least" if ", 5, 5, false" in key else "all_paths"
                    s0 = isa["kernels"][key]["phase_A"]["slot0"][which]
                    step_ins, step_units = s0["instructions"], s0["valu_units"]
                    peak_cells = valu_cap * 64.0 / step_units
                    peak_issue = ceiling * 64.0 / step_ins
        except Exception as ex:  # noqa: BLE001
            isa_note = f"no static counts: {ex}"
        dom_cells_per_s = (dom["cells_per_launch"] / (dom["avg_ms"] * 1e-3)) if dom and dom["avg_ms"] > 0 else None
        out["roofline"] = {
            "bound": "valu", "achieved": (cells / (kernel_ms * 1e-3)) if kernel_ms > 0 else None, "peak": peak_cells, "unit": "cells/s",
            "frac": (cells / (kernel_ms * 1e-3) / peak_cells) if (kernel_ms > 0 and peak_cells) else None,
            "frac_levels_of_dominant_kernel": (dom_cells_per_s / peak_cells) if (dom_cells_per_s and peak_cells) else None,
            "valu_ceiling_units_per_s": valu_cap, "valu_units_per_block_step": step_units, "cells_per_block_step": 64, "static_counts": "profiles/r06/isa_block_step.json",
            # round 5's definition, kept beside it: every issue slot of the chip (3.97 instructions per ns and CU of any kind) carrying an instruction of the block step
            "frac_total_issue": (cells / (kernel_ms * 1e-3) / peak_issue) if (kernel_ms > 0 and peak_issue) else None,
            "issue_ceiling_instr_per_s": ceiling, "instructions_per_block_step": step_ins,
            "static_counts_note": isa_note,
            # fp32 VALU utilisation = F_cell x band cells / DP-kernel time against the fp32 vector peak (FMA counted as 2: bit-exactness forbids
            # FMA here -- TALCO-XDrop.cpp:378-395 rounds every product and sum -- so half of that peak is out of reach by construction)
            "valu_flops_frac": (F_CELL[cfg["P"]] * cells / (kernel_ms * 1e-3) / 1e12 / VALU_FP32_PEAK_TFLOPS) if kernel_ms > 0 else None,
            "valu_tflops": (F_CELL[cfg["P"]] * cells / (kernel_ms * 1e-3) / 1e12) if kernel_ms > 0 else None,
            "valu_flops_per_cell": F_CELL[cfg["P"]], "valu_fp32_peak_tflops": VALU_FP32_PEAK_TFLOPS,
            # ... and the classic figure: NOMINAL R x Q cells (whole matrices, not bands) per second, over the DP kernels and over the timed region
            "nominal_gcups_dp_kernels": (nominal_cells / (kernel_ms * 1e-3) / 1e9) if (kernel_ms > 0 and nominal_cells and world == 1) else None,
            "nominal_gcups_whole_job": (nominal_cells / dt_max / 1e9) if (nominal_cells and world == 1) else None,
            "nominal_cells": (int(nominal_cells // steps) if (nominal_cells and world == 1) else None),
            "contract_bound": "hbm", "contract_achieved_gb_s": achieved, "contract_peak_gb_s": HBM_PEAK_GBS, "contract_frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_live": False,
            "traffic_bytes_per_cell": (traffic / (cells / steps)) if traffic else None,
            "traffic_frac_of_hbm_peak": (traffic / (kernel_ms / steps * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "traceback_share_of_traffic": pmc_tb_share,
            "compulsory_bytes_per_cell": 0.07, "compulsory_bytes_frac": (0.07 * (cells / steps) / (kernel_ms / steps * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernel_ms > 0 else None,
            "traffic_source": pmc_note,
            "library": lib_ver,
            "issue": issue,
            "algorithmic_bytes_per_cell": bcell, "cells": int(cells // steps), "kernel_ms": kernel_ms / steps,
            "dominant_kernel": dom, "kernels": kernels,
            "note": "what binds this path is the VECTOR unit (round 6: tools/micro/issue_rates5.hip -- an fp32 add / mul / fma on vector registers issues at 3.6-3.8 per ns and CU, every "
                    "integer, compare, select, DPP or scalar-operand form at 2.15-2.27, packed fp32 at 1.85; 70 % of the step is of the slow kinds, and weighted that way the vector unit of a "
                    "CU is ~90 % busy in the throughput kernels), so `frac` = band cells of a pass / DP-kernel time of a pass / (256 CUs x 3.7e9 full-rate vector instructions/s x 64 cells / "
                    "VALU units of the dominant kernel's block step, counted in its disassembly with those weights); `frac_total_issue` = round 5's definition (3.97e9 instructions of any kind, "
                    "instructions of the block step); `frac_levels_of_dominant_kernel` = `frac` for the levels that start on the dominant kernel alone.  `contract_frac` keeps the figure of BASELINE.md section 3 -- band "
                    "cells x 64 (192) operand bytes / DP-kernel time over ALL launches of a pass against 8 TB/s; that operand stream is notional (columns are reused from LDS / registers), "
                    "so it exceeds 1 and measures nothing.  Real HBM traffic (`traffic`, counters) is ~1.1 B per cell, most of it traceback words; compulsory 0.07 B per cell.  "
                    "`issue` = the counters of the profiled run per kernel.  DESIGN.md section 3",
        }
        if world == 1 and not args.no_peak and family:
            try:
                peak, pk_batch, matrix, pk, gp, gl = peak_level(twl, dev, local_rank, cfg)
                out["peak_level"] = peak
            except Exception as e:  # noqa: BLE001
                out["peak_level"] = {"note": f"failed: {e}"}
        if world == 1 and not args.no_cpu:
            try:
                if pk_batch is None:
                    peak2, pk_batch, matrix, pk, gp, gl = peak_level(twl, dev, local_rank, cfg, pairs=64, reps=1)
                out["cpu_baseline"] = cpu_baseline(pk_batch, matrix, pk, gp, gl)
            except Exception as e:  # the checker must never take the bench line down
                out["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        if e2e is not None:
            out["wallclock_to_msa"] = e2e
        if world == 1 and family and args.config == "rnasim10k" and args.workload == "calibrated" and not args.no_survey8d and not args.leaves and not args.length:
            try:
                out["survey8d"] = survey8d_record(cfg, local_rank, base)
            except Exception as e:  # noqa: BLE001
                out["survey8d"] = {"note": f"failed: {e}"}
            # ... and its headline figures where a reader of `config` sees them
            s8 = out["survey8d"]
            if "cells_per_s" in s8:
                out["config"]["survey8d_family_as_written"] = {"cells_per_s": s8["cells_per_s"], "s_per_pass": s8["s_per_pass"], "contract_frac": s8["frac_of_hbm_roofline"],
                                                               "msa_md5": s8["msa_md5"], "note": "one pass over the family generated with SURVEY.md 8d's parameters (record `survey8d` of this line)"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        if family and not args.keep:
            for f in (tree, fasta, os.path.join(base, "ready")):
                try:
                    os.remove(f)
                except OSError:
                    pass
            try:
                os.rmdir(base)
            except OSError:
                pass

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    twl.shutdown()


if __name__ == "__main__":
    main()
