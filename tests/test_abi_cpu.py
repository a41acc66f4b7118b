"""The C-ABI library builds for gfx950, loads, and exports every symbol include/twl_align.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="twl_align.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(twl_[a-z_]+)\s*\(", text)))


def test_header_symbols_are_exported(built):
    import twilight_amd as twl

    lib = twl.load_library()
    declared = _declared_symbols()
    assert set(declared) == set(twl.exported_symbols())
    for name in declared:
        assert getattr(lib, name) is not None, name


def test_level_header_symbols_are_exported(built):
    import twilight_amd as twl
    from twilight_amd import level

    lib = twl.load_library()
    declared = _declared_symbols("twl_level.h")
    assert set(declared) == set(level.exported_symbols())
    for name in declared:
        assert getattr(lib, name) is not None, name
    assert C.sizeof(level.TwlSide) == 8 * 4


def test_host_library_header_symbols_are_exported(built):
    """libtwl_host.so (include/twl_msa.h): the caller of the hot path as a C ABI; loads without a GPU and exports every entry point."""
    from twilight_amd import msa

    lib = msa.load_library()
    declared = _declared_symbols("twl_msa.h")
    assert set(declared) == set(msa.exported_symbols())
    for name in declared:
        assert getattr(lib, name) is not None, name
    assert C.sizeof(msa.MsaLevel) == 2 * 4 + 2 * 8 + 3 * 8 + 4 * 4 + 160
    assert C.sizeof(msa.MsaTotals) == 4 * 4 + 3 * 8 + 3 * 8 + 8      # (+ nominal_cells, round 6)


def test_struct_layouts_match_header(built):
    import twilight_amd as twl

    # twl_params: int32 + 441 floats + 4 floats + 3 int32; twl_stats: 2 u64 + 3 double + 10 int32
    assert C.sizeof(twl.TwlParams) == 4 + 441 * 4 + 4 * 4 + 3 * 4
    assert C.sizeof(twl.TwlStats) == 2 * 8 + 3 * 8 + 10 * 4 + 160


def test_calls_fail_loudly_without_init_or_gpu(built):
    import numpy as np

    import twilight_amd as twl
    from twilight_amd import synth

    lib = twl.load_library()
    st = twl.TwlStats()
    # not initialised -> explicit error code, never a silent CPU result
    rc = lib.twl_get_stats(C.c_int(0), C.byref(st))
    if rc == 0:
        pytest.skip("library already initialised in this process")
    assert rc == -1
    assert b"twl_init" in lib.twl_last_error()
    import torch

    if not torch.cuda.is_available():
        batch = synth.make_level_batch(1, 20, members=(1, 1), seed=1)
        with pytest.raises(twl.TwlError):
            twl.init([0])
        with pytest.raises(twl.TwlError):
            twl.align_batch(twl.make_params(synth.nucleotide_matrix()), batch)


def test_product_does_not_reference_oracle():
    # the shipped path must never import/link the CPU checker
    for dirpath, _, files in os.walk(os.path.join(ROOT, "twilight_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in text and "libtwl_oracle" not in text and "talco_oracle" not in text, f


def test_forked_cli_leaves_nobody_behind_when_a_rank_fails(built, tmp_path):
    """twilight-mi355x --gpu-index a,b,c forks one process per GPU before anything touches a device.  On a box without those devices rank 0 fails in
    twl_init -- the children, which wait on the shared page for the communicator id, must end too (they used to spin for ever)."""
    import gzip
    import signal
    import subprocess
    import time

    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 3:
        pytest.skip("three devices present: the ranks would come up")
    fa = tmp_path / "s.fa"
    fa.write_bytes(gzip.open(os.path.join(ROOT, "tests", "golden", "sars_20.fa.gz")).read())
    exe = os.path.join(ROOT, "twilight_amd", "twilight-mi355x")
    p = subprocess.Popen([exe, "-t", os.path.join(ROOT, "tests", "golden", "sars_20.nwk"), "-i", str(fa), "-o", str(tmp_path / "o.aln"), "--gpu-index", "0,1,2"],
                         stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
    try:
        _, err = p.communicate(timeout=120)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        raise
    assert p.returncode != 0, err
    deadline = time.time() + 10
    while time.time() < deadline:
        try:
            os.killpg(p.pid, 0)          # anybody left in the session's process group?
        except ProcessLookupError:
            break
        time.sleep(0.05)
    else:
        os.killpg(p.pid, signal.SIGKILL)
        raise AssertionError("children of a failed run are still alive")


def _nobody_left(p, grace=10.0):
    import signal
    import time

    deadline = time.time() + grace
    while time.time() < deadline:
        try:
            os.killpg(p.pid, 0)
        except ProcessLookupError:
            return
        time.sleep(0.05)
    os.killpg(p.pid, signal.SIGKILL)
    raise AssertionError("processes of the run are still alive")


@pytest.mark.timeout(300)
@pytest.mark.parametrize("ranks", [2, 3])
def test_forked_cli_aligns_on_the_cpu_check_build(built, tmp_path, ranks):
    """The product's CLI flow for several GPUs -- fork one process per listed device, hand the communicator id over through the shared page, deal the pairs of every
    level, all-gather the paths through the LIBRARY's collective, rank 0 writes -- on a box without a GPU: oracle/twilight-cpucheck is main.cpp linked against the
    CPU-check host library, whose device is the oracle and whose communicator is a shared-memory segment.  What stays unexecuted without a
    multi-GPU node is RCCL itself and the device-resident kernel's subtree ownership (its level kernels have no CPU form)."""
    import gzip
    import hashlib
    import signal
    import subprocess

    exe = os.path.join(ROOT, "oracle", "twilight-cpucheck")
    assert os.path.exists(exe), "run `make -C oracle` (build() does)"
    fa = tmp_path / "s.fa"
    fa.write_bytes(gzip.open(os.path.join(ROOT, "tests", "golden", "sars_20.fa.gz")).read())
    out = tmp_path / "o.aln"
    p = subprocess.Popen([exe, "-t", os.path.join(ROOT, "tests", "golden", "sars_20.nwk"), "-i", str(fa), "-o", str(out), "--host-staged", "--test-fork-host-staged",
                          "--gpu-index", ",".join(str(r) for r in range(ranks))], stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True, text=True)
    try:
        so, se = p.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        raise
    assert p.returncode == 0, se[-2000:]
    _nobody_left(p)
    md5 = hashlib.md5(out.read_bytes()).hexdigest()
    assert md5.startswith("53ccbd43"), md5                                       # the reference's recorded MSA of sars_20
    assert "468765465 band cells" in se, se[-500:]                               # every pair aligned once, by one rank


@pytest.mark.timeout(120)
def test_forked_cli_ends_when_a_rank_dies_inside_a_collective(built, tmp_path):
    """Rank 1 dies inside its second all-gather; rank 0 is left in the collective, from which no return code ever comes: the watchdog of the CLI must end the run with an
    error and leave nobody behind (ADVICE rounds 4 and 5)."""
    import gzip
    import signal
    import subprocess

    exe = os.path.join(ROOT, "oracle", "twilight-cpucheck")
    fa = tmp_path / "s.fa"
    fa.write_bytes(gzip.open(os.path.join(ROOT, "tests", "golden", "sars_20.fa.gz")).read())
    env = dict(os.environ, TWL_CPUCHECK_DIE_AT="2")
    p = subprocess.Popen([exe, "-t", os.path.join(ROOT, "tests", "golden", "sars_20.nwk"), "-i", str(fa), "-o", str(tmp_path / "o.aln"), "--host-staged",
                          "--test-fork-host-staged", "--gpu-index", "0,1"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True, env=env)
    try:
        _, err = p.communicate(timeout=100)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        raise AssertionError("the run hung after a rank died inside a collective")
    assert p.returncode != 0, err
    _nobody_left(p)
