"""oracle/level_oracle.py (numpy) against the C++ host mirror on the same inputs, bit for bit.  No GPU.

The C++ mirror (twilight_amd/csrc/host/helpers.cpp) is what the end-to-end pins validate against the reference's MSAs; this test
transfers that pin to the numpy restatement that the GPU tests of the device-resident level kernels use as their checker.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "twilight_amd", "csrc", "host")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import level_oracle as LO  # noqa: E402
from twilight_amd import synth  # noqa: E402
import level_cases as LC  # noqa: E402


@pytest.fixture(scope="module")
def dump_exe(tmp_path_factory):
    exe = tmp_path_factory.mktemp("lvl") / "host_level_dump"
    srcs = [os.path.join(HOST, f) for f in ("phylo.cpp", "seqdb_io.cpp", "helpers.cpp", "progressive.cpp", "driver.cpp")]
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tests", "host_level_dump.cpp")] + srcs + ["-lz"])
    return str(exe)


def _hex(a):
    return " ".join(f"{int(x):08x}" for x in np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).ravel())


def _run_case(exe, tmp_path, case, tag):
    P = case.P
    f = tmp_path / f"{tag}.txt"
    with open(f, "w") as out:
        out.write(f"type {case.seq_type}\nthr {case.thr:.9g}\n")
        for sd in range(2):
            s = case.sides[sd]
            out.write(f"side {sd} {s.group_weight:.9g} {len(s.rows)} {1 if s.cache is not None else 0}\n")
            for w, row in zip(s.seq_weights, s.rows):
                out.write(f"{w:.9g} {row.decode()}\n")
            if s.cache is not None:
                out.write(_hex(s.cache) + "\n")
        exp = LC.expected(case)
        out.write("path " + ("".join(str(int(c)) for c in exp["path_wo_gc"]) or "-") + "\n")
    r = subprocess.run([exe, str(f)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {}
    for line in r.stdout.splitlines():
        k, _, rest = line.partition(" ")
        got.setdefault(k, []).append(rest)
    assert got["lens"][0] == f"{exp['lens'][0]} {exp['lens'][1]}"
    for sd in range(2):
        words = got["cols"][sd].split()[1:]
        assert " ".join(words) == _hex(exp["cols"][sd]), f"{tag}: packed columns of side {sd} differ"
        runs = " ".join(f"{a},{b}" for a, b in exp["runs"][sd])
        assert got["runs"][sd].split(" ", 1)[1:] == ([runs] if runs else []), f"{tag}: gappy runs of side {sd}"
        letters = (LO.NUC + "N") if case.seq_type == "n" else (LO.AA + "X")
        assert got["cons"][sd].split()[1] == "".join(letters[i] for i in exp["cons"][sd])
        if exp["cache_after_prepare"][sd] is not None:
            assert " ".join(got["cache"][sd].split()[1:]) == _hex(exp["cache_after_prepare"][sd]), f"{tag}: stored cache of side {sd}"
    assert got["full"][0] == "".join(str(int(c)) for c in exp["path_full"])
    rows = {int(x.split()[0]): x.split()[1] for x in got["row"]}
    for i, row in enumerate(exp["rows_after"]):
        assert rows[i] == row.decode(), f"{tag}: row {i} after write-back"
    if exp["merged"] is not None:
        assert " ".join(got["merged"][0].split()) == _hex(exp["merged"]), f"{tag}: merged cache"


@pytest.mark.parametrize("seq_type", ["n", "p"])
def test_numpy_restatement_matches_host_mirror(dump_exe, tmp_path, seq_type):
    for seed in range(6):
        case = LC.make_case(seq_type, seed, cached=(seed % 3), length=60 + 17 * seed)
        _run_case(dump_exe, tmp_path, case, f"{seq_type}{seed}")


def test_matrix_tables_agree():
    # the restatement's pairwiseGlobal indexes the same matrices the product uses
    assert synth.nucleotide_matrix().shape == (5, 5) and synth.protein_matrix().shape == (21, 21)
