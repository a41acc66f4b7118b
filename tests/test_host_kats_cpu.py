"""Known-answer tests of the host mirror's helpers (C++), written from the reference source; runs without a GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "twilight_amd", "csrc", "host")


def test_host_helper_known_answers(tmp_path):
    exe = tmp_path / "host_kats"
    srcs = [os.path.join(HOST, f) for f in ("phylo.cpp", "seqdb_io.cpp", "helpers.cpp", "progressive.cpp", "driver.cpp")]
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tests", "host_kats.cpp")] + srcs + ["-lz"])
    r = subprocess.run([str(exe), str(tmp_path), os.path.join(ROOT, "tests", "golden", "substitution.txt")], capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith(("OK", "FAIL"))]
    failed = [l for l in lines if l.startswith("FAIL")]
    assert not failed and r.returncode == 0, r.stdout + r.stderr
    assert len(lines) >= 19
    # every built-in protein matrix equals 5 x the reference's table (fixture: tests/golden/blosum_tables.json, made by make_blosum_fixture.py)
    import json

    import numpy as np

    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "blosum_tables.json")))
    got = np.loadtxt(tmp_path / "blosum_dump.txt").reshape(3, 20, 20)
    for k, b in enumerate(("45", "62", "80")):
        assert np.array_equal(got[k], 5.0 * np.array(ref[b], dtype=np.float64)), b
    assert got[2][7][17] == 15 and got[2][17][7] == 5
