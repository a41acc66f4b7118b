"""BASELINE configurations 3, 4 and 5 at full size on the GPU, against what the CPU checker produced on the same seeded families
(tests/golden/e2e_synthetic_expected.json, written by tests/golden/make_e2e_synthetic.py: MSA md5, length, band-cell total, pairs per level).
Config 4 (100 000 x 1.6 kbp, "8 GPUs") also runs with 8 virtual device replicas on the one GPU: the multi-device orchestration at full
size, which must give the byte-identical MSA."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP_PATH = os.path.join(ROOT, "tests", "golden", "e2e_synthetic_expected.json")
EXP = json.load(open(EXP_PATH)) if os.path.exists(EXP_PATH) else {}
pytestmark = pytest.mark.gpu


def _family(tmp, f):
    from twilight_amd import synth

    sys.setrecursionlimit(1000000)
    nwk, seqs = synth.make_family(f["leaves"], f["length"], P=f["P"], seed=f["seed"], indel=f["indel"],
                                  **({"sub_range": tuple(f["sub_range"])} if f.get("sub_range") else {"sub": f["sub"]}))
    t, fa = os.path.join(tmp, "t.nwk"), os.path.join(tmp, "s.fa")
    open(t, "w").write(nwk + "\n")
    with open(fa, "w") as fh:
        for n, s in seqs:
            fh.write(f">{n}\n{s}\n")
    return t, fa


def _run(tree, fasta, out, typ, extra=()):
    r = subprocess.run([os.path.join(ROOT, "twilight_amd", "twilight-mi355x"), "-t", tree, "-i", fasta, "-o", out, "--type", typ, "--check", "-v"] + list(extra),
                       capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "illegal alignment" not in r.stderr                      # --check: every row reproduces its input, all rows equally long
    tail = [l for l in r.stderr.splitlines() if l.startswith("Wrote")][-1]
    cells = int(tail.split(" band cells")[0].split()[-1])
    levels = [int(l.split("aligned ")[1].split()[0]) for l in r.stderr.splitlines() if l.startswith("Level ")]
    md5 = hashlib.md5(open(out, "rb").read()).hexdigest()
    os.remove(out)
    return md5, cells, levels


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("name", ["rnasim10k", "protein5k", "rnasim10k_survey8d"])
def test_full_size_configuration_reproduces_the_cpu_checker(built, tmp_path, name):
    if name not in EXP:
        pytest.skip("fixture not generated")
    f = EXP[name]
    tree, fasta = _family(str(tmp_path), f)
    md5, cells, levels = _run(tree, fasta, str(tmp_path / "o.aln"), f["type"])
    assert levels == f["pairs_per_level"]
    assert cells == f["band_cells"]
    assert md5 == f["md5"]


@pytest.mark.timeout(1800)
def test_config4_100k_sequences_one_and_eight_replicas(built, tmp_path):
    if "rnasim100k" not in EXP:
        pytest.skip("fixture not generated")
    f = EXP["rnasim100k"]
    tree, fasta = _family(str(tmp_path), f)
    one = _run(tree, fasta, str(tmp_path / "o1.aln"), "n", ["--test-virtual-devices", "1"])
    eight = _run(tree, fasta, str(tmp_path / "o8.aln"), "n", ["--test-virtual-devices", "8"])
    assert one[0] == eight[0] == f["md5"]
    assert one[2] == eight[2] == f["pairs_per_level"]
    assert one[1] == f["band_cells"]          # (virtual replicas share the device's counters; the single-replica run gives the total)
