"""End-to-end pin of the oracle (and, with -m gpu, of the product) against the reference's recorded outputs.

Inputs: the reference's own sample data (tests/golden/*.fa.gz, *.nwk).  Expected: MSA md5 / dimensions / band-cell totals /
pairs per level that BASELINE.md section 2 records for the reference CPU path.  The CPU leg runs oracle/e2e_oracle (host mirror +
oracle DP); the GPU leg runs the product CLI twilight-mi355x (host mirror + libtwl_align through the C ABI).
"""
import hashlib
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
EXP = json.load(open(os.path.join(G, "e2e_expected.json")))


def _md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def _rows(path):
    names, rows = [], []
    for line in open(path):
        if line.startswith(">"):
            names.append(line[1:].strip())
        else:
            rows.append(line.rstrip("\n"))
    return names, rows


@pytest.mark.parametrize("name", ["sars_20", "RNASim"])
@pytest.mark.timeout(600)
def test_oracle_reproduces_reference_msa(built, tmp_path, name):
    e = EXP[name]
    out = tmp_path / "out.aln"
    r = subprocess.run([os.path.join(ROOT, "oracle", "e2e_oracle"), "-t", os.path.join(G, e["tree"]), "-i", os.path.join(G, e["sequences"]),
                        "-o", str(out), "--check"], capture_output=True, text=True, check=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("E2E")][0]
    got = dict(kv.split("=") for kv in line.split()[1:])
    names, rows = _rows(out)
    assert len(names) == e["n_seqs"] and all(len(x) == e["aln_len"] for x in rows)
    assert [int(x) for x in got["pairs_per_level"].split("/")] == e["pairs_per_level"]
    assert int(got["band_cells"]) == e["band_cells"]          # pins band evolution and tile boundaries, not just the final rows
    assert int(got["max_width"]) == e["max_width"]
    assert _md5(out) == e["md5"]
    assert "illegal alignment" not in r.stderr                # --check: every row reproduces its input sequence


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sars_20", "RNASim"])
@pytest.mark.timeout(900)
def test_product_cli_reproduces_reference_msa_on_gpu(built, tmp_path, name):
    e = EXP[name]
    out = tmp_path / "out.aln"
    exe = os.path.join(ROOT, "twilight_amd", "twilight-mi355x")
    r = subprocess.run([exe, "-t", os.path.join(G, e["tree"]), "-i", os.path.join(G, e["sequences"]), "-o", str(out), "--check", "-v"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    names, rows = _rows(out)
    assert len(names) == e["n_seqs"] and all(len(x) == e["aln_len"] for x in rows)
    assert _md5(out) == e["md5"]
    tail = [l for l in r.stderr.splitlines() if "band cells" in l][-1]
    cells = int(tail.split(" band cells")[0].split()[-1])
    assert cells == e["band_cells"]
