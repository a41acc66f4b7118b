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


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,length", [("n", 40, 1200), ("p", 24, 500)])
@pytest.mark.timeout(900)
def test_product_equals_oracle_on_synthetic_family(built, tmp_path, kind, n, length):
    """No reference output is on record for proteins: here the product CLI (GPU) must equal the CPU checker end to end."""
    from twilight_amd import synth

    nwk, seqs = synth.make_family(n, length, P=(6 if kind == "n" else 22), seed=5 + n)
    (tmp_path / "t.nwk").write_text(nwk + "\n")
    (tmp_path / "s.fa").write_text("".join(f">{name}\n{seq}\n" for name, seq in seqs))
    outs = {}
    for tag, exe in (("oracle", os.path.join(ROOT, "oracle", "e2e_oracle")), ("gpu", os.path.join(ROOT, "twilight_amd", "twilight-mi355x"))):
        out = tmp_path / f"{tag}.aln"
        r = subprocess.run([exe, "-t", str(tmp_path / "t.nwk"), "-i", str(tmp_path / "s.fa"), "-o", str(out), "--type", kind, "--check"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "illegal alignment" not in r.stderr
        outs[tag] = out
    assert _md5(outs["gpu"]) == _md5(outs["oracle"])
    names, rows = _rows(outs["gpu"])
    assert len(names) == n and len(set(len(x) for x in rows)) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("kind,flags", [
    ("n", ["-r", "0.7"]),                                   # more columns count as gappy: runs on both sides, pairwiseGlobal splices
    ("n", ["-r", "1"]),                                     # gappy-column removal off (alignment-helper.cpp:77)
    ("n", ["-w", "--gap-open", "-30", "--gap-extend", "-3"]),   # wildcard matrix (general 5x5 kernel mode), other gap penalties / X-drop
    ("n", ["--length-deviation", "0.004"]),                 # some sequences are deferred: main pass on the device, deferred pass after the hand-over
    ("n", ["--length-deviation", "0.004", "--filter"]),     # the same sequences are excluded instead: empty sides (alignment-cpu.cpp:89-90)
    ("p", ["-r", "0.8", "--gap-open", "-40"]),
    ("p", ["--length-deviation", "0.01", "-w"]),
])
@pytest.mark.timeout(900)
def test_product_equals_oracle_with_cli_variants(built, tmp_path, kind, flags):
    """Device-resident CLI path vs the CPU checker under non-default options; plus resident == host-staged."""
    from twilight_amd import synth

    n, length = (48, 900) if kind == "n" else (28, 400)
    nwk, seqs = synth.make_family(n, length, P=(6 if kind == "n" else 22), seed=77, sub=0.03, indel=0.006)
    (tmp_path / "t.nwk").write_text(nwk + "\n")
    (tmp_path / "s.fa").write_text("".join(f">{name}\n{seq}\n" for name, seq in seqs))
    outs = {}
    runs = (("oracle", os.path.join(ROOT, "oracle", "e2e_oracle"), []), ("gpu", os.path.join(ROOT, "twilight_amd", "twilight-mi355x"), []),
            ("staged", os.path.join(ROOT, "twilight_amd", "twilight-mi355x"), ["--host-staged"]))
    for tag, exe, extra in runs:
        out = tmp_path / f"{tag}.aln"
        r = subprocess.run([exe, "-t", str(tmp_path / "t.nwk"), "-i", str(tmp_path / "s.fa"), "-o", str(out), "--type", kind, "--check"] + flags + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "illegal alignment" not in r.stderr
        outs[tag] = (out, r.stderr)
    assert _md5(outs["gpu"][0]) == _md5(outs["oracle"][0]), flags
    assert _md5(outs["staged"][0]) == _md5(outs["oracle"][0]), flags
    if "--length-deviation" in flags and "--filter" not in flags:
        assert "Realign profiles that have been deferred" in outs["gpu"][1], "the variant was meant to exercise the deferred pass"


@pytest.mark.gpu
@pytest.mark.parametrize("replicas,kind,flags,th", [(2, "n", [], 0), (3, "n", ["--length-deviation", "0.004"], 0), (2, "p", ["-r", "0.8"], 0), (3, "n", [], 6)])
@pytest.mark.timeout(900)
def test_replicated_resident_path_equals_oracle(built, tmp_path, replicas, kind, flags, th):
    """Several device replicas (here: virtual ones on the first GPU): every replica prepares and commits the whole level, the DP is
    sharded over them by mask, paths meet on the host.  The MSA must equal the CPU checker's."""
    from twilight_amd import synth

    n, length = (56, 800) if kind == "n" else (30, 400)
    nwk, seqs = synth.make_family(n, length, P=(6 if kind == "n" else 22), seed=91, sub=0.04, indel=0.004)
    (tmp_path / "t.nwk").write_text(nwk + "\n")
    (tmp_path / "s.fa").write_text("".join(f">{name}\n{seq}\n" for name, seq in seqs))
    if th:
        flags = flags + ["--test-cal-profile-th", str(th), "--test-update-seq-th", str(th)]
    outs = {}
    for tag, exe, extra in (("oracle", os.path.join(ROOT, "oracle", "e2e_oracle"), []),
                            ("gpu", os.path.join(ROOT, "twilight_amd", "twilight-mi355x"), ["--test-virtual-devices", str(replicas)])):
        out = tmp_path / f"{tag}.aln"
        r = subprocess.run([exe, "-t", str(tmp_path / "t.nwk"), "-i", str(tmp_path / "s.fa"), "-o", str(out), "--type", kind, "--check", "-v"] + flags + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = (out, r.stderr)
    assert f"resident on {replicas} device replica(s)" in outs["gpu"][1]
    assert _md5(outs["gpu"][0]) == _md5(outs["oracle"][0])
