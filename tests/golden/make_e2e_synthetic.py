"""Writes tests/golden/e2e_synthetic_expected.json: what the CPU checker (oracle/e2e_oracle = host mirror + oracle DP) produces on the
synthetic families of BASELINE configs 3, 4 and 5 (and config 3 with the family parameters of SURVEY.md 8d as written) -- MSA md5, alignment length, band-cell total, pairs per level.  The GPU suite
regenerates the same families (twilight_amd.synth.make_family is seeded) and requires the product to reproduce these numbers.
Takes ~15 minutes of CPU; only md5s and counters are committed, not the MSAs.   python tests/golden/make_e2e_synthetic.py [name ...]"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from twilight_amd import synth  # noqa: E402

FAMILIES = {      # the generator arguments ARE the fixture's inputs: bench.py CONFIGS uses the same ones
    "rnasim10k": dict(leaves=10000, length=10000, P=6, type="n", seed=20260501, sub=0.015, indel=0.001),
    "rnasim100k": dict(leaves=100000, length=1600, P=6, type="n", seed=20260501, sub=0.015, indel=0.001),
    "protein5k": dict(leaves=5000, length=2000, P=22, type="p", seed=20260501, sub=0.015, indel=0.001),
    # SURVEY.md 8d as written (bench.py --workload survey8d): per-branch substitution U(0.03, 0.10), indel 0.005/site, seed 20260501 + config index.
    # ~30 minutes of CPU on 6 threads: 1.8e11 band cells, bands up to 2803 rows, 47 deferred profiles
    "rnasim10k_survey8d": dict(leaves=10000, length=10000, P=6, type="n", seed=20260503, sub=None, indel=0.005, sub_range=[0.03, 0.10]),
}


def main():
    out_path = os.path.join(HERE, "e2e_synthetic_expected.json")
    res = json.load(open(out_path)) if os.path.exists(out_path) else {}
    sys.setrecursionlimit(1000000)
    for name in (sys.argv[1:] or list(FAMILIES)):
        f = FAMILIES[name]
        with tempfile.TemporaryDirectory(prefix="twl_fix_") as d:
            nwk, seqs = synth.make_family(f["leaves"], f["length"], P=f["P"], seed=f["seed"], indel=f["indel"],
                                          **({"sub_range": tuple(f["sub_range"])} if f.get("sub_range") else {"sub": f["sub"]}))
            open(os.path.join(d, "t.nwk"), "w").write(nwk + "\n")
            with open(os.path.join(d, "s.fa"), "w") as fh:
                for n, s in seqs:
                    fh.write(f">{n}\n{s}\n")
            t0 = time.time()
            r = subprocess.run([os.path.join(ROOT, "oracle", "e2e_oracle"), "-t", os.path.join(d, "t.nwk"), "-i", os.path.join(d, "s.fa"), "-o", os.path.join(d, "o.aln"),
                                "--type", f["type"], "--check"], capture_output=True, text=True, check=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("E2E")][-1]
            kv = dict(x.split("=") for x in line.split()[1:])
            assert "illegal alignment" not in r.stderr
            res[name] = dict(f, md5=hashlib.md5(open(os.path.join(d, "o.aln"), "rb").read()).hexdigest(), aln_len=int(kv["aln_len"]),
                             band_cells=int(kv["band_cells"]), pairs_per_level=[int(x) for x in kv["pairs_per_level"].split("/")], max_width=int(kv["max_width"]),
                             cpu_checker_seconds=round(time.time() - t0, 1), cpu_threads=os.cpu_count())
            print(name, res[name]["md5"], res[name]["band_cells"], res[name]["cpu_checker_seconds"], flush=True)
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
