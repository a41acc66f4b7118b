"""Re-derive tests/golden/e2e_expected.json with the CPU checker and compare (run from the repo root).

The expected values themselves come from the reference; this script shows how they are
reproduced here and refuses to overwrite the file with anything that differs.
"""
import hashlib, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = os.path.join(ROOT, "tests", "golden")
exp = json.load(open(os.path.join(G, "e2e_expected.json")))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "e2e_oracle"], stdout=subprocess.DEVNULL)
ok = True
for name in ("sars_20", "RNASim"):
    e = exp[name]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "out.aln")
        r = subprocess.run([os.path.join(ROOT, "oracle", "e2e_oracle"), "-t", os.path.join(G, e["tree"]), "-i", os.path.join(G, e["sequences"]), "-o", out],
                           capture_output=True, text=True, check=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("E2E")][0]
        got = dict(kv.split("=") for kv in line.split()[1:])
        md5 = hashlib.md5(open(out, "rb").read()).hexdigest()
    same = (md5 == e["md5"] and int(got["band_cells"]) == e["band_cells"] and int(got["aln_len"]) == e["aln_len"]
            and [int(x) for x in got["pairs_per_level"].split("/")] == e["pairs_per_level"] and int(got["max_width"]) == e["max_width"])
    print(name, "md5", md5, "cells", got["band_cells"], "OK" if same else "MISMATCH")
    ok &= same
sys.exit(0 if ok else 1)
