"""Writes tests/golden/e2e_variants.json from the INDEPENDENT replay (oracle/msa_replay.py on the dump of oracle/schedule_dump): MSA md5,
alignment length, band cells and pairs per level of every variant in tests/variants.py.  Both the CPU checker (tests/test_replay_cpu.py)
and the product CLI on the GPU (tests/test_gpu_variants.py) must reproduce these.     python tests/golden/make_e2e_variants.py"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from variants import VARIANTS, write_family  # noqa: E402

OUT = os.path.join(HERE, "e2e_variants.json")
only = set(sys.argv[1:])      # names on the command line: regenerate only those, keep the other entries
out = json.load(open(OUT)) if (only and os.path.exists(OUT)) else {}
for name, fam, ins, flags, env in VARIANTS:
    if only and name not in only:
        continue
    with tempfile.TemporaryDirectory() as d:
        t, f, typ = write_family(d, fam, ins)
        e = dict(os.environ)
        e.update(env)
        dump = subprocess.run([os.path.join(ROOT, "oracle", "schedule_dump"), "-t", t, "-i", f, "-o", "x", "--type", typ] + flags, capture_output=True, text=True, env=e, check=True)
        open(os.path.join(d, "dump.json"), "w").write(dump.stdout)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "msa_replay.py"), os.path.join(d, "dump.json"), os.path.join(d, "r.aln")] + flags,
                           capture_output=True, text=True, env=e, check=True)
        kv = dict(x.split("=") for x in r.stdout.strip().split()[1:])
        out[name] = {"md5": hashlib.md5(open(os.path.join(d, "r.aln"), "rb").read()).hexdigest(), "aln_len": int(kv["aln_len"]), "band_cells": int(kv["band_cells"]),
                     "pairs_per_level": [int(x) for x in kv["pairs_per_level"].split("/")], "deferred_profiles": int(kv["deferred_profiles"]), "retries": int(kv["retries"])}
        print(name, out[name]["md5"], out[name]["deferred_profiles"], out[name]["retries"], flush=True)
json.dump(out, open(OUT, "w"), indent=1)
