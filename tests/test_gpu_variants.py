"""The product CLI on the GPU (device-resident and host-staged level kernels) against tests/golden/e2e_variants.json, the fixture that
the INDEPENDENT replay wrote (oracle/msa_replay.py; see tests/test_replay_cpu.py): MSA md5, band cells, level batches of every variant."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from variants import VARIANTS, write_family  # noqa: E402

FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_variants.json")))
pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
@pytest.mark.parametrize("name", [v[0] for v in VARIANTS])
def test_product_cli_reproduces_the_replay_fixture(built, tmp_path, name):
    _, fam, ins, flags, env = [v for v in VARIANTS if v[0] == name][0]
    d = str(tmp_path)
    t, f, typ = write_family(d, fam, ins)
    e = dict(os.environ)
    e.update(env)
    fx = FIX[name]
    for extra in ([], ["--host-staged"]):
        out = os.path.join(d, "gpu.aln")
        r = subprocess.run([os.path.join(ROOT, "twilight_amd", "twilight-mi355x"), "-t", t, "-i", f, "-o", out, "--type", typ, "-v", "--check"] + flags + extra,
                           capture_output=True, text=True, env=e)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert "illegal alignment" not in r.stderr
        assert hashlib.md5(open(out, "rb").read()).hexdigest() == fx["md5"], extra
        tail = [l for l in r.stderr.splitlines() if l.startswith("Wrote")][-1]
        assert int(tail.split(" band cells")[0].split()[-1]) == fx["band_cells"], extra
        levels = [int(l.split("aligned ")[1].split()[0]) for l in r.stderr.splitlines() if l.startswith("Level ")]
        assert levels == fx["pairs_per_level"], extra
