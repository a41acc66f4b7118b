"""The host mirror's rows checked against something outside itself (no GPU needed).

oracle/e2e_oracle is built from the product's host sources + the oracle DP.  oracle/msa_replay.py is a separate implementation of
everything after the level schedule, written in numpy from the reference source.  For every variant of tests/variants.py (gappy
thresholds, wildcard with ambiguous letters, gap penalties, proteins with BLOSUM45/62/80, low-quality deferrals with and without
--filter, the cached-profile and compressed-group branches through lowered thresholds, an X-drop failure retried in the deferred pass)
the two must write the same MSA, count the same band cells and run the same level batches -- and both must equal the committed fixture
tests/golden/e2e_variants.json, which the GPU suite holds the product CLI to."""
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


def _md5(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("name", [v[0] for v in VARIANTS])
def test_cpu_checker_equals_independent_replay_and_fixture(built, tmp_path, name):
    _, fam, ins, flags, env = [v for v in VARIANTS if v[0] == name][0]
    d = str(tmp_path)
    t, f, typ = write_family(d, fam, ins)
    e = dict(os.environ)
    e.update(env)
    ref = os.path.join(d, "ref.aln")
    r = subprocess.run([os.path.join(ROOT, "oracle", "e2e_oracle"), "-t", t, "-i", f, "-o", ref, "--type", typ, "--threads", "4"] + flags, capture_output=True, text=True, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("E2E")][-1]
    kv = dict(x.split("=") for x in line.split()[1:])
    dump = subprocess.run([os.path.join(ROOT, "oracle", "schedule_dump"), "-t", t, "-i", f, "-o", "x", "--type", typ] + flags, capture_output=True, text=True, env=e, check=True)
    open(os.path.join(d, "dump.json"), "w").write(dump.stdout)
    rep = os.path.join(d, "rep.aln")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "msa_replay.py"), os.path.join(d, "dump.json"), rep] + flags, capture_output=True, text=True, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    rk = dict(x.split("=") for x in p.stdout.strip().split()[1:])
    assert _md5(ref) == _md5(rep)
    assert kv["band_cells"] == rk["band_cells"] and kv["pairs_per_level"] == rk["pairs_per_level"] and kv["aln_len"] == rk["aln_len"]
    fx = FIX[name]
    assert _md5(ref) == fx["md5"] and int(kv["band_cells"]) == fx["band_cells"] and [int(x) for x in kv["pairs_per_level"].split("/")] == fx["pairs_per_level"]
    if "deferr" in name or "retried" in name:
        assert fx["deferred_profiles"] > 0 and "Realign profiles that have been deferred" in r.stderr
    if "retried" in name:
        assert fx["retries"] > 0
