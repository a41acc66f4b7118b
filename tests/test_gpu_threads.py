"""Concurrent callers of the C ABI ("must also be callable concurrently from several host threads": the reference calls its kernel from TBB
workers, alignment-cpu.cpp:46, and the deferred / merge paths call it with one pair per level, progressive.cpp:286-291).

What the library promises (include/twl_align.h, "Threads"): every entry point may be called from any thread; calls that touch the same device are
SERIALISED on that device's lock (a call is a whole level batch: it fills the device by itself), error strings are per thread, twl_get_stats reports the
last call of ANY thread on the device (concurrent callers read their own results from their own output arrays, which is what this test checks)."""
import hashlib
import json
import os
import sys
import threading

import numpy as np
import pytest

import oracle_lib as O
from twilight_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from variants import VARIANTS, write_family  # noqa: E402

FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_variants.json")))
pytestmark = pytest.mark.gpu
M = synth.nucleotide_matrix()


@pytest.mark.timeout(600)
def test_four_threads_on_twl_align_batch_and_one_on_the_level_path(gpu, tmp_path):
    from twilight_amd import msa

    # four different batches: leaf pairs, profiles, a multi-tile case with a small marker, one that ends with errorType 2 on some pairs
    jobs = [
        (synth.make_level_batch(24, 900, members=(1, 1), seed=101), {}),
        (synth.make_level_batch(16, 1400, members=((2, 6), (2, 6)), seed=102), {}),
        (synth.make_level_batch(12, 800, members=((1, 4), (1, 4)), seed=103), {"marker": 128}),
        (synth.make_level_batch(10, 1000, members=((1, 3), (1, 3)), seed=104, sub=0.12), {"flen": 96}),
    ]
    expect = []
    for batch, pk in jobs:
        expect.append(O.align_batch(O.make_params(M, **pk), batch, threads=8))
    rounds = 6
    results = [[None] * rounds for _ in jobs]
    errors = []
    start = threading.Barrier(len(jobs) + 1)

    def caller(k):
        try:
            batch, pk = jobs[k]
            p = gpu.make_params(M, **pk)
            start.wait()
            for r in range(rounds):
                results[k][r] = gpu.align_batch(p, batch)
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    # meanwhile: a whole progressive run through libtwl_host -> twl_level_* (prepare / align / restore / commit on the same device)
    name = "nuc_deferrals_cache_compress"
    _, fam, ins, flags, env = [v for v in VARIANTS if v[0] == name][0]
    t, f, typ = write_family(str(tmp_path), fam, ins)
    level_md5 = []

    def level_caller():
        try:
            start.wait()
            for r in range(3):
                out = os.path.join(str(tmp_path), f"lv{r}.aln")
                m = msa.Msa(["-t", t, "-i", f, "-o", out, "--type", typ, "--gpu-index", "0"] + list(flags))
                m.upload()
                m.align()
                m.write(out)
                m.close()
                level_md5.append(hashlib.md5(open(out, "rb").read()).hexdigest())
        except Exception as e:  # noqa: BLE001
            errors.append(("level", repr(e)))

    threads = [threading.Thread(target=caller, args=(k,)) for k in range(len(jobs))] + [threading.Thread(target=level_caller)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=500)
        assert not th.is_alive(), "a caller is stuck"
    assert not errors, errors
    for k, (batch, pk) in enumerate(jobs):
        oa, on, oerr, _ = expect[k]
        for r in range(rounds):
            aln, n, err = results[k][r]
            assert np.array_equal(err, oerr), (k, r, err.tolist(), oerr.tolist())
            assert np.array_equal(n, on), (k, r)
            for i in range(batch.n_pairs):
                assert np.array_equal(aln[i, : n[i]], oa[i, : on[i]]), (k, r, i)
    assert expect[3][2].any(), "the fLen 96 batch was meant to fail some pairs"
    assert level_md5 == [FIX[name]["md5"]] * 3, level_md5
