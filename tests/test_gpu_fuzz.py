"""Randomized GPU-vs-oracle parity campaign (-m gpu): many small batches with random shapes and parameters.

Every case must match the oracle bit-exactly in error codes, path lengths, paths and band-cell counts.  The seed list is
fixed so a failure is reproducible; tools/fuzz_gpu.py runs the same generator over arbitrary seed ranges.
"""
import numpy as np
import pytest

import oracle_lib as O
from twilight_amd import synth

pytestmark = pytest.mark.gpu


def random_case(seed):
    rng = np.random.default_rng(seed)
    prot = rng.random() < 0.25
    P = 22 if prot else 6
    n = int(rng.integers(1, 9))
    length = int(rng.choice([12, 60, 150, 400, 900, 1500] if not prot else [12, 60, 150, 400, 700]))
    lo_r, lo_q = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    members = ((1, lo_r * int(rng.integers(1, 4))), (1, lo_q * int(rng.integers(1, 4)))) if rng.random() < 0.7 else (1, 1)
    batch = synth.make_level_batch(n, length, P=P, members=members, seed=int(rng.integers(1 << 30)),
                                   sub=float(rng.choice([0.02, 0.06, 0.15, 0.3])), indel=float(rng.choice([0.0, 0.005, 0.02])),
                                   gap_col_rate=float(rng.choice([0.0, 0.03, 0.2])), weights=bool(rng.random() < 0.8),
                                   length_jitter=float(rng.choice([0.02, 0.3])),
                                   gap_open=float(rng.choice([-50, -30, -10])), gap_extend=float(rng.choice([-5, -2, -1])))
    if rng.random() < 0.2:          # make some pairs very unequal in length
        k = int(rng.integers(0, n))
        batch.len[k, int(rng.integers(0, 2))] = max(1, int(batch.len[k].min() * rng.uniform(0.05, 0.6)))
    if prot:
        matrix = synth.protein_matrix(wildcard=bool(rng.random() < 0.3))
    else:
        mode = rng.random()
        if mode < 0.5:
            matrix = synth.nucleotide_matrix()                                  # matrix mode 2
        elif mode < 0.7:
            matrix = synth.nucleotide_matrix(match=10, mismatch=-9, transition=-9)   # still structured
        elif mode < 0.85:
            matrix = rng.integers(-9, 19, size=(5, 5)).astype(np.float32)         # general 5x5 (mode 0)
        else:
            matrix = rng.integers(-9, 19, size=(5, 5)).astype(np.float32)
            matrix[4, :] = 0
            matrix[:, 4] = 0                                                       # zero N row/column only (mode 1)
    ge = float(batch.gap_extend[batch.gap_extend != 0].max()) if np.any(batch.gap_extend != 0) else -5.0
    pk = dict(gap_open=float(rng.choice([-50, -30, -10])), gap_extend=ge if ge < 0 else -5.0,
              marker=int(rng.choice([8, 31, 64, 200, 1024])), xdrop=int(rng.choice([150, 600, 2000, 5000])),
              flen=int(rng.choice([64, 300, 4096])))
    if rng.random() < 0.3:
        pk["gap_char"] = 0.0            # alignment-cpu.cpp:88
    return batch, matrix, pk


def check_case(twl, seed):
    batch, matrix, pk = random_case(seed)
    p = twl.make_params(matrix, **pk)
    aln, n, err = twl.align_batch(p, batch)
    oa, on, oerr, ost = O.align_batch(O.make_params(matrix, **pk), batch, threads=4)
    tag = f"seed {seed} P={batch.P} n={batch.n_pairs} len={batch.len.tolist()} num={batch.num.tolist()} params={pk}"
    assert np.array_equal(err, oerr), f"{tag}: errorType gpu {err.tolist()} oracle {oerr.tolist()}"
    assert np.array_equal(n, on), f"{tag}: path length gpu {n.tolist()} oracle {on.tolist()}"
    for i in range(batch.n_pairs):
        assert np.array_equal(aln[i, : n[i]], oa[i, : on[i]]), f"{tag}: path of pair {i} differs"
    if np.all(oerr == 0):
        assert twl.get_stats(0).band_cells == ost.cells, f"{tag}: band cells gpu {twl.get_stats(0).band_cells} oracle {ost.cells}"
    return int(np.count_nonzero(oerr)), ost


@pytest.mark.parametrize("seed0", [1000, 2000, 3000, 4000])
def test_random_campaign(gpu, seed0):
    errs = 0
    for seed in range(seed0, seed0 + 25):
        e, _ = check_case(gpu, seed)
        errs += e
