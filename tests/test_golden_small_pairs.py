"""tests/golden/small_pairs.npz (full fp32 inputs + expected paths, errorTypes, band cells): the oracle must reproduce it on the CPU
and the HIP path must reproduce it through the C ABI on the GPU."""
import os

import numpy as np
import pytest

import oracle_lib as O
from twilight_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "small_pairs.npz")


def _cases():
    z = np.load(GOLD)
    names = sorted({k.split("/")[0] for k in z.files})
    for name in names:
        g = {k.split("/")[1]: z[k] for k in z.files if k.startswith(name + "/")}
        P = g["freq"].shape[-1]
        batch = synth.LevelBatch(P=P, seq_len=g["freq"].shape[2], freq=g["freq"], gap_open=g["gap_open"], gap_extend=g["gap_extend"], len=g["len"], num=g["num"])
        marker, xdrop, flen, zero_gc = (int(v) for v in g["params"])
        pk = {"marker": marker, "flen": flen}
        if xdrop >= 0:
            pk["xdrop"] = xdrop
        if zero_gc:
            pk["gap_char"] = 0.0
        yield name, batch, (synth.protein_matrix() if P == 22 else synth.nucleotide_matrix()), pk, g


def _check(name, aln, n, err, g):
    assert np.array_equal(err, g["err"]), f"{name}: errorType {err.tolist()} != {g['err'].tolist()}"
    assert np.array_equal(n, g["aln_len"]), f"{name}: path lengths"
    for i in range(len(n)):
        assert np.array_equal(aln[i, : n[i]], g["aln"][i, : n[i]]), f"{name}: path of pair {i}"


def test_oracle_reproduces_golden_small_pairs():
    count = 0
    for name, batch, M, pk, g in _cases():
        aln, n, err, st = O.align_batch(O.make_params(M, **pk), batch, threads=2)
        _check(name, aln, n, err, g)
        assert st.cells == int(g["cells"].sum()), f"{name}: band cells"
        count += 1
    assert count == 8


@pytest.mark.gpu
def test_gpu_reproduces_golden_small_pairs(gpu):
    for name, batch, M, pk, g in _cases():
        aln, n, err = gpu.align_batch(gpu.make_params(M, **pk), batch)
        _check(name, aln, n, err, g)
        cells = gpu.get_pair_cells(batch.n_pairs)
        ok = g["err"] == 0
        assert np.array_equal(cells[ok], g["cells"][ok]), f"{name}: per-pair band cells"
