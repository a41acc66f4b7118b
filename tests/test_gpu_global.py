"""The last stage of the re-run chain: TALCO-XDrop with its DP rows in global memory (twilight_amd/csrc/talco_global.hip.h), for bands wider than the
4608-row window of the widest register kernel.  Such bands need fLen > 4608: the retries of the deferred pass raise fLen to
min(int(fLen * 1.2) << 1, min(R, Q)) (/root/reference/src/alignment-cpu.cpp:116-129, TALCO-XDrop.cpp:258,331-338).  Until round 5 the run ended with
TWL_ERR_UNSUPPORTED there."""
import os

import numpy as np
import pytest

import oracle_lib as O
from twilight_amd import api, synth

pytestmark = pytest.mark.gpu
M = synth.nucleotide_matrix()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def forced(gpu):
    gpu.set_knob(api.KNOB_FORCE_GLOBAL, 1)
    yield gpu
    gpu.set_knob(api.KNOB_FORCE_GLOBAL, 0)


def _compare(twl, batch, matrix=None, **pk):
    mat = M if matrix is None else matrix
    p = twl.make_params(mat, **pk)
    aln, n, err = twl.align_batch(p, batch)
    oa, on, oerr, ost = O.align_batch(O.make_params(mat, **pk), batch, threads=8)
    assert np.array_equal(err, oerr), f"errorType differs: gpu {err.tolist()} oracle {oerr.tolist()}"
    assert np.array_equal(n, on), f"path length differs: gpu {n.tolist()} oracle {on.tolist()}"
    for i in range(batch.n_pairs):
        assert np.array_equal(aln[i, : n[i]], oa[i, : on[i]]), f"pair {i}: path differs"
    st = twl.get_stats(0)
    assert st.band_cells == ost.cells, (st.band_cells, ost.cells)      # (failed pairs included: every kernel counts the cells up to the failure)
    return st, ost


@pytest.mark.timeout(600)
def test_band_of_6500_rows_with_flen_9830(gpu):
    """What a retry of the deferred pass asks for: fLen = min(int(4096 * 1.2) << 1, ...) = 9830, a large X-drop, divergent pairs of 6.5 kbp -- the band
    grows to the whole shorter side, far beyond the 4480 rows the widest register window holds."""
    batch = synth.make_level_batch(2, 6500, members=((1, 4), (1, 4)), seed=105, sub=0.25, indel=0.01)
    st, ost = _compare(gpu, batch, xdrop=60000, flen=9830)
    assert ost.max_width > 4480, ost.max_width
    assert st.n_relaunched >= batch.n_pairs and st.window >= ost.max_width, (st.n_relaunched, st.window)


def test_flen_between_the_widest_window_and_the_band_is_error_type_2(gpu):
    """fLen 5000 < band: the pair outgrows every register window first and then fails with errorType 2 exactly where the reference does."""
    batch = synth.make_level_batch(1, 6500, members=((1, 4), (1, 4)), seed=105, sub=0.25, indel=0.01)
    st, ost = _compare(gpu, batch, xdrop=60000, flen=5000)
    oa, on, oerr, _ = O.align_batch(O.make_params(M, xdrop=60000, flen=5000), batch, threads=2)
    assert oerr.tolist() == [2]


@pytest.mark.parametrize("members", [(1, 1), ((2, 6), (2, 6)), (1, (3, 9))])
def test_small_pairs_through_the_global_kernel(forced, members):
    batch = synth.make_level_batch(8, 600, members=members, seed=11)
    st, ost = _compare(forced, batch)
    assert b"talco_global_kernel" in st.kernel


@pytest.mark.parametrize("marker", [16, 33, 128, 250])
def test_markers_through_the_global_kernel(forced, marker):
    batch = synth.make_level_batch(6, 700, members=((1, 4), (1, 4)), seed=5 + marker)
    st, ost = _compare(forced, batch, marker=marker)
    assert ost.tiles > batch.n_pairs


def test_error_types_through_the_global_kernel(forced):
    batch = synth.make_level_batch(8, 900, members=((1, 3), (1, 3)), seed=104, sub=0.12)
    st, ost = _compare(forced, batch, flen=96)                  # errorType 2 on most pairs
    oerr = O.align_batch(O.make_params(M, flen=96), batch, threads=4)[2]
    assert (oerr == 2).any()
    rng = np.random.default_rng(7)
    far = synth.make_level_batch(4, 800, members=((1, 2), (1, 2)), seed=9, sub=0.7)      # unrelated sequences, tiny X-drop: the band empties (errorType 1)
    st, ost = _compare(forced, far, xdrop=40)
    assert (O.align_batch(O.make_params(M, xdrop=40), far, threads=4)[2] == 1).any()
    del rng


def test_unequal_lengths_and_empty_sides_through_the_global_kernel(forced):
    batch = synth.make_level_batch(6, 900, members=((1, 3), (1, 3)), seed=21)
    batch.len[0] = (900 // 3, batch.len[0][1])      # a short reference against a long query: trailing run of 1s
    batch.len[1] = (batch.len[1][0], 200)
    batch.len[2] = (0, batch.len[2][1])             # an empty side: length 0, errorType 0 (the caller fills the all-gap path)
    _compare(forced, batch)


def test_protein_through_the_global_kernel(forced):
    Mp = synth.protein_matrix()
    batch = synth.make_level_batch(5, 400, members=((1, 5), (1, 5)), seed=33, P=22, sub=0.15)
    st, ost = _compare(forced, batch, matrix=Mp)
    assert b"talco_global_kernel<22>" in st.kernel
    _compare(forced, batch, matrix=Mp, marker=64)


def test_golden_small_pairs_through_the_global_kernel(forced):
    """The committed fixture (tests/golden/small_pairs.npz: inputs and the checker's paths, error codes and band cells) through the global-memory kernel."""
    from test_golden_small_pairs import _cases, _check

    for name, batch, mat, pk, g in _cases():
        aln, n, err = forced.align_batch(forced.make_params(mat, **pk), batch)
        _check(name, aln, n, err, g)
        cells = forced.get_pair_cells(batch.n_pairs)
        ok = g["err"] == 0
        assert np.array_equal(cells[ok], g["cells"][ok]), f"{name}: per-pair band cells"
