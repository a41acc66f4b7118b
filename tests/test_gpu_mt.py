"""Tile-parallel alignment (talco_nuc.hip.h, MT kernels: scouts -> chain -> tiles -> stitch) vs the oracle, through the C ABI (-m gpu).

Levels with few pairs of many tiles each run all tiles of all pairs side by side from PREDICTED start cells; the stitch launch keeps a
tile only when its true start equals the predicted one and computes it in line otherwise, so paths, error codes and band cells must be
those of the plain tile loop (Align_freq, /root/reference/src/TALCO-XDrop.cpp:62-108) whatever the predictions were."""
import numpy as np
import pytest

import oracle_lib as O
from twilight_amd import api, synth

pytestmark = pytest.mark.gpu

M = synth.nucleotide_matrix()


@pytest.fixture()
def knobs(gpu):
    gpu.set_knob(api.KNOB_THR_SMALL, 0)          # (also forgets what the levels of earlier tests found of the 512-row window)
    yield gpu
    gpu.set_knob(api.KNOB_MT_PERTURB, 0)
    gpu.set_knob(api.KNOB_MT_MAX_PAIRS, 1024)
    gpu.set_knob(api.KNOB_MT_MIN_MARKER, 512)
    gpu.set_knob(api.KNOB_MT_LEAD, 320)
    gpu.set_knob(api.KNOB_MT_MARGIN, -1)         # (negative: the defaults -- 40 diagonals, 64 on levels of at most 2048 tile jobs)
    gpu.set_knob(api.KNOB_MT_ROUNDS, 2)
    gpu.set_knob(api.KNOB_MT_THR_JOBS, 256)
    gpu.set_knob(api.KNOB_MT_WIDE, 1)
    gpu.set_knob(api.KNOB_THR_SMALL, 0)
    gpu.set_knob(api.KNOB_MT_ANCHOR, 1)
    gpu.set_knob(api.KNOB_MT_LEAD2, -1)          # (96 / 128)
    gpu.set_knob(api.KNOB_PROT_CORRIDOR, 448)


def _compare(twl, batch, **pk):
    p = twl.make_params(M, **pk)
    aln, n, err = twl.align_batch(p, batch)
    st = twl.get_stats(0)
    oa, on, oerr, ost = O.align_batch(O.make_params(M, **pk), batch, threads=8)
    assert np.array_equal(err, oerr), f"errorType differs: gpu {err.tolist()} oracle {oerr.tolist()}"
    assert np.array_equal(n, on), f"path length differs: gpu {n.tolist()} oracle {on.tolist()}"
    for i in range(batch.n_pairs):
        assert np.array_equal(aln[i, : n[i]], oa[i, : on[i]]), f"pair {i}: path differs"
    if np.all(oerr == 0) or st.n_relaunched == 0:      # (band cells of failed pairs count as well; only attempts in a window that was outgrown do not)
        assert st.band_cells == ost.cells, f"band cells gpu {st.band_cells} oracle {ost.cells}"
    return st, ost


def test_tile_parallel_default_params(knobs):
    """6 pairs x ~6000 columns, marker 1024: ~12 tiles per pair, all predicted."""
    batch = synth.make_level_batch(6, 6000, members=((1, 8), (1, 8)), seed=31)
    st, ost = _compare(knobs, batch)
    assert st.speculative == 3
    assert st.mt_tiles_predicted + st.mt_tiles_inline == ost.tiles
    assert st.mt_tiles_predicted >= 0.8 * ost.tiles, (st.mt_tiles_predicted, st.mt_tiles_inline, st.mt_scouts_failed)


@pytest.mark.parametrize("rounds", [1, 2, 4])
@pytest.mark.parametrize("perturb", [1, 2, 3])
def test_wrong_predictions_are_repaired(knobs, perturb, rounds):
    """Every n-th predicted start is moved by one cell: each round of predict / run / verify gets a pair past one such tile (the
    prediction is redone from the true cell), what is left after the last round is computed in line; results unchanged."""
    knobs.set_knob(api.KNOB_MT_PERTURB, perturb)
    knobs.set_knob(api.KNOB_MT_ROUNDS, rounds)
    batch = synth.make_level_batch(5, 5000, members=((1, 6), (1, 6)), seed=32 + perturb)
    st, ost = _compare(knobs, batch)
    assert st.speculative == 3
    assert st.mt_tiles_predicted + st.mt_tiles_inline == ost.tiles
    if rounds == 1:
        assert st.mt_tiles_inline >= (ost.tiles - batch.n_pairs) // (perturb + 1), (st.mt_tiles_predicted, st.mt_tiles_inline)


def test_throughput_geometry_for_scouts_and_tiles(knobs):
    """The throughput geometry (4 waves x 3 blocks, four workgroups per CU) for the scout and tile launches, forced on a small level."""
    knobs.set_knob(api.KNOB_MT_THR_JOBS, 0)
    batch = synth.make_level_batch(6, 6000, members=((1, 8), (1, 8)), seed=31)
    st, ost = _compare(knobs, batch)
    assert st.speculative == 3 and b"<6, 4, 3" in st.kernel
    assert st.mt_tiles_predicted >= 0.8 * ost.tiles


@pytest.mark.parametrize("marker", [128, 250, 600])
def test_small_markers(knobs, marker):
    knobs.set_knob(api.KNOB_MT_MIN_MARKER, 64)
    batch = synth.make_level_batch(4, 2500, members=((1, 5), (1, 5)), seed=40 + marker)
    st, ost = _compare(knobs, batch, marker=marker)
    assert st.speculative == 3
    assert st.mt_tiles_predicted + st.mt_tiles_inline == ost.tiles


def test_short_lead_and_margin(knobs):
    """Scouts that start too late to find the path: predictions fail, the stitch launch does the work; results unchanged."""
    knobs.set_knob(api.KNOB_MT_LEAD, 16)
    knobs.set_knob(api.KNOB_MT_MARGIN, 2)
    batch = synth.make_level_batch(4, 5000, members=((2, 6), (2, 6)), seed=44, indel=0.02)
    st, ost = _compare(knobs, batch)
    assert st.speculative == 3


@pytest.mark.parametrize("thr_jobs", [256, 0])
def test_anchored_scouts_find_the_starts_a_late_straight_line_scout_cannot(knobs, thr_jobs):
    """The scouts start from the cell on which the consensus letters of the two profiles agree (mt_anchor_kernel), a short lead ahead.  With the
    LONG lead cut to 16 diagonals a scout that starts on the straight line is too late to run into the path; the anchored ones still predict
    most tiles.  Predictions only: the results are the oracle's either way."""
    knobs.set_knob(api.KNOB_MT_LEAD, 16)
    knobs.set_knob(api.KNOB_MT_THR_JOBS, thr_jobs)
    batch = synth.make_level_batch(6, 8000, members=((2, 8), (2, 8)), seed=77, indel=0.01)
    knobs.set_knob(api.KNOB_MT_ANCHOR, 0)
    plain, ost = _compare(knobs, batch)
    knobs.set_knob(api.KNOB_MT_ANCHOR, 1)
    anchored, _ = _compare(knobs, batch)
    assert anchored.mt_tiles_predicted + anchored.mt_tiles_inline == ost.tiles
    assert anchored.mt_tiles_predicted >= 0.7 * ost.tiles, (anchored.mt_tiles_predicted, ost.tiles)
    assert anchored.mt_tiles_predicted > plain.mt_tiles_predicted + ost.tiles // 4, (anchored.mt_tiles_predicted, plain.mt_tiles_predicted, ost.tiles)


@pytest.mark.parametrize("lead2", [16, 48, 200])
def test_anchored_scouts_with_other_leads_and_leaf_pairs(knobs, lead2):
    """Leaf x leaf pairs (one-letter columns) and profiles, anchored scouts 16 / 48 / 200 diagonals ahead: whatever they predict, the oracle's paths."""
    knobs.set_knob(api.KNOB_MT_LEAD2, lead2)
    _compare(knobs, synth.make_level_batch(5, 6000, members=(1, 1), seed=78 + lead2, indel=0.01))
    _compare(knobs, synth.make_level_batch(5, 6000, members=((1, 30), (1, 30)), seed=79 + lead2, indel=0.02, gap_col_rate=0.1))


def test_error_types_through_the_tile_parallel_path(knobs):
    """errorType 2 (band wider than fLen) and 1 (X-drop emptied the band) must surface exactly as from the plain loop."""
    batch = synth.make_level_batch(5, 4000, members=(1, 1), seed=9)
    st, ost = _compare(knobs, batch, flen=128)
    assert st.speculative == 3
    b2 = synth.make_level_batch(4, 4000, members=(1, 1), seed=21, sub=0.75, indel=0.05)
    _compare(knobs, b2, xdrop=40)


def test_unequal_lengths_and_trailing_runs(knobs):
    """One side much shorter than the other: the straight-line start cells are far from the path, tiles end with trailing runs."""
    batch = synth.make_level_batch(4, 5000, members=((1, 4), (1, 4)), seed=51)
    ln = batch.len.copy()
    ln[0, 0] = 3000
    ln[1, 1] = 2500
    ln[2, 0] = 1100
    b = synth.LevelBatch(P=batch.P, seq_len=batch.seq_len, freq=batch.freq, gap_open=batch.gap_open, gap_extend=batch.gap_extend, len=ln, num=batch.num)
    _compare(knobs, b)


def test_many_pairs_more_jobs_than_compute_units(knobs):
    """100 pairs x 4000 columns: ~800 tile jobs on 256 compute units (several rounds per launch)."""
    base = synth.make_level_batch(10, 4000, members=((1, 6), (1, 6)), seed=61)
    idx = np.arange(100) % 10
    big = synth.LevelBatch(P=base.P, seq_len=base.seq_len, freq=base.freq[idx], gap_open=base.gap_open[idx], gap_extend=base.gap_extend[idx], len=base.len[idx], num=base.num[idx])
    p = knobs.make_params(M)
    aln, n, err = knobs.align_batch(p, big)
    st = knobs.get_stats(0)
    assert st.speculative == 3
    oa, on, oerr, ost = O.align_batch(O.make_params(M), base, threads=8)
    assert np.array_equal(n, on[idx]) and np.array_equal(err, oerr[idx])
    for i in range(big.n_pairs):
        assert np.array_equal(aln[i, : n[i]], oa[idx[i], : on[idx[i]]]), f"pair {i}"
    assert st.band_cells == 10 * ost.cells


def test_long_pair_100k(knobs):
    batch = synth.make_level_batch(2, 100000, members=((1, 3), (1, 3)), seed=77, sub=0.03, indel=0.002)
    st, ost = _compare(knobs, batch)
    assert st.speculative == 3 and st.mt_tiles_predicted > 150


# ---- pairs whose band outgrows the 1024-row window: all tiles at once on the 3072-row geometry (round 4) ----
@pytest.mark.parametrize("xdrop,lo,hi", [(14000, 1024, 2048), (26000, 2560, 2944)])
def test_wide_band_pairs_rerun_tile_parallel_on_the_3072_row_window(knobs, xdrop, lo, hi):
    """A larger X-drop widens the band past the 1024-row window of the fast geometries: the pairs come back with the internal window code and
    re-run -- scouts, tiles, stitch -- on 16 waves x 3 blocks.  Paths, error codes and band cells are the oracle's, and those of the
    tile-after-tile path of round 3 (TWL_KNOB_MT_WIDE 0)."""
    batch = synth.make_level_batch(3, 6000, members=((1, 6), (1, 6)), seed=101, sub=0.12, indel=0.01)
    st, ost = _compare(knobs, batch, xdrop=xdrop)
    assert lo < ost.max_width <= hi, ost.max_width            # the case is what it claims to be
    assert st.n_relaunched > 0 and st.mt_tiles_predicted + st.mt_tiles_inline >= ost.tiles, (st.n_relaunched, st.mt_tiles_predicted, st.mt_tiles_inline, ost.tiles)
    p = knobs.make_params(M, xdrop=xdrop)
    aln, n, err = knobs.align_batch(p, batch)
    knobs.set_knob(api.KNOB_MT_WIDE, 0)
    aln0, n0, err0 = knobs.align_batch(p, batch)
    st0 = knobs.get_stats(0)
    assert np.array_equal(aln, aln0) and np.array_equal(n, n0) and np.array_equal(err, err0) and st0.band_cells == st.band_cells


def test_wide_band_rerun_with_spoiled_predictions_and_one_round(knobs):
    knobs.set_knob(api.KNOB_MT_PERTURB, 2)
    knobs.set_knob(api.KNOB_MT_ROUNDS, 1)
    batch = synth.make_level_batch(2, 5000, members=((1, 6), (1, 6)), seed=102, sub=0.12, indel=0.01)
    st, ost = _compare(knobs, batch, xdrop=14000)
    assert st.mt_tiles_inline >= 1      # (re-run, or started on the wide window after a streak the tests above left: either way through the wide tiles)


def test_a_streak_of_wide_pairs_starts_on_the_wide_window(knobs):
    """The deferred pass aligns one pair per level against the same growing root: when the pairs of consecutive small calls all outgrew the fast
    window the next call does not try it first (every 8th does).  Same results either way."""
    batch = synth.make_level_batch(1, 5000, members=((1, 6), (1, 6)), seed=102, sub=0.12, indel=0.01)
    narrow = synth.make_level_batch(1, 5000, members=((1, 6), (1, 6)), seed=31)
    _compare(knobs, narrow)                        # (whatever earlier tests left behind: a small call that fits the fast window ends a streak)
    relaunched = []
    for _ in range(10):
        st, ost = _compare(knobs, batch, xdrop=14000)
        relaunched.append(int(st.n_relaunched))
    assert relaunched[:2] == [1, 1] and relaunched[2:7] == [0] * 5 and relaunched[7] == 1 and relaunched[8:] == [0, 0], relaunched
    st, ost = _compare(knobs, narrow)              # a pair that fits the fast window, through the wide one: still the same answer ...
    assert st.n_relaunched == 0
    st, ost = _compare(knobs, narrow)
    st, ost = _compare(knobs, batch, xdrop=14000)  # ... and the streak is over once the probe call (every 8th) finds the fast window wide enough
    assert st.n_relaunched in (0, 1)


def test_band_wider_than_the_3072_row_window_moves_on_to_the_widest_kernel(knobs):
    """... and a band that outgrows that one too ends on the 4608-row kernel, as before."""
    batch = synth.make_level_batch(2, 5000, members=((1, 4), (1, 4)), seed=103, sub=0.2, indel=0.01)
    st, ost = _compare(knobs, batch, xdrop=30000)
    assert ost.max_width > 2944 and st.n_relaunched >= batch.n_pairs, (ost.max_width, st.n_relaunched)      # (twice that unless the call started on the wide window: a streak left by the tests above)


def test_wide_band_rerun_in_a_large_level_touches_only_its_own_rows(knobs):
    """300 pairs of which two outgrow the window: the tables of the re-run are sized by the two, results unchanged."""
    base = synth.make_level_batch(6, 4400, members=((1, 6), (1, 6)), seed=104)
    wide = synth.make_level_batch(2, 4400, members=((1, 6), (1, 6)), seed=105, sub=0.3, indel=0.02)
    idx = np.arange(300) % 6
    def cat(a, b): return np.concatenate([a[idx], b])
    big = synth.LevelBatch(P=6, seq_len=max(base.seq_len, wide.seq_len), freq=None, gap_open=None, gap_extend=None, len=None, num=None)
    sl = big.seq_len
    def pad(x): return np.pad(x, [(0, 0), (0, 0), (0, sl - x.shape[2])] + [(0, 0)] * (x.ndim - 3))
    big = synth.LevelBatch(P=6, seq_len=sl, freq=cat(pad(base.freq), pad(wide.freq)), gap_open=cat(pad(base.gap_open), pad(wide.gap_open)),
                           gap_extend=cat(pad(base.gap_extend), pad(wide.gap_extend)), len=cat(base.len, wide.len), num=cat(base.num, wide.num))
    pk = dict(xdrop=9000)
    p = knobs.make_params(M, **pk)
    aln, n, err = knobs.align_batch(p, big)
    st = knobs.get_stats(0)
    ob = O.align_batch(O.make_params(M, **pk), base, threads=8)
    ow = O.align_batch(O.make_params(M, **pk), wide, threads=8)
    for i in range(big.n_pairs):
        oa, on, oe = (ob[0][idx[i]], ob[1][idx[i]], ob[2][idx[i]]) if i < 300 else (ow[0][i - 300], ow[1][i - 300], ow[2][i - 300])
        assert err[i] == oe and n[i] == on and np.array_equal(aln[i, : n[i]], oa[:on]), f"pair {i}"
    assert st.band_cells == 50 * ob[3].cells + ow[3].cells
    assert ow[3].max_width > 960 or st.n_relaunched == 0


@pytest.mark.parametrize("onehot", [0, 1])
def test_throughput_level_pairs_that_outgrow_the_768_row_window(knobs, onehot):
    """A level of more pairs than CUs runs on 4 waves x 3 blocks (768-row window, bands up to 640 rows); pairs whose band is wider re-run on
    8 waves x 2 blocks (1024 rows), and on from there.  Pool of 8 pairs (X-drop 9000: bands of ~700-900 rows in most) replicated to 1104."""
    pool = synth.make_level_batch(8, 3000, members=((1, 6), (1, 1) if onehot else (1, 6)), seed=106)
    idx = np.arange(1104) % pool.n_pairs      # (more than one round of 4 workgroups on each of 256 CUs: the throughput launch, not the tile-parallel path)
    batch = synth.LevelBatch(P=pool.P, seq_len=pool.seq_len, freq=pool.freq[idx], gap_open=pool.gap_open[idx], gap_extend=pool.gap_extend[idx], len=pool.len[idx], num=pool.num[idx])
    pk = dict(xdrop=9000)
    p = knobs.make_params(M, **pk)
    knobs.set_knob(api.KNOB_ASSUME_ONEHOT_QUERY, onehot)
    knobs.set_knob(api.KNOB_THR_SMALL, 1)          # (this test is about the 768-row window: not the 512-row one first)
    try:
        aln, ln, err = knobs.align_batch(p, batch)
        st = knobs.get_stats(0)
    finally:
        knobs.set_knob(api.KNOB_ASSUME_ONEHOT_QUERY, 0)
    oa, on, oerr, ost = O.align_batch(O.make_params(M, **pk), pool, threads=8)
    assert 640 < ost.max_width <= 960, ost.max_width
    assert b"<6, 4, 3" in st.kernel and st.n_relaunched > 0 and st.matrix_mode == (5 if onehot else 2), (st.kernel, st.n_relaunched, st.matrix_mode)
    assert np.array_equal(err, oerr[idx]) and np.array_equal(ln, on[idx])
    for i in range(1104):
        assert np.array_equal(aln[i, : ln[i]], oa[idx[i], : on[idx[i]]]), f"pair {i}: path differs"
    assert st.band_cells == ost.cells * 138


@pytest.mark.parametrize("onehot", [0, 1])
def test_levels_of_short_pairs_take_the_512_row_window_until_one_outgrows_it(knobs, onehot):
    """Levels of more pairs than CUs whose pairs are short (R + Q <= 4096) run on 4 waves x 2 blocks, five workgroups per CU (512-row window: a band fits
    when it touches at most 8 row blocks -- always up to 449 rows).  Pairs that outgrow it re-run on the 768-row geometry (and on from there); a level that
    sent more than 3 % of its pairs there keeps the next eligible levels on 768 rows, one that fitted lets them start small.  Results are those of the oracle
    whichever way a level went."""
    pool = synth.make_level_batch(8, 1500, members=((1, 6), (1, 1) if onehot else (1, 6)), seed=107, sub=0.03)
    idx = np.arange(1400) % pool.n_pairs
    batch = synth.LevelBatch(P=pool.P, seq_len=pool.seq_len, freq=pool.freq[idx], gap_open=pool.gap_open[idx], gap_extend=pool.gap_extend[idx], len=pool.len[idx], num=pool.num[idx])
    knobs.set_knob(api.KNOB_THR_SMALL, 0)          # (also clears what earlier levels of this process found)
    knobs.set_knob(api.KNOB_ASSUME_ONEHOT_QUERY, onehot)
    try:
        seen = []
        for pk in (dict(xdrop=3000), dict(xdrop=9000), dict(xdrop=3000)):
            aln, ln, err = knobs.align_batch(knobs.make_params(M, **pk), batch)
            st = knobs.get_stats(0)
            oa, on, oerr, ost = O.align_batch(O.make_params(M, **pk), pool, threads=8)
            seen.append((bytes(st.kernel), st.n_relaunched, ost.max_width))
            assert st.matrix_mode == (5 if onehot else 2)
            assert np.array_equal(err, oerr[idx]) and np.array_equal(ln, on[idx]), pk
            for i in range(1400):
                assert np.array_equal(aln[i, : ln[i]], oa[idx[i], : on[idx[i]]]), f"{pk} pair {i}: path differs"
            assert st.band_cells == ost.cells * 175, pk      # (no failed pair: attempts in an outgrown window do not count)
    finally:
        knobs.set_knob(api.KNOB_ASSUME_ONEHOT_QUERY, 0)
    # X-drop 3000: narrow bands, the level tries the small window and fits; X-drop 9000: starts there on that memory, every pair outgrows it (and the 768-row
    # window after it); the next level stays on 768 rows although it would fit
    assert b"<6, 4, 2" in seen[0][0] and seen[0][1] == 0 and seen[0][2] <= 449, seen
    assert b"<6, 4, 2" in seen[1][0] and seen[1][1] >= 1400 and seen[1][2] > 512, seen
    assert b"<6, 4, 3" in seen[2][0] and seen[2][1] == 0, seen


@pytest.mark.parametrize("xdrop,expect", [(3000, b"<6, 4, 2"), (9000, b"<6, 4, 3")])
def test_a_level_of_eight_rounds_asks_a_sample_of_its_own_pairs(knobs, xdrop, expect):
    """2304 short pairs (nine rounds of one workgroup per CU) and nothing remembered: one pair per CU runs on the 512-row window first and the share that
    outgrew it chooses the window of the rest -- narrow bands: the rest stays on 512 rows; X-drop 9000: the rest runs on 768 rows and the sample's own pairs
    re-run.  Same results."""
    pool = synth.make_level_batch(8, 1500, members=((1, 6), (1, 6)), seed=108, sub=0.03)
    idx = np.arange(2304) % pool.n_pairs
    batch = synth.LevelBatch(P=pool.P, seq_len=pool.seq_len, freq=pool.freq[idx], gap_open=pool.gap_open[idx], gap_extend=pool.gap_extend[idx], len=pool.len[idx], num=pool.num[idx])
    pk = dict(xdrop=xdrop)
    knobs.set_knob(api.KNOB_THR_SMALL, 0)
    aln, ln, err = knobs.align_batch(knobs.make_params(M, **pk), batch)
    st = knobs.get_stats(0)
    oa, on, oerr, ost = O.align_batch(O.make_params(M, **pk), pool, threads=8)
    assert expect in bytes(st.kernel), (st.kernel, st.n_relaunched, ost.max_width)
    assert (st.n_relaunched > 0) == (xdrop == 9000), (st.n_relaunched, ost.max_width)
    assert np.array_equal(err, oerr[idx]) and np.array_equal(ln, on[idx])
    for i in range(2304):
        assert np.array_equal(aln[i, : ln[i]], oa[idx[i], : on[idx[i]]]), f"pair {i}: path differs"
    assert st.band_cells == ost.cells * 288
    # ... and the next level does as the sample said, without asking again
    aln2, ln2, err2 = knobs.align_batch(knobs.make_params(M, **pk), batch)
    assert expect in bytes(knobs.get_stats(0).kernel) and np.array_equal(ln2, ln) and np.array_equal(aln2, aln)


@pytest.mark.parametrize("onehot", [0, 1])
@pytest.mark.parametrize("xdrop", [5000, 9000])
def test_long_pairs_that_outgrow_the_512_row_window_rerun_tile_parallel(knobs, onehot, xdrop):
    """A level of long pairs tries the 512-row window too; the pairs that outgrow it re-run with all their tiles at once (up to TWL_KNOB_MT_MAX_PAIRS of them,
    8+ tiles each) instead of one pair after the other -- a 10 kbp pair's full latency is ~18 ms, its tiles side by side ~3 ms -- and go on to the 3072-row
    stage from there.  1200 pairs of ~5 kbp (94 % of one round of five workgroups per CU: the throughput launch); X-drop 9000: every pair outgrows 512 rows."""
    pool = synth.make_level_batch(8, 5000, members=((1, 6), (1, 1) if onehot else (1, 6)), seed=109)
    idx = np.arange(1200) % pool.n_pairs
    batch = synth.LevelBatch(P=pool.P, seq_len=pool.seq_len, freq=pool.freq[idx], gap_open=pool.gap_open[idx], gap_extend=pool.gap_extend[idx], len=pool.len[idx], num=pool.num[idx])
    pk = dict(xdrop=xdrop)
    knobs.set_knob(api.KNOB_THR_SMALL, 0)
    knobs.set_knob(api.KNOB_MT_MAX_PAIRS, 2048)       # (all 1200 through the tile-parallel re-run)
    knobs.set_knob(api.KNOB_ASSUME_ONEHOT_QUERY, onehot)
    try:
        aln, ln, err = knobs.align_batch(knobs.make_params(M, **pk), batch)
        st = knobs.get_stats(0)
    finally:
        knobs.set_knob(api.KNOB_ASSUME_ONEHOT_QUERY, 0)
    oa, on, oerr, ost = O.align_batch(O.make_params(M, **pk), pool, threads=8)
    assert b"<6, 4, 2" in bytes(st.kernel) and st.matrix_mode == (5 if onehot else 2), (st.kernel, st.matrix_mode)
    if xdrop == 9000:
        assert ost.max_width > 512 and st.n_relaunched >= 1200 and st.mt_tiles_predicted + st.mt_tiles_inline > 0, (ost.max_width, st.n_relaunched)
    assert np.array_equal(err, oerr[idx]) and np.array_equal(ln, on[idx])
    for i in range(1200):
        assert np.array_equal(aln[i, : ln[i]], oa[idx[i], : on[idx[i]]]), f"pair {i}: path differs"
    assert st.band_cells == ost.cells * 150


def test_tile_jobs_take_the_512_row_window_while_the_pass_fits_it(knobs):
    """After a throughput level that fitted the 512-row window the tile jobs of the tile-parallel levels above it run on that geometry too (five workgroups
    per CU); a tile that outgrows it is computed in line by the stitch launch (its failed record is of a window narrower than the stitch launch's), and
    once a level had tiles go that way the rest of the pass is back on 768 rows.  Same results every time."""
    knobs.set_knob(api.KNOB_THR_SMALL, 0)
    pool = synth.make_level_batch(8, 1500, members=((1, 6), (1, 6)), seed=107, sub=0.03)
    idx = np.arange(1400) % pool.n_pairs
    wide_level = synth.LevelBatch(P=pool.P, seq_len=pool.seq_len, freq=pool.freq[idx], gap_open=pool.gap_open[idx], gap_extend=pool.gap_extend[idx], len=pool.len[idx], num=pool.num[idx])
    knobs.align_batch(knobs.make_params(M, xdrop=3000), wide_level)
    assert b"<6, 4, 2" in bytes(knobs.get_stats(0).kernel) and knobs.get_stats(0).n_relaunched == 0
    pool2 = synth.make_level_batch(10, 6000, members=((1, 6), (1, 6)), seed=110, sub=0.03)
    idx2 = np.arange(100) % pool2.n_pairs
    level = synth.LevelBatch(P=pool2.P, seq_len=pool2.seq_len, freq=pool2.freq[idx2], gap_open=pool2.gap_open[idx2], gap_extend=pool2.gap_extend[idx2], len=pool2.len[idx2], num=pool2.num[idx2])
    seen = []
    for xdrop in (3000, 9000, 3000):
        pk = dict(xdrop=xdrop)
        aln, ln, err = knobs.align_batch(knobs.make_params(M, **pk), level)
        st = knobs.get_stats(0)
        oa, on, oerr, ost = O.align_batch(O.make_params(M, **pk), pool2, threads=8)
        seen.append((bytes(st.kernel)[:48], st.mt_tiles_predicted, st.mt_tiles_inline, ost.max_width))
        assert st.speculative == 3
        assert np.array_equal(err, oerr[idx2]) and np.array_equal(ln, on[idx2]), pk
        for i in range(100):
            assert np.array_equal(aln[i, : ln[i]], oa[idx2[i], : on[idx2[i]]]), f"{pk} pair {i}: path differs"
        assert st.band_cells == ost.cells * 10, pk
    assert b"<6, 4, 2, 2, 5" in seen[0][0] and seen[0][2] == 0, seen                  # small tiles, every one kept
    assert b"<6, 4, 2, 2, 5" in seen[1][0] and seen[1][2] > 0 and seen[1][3] > 512, seen      # tiles outgrew the window: computed in line
    assert b"<6, 4, 3, 2, 4" in seen[2][0] and seen[2][2] == 0, seen                  # the pass is off the small window


# ---- protein (P = 22): tile-parallel on the precomputed column scores ----
PM = synth.protein_matrix()


def _compare_p(twl, batch, **pk):
    p = twl.make_params(PM, **pk)
    aln, n, err = twl.align_batch(p, batch)
    st = twl.get_stats(0)
    oa, on, oerr, ost = O.align_batch(O.make_params(PM, **pk), batch, threads=8)
    assert np.array_equal(err, oerr) and np.array_equal(n, on), (err.tolist(), oerr.tolist(), n.tolist(), on.tolist())
    for i in range(batch.n_pairs):
        assert np.array_equal(aln[i, : n[i]], oa[i, : on[i]]), f"pair {i}: path differs"
    if np.all(oerr == 0):
        assert st.band_cells == ost.cells, f"band cells gpu {st.band_cells} oracle {ost.cells}"
    return st, ost


@pytest.mark.parametrize("thr_jobs", [256, 0])
def test_protein_tile_parallel(knobs, thr_jobs):
    """5 pairs x ~2500 aa (5-6 tiles each), on either geometry of the scout / tile launches."""
    knobs.set_knob(api.KNOB_MT_THR_JOBS, thr_jobs)
    batch = synth.make_level_batch(5, 2500, P=22, members=((1, 6), (1, 6)), seed=91, sub=0.15)
    st, ost = _compare_p(knobs, batch)
    assert st.speculative == 3 and st.matrix_mode == 4
    assert st.mt_tiles_predicted + st.mt_tiles_inline == ost.tiles


def test_protein_tile_parallel_with_spoiled_predictions(knobs):
    knobs.set_knob(api.KNOB_MT_PERTURB, 2)
    batch = synth.make_level_batch(4, 3000, P=22, members=((2, 5), (2, 5)), seed=92, sub=0.2)
    st, ost = _compare_p(knobs, batch)
    assert st.speculative == 3 and st.mt_tiles_inline >= 1


@pytest.mark.parametrize("corridor", [0, 64, 160, 448])
def test_protein_score_corridor_changes_no_result(knobs, corridor):
    """Protein levels of few pairs precompute their column scores in a corridor around the straight line between the corners (round 6); the tiles outside it hold NaN, and
    a pair whose band reads one is re-run by the kernel that scores in line.  64 rows is narrower than any band (every pair re-runs), 160 cuts some, 0 is the whole matrix."""
    knobs.set_knob(api.KNOB_PROT_CORRIDOR, corridor)
    batch = synth.make_level_batch(5, 2500, P=22, members=((1, 6), (1, 6)), seed=91, sub=0.15, indel=0.01)
    batch.len[1, 1] = int(batch.len[1, 1] * 0.6)      # an unequal pair: its path leaves the straight line by hundreds of rows
    st, ost = _compare_p(knobs, batch)
    if corridor == 0:
        assert st.n_relaunched == 0
    if corridor == 64:
        assert st.n_relaunched >= 4


def test_protein_tile_parallel_small_marker_and_blosum80(knobs):
    knobs.set_knob(api.KNOB_MT_MIN_MARKER, 64)
    batch = synth.make_level_batch(4, 1500, P=22, members=((1, 4), (1, 4)), seed=93, sub=0.25)
    st, ost = _compare_p(knobs, batch, marker=200)
    assert st.speculative == 3


# ---- randomized campaign on the tile-parallel path (tools/fuzz_mt.py runs it over arbitrary seed ranges) ----
def random_mt_case(seed):
    rng = np.random.default_rng(seed)
    prot = rng.random() < 0.2
    P = 22 if prot else 6
    n = int(rng.integers(1, 10))
    length = int(rng.choice([2200, 3500, 5000, 8000] if not prot else [1800, 2600]))
    members = ((1, int(rng.integers(1, 8))), (1, int(rng.integers(1, 8)))) if rng.random() < 0.8 else (1, 1)
    batch = synth.make_level_batch(n, length, P=P, members=members, seed=int(rng.integers(1 << 30)),
                                   sub=float(rng.choice([0.02, 0.06, 0.15, 0.3] if not prot else [0.1, 0.25])), indel=float(rng.choice([0.0, 0.005, 0.02])),
                                   gap_col_rate=float(rng.choice([0.0, 0.03, 0.2])), length_jitter=float(rng.choice([0.02, 0.3])))
    if rng.random() < 0.25:         # some pairs very unequal in length: straight-line starts far from the path, trailing runs
        k = int(rng.integers(0, n))
        batch.len[k, int(rng.integers(0, 2))] = max(1, int(batch.len[k].min() * rng.uniform(0.1, 0.7)))
    marker = int(rng.choice([200, 512, 1024]))
    pk = dict(marker=marker, xdrop=int(rng.choice([600, 2000, 5000])), flen=int(rng.choice([200, 700, 4096])))
    if not prot and rng.random() < 0.12:      # bands that outgrow the fast window: the 768 -> 1024 -> 3072 -> 4608 row stages, pair scouts (round 4)
        pk["xdrop"] = int(rng.choice([9000, 14000, 26000, 40000]))
        pk["flen"] = 4096
    if rng.random() < 0.25:
        pk["gap_char"] = 0.0
    knobs = {api.KNOB_MT_MIN_MARKER: 64, api.KNOB_MT_PERTURB: int(rng.choice([0, 0, 1, 2, 5])), api.KNOB_MT_ROUNDS: int(rng.choice([1, 2, 3])),
             api.KNOB_MT_LEAD: int(rng.choice([16, 128, 320])), api.KNOB_MT_MARGIN: int(rng.choice([2, 16, 40])), api.KNOB_MT_THR_JOBS: int(rng.choice([0, 256])),
             api.KNOB_MT_ANCHOR: int(rng.choice([1, 1, 0])), api.KNOB_MT_LEAD2: int(rng.choice([16, 64, 96, 160]))}
    return batch, (PM if prot else M), pk, knobs


def check_mt_case(twl, seed):
    batch, matrix, pk, knobs = random_mt_case(seed)
    for k, v in knobs.items():
        twl.set_knob(k, v)
    p = twl.make_params(matrix, **pk)
    aln, n, err = twl.align_batch(p, batch)
    st = twl.get_stats(0)
    oa, on, oerr, ost = O.align_batch(O.make_params(matrix, **pk), batch, threads=8)
    tag = f"seed {seed} P={batch.P} len={batch.len.tolist()} params={pk} knobs={knobs} speculative={st.speculative}"
    assert np.array_equal(err, oerr), f"{tag}: errorType gpu {err.tolist()} oracle {oerr.tolist()}"
    assert np.array_equal(n, on), f"{tag}: path length gpu {n.tolist()} oracle {on.tolist()}"
    for i in range(batch.n_pairs):
        assert np.array_equal(aln[i, : n[i]], oa[i, : on[i]]), f"{tag}: path of pair {i} differs"
    if np.all(oerr == 0):
        assert st.band_cells == ost.cells, f"{tag}: band cells gpu {st.band_cells} oracle {ost.cells}"
    return st.speculative == 3, int(np.count_nonzero(oerr))


def test_random_campaign_on_the_tile_parallel_path(knobs):
    took = 0
    for seed in range(7000, 7040):
        mt, _ = check_mt_case(knobs, seed)
        took += mt
    assert took >= 20, took      # most cases are long enough to take the tile-parallel path


def test_badly_filled_last_round_goes_through_tiles(knobs):
    """More pairs than persistent workgroups (4 per CU) with a small remainder: the full rounds run on the throughput kernel, the remainder
    (the shortest pairs) on the tile-parallel path behind it -- one call, two kinds of launches, same results.  12 distinct pairs replicated."""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n = 4 * cus + 27
    pool = synth.make_level_batch(12, 4400, members=((1, 6), (1, 6)), seed=77)      # 8+ tiles of 1024 anti-diagonals each
    idx = np.arange(n) % pool.n_pairs
    batch = synth.LevelBatch(P=pool.P, seq_len=pool.seq_len, freq=pool.freq[idx], gap_open=pool.gap_open[idx], gap_extend=pool.gap_extend[idx], len=pool.len[idx], num=pool.num[idx])
    p = knobs.make_params(M)
    aln, ln, err = knobs.align_batch(p, batch)
    st = knobs.get_stats(0)
    oa, on, oerr, ost = O.align_batch(O.make_params(M), pool, threads=8)
    assert st.mt_tiles_predicted + st.mt_tiles_inline > 0, "the remainder did not take the tile-parallel path"
    assert np.array_equal(err, oerr[idx]) and np.array_equal(ln, on[idx])
    for i in range(n):
        assert np.array_equal(aln[i, : ln[i]], oa[idx[i], : on[idx[i]]]), f"pair {i}: path differs"
    knobs.set_knob(api.KNOB_MT_TAIL_PCT, 0)      # never: the same call on the throughput kernel alone
    try:
        aln2, ln2, err2 = knobs.align_batch(p, batch)
        st2 = knobs.get_stats(0)
    finally:
        knobs.set_knob(api.KNOB_MT_TAIL_PCT, 70)
    assert st2.mt_tiles_predicted + st2.mt_tiles_inline == 0
    assert np.array_equal(aln, aln2) and np.array_equal(ln, ln2) and np.array_equal(err, err2)
    assert st.band_cells == st2.band_cells


def test_badly_filled_last_round_of_a_leaf_level(knobs):
    """The same with one-letter query rows (a leaf level): full rounds and tiles both take the four-product form of the column score (matrix mode 5)."""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n = 4 * cus + 19
    pool = synth.make_level_batch(10, 4400, members=((1, 6), (1, 1)), seed=78)
    idx = np.arange(n) % pool.n_pairs
    batch = synth.LevelBatch(P=pool.P, seq_len=pool.seq_len, freq=pool.freq[idx], gap_open=pool.gap_open[idx], gap_extend=pool.gap_extend[idx], len=pool.len[idx], num=pool.num[idx])
    p = knobs.make_params(M)
    knobs.set_knob(api.KNOB_ASSUME_ONEHOT_QUERY, 1)
    try:
        aln, ln, err = knobs.align_batch(p, batch)
        st = knobs.get_stats(0)
    finally:
        knobs.set_knob(api.KNOB_ASSUME_ONEHOT_QUERY, 0)
    oa, on, oerr, ost = O.align_batch(O.make_params(M), pool, threads=8)
    assert st.matrix_mode == 5 and st.mt_tiles_predicted + st.mt_tiles_inline > 0, (st.matrix_mode, st.mt_tiles_predicted)
    assert np.array_equal(err, oerr[idx]) and np.array_equal(ln, on[idx])
    for i in range(n):
        assert np.array_equal(aln[i, : ln[i]], oa[idx[i], : on[idx[i]]]), f"pair {i}: path differs"
