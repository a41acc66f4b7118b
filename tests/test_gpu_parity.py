"""HIP path vs oracle through the C ABI: bit-exact paths, error codes and band-cell counts (-m gpu)."""
import numpy as np
import pytest

import oracle_lib as O
from twilight_amd import api, synth

pytestmark = pytest.mark.gpu

M = synth.nucleotide_matrix()


@pytest.fixture(autouse=True)
def _reset_knobs(gpu):
    """Tests that force a kernel variant (twl_set_knob) leave the defaults behind them."""
    yield
    gpu.set_knob(api.KNOB_PROT_MODE, 0)
    gpu.set_knob(api.KNOB_ASSUME_ONEHOT_QUERY, 0)


def _compare(twl, batch, matrix=None, **pk):
    mat = M if matrix is None else matrix
    p = twl.make_params(mat, **pk)
    aln, n, err = twl.align_batch(p, batch)
    oa, on, oerr, ost = O.align_batch(O.make_params(mat, **pk), batch, threads=8)
    assert np.array_equal(err, oerr), f"errorType differs: gpu {err.tolist()} oracle {oerr.tolist()}"
    assert np.array_equal(n, on), f"path length differs: gpu {n.tolist()} oracle {on.tolist()}"
    for i in range(batch.n_pairs):
        if not np.array_equal(aln[i, : n[i]], oa[i, : on[i]]):
            first = int(np.flatnonzero(aln[i, : n[i]] != oa[i, : on[i]])[0])
            raise AssertionError(f"pair {i}: path differs first at step {first} of {n[i]}")
    st = twl.get_stats(0)
    cells_ok = ost.cells
    if np.all(oerr == 0):
        assert st.band_cells == cells_ok, f"band cells gpu {st.band_cells} oracle {cells_ok}"
    return st, ost


@pytest.mark.parametrize("members", [(1, 1), ((2, 6), (2, 6)), (1, (3, 9))])
def test_small_pairs_default_params(gpu, members):
    batch = synth.make_level_batch(12, 600, members=members, seed=11)
    _compare(gpu, batch)


def test_multi_tile_small_marker(gpu):
    # marker 128 -> many tiles per pair, exercises convergence pointers, tile stitching, tb flush of partial groups
    batch = synth.make_level_batch(10, 900, members=((1, 5), (1, 5)), seed=3)
    st, ost = _compare(gpu, batch, marker=128)
    assert ost.tiles > 3 * batch.n_pairs


@pytest.mark.parametrize("marker", [16, 33, 250])
def test_marker_sweep(gpu, marker):
    batch = synth.make_level_batch(6, 500, members=((1, 3), (1, 3)), seed=5 + marker)
    _compare(gpu, batch, marker=marker)


def test_rnasim_shaped_1600(gpu):
    batch = synth.make_level_batch(24, 1600, members=((1, 8), (1, 8)), seed=20260502)
    st, ost = _compare(gpu, batch)
    assert ost.max_width > 300


def test_error_type_2_small_flen(gpu):
    batch = synth.make_level_batch(8, 800, members=(1, 1), seed=9)
    p = dict(flen=128)
    st, ost = _compare(gpu, batch, **p)


def test_error_type_1_tiny_xdrop(gpu):
    # unrelated sequences with a tiny X-drop: the band dies
    rng = np.random.default_rng(4)
    batch = synth.make_level_batch(6, 400, members=(1, 1), seed=21, sub=0.75, indel=0.05)
    _compare(gpu, batch, xdrop=40)


def test_ragged_and_tiny(gpu):
    b1 = synth.make_level_batch(5, 40, members=((1, 3), (1, 3)), seed=1, length_jitter=0.5)
    _compare(gpu, b1)
    b2 = synth.make_level_batch(4, 9, members=(1, 1), seed=2, indel=0.0)
    _compare(gpu, b2)


def test_empty_side_and_single_column(gpu):
    batch = synth.make_level_batch(4, 50, members=(1, 1), seed=8)
    batch.len[1, 0] = 0          # empty reference side -> aln_len 0, err 0 (caller emits the all-gap path)
    batch.len[2, 1] = 1          # single query column
    p = gpu.make_params(M)
    aln, n, err = gpu.align_batch(p, batch)
    oa, on, oerr, _ = O.align_batch(O.make_params(M), batch, threads=1)
    assert n[1] == 0 and err[1] == 0
    assert np.array_equal(n, on) and np.array_equal(err, oerr)
    for i in range(batch.n_pairs):
        assert np.array_equal(aln[i, : n[i]], oa[i, : on[i]])


def test_length_mismatch_pairs(gpu):
    # very different lengths: long end gaps, band hugging one edge, trailing runs at tile exit
    rng = np.random.default_rng(12)
    b = synth.make_level_batch(6, 700, members=(1, 1), seed=31)
    b.len[:, 1] = (b.len[:, 1] * np.array([0.3, 0.5, 0.7, 0.9, 0.2, 1.0])).astype(np.int32).clip(1)
    _compare(gpu, b, marker=64)


# ---- protein: P = 22, 5 x BLOSUM62 (BASELINE config 5) ----
MP = synth.protein_matrix()


@pytest.mark.parametrize("members", [(1, 1), ((2, 5), (1, 4))])
def test_protein_small(gpu, members):
    batch = synth.make_level_batch(8, 400, P=synth.PROT_P, members=members, seed=17, sub=0.15)
    _compare(gpu, batch, matrix=MP)


def test_protein_multi_tile_and_wildcard(gpu):
    batch = synth.make_level_batch(6, 700, P=synth.PROT_P, members=((1, 4), (1, 4)), seed=23, sub=0.12)
    st, ost = _compare(gpu, batch, matrix=synth.protein_matrix(wildcard=True), marker=96)
    assert ost.tiles > 3 * batch.n_pairs


def test_protein_2k(gpu):
    batch = synth.make_level_batch(6, 2000, P=synth.PROT_P, members=((1, 6), (1, 6)), seed=29, sub=0.2)
    _compare(gpu, batch, matrix=MP)


# ---- device-resident entry point (twl_align_batch_device: HBM in, HBM out; what bench.py times) ----
def test_device_pointer_form_equals_host_form(gpu):
    import torch

    batch = synth.make_level_batch(20, 900, members=((1, 6), (1, 6)), seed=41, length_jitter=0.3)
    p = gpu.make_params(M)
    a_host, n_host, e_host = gpu.align_batch(p, batch)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(np.ascontiguousarray(getattr(batch, k))).to(dev) for k in ("freq", "gap_open", "gap_extend", "len", "num")}
    n, sl = batch.n_pairs, batch.seq_len
    aln = torch.zeros((n, 2 * sl), dtype=torch.int8, device=dev)
    alen = torch.zeros(n, dtype=torch.int32, device=dev)
    err = torch.full((n,), 7, dtype=torch.int16, device=dev)
    torch.cuda.synchronize()      # the inputs were produced on torch's stream; with stream=None the library runs on its own (twl_align.h)
    for _ in range(2):      # twice: buffers and the work queue are reused across calls
        gpu.align_batch_device(p, n, sl, t["freq"].data_ptr(), t["gap_open"].data_ptr(), t["gap_extend"].data_ptr(), t["len"].data_ptr(),
                               t["num"].data_ptr(), aln.data_ptr(), alen.data_ptr(), err.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(alen.cpu().numpy(), n_host) and np.array_equal(err.cpu().numpy(), e_host)
        d_aln = aln.cpu().numpy()
        for i in range(n):
            assert np.array_equal(d_aln[i, : n_host[i]], a_host[i, : n_host[i]])
    st = gpu.get_stats(0)
    cells = gpu.get_pair_cells(n)
    assert int(cells.sum()) == st.band_cells and st.kernel_ms > 0
    oa, on, oerr, ost = O.align_batch(O.make_params(M), batch, threads=8)
    assert st.band_cells == ost.cells


def test_device_pointer_form_with_unaligned_outputs(gpu):
    """The caller's length / error-code arrays need not be 16-byte aligned (views into larger tensors): the zero-fill in front of the DP
    kernels takes its byte-wise path for them, the result block comes back the same."""
    import torch

    batch = synth.make_level_batch(7, 700, members=((1, 4), (1, 4)), seed=43, length_jitter=0.3)
    p = gpu.make_params(M)
    a_host, n_host, e_host = gpu.align_batch(p, batch)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(np.ascontiguousarray(getattr(batch, k))).to(dev) for k in ("freq", "gap_open", "gap_extend", "len", "num")}
    n, sl = batch.n_pairs, batch.seq_len
    aln = torch.zeros((n, 2 * sl), dtype=torch.int8, device=dev)
    alen_big = torch.full((n + 3,), 99, dtype=torch.int32, device=dev)
    err_big = torch.full((n + 5,), 7, dtype=torch.int16, device=dev)
    alen, err = alen_big[1 : n + 1], err_big[3 : n + 3]          # 4 and 6 bytes past an aligned address
    assert alen.data_ptr() % 16 != 0 and err.data_ptr() % 16 != 0
    torch.cuda.synchronize()
    gpu.align_batch_device(p, n, sl, t["freq"].data_ptr(), t["gap_open"].data_ptr(), t["gap_extend"].data_ptr(), t["len"].data_ptr(),
                           t["num"].data_ptr(), aln.data_ptr(), alen.data_ptr(), err.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(alen.cpu().numpy(), n_host) and np.array_equal(err.cpu().numpy(), e_host)
    assert int(alen_big[0]) == 99 and int(alen_big[n + 1]) == 99 and int(err_big[2]) == 7 and int(err_big[n + 3]) == 7      # nothing written outside the views
    d_aln = aln.cpu().numpy()
    for i in range(n):
        assert np.array_equal(d_aln[i, : n_host[i]], a_host[i, : n_host[i]])


def test_many_small_pairs_more_than_workgroups(gpu):
    # more work items than persistent workgroups: exercises the device-side work queue
    batch = synth.make_level_batch(48, 120, members=((1, 3), (1, 3)), seed=43, length_jitter=0.5)
    idx = np.arange(1500) % 48
    big = synth.LevelBatch(P=batch.P, seq_len=batch.seq_len, freq=batch.freq[idx], gap_open=batch.gap_open[idx], gap_extend=batch.gap_extend[idx],
                           len=batch.len[idx], num=batch.num[idx])
    p = gpu.make_params(M)
    aln, n, err = gpu.align_batch(p, big)
    oa, on, oerr, ost = O.align_batch(O.make_params(M), batch, threads=8)
    assert np.array_equal(n, on[idx]) and np.array_equal(err, oerr[idx])
    for i in range(big.n_pairs):
        assert np.array_equal(aln[i, : n[i]], oa[idx[i], : on[idx[i]]])


def _rnasim_leaf_batch(n_pairs):
    """Leaf-vs-leaf pairs from the reference's own RNASim sample (tests/golden/RNASim.fa.gz): one-hot columns, plain gap penalties."""
    import gzip
    import os

    seqs, cur = [], []
    with gzip.open(os.path.join(os.path.dirname(__file__), "golden", "RNASim.fa.gz"), "rt") as f:
        for line in f:
            if line.startswith(">"):
                if cur:
                    seqs.append("".join(cur))
                cur = []
            else:
                cur.append(line.strip())
    if cur:
        seqs.append("".join(cur))
    idx = {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3}
    n = min(n_pairs, len(seqs) // 2)
    sl = max(len(s) for s in seqs[: 2 * n])
    freq = np.zeros((n, 2, sl, 6), dtype=np.float32)
    gop = np.zeros((n, 2, sl), dtype=np.float32)
    gex = np.zeros((n, 2, sl), dtype=np.float32)
    ln = np.zeros((n, 2), dtype=np.int32)
    for i in range(n):
        for sd in range(2):
            s = seqs[2 * i + sd].upper()
            code = np.array([idx.get(c, 4) for c in s])
            freq[i, sd, np.arange(len(s)), code] = 1.0
            gop[i, sd, : len(s)] = -50.0
            gex[i, sd, : len(s)] = -5.0
            ln[i, sd] = len(s)
    return synth.LevelBatch(P=6, seq_len=sl, freq=freq, gap_open=gop, gap_extend=gex, len=ln, num=np.ones((n, 2), dtype=np.int32))


def test_config2_rnasim_leaf_level_band512(gpu):
    """BASELINE configs[1]: RNASim leaf pairs as one level batch with fLen = 512, xdrop = 4000.  GPU and oracle must
    agree on every path and on which pairs overflow the 512-wide band (errorType 2: deferred in the first pass)."""
    batch = _rnasim_leaf_batch(289)
    st, ost = _compare(gpu, batch, flen=512, xdrop=4000)
    # and with the reference's default parameters (xdrop 5000, fLen 4096) nothing fails on this data
    p = gpu.make_params(M)
    aln, n, err = gpu.align_batch(p, batch)
    assert not err.any() and (n > 0).all()


@pytest.mark.parametrize("mode", ["dense", "sparse", "presim"])
def test_protein_score_modes_match_oracle(gpu, mode, monkeypatch):
    """The three protein column-score paths (dense loop, loop over non-zero reference letters, scores precomputed by
    score_matrix_kernel) are the same arithmetic in the same order: each must reproduce the oracle bit for bit."""
    gpu.set_knob(api.KNOB_PROT_MODE, api.PROT_MODES[mode])
    PM = synth.protein_matrix()
    for seed, members, length in ((3, (1, 1), 500), (4, ((2, 9), (1, 6)), 700), (5, ((20, 40), (20, 40)), 300)):
        batch = synth.make_level_batch(6, length, members=members, seed=seed, P=22, sub=0.25)
        _compare(gpu, batch, matrix=PM)
    # multi-tile (small marker) and a zero gap-letter score (deferred-pass rule, alignment-cpu.cpp:88)
    batch = synth.make_level_batch(5, 900, members=((1, 5), (1, 5)), seed=8, P=22, sub=0.2)
    _compare(gpu, batch, matrix=PM, marker=128)
    _compare(gpu, batch, matrix=PM, gap_char=0.0)


@pytest.mark.timeout(900)
def test_full_size_level_properties(gpu):
    """BASELINE size (10 kbp profiles, a level-sized batch): properties that do not need the oracle at full size -- every path
    consumes exactly (R, Q), replicas of a pair get the identical path and band-cell count whichever workgroup runs them, a second
    run is identical -- plus the oracle on a sample of the full-size pairs."""
    pool = synth.make_level_batch(24, 10000, members=((1, 8), (1, 8)), seed=20260501)
    idx = np.arange(600) % pool.n_pairs
    big = synth.LevelBatch(P=pool.P, seq_len=pool.seq_len, freq=pool.freq[idx], gap_open=pool.gap_open[idx], gap_extend=pool.gap_extend[idx],
                           len=pool.len[idx], num=pool.num[idx])
    p = gpu.make_params(M)
    aln, n, err = gpu.align_batch(p, big)
    cells = gpu.get_pair_cells(big.n_pairs)
    assert not err.any()
    for i in range(big.n_pairs):
        assert synth.path_consumes(aln[i], int(n[i])) == (int(big.len[i, 0]), int(big.len[i, 1])), f"pair {i}"
        j = int(idx[i])
        if i != j:
            assert n[i] == n[j] and np.array_equal(aln[i, : n[i]], aln[j, : n[j]]) and cells[i] == cells[j], f"replica {i} of {j}"
    aln2, n2, err2 = gpu.align_batch(p, big)
    assert np.array_equal(n, n2) and np.array_equal(cells, gpu.get_pair_cells(big.n_pairs))
    assert all(np.array_equal(aln[i, : n[i]], aln2[i, : n[i]]) for i in range(big.n_pairs))
    sample = synth.LevelBatch(P=pool.P, seq_len=pool.seq_len, freq=pool.freq[:4], gap_open=pool.gap_open[:4], gap_extend=pool.gap_extend[:4],
                              len=pool.len[:4], num=pool.num[:4])
    oa, on, oerr, ost = O.align_batch(O.make_params(M), sample, threads=8)
    assert np.array_equal(on, n[:4]) and not oerr.any()
    for i in range(4):
        assert np.array_equal(oa[i, : on[i]], aln[i, : n[i]])
    assert ost.cells == int(cells[:4].sum())


@pytest.mark.parametrize("kind", ["nuc", "nuc_wildcard", "prot"])
def test_column_scores_match_oracle(gpu, kind):
    """SP column scores (similarScore, TALCO-XDrop.cpp:444): the kernels' values against the oracle's twlo_column_score for every cell
    of small profile pairs -- equal as floats (the kernels may return -0 where the oracle returns +0; nothing downstream can see that)."""
    P = 22 if kind == "prot" else 6
    mat = (synth.protein_matrix() if kind == "prot" else synth.nucleotide_matrix()).copy()
    if kind == "nuc_wildcard":          # -w: N scores like a match (scoring-matrix.cpp:104-110) -> the general 5x5 arithmetic matters
        mat[4, :] = 18.0
        mat[:, 4] = 18.0
    b = synth.make_level_batch(2, 80, members=((2, 7), (1, 5)), seed=31, P=P, sub=0.2, gap_col_rate=0.1)
    for gap_char in (None, 0.0):
        pk = {} if gap_char is None else {"gap_char": gap_char}
        p, op = gpu.make_params(mat, **pk), O.make_params(mat, **pk)
        for i in range(b.n_pairs):
            R, Q = int(b.len[i, 0]), int(b.len[i, 1])
            ref, qry = b.freq[i, 0, :R], b.freq[i, 1, :Q]
            got = gpu.column_scores(p, ref, qry, int(b.num[i, 0]), int(b.num[i, 1]))
            denom = float(np.float32(b.num[i, 0]) * np.float32(b.num[i, 1]))
            want = np.array([[O.column_score(op, ref[j], qry[ii], denom) for j in range(R)] for ii in range(Q)], dtype=np.float32)
            assert got.shape == want.shape and np.array_equal(got, want), f"{kind} pair {i}: {int((got != want).sum())} of {got.size} cells differ"


@pytest.mark.parametrize("kind", ["nuc_mode2", "nuc_mode2_leaf", "nuc_mode5_leaf_query", "nuc_mode5_leaf_query_general_core", "nuc_mode1", "nuc_mode0_wildcard",
                                  "nuc_mode0_random", "prot_sparse"])
def test_scores_inside_the_dp_kernel_match_oracle(gpu, kind, monkeypatch):
    """The column score of every cell the band visits, written out by a diagnostics instantiation of the DP kernel itself (same code:
    hoisted-reciprocal division, row-pair packed products, sparse protein loop), against twlo_column_score -- equal as floats."""
    rng = np.random.default_rng(5)
    P = 22 if kind == "prot_sparse" else 6
    if kind == "prot_sparse": mat = synth.protein_matrix().copy()
    elif kind in ("nuc_mode2", "nuc_mode2_leaf", "nuc_mode5_leaf_query"): mat = synth.nucleotide_matrix().copy()
    elif kind in ("nuc_mode1", "nuc_mode5_leaf_query_general_core"):
        mat = rng.integers(-9, 19, size=(5, 5)).astype(np.float32)
        mat[4, :] = 0
        mat[:, 4] = 0
    elif kind == "nuc_mode0_wildcard":
        mat = synth.nucleotide_matrix().copy()
        mat[4, :] = 18.0
        mat[:, 4] = 18.0
    else: mat = rng.integers(-9, 19, size=(5, 5)).astype(np.float32)
    members = (1, 1) if kind == "nuc_mode2_leaf" else ((2, 7), (1, 5))
    if kind.startswith("nuc_mode5"):
        # single-sequence query sides (profiles on the reference side): the device-resident level path tells the kernels, which then take
        # the one-letter form of the column score (matrix mode 5); through this entry the test has to say so itself
        members = ((2, 7), 1)
        gpu.set_knob(api.KNOB_ASSUME_ONEHOT_QUERY, 1)
    b = synth.make_level_batch(2, 700, members=members, seed=37, P=P, sub=0.1, gap_col_rate=0.1)
    for gap_char in (None, 0.0):
        pk = dict(marker=128)          # several tiles: the tile offsets of the dump are exercised too
        if gap_char is not None: pk["gap_char"] = gap_char
        p, op = gpu.make_params(mat, **pk), O.make_params(mat, **pk)
        for i in range(b.n_pairs):
            R, Q = int(b.len[i, 0]), int(b.len[i, 1])
            got = gpu.dp_column_scores(p, b, i)
            seen = ~np.isnan(got)
            assert seen.sum() > 20 * (R + Q), f"{kind}: the band visited only {int(seen.sum())} cells"
            denom = float(np.float32(b.num[i, 0]) * np.float32(b.num[i, 1]))
            ref, qry = b.freq[i, 0, :R], b.freq[i, 1, :Q]
            ii, jj = np.nonzero(seen)
            step = max(1, len(ii) // 6000)          # every visited cell of short pairs, an even sample of long ones
            bad = 0
            for t in range(0, len(ii), step):
                want = np.float32(O.column_score(op, ref[jj[t]], qry[ii[t]], denom))
                bad += int(got[ii[t], jj[t]] != want)
            assert bad == 0, f"{kind} pair {i}: {bad} sampled cells differ"


@pytest.mark.timeout(900)
def test_long_pairs_100k(gpu):
    """Two pairs of ~100 kbp profiles (~100 tiles each, ~10^8 band cells): index arithmetic, step budget and tile stitching at a length
    ten times the benchmark's; the oracle needs a few seconds per pair."""
    batch = synth.make_level_batch(2, 100000, members=((1, 3), (1, 3)), seed=77)
    st, ost = _compare(gpu, batch)
    assert ost.cells > 5e7
    assert st.n_relaunched == 0          # the round-2 kernel takes any length (its 16-bit tags are tile-local)


def test_retry_parameters_beyond_flen_4096(gpu):
    """The deferred pass retries with xdrop *= 2 and fLen = min((xdrop * 4) << 1, min(R, Q)) (alignment-cpu.cpp:124-128): on profiles of
    more than 4096 columns that is an fLen above the default cap.  fLen only caps the anti-diagonal width, so the kernels must take it."""
    batch = synth.make_level_batch(4, 6400, members=((1, 4), (1, 4)), seed=77, sub=0.2, indel=0.02)
    minlen = int(batch.len.min())
    assert minlen > 4096
    st, ost = _compare(gpu, batch, xdrop=10000, flen=minlen)
    assert ost.max_width > 500
    # errorType 2 -> fLen = min(int(fLen * 1.2) << 1, min(R, Q)) (:116-119): 9830 for 10 kbp inputs; here capped by the lengths
    _compare(gpu, batch, flen=min(int(4096 * 1.2) << 1, minlen))


# ---- asymmetric matrices: the DP indexes scoreMatrix[l][m] with l = reference letter, m = query letter (TALCO-XDrop.cpp:382) ----
def _blosum(which):
    import json
    import os

    t = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "blosum_tables.json")))[which]
    m = np.zeros((21, 21), dtype=np.float32)
    m[:20, :20] = 5.0 * np.asarray(t, dtype=np.float32)
    return m


@pytest.mark.parametrize("mode", ["dense", "sparse", "presim"])
def test_blosum80_asymmetric_entry_is_indexed_like_the_reference(gpu, mode, monkeypatch):
    """5 x BLOSUM80 as the reference ships it has [I][V] = 15 but [V][I] = 5 (blosum.hpp:65,75).  Profiles rich in I and V make the
    entry count; a kernel that indexed M[m][l] would score these pairs differently from the oracle (which follows the x86 order)."""
    gpu.set_knob(api.KNOB_PROT_MODE, api.PROT_MODES[mode])
    m80 = _blosum("80")
    assert m80[7, 17] == 15 and m80[17, 7] == 5
    batch = synth.make_level_batch(6, 500, members=((1, 4), (1, 4)), seed=80, P=22, sub=0.3)
    # bias the columns towards I (7) on the reference side and V (17) on the query side, keeping the column sums
    rng = np.random.default_rng(1)
    for n in range(batch.n_pairs):
        for side, (src, dst) in enumerate(((17, 7), (7, 17))):
            L = int(batch.len[n, side])
            cols = rng.random(L) < 0.5
            f = batch.freq[n, side, :L]
            tot = f[cols, :20].sum(axis=1)
            f[cols, :20] = 0
            f[cols, dst] = tot * 0.75
            f[cols, src] = tot * 0.25
    _compare(gpu, batch, matrix=m80)
    sw = m80.copy()
    sw[7, 17], sw[17, 7] = sw[17, 7], sw[7, 17]
    a1, n1, _ = gpu.align_batch(gpu.make_params(m80), batch)
    a2, n2, _ = gpu.align_batch(gpu.make_params(sw), batch)
    assert any(n1[i] != n2[i] or not np.array_equal(a1[i, : n1[i]], a2[i, : n2[i]]) for i in range(batch.n_pairs)), \
        "the asymmetric entry did not influence any path: the test would not notice a transposed matrix"


def test_random_asymmetric_nucleotide_matrix(gpu):
    """General 5 x 5 matrix with every entry different (matrix mode 0 of the nucleotide kernel): transposition would show."""
    rng = np.random.default_rng(3)
    mat = rng.integers(-12, 20, size=(5, 5)).astype(np.float32)
    mat[np.arange(5), np.arange(5)] = 18 + np.arange(5)
    assert not np.array_equal(mat, mat.T)
    batch = synth.make_level_batch(8, 900, members=((1, 6), (1, 6)), seed=15)
    _compare(gpu, batch, matrix=mat)
    _compare(gpu, batch, matrix=mat.T.copy())


@pytest.mark.parametrize("where", ["query_row_700", "ref_col_300", "first_row"])
def test_one_profile_entry_outside_the_fast_division_range(gpu, where):
    # The hoisted-reciprocal division of the round-2 kernels is exact only for entries in [2^-20, 2^30]; anything else re-runs on the
    # IEEE-division kernel.  ONE such entry, seen by one wave of the workgroup only, must send the whole pair there -- same result.
    batch = synth.make_level_batch(6, 1400, members=((2, 6), (2, 6)), seed=77)
    f = batch.freq.copy()
    pairs = (1, 4)
    for n in pairs:
        if where == "query_row_700": f[n, 1, 700, 2] = np.float32(3e-9)
        elif where == "ref_col_300": f[n, 0, 300, 1] = np.float32(7e11)
        else: f[n, 1, 0, 0] = np.float32(1e-12)
    b2 = synth.LevelBatch(P=batch.P, seq_len=batch.seq_len, freq=f, gap_open=batch.gap_open, gap_extend=batch.gap_extend, len=batch.len, num=batch.num)
    _compare(gpu, b2)
    # (an entry that large can also widen the band past the 1024-row window first: such a pair is re-run twice, window then division)
    assert gpu.get_stats(0).n_relaunched in (len(pairs), 2 * len(pairs))


def _replicated(batch, n):
    idx = np.arange(n) % batch.n_pairs
    return idx, synth.LevelBatch(P=batch.P, seq_len=batch.seq_len, freq=batch.freq[idx], gap_open=batch.gap_open[idx], gap_extend=batch.gap_extend[idx],
                                 len=batch.len[idx], num=batch.num[idx])


@pytest.mark.parametrize("kind", ["nuc", "prot"])
def test_mid_size_levels_take_speculative_teams_two_to_a_cu(gpu, kind):
    """CUs/2 < pairs <= CUs: two workgroups per pair of the throughput geometry (nucleotide 8 waves x 2 blocks, protein 8 waves on the
    512-row window with precomputed scores), all resident at once.  Several tiles per pair (marker 128); same paths, error codes and
    band cells as the oracle."""
    P = 22 if kind == "prot" else 6
    mat = synth.protein_matrix() if kind == "prot" else M
    base = synth.make_level_batch(24, 700 if kind == "nuc" else 420, P=P, members=((1, 5), (1, 5)), seed=71, length_jitter=0.2)
    idx, big = _replicated(base, 170)
    p = gpu.make_params(mat, marker=128)
    aln, n, err = gpu.align_batch(p, big)
    st = gpu.get_stats(0)
    assert st.speculative == 2 and st.grid == 2 * 170, (st.speculative, st.grid)
    oa, on, oerr, ost = O.align_batch(O.make_params(mat, marker=128), base, threads=8)
    assert np.array_equal(n, on[idx]) and np.array_equal(err, oerr[idx])
    for i in range(big.n_pairs):
        assert np.array_equal(aln[i, : n[i]], oa[idx[i], : on[idx[i]]]), f"pair {i}"
    cells = gpu.get_pair_cells(big.n_pairs)
    _, _, _, one = O.align_batch(O.make_params(mat, marker=128), synth.LevelBatch(P=base.P, seq_len=base.seq_len, freq=base.freq[:1], gap_open=base.gap_open[:1],
                                                                                   gap_extend=base.gap_extend[:1], len=base.len[:1], num=base.num[:1]), threads=1)
    assert int(cells[0]) == one.cells and int(cells[24]) == one.cells


def test_protein_band_wider_than_the_512_row_window_moves_on(gpu):
    """Protein levels with more pairs than CUs start on the 512-row window; with a large X-drop the band of 1.5 kaa pairs outgrows it and
    those pairs must come back, bit-identical, from the 1024-row kernel."""
    mat = synth.protein_matrix()
    base = synth.make_level_batch(6, 1500, P=22, members=((1, 4), (1, 4)), seed=19, sub=0.3)
    idx, big = _replicated(base, 300)
    pk = dict(xdrop=30000)
    p = gpu.make_params(mat, **pk)
    aln, n, err = gpu.align_batch(p, big)
    st = gpu.get_stats(0)
    oa, on, oerr, ost = O.align_batch(O.make_params(mat, **pk), base, threads=8)
    assert ost.max_width > 520, ost.max_width
    assert st.n_relaunched > 0
    assert np.array_equal(n, on[idx]) and np.array_equal(err, oerr[idx])
    for i in range(big.n_pairs):
        assert np.array_equal(aln[i, : n[i]], oa[idx[i], : on[idx[i]]]), f"pair {i}"


def test_nothing_to_align(gpu):
    """Every pair of the call masked out (empty sides): nothing is launched, every pair comes back with length 0 and errorType 0."""
    b = synth.make_level_batch(5, 300, members=(1, 1), seed=3)
    ln = b.len.copy()
    ln[:, 1] = 0
    b0 = synth.LevelBatch(P=b.P, seq_len=b.seq_len, freq=b.freq, gap_open=b.gap_open, gap_extend=b.gap_extend, len=ln, num=b.num)
    aln, n, err = gpu.align_batch(gpu.make_params(M), b0)
    assert np.all(n == 0) and np.all(err == 0)
    assert gpu.get_stats(0).band_cells == 0
