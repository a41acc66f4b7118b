"""Device-resident level kernels (include/twl_level.h) against oracle/level_oracle.py and the DP oracle, bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import level_cases as LC  # noqa: E402
import level_oracle as LO  # noqa: E402
import oracle_lib as O  # noqa: E402

pytestmark = pytest.mark.gpu
F = np.float32


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _level(cases, store_all=False):
    """Flatten cases into one store + one level description.  Returns (seqs, pairs, ids) with ids[i][sd] = sequence ids."""
    from twilight_amd import level as L

    seqs, pairs, ids = [], [], []
    next_cache = 0
    for c in cases:
        pr, idp = [], []
        for sd in range(2):
            s = c.sides[sd]
            k = len(s.rows)
            members = list(range(len(seqs), len(seqs) + k))
            seqs.extend(s.rows)
            side = L.Side(members=members, member_weight=LO.member_weights(s.seq_weights, s.group_weight, k), len=len(s.rows[0]), num=k,
                          weight=float(F(s.group_weight)))
            if store_all:
                side.store_id = next_cache
                next_cache += 1
            pr.append(side)
            idp.append(members)
        pairs.append(pr)
        ids.append(idp)
    return seqs, pairs, ids


@pytest.mark.parametrize("seq_type", ["n", "p"])
def test_prepare_align_commit_match_oracle(gpu, seq_type):
    import twilight_amd as twl
    from twilight_amd import level as L, synth

    cases = [LC.make_case(seq_type, seed, cached=0, length=70 + 31 * seed) for seed in range(7)]
    cases.append(LC.make_case(seq_type, 11, cached=0, length=1500, thr=0.6))       # spans many 256-column chunks
    seqs, pairs, ids = _level(cases)
    M = LC.matrix_of(seq_type)
    p = twl.make_params(M)
    st = L.Store(seqs, seq_type)
    lens, info = st.prepare(p, pairs, gappy_threshold=0.95)
    # gappy threshold is per level in the product; check the 0.6 case in its own level below
    exps = [LC.expected(c) if c.thr == 0.95 else None for c in cases]
    for i, (c, e) in enumerate(zip(cases, exps)):
        if e is None:
            continue
        assert tuple(lens[i]) == e["lens"], f"pair {i}: lengths after gappy removal"
        for sd in range(2):
            assert np.array_equal(_bits(st.columns(i, sd)), _bits(e["cols"][sd])), f"pair {i} side {sd}: packed columns"
            n0 = len(c.sides[sd].rows[0])
            assert np.array_equal(info[i, sd, :n0], e["info"][sd]), f"pair {i} side {sd}: consensus / gappy flags"
    # DP on the prepared columns == DP oracle on the oracle's columns
    aln, n, err = st.align(p)
    op = O.make_params(M)
    for i, e in enumerate(exps):
        if e is None:
            continue
        cr, cq = e["cols"]
        P = st.P
        k_r, k_q = len(cases[i].sides[0].rows), len(cases[i].sides[1].rows)
        oa, oerr, _ = O.align_pair(op, cr[:, :P], cq[:, :P], cr[:, P], cr[:, P + 1], cq[:, P], cq[:, P + 1], k_r, k_q)
        assert err[i] == oerr and n[i] == len(oa) and np.array_equal(aln[i, : n[i]], oa), f"pair {i}: DP path"
    # write-back with the oracle's random paths (gappy columns restored); pair 2 is left out like a deferred pair
    paths = []
    for i, e in enumerate(exps):
        paths.append(np.zeros(0, dtype=np.int8) if (e is None or i == 2) else e["path_full"])
    st.commit(paths)
    rows = st.rows()
    for i, e in enumerate(exps):
        flat = ids[i][0] + ids[i][1]
        if e is None or i == 2:
            for sid in flat:
                assert rows[sid] == seqs[sid], f"pair {i}: untouched row {sid} changed"
            continue
        for sid, want in zip(flat, e["rows_after"]):
            assert rows[sid] == want, f"pair {i}: row {sid} after write-back"
    st.close()

    # the 0.6-threshold, 1500-column case as its own level
    c = cases[-1]
    seqs2, pairs2, ids2 = _level([c])
    st = L.Store(seqs2, seq_type)
    lens, info = st.prepare(p, pairs2, gappy_threshold=0.6)
    e = LC.expected(c)
    assert tuple(lens[0]) == e["lens"]
    for sd in range(2):
        assert np.array_equal(_bits(st.columns(0, sd)), _bits(e["cols"][sd]))
    st.commit([e["path_full"]])
    rows = st.rows()
    for sid, want in zip(ids2[0][0] + ids2[0][1], e["rows_after"]):
        assert rows[sid] == want
    st.close()


@pytest.mark.parametrize("seq_type", ["n", "p"])
def test_cached_profiles_two_levels(gpu, seq_type):
    """Level 1 stores both nodes' profiles (as for >= 1000-sequence nodes) and merges them on commit; level 2 uses the merged cache."""
    import twilight_amd as twl
    from twilight_amd import level as L

    a = LC.make_case(seq_type, 21, cached=0, length=300)
    seqs, pairs, ids = _level([a], store_all=True)
    z_rows = LC.make_case(seq_type, 22, cached=0, length=340).sides[1]
    z_ids = list(range(len(seqs), len(seqs) + len(z_rows.rows)))
    seqs = seqs + z_rows.rows
    p = twl.make_params(LC.matrix_of(seq_type))
    st = L.Store(seqs, seq_type)
    st.prepare(p, pairs)
    P = st.P
    profs = [LC.side_profile(a, sd) for sd in range(2)]
    caches = [LO.cache_from_profile(profs[sd], a.sides[sd].group_weight, len(a.sides[sd].rows)) for sd in range(2)]
    for sd in range(2):
        assert np.array_equal(_bits(st.cache(sd)), _bits(caches[sd])), f"stored cache of side {sd}"
    e = LC.expected(a)
    st.commit([e["path_full"]])
    merged = LO.update_frequency(caches[0], caches[1], e["path_full"], a.sides[0].group_weight, a.sides[1].group_weight)
    assert np.array_equal(_bits(st.cache(0)), _bits(merged)), "merged cache"
    with pytest.raises(twl.TwlError):
        st.cache(1)                                  # the query node's cache is dropped by the merge
    # level 2: (merged node, Z).  The merged node's profile comes from its cache; Z's is stored.
    k = len(a.sides[0].rows) + len(a.sides[1].rows)
    gw = float(F(F(a.sides[0].group_weight) + F(a.sides[1].group_weight)))
    total = len(e["path_full"])
    w_all = np.concatenate([a.sides[0].seq_weights, a.sides[1].seq_weights])
    ref = L.Side(members=ids[0][0] + ids[0][1], member_weight=LO.member_weights(w_all, gw, k), len=total, num=k, weight=gw, cache_id=0)
    kz = len(z_rows.rows)
    qry = L.Side(members=z_ids, member_weight=LO.member_weights(z_rows.seq_weights, z_rows.group_weight, kz), len=len(z_rows.rows[0]), num=kz,
                 weight=float(F(z_rows.group_weight)), store_id=7)
    lens, info = st.prepare(p, [[ref, qry]])
    prof_ref = LO.profile_from_cache(merged, gw, k)
    cols, _, _ = LO.prepare_side(prof_ref, k, 0.95, LC.GAP_OPEN, LC.GAP_EXTEND, seq_type)
    assert lens[0, 0] == cols.shape[0]
    assert np.array_equal(_bits(st.columns(0, 0)), _bits(cols)), "columns from the merged cache"
    prof_z = LO.calculate_profile(z_rows.rows, LO.member_weights(z_rows.seq_weights, z_rows.group_weight, kz), P, seq_type)
    assert np.array_equal(_bits(st.cache(7)), _bits(LO.cache_from_profile(prof_z, z_rows.group_weight, kz)))
    st.close()


def test_rows_grow_past_initial_capacity(gpu):
    """A commit whose paths are longer than the store's row pitch re-pitches the planes without losing rows."""
    import twilight_amd as twl
    from twilight_amd import level as L

    rng = np.random.default_rng(5)
    N = 20
    seqs = [bytes(rng.choice(list(b"ACGT"), size=400).astype(np.uint8)) for _ in range(N)] + [b"ACGTAC"]
    st = L.Store(seqs, "n")                      # initial pitch 8 x 400 + 256 = 3456 columns for 21 sequences of 400 letters
    p = twl.make_params(LC.matrix_of("n"))
    members, total = [0], 400
    for nxt in range(1, N):                      # end-to-end concatenation: 800, 1200, ... 8000 columns (past the pitch from 3600 on)
        k = len(members)
        st.prepare(p, [[L.Side(list(members), np.full(k, 1.0, dtype=F), total, k, float(k)), L.Side([nxt], np.asarray([1.0], dtype=F), 400, 1, 1.0)]],
                   gappy_threshold=1.0)
        st.commit([np.concatenate([np.full(total, 2, np.int8), np.full(400, 1, np.int8)])])
        members.append(nxt)
        total += 400
    rows = st.rows()
    for i in range(N):
        assert rows[i] == b"-" * (400 * i) + seqs[i] + b"-" * (400 * (N - 1 - i)), f"row {i}"
    assert rows[N] == seqs[N]
    st.close()


def test_level_edge_cases(gpu):
    """Empty side, run masks, a level that commits nothing, and argument errors of the level API."""
    import twilight_amd as twl
    from twilight_amd import level as L

    rng = np.random.default_rng(9)
    mk = lambda n: bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8))
    seqs = [mk(120), mk(130), mk(90), b"", mk(64), mk(64)]
    one = np.asarray([1.0], dtype=F)
    p = twl.make_params(LC.matrix_of("n"))
    st = L.Store(seqs, "n")
    pairs = [[L.Side([0], one, 120, 1, 1.0), L.Side([1], one, 130, 1, 1.0)],
             [L.Side([2], one, 90, 1, 1.0), L.Side([3], one, 0, 1, 1.0)],          # empty query side (alignment-cpu.cpp:89-90)
             [L.Side([4], one, 64, 1, 1.0), L.Side([5], one, 64, 1, 1.0)]]
    lens, info = st.prepare(p, pairs)
    assert lens.tolist() == [[120, 130], [90, 0], [64, 64]]
    aln, n, err = st.align(p, run_mask=np.array([1, 1, 0], dtype=np.uint8))
    assert n[0] > 0 and n[1] == 0 and n[2] == 0 and not err.any()                 # empty side and masked pair: no path, no error
    aln2, n2, err2 = st.align(p, run_mask=np.array([0, 0, 1], dtype=np.uint8))    # second call on the same prepared level
    assert n2[0] == 0 and n2[2] > 0
    full = st.align(p)
    assert np.array_equal(full[0][0, : n[0]], aln[0, : n[0]]) and np.array_equal(full[0][2, : n2[2]], aln2[2, : n2[2]])
    # commit: pair 0 its path, pair 1 the all-gap path the caller builds for an empty side, pair 2 deferred
    st.commit([aln[0, : n[0]], np.full(90, 2, np.int8), np.zeros(0, np.int8)])
    rows = st.rows()
    assert len(rows[0]) == n[0] == len(rows[1]) and rows[0].replace(b"-", b"") == seqs[0] and rows[1].replace(b"-", b"") == seqs[1]
    assert rows[2] == seqs[2] and rows[3] == b"-" * 90 and rows[4] == seqs[4] and rows[5] == seqs[5]
    # a level without any pair to commit, then errors
    st.prepare(p, [[L.Side([4], one, 64, 1, 1.0), L.Side([5], one, 64, 1, 1.0)]])
    st.commit([np.zeros(0, np.int8)])
    with pytest.raises(twl.TwlError):
        st.align(p)                                                               # nothing prepared any more
    with pytest.raises(twl.TwlError):
        st.prepare(p, [[L.Side([0], one, 120, 1, 1.0), L.Side([1], one, 130, 1, 1.0)]])   # rows 0/1 are n[0] long now
    with pytest.raises(twl.TwlError):
        st.prepare(p, [[L.Side([4], one, 64, 1, 1.0), L.Side([99], one, 64, 1, 1.0)]])
    with pytest.raises(twl.TwlError):
        st.prepare(p, [[L.Side([4], one, 64, 1, 1.0, cache_id=5), L.Side([5], one, 64, 1, 1.0)]])
    with pytest.raises(twl.TwlError):
        st.prepare(twl.make_params(LC.matrix_of("p")), [[L.Side([4], one, 64, 1, 1.0), L.Side([5], one, 64, 1, 1.0)]])
    st.close()


@pytest.mark.parametrize("seq_type", ["n", "p"])
def test_commit_from_dp_output_equals_commit_of_downloaded_paths(gpu, seq_type):
    """Pairs in which no gappy column was removed keep their DP path in HBM: twl_level_align without a path buffer, twl_level_commit_from_dp
    with those pairs marked (one pair fetched with twl_level_read_path and brought back by the caller, one left out like a deferred
    pair) must leave exactly the rows and lengths that the download / upload form leaves."""
    import twilight_amd as twl
    from twilight_amd import level as L

    cases = [LC.make_case(seq_type, seed, cached=0, length=90 + 53 * seed) for seed in range(6)]
    seqs, pairs, ids = _level(cases)
    p = twl.make_params(LC.matrix_of(seq_type))
    # gappy threshold 1: nothing is removed, so every DP path is a final path
    a = L.Store(seqs, seq_type)
    a.prepare(p, pairs, gappy_threshold=1.0)
    aln, n, err = a.align(p)
    assert np.all(err == 0) and np.all(n > 0)
    full = [aln[i, : n[i]].copy() for i in range(len(cases))]
    full[3] = np.zeros(0, dtype=np.int8)
    a.commit(full)
    want = a.rows()
    a.close()

    b = L.Store(seqs, seq_type)
    b.prepare(p, pairs, gappy_threshold=1.0)
    n2, err2 = b.align_in_hbm(p)
    assert np.array_equal(n2, n) and np.array_equal(err2, err)
    fetched = b.read_path(1, int(n2[1]))
    assert np.array_equal(fetched, aln[1, : n[1]])
    plen = [int(x) for x in n2]
    plen[3] = 0
    b.commit_from_dp([None, fetched, None, None, None, None], plen)
    got = b.rows()
    b.close()
    assert got == want


def test_geometry_follows_the_pairs_that_run(gpu):
    """A rank of a sharded run (or one gap-character group) aligns a masked share of the level: the kernel geometry must be chosen by the
    number of pairs that RUN -- 10 of 300 pairs take the speculative two-workgroup kernel (20 workgroups), not the throughput kernel
    the whole level would take -- and the masked-out pairs come back with length 0 and errorType 0."""
    import twilight_amd as twl
    from twilight_amd import level as L

    cases = [LC.make_case("n", seed, cached=0, length=80 + (seed % 7) * 9) for seed in range(300)]
    seqs, pairs, ids = _level(cases)
    p = twl.make_params(LC.matrix_of("n"))
    st = L.Store(seqs, "n")
    st.prepare(p, pairs, gappy_threshold=0.95)
    aln_all, n_all, err_all = st.align(p)
    s_all = twl.get_stats(0)
    assert s_all.speculative == 0 and s_all.grid > 256
    mask = np.zeros(len(cases), dtype=np.uint8)
    mask[5:15] = 1
    aln_m, n_m, err_m = st.align(p, run_mask=mask)
    s_m = twl.get_stats(0)
    assert s_m.speculative == 1 and s_m.grid == 20, (s_m.speculative, s_m.grid)
    for i in range(len(cases)):
        if mask[i]:
            assert err_m[i] == err_all[i] and n_m[i] == n_all[i] and np.array_equal(aln_m[i, : n_m[i]], aln_all[i, : n_all[i]]), f"pair {i}"
        else:
            assert n_m[i] == 0 and err_m[i] == 0, f"masked-out pair {i}"
    st.close()


def test_row_planes_fall_back_to_the_minimal_pitch_together(gpu):
    """grow_rows: when an allocation at the generous pitch fails, BOTH planes are retried at the pitch that is needed (one pitch for
    both planes); a failure there too is reported, not half applied.  The write-back that follows re-pitches again and must still be
    right."""
    import twilight_amd as twl
    from twilight_amd import api, level as L

    cases = [LC.make_case("n", seed, cached=0, length=90 + 17 * seed) for seed in range(5)]
    seqs, pairs, ids = _level(cases)
    p = twl.make_params(LC.matrix_of("n"))
    twl.set_knob(api.KNOB_FAIL_ROW_ALLOCS, 2)       # generous pitch fails, and so does the retry
    with pytest.raises(Exception):
        L.Store(seqs, "n")
    twl.set_knob(api.KNOB_FAIL_ROW_ALLOCS, 1)       # generous pitch fails on the first plane: both planes at the minimal pitch
    st = L.Store(seqs, "n")
    twl.set_knob(api.KNOB_FAIL_ROW_ALLOCS, 0)
    st.prepare(p, pairs, gappy_threshold=0.95)
    exps = [LC.expected(c) for c in cases]
    st.commit([e["path_full"] for e in exps])       # longer rows than the minimal pitch holds: grows again
    rows = st.rows()
    for i, e in enumerate(exps):
        for sid, want in zip(ids[i][0] + ids[i][1], e["rows_after"]):
            assert rows[sid] == want, f"pair {i}: row {sid} after write-back"
    st.close()


@pytest.mark.parametrize("seq_type", ["n", "p"])
def test_gappy_columns_back_on_the_device(gpu, seq_type):
    """twl_level_restore == addGappyColumnsBack + pairwiseGlobal (alignment-helper.cpp:324-375, :243-322) as oracle/level_oracle.py walks
    them, on the DP paths the device itself produced; the write-back that follows takes the restored rows straight from HBM."""
    import twilight_amd as twl
    from twilight_amd import level as L

    cases = [LC.make_case(seq_type, seed, cached=0, length=60 + 41 * seed, thr=(0.6 if seed % 3 == 0 else 0.95)) for seed in range(1, 9)]
    cases = [c for c in cases if c.thr == 0.6] + [LC.make_case(seq_type, 40 + k, cached=0, length=900 + 300 * k, thr=0.6) for k in range(3)]
    seqs, pairs, ids = _level(cases)
    p = twl.make_params(LC.matrix_of(seq_type))
    st = L.Store(seqs, seq_type)
    lens, info = st.prepare(p, pairs, gappy_threshold=0.6)
    n, err = st.align_in_hbm(p)
    assert not err.any()
    stride = max(len(c.sides[0].rows[0]) + len(c.sides[1].rows[0]) for c in cases)
    lost = [i for i, c in enumerate(cases) if tuple(lens[i]) != (len(c.sides[0].rows[0]), len(c.sides[1].rows[0]))]
    assert len(lost) >= len(cases) - 1      # a 0.6 threshold removes columns nearly everywhere
    fin = st.restore(p, lost, stride)
    assert (fin > 0).all(), fin
    exps = {}
    for t, i in enumerate(lost):
        e = LC.expected(cases[i], path_wo_gc=st.read_path(i, int(n[i])))
        exps[i] = e
        got = st.read_final(i, int(fin[t]))
        assert len(e["path_full"]) == fin[t] and np.array_equal(got, e["path_full"]), f"pair {i}: restored path"
    plen = [int(fin[lost.index(i)]) if i in exps else int(n[i]) for i in range(len(cases))]
    st.commit_from_dp([None] * len(cases), plen, stride=stride, restored=lost)
    rows = st.rows()
    for i, e in exps.items():
        for sid, want in zip(ids[i][0] + ids[i][1], e["rows_after"]):
            assert rows[sid] == want, f"pair {i}: row {sid} after write-back"
    st.close()


def test_rows_and_cached_profiles_of_a_subtree_move_between_stores(gpu):
    """What a sharded run exchanges where its subtrees meet (host/align_owned.cpp): the current rows of a list of sequences out of one store (host block and
    device block) and into another (both ways), and a cached profile under an id new to the receiving store."""
    import twilight_amd as twl
    from twilight_amd import level as L

    rng = np.random.default_rng(5)
    seqs = [bytes(rng.choice(list(b"ACGT-"), size=int(rng.integers(40, 400))).astype(np.uint8)) for _ in range(23)]
    a, b = L.Store(seqs, "n"), L.Store([s[:7] for s in seqs], "n")          # b holds stubs: the rows arrive from a
    ids = [3, 22, 0, 11, 7]
    got = a.rows_of(ids)
    assert got == [seqs[i] for i in ids] and a.rows_to_block(ids) == got
    longer = [s + b"-" * 1500 for s in got]                                   # longer than b's row pitch: the planes grow
    b.write_rows(ids[:3], longer[:3])
    b.write_rows(ids[3:], longer[3:], via_device_block=True)
    rows = b.rows()
    for k, i in enumerate(ids):
        assert rows[i] == longer[k]
    for i in set(range(23)) - set(ids):
        assert rows[i] == seqs[i][:7]
    prof = rng.random((321, 6)).astype(np.float32)
    b.write_cache(40, prof)
    assert np.array_equal(b.cache(40), prof)
    with pytest.raises(Exception):
        b.write_cache(40, prof)                                               # the id is taken
    a.close(); b.close()


@pytest.mark.parametrize("regime", ["throughput", "few", "tile_parallel", "protein_few", "protein_many"])
def test_one_launch_takes_pairs_of_both_gap_character_kinds(gpu, regime):
    """The reference decides gapCharScore per pair (alignment-cpu.cpp:88: 0 when a side holds more than 10 000 sequences).  twl_level_align_mixed
    takes the pairs of both kinds in ONE launch; every pair must come out as in the call of its own kind, on each kernel family the level can take."""
    import twilight_amd as twl
    from twilight_amd import level as L

    seq_type = "p" if regime.startswith("protein") else "n"
    if regime == "throughput":
        cases = [LC.make_case("n", seed, cached=0, length=90 + (seed % 11) * 7) for seed in range(1100)]
    elif regime == "few":
        cases = [LC.make_case("n", seed, cached=0, length=200 + 13 * seed) for seed in range(10)]
    elif regime == "tile_parallel":
        cases = [LC.make_case("n", seed, cached=0, length=1700 + 40 * seed) for seed in range(6)]
    elif regime == "protein_few":
        cases = [LC.make_case("p", seed, cached=0, length=150 + 20 * seed) for seed in range(8)]
    else:
        cases = [LC.make_case("p", seed, cached=0, length=80 + (seed % 9) * 6) for seed in range(300)]
    seqs, pairs, ids = _level(cases)
    p = twl.make_params(LC.matrix_of(seq_type), marker=512 if regime == "tile_parallel" else 1024)
    p0 = twl.make_params(LC.matrix_of(seq_type), marker=p.marker, gap_char=0.0)
    st = L.Store(seqs, seq_type)
    st.prepare(p, pairs, gappy_threshold=0.95)
    n = len(cases)
    zero = (np.arange(n) % 3 == 1).astype(np.uint8)
    aln_p, n_p, err_p = st.align(p, run_mask=1 - zero)
    aln_z, n_z, err_z = st.align(p0, run_mask=zero)
    aln_a, n_a, err_a = st.align(p0)              # every pair with score 0: how many pairs the score matters for
    aln_m, n_m, err_m = st.align(p, zero_gap=zero)
    if regime == "tile_parallel":
        assert twl.get_stats(0).speculative == 3
    differ = 0
    for i in range(n):
        want_a, want_n, want_e = (aln_z, n_z, err_z) if zero[i] else (aln_p, n_p, err_p)
        assert err_m[i] == want_e[i] and n_m[i] == want_n[i] and np.array_equal(aln_m[i, : n_m[i]], want_a[i, : want_n[i]]), f"pair {i} (zero_gap {zero[i]})"
        if not zero[i]:
            differ += int(n_a[i] != n_p[i] or not np.array_equal(aln_a[i, : n_a[i]], aln_p[i, : n_p[i]]))
    assert differ > 0, "the gap-character score changes no path of this level: the test does not discriminate"
    # a mask of zeros is the plain call; all ones is the call with gap_char 0
    aln_1, n_1, err_1 = st.align(p, zero_gap=np.ones(n, dtype=np.uint8))
    assert np.array_equal(n_1, n_a) and np.array_equal(err_1, err_a) and all(np.array_equal(aln_1[i, : n_1[i]], aln_a[i, : n_a[i]]) for i in range(n))
    st.close()


def test_leaf_level_runs_the_step_without_gap_terms_and_division(gpu):
    """A level whose pairs are all single sequences on both sides (the bottom of every guide tree) runs talco_lean_kernel<..., SP = 1>: no gap-letter terms, no
    division, none of their per-block tests.  Same paths, lengths, error codes and band cells as the general step (TWL_KNOB_LEAF_STEP 0) and as the oracle;
    a level with ONE two-sequence side does not take it."""
    import twilight_amd as twl
    from twilight_amd import api, level as L, synth

    rng = np.random.default_rng(5)
    n = 600                                        # more pairs than CUs: the throughput launch (+ its tile-parallel remainder when the last round is badly filled)
    M = LC.matrix_of("n")
    p = twl.make_params(M)

    def leaf_pairs(extra_member):
        seqs, pairs = [], []
        for i in range(n):
            a = "".join("ACGT"[c] for c in rng.integers(0, 4, size=900 + int(rng.integers(0, 60))))
            b = list(a)
            for _ in range(25):
                b[int(rng.integers(0, len(b)))] = "ACGTN"[int(rng.integers(0, 5))]
            if i % 3 == 0: del b[100:104]
            b = "".join(b)
            ids_a = [len(seqs)]; seqs.append(a.encode())
            ids_b = [len(seqs)]; seqs.append(b.encode())
            if extra_member and i == 17:
                ids_a.append(len(seqs)); seqs.append(a.encode())      # (same length as its row mate)
            sides = [L.Side(members=m, member_weight=[1.0] * len(m), len=len(seqs[m[0]]), num=len(m), weight=float(len(m))) for m in (ids_a, ids_b)]
            pairs.append(sides)
        return seqs, pairs

    seqs, pairs = leaf_pairs(False)
    st = L.Store(seqs, "n")
    st.prepare(p, pairs, gappy_threshold=0.95)
    aln1, n1, err1 = st.align(p)
    s1 = twl.get_stats(0)
    assert b", 0, 1>" in s1.kernel and s1.matrix_mode == 5, s1.kernel
    twl.set_knob(api.KNOB_LEAF_STEP, 0)
    try:
        aln0, n0, err0 = st.align(p)
        s0 = twl.get_stats(0)
    finally:
        twl.set_knob(api.KNOB_LEAF_STEP, 1)
    assert b", 0, 1>" not in s0.kernel, s0.kernel
    assert np.array_equal(n1, n0) and np.array_equal(err1, err0) and np.array_equal(aln1, aln0) and s1.band_cells == s0.band_cells
    # (against the oracle: every end-to-end pin of tests/test_e2e_pin.py and tests/test_gpu_variants.py starts with such a level)
    st.close()
    seqs, pairs = leaf_pairs(True)
    st = L.Store(seqs, "n")
    st.prepare(p, pairs, gappy_threshold=0.95)
    st.align(p)
    assert b", 0, 1>" not in twl.get_stats(0).kernel
    st.close()
