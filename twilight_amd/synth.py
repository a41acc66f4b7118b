"""Synthetic RNASim-shaped level batches for tests and bench.

A *level batch* is what ``parallelAlignmentCPU`` (reference alignment-cpu.cpp:36-183)
hands to the DP for one guide-tree level after profile building, gappy-column
removal and PSGP: for every sibling pair a reference profile ``[R][P]`` and a query
profile ``[Q][P]`` of weighted letter counts (each column sums to the number of
member sequences, alignment-helper.cpp:23-34), position-specific gap penalties
(alignment-helper.cpp:168-219) and the member counts.  The flat layout is the one
``include/twl_align.h`` takes: ``freq[n_pairs][2][seq_len][P]``,
``gap_open/gap_extend[n_pairs][2][seq_len]``, ``len[n_pairs][2]``, ``num[n_pairs][2]``.

Everything here is numpy on the host; nothing touches the GPU or the oracle.
"""
from __future__ import annotations

import dataclasses

import numpy as np

NUC_P = 6
PROT_P = 22

# BLOSUM62 background frequencies (ACDEFGHIKLMNPQRSTVWY order of letterIdx,
# reference scoring-matrix.cpp:40-79), used only to draw synthetic residues.
_AA_BG = np.array([0.0825, 0.0137, 0.0545, 0.0675, 0.0386, 0.0707, 0.0227, 0.0596, 0.0584, 0.0966,
                   0.0242, 0.0406, 0.0470, 0.0393, 0.0553, 0.0656, 0.0534, 0.0687, 0.0108, 0.0292])
_AA_BG = _AA_BG / _AA_BG.sum()


@dataclasses.dataclass
class LevelBatch:
    P: int
    seq_len: int
    freq: np.ndarray        # float32 [n][2][seq_len][P]
    gap_open: np.ndarray    # float32 [n][2][seq_len]
    gap_extend: np.ndarray  # float32 [n][2][seq_len]
    len: np.ndarray         # int32 [n][2]
    num: np.ndarray         # int32 [n][2]

    @property
    def n_pairs(self) -> int:
        return int(self.len.shape[0])


def nucleotide_matrix(match=18.0, mismatch=-8.0, transition=-4.0) -> np.ndarray:
    """Default nucleotide matrix, reference scoring-matrix.cpp:97-110 (5x5, N row/col = 0)."""
    m = np.zeros((5, 5), dtype=np.float32)
    for i in range(5):
        for j in range(5):
            if i == 4 or j == 4:
                m[i, j] = 0.0
            elif i == j:
                m[i, j] = match
            elif abs(i - j) == 2:
                m[i, j] = transition
            else:
                m[i, j] = mismatch
    return m


_BLOSUM62 = [   # ACDEFGHIKLMNPQRSTVWY order of letterIdx (public NCBI BLOSUM62)
    [4, 0, -2, -1, -2, 0, -2, -1, -1, -1, -1, -2, -1, -1, -1, 1, 0, 0, -3, -2],
    [0, 9, -3, -4, -2, -3, -3, -1, -3, -1, -1, -3, -3, -3, -3, -1, -1, -1, -2, -2],
    [-2, -3, 6, 2, -3, -1, -1, -3, -1, -4, -3, 1, -1, 0, -2, 0, -1, -3, -4, -3],
    [-1, -4, 2, 5, -3, -2, 0, -3, 1, -3, -2, 0, -1, 2, 0, 0, -1, -2, -3, -2],
    [-2, -2, -3, -3, 6, -3, -1, 0, -3, 0, 0, -3, -4, -3, -3, -2, -2, -1, 1, 3],
    [0, -3, -1, -2, -3, 6, -2, -4, -2, -4, -3, 0, -2, -2, -2, 0, -2, -3, -2, -3],
    [-2, -3, -1, 0, -1, -2, 8, -3, -1, -3, -2, 1, -2, 0, 0, -1, -2, -3, -2, 2],
    [-1, -1, -3, -3, 0, -4, -3, 4, -3, 2, 1, -3, -3, -3, -3, -2, -1, 3, -3, -1],
    [-1, -3, -1, 1, -3, -2, -1, -3, 5, -2, -1, 0, -1, 1, 2, 0, -1, -2, -3, -2],
    [-1, -1, -4, -3, 0, -4, -3, 2, -2, 4, 2, -3, -3, -2, -2, -2, -1, 1, -2, -1],
    [-1, -1, -3, -2, 0, -3, -2, 1, -1, 2, 5, -2, -2, 0, -1, -1, -1, 1, -1, -1],
    [-2, -3, 1, 0, -3, 0, 1, -3, 0, -3, -2, 6, -2, 0, 0, 1, 0, -3, -4, -2],
    [-1, -3, -1, -1, -4, -2, -2, -3, -1, -3, -2, -2, 7, -1, -2, -1, -1, -2, -4, -3],
    [-1, -3, 0, 2, -3, -2, 0, -3, 1, -2, 0, 0, -1, 5, 1, 0, -1, -2, -2, -1],
    [-1, -3, -2, 0, -3, -2, 0, -3, 2, -2, -1, 0, -2, 1, 5, -1, -1, -3, -3, -2],
    [1, -1, 0, 0, -2, 0, -1, -2, 0, -2, -1, 1, -1, 0, -1, 4, 1, -2, -3, -2],
    [0, -1, -1, -1, -2, -2, -2, -1, -1, -1, -1, 0, -1, -1, -1, 1, 5, 0, -2, -2],
    [0, -1, -3, -2, -1, -3, -3, 3, -2, 1, 1, -3, -2, -2, -3, -2, 0, 4, -3, -1],
    [-3, -2, -4, -3, 1, -2, -2, -3, -3, -2, -1, -4, -4, -2, -3, -3, -2, -3, 11, 2],
    [-2, -2, -3, -2, 3, -3, 2, -1, -2, -1, -1, -2, -3, -1, -2, -2, -2, -1, 2, 7]]


def protein_matrix(wildcard: bool = False) -> np.ndarray:
    """5 x BLOSUM62 with the X row/column 0 (or 5 x mean diagonal with -w), reference scoring-matrix.cpp:112-127."""
    m = np.zeros((21, 21), dtype=np.float32)
    b = np.asarray(_BLOSUM62, dtype=np.float32)
    m[:20, :20] = 5 * b
    if wildcard:
        n = np.float32(np.float32(np.trace(b)) / np.float32(20))
        m[20, :] = 5 * n
        m[:, 20] = 5 * n
    return m


def _mutate(seq: np.ndarray, rng: np.random.Generator, n_letters: int, sub: float, indel: float,
            bg: np.ndarray | None) -> np.ndarray:
    """Substitutions with prob `sub` per site, indel events with prob `indel` per site, geometric length mean 3."""
    s = seq.copy()
    hit = rng.random(s.size) < sub
    if n_letters == 4:
        # transitions (A<->G, C<->T: index +-2) twice as likely as each transversion
        r = rng.random(s.size)
        trans = (s + 2) % 4
        tv1 = (s + 1) % 4
        tv2 = (s + 3) % 4
        new = np.where(r < 0.5, trans, np.where(r < 0.75, tv1, tv2))
    else:
        new = rng.choice(n_letters, size=s.size, p=bg)
    s = np.where(hit, new, s).astype(np.int8)
    if indel <= 0:
        return s
    out = []
    pos = 0
    events = np.flatnonzero(rng.random(s.size) < indel)
    for e in events:
        if e < pos:
            continue
        out.append(s[pos:e])
        length = int(rng.geometric(1.0 / 3.0))
        if rng.random() < 0.5:   # deletion
            pos = min(s.size, e + length)
        else:                    # insertion
            ins = rng.integers(0, n_letters, size=length) if bg is None else rng.choice(n_letters, size=length, p=bg)
            out.append(ins.astype(np.int8))
            pos = e
    out.append(s[pos:])
    return np.concatenate(out).astype(np.int8)


def _psgp(gapcount: np.ndarray, num: int, gap_open: float, gap_extend: float, scale: float):
    """Position-specific gap penalties, reference alignment-helper.cpp:168-219 (double RHS, then float)."""
    g = gapcount.astype(np.float32)
    frac = (np.float32(num) - g).astype(np.float64) * 1.0 / np.float64(num)      # ((num-g)*1.0/num): float sub, double div
    go = np.float64(np.float32(gap_open) * np.float32(scale)) * frac
    ge = np.float64(np.float32(gap_extend)) * frac
    min_go = np.float32(np.float32(gap_open) * np.float32(0.1))
    min_ge = np.float32(np.float32(gap_extend) * np.float32(0.2))
    go = np.minimum(min_go, go.astype(np.float32))
    ge = np.minimum(min_ge, ge.astype(np.float32))
    go = np.where(g > 0, go, np.float32(gap_open)).astype(np.float32)
    ge = np.where(g > 0, ge, np.float32(gap_extend)).astype(np.float32)
    return go, ge


def _side_profile(anc: np.ndarray, n_members: int, rng: np.random.Generator, P: int, sub: float,
                  gap_col_rate: float, weights: bool):
    """A small gap-free-by-construction MSA around `anc` turned into a weighted-count profile [len][P].

    Members differ by substitutions; a fraction of columns carries gaps in a random subset of members.
    Column sums equal n_members (reference alignment-helper.cpp:23-34: w = weight/groupWeight*num)."""
    n_letters = 4 if P == NUC_P else 20
    L = anc.size
    prof = np.zeros((L, P), dtype=np.float32)
    if n_members == 1:
        prof[np.arange(L), anc] = 1.0
        return prof
    w = rng.uniform(0.5, 1.5, size=n_members) if weights else np.ones(n_members)
    w = (w.astype(np.float32) / np.float32(w.astype(np.float32).sum()) * np.float32(n_members)).astype(np.float32)
    bg = None if P == NUC_P else _AA_BG
    gapcols = rng.random(L) < gap_col_rate
    for m in range(n_members):
        s = _mutate(anc, rng, n_letters, sub, 0.0, bg)
        letters = s.astype(np.int64)
        g = gapcols & (rng.random(L) < 0.5)
        letters = np.where(g, P - 1, letters)
        # sequential float32 accumulation, one member at a time (profile[...] += 1.0*w)
        prof[np.arange(L), letters] += w[m]
    return prof


def make_level_batch(n_pairs: int, length: int, *, seed: int = 20260501, P: int = NUC_P, members=(1, 1),
                     sub: float = 0.06, indel: float = 0.005, gap_col_rate: float = 0.03, weights: bool = True,
                     gap_open: float = -50.0, gap_extend: float = -5.0, length_jitter: float = 0.02,
                     pad_to: int | None = None) -> LevelBatch:
    """Build a synthetic level batch of `n_pairs` sibling pairs whose ancestor has ~`length` columns.

    members=(a, b): member-sequence counts (ref, query); ints or (lo, hi) ranges drawn per pair.
    members=(1, 1) gives leaf pairs (one-hot columns); anything larger gives fractional profiles."""
    rng = np.random.default_rng(seed)
    n_letters = 4 if P == NUC_P else 20
    bg = None if P == NUC_P else _AA_BG
    scale = 0.5 if P == NUC_P else 1.0
    sides = []
    lens = np.zeros((n_pairs, 2), dtype=np.int32)
    nums = np.zeros((n_pairs, 2), dtype=np.int32)

    def draw(m):
        if isinstance(m, (tuple, list)):
            return int(rng.integers(m[0], m[1] + 1))
        return int(m)

    for n in range(n_pairs):
        Ln = max(8, int(round(length * (1.0 + rng.uniform(-length_jitter, length_jitter)))))
        root = (rng.integers(0, 4, size=Ln) if P == NUC_P else rng.choice(20, size=Ln, p=bg)).astype(np.int8)
        pair = []
        for side in range(2):
            anc = _mutate(root, rng, n_letters, sub, indel, bg)
            nm = draw(members[side])
            prof = _side_profile(anc, nm, rng, P, sub * 0.5, gap_col_rate, weights)
            pair.append((prof, nm))
            lens[n, side] = prof.shape[0]
            nums[n, side] = nm
        sides.append(pair)
    seq_len = int(lens.max()) if pad_to is None else int(pad_to)
    assert seq_len >= int(lens.max())
    freq = np.zeros((n_pairs, 2, seq_len, P), dtype=np.float32)
    gop = np.zeros((n_pairs, 2, seq_len), dtype=np.float32)
    gex = np.zeros((n_pairs, 2, seq_len), dtype=np.float32)
    for n, pair in enumerate(sides):
        for side, (prof, nm) in enumerate(pair):
            L = prof.shape[0]
            freq[n, side, :L] = prof
            go, ge = _psgp(prof[:, P - 1], nm, gap_open, gap_extend, scale)
            gop[n, side, :L] = go
            gex[n, side, :L] = ge
    return LevelBatch(P=P, seq_len=seq_len, freq=freq, gap_open=gop, gap_extend=gex, len=lens, num=nums)


def path_consumes(aln: np.ndarray, n: int) -> tuple[int, int]:
    """(#ref columns, #query columns) an alignment path consumes: 0 both, 1 query only, 2 ref only."""
    a = aln[:n]
    return int(np.count_nonzero(a != 1)), int(np.count_nonzero(a != 2))


def make_family(n_leaves: int, length: int, *, P: int = NUC_P, seed: int = 1, sub: float = 0.015, indel: float = 0.001,
                sub_range: tuple[float, float] | None = None):
    """A synthetic sequence family for end-to-end runs: (newick, [(name, sequence), ...]).

    A root sequence is evolved down a random binary tree (Yule-like splits); branch lengths are the per-branch
    substitution rates.  Leaves are named s0..s{n-1} in file order.  The defaults are calibrated on the reference's RNASim
    sample (579 x 1.56 kbp -> 3988 columns, band avg 336-421 / max 685): 579 x 1560 here gives ~3750 columns, band avg ~310 / max ~550."""
    rng = np.random.default_rng(seed)
    n_letters = 4 if P == NUC_P else 20
    bg = None if P == NUC_P else _AA_BG
    alphabet = "ACGT" if P == NUC_P else "ACDEFGHIKLMNPQRSTVWY"
    root = (rng.integers(0, 4, size=length) if P == NUC_P else rng.choice(20, size=length, p=bg)).astype(np.int8)
    leaves = []

    def grow(seq, n):
        if n == 1:
            name = f"s{len(leaves)}"
            leaves.append((name, "".join(alphabet[c] for c in seq)))
            return name
        left = int(rng.integers(1, n))
        parts = []
        for k in (left, n - left):
            # per-branch substitution probability: sub x U(0.5, 1.5), or U(lo, hi) when sub_range is given (0.03-0.10)
            b = float(rng.uniform(sub_range[0], sub_range[1])) if sub_range else float(rng.uniform(0.5, 1.5) * sub)
            parts.append(f"{grow(_mutate(seq, rng, n_letters, b, indel, bg), k)}:{b:.6f}")
        return "(" + ",".join(parts) + ")"

    return grow(root, n_leaves) + ";", leaves
