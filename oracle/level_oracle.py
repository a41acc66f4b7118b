"""oracle/level_oracle.py -- CPU restatement (numpy) of the per-pair pre/post-processing around the DP.

TEST INFRASTRUCTURE ONLY: used by tests/ to check the device kernels of include/twl_level.h; the product never imports it.
Each function follows the reference lines it cites (/root/reference/src/alignment-helper.cpp) expression by expression,
including where the reference mixes float and double.

PARITY STATUS: the same arithmetic, in C++ (twilight_amd/csrc/host/helpers.cpp), is what the end-to-end pins of
tests/test_e2e_pin.py run on the CPU side (oracle/e2e_oracle) and those pins reproduce the reference's MSAs byte for byte
(sars_20 and RNASim md5s).  This numpy form is checked against that C++ form by tests/test_host_kats_cpu.py
(profile, PSGP, gappy runs, consensus on shared inputs).  The cached-profile branch (alignment-helper.cpp:16-21,35-40,
506-539) only triggers at >= 1000 sequences per node, which neither pinned dataset reaches: for that branch parity is
"unpinned" against reference outputs and rests on this restatement.
"""
from __future__ import annotations

import numpy as np

F = np.float32
NUC = "ACGT"
AA = "ACDEFGHIKLMNPQRSTVWY"


def letter_idx(seq_type: str, ch: str) -> int:
    """letterIdx(type, toupper(c)) -- scoring-matrix.cpp:26-79."""
    u = ch.upper()
    if seq_type == "n":
        if u in ("-", "."):
            return 5
        return {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3}.get(u, 4)
    if u in ("-", "."):
        return 21
    k = AA.find(u)
    return k if k >= 0 else 20


def lut(seq_type: str) -> np.ndarray:
    return np.array([letter_idx(seq_type, chr(c)) for c in range(256)], dtype=np.int64)


def member_weights(seq_weights, group_weight, num) -> np.ndarray:
    """w = seq.weight / groupWeight * num in fp32 -- alignment-helper.cpp:27."""
    return (np.asarray(seq_weights, dtype=F) / F(group_weight)) * F(num)


def calculate_profile(rows, weights_w, P: int, seq_type: str) -> np.ndarray:
    """alignment-helper.cpp:23-34: profile[t][letterIdx] += w, member by member (fp32 accumulation in member order)."""
    L = len(rows[0]) if rows else 0
    prof = np.zeros((L, P), dtype=F)
    table = lut(seq_type)
    cols = np.arange(L)
    for row, w in zip(rows, weights_w):
        idx = table[np.frombuffer(row, dtype=np.uint8)]
        prof[cols, idx] = prof[cols, idx] + F(w)
    return prof


def profile_from_cache(msa_freq: np.ndarray, group_weight, num) -> np.ndarray:
    """:16-21  profile = msaFreq / weight * num."""
    return (msa_freq.astype(F) / F(group_weight)) * F(num)


def cache_from_profile(profile: np.ndarray, group_weight, num) -> np.ndarray:
    """:35-40  msaFreq = profile / num * weight."""
    return (profile / F(num)) * F(group_weight)


def consensus_idx(profile: np.ndarray) -> np.ndarray:
    """getConsensus :221-241: first strict maximum over the first P-2 letters; all-zero column -> N / X (index P-2)."""
    P = profile.shape[1]
    best = np.full(profile.shape[0], P - 2, dtype=np.int64)
    count = np.zeros(profile.shape[0], dtype=F)
    for j in range(P - 2):
        better = profile[:, j] > count
        best[better] = j
        count[better] = profile[better, j]
    return best


def gappy_mask(profile: np.ndarray, num, thr) -> np.ndarray:
    """removeGappyColumns :84,105: freq[gap] / num > gappyVertical (fp32)."""
    return (profile[:, -1] / F(num)) > F(thr)


def runs_of(mask: np.ndarray):
    runs, start = [], -1
    for i, g in enumerate(mask):
        if g and start < 0:
            start = i
        if not g and start >= 0:
            runs.append((start, i - start))
            start = -1
    if start >= 0:
        runs.append((start, len(mask) - start))
    return runs


def psgp(profile: np.ndarray, num, gap_open, gap_extend, seq_type: str):
    """calculatePSGP :168-219.  The right-hand sides are evaluated in double and narrowed once (:188-189,203-204)."""
    scale = F(0.5) if seq_type == "n" else F(1.0)
    min_ge = F(np.float64(F(gap_extend)) * 0.2)
    min_go = F(np.float64(F(gap_open)) * 0.1)
    g = profile[:, -1].astype(F)
    frac = (np.float64(1.0) * (F(num) - g).astype(np.float64)) / np.float64(num)
    go = np.minimum(min_go, (np.float64(F(gap_open) * scale) * frac).astype(F)).astype(F)
    ge = np.minimum(min_ge, (np.float64(F(gap_extend)) * frac).astype(F)).astype(F)
    pos = g > 0
    return np.where(pos, go, F(gap_open)).astype(F), np.where(pos, ge, F(gap_extend)).astype(F)


def prepare_side(profile: np.ndarray, num, thr, gap_open, gap_extend, seq_type: str):
    """Columns the DP sees for one side: gappy columns removed (unless thr == 1), then [P freq | gapOpen | gapExtend]."""
    mask = gappy_mask(profile, num, thr) if thr != 1.0 else np.zeros(profile.shape[0], dtype=bool)
    kept = profile[~mask]
    go, ge = psgp(kept, num, gap_open, gap_extend, seq_type)
    cols = np.concatenate([kept, go[:, None], ge[:, None]], axis=1).astype(F)
    info = consensus_idx(profile).astype(np.uint8) | (gappy_mask(profile, num, thr).astype(np.uint8) << 7)
    return cols, info, runs_of(mask)


def pairwise_global(s1, s2, matrix: np.ndarray, gap_open, gap_extend):
    """pairwiseGlobal :243-322 on letter-index sequences: affine NW, free leading gaps; traceback prefers M, then Y(1), then X(2)."""
    m, n = len(s1), len(s2)
    go, ge = F(gap_open), F(gap_extend)
    M = np.zeros((m + 1, n + 1), dtype=F)
    X = np.zeros((m + 1, n + 1), dtype=F)
    Y = np.zeros((m + 1, n + 1), dtype=F)
    tb = np.zeros((m + 1, n + 1), dtype=np.int8)
    for i in range(1, m + 1):
        Y[i, 0] = F(-1e9)
        tb[i, 0] = 2
    for j in range(1, n + 1):
        X[0, j] = F(-1e9)
        tb[0, j] = 1
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            base = F(matrix[s1[i - 1], s2[j - 1]])
            M[i, j] = base + max(M[i - 1, j - 1], X[i - 1, j - 1], Y[i - 1, j - 1])
            X[i, j] = max(M[i - 1, j] + go, X[i - 1, j] + ge)
            Y[i, j] = max(M[i, j - 1] + go, Y[i, j - 1] + ge)
            best = max(M[i, j], X[i, j], Y[i, j])
            tb[i, j] = 0 if best == M[i, j] else (1 if best == Y[i, j] else 2)
    path, i, j = [], m, n
    while i > 0 or j > 0:
        d = int(tb[i, j])
        path.append(d)
        if d == 0:
            i -= 1
            j -= 1
        elif d == 1:
            j -= 1
        else:
            i -= 1
    return path[::-1]


def add_gappy_columns_back(path, runs_r, runs_q, cons_r, cons_q, matrix, gap_open, gap_extend):
    """addGappyColumnsBack :324-375.  cons_* are consensus letter-index arrays of the ORIGINAL (un-compacted) sides; an index of
    P-2 (N / X) maps to the wildcard row of the matrix like letterIdx of 'N' / 'X' does."""
    out, r, q, gr, gq = [], 0, 0, 0, 0
    for a in range(len(path) + 1):
        gapR = gr < len(runs_r) and r == runs_r[gr][0]
        gapQ = gq < len(runs_q) and q == runs_q[gq][0]
        if gapR and gapQ:
            lr, lq = runs_r[gr][1], runs_q[gq][1]
            out.extend(pairwise_global(list(cons_r[r: r + lr]), list(cons_q[q: q + lq]), matrix, gap_open, gap_extend))
            gr += 1
            gq += 1
            r += lr
            q += lq
        else:
            if gapR:
                out.extend([2] * runs_r[gr][1])
                r += runs_r[gr][1]
                gr += 1
            if gapQ:
                out.extend([1] * runs_q[gq][1])
                q += runs_q[gq][1]
                gq += 1
        if a < len(path):
            c = int(path[a])
            out.append(c)
            if c == 0:
                r += 1
                q += 1
            elif c == 1:
                q += 1
            else:
                r += 1
    return np.asarray(out, dtype=np.int8)


def apply_path(row: bytes, path: np.ndarray, keep_code: int) -> bytes:
    """updateAlignment :389-400 (keep_code 2, reference side) / :436-447 (keep_code 1, query side)."""
    src = np.frombuffer(row, dtype=np.uint8)
    keep = (path == 0) | (path == keep_code)
    out = np.full(len(path), ord("-"), dtype=np.uint8)
    out[keep] = src[: int(keep.sum())]
    return out.tobytes()


def update_frequency(fr: np.ndarray, fq: np.ndarray, path: np.ndarray, ref_weight, qry_weight) -> np.ndarray:
    """updateFrequency :506-539: merged cached profile along the path."""
    P = fr.shape[1]
    out = np.zeros((len(path), P), dtype=F)
    r = q = 0
    for j, c in enumerate(path):
        if c == 0:
            out[j] = fr[r] + fq[q]
            r += 1
            q += 1
        elif c == 1:
            out[j, : P - 1] = fq[q, : P - 1]
            out[j, P - 1] = F(np.float64(fq[q, P - 1]) + 1.0 * np.float64(F(ref_weight)))
            q += 1
        else:
            out[j, : P - 1] = fr[r, : P - 1]
            out[j, P - 1] = F(np.float64(fr[r, P - 1]) + 1.0 * np.float64(F(qry_weight)))
            r += 1
    return out
