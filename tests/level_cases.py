"""Seeded pair cases for the level pre/post-processing tests, and their expected values from oracle/level_oracle.py."""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import level_oracle as LO  # noqa: E402
from twilight_amd import synth  # noqa: E402

F = np.float32
GAP_OPEN, GAP_EXTEND = -50.0, -5.0


@dataclass
class SideCase:
    rows: List[bytes]
    seq_weights: np.ndarray          # per-sequence weights (SequenceInfo::weight)
    group_weight: float              # Node::alnWeight
    cache: Optional[np.ndarray] = None   # Node::msaFreq when the node carries a cached profile


@dataclass
class PairCase:
    seq_type: str
    thr: float
    sides: List[SideCase] = field(default_factory=list)
    seed: int = 0

    @property
    def P(self):
        return 6 if self.seq_type == "n" else 22


def _rows(rng, seq_type, k, length, lead_gap, gap_cols):
    alpha = "ACGT" if seq_type == "n" else LO.AA
    odd = "NRY" if seq_type == "n" else "XBZ"
    base = rng.choice(list(alpha), size=length)
    rows = []
    for _ in range(k):
        r = base.copy()
        mut = rng.random(length) < 0.15
        r[mut] = rng.choice(list(alpha), size=int(mut.sum()))
        amb = rng.random(length) < 0.02
        r[amb] = rng.choice(list(odd), size=int(amb.sum()))
        low = rng.random(length) < 0.1
        r = np.array([c.lower() if l else c for c, l in zip(r, low)])
        gaps = rng.random(length) < 0.08
        r[gaps] = "-"
        r[gap_cols] = "-"                       # columns that are gaps in every member: removed as gappy
        r[:lead_gap] = "-"
        rows.append("".join(r).encode())
    return rows


def make_case(seq_type: str, seed: int, cached: int = 0, length: int = 80, thr: Optional[float] = None) -> PairCase:
    """cached: 0 = no cached profiles, 1 = the reference side carries one, 2 = both sides do."""
    rng = np.random.default_rng(1000 * seed + (7 if seq_type == "n" else 13))
    case = PairCase(seq_type=seq_type, thr=(1.0 if seed % 4 == 3 else 0.95) if thr is None else thr, seed=seed)
    lead = 3 if seed % 2 == 0 else 0              # both sides start with a gappy run -> pairwiseGlobal on the consensus substrings
    for sd in range(2):
        k = int(rng.integers(1, 7))
        L = length + (5 * sd)
        gap_cols = np.zeros(L, dtype=bool)
        for _ in range(3):
            s = int(rng.integers(lead + 2, L - 6))
            gap_cols[s: s + int(rng.integers(1, 4))] = True
        rows = _rows(rng, seq_type, k, L, lead, gap_cols)
        w = rng.uniform(0.5, 2.0, size=k).astype(F)
        gw = F(0)
        for x in w:
            gw = F(gw + x)
        side = SideCase(rows=rows, seq_weights=w, group_weight=float(gw))
        if cached == 2 or (cached == 1 and sd == 0):
            prof = LO.calculate_profile(rows, LO.member_weights(w, gw, k), case.P, seq_type)
            side.cache = LO.cache_from_profile(prof, gw, k)
        case.sides.append(side)
    return case


def matrix_of(seq_type):
    return synth.nucleotide_matrix() if seq_type == "n" else synth.protein_matrix()


def random_path(rng, r_len, q_len):
    path, r, q = [], 0, 0
    while r < r_len or q < q_len:
        c = int(rng.choice([0, 0, 0, 0, 1, 2]))
        if c == 0 and r < r_len and q < q_len:
            r += 1
            q += 1
        elif c == 1 and q < q_len:
            q += 1
        elif c == 2 and r < r_len:
            r += 1
        else:
            continue
        path.append(c)
    return np.asarray(path, dtype=np.int8)


def side_profile(case: PairCase, sd: int) -> np.ndarray:
    s = case.sides[sd]
    k = len(s.rows)
    if s.cache is not None:
        return LO.profile_from_cache(s.cache, s.group_weight, k)
    return LO.calculate_profile(s.rows, LO.member_weights(s.seq_weights, s.group_weight, k), case.P, case.seq_type)


def expected(case: PairCase, path_wo_gc: Optional[np.ndarray] = None) -> dict:
    store = any(s.cache is not None for s in case.sides)           # storeFreq, alignment-helper.cpp:14
    out = {"cols": [], "info": [], "runs": [], "cons": [], "cache_after_prepare": []}
    for sd in range(2):
        s = case.sides[sd]
        k = len(s.rows)
        prof = side_profile(case, sd)
        cols, info, runs = LO.prepare_side(prof, k, case.thr, GAP_OPEN, GAP_EXTEND, case.seq_type)
        out["cols"].append(cols)
        out["info"].append(info)
        out["runs"].append(runs)
        out["cons"].append(LO.consensus_idx(prof))
        out["cache_after_prepare"].append(s.cache if s.cache is not None else (LO.cache_from_profile(prof, s.group_weight, k) if store else None))
    out["lens"] = (out["cols"][0].shape[0], out["cols"][1].shape[0])
    if path_wo_gc is None:
        path_wo_gc = random_path(np.random.default_rng(99 + case.seed), *out["lens"])
    out["path_wo_gc"] = path_wo_gc
    full = LO.add_gappy_columns_back(path_wo_gc, out["runs"][0], out["runs"][1], out["cons"][0], out["cons"][1], matrix_of(case.seq_type), GAP_OPEN, GAP_EXTEND)
    out["path_full"] = full
    out["rows_after"] = [LO.apply_path(r, full, 2) for r in case.sides[0].rows] + [LO.apply_path(r, full, 1) for r in case.sides[1].rows]
    c0, c1 = out["cache_after_prepare"]
    out["merged"] = LO.update_frequency(c0, c1, full, case.sides[0].group_weight, case.sides[1].group_weight) if (c0 is not None and c1 is not None) else None
    return out
