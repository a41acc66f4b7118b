"""oracle/msa_replay.py -- TEST INFRASTRUCTURE ONLY.  An independent replay of TWILIGHT's progressive alignment AFTER the level schedule.

oracle/e2e_oracle (the CPU checker every "product == checker" test uses) is built from the product's own host sources plus the
oracle DP, so on its own it cannot tell whether those host sources are right.  This module restates, in numpy and straight from the
REFERENCE source, everything the host does between the schedule and the final rows, sharing no code with
twilight_amd/csrc/host/{helpers,progressive,align_*}.cpp:

    progressive::updateNode            /root/reference/src/progressive.cpp:126-172
    cpu::parallelAlignmentCPU          /root/reference/src/alignment-cpu.cpp:36-183   (policy: empty sides, low-quality singletons, defer /
                                                                                        retry, gapCharScore rule)
    calculateProfile / getConsensus / removeGappyColumns / calculatePSGP / addGappyColumnsBack / pairwiseGlobal
                                       alignment-helper.cpp:8-375   (oracle/level_oracle.py)
    updateFrequency / updateAlignment (with the > 1000-member compression into a negative id) / fallback2cpu
                                       alignment-helper.cpp:377-591
    msaOnSubtree: result to the root, expansion of compressed members, the deferred pass
                                       progressive.cpp:194-299
    msa::Params (built-in matrices)    scoring-matrix.cpp:81-135

The inputs are the sequences (with weights and quality flags), the subtree's nodes and the level batches as oracle/schedule_dump
prints them -- the tree / reroot / scheduling part is validated separately, through the pairs-per-level the reference recorded for its
two sample datasets.  The DP itself is the C oracle (oracle/libtwl_oracle.so).

    python oracle/msa_replay.py dump.json out.aln [--remove-gappy 0.95] [--wildcard] [--blosum 62] [--filter] ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))

import level_oracle as L  # noqa: E402
import oracle_lib as O  # noqa: E402

F = np.float32


def libstdcxx_sort(v, less):
    """std::sort as libstdc++ implements it (bits/stl_algo.h: introsort with a median-of-three pivot, then insertion sort; threshold 16).
    The reference orders the deferred profiles with std::sort and a comparator that has ties (progressive.cpp:281-284); the order
    of tied elements is unspecified by the standard but fixed by this algorithm, and it decides which profile joins the root first."""
    def linear_insert(last):
        val = v[last]
        nxt = last - 1
        while less(val, v[nxt]):
            v[last] = v[nxt]
            last = nxt
            nxt -= 1
        v[last] = val

    def insertion(first, last):
        for i in range(first + 1, last):
            if less(v[i], v[first]):
                val = v[i]
                v[first + 1: i + 1] = v[first: i]
                v[first] = val
            else:
                linear_insert(i)

    def median_to_first(result, a, b, c):
        if less(v[a], v[b]):
            if less(v[b], v[c]):
                k = b
            elif less(v[a], v[c]):
                k = c
            else:
                k = a
        elif less(v[a], v[c]):
            k = a
        elif less(v[b], v[c]):
            k = c
        else:
            k = b
        v[result], v[k] = v[k], v[result]

    def partition(first, last, pivot):
        while True:
            while less(v[first], v[pivot]):
                first += 1
            last -= 1
            while less(v[pivot], v[last]):
                last -= 1
            if not first < last:
                return first
            v[first], v[last] = v[last], v[first]
            first += 1

    def introsort(first, last, depth):
        while last - first > 16:
            if depth == 0:
                raise NotImplementedError("heapsort fallback of std::sort (not reached at these sizes)")
            depth -= 1
            mid = first + (last - first) // 2
            median_to_first(first, first + 1, mid, last - 1)
            cut = partition(first + 1, last, first)
            introsort(cut, last, depth)
            last = cut

    n = len(v)
    if n:
        introsort(0, n, 2 * (n.bit_length() - 1))
        if n > 16:
            insertion(0, 16)
            for i in range(16, n):
                linear_insert(i)
        else:
            insertion(0, n)
    return v


class Node:
    def __init__(self, ident, leaf, grp, children):
        self.id, self.leaf, self.grp, self.children = ident, bool(leaf), grp, children
        self.seqs = []            # seqsIncluded
        self.freq = None          # msaFreq (None = empty)
        self.aln_len = 0
        self.aln_num = 0
        self.aln_weight = F(0)


def nucleotide_matrix(match, mismatch, transition, wildcard):
    """scoring-matrix.cpp:101-110."""
    m = np.zeros((5, 5), dtype=F)
    for i in range(5):
        for j in range(5):
            if i == 4 or j == 4:
                m[i, j] = match if wildcard else 0.0
            elif i == j:
                m[i, j] = match
            elif abs(i - j) == 2:
                m[i, j] = transition
            else:
                m[i, j] = mismatch
    return m


def protein_matrix(which, wildcard):
    """scoring-matrix.cpp:112-135; the tables are the fixture tests/golden/blosum_tables.json (= blosum.hpp)."""
    tabs = json.load(open(os.path.join(os.path.dirname(HERE), "tests", "golden", "blosum_tables.json")))
    if str(which) not in tabs:
        which = 62
    m = np.zeros((21, 21), dtype=F)
    nscore = F(0)
    for i in range(20):
        nscore = F(nscore + F(tabs["62"][i][i]))
    nscore = F(nscore / F(20))
    m[:, 20] = F(5) * nscore if wildcard else 0.0
    m[20, :] = F(5) * nscore if wildcard else 0.0
    m[:20, :20] = F(5) * np.asarray(tabs[str(which)], dtype=F)
    return m


class Replay:
    def __init__(self, dump, *, gappy=0.95, wildcard=False, match=18.0, mismatch=-8.0, transition=-4.0, gap_open=-50.0, gap_extend=-5.0,
                 blosum=62, no_filter=True, cal_profile_th=1000, update_seq_th=1000):
        self.type = dump["type"]
        self.P = 6 if self.type == "n" else 22
        self.gappy = float(np.float32(gappy))
        self.gap_open, self.gap_extend = F(gap_open), F(gap_extend)
        self.matrix = nucleotide_matrix(match, mismatch, transition, wildcard) if self.type == "n" else protein_matrix(blosum, wildcard)
        self.no_filter = no_filter
        self.cal_th, self.upd_th = cal_profile_th, update_seq_th
        self.seqs = dump["sequences"]
        self.rows = [s["seq"].encode() for s in self.seqs]            # current aligned row of every sequence
        self.weight = [F(s["weight"]) for s in self.seqs]
        self.low_q = [bool(s["low_quality"]) for s in self.seqs]
        self.subtree_idx = [int(s["subtree_idx"]) for s in self.seqs]
        self.name_to_id = {s["name"]: s["id"] for s in self.seqs}
        self.nodes = {k: Node(k, v["leaf"], v["grp"], v["children"]) for k, v in dump["nodes"].items()}
        self.root = self.nodes[dump["root"]]
        self.levels = [[(self.nodes[a], self.nodes[b]) for a, b in lv] for lv in dump["levels"]]
        self.subtree_aln = {}     # database->subtreeAln: negative id -> path of 0 (residue kept) / 1 (gap)
        self.fallback = []        # database->fallback_nodes
        self.task = 0
        self.cells = 0
        self.pairs_per_level = []
        self.retries = 0

    # ---- progressive.cpp:126-172 ----
    def _materialise(self, n, partner):
        if n.leaf and not n.seqs:
            sid = self.name_to_id[n.id]
            n.seqs = [sid]
            n.aln_len = len(self.rows[sid])
            n.aln_num = 1
            n.aln_weight = self.weight[sid]
        elif not n.seqs:
            for cid in n.children:
                c = self.nodes[cid]
                if (c.grp == -1 or c.grp == n.grp) and c.id != partner.id:
                    n.freq, c.freq = c.freq, None
                    n.seqs = list(c.seqs)
                    n.aln_len, n.aln_num, n.aln_weight = c.aln_len, c.aln_num, c.aln_weight
                    break

    def update_node(self, pairs):
        for a, b in pairs:
            self._materialise(a, b)
            self._materialise(b, a)

    # ---- alignment-helper.cpp:8-72 ----
    def _profile(self, n, store):
        if n.freq is not None:
            return L.profile_from_cache(n.freq, n.aln_weight, n.aln_num)
        rows = [self.rows[s][: n.aln_len] for s in n.seqs]
        w = [F(F(self.weight[s] / n.aln_weight) * F(n.aln_num)) for s in n.seqs]
        prof = L.calculate_profile(rows, w, self.P, self.type) if n.aln_len > 0 else np.zeros((0, self.P), dtype=F)
        if store:
            n.freq = L.cache_from_profile(prof, n.aln_weight, n.aln_num)
        return prof

    # ---- alignment-helper.cpp:377-503 ----
    def _update_alignment(self, a, b, path):
        total = len(path)
        for node, keep in ((a, 2), (b, 1)):
            for sid in node.seqs:
                if self.task != 2 and sid >= 0:
                    self.rows[sid] = L.apply_path(self.rows[sid], path, keep)
                else:
                    org = self.subtree_aln[sid]
                    keepm = (path == 0) | (path == keep)
                    upd = np.ones(total, dtype=np.int8)
                    upd[keepm] = org[: int(keepm.sum())]
                    self.subtree_aln[sid] = upd
        a.aln_num += b.aln_num
        a.aln_len = total
        a.aln_weight = F(a.aln_weight + b.aln_weight)
        a.seqs = a.seqs + b.seqs
        b.seqs = []
        if len(a.seqs) > self.upd_th and a.freq is not None and self.task != 2:
            count, first = 0, 0
            for idx in a.seqs:
                if idx > 1:                               # (sic: ids 0 and 1 are not counted, :483)
                    if first == 0:
                        first = -idx
                    count += 1
            if count >= self.upd_th:
                self.subtree_aln[first] = np.zeros(total, dtype=np.int8)
                new = [first]
                for idx in a.seqs:
                    if idx >= 0:
                        self.subtree_idx[idx] = first
                    else:
                        new.append(idx)
                a.seqs = new

    # ---- alignment-helper.cpp:541-591 ----
    def _fallback2cpu(self, idxs, pairs):
        filtering = not self.no_filter
        for i in sorted(idxs):
            a, b = pairs[i]
            rn, qn = a.aln_num, b.aln_num
            lq_r = False if rn > 1 else self.low_q[a.seqs[0]]
            lq_q = False if qn > 1 else self.low_q[b.seqs[0]]
            if rn < qn or lq_r:
                if (not filtering) or (not lq_r):
                    self.fallback.append(b)
                    if lq_r:
                        self.low_q[a.seqs[0]] = False
                a.aln_len, b.aln_len = b.aln_len, a.aln_len
                a.aln_num, b.aln_num = b.aln_num, a.aln_num
                a.aln_weight, b.aln_weight = b.aln_weight, a.aln_weight
                a.seqs, b.seqs = b.seqs, a.seqs
                a.freq, b.freq = b.freq, a.freq
            else:
                if (not filtering) or (not lq_q):
                    self.fallback.append(b)
                    if lq_q:
                        self.low_q[b.seqs[0]] = False

    # ---- alignment-cpu.cpp:36-183 ----
    def level(self, pairs):
        deferred = []
        for i, (a, b) in enumerate(pairs):
            ref_len, qry_len, ref_num, qry_num = a.aln_len, b.aln_len, a.aln_num, b.aln_num
            store = (ref_num >= self.cal_th or qry_num >= self.cal_th) or (a.freq is not None or b.freq is not None)
            pr = self._profile(a, store)
            pq = self._profile(b, store)
            cols_r, info_r, runs_r = L.prepare_side(pr, ref_num, self.gappy, self.gap_open, self.gap_extend, self.type)
            cols_q, info_q, runs_q = L.prepare_side(pq, qry_num, self.gappy, self.gap_open, self.gap_extend, self.type)
            cons_r, cons_q = (info_r & 0x7F).astype(np.int64), (info_q & 0x7F).astype(np.int64)
            gap_char = F(0) if (self.task in (1, 2) or ref_num > 10000 or qry_num > 10000) else self.gap_extend      # :88
            xdrop, flen = int(np.int32(1000 * -1 * float(self.gap_extend))), 1 << 12                                   # TALCO-XDrop.cpp:49-50
            path = np.zeros(0, dtype=np.int8)
            if ref_len == 0:
                path = np.ones(qry_len, dtype=np.int8)
            if qry_len == 0:
                path = np.concatenate([path, np.full(ref_len, 2, dtype=np.int8)])
            lq_r = False if ref_num > 1 else self.low_q[a.seqs[0]]
            lq_q = False if qry_num > 1 else self.low_q[b.seqs[0]]
            failed = False
            if not lq_r and not lq_q:
                R, Q = cols_r.shape[0], cols_q.shape[0]
                while len(path) == 0:
                    p = O.make_params(self.matrix, gap_open=float(self.gap_open), gap_extend=float(self.gap_extend), gap_char=float(gap_char), xdrop=xdrop, flen=flen)
                    got, err, st = O.align_pair(p, cols_r[:, : self.P], cols_q[:, : self.P], cols_r[:, self.P], cols_r[:, self.P + 1], cols_q[:, self.P],
                                                cols_q[:, self.P + 1], float(ref_num), float(qry_num))
                    self.cells += st.cells
                    if err == 0:
                        path = got
                    if self.task == 0 and err != 0:
                        path = np.zeros(0, dtype=np.int8)
                        failed = True
                        break
                    if err == 2:
                        flen = min(int(np.int32(flen * 1.2)) << 1, min(R, Q))
                        self.retries += 1
                    elif err == 3:
                        raise RuntimeError("errorType 3")
                    elif err == 1:
                        xdrop = int(np.int32(xdrop * 2))
                        flen = min(int(np.int32(xdrop * 4)) << 1, min(R, Q))
                        self.retries += 1
            if failed:
                deferred.append(i)
            if self.task == 0 and (ref_num == 1 or qry_num == 1) and (lq_r or lq_q):
                path = np.zeros(0, dtype=np.int8)
                deferred.append(i)
            if len(path):
                full = L.add_gappy_columns_back(path, runs_r, runs_q, cons_r, cons_q, self.matrix, self.gap_open, self.gap_extend)
                assert int(((full == 0) | (full == 2)).sum()) == ref_len and int(((full == 0) | (full == 1)).sum()) == qry_len
                rw, qw = a.aln_weight, b.aln_weight
                if a.freq is not None and b.freq is not None:                     # updateFrequency :506-539
                    a.freq = L.update_frequency(a.freq, b.freq, full, rw, qw)
                    b.freq = None
                    a.aln_len = len(full)
                self._update_alignment(a, b, full)
        if deferred:
            self._fallback2cpu(deferred, pairs)

    def _progressive(self, levels):
        for pairs in levels:
            self.update_node(pairs)
            self.level(pairs)
            self.pairs_per_level.append(len(pairs))

    # ---- progressive.cpp:194-230 ----
    def _expand(self, node):
        for sid in range(len(self.rows)):
            if self.subtree_idx[sid] < -1:
                aln = self.subtree_aln[self.subtree_idx[sid]]
                self.rows[sid] = L.apply_path(self.rows[sid], np.where(aln == 0, 0, 1).astype(np.int8), 2)
        new = [s for s in node.seqs if s >= 0]
        new += [sid for sid in range(len(self.rows)) if self.subtree_idx[sid] < 0]
        node.seqs = new

    # ---- progressive.cpp:232-299 ----
    def run(self):
        self._progressive(self.levels)
        last = self.levels[-1][0][0]
        root = self.root
        root.seqs = last.seqs
        if last.freq is not None:
            root.freq = last.freq
        root.aln_len, root.aln_num, root.aln_weight = last.aln_len, last.aln_num, last.aln_weight
        if last is not root:
            last.seqs, last.freq = [], None
        if not self.fallback:
            self._expand(root)
            return
        self.task = 1
        bad = libstdcxx_sort(list(self.fallback), lambda x, y: (x.aln_len > y.aln_len) if x.aln_num == y.aln_num else (x.aln_num > y.aln_num))
        self.deferred_profiles = len(bad)
        self.fallback = []
        self._progressive([[(root, n)] for n in bad])
        self._expand(root)
        self.task = 0

    def write(self, path):
        """io.cpp:512-525: every sequence that is not flagged low quality, in input order, first aln_len columns of its row."""
        n = self.root.aln_len
        with open(path, "wb") as f:
            for s, row, lq in zip(self.seqs, self.rows, self.low_q):
                if not lq:
                    f.write(b">" + s["name"].encode() + b"\n" + row[:n] + b"\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dump")
    ap.add_argument("out")
    ap.add_argument("-r", "--remove-gappy", type=float, default=0.95)
    ap.add_argument("-w", "--wildcard", action="store_true")
    ap.add_argument("--match", type=float, default=18)
    ap.add_argument("--mismatch", type=float, default=-8)
    ap.add_argument("--transition", type=float, default=-4)
    ap.add_argument("--gap-open", type=float, default=-50)
    ap.add_argument("--gap-extend", type=float, default=-5)
    ap.add_argument("-b", "--blosum", type=int, default=62)
    ap.add_argument("--filter", action="store_true")
    for ign in ("--length-deviation", "--max-ambig", "--max-len", "--min-len", "--type", "--xdrop"):      # already applied in the dump / unused (TALCO ignores --xdrop)
        ap.add_argument(ign, default=None, help="accepted for symmetry with the CLI; the dump already reflects it")
    ap.add_argument("--rooted", action="store_true", help="accepted for symmetry with the CLI")
    ap.add_argument("--cal-profile-th", "--test-cal-profile-th", dest="cal_profile_th", type=int, default=1000)      # (the CLI's development flags, include/twl_msa.h)
    ap.add_argument("--update-seq-th", "--test-update-seq-th", dest="update_seq_th", type=int, default=1000)
    a = ap.parse_args()
    r = Replay(json.load(open(a.dump)), gappy=a.remove_gappy, wildcard=a.wildcard, match=a.match, mismatch=a.mismatch, transition=a.transition,
               gap_open=a.gap_open, gap_extend=a.gap_extend, blosum=a.blosum, no_filter=not a.filter, cal_profile_th=a.cal_profile_th,
               update_seq_th=a.update_seq_th)
    r.run()
    r.write(a.out)
    print(f"REPLAY levels={len(r.pairs_per_level)} pairs_per_level={'/'.join(map(str, r.pairs_per_level))} band_cells={r.cells} aln_len={r.root.aln_len} "
          f"deferred_profiles={getattr(r, 'deferred_profiles', 0)} retries={r.retries}")


if __name__ == "__main__":
    main()
