"""Generates tests/golden/small_pairs.npz: small level batches with full fp32 inputs and the expected paths / errorTypes / band-cell
counts.  The expectations come from oracle/talco_oracle.c, which is pinned end to end against the reference's
recorded outputs (tests/test_e2e_pin.py); the fixture freezes them so that neither the oracle nor the kernels can drift unnoticed.

    python tests/golden/make_small_pairs.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from twilight_amd import synth  # noqa: E402

CASES = [
    # name, P, members, length, seed, params
    ("nuc_leaf", 6, (1, 1), 280, 1, {}),
    ("nuc_profiles", 6, ((2, 7), (1, 5)), 300, 2, {}),
    ("nuc_multi_tile_marker64", 6, ((1, 4), (1, 4)), 300, 3, {"marker": 64}),
    ("nuc_xdrop_empties_band", 6, ((1, 3), (1, 3)), 260, 4, {"xdrop": 50}),          # errorType 1
    ("nuc_band_wider_than_flen", 6, ((1, 3), (1, 3)), 260, 5, {"flen": 110, "xdrop": 1200}),   # errorType 2, late in the pair
    ("nuc_zero_gap_char", 6, ((3, 8), (3, 8)), 240, 6, {"gap_char": 0.0}),
    ("prot_leaf", 22, (1, 1), 200, 7, {}),
    ("prot_profiles", 22, ((2, 6), (2, 6)), 220, 8, {"marker": 96}),
]


def main():
    out = {}
    for name, P, members, length, seed, pk in CASES:
        b = synth.make_level_batch(4, length, members=members, seed=seed, P=P, sub=(0.2 if P == 22 else 0.08))
        M = synth.protein_matrix() if P == 22 else synth.nucleotide_matrix()
        aln, n, err, st = O.align_batch(O.make_params(M, **pk), b, threads=1)
        cells = np.zeros(b.n_pairs, dtype=np.uint64)
        for i in range(b.n_pairs):     # per-pair band cells
            _, _, s1 = O.align_pair(O.make_params(M, **pk), b.freq[i, 0, : b.len[i, 0]], b.freq[i, 1, : b.len[i, 1]], b.gap_open[i, 0], b.gap_extend[i, 0],
                                    b.gap_open[i, 1], b.gap_extend[i, 1], b.num[i, 0], b.num[i, 1])
            cells[i] = s1.cells
        out[f"{name}/freq"] = b.freq
        out[f"{name}/gap_open"] = b.gap_open
        out[f"{name}/gap_extend"] = b.gap_extend
        out[f"{name}/len"] = b.len
        out[f"{name}/num"] = b.num
        out[f"{name}/aln"] = aln[:, : int(n.max()) if n.max() > 0 else 1]
        out[f"{name}/aln_len"] = n
        out[f"{name}/err"] = err
        out[f"{name}/cells"] = cells
        out[f"{name}/params"] = np.array([pk.get("marker", 1024), pk.get("xdrop", -1), pk.get("flen", 4096), 1 if "gap_char" in pk else 0], dtype=np.int64)
        print(name, "err", err.tolist(), "len", n.tolist(), "cells", cells.tolist())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "small_pairs.npz"), **out)


if __name__ == "__main__":
    main()
