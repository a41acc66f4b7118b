"""Classify the errorType-3 outcomes of the randomized campaign (CPU only: the checker, pair by pair).

    python tools/classify_err3.py START COUNT [out.json]

errorType 3 is the reference's "There might be some bugs in the code!" exit (TALCO-XDrop.cpp:108-112 and the consistency tests of
Tile); the checker records which test fired (oracle/talco_oracle.h: err3_reason 1 entry lengths, 2 marker state is a sentinel /
unset convergence value (:645-652), 3 start address, 4 negative reference step, 5 lengths after the advance, 6 path overflow) and how
often a cell took the match pointer without a valid diagonal predecessor (oob_diag, the unguarded read at :541).  The point of the
tally: reason 2 only ever appears on pairs where that unguarded read happened, i.e. where the reference itself reads outside its band.
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from test_gpu_fuzz import random_case


def classify(start, count):
    tally = {"cases": 0, "pairs": 0, "errorType": {0: 0, 1: 0, 2: 0, 3: 0}, "err3_reason": {}, "err3_with_oob_diag_0": {}, "pairs_with_oob_diag": 0,
             "reason2_without_oob": []}
    for seed in range(start, start + count):
        batch, matrix, pk = random_case(seed)
        p = O.make_params(matrix, **pk)
        tally["cases"] += 1
        for i in range(batch.n_pairs):
            R, Q = int(batch.len[i, 0]), int(batch.len[i, 1])
            if R <= 0 or Q <= 0: continue
            _, err, st = O.align_pair(p, batch.freq[i, 0, :R], batch.freq[i, 1, :Q], batch.gap_open[i, 0, :R], batch.gap_extend[i, 0, :R],
                                      batch.gap_open[i, 1, :Q], batch.gap_extend[i, 1, :Q], float(batch.num[i, 0]), float(batch.num[i, 1]))
            tally["pairs"] += 1
            tally["errorType"][err] += 1
            if st.oob_diag: tally["pairs_with_oob_diag"] += 1
            if err == 3:
                r = int(st.err3_reason)
                tally["err3_reason"][r] = tally["err3_reason"].get(r, 0) + 1
                if st.oob_diag == 0:
                    tally["err3_with_oob_diag_0"][r] = tally["err3_with_oob_diag_0"].get(r, 0) + 1
                    if r == 2: tally["reason2_without_oob"].append([seed, i])
    return tally


if __name__ == "__main__":
    start, count = int(sys.argv[1]), int(sys.argv[2])
    t = classify(start, count)
    t["seeds"] = [start, start + count - 1]
    s = json.dumps(t, indent=1)
    print(s)
    if len(sys.argv) > 3: open(sys.argv[3], "w").write(s + "\n")
