"""errorType 3 of the randomized campaign, classified with the checker (CPU): the sentinel / unset convergence value at the tile exit
(TALCO-XDrop.cpp:645-652, where the reference indexes out of range) only occurs on pairs where the reference's unguarded read of a
diagonal predecessor outside the band (:541) happened first."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from classify_err3 import classify


def test_sentinel_convergence_value_needs_an_out_of_band_read():
    t = classify(1000, 120)
    assert t["pairs"] > 300
    assert t["reason2_without_oob"] == [], t
    # every other consistency exit is unreachable on these inputs as well
    assert set(t["err3_with_oob_diag_0"]) <= {2} and not t["err3_with_oob_diag_0"], t
