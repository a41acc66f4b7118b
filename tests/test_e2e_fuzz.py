"""A slice of the randomized end-to-end campaign (tools/fuzz_e2e.py): device-resident CLI == host-staged CLI == CPU checker on random
families and options, including test-only low thresholds that make small trees take the cached-profile / compressed-group branches."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_random_families_all_three_paths_agree(built):
    import fuzz_e2e

    cached = 0
    for seed in range(1, 25):
        ok, same_fail, desc, res = fuzz_e2e.one(seed)
        assert ok or same_fail, f"{desc}: {res}"
        cached += "th=0" not in desc
    assert cached >= 8, "the slice is meant to include cases with lowered cache thresholds"
