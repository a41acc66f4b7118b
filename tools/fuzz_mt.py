"""Randomized campaign on the tile-parallel path (long pairs, random markers, spoiled predictions, rounds, leads, both geometries) against
the oracle:  python tools/fuzz_mt.py START COUNT      (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import twilight_amd as twl
from test_gpu_mt import check_mt_case

start, count = int(sys.argv[1]), int(sys.argv[2])
twl.init([0])
twl.set_knob(twl.knobs.KNOB_POISON_TB, 1)      # a traceback word whose store was wrongly skipped must read as garbage, not as zeros
bad = took = errs = 0
for seed in range(start, start + count):
    try:
        mt, e = check_mt_case(twl, seed)
        took += mt; errs += 1 if e else 0
    except AssertionError as ex:
        bad += 1
        print("FAIL", ex, flush=True)
print(f"seeds {start}..{start+count-1}: {bad} failures; {took} cases through the tile-parallel path, {errs} cases with algorithmic errorTypes", flush=True)
sys.exit(1 if bad else 0)
