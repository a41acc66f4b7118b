"""Run the randomized GPU-vs-oracle parity campaign over a seed range:  python tools/fuzz_gpu.py START COUNT"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import twilight_amd as twl
from test_gpu_fuzz import check_case

start, count = int(sys.argv[1]), int(sys.argv[2])
twl.init([0])
twl.set_knob(twl.knobs.KNOB_POISON_TB, 1)      # a traceback word whose store was wrongly skipped must read as garbage, not as zeros
bad = 0
kinds = {0: 0, 1: 0}
for seed in range(start, start + count):
    try:
        e, st = check_case(twl, seed)
        kinds[1 if e else 0] += 1
    except AssertionError as ex:
        bad += 1
        print("FAIL", ex, flush=True)
print(f"seeds {start}..{start+count-1}: {bad} failures; {kinds[0]} clean cases, {kinds[1]} cases with algorithmic errorTypes", flush=True)
sys.exit(1 if bad else 0)
