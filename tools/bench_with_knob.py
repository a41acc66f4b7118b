"""Run bench.py with launch-policy knobs set first (development): python tools/bench_with_knob.py KNOB=VALUE[,KNOB=VALUE] [bench.py flags...]
Knob names as in twilight_amd/api.py without the KNOB_ prefix, e.g. MT_TAIL_PCT=0."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twilight_amd import api

api.init([int(os.environ.get("LOCAL_RANK", "0"))])
for kv in sys.argv[1].split(","):
    k, v = kv.split("=")
    api.set_knob(getattr(api, "KNOB_" + k), int(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
