#!/bin/bash
# Development: bench passes of the 10k x 10 kbp family under different tile-parallel knobs (needs a TWL_DEV build of the library).
#   tools/mt_sweep.sh <outdir under the repo> "<ENV=V ...>" ["<ENV=V ...>" ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/$1; shift
mkdir -p $O
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-peak --no-e2e > $O/sweep_$i.json 2> $O/sweep_$i.err
  python3 - "$O/sweep_$i.json" "$cfg" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
lv = d["levels"]
print(sys.argv[2], "| ms/pass %.1f dp %.1f | md5 %s | hit %.3f | levels 8-14: %s | 15-31 sum %.1f" % (
    d["ms_per_step"], d["dp_kernel"]["kernel_ms_per_pass"], d["config"]["msa_md5"][:8], d["tile_parallel"]["hit_rate"],
    " ".join("%.1f" % l["kernel_ms"] for l in lv[7:14]), sum(l["kernel_ms"] for l in lv[14:])))
PY
done
