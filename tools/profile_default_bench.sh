#!/bin/bash
# rocprofv3 kernel trace of the DEFAULT bench command (python3 bench.py), so that the committed average kernel duration is the one the
# bench line itself reports: tools/profile_default_bench.sh <outdir under the repo>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/$1
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-e2e > $O/bench_default_under_rocprof.json 2> $O/kt.err
find $O -name "*kernel_stats.csv" -exec cp {} $O/default_kernel_stats.csv \;
head -3 $O/default_kernel_stats.csv | cut -c1-200
tail -c 900 $O/bench_default_under_rocprof.json
