#!/bin/bash
# rocprofv3 kernel trace of the product CLI on a generated family (run on the GPU box): tools/prof_e2e.sh <leaves> <length> <outdir> [n|p]
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
D=/tmp/twl_prof_fam
T=${4:-n}
python $R/tools/e2e_bench.py --leaves $1 --length $2 --type $T --keep $D --generate-only
mkdir -p $R/$3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$3 -o e2e -- $R/twilight_amd/twilight-mi355x -t $D/t.nwk -i $D/s.fa -o $D/out.aln --type $T -v > $R/$3/cli.log 2>&1
ls $R/$3
