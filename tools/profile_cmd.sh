#!/bin/bash
# rocprofv3 counters of any command's DP kernel (run on the GPU box): tools/profile_cmd.sh <outdir under the repo> <python args...>
# PMC counters in separate passes with --kernel-trace only (never with other trace domains).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/$1; shift
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 "$@" > $O/cmd.out 2> $O/kt.err
rocprofv3 --kernel-trace --output-format csv -d $O/p1 -o p1 --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES -- python3 "$@" > /dev/null 2> $O/p1.err
rocprofv3 --kernel-trace --output-format csv -d $O/p2 -o p2 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -- python3 "$@" > /dev/null 2> $O/p2.err
python3 $R/tools/summarize_pmc.py $O > $O/pmc_summary.json
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cat $O/pmc_summary.json
