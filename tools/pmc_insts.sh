#!/bin/bash
# Instruction counters of the bench command on the GPU box (kernel trace + the two SQ passes of tools/profile_bench.sh): tools/pmc_insts.sh <outdir under the repo> [bench.py args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/$1; shift
mkdir -p $O
ARGS="--steps 2 --warmup 1 --no-cpu --no-e2e --no-peak --no-survey8d $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py $ARGS > $O/bench_under_rocprof.json 2> $O/kt.err
rocprofv3 --kernel-trace --output-format csv -d $O/p1 -o p1 --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES -- python3 $R/bench.py $ARGS > /dev/null 2> $O/p1.err
rocprofv3 --kernel-trace --output-format csv -d $O/p2 -o p2 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -- python3 $R/bench.py $ARGS > /dev/null 2> $O/p2.err
python3 $R/tools/summarize_pmc.py $O 3 > $O/pmc_summary.json
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt $O/p1 $O/p2
ls $O
