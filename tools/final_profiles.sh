# Round-5 evidence under gpurun_out/prof_r05 (copied to profiles/r05 by hand): bash tools/final_profiles.sh   (GPU box)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r05
mkdir -p $O
bash tools/profile_bench.sh $O/pmc --no-survey8d > $O/pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp
for c in protein5k rnasim100k; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_$c -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --config $c --no-cpu --no-peak --no-e2e > $GRAFT_REPO_ROOT/$O/${c}_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/${c}.err
  find $GRAFT_REPO_ROOT/$O/kt_$c -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/$O/${c}_kernel_stats.csv \;
  rm -rf $GRAFT_REPO_ROOT/$O/kt_$c
done
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_line_default.json 2> $O/bench_line_default.err
python bench.py --config protein5k --no-e2e > $O/bench_line_protein5k.json 2>/dev/null
python bench.py --config rnasim100k --no-cpu --no-peak --no-e2e > $O/bench_line_rnasim100k.json 2>/dev/null
python bench.py --config rnasim1k_band512 --no-e2e > $O/bench_line_rnasim1k_band512.json 2>/dev/null
python bench.py --workload survey8d --steps 2 --warmup 1 --no-cpu --no-peak --no-e2e > $O/bench_line_survey8d.json 2>/dev/null
TWL_BENCH_FORCE_SHARD=1 python bench.py --no-cpu --no-peak --no-e2e --no-survey8d > $O/bench_line_forced_shard_1rank_rccl.json 2> $O/shard1.err
TWL_BENCH_ONE_GPU=1 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu --no-peak --no-e2e > $O/bench_line_two_ranks_one_gpu_gloo.json 2> $O/two_ranks.err      # (a plain command: bench.py starts its own launcher)
TWL_BENCH_VERBOSE=1 python bench.py --no-cpu --no-peak --no-e2e --no-survey8d --steps 1 > /dev/null 2> $O/verbose10k.err
TWL_BENCH_VERBOSE=1 python bench.py --config rnasim100k --no-cpu --no-peak --no-e2e --steps 1 > /dev/null 2> $O/verbose100k.err
TWL_BENCH_VERBOSE=1 python bench.py --workload survey8d --no-cpu --no-peak --no-e2e --steps 1 --warmup 0 > /dev/null 2> $O/verbose_survey8d.err
ls $O
