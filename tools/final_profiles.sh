# Round-6 evidence under gpurun_out/prof_r06 (copied to profiles/r06 by tools/copy_profiles.sh): bash tools/final_profiles.sh   (GPU box)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r06
mkdir -p $O
# the default configuration: kernel trace + the four counter passes of the bench command
bash tools/profile_bench.sh $O/pmc --no-survey8d > $O/pmc.log 2>&1
# the other BASELINE configurations: kernel trace + the same counter passes
bash tools/profile_bench.sh $O/pmc_protein5k --config protein5k > $O/pmc_protein5k.log 2>&1
bash tools/profile_bench.sh $O/pmc_rnasim100k --config rnasim100k > $O/pmc_rnasim100k.log 2>&1
for c in pmc pmc_protein5k pmc_rnasim100k; do rm -rf $O/$c/kt $O/$c/p1 $O/$c/p2 $O/$c/p3 $O/$c/p4; done
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
