set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r03c
bash tools/profile_bench.sh gpurun_out/prof_r03c > gpurun_out/prof_r03c.log 2>&1
cd /tmp && export TMPDIR=/tmp
for c in protein5k rnasim100k; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03c_$c -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --config $c --no-cpu --no-peak --no-e2e > $GRAFT_REPO_ROOT/gpurun_out/prof_r03c_$c.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r03c_$c.err
  find $GRAFT_REPO_ROOT/gpurun_out/prof_r03c_$c -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_r03c_${c}_kernel_stats.csv \;
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r03c_$c
done
cd $GRAFT_REPO_ROOT
python bench.py --config protein5k --no-e2e > gpurun_out/prof_r03c_line_protein5k.json 2>/dev/null
python bench.py --config rnasim100k --no-cpu --no-peak --no-e2e > gpurun_out/prof_r03c_line_rnasim100k.json 2>/dev/null
TWL_BENCH_FORCE_SHARD=1 python bench.py --no-cpu --no-peak --no-e2e > gpurun_out/prof_r03c_shard1.json 2>gpurun_out/prof_r03c_shard1.err
TWL_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu --no-peak --no-e2e > gpurun_out/prof_r03c_two_ranks_one_gpu.json 2> gpurun_out/prof_r03c_two_ranks_one_gpu.err
python bench.py > gpurun_out/prof_r03c_line_default.json 2> gpurun_out/prof_r03c_line_default.err
TWL_BENCH_VERBOSE=1 python bench.py --no-cpu --no-peak --no-e2e --steps 2 > /dev/null 2> gpurun_out/prof_r03c_verbose10k.err
TWL_BENCH_VERBOSE=1 python bench.py --config rnasim100k --no-cpu --no-peak --no-e2e --steps 2 > /dev/null 2> gpurun_out/prof_r03c_verbose100k.err
ls gpurun_out | grep prof_r03c
