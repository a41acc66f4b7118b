#!/bin/bash
# What tools/final_profiles.sh left under gpurun_out/prof_r06 -> profiles/r06 (run in the build container after the gpurun call):  bash tools/copy_profiles.sh
# Then, for bench lines that carry the counters: `python bench.py ...` once more on the GPU box (the counters are attached only when the *_pmc_summary.json of the
# snapshot was taken on the running library's sources), copied over profiles/r06/bench_line_*.json.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/prof_r06; P=profiles/r06
mkdir -p $P
cp $O/pmc/kernel_stats.csv $P/bench_kernel_stats.csv
cp $O/pmc/pmc_summary.json $P/bench_pmc_summary.json
cp $O/pmc/bench_under_rocprof.json $P/bench_under_rocprof.json
for c in protein5k rnasim100k; do
  cp $O/pmc_$c/kernel_stats.csv $P/${c}_kernel_stats.csv
  cp $O/pmc_$c/pmc_summary.json $P/${c}_pmc_summary.json
  cp $O/pmc_$c/bench_under_rocprof.json $P/${c}_under_rocprof.json
done
for f in default protein5k rnasim100k rnasim1k_band512 survey8d forced_shard_1rank_rccl two_ranks_one_gpu_gloo; do cp $O/bench_line_$f.json $P/; done
python tools/level_table.py $O/verbose10k.err 31 > $P/level_table_rnasim10k.txt
python tools/level_table.py $O/verbose100k.err 39 > $P/level_table_rnasim100k.txt
python tools/level_table.py $O/verbose_survey8d.err 76 > $P/level_table_survey8d.txt
python tools/scaling_model.py $P/bench_line_default.json > $P/scaling_model_rnasim10k.txt
python tools/scaling_model.py $P/bench_line_rnasim100k.json > $P/scaling_model_rnasim100k.txt
H=$(python -c "import __graft_entry__ as g; print(g.source_hash())")
grep -c "$H" $P/bench_pmc_summary.json $P/protein5k_pmc_summary.json $P/rnasim100k_pmc_summary.json $P/isa_block_step.json
