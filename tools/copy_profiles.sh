#!/bin/bash
# What tools/final_profiles.sh left under gpurun_out/prof_r05 -> profiles/r05 (run in the build container after the gpurun call):  bash tools/copy_profiles.sh
# Then, for a bench line that carries the counters: one more `python bench.py > gpurun_out/prof_r05/bench_line_default.json` on the GPU box (the counters are attached
# only when profiles/r05/bench_pmc_summary.json in the snapshot was taken on the running library's sources) and `cp` it over profiles/r05/bench_line_default.json.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/prof_r05; P=profiles/r05
cp $O/pmc/kernel_stats.csv $P/bench_kernel_stats.csv
cp $O/pmc/pmc_summary.json $P/bench_pmc_summary.json
cp $O/pmc/bench_under_rocprof.json $P/bench_under_rocprof.json
for f in default protein5k rnasim100k rnasim1k_band512 survey8d forced_shard_1rank_rccl two_ranks_one_gpu_gloo; do cp $O/bench_line_$f.json $P/; done
cp $O/protein5k_kernel_stats.csv $O/rnasim100k_kernel_stats.csv $P/
python tools/level_table.py $O/verbose10k.err 31 > $P/level_table_rnasim10k.txt
python tools/level_table.py $O/verbose100k.err 39 > $P/level_table_rnasim100k.txt
python tools/level_table.py $O/verbose_survey8d.err 76 > $P/level_table_survey8d.txt
python tools/scaling_model.py $P/bench_line_default.json > $P/scaling_model_rnasim10k.txt
python tools/scaling_model.py $P/bench_line_rnasim100k.json > $P/scaling_model_rnasim100k.txt
# the static instruction counts belong to the step loops, which tools/isa_block_step.py re-derives from the sources: the listings must not have changed
H=$(python -c "import __graft_entry__ as g; print(g.source_hash())")
mkdir -p /tmp/isa_chk
python tools/isa_block_step.py "<6, 4, 2, 2, 5, false, false, 0, 0>" /tmp/isa_chk > /dev/null
python tools/isa_block_step.py "<6, 4, 2, 5, 5, false, false, 0, 1>" /tmp/isa_chk > /dev/null
if diff <(tail -n +2 /tmp/isa_chk/isa_step_6_4_2_2_5_false_false_0_0.s) <(tail -n +2 $P/isa_step_6_4_2_2_5_false_false_0.s) > /dev/null && cmp -s /tmp/isa_chk/isa_step_6_4_2_5_5_false_false_0_1.s $P/isa_step_6_4_2_5_5_false_false_0_1.s; then
  sed -i "s/\"source_hash\": \"[0-9a-f]*\"/\"source_hash\": \"$H\"/" $P/isa_block_step.json
  echo "step listings unchanged: isa_block_step.json stamped $H"
else
  echo "STEP LISTINGS CHANGED: re-derive profiles/r05/isa_block_step.json with tools/isa_block_step.py" >&2; exit 1
fi
grep -c "$H" $P/bench_pmc_summary.json $P/isa_block_step.json
