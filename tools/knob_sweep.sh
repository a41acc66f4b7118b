# Development: the bench under a few values of launch-policy knobs (tools/bench_with_knob.py): bash tools/knob_sweep.sh "K=V K=V,K2=V2 ..."
for kv in ${1:-MT_MARGIN=40 MT_MARGIN=64 MT_LEAD2=128}; do
  python tools/bench_with_knob.py $kv --steps 3 --warmup 1 --no-cpu --no-survey8d --no-e2e --no-peak 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
lv=d['levels']
print('$kv', round(d['ms_per_step'],1), round(d['dp_kernel']['kernel_ms_per_pass'],1), 'tiles', d['tile_parallel']['tiles_predicted'], d['tile_parallel']['tiles_inline'], 'lv1-11', [round(l['kernel_ms'],1) for l in lv[:11]], 'lv12-31', round(sum(l['kernel_ms'] for l in lv[11:]),2))"
done
