for kv in MT_MARGIN=40 MT_MARGIN=64 MT_MARGIN=96 MT_LEAD2=128 MT_MARGIN=64,MT_LEAD2=128; do
  python tools/bench_with_knob.py $kv --steps 3 --warmup 1 --no-cpu --no-survey8d --no-e2e --no-peak 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
lv=d['levels']
print('$kv', round(d['ms_per_step'],1), round(d['dp_kernel']['kernel_ms_per_pass'],1), 'tiles', d['tile_parallel']['tiles_predicted'], d['tile_parallel']['tiles_inline'], 'lv12-31', round(sum(l['kernel_ms'] for l in lv[11:]),2), [round(l['kernel_ms'],2) for l in lv[11:22]])"
done
