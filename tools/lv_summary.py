import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], '%.4e'%d['value'], round(d['ms_per_step'],1), d['config'].get('msa_md5')[:8], round(d['dp_kernel']['kernel_ms_per_pass'],1), d['config'].get('pairs_rerun_in_wider_window'), [round(l['kernel_ms'],1) for l in d['levels'][:9]])
