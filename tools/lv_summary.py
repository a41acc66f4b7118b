"""One line of a bench line (development): value, ms per step, MSA md5, DP-kernel ms per pass, pairs re-run in a wider window, kernel ms of the first nine levels.
   python tools/lv_summary.py <bench_line.json>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
md5 = (d["config"].get("msa_md5") or "-")[:8]
dp = d.get("dp_kernel") or {}
kms = dp.get("kernel_ms_per_pass")
print(sys.argv[1], "%.4e" % d["value"], round(d["ms_per_step"], 1), md5, round(kms, 1) if kms is not None else "-", d["config"].get("pairs_rerun_in_wider_window"),
      [round(l["kernel_ms"], 1) for l in (d.get("levels") or [])[:9]])
