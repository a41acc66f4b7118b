"""Summarise the rocprofv3 PMC passes written by tools/profile_bench.sh / tools/profile_cmd.sh into one JSON (stdout): counter sums of the
DP kernels' dispatches, per kernel name, and the HBM bytes per bench pass (FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE is doubled on
gfx950 per MI355X_MICROARCH.md).  Usage: summarize_pmc.py <dir> [passes of the family per command run]"""
import csv, glob, json, os, sys

root = sys.argv[1]
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out = {}
for i in (1, 2, 3, 4):
    files = glob.glob(os.path.join(root, f"p{i}", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    sums, disp, per = {}, set(), {}
    for row in csv.DictReader(open(files[0])):
        if "talco_" not in row["Kernel_Name"]:
            continue
        disp.add(row["Dispatch_Id"])
        sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        k = per.setdefault(row["Kernel_Name"], {"dispatches": set(), "wg": row["Workgroup_Size"], "lds": row["LDS_Block_Size"], "vgpr": row["VGPR_Count"], "sum": {}})
        k["dispatches"].add(row["Dispatch_Id"])
        k["sum"][row["Counter_Name"]] = k["sum"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    for k in per.values():
        k["dispatches"] = len(k["dispatches"])
    out[f"pass{i}"] = {"dispatches": len(disp), "sum_over_dispatches": sums, "per_kernel": per}
bench = None
try:
    for line in open(os.path.join(root, "bench_under_rocprof.json")):
        if line.startswith("{"):
            bench = json.loads(line)
except OSError:
    pass
if "pass3" in out and "pass4" in out and bench:
    fetch = out["pass3"]["sum_over_dispatches"]["FETCH_SIZE"] * 1024 / passes
    write = out["pass4"]["sum_over_dispatches"]["WRITE_SIZE"] * 1024 / passes
    cells = bench["roofline"]["cells"]
    out["hbm_per_pass"] = {"fetch_bytes_reported": fetch, "fetch_bytes_corrected_x2": 2 * fetch, "write_bytes": write, "traffic_bytes": 2 * fetch + write,
                           "band_cells_per_pass": cells, "traffic_bytes_per_cell": (2 * fetch + write) / cells, "family_passes_per_run": passes,
                           "note": "DP kernels only, all launches of one pass over the family; rocprofv3 --pmc in separate passes with --kernel-trace only; FETCH_SIZE/WRITE_SIZE are KB; "
                                   "FETCH doubled per MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at half)"}
# ---- what binds: issue and wait fractions per kernel, from the counters above and the kernel-trace durations of the same command ----
# MI355X: 256 CUs x 4 SIMDs at 2.4 GHz; a wave's VALU instruction occupies its SIMD for 2 cycles (64 lanes on a SIMD-32), the scalar
# unit of a CU retires at most ~1 instruction per cycle (MI355X_MICROARCH.md; tools/micro/issue_rates*.hip).
CLK, CUS, SIMDS = 2.4e9, 256, 1024
dur = {}
ks = glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True) + glob.glob(os.path.join(root, "kernel_stats.csv"))
if ks:
    for row in csv.DictReader(open(ks[0])):
        dur[row["Name"]] = float(row["TotalDurationNs"]) * 1e-9
issue = {}
if "pass1" in out and dur:
    p2 = out.get("pass2", {}).get("per_kernel", {})
    p3 = out.get("pass3", {}).get("per_kernel", {})
    p4 = out.get("pass4", {}).get("per_kernel", {})
    for name, k in out["pass1"]["per_kernel"].items():
        t = dur.get(name)
        if not t:
            continue
        c = k["sum"]
        e = {"seconds_in_run": t, "dispatches": k["dispatches"],
             "valu_issue_frac": c.get("SQ_INSTS_VALU", 0.0) * 2.0 / (SIMDS * CLK * t),
             "salu_per_cycle_per_cu": c.get("SQ_INSTS_SALU", 0.0) / (CUS * CLK * t),
             "lds_insts_per_cycle_per_cu": c.get("SQ_INSTS_LDS", 0.0) / (CUS * CLK * t)}
        # every instruction the waves issued (vector, scalar, LDS, vector / scalar memory, branches) per ns and CU, against the measured ceiling of 3.97
        # (tools/micro/issue_rates.hip: 16 waves per CU, interleaved vector and scalar instructions) -- no clock enters
        br = p2.get(name, {}).get("sum", {}).get("SQ_INSTS_BRANCH", 0.0)
        tot = sum(c.get(f, 0.0) for f in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM")) + br
        e["insts_per_ns_per_cu"] = tot / (CUS * t * 1e9)
        e["issue_frac_of_measured_ceiling"] = e["insts_per_ns_per_cu"] / 3.97
        e["insts_by_kind"] = {f[len("SQ_INSTS_"):].lower(): c.get(f, 0.0) for f in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM")}
        e["insts_by_kind"]["branch"] = br
        if name in p2:
            c2 = p2[name]["sum"]
            if c.get("SQ_WAVE_CYCLES"):
                e["wait_frac_of_wave_cycles"] = c2.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
                e["active_inst_frac_of_wave_cycles"] = c2.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        if name in p3 and name in p4:
            e["hbm_gb_per_s"] = (2 * p3[name]["sum"].get("FETCH_SIZE", 0.0) + p4[name]["sum"].get("WRITE_SIZE", 0.0)) * 1024 / t / 1e9
        issue[name] = e
    out["issue_per_kernel"] = issue
# ---- the dominant kernel in ONE place: launches and average duration from the kernel trace of this command, band cells per launch from the
# bench line of the same command, instructions of a block step from the disassembly -> the issue-ceiling fraction, recomputable from these few numbers ----
try:
    isa = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06", "isa_block_step.json")))
    dk = (bench or {}).get("roofline", {}).get("dominant_kernel")
    if dk and ks:
        # (the profiler spells a defaulted last template argument out: "..., false, 0>" of the library is "..., false, 0, 0>" in the trace)
        names = (dk["kernel"] + "(", dk["kernel"][:-1] + ", 0>(")
        rows = [r for r in csv.DictReader(open(ks[0])) if any(nm in r["Name"] for nm in names)]
        import re as _re
        _m = _re.search(r"talco_lean_kernel<(\d+, \d+, \d+, \d+, \d+, false, false, )([^>]*)>", dk["kernel"])
        key = None
        if _m:
            _want = "talco_lean_kernel<" + _m.group(1) + ("1" if "/" in _m.group(2) else _m.group(2)) + ">"
            key = _want if _want in isa["kernels"] else None
        if rows and key:
            allrows = [r for r in csv.DictReader(open(ks[0])) if "talco_" in r["Name"]]
            dp_s_per_pass = sum(float(r["TotalDurationNs"]) for r in allrows) * 1e-9 / passes
            cells_per_pass = bench["roofline"]["cells"]
            avg_ns = sum(float(r["TotalDurationNs"]) for r in rows) / max(1, sum(int(r["Calls"]) for r in rows))
            s0 = isa["kernels"][key]["phase_A"]["slot0"]["least" if ", 5, 5, false" in key else "all_paths"]
            cap = isa["valu_ceiling"]["cus"] * isa["valu_ceiling"]["units_per_ns_per_cu"] * 1e9
            peak = cap * 64.0 / s0["valu_units"]
            peak_issue = isa["issue_ceiling"]["cus"] * isa["issue_ceiling"]["instr_per_ns_per_cu"] * 1e9 * 64.0 / s0["instructions"]
            out["dominant_kernel"] = {"kernel": dk["kernel"], "calls_in_trace": sum(int(r["Calls"]) for r in rows), "avg_ns_in_trace": avg_ns,
                                      "valu_units_per_block_step": s0["valu_units"], "instructions_per_block_step": s0["instructions"], "cells_per_block_step": 64,
                                      "valu_ceiling_units_per_ns_per_cu": isa["valu_ceiling"]["units_per_ns_per_cu"], "cus": isa["valu_ceiling"]["cus"],
                                      "peak_cells_per_s": peak, "peak_cells_per_s_total_issue": peak_issue,
                                      # what bench.py calls a launch of this kernel is a LEVEL that starts on it (its remainder runs as tile jobs of the same family):
                                      "bench_level_cells_per_launch": dk["cells_per_launch"], "bench_level_dp_ms_per_launch": dk["avg_ms"],
                                      "frac_levels_of_this_kernel": dk["cells_per_launch"] / (dk["avg_ms"] * 1e-3) / peak,
                                      # the whole pass, from the trace alone: every talco_* launch of the command / passes of the family in it
                                      "trace_dp_kernel_s_per_pass": dp_s_per_pass, "bench_dp_kernel_s_per_pass": bench["roofline"]["kernel_ms"] * 1e-3,
                                      "band_cells_per_pass": cells_per_pass,
                                      "frac_whole_pass": cells_per_pass / dp_s_per_pass / peak,
                                      "frac_whole_pass_total_issue": cells_per_pass / dp_s_per_pass / peak_issue,
                                      "note": "frac_whole_pass = band_cells_per_pass / (sum of TotalDurationNs of the talco_* rows of the kernel_stats csv / passes in that command) / "
                                              "(cus x 3.7e9 x 64 / valu_units_per_block_step); ..._total_issue = round 5's definition (3.97e9 instructions of any kind, instructions of the block step)"}
            # the vector unit's load as the counters see it: vector instructions per ns and CU of the dominant kernel (the slow classes top out at 2.15-2.27, issue_rates5.log)
            for nm, e in issue.items():
                if any(x in nm for x in names) and "insts_by_kind" in e:
                    out["dominant_kernel"]["valu_insts_per_ns_per_cu"] = e["insts_by_kind"]["valu"] / (CUS * e["seconds_in_run"] * 1e9)
except Exception as ex:  # noqa: BLE001
    out["dominant_kernel"] = {"note": f"not formed: {ex}"}
try:      # which code these counters belong to: the hash of the kernel sources the profiled library was built from (twl_version carries the same)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as _g
    out["source_hash"] = _g.source_hash()
except Exception:
    out["source_hash"] = None
print(json.dumps(out, indent=1))
