"""Static instruction count of one anti-diagonal step of a talco_lean_kernel instantiation, from its gfx950 disassembly (cross-compiles without a GPU).

    python tools/isa_block_step.py "<6, 4, 2, 2, 5, false, false, 0>" [out_dir]

Compiles the kernel alone (seconds), cuts the phase-A loop (k < marker - 1: the loop most diagonals of a tile run in) out of the listing, splits it into
basic blocks and counts instructions per block and kind.  Writes <out_dir>/isa_step_<tag>.s (the loop, every block headed by its counts) and prints a JSON
summary.  Which blocks a diagonal actually executes depends on the data (gap letters present, denominators, activity of a slot); the summary therefore gives
three sums: `always` (blocks on every path through an ACTIVE slot with gap letters and a general denominator -- found by walking fall-through and
unconditional branches from the slot's first block and taking the NOT-taken side of every forward conditional branch whose target lies inside the slot),
`slot_text` (all instructions in the slot's text range, rare paths included) and `per_diagonal` (the text outside the slots: step head, hook, barrier, band update)."""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "twilight_amd", "csrc")


def compile_one(targs):
    d = tempfile.mkdtemp(prefix="twl_isa_")
    src = os.path.join(d, "k.hip")
    open(src, "w").write('#include <type_traits>\n#include "talco_nuc.hip.h"\ntemplate __global__ void twl::talco_lean_kernel%s(twl::NArgs);\n' % targs)
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-I" + CSRC, "-S",
                           "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kind(ins):
    op = ins.split()[0]
    if op == "s_nop": return "nop"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op in ("s_barrier", "s_waitcnt", "s_setprio"): return "sync"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")): return "vmem"
    if op.startswith("v_"): return "valu"
    return "other"


FAST = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|mad|mov)_(f32|b32)(_e32|_e64)?$")


def valu_units(ins):
    """Issue cost of a vector instruction in units of one full-rate fp32 instruction, as tools/micro/issue_rates5.hip measured it on MI355X at the occupancy of the
    throughput kernels (profiles/r06/issue_rates5.log): fp32 add / mul / fma / mov on vector registers and inline constants 3.6-3.8 per ns and CU (1 unit); anything with
    an SGPR operand, every integer, compare, select and DPP form 2.15-2.27 (1.7 units); packed fp32 1.85 (2 units, two operations)."""
    op, _, rest = ins.partition(" ")
    if op.startswith("v_pk_"): return 2.0
    if "dpp" in ins or "sdwa" in ins: return 1.7
    if FAST.match(op):
        ops = [o.strip() for o in rest.split(",")]
        if any(re.match(r"^-?\|?(s\d+|s\[|vcc|exec|m0|ttmp)", o) for o in ops[1:]): return 1.7      # a scalar register among the sources
        if any(re.match(r"^(0x[0-9a-f]+|-?\d{3,})$", o) for o in ops[1:]): return 1.7                   # a 32-bit literal (not an inline constant)
        return 1.0
    return 1.7


def main():
    targs = sys.argv[1]
    out_dir = sys.argv[2] if len(sys.argv) > 2 else "."
    lines = compile_one(targs)
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3twl17talco_lean_kernel"))
    end = next(i for i, l in enumerate(lines) if l.startswith(".Lfunc_end") and i > start)
    k = lines[start:end]
    # the step leaves a comment in the listing ("; TWL_STEP PH=p GEN=g"): the loop around the comment of the wanted variant is cut out.
    # WHICH (argv[3], default "0,0") = phase,general: "0,0" the plain phase-A step most diagonals of a tile run in, "2,0" the plain phase-C step
    ph, gen = (sys.argv[3] if len(sys.argv) > 3 else "0,0").split(",")
    mark = next(i for i, l in enumerate(k) if f"TWL_STEP PH={ph} GEN={gen}" in l)
    best = None
    for i, l in enumerate(k):
        if "Loop Header" not in l or i > mark: continue
        a = i
        while not k[a].startswith(".LBB"): a -= 1
        head_label = k[a].split(":")[0]
        backs = [j for j in range(mark, len(k)) if re.match(r"\s+s_c?branch\w*\s+%s\b" % re.escape(head_label), k[j])]
        if backs and (best is None or a > best[0]): best = (a, backs[-1])      # the innermost loop that holds the comment: the latest header with a branch back behind it
    a, z = best
    body = k[a:z + 1]
    blocks, cur = [], None
    for l in body:
        if l.startswith(".LBB"):
            cur = {"label": l.split(":")[0], "ins": []}
            blocks.append(cur)
        elif l.startswith("\t") and not l.strip().startswith((";", ".")) and cur is not None:
            ins = l.strip()
            if ins.startswith(";;#"): continue
            cur["ins"].append(ins)
            if ins.split()[0].startswith(("s_cbranch", "s_branch")):
                cur = {"label": cur["label"] + "+", "ins": []}      # a conditional branch ends a basic block: the fall-through continues under a derived label
                blocks.append(cur)
    blocks = [b for b in blocks if b["ins"]]
    tot = {}
    for b in blocks:
        c = {}
        for ins in b["ins"]:
            c[kind(ins)] = c.get(kind(ins), 0) + 1
        b["count"] = c
        b["units"] = round(sum(valu_units(i) for i in b["ins"] if kind(i) == "valu"), 1)
        for kk, v in c.items():
            tot[kk] = tot.get(kk, 0) + v
    tag = re.sub(r"[^0-9a-z]+", "_", targs.lower()).strip("_")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"isa_step_{tag}.s"), "w") as f:
        f.write(f"; talco_lean_kernel{targs}: the phase-A loop (one anti-diagonal per iteration), basic blocks with instruction counts -- tools/isa_block_step.py\n")
        for b in blocks:
            f.write(f"; ---- {b['label']}: {sum(b['count'].values())} instructions {json.dumps(b['count'])}, {b['units']} VALU units\n")
            for ins in b["ins"]:
                f.write("\t" + ins + "\n")
    print(json.dumps({"kernel": "talco_lean_kernel" + targs, "loop_text_instructions": sum(tot.values()), "by_kind": tot, "valu_units_text": round(sum(b["units"] for b in blocks), 1), "basic_blocks": len(blocks),
                      "listing": f"isa_step_{tag}.s"}))


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "--all"):
    main()


# ---- round 6: the summary of every kernel the bench configurations are dominated by, in one file (profiles/r06/isa_block_step.json) ----
KERNELS = {
    "talco_lean_kernel<6, 4, 2, 2, 5, false, false, 0>": "throughput, 512-row window, profiles (levels 2+ of the nucleotide configurations)",
    "talco_lean_kernel<6, 4, 2, 5, 5, false, false, 0>": "throughput, 512-row window, one-letter query rows",
    "talco_lean_kernel<6, 4, 2, 5, 5, false, false, 0, 1>": "throughput, 512-row window, the leaf x leaf step (SP 1)",
    "talco_lean_kernel<6, 4, 3, 2, 4, false, false, 0>": "throughput, 768-row window",
    "talco_lean_kernel<6, 4, 3, 2, 4, false, false, 1>": "tile jobs on the 768-row throughput geometry",
    "talco_lean_kernel<6, 4, 2, 2, 5, false, false, 1>": "tile jobs on the 512-row throughput geometry",
    "talco_lean_kernel<6, 16, 1, 2, 1, false, false, 1>": "tile jobs on the latency geometry",
    "talco_lean_kernel<6, 16, 3, 2, 1, false, false, 1>": "tile jobs of the wide re-runs (3072-row window): the family of SURVEY 8d as written",
    "talco_lean_kernel<22, 8, 1, 3, 4, false, false, 0>": "protein throughput: the loop over the non-zero letters of the reference column, counted for ONE letter per column (the least a column can hold)",
    "talco_lean_kernel<22, 16, 1, 4, 1, false, false, 1>": "protein tile jobs on precomputed scores",
    "talco_lean_kernel<22, 8, 1, 4, 4, false, false, 1>": "protein tile jobs on precomputed scores, throughput geometry",
}


def slot_summary(blocks):
    """Slot 0 of the loop = the blocks from the one that raises the wave's priority (the first instruction of an active block) to the one in front of the next such block
    (or of the barrier).  `optional` = blocks a forward conditional branch inside the slot jumps over (gap-letter terms, division): left out of the `least` sums."""
    idx = {b["label"]: i for i, b in enumerate(blocks)}
    prio = [i for i, b in enumerate(blocks) if any(re.match(r"s_setprio [123]$", x) for x in b["ins"])]
    bar = next(i for i, b in enumerate(blocks) if any(x.startswith("s_barrier") for x in b["ins"]))
    if not prio: return None
    lo = prio[0]
    hi = next((p for p in prio[1:] if p > lo and p < bar), bar)
    # the slot ends in front of the tests of the next slot / the barrier block: walk back over pure scalar-test blocks
    while hi - 1 > lo and all(kind(x) in ("salu", "branch") for x in blocks[hi - 1]["ins"]): hi -= 1
    optional = set()
    for i in range(lo, hi):
        last = blocks[i]["ins"][-1].split()
        if last[0].startswith("s_cbranch") and last[-1] in idx and i < idx[last[-1]] <= hi:
            for j in range(i + 1, idx[last[-1]]): optional.add(j)
    def tot(sel):
        return {"instructions": sum(len(blocks[i]["ins"]) for i in sel), "valu_units": round(sum(blocks[i]["units"] for i in sel), 1),
                "valu": sum(blocks[i]["count"].get("valu", 0) for i in sel)}
    every = list(range(lo, hi))
    return {"blocks": f"{blocks[lo]['label']} .. {blocks[hi - 1]['label']}", "all_paths": tot(every), "least": tot([i for i in every if i not in optional]),
            "slot_range": (lo, hi)}


def summarise(targs, out_dir):
    global_argv = sys.argv
    res = {}
    for which, name in (("0,0", "phase_A"), ("2,0", "phase_C")):
        sys.argv = [global_argv[0], targs, out_dir, which]
        lines = compile_one(targs)
        start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3twl17talco_lean_kernel"))
        end = next(i for i, l in enumerate(lines) if l.startswith(".Lfunc_end") and i > start)
        k = lines[start:end]
        ph, gen = which.split(",")
        mark = next(i for i, l in enumerate(k) if f"TWL_STEP PH={ph} GEN={gen}" in l)
        best = None
        for i, l in enumerate(k):
            if "Loop Header" not in l or i > mark: continue
            a = i
            while not k[a].startswith(".LBB"): a -= 1
            head_label = k[a].split(":")[0]
            backs = [j for j in range(mark, len(k)) if re.match(r"\s+s_c?branch\w*\s+%s\b" % re.escape(head_label), k[j])]
            if backs and (best is None or a > best[0]): best = (a, backs[-1])
        a, z = best
        blocks, cur = [], None
        for l in k[a:z + 1]:
            if l.startswith(".LBB"):
                cur = {"label": l.split(":")[0], "ins": []}; blocks.append(cur)
            elif l.startswith("\t") and not l.strip().startswith((";", ".")) and cur is not None:
                ins = l.strip()
                if ins.startswith(";;#"): continue
                cur["ins"].append(ins)
                if ins.split()[0].startswith(("s_cbranch", "s_branch")):
                    cur = {"label": cur["label"] + "+", "ins": []}; blocks.append(cur)
        blocks = [b for b in blocks if b["ins"]]
        for b in blocks:
            c = {}
            for ins in b["ins"]: c[kind(ins)] = c.get(kind(ins), 0) + 1
            b["count"] = c
            b["units"] = round(sum(valu_units(i) for i in b["ins"] if kind(i) == "valu"), 1)
        ss = slot_summary(blocks)
        tag = re.sub(r"[^0-9a-z]+", "_", targs.lower()).strip("_")
        fn = f"isa_step_{tag}_{name}.s"
        with open(os.path.join(out_dir, fn), "w") as f:
            f.write(f"; talco_lean_kernel{targs}: the plain {name.replace('_', ' ')} loop (one anti-diagonal per iteration), basic blocks with instruction counts and VALU units -- tools/isa_block_step.py\n")
            for i, b in enumerate(blocks):
                inslot = ss and ss["slot_range"][0] <= i < ss["slot_range"][1]
                f.write(f"; ---- {b['label']}{' [slot 0]' if inslot else ''}: {sum(b['count'].values())} instructions {json.dumps(b['count'])}, {b['units']} VALU units\n")
                for ins in b["ins"]: f.write("\t" + ins + "\n")
        if ss: ss.pop("slot_range")
        res[name] = {"listing": fn, "loop_text_instructions": sum(len(b["ins"]) for b in blocks), "loop_text_valu_units": round(sum(b["units"] for b in blocks), 1), "slot0": ss}
    sys.argv = global_argv
    return res


def main_all(out_dir):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    os.makedirs(out_dir, exist_ok=True)
    out = {"source_hash": g.source_hash(),
           "how": "tools/isa_block_step.py --all: every kernel alone through hipcc -S; the plain phase-A and phase-C loops (one anti-diagonal per iteration) are found by the comment the step "
                  "leaves in the listing, cut into basic blocks and counted.  slot0 = the text of one ACTIVE 64-row block (from the instruction that raises the wave's priority to the "
                  "tests of the next slot or the barrier): `all_paths` every instruction in it (profiles with gap letters on both sides and a general denominator run all of it), "
                  "`least` without the blocks a forward branch jumps over (gap-letter terms :394-395, the division :444).  valu_units = vector instructions weighted by their measured "
                  "issue cost (issue_rates5.log next to this file): fp32 add / mul / fma / mov on vector registers 1, anything with a scalar operand and every integer, compare, select "
                  "and DPP form 1.7, packed fp32 2.",
           "valu_ceiling": {"units_per_ns_per_cu": 3.7, "cus": 256,
                            "source": "profiles/r06/issue_rates5.log (tools/micro/issue_rates5.hip): full-rate fp32 vector instructions, 5 workgroups x 4 waves per CU 3.59-3.78 per ns and CU, "
                                      "16 waves 3.89-4.18; the slow classes 2.15-2.27 (ratio 1.7), packed fp32 1.82-1.88 (2)"},
           "issue_ceiling": {"instr_per_ns_per_cu": 3.97, "cus": 256, "source": "profiles/r02/issue_rates.log (the total-issue ceiling round 5's frac was measured against; kept for continuity)"},
           "kernels": {}}
    for targs_full, what in KERNELS.items():
        targs = targs_full[len("talco_lean_kernel"):]
        r = summarise(targs, out_dir)
        r["what"] = what
        out["kernels"][targs_full] = r
        s0 = r["phase_A"]["slot0"]
        print(targs_full, "phase A slot0:", s0["all_paths"] if s0 else None, "least", s0["least"] if s0 else None, file=sys.stderr)
    json.dump(out, open(os.path.join(out_dir, "isa_block_step.json"), "w"), indent=1)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--all":
    main_all(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r06"))
