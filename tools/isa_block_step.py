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


if __name__ == "__main__":
    main()
