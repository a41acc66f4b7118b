"""Development aid: compile ONE talco_lean_kernel instantiation to gfx950 assembly (seconds, no GPU) and report its registers, scratch, LDS, and the scratch /
spill traffic inside its diagonal loops (loop depth >= 3: a reload there sits on every anti-diagonal's critical path).

    python tools/kernel_check.py "<6, 4, 2, 2, 5, false, false, 0>" [more instantiations ...]      ->  one line each; the listing stays in /tmp/twl_kc_<tag>.s"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "twilight_amd", "csrc")


def check(targs, extra=()):
    tag = re.sub(r"[^0-9a-z]+", "_", targs.lower()).strip("_")
    d = tempfile.mkdtemp(prefix="twl_kc_")
    src = os.path.join(d, "k.hip")
    open(src, "w").write('#include <type_traits>\n#include "talco_nuc.hip.h"\ntemplate __global__ void twl::talco_lean_kernel%s(twl::NArgs);\n' % targs)
    out = f"/tmp/twl_kc_{tag}.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-I" + CSRC, "-S",
                        "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", out, src] + list(extra), capture_output=True, text=True)
    if r.returncode:
        print(targs, "FAILED\n", r.stderr[-2000:]); return
    res = {}
    on = False
    for l in r.stderr.splitlines():
        if "Function Name" in l: on = "talco_lean_kernel" in l
        m = re.search(r"remark:\s+(\S[^:]*): (\S+)", l)
        if on and m: res[m.group(1).strip()] = m.group(2)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3twl17talco_lean_kernel"))
    # a block's loop depth = the deepest "Depth=" of its label line and of the comment lines under it; then: scratch traffic per depth
    depth, pending, by_depth = 0, False, {}
    for l in lines[start:]:
        if l.startswith(".LBB") or l.startswith("; %bb."):
            m = re.findall(r"Depth=(\d+)", l)
            depth = max([int(x) for x in m], default=0)
            pending = True
            continue
        if pending and l.lstrip().startswith(";"):
            m = re.findall(r"Depth=(\d+)", l)
            if m: depth = max(depth, max(int(x) for x in m))
            continue
        pending = False
        if "scratch_" in l: by_depth[depth] = by_depth.get(depth, 0) + 1
    inloop = {d: n for d, n in sorted(by_depth.items()) if d >= 3}
    print(f"{targs:48s} VGPR {res.get('VGPRs')} SGPR {res.get('TotalSGPRs')} scratch {res.get('ScratchSize [bytes/lane]')} B spillV {res.get('VGPRs Spill')} spillS {res.get('SGPRs Spill')} "
          f"LDS {res.get('LDS Size [bytes/block]')} occ {res.get('Occupancy [waves/SIMD]')} | scratch ops by loop depth (>= 3): {inloop}")


if __name__ == "__main__":
    for t in sys.argv[1:]:
        check(t)
