"""Wall-clock to final MSA on a synthetic RNASim-shaped family: product CLI (GPU) vs the CPU checker, same box.

    python tools/e2e_bench.py --leaves 1000 --length 1600 [--type n|p] [--cpu] [--out gpurun_out/e2e.json]
"""
import argparse, hashlib, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twilight_amd import synth  # noqa: E402


def run(cmd):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True)
    return time.perf_counter() - t0, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leaves", type=int, default=1000)
    ap.add_argument("--length", type=int, default=1600)
    ap.add_argument("--type", default="n")
    ap.add_argument("--cpu", action="store_true", help="also run oracle/e2e_oracle (CPU) and compare the MSAs")
    ap.add_argument("--sub", type=float, default=0.015)
    ap.add_argument("--indel", type=float, default=0.001)
    ap.add_argument("--seed", type=int, default=20260501)
    ap.add_argument("--out", default="")
    ap.add_argument("--keep", default="", help="write the generated tree/sequences into this directory (t.nwk, s.fa) and keep them")
    ap.add_argument("--generate-only", action="store_true")
    a = ap.parse_args()
    P = 6 if a.type == "n" else 22
    sys.setrecursionlimit(100000)
    t0 = time.perf_counter()
    nwk, seqs = synth.make_family(a.leaves, a.length, P=P, seed=a.seed, sub=a.sub, indel=a.indel)
    d = a.keep or tempfile.mkdtemp(prefix="twl_e2e_")
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "t.nwk"), "w").write(nwk + "\n")
    with open(os.path.join(d, "s.fa"), "w") as f:
        for name, s in seqs:
            f.write(f">{name}\n{s}\n")
    gen = time.perf_counter() - t0
    res = {"leaves": a.leaves, "length": a.length, "type": a.type, "generate_s": gen}
    if a.generate_only:
        print(json.dumps(res)); return
    gpu_out = os.path.join(d, "gpu.aln")
    wall, r = run([os.path.join(ROOT, "twilight_amd", "twilight-mi355x"), "-t", os.path.join(d, "t.nwk"), "-i", os.path.join(d, "s.fa"), "-o", gpu_out,
                   "--type", a.type, "-v"])
    if r.returncode != 0:
        print(r.stderr[-3000:]); sys.exit(1)
    tail = [l for l in r.stderr.splitlines() if l.startswith("Wrote")][-1]
    levels = [l for l in r.stderr.splitlines() if l.startswith("Level ")]
    phases = [l for l in r.stderr.splitlines() if l.startswith("Host phases") or l.startswith("Driver phases")]
    res["gpu"] = {"wall_s": wall, "summary": tail, "phases": phases, "levels": len(levels), "md5": hashlib.md5(open(gpu_out, "rb").read()).hexdigest()}
    first = open(gpu_out).readlines()[1].strip()
    res["aln_len"] = len(first)
    if a.out:
        open(a.out + ".stderr.txt", "w").write(r.stderr)
    if a.cpu:
        cpu_out = os.path.join(d, "cpu.aln")
        wall_c, rc = run([os.path.join(ROOT, "oracle", "e2e_oracle"), "-t", os.path.join(d, "t.nwk"), "-i", os.path.join(d, "s.fa"), "-o", cpu_out, "--type", a.type])
        line = [l for l in rc.stdout.splitlines() if l.startswith("E2E")]
        res["cpu"] = {"wall_s": wall_c, "summary": line[-1] if line else rc.stderr[-500:], "md5": hashlib.md5(open(cpu_out, "rb").read()).hexdigest() if os.path.exists(cpu_out) else None}
        res["msa_equal"] = res["cpu"]["md5"] == res["gpu"]["md5"]
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
    if not a.keep:                      # the family and the (possibly GB-sized) MSAs were only needed for the comparison
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
