"""Randomized end-to-end campaign: product CLI on the device-resident path, product CLI on the host-staged path and the CPU
checker must write the same MSA.  Random families, options and (test-only) low thresholds for the cached-profile branch.

    python tools/fuzz_e2e.py [n_cases] [first_seed]        (GPU box)
"""
import hashlib, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twilight_amd import synth

CLI = os.path.join(ROOT, "twilight_amd", "twilight-mi355x")
CPU = os.path.join(ROOT, "oracle", "e2e_oracle")


def md5(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


def one(seed):
    rng = np.random.default_rng(seed)
    kind = "p" if rng.random() < 0.3 else "n"
    n = int(rng.integers(6, 90))
    length = int(rng.integers(150, 1400 if kind == "n" else 600))
    sub = float(rng.choice([0.01, 0.03, 0.08, 0.2, 0.35]))
    indel = float(rng.choice([0.001, 0.005, 0.02]))
    flags = []
    if rng.random() < 0.5:
        flags += ["-r", str(rng.choice([0.5, 0.7, 0.9, 1]))]
    if rng.random() < 0.3:
        flags += ["-w"]
    if rng.random() < 0.3:
        flags += ["--gap-open", str(int(rng.integers(-80, -10))), "--gap-extend", str(int(rng.integers(-8, -1)))]
    if rng.random() < 0.3:
        flags += ["--length-deviation", str(rng.choice([0.003, 0.01, 0.03]))]
        if rng.random() < 0.4:
            flags += ["--filter"]
    if rng.random() < 0.2:
        flags += ["--rooted"]
    env = dict(os.environ)
    th = int(rng.choice([0, 3, 6, 12]))
    if th:
        flags = flags + ["--test-cal-profile-th", str(th), "--test-update-seq-th", str(int(rng.choice([th, 2 * th])))]
    replicas = int(rng.choice([1, 1, 2, 3]))
    d = tempfile.mkdtemp(prefix="twl_fz_")
    sys.setrecursionlimit(100000)
    nwk, seqs = synth.make_family(n, length, P=(6 if kind == "n" else 22), seed=seed, sub=sub, indel=indel)
    open(os.path.join(d, "t.nwk"), "w").write(nwk + "\n")
    open(os.path.join(d, "s.fa"), "w").write("".join(f">{a}\n{b}\n" for a, b in seqs))
    res = {}
    for tag, exe, extra in (("cpu", CPU, []), ("resident", CLI, []), ("staged", CLI, ["--host-staged"])):
        out = os.path.join(d, tag + ".aln")
        r = subprocess.run([exe, "-t", os.path.join(d, "t.nwk"), "-i", os.path.join(d, "s.fa"), "-o", out, "--type", kind, "--check"] + flags + extra +
                           (["--test-virtual-devices", str(replicas)] if tag == "resident" else []), capture_output=True, text=True, env=env)
        if r.returncode != 0:
            res[tag] = f"rc {r.returncode}: " + (r.stdout + r.stderr)[-300:].replace("\n", " | ")
        else:
            res[tag] = md5(out)
    ok = res["cpu"] == res["resident"] == res["staged"] and not res["cpu"].startswith("rc")
    same_fail = all(v.startswith("rc") for v in res.values())
    desc = f"seed {seed} {kind} n={n} len={length} sub={sub} indel={indel} th={th} replicas={replicas} flags={' '.join(flags)}"
    return ok, same_fail, desc, res


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = agree_fail = 0
    for seed in range(first, first + n_cases):
        ok, same_fail, desc, res = one(seed)
        if ok:
            continue
        if same_fail:
            agree_fail += 1
            print("ALL THREE FAILED ALIKE:", desc, res["cpu"][:160])
            continue
        bad += 1
        print("MISMATCH:", desc, res)
    print(f"fuzz_e2e: {n_cases} cases, {bad} mismatches, {agree_fail} cases where all three runs stopped with an error")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
