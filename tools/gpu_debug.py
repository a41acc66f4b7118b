"""Staged GPU bring-up: each case runs in its own subprocess under a short timeout and logs to gpurun_out/.

Usage on the GPU box:  python tools/gpu_debug.py [case ...]
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)

CASES = {
    # name: (n_pairs, length, members, params)
    "one_tiny": (1, 30, (1, 1), {}),
    "one_300": (1, 300, (1, 1), {}),
    "one_700_two_tiles": (1, 700, (1, 1), {}),
    "one_700_marker64": (1, 700, (1, 1), {"marker": 64}),
    "twelve_600": (12, 600, (1, 1), {}),
    "profiles_600": (12, 600, ((2, 6), (2, 6)), {}),
}


def child(name):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    t0 = time.time()

    def log(msg):
        print(f"[{name} +{time.time() - t0:6.2f}s] {msg}", flush=True)

    import numpy as np

    log("imports")
    import twilight_amd as twl
    from twilight_amd import synth
    import oracle_lib as O

    n, length, members, pk = CASES[name]
    M = synth.nucleotide_matrix()
    batch = synth.make_level_batch(n, length, members=members, seed=11)
    log(f"batch built len={batch.len.tolist()}")
    oa, on, oerr, ost = O.align_batch(O.make_params(M, **pk), batch, threads=4)
    log(f"oracle: err={oerr.tolist()} n={on.tolist()} cells={ost.cells} tiles={ost.tiles} maxw={ost.max_width}")
    twl.init([0])
    log("twl.init ok")
    aln, gn, gerr = twl.align_batch(twl.make_params(M, **pk), batch)
    st = twl.get_stats(0)
    log(f"gpu: err={gerr.tolist()} n={gn.tolist()} cells={st.band_cells} kernel_ms={st.kernel_ms:.3f} grid={st.grid}")
    ok = np.array_equal(gerr, oerr) and np.array_equal(gn, on) and st.band_cells == ost.cells
    for i in range(n):
        same = np.array_equal(aln[i, : gn[i]], oa[i, : on[i]])
        ok &= same
        if not same:
            m = min(gn[i], on[i])
            d = np.flatnonzero(aln[i, :m] != oa[i, :m])
            log(f"pair {i}: MISMATCH first at {d[0] if d.size else m} gpu_len={gn[i]} oracle_len={on[i]}")
            lo = max(0, (d[0] if d.size else m) - 5)
            log(f"   gpu    {aln[i, lo:lo + 30].tolist()}")
            log(f"   oracle {oa[i, lo:lo + 30].tolist()}")
    log("PARITY OK" if ok else "PARITY FAIL")


def main():
    names = sys.argv[1:] or list(CASES)
    if names[0] == "--child":
        child(names[1])
        return
    env = dict(os.environ, TWL_DEBUG="1")
    for name in names:
        log = os.path.join(OUT, f"debug_{name}.log")
        with open(log, "w") as f:
            try:
                rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--child", name], stdout=f, stderr=subprocess.STDOUT,
                                     env=env, timeout=60)
            except subprocess.TimeoutExpired:
                rc = "TIMEOUT"
        print(f"== {name}: rc={rc}")
        print(open(log).read()[-3000:])
        if rc == "TIMEOUT":
            print("stopping after first hang")
            break


if __name__ == "__main__":
    main()
