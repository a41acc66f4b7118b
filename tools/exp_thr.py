"""Throughput-geometry experiments: one wide level (2048 pairs x 10 kbp, the bench's peak_level workload) and the leaf-shaped variant on a
cross-compiled variant of the library (build_exp/*.so).   python tools/exp_thr.py <library.so | base> [pairs] [length] [TWL_KNOB_THR_SMALL]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from twilight_amd import synth, api
which = sys.argv[1] if len(sys.argv) > 1 else "base"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
length = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
small = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if which != "base":
    api.LIB_PATH = os.path.abspath(which)
api.init([0])
if small:
    api.set_knob(api.KNOB_THR_SMALL, small)
dev = torch.device("cuda:0")
p = api.make_params(synth.nucleotide_matrix())
for label, members, sub in (("profiles 1-8 x 1-8", ((1, 8), (1, 8)), 0.06), ("bench-like 0.03", ((1, 8), (1, 8)), 0.03), ("leaf x leaf", (1, 1), 0.03)):
    api.set_knob(api.KNOB_ASSUME_ONEHOT_QUERY, 1 if members == (1, 1) else 0)
    b = synth.make_level_batch(64, length, members=members, seed=20260501 + 3, sub=sub)
    idx = np.arange(n) % b.n_pairs
    t = lambda a: torch.from_numpy(a[idx]).to(dev)
    freq, gop, gex, ln, nm = t(b.freq), t(b.gap_open), t(b.gap_extend), t(b.len), t(b.num)
    aln = torch.zeros((n, 2 * b.seq_len), dtype=torch.int8, device=dev); alen = torch.zeros(n, dtype=torch.int32, device=dev); err = torch.zeros(n, dtype=torch.int16, device=dev)
    best = 1e9
    for r in range(4):
        torch.cuda.synchronize()
        api.align_batch_device(p, n, b.seq_len, freq.data_ptr(), gop.data_ptr(), gex.data_ptr(), ln.data_ptr(), nm.data_ptr(), aln.data_ptr(), alen.data_ptr(), err.data_ptr())
        torch.cuda.synchronize()
        st = api.get_stats(0)
        if r: best = min(best, st.kernel_ms)
    h = hashlib.md5(aln.cpu().numpy().tobytes() + alen.cpu().numpy().tobytes() + err.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"{os.path.basename(which):22s} {label:20s} pairs {n}: kernel {best:8.2f} ms  {st.band_cells / best / 1e6:7.2f} Gcells/s  relaunched {st.n_relaunched:4d} grid {st.grid} paths {h}  {st.kernel.decode()[:48]}", flush=True)
