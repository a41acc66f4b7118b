"""Development aid: a lone pair / a few pairs through the tile-parallel path on either geometry of its tile launches (TWL_KNOB_MT_THR_JOBS).
   python tools/lone_pair_probe.py <pairs> <length> <thr_jobs> [library.so]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from twilight_amd import synth, api
import torch
import twilight_amd as twl
n = int(sys.argv[1]); length = int(sys.argv[2]); thr = int(sys.argv[3])
if len(sys.argv) > 4 and sys.argv[4] != "base":
    api.LIB_PATH = os.path.abspath(sys.argv[4])
b = synth.make_level_batch(min(n, 8), length, members=((1, 8), (1, 8)), seed=5)
twl.init([0])
twl.set_knob(api.KNOB_MT_THR_JOBS, thr)
idx = np.arange(n) % b.n_pairs
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a[idx]).to(dev)
freq, gop, gex, ln, nm = t(b.freq), t(b.gap_open), t(b.gap_extend), t(b.len), t(b.num)
aln = torch.zeros((n, 2 * b.seq_len), dtype=torch.int8, device=dev); alen = torch.zeros(n, dtype=torch.int32, device=dev); err = torch.zeros(n, dtype=torch.int16, device=dev)
p = twl.make_params(synth.nucleotide_matrix())
best = 1e9
for r in range(5):
    torch.cuda.synchronize()
    twl.align_batch_device(p, n, b.seq_len, freq.data_ptr(), gop.data_ptr(), gex.data_ptr(), ln.data_ptr(), nm.data_ptr(), aln.data_ptr(), alen.data_ptr(), err.data_ptr())
    torch.cuda.synchronize()
    st = twl.get_stats(0)
    if r: best = min(best, st.kernel_ms)
print(f"{os.path.basename(sys.argv[4]) if len(sys.argv) > 4 else 'base':20s} pairs {n} len {length} thr_jobs {thr}: kernel {best:.3f} ms, tiles {st.mt_tiles_predicted}/{st.mt_tiles_inline}, {st.kernel.decode()[:70]}", flush=True)
