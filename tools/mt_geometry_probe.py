"""Development: kernel time of tile-parallel levels of n pairs x 10 kbp on the two geometries of the scout / tile launches.
    python tools/mt_geometry_probe.py      (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from twilight_amd import synth, api

api.init([0])
dev = torch.device("cuda:0")
p = api.make_params(synth.nucleotide_matrix())
b0 = synth.make_level_batch(32, 10000, members=((1, 8), (1, 8)), seed=5)
for n in (1, 4, 12, 24, 48):
    for thr in (512, 0):
        api.set_knob(api.KNOB_MT_THR_JOBS, thr)
        idx = np.arange(n) % b0.n_pairs
        t = lambda a: torch.from_numpy(a[idx]).to(dev)
        freq, gop, gex, ln, nm = t(b0.freq), t(b0.gap_open), t(b0.gap_extend), t(b0.len), t(b0.num)
        aln = torch.zeros((n, 2 * b0.seq_len), dtype=torch.int8, device=dev); alen = torch.zeros(n, dtype=torch.int32, device=dev); err = torch.zeros(n, dtype=torch.int16, device=dev)
        best = 1e9
        for r in range(4):
            torch.cuda.synchronize()
            api.align_batch_device(p, n, b0.seq_len, freq.data_ptr(), gop.data_ptr(), gex.data_ptr(), ln.data_ptr(), nm.data_ptr(), aln.data_ptr(), alen.data_ptr(), err.data_ptr())
            torch.cuda.synchronize()
            best = min(best, api.get_stats(0).kernel_ms)
        print(f"pairs {n:3d} thr_jobs {thr:3d}: kernel {best:.2f} ms", flush=True)
