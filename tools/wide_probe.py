"""Development aid: pairs whose band outgrows the 1024-row window (a large X-drop on diverged pairs) through twl_align_batch_device, with the
kernel time of the call and the re-run counters.    python tools/wide_probe.py <pairs> <length> <xdrop> [sub] [reps] [wide 0/1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from twilight_amd import synth, api

n = int(sys.argv[1]); length = int(sys.argv[2]); xdrop = int(sys.argv[3])
sub = float(sys.argv[4]) if len(sys.argv) > 4 else 0.12
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
wide = int(sys.argv[6]) if len(sys.argv) > 6 else 1
pool = min(n, 8)
b = synth.make_level_batch(pool, length, members=((1, 6), (1, 6)), seed=101, sub=sub, indel=0.01)
import torch
import twilight_amd as twl
twl.init([0])
twl.set_knob(api.KNOB_MT_WIDE, wide)
idx = np.arange(n) % pool
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a[idx]).to(dev)
freq, gop, gex, ln, nm = t(b.freq), t(b.gap_open), t(b.gap_extend), t(b.len), t(b.num)
aln = torch.zeros((n, 2 * b.seq_len), dtype=torch.int8, device=dev); alen = torch.zeros(n, dtype=torch.int32, device=dev); err = torch.zeros(n, dtype=torch.int16, device=dev)
p = twl.make_params(synth.nucleotide_matrix(), xdrop=xdrop)
for r in range(reps):
    torch.cuda.synchronize(); t0 = time.time()
    twl.align_batch_device(p, n, b.seq_len, freq.data_ptr(), gop.data_ptr(), gex.data_ptr(), ln.data_ptr(), nm.data_ptr(), aln.data_ptr(), alen.data_ptr(), err.data_ptr())
    torch.cuda.synchronize(); dt = time.time() - t0
    st = twl.get_stats(0)
    print(f"rep {r}: wide={wide} n={n} len={length} xdrop={xdrop}: cells={st.band_cells:.4g} kernel={st.kernel_ms:.2f} ms wall={dt*1e3:.2f} ms relaunched={st.n_relaunched} launches={st.n_launches} "
          f"tiles predicted/inline {st.mt_tiles_predicted}/{st.mt_tiles_inline} errs={int((err != 0).sum())}", flush=True)
