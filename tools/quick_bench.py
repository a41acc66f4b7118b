"""Quick device-resident timing of the DP kernel (development aid; bench.py is the contract)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from twilight_amd import synth

def main():
    n = int(sys.argv[1]); length = int(sys.argv[2]); mem = sys.argv[3] if len(sys.argv) > 3 else "prof"
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    members = (1, 1) if mem.startswith("leaf") else (((32, 64), (32, 64)) if mem.startswith("deep") else ((1, 8), (1, 8)))
    prot = mem.endswith("_p")
    pool = min(n, 32)
    t0 = time.time()
    b = synth.make_level_batch(pool, length, members=members, seed=5, P=(22 if prot else 6), sub=((0.4 if mem.startswith("deep") else 0.15) if prot else 0.06))
    print(f"gen {pool} pairs in {time.time()-t0:.1f}s", flush=True)
    import torch
    import twilight_amd as twl
    twl.init([0])
    idx = np.arange(n) % pool
    dev = torch.device("cuda:0")
    freq = torch.from_numpy(b.freq[idx]).to(dev); gop = torch.from_numpy(b.gap_open[idx]).to(dev); gex = torch.from_numpy(b.gap_extend[idx]).to(dev)
    ln = torch.from_numpy(b.len[idx]).to(dev); nm = torch.from_numpy(b.num[idx]).to(dev)
    aln = torch.zeros((n, 2 * b.seq_len), dtype=torch.int8, device=dev); alen = torch.zeros(n, dtype=torch.int32, device=dev); err = torch.zeros(n, dtype=torch.int16, device=dev)
    p = twl.make_params(synth.protein_matrix() if prot else synth.nucleotide_matrix())
    bcell = 192 if prot else 64
    for r in range(reps):
        torch.cuda.synchronize(); t = time.time()
        twl.align_batch_device(p, n, b.seq_len, freq.data_ptr(), gop.data_ptr(), gex.data_ptr(), ln.data_ptr(), nm.data_ptr(), aln.data_ptr(), alen.data_ptr(), err.data_ptr())
        torch.cuda.synchronize(); dt = time.time() - t
        st = twl.get_stats(0)
        print(f"rep {r}: n={n} len={length} {mem}: cells={st.band_cells:.4g} kernel={st.kernel_ms:.2f}ms pack={st.pack_ms:.3f}ms wall={dt*1e3:.2f}ms  "
              f"{st.band_cells/st.kernel_ms/1e6:.3f} Gcells/s  roofline={st.band_cells/st.kernel_ms*1e3*bcell/8e12:.4f} grid={st.grid} errs={int((err!=0).sum())} relaunched={st.n_relaunched}", flush=True)
main()
