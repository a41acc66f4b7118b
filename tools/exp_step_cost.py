"""Timing experiments on copies of the library built with an experiment flag (-DTWL_EXP_SALU / -DTWL_EXP_VALU: 24 extra scalar / vector
instructions per wave and diagonal): what does a wide level and a lone pair cost then?  (The convergence-test experiment this script was
written for led to the pre-test of DESIGN.md section 3.1.)
    python tools/exp_step_cost.py [base | TWL_EXP_SALU | TWL_EXP_VALU | <label> <compiler flags...>]        (GPU box; does not touch the product library;
    prints a checksum of the paths so that a flag experiment can be compared with the base build)"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from twilight_amd import synth, api

so = os.path.join(tempfile.mkdtemp(), "libtwl_exp.so")
which = sys.argv[1] if len(sys.argv) > 1 else "base"      # base | TWL_EXP_SALU | TWL_EXP_VALU (a -D flag of an experiment build) | any label followed by compiler flags
if which.endswith(".so"):          # a library built elsewhere (cross-compiled variants travel with the repository snapshot)
    api.LIB_PATH = os.path.abspath(which)
    which = os.path.basename(which)
elif which != "base":
    extra = sys.argv[2:] if len(sys.argv) > 2 else ["-D" + which]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17"] + extra +
                          ["-o", so, os.path.join(ROOT, "twilight_amd", "csrc", "twl_align.hip")])
    api.LIB_PATH = so
api.init([0])
dev = torch.device("cuda:0")
p = api.make_params(synth.nucleotide_matrix())
cases = [(2048, 1024, 6), (1, 1024, 6), (1, 0, 6), (100, 1024, 6), (2048, 1024, 22), (64, 1024, 22)]      # pairs, MT_MAX_PAIRS, P (22: 2 kaa protein pairs)
for n, mt, P in cases:
    api.set_knob(api.KNOB_MT_MAX_PAIRS, mt)
    if P == 22: p = api.make_params(synth.protein_matrix())
    b = synth.make_level_batch(min(n, 32), 10000, members=((1, 8), (1, 8)), seed=5) if P == 6 else synth.make_level_batch(min(n, 32), 2000, members=((1, 8), (1, 8)), seed=5, P=22, sub=0.15)
    idx = np.arange(n) % b.n_pairs
    t = lambda a: torch.from_numpy(a[idx]).to(dev)
    freq, gop, gex, ln, nm = t(b.freq), t(b.gap_open), t(b.gap_extend), t(b.len), t(b.num)
    aln = torch.zeros((n, 2 * b.seq_len), dtype=torch.int8, device=dev); alen = torch.zeros(n, dtype=torch.int32, device=dev); err = torch.zeros(n, dtype=torch.int16, device=dev)
    for r in range(3):
        torch.cuda.synchronize()
        api.align_batch_device(p, n, b.seq_len, freq.data_ptr(), gop.data_ptr(), gex.data_ptr(), ln.data_ptr(), nm.data_ptr(), aln.data_ptr(), alen.data_ptr(), err.data_ptr())
        torch.cuda.synchronize()
        st = api.get_stats(0)
    import hashlib
    h = hashlib.md5(aln.cpu().numpy().tobytes() + alen.cpu().numpy().tobytes() + err.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"{which}: P {P} pairs {n} mt_max {mt}: kernel {st.kernel_ms:.2f} ms, cells {st.band_cells:.4g}, paths {h}, {st.kernel.decode()[:60]}", flush=True)
