"""Build a stamped debug copy of the library and print where one workgroup's diagonal step spends its cycles.

    python tools/step_stamps.py [n_pairs] [length]     (GPU box; does not touch the product library)
"""
import ctypes as C, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (share one HIP runtime)
from twilight_amd import synth, api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
length = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
so = os.path.join(tempfile.mkdtemp(), "libtwl_stamps.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17",
                       "-DTWL_KERNEL_STAMPS", "-DTWL_DEV", "-o", so, os.path.join(ROOT, "twilight_amd", "csrc", "twl_align.hip")])
api.LIB_PATH = so
os.environ["TWL_DEBUG"] = "1"
b = synth.make_level_batch(min(n, 32), length, members=((1, 8), (1, 8)), seed=5)
idx = np.arange(n) % b.n_pairs
big = synth.LevelBatch(P=b.P, seq_len=b.seq_len, freq=b.freq[idx], gap_open=b.gap_open[idx], gap_extend=b.gap_extend[idx], len=b.len[idx], num=b.num[idx])
api.init([0])
api.set_knob(api.KNOB_MT_MAX_PAIRS, 0)      # the plain tile loop, not the tile-parallel path
api.align_batch(api.make_params(synth.nucleotide_matrix()), big)
lib = api.load_library()
lib.twl_debug_read.restype = C.c_int
buf = (C.c_longlong * 136)()
rc = lib.twl_debug_read(0, buf, 136)
print("rc", rc)
for w in range(16):
    g = [buf[8 * w + t] for t in range(8)]
    if g[3]:
        print(f"wave {w}: steps {g[3]} active(slot0) {g[4]}  cycles/step: slots {g[0]/g[3]:.0f}  barrier wait {g[1]/g[3]:.0f}  post {g[2]/g[3]:.0f}  total/step {g[5]/g[3]:.0f};  whole pair {g[5]}  in steps {g[0]+g[1]+g[2]}  tile setup {g[7]}  tile exit+traceback {g[6]}")
