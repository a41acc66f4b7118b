import ctypes as C, os, subprocess, sys, tempfile
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
import numpy as np
import torch
from twilight_amd import synth, api
so = os.path.join(tempfile.mkdtemp(), "libtwl_dbg.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17", "-w", "-DTWL_SPEC_DEBUG", "-o", so, os.path.join(ROOT, "twilight_amd", "csrc", "twl_align.hip")])
api.LIB_PATH = so
b = synth.make_level_batch(1, 10000, members=((1, 8), (1, 8)), seed=5)
api.init([0])
api.align_batch(api.make_params(synth.nucleotide_matrix()), b)
