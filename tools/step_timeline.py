"""Stamped debug build: the raw per-diagonal timeline of one workgroup (tile 4 of a lone pair), 96 diagonals of phase A and 96 of phase C.

    python tools/step_timeline.py [length]      (GPU box; does not touch the product library)
Prints, per diagonal, for every wave: cycles from the diagonal's first head stamp to its head / barrier arrival / barrier exit / end.
"""
import ctypes as C, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from twilight_amd import synth, api

length = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
so = os.environ.get("TWL_STAMPS_SO") or os.path.join(tempfile.mkdtemp(), "libtwl_stamps.so")      # (a library cross-compiled with -DTWL_KERNEL_STAMPS -DTWL_DEV saves the GPU box the build)
if not os.environ.get("TWL_STAMPS_SO"):
  subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17",
                       "-DTWL_KERNEL_STAMPS", "-DTWL_DEV", "-o", so, os.path.join(ROOT, "twilight_amd", "csrc", "twl_align.hip")])
api.LIB_PATH = so
os.environ["TWL_DEBUG"] = "1"
b = synth.make_level_batch(1, length, members=((1, 8), (1, 8)), seed=5)
api.init([0])
api.set_knob(api.KNOB_MT_MAX_PAIRS, 0)      # the plain tile loop, not the tile-parallel path
if not os.environ.get("TWL_TIMELINE_SPEC"):      # TWL_TIMELINE_SPEC=1: the speculative kernel (tile 4 is one of workgroup 0's)
    api.set_knob(api.KNOB_NO_SPEC, 1)
api.align_batch(api.make_params(synth.nucleotide_matrix()), b)
lib = api.load_library()
lib.twl_debug_read.restype = C.c_int
n = 256 + 16 * 192 * 4
buf = (C.c_longlong * n)()
print("rc", lib.twl_debug_read(0, buf, n))
a = np.array(buf[256:], dtype=np.int64).reshape(16, 192, 4)
for name, lo in (("phase A, k = 600..", 0), ("phase C, k = marker+100..", 96)):
    print(name)
    seg = a[:, lo:lo + 96, :]
    prev_end = None
    for t in range(0, 96):
        if not seg[:, t, 0].any(): continue
        t0 = seg[:, t, 0].min()
        head = seg[:, t, 0] - t0; slots = seg[:, t, 1] - t0; bar = seg[:, t, 2] - t0; end = seg[:, t, 3] - t0
        gap = "" if prev_end is None else " since last end %d" % (t0 - prev_end)
        prev_end = seg[:, t, 3].max()
        if t < 24 or t % 8 == 0:
            print(" diag +%2d: step %5d cycles%s" % (t, seg[:, t, 3].max() - t0, gap))
            print("   head  ", " ".join("%4d" % x for x in head))
            print("   arrive", " ".join("%4d" % x for x in slots))
            print("   leave ", " ".join("%4d" % x for x in bar))
            print("   end   ", " ".join("%4d" % x for x in end))
    tot = seg[:, 95, 3].max() - seg[:, 0, 0].min()
    print(" 96 diagonals: %d cycles, %.0f per diagonal; busiest wave: work %.0f, at barrier %.0f, post %.0f" % (
        tot, tot / 96.0, (seg[:, :, 1] - seg[:, :, 0]).mean(1).max(), (seg[:, :, 2] - seg[:, :, 1]).mean(1).min(), (seg[:, :, 3] - seg[:, :, 2]).mean(1).max()))
