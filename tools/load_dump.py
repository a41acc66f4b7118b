"""Read the calls a TWL_DEV host library wrote with TWL_DUMP_BATCH=<file> (align_gpu.cpp) back as synth.LevelBatch objects + parameters."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twilight_amd import synth


def load(path):
    out = []
    with open(path, "rb") as f:
        while True:
            hd = np.fromfile(f, np.int32, 8)
            if hd.size < 8:
                break
            assert hd[0] == 0x7477626c
            _, P, n, stride, xdrop, flen, marker, _ = [int(x) for x in hd]
            gp = np.fromfile(f, np.float32, 4)
            M = np.fromfile(f, np.float32, (P - 1) * (P - 1)).reshape(P - 1, P - 1)
            ln = np.fromfile(f, np.int32, 2 * n).reshape(n, 2)
            nm = np.fromfile(f, np.int32, 2 * n).reshape(n, 2)
            freq = np.fromfile(f, np.float32, n * 2 * stride * P).reshape(n, 2, stride, P)
            gop = np.fromfile(f, np.float32, n * 2 * stride).reshape(n, 2, stride)
            gex = np.fromfile(f, np.float32, n * 2 * stride).reshape(n, 2, stride)
            b = synth.LevelBatch(P=P, seq_len=stride, freq=freq, gap_open=gop, gap_extend=gex, len=ln, num=nm)
            out.append((b, M, dict(xdrop=xdrop, flen=flen, marker=marker, gap_open=float(gp[0]), gap_extend=float(gp[1]), gap_char=float(gp[2]))))
    return out
