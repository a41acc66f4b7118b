"""ctypes binding of oracle/libtwl_oracle.so -- the CPU checker.  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None


class Params(C.Structure):
    _fields_ = [("P", C.c_int32), ("matrix", C.POINTER(C.c_float)), ("gap_open", C.c_float),
                ("gap_extend", C.c_float), ("gap_char", C.c_float), ("xdrop", C.c_int32),
                ("flen", C.c_int32), ("marker", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("cells", C.c_uint64), ("diags", C.c_uint64), ("tiles", C.c_int32), ("max_width", C.c_int32),
                ("empty_reduce", C.c_uint64), ("oob_diag", C.c_uint64), ("err3_reason", C.c_int32), ("err3_tile", C.c_int32)]


TRACE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "libtwl_oracle.so")
        src = os.path.join(ORACLE_DIR, "talco_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "libtwl_oracle.so"], stdout=subprocess.DEVNULL)
        _LIB = C.CDLL(so)
        _LIB.twlo_align_batch.restype = C.c_int
        _LIB.twlo_align_pair.restype = C.c_int
        _LIB.twlo_column_score.restype = C.c_float
    return _LIB


def make_params(matrix: np.ndarray, *, gap_open=-50.0, gap_extend=-5.0, gap_char=None, xdrop=None, flen=4096,
                marker=1024):
    """Talco_xdrop::Params defaults (reference TALCO-XDrop.cpp:36-53): xdrop = 1000*-gapExtend."""
    m = np.ascontiguousarray(matrix, dtype=np.float32)
    p = Params()
    p.P = m.shape[0] + 1
    p.matrix = m.ctypes.data_as(C.POINTER(C.c_float))
    p.gap_open = gap_open
    p.gap_extend = gap_extend
    p.gap_char = gap_extend if gap_char is None else gap_char
    p.xdrop = int(1000 * -gap_extend) if xdrop is None else int(xdrop)
    p.flen = flen
    p.marker = marker
    p._keep = m   # keep the matrix alive
    return p


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def align_batch(params: Params, batch, threads: int = 1):
    """Run the oracle over a LevelBatch; returns (aln[n][2*seq_len] int8, aln_len[n], err[n], Stats)."""
    n, sl = batch.n_pairs, batch.seq_len
    aln = np.zeros((n, 2 * sl), dtype=np.int8)
    aln_len = np.zeros(n, dtype=np.int32)
    err = np.zeros(n, dtype=np.int16)
    st = Stats()
    freq = np.ascontiguousarray(batch.freq, dtype=np.float32)
    go = np.ascontiguousarray(batch.gap_open, dtype=np.float32)
    ge = np.ascontiguousarray(batch.gap_extend, dtype=np.float32)
    ln = np.ascontiguousarray(batch.len, dtype=np.int32)
    nm = np.ascontiguousarray(batch.num, dtype=np.int32)
    rc = lib().twlo_align_batch(C.byref(params), C.c_int32(n), C.c_int32(sl), _fp(freq), _fp(go), _fp(ge),
                                ln.ctypes.data_as(C.POINTER(C.c_int32)), nm.ctypes.data_as(C.POINTER(C.c_int32)),
                                aln.ctypes.data_as(C.POINTER(C.c_int8)), aln_len.ctypes.data_as(C.POINTER(C.c_int32)),
                                err.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int32(threads), C.byref(st))
    assert rc == 0
    return aln, aln_len, err, st


def align_pair(params: Params, ref, qry, gop_r, gex_r, gop_q, gex_q, ref_num, qry_num, trace=None):
    ref = np.ascontiguousarray(ref, dtype=np.float32)
    qry = np.ascontiguousarray(qry, dtype=np.float32)
    R, Q = ref.shape[0], qry.shape[0]
    aln = np.zeros(R + Q, dtype=np.int8)
    n = C.c_int32(0)
    err = C.c_int16(0)
    st = Stats()
    arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (gop_r, gex_r, gop_q, gex_q)]
    cb = TRACE_FN(trace) if trace is not None else C.cast(None, TRACE_FN)
    lib().twlo_align_pair(C.byref(params), _fp(ref), C.c_int32(R), _fp(qry), C.c_int32(Q), *[_fp(a) for a in arrs],
                          C.c_float(ref_num), C.c_float(qry_num), aln.ctypes.data_as(C.POINTER(C.c_int8)),
                          C.byref(n), C.byref(err), C.byref(st), cb, None)
    return aln[: n.value].copy(), int(err.value), st


def column_score(params: Params, r, q, denom):
    r = np.ascontiguousarray(r, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    return float(lib().twlo_column_score(C.byref(params), _fp(r), _fp(q), C.c_float(denom)))


_FLIB = None


def faithful_lib():
    """oracle/libtwl_faithful.so: the checker's algorithm in the reference's data layout and allocation pattern (what bench.py times as the
    reference's own code); held to the checker bit for bit by tests/test_oracle_cpu.py."""
    global _FLIB
    if _FLIB is None:
        so = os.path.join(ORACLE_DIR, "libtwl_faithful.so")
        src = os.path.join(ORACLE_DIR, "talco_faithful.cpp")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "libtwl_faithful.so"], stdout=subprocess.DEVNULL)
        _FLIB = C.CDLL(so)
        _FLIB.twlf_align_batch.restype = C.c_int
    return _FLIB


def align_batch_faithful(params: Params, batch, threads: int = 1):
    """Same outputs as align_batch (paths, lengths, error codes) + the band-cell count, from the reference-layout restatement."""
    n, sl = batch.n_pairs, batch.seq_len
    aln = np.zeros((n, 2 * sl), dtype=np.int8)
    aln_len = np.zeros(n, dtype=np.int32)
    err = np.zeros(n, dtype=np.int16)
    cells = C.c_uint64(0)
    freq = np.ascontiguousarray(batch.freq, dtype=np.float32)
    go = np.ascontiguousarray(batch.gap_open, dtype=np.float32)
    ge = np.ascontiguousarray(batch.gap_extend, dtype=np.float32)
    ln = np.ascontiguousarray(batch.len, dtype=np.int32)
    nm = np.ascontiguousarray(batch.num, dtype=np.int32)
    rc = faithful_lib().twlf_align_batch(C.byref(params), C.c_int32(n), C.c_int32(sl), _fp(freq), _fp(go), _fp(ge),
                                         ln.ctypes.data_as(C.POINTER(C.c_int32)), nm.ctypes.data_as(C.POINTER(C.c_int32)),
                                         aln.ctypes.data_as(C.POINTER(C.c_int8)), aln_len.ctypes.data_as(C.POINTER(C.c_int32)),
                                         err.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int32(threads), C.byref(cells))
    assert rc == 0
    return aln, aln_len, err, int(cells.value)
