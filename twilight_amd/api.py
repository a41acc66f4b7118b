"""ctypes binding of libtwl_align.so (C ABI: include/twl_align.h).  No fallback path."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TWL_LIB") or os.path.join(_HERE, "libtwl_align.so")   # TWL_LIB: development override for A/B builds
TWL_MAX_MATRIX = 21

_lib = None


class TwlError(RuntimeError):
    pass


class TwlParams(C.Structure):
    """twl_params -- replaces Talco_xdrop::Params (reference TALCO-XDrop.hpp:37-54)."""
    _fields_ = [("P", C.c_int32), ("matrix", C.c_float * (TWL_MAX_MATRIX * TWL_MAX_MATRIX)), ("gap_open", C.c_float),
                ("gap_extend", C.c_float), ("gap_boundary", C.c_float), ("gap_char", C.c_float), ("xdrop", C.c_int32),
                ("flen", C.c_int32), ("marker", C.c_int32)]


class TwlStats(C.Structure):
    _fields_ = [("band_cells", C.c_uint64), ("nominal_cells", C.c_uint64), ("kernel_ms", C.c_double),
                ("pack_ms", C.c_double), ("total_ms", C.c_double), ("n_launches", C.c_int32),
                ("n_relaunched", C.c_int32), ("window", C.c_int32), ("grid", C.c_int32), ("matrix_mode", C.c_int32), ("speculative", C.c_int32),
                ("mt_tiles_predicted", C.c_int32), ("mt_tiles_inline", C.c_int32), ("mt_scouts_failed", C.c_int32), ("reserved", C.c_int32),
                ("kernel", C.c_char * 160)]


_SYMBOLS = ["twl_init", "twl_shutdown", "twl_last_error", "twl_version", "twl_align_batch", "twl_align_batch_device",
            "twl_get_stats", "twl_get_pair_cells", "twl_column_scores", "twl_dp_column_scores", "twl_host_alloc", "twl_host_free", "twl_set_knob",
            "twl_copy_to_device", "twl_copy_from_device", "twl_copy_rows_from_device",
            "twl_comm_unique_id", "twl_comm_init", "twl_comm_all_gather", "twl_comm_all_gather_host", "twl_comm_destroy", "twl_plan_describe"]


def exported_symbols():
    return list(_SYMBOLS)


def load_library():
    """dlopen the in-tree HIP library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TwlError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        # PyTorch-ROCm bundles its own libamdhip64.  Two HIP runtimes in one process each try to own the device and the second
        # one sees "no GPUs", so when torch is installed it is imported first and our library binds to the runtime already loaded.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        lib = C.CDLL(LIB_PATH)
        lib.twl_last_error.restype = C.c_char_p
        lib.twl_version.restype = C.c_char_p
        lib.twl_init.restype = C.c_int
        lib.twl_align_batch.restype = C.c_int
        lib.twl_align_batch_device.restype = C.c_int
        lib.twl_get_stats.restype = C.c_int
        lib.twl_get_pair_cells.restype = C.c_int
        lib.twl_column_scores.restype = C.c_int
        lib.twl_dp_column_scores.restype = C.c_int
        lib.twl_shutdown.restype = None
        lib.twl_set_knob.restype = C.c_int
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise TwlError(f"libtwl_align error {rc}: {load_library().twl_last_error().decode()}")


def make_params(matrix, *, gap_open=-50.0, gap_extend=-5.0, gap_boundary=None, gap_char=None, xdrop=None, flen=4096,
                marker=1024) -> TwlParams:
    """Defaults follow Talco_xdrop::Params(msa::Params&), reference TALCO-XDrop.cpp:36-53."""
    m = np.asarray(matrix, dtype=np.float32)
    n = m.shape[0]
    p = TwlParams()
    p.P = n + 1
    flat = np.zeros(TWL_MAX_MATRIX * TWL_MAX_MATRIX, dtype=np.float32)
    flat[: n * n] = m.reshape(-1)
    p.matrix[:] = flat.tolist()
    p.gap_open = gap_open
    p.gap_extend = gap_extend
    p.gap_boundary = gap_extend if gap_boundary is None else gap_boundary
    p.gap_char = gap_extend if gap_char is None else gap_char
    p.xdrop = int(1000 * -gap_extend) if xdrop is None else int(xdrop)
    p.flen = flen
    p.marker = marker
    return p


def init(device_ids=None):
    lib = load_library()
    if device_ids is None:
        _check(lib.twl_init(None, 0))
    else:
        arr = (C.c_int * len(device_ids))(*device_ids)
        _check(lib.twl_init(arr, len(device_ids)))


def shutdown():
    load_library().twl_shutdown()


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def align_batch(params: TwlParams, batch):
    """Host-buffer form (twl_align_batch).  `batch` is a synth.LevelBatch-like object."""
    lib = load_library()
    n, sl = batch.n_pairs, batch.seq_len
    freq = np.ascontiguousarray(batch.freq, dtype=np.float32)
    go = np.ascontiguousarray(batch.gap_open, dtype=np.float32)
    ge = np.ascontiguousarray(batch.gap_extend, dtype=np.float32)
    ln = np.ascontiguousarray(batch.len, dtype=np.int32)
    nm = np.ascontiguousarray(batch.num, dtype=np.int32)
    aln = np.zeros((n, 2 * sl), dtype=np.int8)
    aln_len = np.zeros(n, dtype=np.int32)
    err = np.zeros(n, dtype=np.int16)
    _check(lib.twl_align_batch(C.byref(params), C.c_int32(n), C.c_int32(sl), _ptr(freq, C.c_float), _ptr(go, C.c_float),
                               _ptr(ge, C.c_float), _ptr(ln, C.c_int32), _ptr(nm, C.c_int32), _ptr(aln, C.c_int8),
                               _ptr(aln_len, C.c_int32), _ptr(err, C.c_int16)))
    return aln, aln_len, err


def align_batch_device(params: TwlParams, n_pairs, seq_len, d_freq, d_gop, d_gex, d_len, d_num, d_aln, d_aln_len, d_err,
                       device=0, stream=None):
    """Device-pointer form (twl_align_batch_device).  Pointers are integers (e.g. torch.Tensor.data_ptr())."""
    lib = load_library()
    vp = C.c_void_p
    _check(lib.twl_align_batch_device(C.c_int(device), vp(stream or 0), C.byref(params), C.c_int32(n_pairs), C.c_int32(seq_len),
                                      vp(d_freq), vp(d_gop), vp(d_gex), vp(d_len), vp(d_num), vp(d_aln), vp(d_aln_len),
                                      vp(d_err)))


KNOB_MT_PERTURB, KNOB_MT_MAX_PAIRS, KNOB_MT_MIN_MARKER, KNOB_MT_LEAD, KNOB_MT_MARGIN, KNOB_MT_ROUNDS, KNOB_MT_THR_JOBS, KNOB_FAIL_ROW_ALLOCS = 1, 2, 3, 4, 5, 6, 7, 8
KNOB_PROT_MODE, KNOB_ASSUME_ONEHOT_QUERY, KNOB_MT_TAIL_PCT, KNOB_MT_WIDE, KNOB_NO_SPEC, KNOB_SCOUT_XDROP_PCT = 9, 10, 11, 12, 13, 14
KNOB_THR_SMALL = 15
KNOB_FORCE_GLOBAL = 16
KNOB_LEAF_STEP = 17
KNOB_POISON_TB = 18
KNOB_MT_ANCHOR = 19
KNOB_PROT_CORRIDOR = 21
KNOB_MT_LEAD2 = 20
PROT_MODES = {"auto": 0, "dense": 1, "sparse": 2, "presim": 3, "r1": 4, "lean_sparse": 5, "lean_presim": 6}


def set_knob(key: int, value: int):
    """twl_set_knob: development / test knobs of the launch policy (include/twl_align.h)."""
    _check(load_library().twl_set_knob(C.c_int(key), C.c_int(value)))


def plan_describe(params, lens, num_cu=256, qry_onehot=False, wide_streak=0, small_state=0) -> str:
    """twl_plan_describe: the launch plan of a nucleotide call in words (no device needed)."""
    wide_streak = int(wide_streak) + (100000 * (32 + int(small_state)) if small_state else 0)      # (1 / -1: the earlier levels of the pass fitted / outgrew the 512-row throughput window)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    buf = C.create_string_buffer(256)
    _check(load_library().twl_plan_describe(C.byref(params), C.c_int32(lens.shape[0]), _ptr(lens, C.c_int32), C.c_int32(num_cu), C.c_int32(int(qry_onehot)), C.c_int32(wide_streak), buf, C.c_int32(256)))
    return buf.value.decode()


def version() -> str:
    """twl_version(): release and the hash of the kernel sources the library was built from (__graft_entry__.source_hash)."""
    return load_library().twl_version().decode()


def get_stats(device=0) -> TwlStats:
    st = TwlStats()
    _check(load_library().twl_get_stats(C.c_int(device), C.byref(st)))
    return st


def get_pair_cells(n, device=0):
    out = np.zeros(n, dtype=np.uint64)
    _check(load_library().twl_get_pair_cells(C.c_int(device), _ptr(out, C.c_uint64), C.c_int32(n)))
    return out


def column_scores(params: TwlParams, ref: np.ndarray, qry: np.ndarray, ref_num: int, qry_num: int) -> np.ndarray:
    """twl_column_scores: similarScore(i, j) for every query row i and reference column j of one pair, shape [Q][R]."""
    ref = np.ascontiguousarray(ref, dtype=np.float32)
    qry = np.ascontiguousarray(qry, dtype=np.float32)
    R, Q, P = ref.shape[0], qry.shape[0], ref.shape[1]
    sl = max(R, Q)
    freq = np.zeros((2, sl, P), dtype=np.float32)
    freq[0, :R] = ref
    freq[1, :Q] = qry
    ln = np.array([R, Q], dtype=np.int32)
    nm = np.array([ref_num, qry_num], dtype=np.int32)
    out = np.zeros((Q, R), dtype=np.float32)
    _check(load_library().twl_column_scores(C.byref(params), C.c_int32(sl), _ptr(freq, C.c_float), _ptr(ln, C.c_int32), _ptr(nm, C.c_int32), _ptr(out, C.c_float)))
    return out


def dp_column_scores(params: TwlParams, batch, pair: int = 0) -> np.ndarray:
    """twl_dp_column_scores: the column scores the DP kernel itself evaluated while aligning pair `pair` of `batch`, shape [Q][R], NaN where
    the band never went."""
    sl = batch.seq_len
    freq = np.ascontiguousarray(batch.freq[pair], dtype=np.float32)
    gop = np.ascontiguousarray(batch.gap_open[pair], dtype=np.float32)
    gex = np.ascontiguousarray(batch.gap_extend[pair], dtype=np.float32)
    ln = np.ascontiguousarray(batch.len[pair], dtype=np.int32)
    nm = np.ascontiguousarray(batch.num[pair], dtype=np.int32)
    out = np.zeros((int(ln[1]), int(ln[0])), dtype=np.float32)
    _check(load_library().twl_dp_column_scores(C.byref(params), C.c_int32(sl), _ptr(freq, C.c_float), _ptr(gop, C.c_float), _ptr(gex, C.c_float),
                                               _ptr(ln, C.c_int32), _ptr(nm, C.c_int32), _ptr(out, C.c_float)))
    return out
