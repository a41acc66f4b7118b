"""ctypes binding of libtwl_host.so (C ABI: include/twl_msa.h): TWILIGHT's tree + sequences alignment flow with the MI355X
level kernel, in steps (open -> [shard] -> upload -> align -> report / write).  No fallback path: the library must be built."""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, List, Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TWL_HOST_LIB") or os.path.join(_HERE, "libtwl_host.so")   # TWL_HOST_LIB: the CPU-check build in tests


class MsaLevel(C.Structure):
    """twl_msa_level -- one level-kernel call (the reference's per-level report line, progressive.cpp:178-189)."""
    _fields_ = [("pairs", C.c_int32), ("task", C.c_int32), ("band_cells", C.c_uint64), ("relaunched", C.c_uint64),
                ("kernel_ms", C.c_double), ("level_ms", C.c_double), ("exchange_ms", C.c_double), ("matrix_mode", C.c_int32), ("speculative", C.c_int32),
                ("mt_tiles_predicted", C.c_int32), ("mt_tiles_inline", C.c_int32), ("kernel", C.c_char * 160)]


class MsaTotals(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("aln_len", C.c_int32), ("n_sequences", C.c_int32), ("reserved", C.c_int32),
                ("pairs", C.c_uint64), ("band_cells", C.c_uint64), ("relaunched", C.c_uint64),
                ("kernel_ms", C.c_double), ("exchange_ms", C.c_double), ("align_s", C.c_double), ("nominal_cells", C.c_uint64)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

_SYMBOLS = ["twl_msa_open", "twl_msa_shard", "twl_msa_shard_device", "twl_msa_rccl_unique_id", "twl_msa_shard_rccl", "twl_msa_upload", "twl_msa_align", "twl_msa_report",
            "twl_msa_write", "twl_msa_close", "twl_msa_last_error"]
_libs = {}


def exported_symbols():
    return list(_SYMBOLS)


class MsaError(RuntimeError):
    pass


def load_library(path: Optional[str] = None):
    path = path or LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise MsaError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        try:                      # bind to the HIP runtime torch already loaded (see api.load_library)
            import torch  # noqa: F401
        except Exception:
            pass
        lib = C.CDLL(path)
        lib.twl_msa_last_error.restype = C.c_char_p
        for name in _SYMBOLS[:-2]:
            getattr(lib, name).restype = C.c_int
        lib.twl_msa_close.restype = None
        lib.twl_msa_close.argtypes = [C.c_void_p]
        _libs[path] = lib
    return _libs[path]


def rccl_unique_id(lib_path: Optional[str] = None) -> bytes:
    """ncclGetUniqueId through the host library (call on one rank after twl_init, i.e. after a handle was opened and uploaded or api.init ran)."""
    lib = load_library(lib_path)
    buf = (C.c_char * 128)()
    if lib.twl_msa_rccl_unique_id(buf) != 0:
        raise MsaError(f"twl_msa_rccl_unique_id failed: {lib.twl_msa_last_error().decode()}")
    return bytes(buf.raw)


class Msa:
    """One alignment run.  `argv` are the CLI flags (-t tree -i sequences -o output [--type n|p] [--gpu-index k] ...)."""

    def __init__(self, argv: Sequence[str], lib_path: Optional[str] = None):
        self._lib = load_library(lib_path)
        args = [b"twl_msa"] + [a.encode() for a in argv]
        arr = (C.c_char_p * len(args))(*args)
        self._h = C.c_void_p()
        self._cb = None
        self._check(self._lib.twl_msa_open(len(args), arr, C.byref(self._h)), "twl_msa_open")

    def _check(self, rc, what):
        if rc != 0:
            raise MsaError(f"{what} failed ({rc}): {self._lib.twl_msa_last_error().decode()}")

    def shard(self, rank: int, world: int, exchange: Optional[Callable[[int, int, int], int]], exchange_device: Optional[Callable[[int, int, int], int]] = None):
        """Align this family together with `world` - 1 other processes.  `exchange(send_ptr, bytes_per_rank, recv_ptr) -> 0` is an
        all-gather of host blocks (twilight_amd.dist.make_exchange); `exchange_device` the same on device pointers
        (twilight_amd.dist.make_device_exchange): with it the device-resident level kernel keeps the paths in HBM end to end."""
        def wrap(fn):
            def _cb(user, send, nbytes, recv):
                # ctypes swallows exceptions raised inside a callback and hands the C side an undefined return value: every failure
                # must come back as a non-zero code, which the host library turns into a fatal error of the run
                try:
                    return int(fn(send, nbytes, recv))
                except BaseException as ex:  # noqa: BLE001
                    import sys
                    import traceback
                    traceback.print_exc()
                    print(f"twilight_amd: the exchange callback failed: {ex!r}", file=sys.stderr, flush=True)
                    return 1
            return EXCHANGE_FN(_cb)

        if exchange_device is not None:
            self._cb_dev = wrap(exchange_device)
            self._cb = wrap(exchange) if exchange is not None else C.cast(None, EXCHANGE_FN)
            self._check(self._lib.twl_msa_shard_device(self._h, rank, world, self._cb_dev, None, self._cb, None), "twl_msa_shard_device")
        elif world > 1 or exchange is not None:
            self._cb = wrap(exchange)
            self._check(self._lib.twl_msa_shard(self._h, rank, world, self._cb, None), "twl_msa_shard")
        return self

    def shard_rccl(self, rank: int, world: int, unique_id: bytes):
        """The same, with the per-level collective made by the library itself (RCCL from its own C++: include/twl_msa.h, twl_msa_shard_rccl):
        `unique_id` = the 128 bytes `rccl_unique_id()` returned on ONE rank, handed to all of them by the launcher."""
        if len(unique_id) != 128:
            raise MsaError("the communicator id has 128 bytes")
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        self._check(self._lib.twl_msa_shard_rccl(self._h, rank, world, buf), "twl_msa_shard_rccl")
        return self

    def upload(self):
        self._check(self._lib.twl_msa_upload(self._h), "twl_msa_upload")
        return self

    def align(self):
        self._check(self._lib.twl_msa_align(self._h), "twl_msa_align")
        return self

    def report(self) -> Tuple[MsaTotals, List[MsaLevel]]:
        t = MsaTotals()
        self._check(self._lib.twl_msa_report(self._h, C.byref(t), None, 0), "twl_msa_report")
        lv = (MsaLevel * max(1, t.n_levels))()
        self._check(self._lib.twl_msa_report(self._h, C.byref(t), lv, t.n_levels), "twl_msa_report")
        return t, [lv[i] for i in range(t.n_levels)]

    def write(self, path: Optional[str] = None):
        self._check(self._lib.twl_msa_write(self._h, path.encode() if path else None), "twl_msa_write")
        return self

    def close(self):
        if self._h:
            self._lib.twl_msa_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
