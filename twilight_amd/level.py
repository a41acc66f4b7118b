"""ctypes binding of the device-resident level API (C ABI: include/twl_level.h).  No fallback path."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import api


class TwlSide(C.Structure):
    """twl_side -- what calculateProfile reads from a Node (reference alignment-helper.cpp:8-40)."""
    _fields_ = [("n_members", C.c_int32), ("member_off", C.c_int32), ("len", C.c_int32), ("num", C.c_int32), ("weight", C.c_float),
                ("cache_id", C.c_int32), ("store_id", C.c_int32), ("reserved", C.c_int32)]


_SYMBOLS = ["twl_store_create", "twl_store_destroy", "twl_store_read_rows", "twl_store_read_rows_of", "twl_store_write_rows", "twl_store_write_cache", "twl_store_rows_to_block", "twl_store_rows_from_block", "twl_store_exchange_buffers", "twl_store_read_cache", "twl_store_drop_cache",
            "twl_level_prepare", "twl_level_read_colinfo", "twl_level_read_colinfo_many", "twl_level_align", "twl_level_align_mixed", "twl_level_read_path", "twl_level_read_paths", "twl_level_commit", "twl_level_commit_from_dp", "twl_level_read_columns", "twl_level_timing",
            "twl_level_restore", "twl_level_read_final", "twl_level_exchange_buffers", "twl_level_paths_to_block", "twl_level_paths_from_block",
            "twl_level_write_final"]


def exported_symbols():
    return list(_SYMBOLS)


def _lib():
    lib = api.load_library()
    for name in _SYMBOLS:
        if name != "twl_store_destroy":
            getattr(lib, name).restype = C.c_int
    lib.twl_store_destroy.restype = None
    return lib


@dataclass
class Side:
    members: Sequence[int]           # sequence ids
    member_weight: Sequence[float]   # seq.weight / groupWeight * num, fp32
    len: int
    num: int
    weight: float
    cache_id: int = -1
    store_id: int = -1


class Store:
    """Aligned rows of every sequence, resident on one device."""

    def __init__(self, seqs: List[bytes], seq_type: str = "n", device: int = 0):
        lib = _lib()
        self.P = 6 if seq_type == "n" else 22
        self.n = len(seqs)
        arr = (C.c_char_p * self.n)(*seqs)
        lens = (C.c_int32 * self.n)(*[len(s) for s in seqs])
        self._h = C.c_void_p()
        api._check(lib.twl_store_create(C.c_int(device), C.c_char(seq_type.encode()), C.c_int32(self.n), arr, lens, C.byref(self._h)))
        self._n_pairs = 0
        self._seq_len = 0

    def close(self):
        if self._h:
            _lib().twl_store_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rows(self) -> List[bytes]:
        lib = _lib()
        lens = (C.c_int32 * self.n)()
        api._check(lib.twl_store_read_rows(self._h, None, lens))
        bufs = [C.create_string_buffer(max(int(n), 1)) for n in lens]
        ptrs = (C.c_void_p * self.n)(*[C.cast(b, C.c_void_p) for b in bufs])
        api._check(lib.twl_store_read_rows(self._h, ptrs, lens))
        return [bufs[i].raw[: lens[i]] for i in range(self.n)]

    def rows_of(self, ids: Sequence[int]) -> List[bytes]:
        """twl_store_read_rows_of: the current rows of a list of sequences."""
        lib = _lib()
        n = len(ids)
        idv = (C.c_int32 * n)(*ids)
        lens = (C.c_int32 * n)()
        api._check(lib.twl_store_read_rows_of(self._h, C.c_int32(n), idv, None, lens))
        buf = C.create_string_buffer(max(1, sum(lens)))
        api._check(lib.twl_store_read_rows_of(self._h, C.c_int32(n), idv, buf, lens))
        out, at = [], 0
        for k in range(n):
            out.append(buf.raw[at: at + lens[k]])
            at += lens[k]
        return out

    def write_rows(self, ids: Sequence[int], rows: Sequence[bytes], via_device_block: bool = False):
        """twl_store_write_rows (host block) or twl_store_rows_from_block (device block of the store's exchange buffers): rows of other ranks' subtrees arriving."""
        lib = _lib()
        n = len(ids)
        idv = (C.c_int32 * n)(*ids)
        lens = (C.c_int32 * n)(*[len(r) for r in rows])
        blob = b"".join(rows)
        if not via_device_block:
            api._check(lib.twl_store_write_rows(self._h, C.c_int32(n), idv, blob, lens))
            return
        send, recv = C.c_void_p(), C.c_void_p()
        api._check(lib.twl_store_exchange_buffers(self._h, C.c_int64(len(blob)), C.c_int64(len(blob)), C.byref(send), C.byref(recv)))
        api._check(lib.twl_copy_to_device(C.c_int(0), recv, blob, C.c_uint64(len(blob))))
        api._check(lib.twl_store_rows_from_block(self._h, C.c_int32(n), idv, lens, recv))

    def rows_to_block(self, ids: Sequence[int]) -> List[bytes]:
        """twl_store_rows_to_block into the store's send buffer, read back: what a rank sends of its subtrees."""
        lib = _lib()
        n = len(ids)
        idv = (C.c_int32 * n)(*ids)
        lens = (C.c_int32 * n)()
        api._check(lib.twl_store_read_rows_of(self._h, C.c_int32(n), idv, None, lens))
        total = max(1, sum(lens))
        send, recv = C.c_void_p(), C.c_void_p()
        api._check(lib.twl_store_exchange_buffers(self._h, C.c_int64(total), C.c_int64(total), C.byref(send), C.byref(recv)))
        api._check(lib.twl_store_rows_to_block(self._h, C.c_int32(n), idv, send, lens))
        buf = C.create_string_buffer(total)
        api._check(lib.twl_copy_from_device(C.c_int(0), buf, send, C.c_uint64(total)))
        out, at = [], 0
        for k in range(n):
            out.append(buf.raw[at: at + lens[k]])
            at += lens[k]
        return out

    def write_cache(self, cache_id: int, profile: np.ndarray):
        """twl_store_write_cache: a cached profile float[len][P] under an id new to this store."""
        prof = np.ascontiguousarray(profile, dtype=np.float32)
        api._check(_lib().twl_store_write_cache(self._h, C.c_int32(cache_id), prof.ctypes.data_as(C.POINTER(C.c_float)), C.c_int32(prof.shape[0])))

    def cache(self, cache_id: int) -> np.ndarray:
        lib = _lib()
        n = C.c_int32(0)
        api._check(lib.twl_store_read_cache(self._h, C.c_int32(cache_id), None, C.byref(n)))
        out = np.zeros((n.value, self.P), dtype=np.float32)
        api._check(lib.twl_store_read_cache(self._h, C.c_int32(cache_id), out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)))
        return out

    def prepare(self, params: api.TwlParams, pairs: Sequence[Sequence[Side]], gappy_threshold: float = 0.95, seq_len: Optional[int] = None):
        """twl_level_prepare.  Returns (len_out[n][2], colinfo[n][2][seq_len])."""
        lib = _lib()
        n = len(pairs)
        sides = (TwlSide * (2 * n))()
        members, weights = [], []
        for i, pr in enumerate(pairs):
            for sd in range(2):
                s = pr[sd]
                x = sides[2 * i + sd]
                x.n_members, x.member_off, x.len, x.num, x.weight = len(s.members), len(members), s.len, s.num, s.weight
                x.cache_id, x.store_id, x.reserved = s.cache_id, s.store_id, 0
                members.extend(int(m) for m in s.members)
                weights.extend(s.member_weight)
        sl = seq_len if seq_len is not None else max([1] + [s.len for pr in pairs for s in pr])
        mem = np.asarray(members, dtype=np.int32)
        w = np.asarray(weights, dtype=np.float32)
        lens = np.zeros((n, 2), dtype=np.int32)
        info = np.zeros((n, 2, sl), dtype=np.uint8)
        api._check(lib.twl_level_prepare(self._h, C.byref(params), C.c_float(gappy_threshold), C.c_int32(n), sides,
                                         mem.ctypes.data_as(C.POINTER(C.c_int32)), w.ctypes.data_as(C.POINTER(C.c_float)), C.c_int32(sl),
                                         lens.ctypes.data_as(C.POINTER(C.c_int32)), info.ctypes.data_as(C.POINTER(C.c_uint8))))
        self._n_pairs, self._seq_len, self._lens = n, sl, lens
        return lens, info

    def columns(self, pair: int, side: int) -> np.ndarray:
        """Packed DP columns [len][P+2] of one side of the prepared level."""
        n = int(self._lens[pair, side])
        out = np.zeros((n, self.P + 2), dtype=np.float32)
        api._check(_lib().twl_level_read_columns(self._h, C.c_int32(pair), C.c_int32(side), out.ctypes.data_as(C.POINTER(C.c_float)), C.c_int32(n)))
        return out

    def align(self, params: api.TwlParams, run_mask: Optional[np.ndarray] = None, zero_gap: Optional[np.ndarray] = None):
        """twl_level_align; with zero_gap (one flag per pair) twl_level_align_mixed: those pairs take gapCharScore 0, the others params.gap_char."""
        n, sl = self._n_pairs, self._seq_len
        aln = np.zeros((n, 2 * sl), dtype=np.int8)
        aln_len = np.zeros(n, dtype=np.int32)
        err = np.zeros(n, dtype=np.int16)
        mask = None
        if run_mask is not None:
            m = np.ascontiguousarray(run_mask, dtype=np.uint8)
            mask = m.ctypes.data_as(C.POINTER(C.c_uint8))
        if zero_gap is not None:
            z = np.ascontiguousarray(zero_gap, dtype=np.uint8)
            assert z.shape == (n,)
            api._check(_lib().twl_level_align_mixed(self._h, C.byref(params), mask, z.ctypes.data_as(C.POINTER(C.c_uint8)), aln.ctypes.data_as(C.POINTER(C.c_int8)),
                                                    aln_len.ctypes.data_as(C.POINTER(C.c_int32)), err.ctypes.data_as(C.POINTER(C.c_int16))))
            return aln, aln_len, err
        api._check(_lib().twl_level_align(self._h, C.byref(params), mask, aln.ctypes.data_as(C.POINTER(C.c_int8)),
                                          aln_len.ctypes.data_as(C.POINTER(C.c_int32)), err.ctypes.data_as(C.POINTER(C.c_int16))))
        return aln, aln_len, err

    def align_in_hbm(self, params: api.TwlParams, run_mask: Optional[np.ndarray] = None):
        """twl_level_align with aln_out = NULL: lengths and error codes only, the paths stay on the device."""
        n = self._n_pairs
        aln_len = np.zeros(n, dtype=np.int32)
        err = np.zeros(n, dtype=np.int16)
        mask = None
        if run_mask is not None:
            m = np.ascontiguousarray(run_mask, dtype=np.uint8)
            mask = m.ctypes.data_as(C.POINTER(C.c_uint8))
        api._check(_lib().twl_level_align(self._h, C.byref(params), mask, None, aln_len.ctypes.data_as(C.POINTER(C.c_int32)), err.ctypes.data_as(C.POINTER(C.c_int16))))
        return aln_len, err

    def read_path(self, pair: int, length: int) -> np.ndarray:
        out = np.zeros(max(1, length), dtype=np.int8)
        api._check(_lib().twl_level_read_path(self._h, C.c_int32(pair), out.ctypes.data_as(C.POINTER(C.c_int8)), C.c_int32(length)))
        return out[:length]

    def restore(self, params: api.TwlParams, pairs: Sequence[int], out_stride: int) -> np.ndarray:
        """twl_level_restore: gappy columns back into the DP paths of `pairs`, on the device; returns their final lengths (-1: host must do it)."""
        sel = np.asarray(list(pairs), dtype=np.int32)
        out = np.zeros(max(1, len(sel)), dtype=np.int32)
        api._check(_lib().twl_level_restore(self._h, C.byref(params), C.c_int32(len(sel)), sel.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(out_stride),
                                            out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out[: len(sel)]

    def read_final(self, pair: int, length: int) -> np.ndarray:
        out = np.zeros(max(1, length), dtype=np.int8)
        api._check(_lib().twl_level_read_final(self._h, C.c_int32(pair), out.ctypes.data_as(C.POINTER(C.c_int8)), C.c_int32(length)))
        return out[:length]

    def commit_from_dp(self, paths: Sequence[Optional[np.ndarray]], path_len: Sequence[int], stride: Optional[int] = None, restored: Sequence[int] = ()):
        """twl_level_commit_from_dp: paths[i] is None -> pair i's path is in HBM already (the DP output, or what twl_level_restore made of it), path_len[i] long."""
        n = self._n_pairs
        stride = int(stride) if stride else max([1] + [int(x) for x in path_len])
        flat = np.zeros((n, stride), dtype=np.int8)
        plen = np.asarray(path_len, dtype=np.int32).copy()
        from_dp = np.zeros(n, dtype=np.uint8)
        for i, p in enumerate(paths):
            if p is None: from_dp[i] = (2 if i in set(restored) else 1) if plen[i] > 0 else 0
            else: flat[i, : len(p)] = p
        api._check(_lib().twl_level_commit_from_dp(self._h, flat.ctypes.data_as(C.POINTER(C.c_int8)), plen.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(stride),
                                                   from_dp.ctypes.data_as(C.POINTER(C.c_uint8))))

    def commit(self, paths: Sequence[np.ndarray]):
        n = self._n_pairs
        assert len(paths) == n
        stride = max([1] + [len(p) for p in paths])
        flat = np.zeros((n, stride), dtype=np.int8)
        plen = np.zeros(n, dtype=np.int32)
        for i, p in enumerate(paths):
            flat[i, : len(p)] = p
            plen[i] = len(p)
        api._check(_lib().twl_level_commit(self._h, flat.ctypes.data_as(C.POINTER(C.c_int8)), plen.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(stride)))
