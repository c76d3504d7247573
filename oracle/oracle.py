"""ctypes front-end for the CPU oracle (oracle/nimble_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libnimble_oracle.so")
_lib = None

_ES = {1: 1, 2: 1, 3: 2, 4: 2, 5: 4, 6: 4, 7: 8, 8: 8, 9: 4, 10: 8, 11: 1}
_NP = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}
DT_BOOL, DT_STRING = 11, 12


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "nimble_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, u8p, u32, u64, i32 = C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.orc_create.restype = vp
        L.orc_create.argtypes = [u8p, u64, i32, C.c_char_p, i32]
        L.orc_destroy.argtypes = [vp]
        for f in ("orc_row_count", "orc_null_count"):
            getattr(L, f).restype = u32
            getattr(L, f).argtypes = [vp]
        for f in ("orc_encoding_type", "orc_data_type", "orc_is_nullable", "orc_reset"):
            getattr(L, f).restype = i32
            getattr(L, f).argtypes = [vp]
        L.orc_last_error.restype = C.c_char_p
        L.orc_last_error.argtypes = [vp]
        L.orc_skip.restype = i32
        L.orc_skip.argtypes = [vp, u32]
        L.orc_materialize.restype = i32
        L.orc_materialize.argtypes = [vp, u32, vp]
        L.orc_view_read.restype = i32
        L.orc_view_read.argtypes = [vp, u32, u32, vp]
        L.orc_materialize_bools_as_bits.restype = i32
        L.orc_materialize_bools_as_bits.argtypes = [vp, u32, vp, i32]
        L.orc_materialize_nullable.restype = u32
        L.orc_materialize_nullable.argtypes = [vp, u32, vp, vp, C.POINTER(i32), vp, u32, u32]
        L.orc_sv_total_bytes.restype = u64
        L.orc_sv_total_bytes.argtypes = [vp, u32]
        L.orc_sv_flatten.argtypes = [vp, u32, vp, vp]
        L.orc_split_stream.restype = C.c_int64
        L.orc_split_stream.argtypes = [u8p, u64, vp, vp, C.c_int64]
        L.orc_find_set_bit.restype = u32
        L.orc_find_set_bit.argtypes = [vp, u32, u32, u32]
        L.orc_sd_create.restype = vp
        L.orc_sd_create.argtypes = [u8p, u64, i32]
        L.orc_sd_destroy.argtypes = [vp]
        L.orc_sd_reset.argtypes = [vp]
        L.orc_sd_last_error.restype = C.c_char_p
        L.orc_sd_last_error.argtypes = [vp]
        L.orc_sd_skip.restype = i32
        L.orc_sd_skip.argtypes = [vp, u32]
        L.orc_sd_next.restype = u32
        L.orc_sd_next.argtypes = [vp, u32, vp, vp, C.POINTER(i32), vp, u32]
        i64 = C.c_int64
        L.orc_cd_decode_nullable.restype = i32
        L.orc_cd_decode_nullable.argtypes = [vp, vp, vp, i64, vp]
        L.orc_cd_next_bools.restype = i32
        L.orc_cd_next_bools.argtypes = [vp, vp, i64, vp, i32, vp]
        L.orc_cd_next_indices.restype = i32
        L.orc_cd_next_indices.argtypes = [vp, vp, i64, vp]
        L.orc_cd_skip.restype = i32
        L.orc_cd_skip.argtypes = [vp, i64]
        L.orc_decode_all.restype = u32
        L.orc_decode_all.argtypes = [u8p, u64, i32, vp, vp]
        L.orc_decode_all_fast.restype = u32
        L.orc_decode_all_fast.argtypes = [u8p, u64, i32, vp, vp]
        L.orc_set_bulk.argtypes = [vp, i32]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Encoding:
    """Mirror of nimble::Encoding (encodings/common/Encoding.h:110-389) over the
    C oracle: rowCount / reset / skip / materialize / materializeNullable /
    materializeBoolsAsBits with the reference's row-cursor semantics."""

    def __init__(self, data, varint_rows: bool = False):
        self._data = np.frombuffer(bytes(data), dtype=np.uint8).copy()  # keep alive + writable ptr
        err = C.create_string_buffer(256)
        self._h = lib().orc_create(_ptr(self._data), self._data.size, int(varint_rows), err, 256)
        if not self._h:
            raise OracleError(err.value.decode())
        self.row_count = lib().orc_row_count(self._h)
        self.encoding_type = lib().orc_encoding_type(self._h)
        self.data_type = lib().orc_data_type(self._h)
        self.is_nullable = bool(lib().orc_is_nullable(self._h))
        self.null_count = lib().orc_null_count(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def _err(self):
        raise OracleError(lib().orc_last_error(self._h).decode())

    def reset(self):
        if lib().orc_reset(self._h):
            self._err()

    def skip(self, n: int):
        if lib().orc_skip(self._h, n):
            self._err()

    def _elem_dtype(self):
        return _NP[_ES[self.data_type]]

    def materialize(self, n: int):
        """Returns physicalType[n] (bool: uint8 0/1; string: list of bytes)."""
        if self.data_type == DT_STRING:
            sv = np.zeros(2 * n, dtype=np.uint64)
            if lib().orc_materialize(self._h, n, _ptr(sv)):
                self._err()
            return self._flatten(sv, n)
        out = np.empty(n, dtype=self._elem_dtype())
        if lib().orc_materialize(self._h, n, _ptr(out)):
            self._err()
        return out

    def read(self, offset: int, length: int):
        """EncodingView::read (encodings/views/EncodingView.h:38-41): rows
        [offset, offset+length) whatever the cursor; numeric / bool only."""
        out = np.empty(length, dtype=self._elem_dtype())
        if lib().orc_view_read(self._h, offset, length, _ptr(out)):
            self._err()
        return out

    def _flatten(self, sv, n):
        total = lib().orc_sv_total_bytes(_ptr(sv), n)
        lengths = np.zeros(n, dtype=np.uint32)
        blob = np.zeros(max(int(total), 1), dtype=np.uint8)
        lib().orc_sv_flatten(_ptr(sv), n, _ptr(lengths), _ptr(blob))
        return lengths, blob[:int(total)]

    def materialize_bools_as_bits(self, n: int, buffer: np.ndarray, begin: int):
        assert buffer.dtype == np.uint64
        if lib().orc_materialize_bools_as_bits(self._h, n, _ptr(buffer), begin):
            self._err()

    def materialize_nullable(self, n: int, buffer: np.ndarray, nulls: np.ndarray,
                             scatter: np.ndarray | None = None, scatter_size: int = 0, offset: int = 0):
        """Encoding::materializeNullable.  `buffer` (physical type; strings are
        not supported here) and `nulls` (uint8 bitmap) are updated in place.
        Returns (non_null_count, nulls_requested)."""
        req = C.c_int(0)
        r = lib().orc_materialize_nullable(
            self._h, n, _ptr(buffer), _ptr(nulls), C.byref(req),
            _ptr(scatter) if scatter is not None else None, scatter_size, offset)
        if r == 0xFFFFFFFF:
            self._err()
        return r, bool(req.value)


def decode_chunk(data, varint_rows: bool = False):
    """Whole-chunk decode as the GPU path defines its output:
    returns dict(values=..., nulls=bitmap-or-None, non_null=count, rows=n).
    numeric/bool: `values` has null slots zero-filled (Encoding::materialize);
    string: values=(lengths uint32[n], blob uint8[...]) with empty strings at nulls."""
    e = Encoding(data, varint_rows)
    n = e.row_count
    vals = e.materialize(n)
    nulls = None
    nn = n
    if e.is_nullable:
        e.reset()
        if e.data_type == DT_STRING:
            # the non-null bitmap is the second child (to the end of the chunk): decode it on its own
            raw = bytes(data)
            pre = 6
            if varint_rows:
                pre = 2
                while raw[pre] & 0x80:
                    pre += 1
                pre += 1
            size = int.from_bytes(raw[pre:pre + 4], "little")
            ne = Encoding(raw[pre + 4 + size:], varint_rows)
            bits = ne.materialize(ne.row_count)
            return dict(values=vals, nulls=np.packbits(bits.astype(np.uint8), bitorder="little"),
                        non_null=int(bits.sum()), rows=n, data_type=e.data_type)
        buf = np.zeros(n, dtype=vals.dtype)
        bm = np.zeros((n + 7) // 8 + 8, dtype=np.uint8)
        nn, req = e.materialize_nullable(n, buf, bm)
        if not req:  # no nulls present: reference leaves the bitmap untouched
            bm[:] = 0
            full = np.ones(n, dtype=np.uint8)
            bm[:(n + 7) // 8] = np.packbits(full, bitorder="little")
        nulls = bm[:(n + 7) // 8].copy()
    return dict(values=vals, nulls=nulls, non_null=nn, rows=n, data_type=e.data_type)


def split_stream(stream: bytes):
    a = np.frombuffer(bytes(stream), dtype=np.uint8).copy()
    n = lib().orc_split_stream(_ptr(a), a.size, None, None, 0)
    if n < 0:
        raise OracleError("bad stream framing" if n == -1 else "compressed chunk")
    offs = np.zeros(max(n, 1), dtype=np.uint64)
    sizes = np.zeros(max(n, 1), dtype=np.uint64)
    lib().orc_split_stream(_ptr(a), a.size, _ptr(offs), _ptr(sizes), n)
    return [(int(offs[i]), int(sizes[i])) for i in range(n)]


class StreamDecoder:
    """Mirror of ChunkedStreamDecoder (velox/ChunkedStreamDecoder.cpp:75-205)."""

    def __init__(self, stream: bytes, varint_rows: bool = False):
        self._data = np.frombuffer(bytes(stream), dtype=np.uint8).copy()
        self._h = lib().orc_sd_create(_ptr(self._data), self._data.size, int(varint_rows))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_sd_destroy(self._h)
            self._h = None

    def _err(self):
        raise OracleError(lib().orc_sd_last_error(self._h).decode())

    def reset(self):
        lib().orc_sd_reset(self._h)

    def skip(self, n):
        if lib().orc_sd_skip(self._h, n):
            self._err()

    def next(self, count, output: np.ndarray, nulls: np.ndarray | None = None,
             scatter: np.ndarray | None = None, scatter_size: int = 0):
        req = C.c_int(0)
        r = lib().orc_sd_next(self._h, count, _ptr(output), _ptr(nulls) if nulls is not None else None,
                              C.byref(req), _ptr(scatter) if scatter is not None else None, scatter_size)
        if r == 0xFFFFFFFF:
            self._err()
        return r, bool(req.value)


class ChunkedDecoder(StreamDecoder):
    """Dense bulk entry points of the selective reader's ChunkedDecoder
    (velox/selective/ChunkedDecoder.h:78-90,206-268).  Bitmaps are uint8 arrays,
    LSB first, 1 = not null."""

    def __init__(self, stream: bytes, varint_rows: bool = False, decode_values_with_nulls: bool = False):
        super().__init__(stream, varint_rows)
        self.decode_values_with_nulls = decode_values_with_nulls
        self.preloaded_values = None

    def decode_nullable(self, nulls, data: np.ndarray, count: int, incoming_nulls=None):
        if lib().orc_cd_decode_nullable(self._h, _ptr(nulls) if nulls is not None else None, _ptr(data), count,
                                        _ptr(incoming_nulls) if incoming_nulls is not None else None):
            self._err()

    def next_bools(self, data: np.ndarray, count: int, incoming_nulls=None):
        assert data.dtype == np.uint64
        if self.decode_values_with_nulls:
            self.preloaded_values = np.zeros(count, dtype=np.uint32)
        if lib().orc_cd_next_bools(self._h, _ptr(data), count, _ptr(incoming_nulls) if incoming_nulls is not None else None,
                                   int(self.decode_values_with_nulls),
                                   _ptr(self.preloaded_values) if self.decode_values_with_nulls else None):
            self._err()

    def next_indices(self, data: np.ndarray, count: int, incoming_nulls=None):
        assert data.dtype in (np.int32, np.uint32)
        if lib().orc_cd_next_indices(self._h, _ptr(data), count, _ptr(incoming_nulls) if incoming_nulls is not None else None):
            self._err()

    def skip_values(self, n: int):
        if lib().orc_cd_skip(self._h, n):
            self._err()
