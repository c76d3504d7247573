"""ctypes access to the C++ host layer (nimble_amd/host): GpuEncodingFactory,
its Encoding objects and GpuStripeDecoder, with the same Python surface as
oracle.oracle.Encoding / StreamDecoder so one contract test drives both."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from nimble_amd import build as nb
from nimble_amd import wire as W

_lib = None


class HostError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (loads the HIP runtime the libraries share)
        from nimble_amd import capi

        capi.lib()
        if not os.path.exists(nb.HOST_LIB_PATH):
            raise ImportError(f"{nb.HOST_LIB_PATH} missing: run __graft_entry__.build()")
        L = C.CDLL(nb.HOST_LIB_PATH)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.nblh_last_error.restype = C.c_char_p
        L.nblh_factory_create.restype = vp
        L.nblh_factory_create.argtypes = [i32, i32]
        L.nblh_factory_free.argtypes = [vp]
        L.nblh_factory_stats.argtypes = [vp, vp]
        L.nblh_encoding_create.restype = vp
        L.nblh_encoding_create.argtypes = [vp, vp, u64]
        L.nblh_encoding_free.argtypes = [vp]
        for f in ("nblh_row_count", "nblh_null_count"):
            getattr(L, f).restype = u32
            getattr(L, f).argtypes = [vp]
        for f in ("nblh_is_nullable", "nblh_data_type", "nblh_encoding_type", "nblh_reset"):
            getattr(L, f).restype = i32
            getattr(L, f).argtypes = [vp]
        L.nblh_debug_string.argtypes = [vp, i32, C.c_char_p, u32]
        L.nblh_skip.argtypes = [vp, u32]
        L.nblh_materialize.argtypes = [vp, u32, vp]
        L.nblh_materialize_bools_as_bits.argtypes = [vp, u32, vp, i32]
        L.nblh_materialize_nullable.argtypes = [vp, u32, vp, vp, C.POINTER(i32), vp, u32, u32, C.POINTER(u32)]
        L.nblh_sv_total.restype = u64
        L.nblh_sv_total.argtypes = [vp, u32]
        L.nblh_sv_flatten.argtypes = [vp, u32, vp, vp]
        L.nblh_dictionary_enabled.restype = i32
        L.nblh_dictionary_enabled.argtypes = [vp]
        L.nblh_dictionary_size.argtypes = [vp, C.POINTER(u32)]
        L.nblh_dictionary_entries.argtypes = [vp, C.POINTER(vp)]
        L.nblh_materialize_indices.argtypes = [vp, u32, vp]
        L.nblh_supports_encoding_view.restype = i32
        L.nblh_supports_encoding_view.argtypes = [i32]
        L.nblh_view_create.restype = vp
        L.nblh_view_create.argtypes = [vp, vp, u64]
        L.nblh_view_free.argtypes = [vp]
        L.nblh_view_row_count.restype = u32
        L.nblh_view_row_count.argtypes = [vp]
        for f in ("nblh_view_encoding_type", "nblh_view_data_type"):
            getattr(L, f).restype = i32
            getattr(L, f).argtypes = [vp]
        L.nblh_view_read.argtypes = [vp, u32, u32, vp]
        L.nblh_view_read_at.argtypes = [vp, u32, vp]
        L.nblh_stripe_create.restype = vp
        L.nblh_stripe_create.argtypes = [vp, C.POINTER(C.c_char_p), vp, u32]
        L.nblh_stripe_free.argtypes = [vp]
        L.nblh_stripe_next.argtypes = [vp, u32, u32, vp, vp, C.POINTER(i32), vp, u32, C.POINTER(u32), C.POINTER(u32)]
        L.nblh_stripe_skip.argtypes = [vp, u32, u32]
        L.nblh_stripe_reset.argtypes = [vp, u32]
        i64 = C.c_int64
        L.nblh_chunked_open.argtypes = [vp, u32, i32]
        L.nblh_chunked_skip.argtypes = [vp, u32, i64]
        L.nblh_chunked_next_bools.argtypes = [vp, u32, vp, i64, vp]
        L.nblh_chunked_next_indices.argtypes = [vp, u32, vp, i64, vp]
        L.nblh_chunked_decode_nullable.argtypes = [vp, u32, vp, vp, i64, vp]
        L.nblh_chunked_preloaded_values.argtypes = [vp, u32, C.POINTER(vp)]
        L.nblh_chunked_state.argtypes = [vp, u32, i32, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]
        L.nblh_decode_to_device.restype = vp
        L.nblh_decode_to_device.argtypes = [vp, C.POINTER(C.c_char_p), vp, u32]
        L.nblh_device_batch_free.argtypes = [vp]
        L.nblh_device_batch_data.restype = u64
        L.nblh_device_batch_data.argtypes = [vp]
        L.nblh_device_batch_bytes.restype = u64
        L.nblh_device_batch_bytes.argtypes = [vp]
        L.nblh_device_batch_layout.argtypes = [vp, u32, vp]
        L.nblh_device_batch_copy_out.argtypes = [vp, vp]
        u64p = C.POINTER(u64)
        L.nblh_tablet_open.restype = vp
        L.nblh_tablet_open.argtypes = [vp, u64]
        L.nblh_tablet_free.argtypes = [vp]
        L.nblh_tablet_stripe_count.restype = u32
        L.nblh_tablet_stripe_count.argtypes = [vp]
        L.nblh_tablet_row_count.restype = u64
        L.nblh_tablet_row_count.argtypes = [vp]
        L.nblh_tablet_stripe_info.argtypes = [vp, u32, C.POINTER(u32), u64p, C.POINTER(u32), C.POINTER(u32)]
        L.nblh_tablet_stream.argtypes = [vp, u32, u32, C.POINTER(vp), u64p]
        L.nblh_tablet_optional_section.argtypes = [vp, C.c_char_p, C.POINTER(vp), u64p]
        L.nblh_tablet_decode_stripe.restype = vp
        L.nblh_tablet_decode_stripe.argtypes = [vp, vp, u32, vp, u32, C.POINTER(u32)]
        L.nblh_assign_stripes.argtypes = [vp, u32, u32, vp]
        L.nblh_tablet_decoder_create.restype = vp
        L.nblh_tablet_decoder_create.argtypes = [vp, vp, u32, i32, u32]
        L.nblh_tablet_decoder_free.argtypes = [vp]
        L.nblh_tablet_decoder_device_of.argtypes = [vp, u32, C.POINTER(i32)]
        L.nblh_tablet_decoder_take.restype = vp
        L.nblh_tablet_decoder_take.argtypes = [vp, u32, vp, u32, C.POINTER(u32)]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise HostError(lib().nblh_last_error().decode(errors="replace"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Factory:
    def __init__(self, varint_rows: bool = False, device: int = 0, keep_dictionary: bool = False, collect_stats: bool = False):
        self._h = lib().nblh_factory_create(int(varint_rows) | (2 if keep_dictionary else 0) | (4 if collect_stats else 0), device)
        if not self._h:
            _chk(-1)

    def stats(self) -> dict:
        """Options::decodingStats of a factory made with collect_stats: per counter {count, sum, min, max}."""
        raw = np.zeros(24, dtype=np.uint64)
        lib().nblh_factory_stats(self._h, _p(raw))
        names = ("decodeCPUTimeNanos", "decompressCPUTimeNanos", "encodedBytes", "decodedBytes", "planNanos", "deviceNanos")
        return {n: dict(count=int(raw[4 * i]), sum=int(raw[4 * i + 1]), min=int(raw[4 * i + 2]), max=int(raw[4 * i + 3])) for i, n in enumerate(names)}

    def create(self, data) -> "HostEncoding":
        return HostEncoding(self, data)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().nblh_factory_free(self._h)
            self._h = None


class HostEncoding:
    def __init__(self, factory: Factory, data):
        self._data = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        self._factory = factory
        self._h = lib().nblh_encoding_create(factory._h, _p(self._data), self._data.size)
        if not self._h:
            _chk(-1)
        self.row_count = lib().nblh_row_count(self._h)
        self.data_type = lib().nblh_data_type(self._h)
        self.encoding_type = lib().nblh_encoding_type(self._h)
        self.is_nullable = bool(lib().nblh_is_nullable(self._h))
        self.null_count = lib().nblh_null_count(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().nblh_encoding_free(self._h)
            self._h = None

    def debug_string(self, offset=0):
        buf = C.create_string_buffer(256)
        lib().nblh_debug_string(self._h, offset, buf, 256)
        return buf.value.decode()

    def reset(self):
        _chk(lib().nblh_reset(self._h))

    def skip(self, n):
        _chk(lib().nblh_skip(self._h, n))

    def materialize(self, n):
        if self.data_type == W.STRING:
            sv = np.zeros(2 * n, dtype=np.uint64)
            _chk(lib().nblh_materialize(self._h, n, _p(sv)))
            total = lib().nblh_sv_total(_p(sv), n)
            lengths = np.zeros(n, dtype=np.uint32)
            blob = np.zeros(max(int(total), 1), dtype=np.uint8)
            lib().nblh_sv_flatten(_p(sv), n, _p(lengths), _p(blob))
            return lengths, blob[: int(total)]
        dt = np.uint8 if self.data_type == W.BOOL else W.physical_dtype(self.data_type)
        out = np.empty(n, dtype=dt)
        _chk(lib().nblh_materialize(self._h, n, _p(out)))
        return out

    @property
    def dictionary_enabled(self):
        return bool(lib().nblh_dictionary_enabled(self._h))

    def dictionary_size(self):
        n = C.c_uint32(0)
        _chk(lib().nblh_dictionary_size(self._h, C.byref(n)))
        return n.value

    def dictionary_entries(self):
        """Alphabet as an array of the physical type, or (lengths, blob) for strings."""
        n = self.dictionary_size()
        p = C.c_void_p()
        _chk(lib().nblh_dictionary_entries(self._h, C.byref(p)))
        if self.data_type == W.STRING:
            total = lib().nblh_sv_total(p, n)
            lengths = np.zeros(n, dtype=np.uint32)
            blob = np.zeros(max(int(total), 1), dtype=np.uint8)
            lib().nblh_sv_flatten(p, n, _p(lengths), _p(blob))
            return lengths, blob[: int(total)]
        dt = W.physical_dtype(self.data_type)
        if n == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,)).view(dt).copy()

    def materialize_indices(self, n):
        out = np.empty(n, dtype=np.uint32)
        _chk(lib().nblh_materialize_indices(self._h, n, _p(out)))
        return out

    def materialize_bools_as_bits(self, n, buffer, begin):
        _chk(lib().nblh_materialize_bools_as_bits(self._h, n, _p(buffer), begin))

    def materialize_nullable(self, n, buffer, nulls, scatter=None, scatter_size=0, offset=0):
        req = C.c_int(0)
        nn = C.c_uint32(0)
        _chk(lib().nblh_materialize_nullable(self._h, n, _p(buffer), _p(nulls), C.byref(req), _p(scatter), scatter_size,
                                             offset, C.byref(nn)))
        return nn.value, bool(req.value)


class HostEncodingView:
    """GpuEncodingFactory::createEncodingView: cursor-free reads of one stream."""

    def __init__(self, factory: Factory, data):
        self._data = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        self._factory = factory
        self._h = lib().nblh_view_create(factory._h, _p(self._data), self._data.size)
        if not self._h:
            _chk(-1)
        self.row_count = lib().nblh_view_row_count(self._h)
        self.data_type = lib().nblh_view_data_type(self._h)
        self.encoding_type = lib().nblh_view_encoding_type(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().nblh_view_free(self._h)
            self._h = None

    def _out(self, n):
        if self.data_type == W.STRING:
            return np.zeros(2 * n, dtype=np.uint64)
        return np.empty(n, dtype=np.uint8 if self.data_type == W.BOOL else W.physical_dtype(self.data_type))

    def _strings(self, sv, n):
        total = lib().nblh_sv_total(_p(sv), n)
        lengths = np.zeros(n, dtype=np.uint32)
        blob = np.zeros(max(int(total), 1), dtype=np.uint8)
        lib().nblh_sv_flatten(_p(sv), n, _p(lengths), _p(blob))
        return lengths, blob[: int(total)]

    def read(self, offset, length):
        out = self._out(length)
        _chk(lib().nblh_view_read(self._h, offset, length, _p(out)))
        return self._strings(out, length) if self.data_type == W.STRING else out

    def read_at(self, index):
        out = self._out(1)
        _chk(lib().nblh_view_read_at(self._h, index, _p(out)))
        return self._strings(out, 1) if self.data_type == W.STRING else out[0]


def supports_encoding_view(encoding_type: int) -> bool:
    return bool(lib().nblh_supports_encoding_view(int(encoding_type)))


class StripeDecoder:
    """GpuStripeDecoder: all streams of a stripe decoded in one GPU batch."""

    def __init__(self, factory: Factory, streams):
        self._factory = factory
        self._streams = [bytes(s) for s in streams]
        arr = (C.c_char_p * len(self._streams))(*self._streams)
        sizes = np.array([len(s) for s in self._streams], dtype=np.uint64)
        self._h = lib().nblh_stripe_create(factory._h, arr, _p(sizes), len(self._streams))
        if not self._h:
            _chk(-1)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().nblh_stripe_free(self._h)
            self._h = None

    def stream(self, i):
        return _StreamView(self, i)

    def chunked(self, i, decode_values_with_nulls=False):
        """The stream through the selective reader's ChunkedDecoder interface."""
        _chk(lib().nblh_chunked_open(self._h, i, int(decode_values_with_nulls)))
        return _ChunkedView(self, i, decode_values_with_nulls)


class _StreamView:
    def __init__(self, stripe, idx):
        self._s, self._i = stripe, idx

    def reset(self):
        _chk(lib().nblh_stripe_reset(self._s._h, self._i))

    def skip(self, n):
        _chk(lib().nblh_stripe_skip(self._s._h, self._i, n))

    def next(self, count, output, nulls=None, scatter=None, scatter_size=0):
        req = C.c_int(0)
        nn = C.c_uint32(0)
        nb_ = C.c_uint32(0)
        _chk(lib().nblh_stripe_next(self._s._h, self._i, count, _p(output), _p(nulls), C.byref(req), _p(scatter),
                                    scatter_size, C.byref(nn), C.byref(nb_)))
        self.last_string_buffers = nb_.value
        return nn.value, bool(req.value)


class _ChunkedView:
    """Same Python surface as oracle.oracle.ChunkedDecoder."""

    def __init__(self, stripe, idx, decode_values_with_nulls):
        self._s, self._i = stripe, idx
        self.decode_values_with_nulls = decode_values_with_nulls
        self.preloaded_values = None

    def decode_nullable(self, nulls, data, count, incoming_nulls=None):
        _chk(lib().nblh_chunked_decode_nullable(self._s._h, self._i, _p(nulls), _p(data), count, _p(incoming_nulls)))

    def next_bools(self, data, count, incoming_nulls=None):
        _chk(lib().nblh_chunked_next_bools(self._s._h, self._i, _p(data), count, _p(incoming_nulls)))
        if self.decode_values_with_nulls:
            p = C.c_void_p()
            _chk(lib().nblh_chunked_preloaded_values(self._s._h, self._i, C.byref(p)))
            self.preloaded_values = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(max(count, 1),))[:count].copy()

    def next_indices(self, data, count, incoming_nulls=None):
        _chk(lib().nblh_chunked_next_indices(self._s._h, self._i, _p(data), count, _p(incoming_nulls)))

    def skip_values(self, n):
        _chk(lib().nblh_chunked_skip(self._s._h, self._i, n))

    def state(self, ensure_loaded=False):
        """(remaining values, encoding type or -1, dictionaryConvertible)"""
        rem, et, dc = C.c_int64(), C.c_int(), C.c_int()
        _chk(lib().nblh_chunked_state(self._s._h, self._i, int(ensure_loaded), C.byref(rem), C.byref(et), C.byref(dc)))
        return rem.value, et.value, bool(dc.value)


class Tablet:
    """nimble_amd::Tablet over file bytes held by this object."""

    def __init__(self, data: bytes):
        self._data = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        self._h = lib().nblh_tablet_open(_p(self._data), self._data.size)
        if not self._h:
            _chk(-1)
        self.stripe_count = lib().nblh_tablet_stripe_count(self._h)
        self.row_count = lib().nblh_tablet_row_count(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().nblh_tablet_free(self._h)
            self._h = None

    def stripe_info(self, i):
        rows, off, size, streams = C.c_uint32(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        _chk(lib().nblh_tablet_stripe_info(self._h, i, C.byref(rows), C.byref(off), C.byref(size), C.byref(streams)))
        return rows.value, off.value, size.value, streams.value

    def stream(self, stripe, stream_id) -> bytes:
        p, n = C.c_void_p(), C.c_uint64()
        _chk(lib().nblh_tablet_stream(self._h, stripe, stream_id, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value) if n.value else b""

    def optional_section(self, name: str) -> bytes:
        p, n = C.c_void_p(), C.c_uint64()
        _chk(lib().nblh_tablet_optional_section(self._h, name.encode(), C.byref(p), C.byref(n)))
        return C.string_at(p, n.value) if n.value else b""

    def decode_stripe(self, factory: "Factory", stripe):
        """(StripeDecoder-like object, stream ids) for all non-empty streams of the stripe."""
        ids = np.zeros(max(self.stripe_info(stripe)[3], 1), dtype=np.uint32)
        n = C.c_uint32()
        h = lib().nblh_tablet_decode_stripe(factory._h, self._h, stripe, _p(ids), ids.size, C.byref(n))
        if not h:
            _chk(-1)
        sd = StripeDecoder.__new__(StripeDecoder)
        sd._factory, sd._streams, sd._h, sd._keep = factory, [], h, self
        return sd, ids[: n.value].tolist()


def assign_stripes(sizes, devices):
    """GpuTabletDecoder::assignStripes: per device, its stripes ascending."""
    a = np.asarray(sizes, dtype=np.uint64)
    owner = np.zeros(max(a.size, 1), dtype=np.uint32)
    _chk(lib().nblh_assign_stripes(_p(a), a.size, devices, _p(owner)))
    return [[s for s in range(a.size) if owner[s] == d] for d in range(devices)]


class TabletDecoder:
    """GpuTabletDecoder: all stripes of a Tablet over a list of devices."""

    def __init__(self, tablet: Tablet, devices, varint_rows=False, keep_dictionary=False, lookahead=2):
        self._tablet = tablet
        dev = np.asarray(devices, dtype=np.int32)
        self._h = lib().nblh_tablet_decoder_create(tablet._h, _p(dev), dev.size, int(varint_rows) | (2 if keep_dictionary else 0), lookahead)
        if not self._h:
            _chk(-1)

    def device_of(self, stripe):
        d = C.c_int32()
        _chk(lib().nblh_tablet_decoder_device_of(self._h, stripe, C.byref(d)))
        return d.value

    def take(self, stripe):
        """(StripeDecoder-like object, stream ids) of a decoded stripe; blocks until it is ready."""
        ids = np.zeros(max(self._tablet.stripe_info(stripe)[3], 1), dtype=np.uint32)
        n = C.c_uint32()
        h = lib().nblh_tablet_decoder_take(self._h, stripe, _p(ids), ids.size, C.byref(n))
        if not h:
            _chk(-1)
        sd = StripeDecoder.__new__(StripeDecoder)
        sd._factory, sd._streams, sd._h, sd._keep = None, [], h, self
        return sd, ids[: n.value].tolist()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().nblh_tablet_decoder_free(self._h)
            self._h = None


class DeviceBatch:
    """GpuEncodingFactory::decodeToDevice: decoded chunks left in device memory."""

    def __init__(self, factory: Factory, chunks):
        from nimble_amd import capi

        self._chunks = [bytes(c) for c in chunks]
        arr = (C.c_char_p * len(self._chunks))(*self._chunks)
        sizes = np.array([len(c) for c in self._chunks], dtype=np.uint64)
        self._h = lib().nblh_decode_to_device(factory._h, arr, _p(sizes), len(self._chunks))
        if not self._h:
            _chk(-1)
        self.device_ptr = lib().nblh_device_batch_data(self._h)
        self.bytes = lib().nblh_device_batch_bytes(self._h)
        self.layouts = []
        for i in range(len(self._chunks)):
            lay = capi.NblChunkLayout()
            _chk(lib().nblh_device_batch_layout(self._h, i, C.byref(lay)))
            self.layouts.append(lay)

    def to_host(self) -> np.ndarray:
        out = np.zeros(max(int(self.bytes), 1), dtype=np.uint8)
        _chk(lib().nblh_device_batch_copy_out(self._h, _p(out)))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().nblh_device_batch_free(self._h)
            self._h = None
