"""Test-vector / synthetic-table generator: a numpy writer for the nine
production stream encodings of Nimble plus the two block encodings nearest to
FixedBitWidth (BlockBitPacking, FOR).

This is the *input fabricator* for tests and benchmarks -- the reference's own
writer cannot be built offline.  Byte layouts follow the reference encoders:

* prefix ............ dwio/nimble/encodings/common/EncodingPrefix.h:41-54
* Trivial ........... dwio/nimble/encodings/TrivialEncoding.h:303-330,
                      TrivialEncoding.cpp:170-231 (string), :383-427 (bool)
* FixedBitWidth ..... dwio/nimble/encodings/FixedBitWidthEncoding.h:351-412
* RLE ............... dwio/nimble/encodings/RLEEncoding.h:62-85,178-226,758-766
* Dictionary ........ dwio/nimble/encodings/DictionaryEncoding.h:391-488
* Nullable .......... dwio/nimble/encodings/NullableEncoding.h:370-405
* SparseBool ........ dwio/nimble/encodings/SparseBoolEncoding.cpp:178-239
* Constant .......... dwio/nimble/encodings/ConstantEncoding.h:163-200
* MainlyConstant .... dwio/nimble/encodings/MainlyConstantEncoding.h:558-570,863-915
* Varint ............ dwio/nimble/encodings/VarintEncoding.h:164-211
* BlockBitPacking ... dwio/nimble/encodings/BlockBitPackingEncoding.h:559-594,1100-1227
* FOR ............... dwio/nimble/encodings/ForEncoding.h:224-240,590-758
* chunk framing ..... dwio/nimble/common/ChunkHeader.h:43-47

Unlike the reference writer there is no selection policy: the caller names the
cascade explicitly with the spec classes below (the same role the reference's
tests give to TestTrivialEncodingSelectionPolicy / hand-assembled cascades).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Optional, Sequence, Union

import numpy as np

from . import wire as W

__all__ = [
    "StringColumn", "Trivial", "FixedBitWidth", "Varint", "Constant", "RLE",
    "Dictionary", "Nullable", "SparseBool", "MainlyConstant", "BlockBitPacking", "FOR", "encode",
    "encode_nullable", "encode_dictionary_from_indices", "nullable_wrap", "mainly_constant_from_parts",
    "chunk", "stream", "bitpack", "varint_bytes",
    "bits_required",
]


# --------------------------------------------------------------------------
# value containers
# --------------------------------------------------------------------------
@dataclass
class StringColumn:
    """A column of byte strings held as (lengths, concatenated bytes)."""

    lengths: np.ndarray  # uint32[n]
    blob: np.ndarray  # uint8[sum(lengths)]

    def __post_init__(self):
        self.lengths = np.ascontiguousarray(self.lengths, dtype=np.uint32)
        self.blob = np.ascontiguousarray(self.blob, dtype=np.uint8)
        assert int(self.lengths.sum(dtype=np.uint64)) == self.blob.size

    def __len__(self):
        return int(self.lengths.size)

    @staticmethod
    def from_list(items: Sequence[bytes]) -> "StringColumn":
        lengths = np.fromiter((len(s) for s in items), dtype=np.uint32, count=len(items))
        blob = np.frombuffer(b"".join(items), dtype=np.uint8)
        return StringColumn(lengths, blob)

    def offsets(self) -> np.ndarray:
        off = np.zeros(self.lengths.size + 1, dtype=np.uint64)
        np.cumsum(self.lengths, dtype=np.uint64, out=off[1:])
        return off

    def to_list(self):
        off = self.offsets()
        raw = self.blob.tobytes()
        return [raw[int(off[i]):int(off[i + 1])] for i in range(len(self))]

    def take(self, indices: np.ndarray) -> "StringColumn":
        indices = np.asarray(indices, dtype=np.int64)
        off = self.offsets()
        lens = self.lengths[indices]
        total = int(lens.sum(dtype=np.uint64))
        if total == 0:
            return StringColumn(lens, np.zeros(0, dtype=np.uint8))
        starts = off[indices].astype(np.int64)
        out_off = np.zeros(indices.size + 1, dtype=np.int64)
        np.cumsum(lens, dtype=np.int64, out=out_off[1:])
        # byte gather: position p of row r reads blob[starts[r] + p - out_off[r]]
        row_of = np.repeat(np.arange(indices.size, dtype=np.int64), lens.astype(np.int64))
        src = starts[row_of] + (np.arange(total, dtype=np.int64) - out_off[row_of])
        return StringColumn(lens, self.blob[src])


Values = Union[np.ndarray, StringColumn]


# --------------------------------------------------------------------------
# cascade specs
# --------------------------------------------------------------------------
@dataclass
class Trivial:
    lengths: object = None  # string only: encoding of the uint32 lengths
    compression: Optional[str] = None  # "zstd" | "lz4": compress the payload (values / bitmap / string blob)


@dataclass
class FixedBitWidth:
    exact_bits: bool = False  # Encoding::Options::fixedBitWidthUseExactBits
    bit_width: Optional[int] = None  # force a width (>= required)
    baseline: Optional[int] = None  # force a baseline (<= min)
    compression: Optional[str] = None  # "zstd" | "lz4": compress the packed bits


@dataclass
class Varint:
    pass


@dataclass
class Constant:
    pass


@dataclass
class RLE:
    lengths: object = field(default_factory=FixedBitWidth)
    values: object = field(default_factory=FixedBitWidth)


@dataclass
class Dictionary:
    alphabet: object = field(default_factory=Trivial)
    indices: object = field(default_factory=FixedBitWidth)


@dataclass
class Nullable:
    data: object = field(default_factory=Trivial)
    nulls: object = field(default_factory=Trivial)


@dataclass
class SparseBool:
    indices: object = field(default_factory=FixedBitWidth)


@dataclass
class MainlyConstant:
    is_common: object = field(default_factory=SparseBool)
    other: object = field(default_factory=Trivial)


@dataclass
class BlockBitPacking:
    """Blocks of `block_size` rows, each with its own baseline and exact bit
    width; blocks that packing would not shrink are stored raw."""

    baselines: object = field(default_factory=Trivial)
    bit_widths: object = field(default_factory=Trivial)
    offsets: object = field(default_factory=Trivial)
    block_size: int = W.BBP_MAX_BLOCK_SIZE
    first_block_rows: Optional[int] = None  # a sliced stream starts with a partial block
    compression: Optional[str] = None  # "zstd" | "lz4": compress the packed blocks


@dataclass
class FOR:
    """Frames of `frame_size` rows: per-frame minimum + residuals packed at a
    per-frame width, frames laid back to back at BIT granularity."""

    bit_widths: object = field(default_factory=Trivial)
    references: object = field(default_factory=Trivial)
    bit_offsets: object = field(default_factory=Trivial)
    frame_size: int = 128
    first_frame_rows: Optional[int] = None  # a sliced stream starts with a partial frame
    exact_bits: bool = False  # widths 0..64 as needed instead of the writer's {1,2,4,8,16,32,64}
    compression: Optional[str] = None


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def bits_required(v: int) -> int:
    """Bits needed for `v`; 1 for zero (a zero width is never written)."""
    return max(1, int(v).bit_length())


def _write_varint(v: int) -> bytes:
    out = bytearray()
    while v >= 128:  # dwio/nimble/common/Varint.h:70-76
        out.append(0x80 | (v & 0x7F))
        v >>= 7
    out.append(v)
    return bytes(out)


def prefix(enc: int, dtype: int, rows: int, varint_rows: bool = False) -> bytes:
    if varint_rows:
        return bytes([enc, dtype]) + _write_varint(rows)
    return struct.pack("<BBI", enc, dtype, rows)


def bitpack(values: np.ndarray, width: int) -> bytes:
    """LSB-first pack: value i occupies bits [i*w, (i+1)*w) of the byte stream;
    result is ceil(n*w/8) + 7 bytes (FixedBitArray::bufferSize)."""
    assert 1 <= width <= 64
    v = np.ascontiguousarray(values).astype(np.uint64, copy=False)
    n = v.size
    nbytes = (n * width + 7) // 8
    if n == 0:
        return bytes(W.FBA_SLOP)
    groups = (n + 63) // 64
    padded = np.zeros(groups * 64, dtype=np.uint64)
    padded[:n] = v
    if width < 64:
        assert int(padded.max()) < (1 << width), "value does not fit bit width"
    padded = padded.reshape(groups, 64)
    words = np.zeros((groups, width), dtype=np.uint64)  # 64 values = `width` words
    for j in range(64):
        bit0 = j * width
        wi, sh = bit0 >> 6, bit0 & 63
        col = padded[:, j]
        words[:, wi] |= col << np.uint64(sh)
        if sh + width > 64:
            words[:, wi + 1] |= col >> np.uint64(64 - sh)
    raw = words.reshape(-1).view(np.uint8)[:nbytes]
    return raw.tobytes() + bytes(W.FBA_SLOP)


def _varint_string(data: bytes) -> bytes:
    """varint length + bytes (encoding::writeVarintString)."""
    return _write_varint(len(data)) + data


def _segments(n: int, size: int, first: Optional[int]):
    """Row ranges of the blocks / frames of an n-row stream: a first segment of
    `first` rows (default `size`), then full segments, then the remainder."""
    first = min(size if first is None else int(first), n)
    assert 0 < first <= size
    bounds = [0, first]
    while bounds[-1] < n:
        bounds.append(min(bounds[-1] + size, n))
    return list(zip(bounds[:-1], bounds[1:]))


def bitpack_variable(values: np.ndarray, widths: np.ndarray) -> tuple:
    """LSB-first pack of value i into widths[i] bits, back to back at bit
    granularity.  Returns (bytes of ceil(total/8), start bit of every value)."""
    v = np.ascontiguousarray(values).astype(np.uint64, copy=False)
    w = np.ascontiguousarray(widths).astype(np.int64, copy=False)
    start = np.zeros(v.size + 1, dtype=np.int64)
    np.cumsum(w, out=start[1:])
    total = int(start[-1])
    bits = np.zeros(total, dtype=np.uint8)
    for k in range(int(w.max()) if w.size else 0):
        m = w > k
        bits[start[:-1][m] + k] = ((v[m] >> np.uint64(k)) & np.uint64(1)).astype(np.uint8)
    return np.packbits(bits, bitorder="little").tobytes(), start[:-1]


def varint_bytes(values: np.ndarray) -> bytes:
    """LEB128, low 7 bits first (dwio/nimble/common/Varint.h:70-76)."""
    v = np.ascontiguousarray(values).astype(np.uint64, copy=False)
    n = v.size
    if n == 0:
        return b""
    sizes = np.ones(n, dtype=np.int64)
    for k in range(1, 10):
        sizes += (v >= np.uint64(1 << (7 * k))).astype(np.int64)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(sizes, out=offs[1:])
    out = np.zeros(int(offs[-1]), dtype=np.uint8)
    for k in range(10):
        m = sizes > k
        if not m.any():
            break
        byte = ((v[m] >> np.uint64(7 * k)) & np.uint64(0x7F)).astype(np.uint8)
        byte |= (sizes[m] > k + 1).astype(np.uint8) << 7
        out[offs[:-1][m] + k] = byte
    return out.tobytes()


def compress_payload(raw: bytes, compression: Optional[str]):
    """(CompressionType byte, payload bytes) as Compression::compress frames them:
    u32 uncompressed size + Zstd frame (compression/ZstdCompressor.cpp:27-59) or
    raw LZ4 block (Lz4Compressor.cpp).  Needs pyarrow's codecs (test vectors only)."""
    if compression is None:
        return W.UNCOMPRESSED, raw
    import pyarrow as pa

    codec = {"zstd": "zstd", "lz4": "lz4_raw"}[compression]
    body = pa.Codec(codec).compress(raw, asbytes=True)
    return {"zstd": 1, "lz4": 3}[compression], struct.pack("<I", len(raw)) + body


def _phys(values: np.ndarray, dtype: int) -> np.ndarray:
    """Reinterpret `values` as the unsigned physical type of `dtype`."""
    pt = W.physical_dtype(dtype)
    a = np.ascontiguousarray(values)
    if a.dtype.itemsize != np.dtype(pt).itemsize:
        raise ValueError(f"array itemsize {a.dtype.itemsize} does not match data type {dtype}")
    return a.view(pt)


def _bitmap(bools: np.ndarray) -> bytes:
    b = np.ascontiguousarray(bools).astype(np.uint8, copy=False)
    return np.packbits(b, bitorder="little").tobytes() + bytes(W.FBA_SLOP)


# --------------------------------------------------------------------------
# encoders
# --------------------------------------------------------------------------
def encode(spec, values: Values, dtype: int, varint_rows: bool = False) -> bytes:
    """Serialise `values` (numpy array of the physical/logical type, bool array,
    or StringColumn) with the cascade `spec`.  Returns one encoding's bytes."""
    vr = varint_rows
    if dtype == W.STRING:
        assert isinstance(values, StringColumn)
        n = len(values)
    else:
        values = np.ascontiguousarray(values)
        n = int(values.size)
        if dtype == W.BOOL:
            values = values.astype(np.bool_, copy=False)
        else:
            values = _phys(values, dtype)

    if isinstance(spec, Trivial):
        if dtype == W.STRING:
            ct, body = compress_payload(values.blob.tobytes(), spec.compression)
            lens = encode(spec.lengths or FixedBitWidth(), values.lengths, W.UINT32, vr)
            return prefix(W.TRIVIAL, dtype, n, vr) + bytes([ct]) + struct.pack("<I", len(lens)) + lens + body
        ct, body = compress_payload(_bitmap(values) if dtype == W.BOOL else values.tobytes(), spec.compression)
        return prefix(W.TRIVIAL, dtype, n, vr) + bytes([ct]) + body

    if isinstance(spec, FixedBitWidth):
        assert W.is_numeric(dtype)
        lo = int(values.min()) if n else 0
        hi = int(values.max()) if n else 0
        base = lo if spec.baseline is None else int(spec.baseline)
        assert base <= lo
        width = bits_required(hi - base)
        if spec.bit_width is not None:
            assert spec.bit_width >= width
            width = spec.bit_width
        elif not spec.exact_bits:
            width = (width + 7) // 8 * 8
        residual = (values - values.dtype.type(base)) if n else values
        pt = W.physical_dtype(dtype)
        ct, body = compress_payload(bitpack(residual, width), spec.compression)
        return (prefix(W.FIXED_BIT_WIDTH, dtype, n, vr) + bytes([ct]) +
                np.array([base], dtype=pt).tobytes() + bytes([width]) + body)

    if isinstance(spec, Varint):
        assert W.elem_bytes(dtype) in (4, 8) and W.is_numeric(dtype)
        base = int(values.min()) if n else 0
        pt = W.physical_dtype(dtype)
        residual = (values - values.dtype.type(base)) if n else values
        return (prefix(W.VARINT, dtype, n, vr) + np.array([base], dtype=pt).tobytes() +
                varint_bytes(residual))

    if isinstance(spec, Constant):
        assert n > 0, "ConstantEncoding cannot be empty"
        if dtype == W.STRING:
            first = values.to_list()[0] if n < 1000 else values.take(np.array([0])).blob.tobytes()
            assert (values.lengths == values.lengths[0]).all()
            return prefix(W.CONSTANT, dtype, n, vr) + struct.pack("<I", len(first)) + first
        assert (values == values[0]).all(), "ConstantEncoding requires constant data"
        if dtype == W.BOOL:
            return prefix(W.CONSTANT, dtype, n, vr) + bytes([int(values[0])])
        return prefix(W.CONSTANT, dtype, n, vr) + values[:1].tobytes()

    if isinstance(spec, RLE):
        if dtype == W.STRING:
            ids = _string_ids(values)
            change = np.ones(n, dtype=bool)
            change[1:] = ids[1:] != ids[:-1]
        else:
            change = np.ones(n, dtype=bool)
            change[1:] = values[1:] != values[:-1]
        starts = np.flatnonzero(change)
        lengths = np.diff(np.append(starts, n)).astype(np.uint32)
        lens = encode(spec.lengths, lengths, W.UINT32, vr)
        head = prefix(W.RLE, dtype, n, vr) + struct.pack("<I", len(lens)) + lens
        if dtype == W.BOOL:
            return head + bytes([int(values[0]) if n else 0])
        run_values = values.take(starts) if dtype == W.STRING else values[starts]
        return head + encode(spec.values, run_values, dtype, vr)

    if isinstance(spec, Dictionary):
        assert dtype != W.BOOL
        keys = _string_ids(values) if dtype == W.STRING else values
        # alphabet in first-occurrence order (DictionaryEncoding.h:432-456)
        uniq, first_idx, inverse = np.unique(keys, return_index=True, return_inverse=True)
        order = np.argsort(first_idx, kind="stable")
        rank = np.empty_like(order)
        rank[order] = np.arange(order.size)
        indices = rank[inverse].astype(np.uint32)
        first_sorted = first_idx[order]
        alphabet = values.take(first_sorted) if dtype == W.STRING else values[first_sorted]
        alpha = encode(spec.alphabet, alphabet, dtype, vr)
        return (prefix(W.DICTIONARY, dtype, n, vr) + struct.pack("<I", len(alpha)) + alpha +
                encode(spec.indices, indices, W.UINT32, vr))

    if isinstance(spec, SparseBool):
        assert dtype == W.BOOL
        set_count = int(values.sum())
        if set_count > (n >> 1):  # SparseBoolEncoding.cpp:190-196
            sparse_value, idx = False, np.flatnonzero(~values)
        else:
            sparse_value, idx = True, np.flatnonzero(values)
        idx = np.append(idx, n).astype(np.uint32)  # sentinel, :215-217
        return (prefix(W.SPARSE_BOOL, dtype, n, vr) + bytes([int(sparse_value)]) +
                encode(spec.indices, idx, W.UINT32, vr))

    if isinstance(spec, MainlyConstant):
        assert dtype != W.BOOL and n > 0
        keys = _string_ids(values) if dtype == W.STRING else values
        uniq, first_idx, counts = np.unique(keys, return_index=True, return_counts=True)
        if dtype == W.STRING:
            # ties -> smaller string; order candidate keys by actual bytes
            cand = np.flatnonzero(counts == counts.max())
            strs = values.take(first_idx[cand]).to_list()
            best = cand[min(range(len(strs)), key=lambda i: strs[i])]
        else:
            best = int(np.flatnonzero(counts == counts.max())[0])  # uniq sorted -> smaller value
        is_common = keys == uniq[best]
        other_rows = np.flatnonzero(~is_common)
        other = values.take(other_rows) if dtype == W.STRING else values[other_rows]
        ic = encode(spec.is_common, is_common, W.BOOL, vr)
        ov = encode(spec.other, other, dtype, vr)
        if dtype == W.STRING:
            cv = values.take(np.array([first_idx[best]])).blob.tobytes()
            common = struct.pack("<I", len(cv)) + cv
        else:
            common = values[first_idx[best]:first_idx[best] + 1].tobytes()
        return (prefix(W.MAINLY_CONSTANT, dtype, n, vr) + struct.pack("<I", len(ic)) + ic +
                struct.pack("<I", len(ov)) + ov + common)

    if isinstance(spec, BlockBitPacking):
        assert W.is_numeric(dtype) and n > 0, "BlockBitPacking cannot encode empty streams."
        es = W.elem_bytes(dtype)
        pt = W.physical_dtype(dtype)
        segs = _segments(n, spec.block_size, spec.first_block_rows)
        assert 0 < spec.block_size <= W.BBP_MAX_BLOCK_SIZE and len(segs) <= W.BBP_MAX_BLOCKS
        baselines = np.zeros(len(segs), dtype=pt)
        widths = np.zeros(len(segs), dtype=np.uint8)
        offsets = np.zeros(len(segs), dtype=np.uint32)
        parts, running = [], 0
        for b, (s0, s1) in enumerate(segs):  # makeBlockInfo, BlockBitPackingEncoding.h:559-594
            block = values[s0:s1]
            lo, hi = int(block.min()), int(block.max())
            bits = 0 if hi == lo else (hi - lo).bit_length()
            raw_size = (s1 - s0) * es
            packed_size = ((s1 - s0) * bits + 7) // 8 + W.FBA_SLOP
            offsets[b] = running
            if bits == 8 * es or packed_size >= raw_size:
                widths[b] = W.BBP_RAW_BLOCK
                parts.append(block.tobytes())
            else:
                baselines[b], widths[b] = lo, bits
                parts.append(bitpack(block - block.dtype.type(lo), bits) if bits else bytes(W.FBA_SLOP))
            running += len(parts[-1])
        ct, body = compress_payload(b"".join(parts), spec.compression)
        unsigned = {1: W.UINT8, 2: W.UINT16, 4: W.UINT32, 8: W.UINT64}[es]
        return (prefix(W.BLOCK_BIT_PACKING, dtype, n, vr) + bytes([ct]) + _write_varint(spec.block_size) +
                _write_varint(len(segs)) + _varint_string(encode(spec.baselines, baselines, unsigned, vr)) +
                _varint_string(encode(spec.bit_widths, widths, W.UINT8, vr)) +
                _varint_string(encode(spec.offsets, offsets, W.UINT32, vr)) + _write_varint(segs[0][1]) + body)

    if isinstance(spec, FOR):
        assert dtype in W.INTEGER_TYPES and n > 0, "Cannot encode empty data with ForEncoding."
        es = W.elem_bytes(dtype)
        pt = W.physical_dtype(dtype)
        segs = _segments(n, spec.frame_size, spec.first_frame_rows)
        refs = np.zeros(len(segs), dtype=pt)
        widths = np.zeros(len(segs), dtype=np.uint8)
        row_width = np.zeros(n, dtype=np.int64)
        residual = np.zeros(n, dtype=np.uint64)
        for f, (s0, s1) in enumerate(segs):  # ForEncoding.h:653-692
            frame = values[s0:s1]
            lo, hi = int(frame.min()), int(frame.max())
            need = (hi - lo).bit_length()
            if spec.exact_bits:
                bw = need
            else:  # minBitWidth, :227-240
                bw = next(w for w in (1, 2, 4, 8, 16, 32, 64) if w >= max(need, 1))
            refs[f], widths[f] = lo, bw
            row_width[s0:s1] = bw
            residual[s0:s1] = (frame - frame.dtype.type(lo)).astype(np.uint64)
        packed, start = bitpack_variable(residual, row_width)
        bit_offsets = np.array([int(start[s0]) if s0 < n else 0 for s0, _ in segs], dtype=np.uint64)
        ct, body = compress_payload(packed, spec.compression)
        unsigned = {1: W.UINT8, 2: W.UINT16, 4: W.UINT32, 8: W.UINT64}[es]
        return (prefix(W.FOR, dtype, n, vr) + bytes([ct]) + _write_varint(spec.frame_size) + _write_varint(len(segs)) +
                _write_varint(segs[0][1]) + _varint_string(encode(spec.bit_widths, widths, W.UINT8, vr)) +
                _varint_string(encode(spec.references, refs, unsigned, vr)) +
                _varint_string(encode(spec.bit_offsets, bit_offsets, W.UINT64, vr)) + _write_varint(len(body)) + body)

    if isinstance(spec, Nullable):
        raise TypeError("use encode_nullable() for Nullable cascades")
    raise TypeError(f"unknown spec {spec!r}")


def encode_nullable(spec: Nullable, non_null_values: Values, non_null_mask: np.ndarray,
                    dtype: int, varint_rows: bool = False) -> bytes:
    """Nullable{data, nulls}: `non_null_values` holds only the non-null rows
    (densely), `non_null_mask[i]` is True when row i is NOT null
    (NullableEncoding.h:370-405)."""
    mask = np.ascontiguousarray(non_null_mask).astype(np.bool_, copy=False)
    n = int(mask.size)
    nn = int(mask.sum())
    assert (len(non_null_values) if dtype == W.STRING else int(np.asarray(non_null_values).size)) == nn
    data = encode(spec.data, non_null_values, dtype, varint_rows)
    nulls = encode(spec.nulls, mask, W.BOOL, varint_rows)
    return prefix(W.NULLABLE, dtype, n, varint_rows) + struct.pack("<I", len(data)) + data + nulls


def encode_dictionary_from_indices(spec: Dictionary, alphabet: Values, indices: np.ndarray, dtype: int,
                                    varint_rows: bool = False) -> bytes:
    """Dictionary encoding of `alphabet[indices]` without materialising the rows
    (for multi-million-row string columns).  `alphabet` must hold distinct
    values; entries are re-ordered to first-occurrence order and unused ones
    dropped, as the reference writer would (DictionaryEncoding.h:432-456)."""
    indices = np.ascontiguousarray(indices, dtype=np.int64)
    n = int(indices.size)
    used, first_idx, inverse = np.unique(indices, return_index=True, return_inverse=True)
    order = np.argsort(first_idx, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    new_indices = rank[inverse].astype(np.uint32)
    picked = used[order]
    alpha_vals = alphabet.take(picked) if dtype == W.STRING else _phys(alphabet, dtype)[picked]
    alpha = encode(spec.alphabet, alpha_vals, dtype, varint_rows)
    return (prefix(W.DICTIONARY, dtype, n, varint_rows) + struct.pack("<I", len(alpha)) + alpha +
            encode(spec.indices, new_indices, W.UINT32, varint_rows))


def mainly_constant_from_parts(spec: MainlyConstant, is_common: np.ndarray, other: np.ndarray, common, dtype: int,
                               varint_rows: bool = False) -> bytes:
    """MainlyConstant assembled from its parts (no frequency count): row i is
    `common` where is_common[i], else the next value of `other`
    (MainlyConstantEncoding.h:69-75)."""
    assert W.is_numeric(dtype)
    is_common = np.ascontiguousarray(is_common).astype(np.bool_, copy=False)
    other = _phys(np.ascontiguousarray(other), dtype)
    assert int((~is_common).sum()) == other.size
    ic = encode(spec.is_common, is_common, W.BOOL, varint_rows)
    ov = encode(spec.other, other, dtype, varint_rows)
    cv = _phys(np.array([common]), dtype).tobytes() if isinstance(common, np.generic) else \
        np.array([common], dtype=W.physical_dtype(dtype)).tobytes()
    return (prefix(W.MAINLY_CONSTANT, dtype, int(is_common.size), varint_rows) + struct.pack("<I", len(ic)) + ic +
            struct.pack("<I", len(ov)) + ov + cv)


def nullable_wrap(data_encoding: bytes, non_null_mask: np.ndarray, nulls_spec, dtype: int, varint_rows: bool = False) -> bytes:
    """Nullable around an already-encoded non-null child (NullableEncoding.h:370-405)."""
    mask = np.ascontiguousarray(non_null_mask).astype(np.bool_, copy=False)
    nulls = encode(nulls_spec, mask, W.BOOL, varint_rows)
    return prefix(W.NULLABLE, dtype, int(mask.size), varint_rows) + struct.pack("<I", len(data_encoding)) + data_encoding + nulls


def _string_ids(col: StringColumn) -> np.ndarray:
    """Dense integer id per row, equal ids <=> equal strings."""
    items = col.to_list()
    table = {}
    ids = np.empty(len(items), dtype=np.int64)
    for i, s in enumerate(items):
        ids[i] = table.setdefault(s, len(table))
    return ids


# --------------------------------------------------------------------------
# chunk / stream framing
# --------------------------------------------------------------------------
def chunk(encoding_bytes: bytes, compression=W.UNCOMPRESSED) -> bytes:
    """u32 length | u8 CompressionType | payload (ChunkHeader.h:43-47).  `compression`
    "zstd" / "lz4" compresses the whole chunk (velox/ChunkedStream.cpp:36-72); an
    integer only labels the header."""
    if isinstance(compression, str):
        ct, body = compress_payload(encoding_bytes, compression)
        return struct.pack("<IB", len(body), ct) + body
    return struct.pack("<IB", len(encoding_bytes), compression) + encoding_bytes


def stream(encodings: Sequence[bytes]) -> bytes:
    return b"".join(chunk(e) for e in encodings)
