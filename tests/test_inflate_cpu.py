"""CPU: the host pre-pass that inflates compressed leaf payloads.  A chunk written
with Zstd / LZ4 leaves must turn into exactly the bytes the same cascade has with
compression off (child-size fields patched), which the planner then accepts."""
import copy
import ctypes as C

import numpy as np
import pytest

from nimble_amd import capi
from nimble_amd import encoder as E
from nimble_amd import wire as W
from nimble_amd.decoder import Plan

from util import random_strings

pa = pytest.importorskip("pyarrow")


def _with_compression(spec, codec):
    """Deep copy of `spec` with every Trivial / FixedBitWidth leaf compressed."""
    s = copy.deepcopy(spec)

    def walk(x):
        if isinstance(x, (E.Trivial, E.FixedBitWidth, E.BlockBitPacking, E.FOR)):
            x.compression = codec
        for name in ("lengths", "values", "alphabet", "indices", "data", "nulls", "is_common", "other",
                     "baselines", "bit_widths", "offsets", "references", "bit_offsets"):
            child = getattr(x, name, None)
            if child is not None and not isinstance(child, (int, bool)):
                walk(child)

    walk(s)
    return s


def _cases(rng):
    n = 3000
    ints = np.repeat(rng.integers(0, 50, n // 10 + 1), 10)[:n].astype(np.int64)
    strs = random_strings(rng, 25).take(rng.integers(0, 25, n))
    bools = rng.random(n) < 0.1
    yield E.Trivial(), ints, W.INT64
    yield E.FixedBitWidth(), ints, W.INT64
    yield E.Trivial(), bools, W.BOOL
    yield E.Trivial(lengths=E.Trivial()), strs, W.STRING
    yield E.RLE(lengths=E.FixedBitWidth(), values=E.Trivial()), ints, W.INT64
    yield E.Dictionary(alphabet=E.Trivial(lengths=E.FixedBitWidth()), indices=E.FixedBitWidth()), strs, W.STRING
    yield E.MainlyConstant(is_common=E.Trivial(), other=E.FixedBitWidth()), ints, W.INT64
    yield E.SparseBool(indices=E.Trivial()), bools, W.BOOL
    yield E.RLE(lengths=E.RLE(lengths=E.Trivial(), values=E.FixedBitWidth())), bools, W.BOOL
    # block encodings: compressed packed blocks, and varint-sized children whose size changes when inflated
    yield E.BlockBitPacking(block_size=64), ints, W.INT64
    yield E.FOR(bit_widths=E.FixedBitWidth(), references=E.Trivial(), bit_offsets=E.Trivial(), frame_size=16), ints, W.INT64
    yield E.RLE(lengths=E.FOR(frame_size=8), values=E.BlockBitPacking(block_size=32, offsets=E.FixedBitWidth())), ints, W.INT64


@pytest.mark.parametrize("codec", ["zstd", "lz4"])
@pytest.mark.parametrize("varint_rows", [False, True])
def test_inflated_chunk_equals_the_uncompressed_encoding(codec, varint_rows):
    rng = np.random.default_rng(8)
    for spec, values, dtype in _cases(rng):
        plain = E.encode(spec, values, dtype, varint_rows)
        packed = E.encode(_with_compression(spec, codec), values, dtype, varint_rows)
        assert packed != plain
        with pytest.raises(capi.NimbleUnsupportedError, match="compressed"):
            Plan.from_encodings([packed], varint_rows)
        assert capi.inflate_chunk(packed, varint_rows) == plain, repr(spec)
        assert capi.inflate_chunk(plain, varint_rows) == plain
        Plan.from_encodings([capi.inflate_chunk(packed, varint_rows)], varint_rows).close()
    # Nullable on top
    mask = rng.random(2000) < 0.6
    v = rng.integers(0, 99, int(mask.sum())).astype(np.int32)
    spec = E.Nullable(data=E.FixedBitWidth(), nulls=E.Trivial())
    plain = E.encode_nullable(spec, v, mask, W.INT32, varint_rows)
    packed = E.encode_nullable(_with_compression(spec, codec), v, mask, W.INT32, varint_rows)
    assert capi.inflate_chunk(packed, varint_rows) == plain


def test_whole_chunk_compression_and_damage():
    rng = np.random.default_rng(9)
    enc = E.encode(E.FixedBitWidth(), rng.integers(0, 1000, 5000).astype(np.int32), W.INT32)
    for codec in ("zstd", "lz4"):
        stream = E.chunk(enc, codec) + E.chunk(enc)
        a = capi.as_u8(stream)
        n = C.c_uint32(0)
        offs, sizes, comp = np.zeros(2, np.uint64), np.zeros(2, np.uint64), np.zeros(2, np.uint8)
        capi.check(capi.lib().nbl_split_stream_ex(a.ctypes.data, a.size, offs.ctypes.data, sizes.ctypes.data, comp.ctypes.data, 2, C.byref(n)))
        assert n.value == 2 and comp.tolist() == [{"zstd": 1, "lz4": 3}[codec], 0]
        need = C.c_uint64(0)
        capi.check(capi.lib().nbl_uncompressed_size(int(comp[0]), a.ctypes.data + int(offs[0]), int(sizes[0]), C.byref(need)))
        assert need.value == len(enc)
        out = np.zeros(need.value, np.uint8)
        capi.check(capi.lib().nbl_uncompress(int(comp[0]), a.ctypes.data + int(offs[0]), int(sizes[0]), out.ctypes.data, out.size))
        assert out.tobytes() == enc
        with pytest.raises(capi.NimbleUnsupportedError):  # the plain splitter still refuses compressed chunks
            capi.split_stream(stream)
        # a damaged frame is an error, not a crash
        bad = bytearray(E.encode(E.Trivial(compression=codec), rng.integers(0, 9, 4000).astype(np.int64), W.INT64))
        bad[len(bad) // 2] ^= 0x5A
        bad[len(bad) // 2 + 1] ^= 0xA5
        try:
            capi.inflate_chunk(bytes(bad))
        except capi.NimbleCorruptError:
            pass
    with pytest.raises(capi.NimbleUnsupportedError, match="compression type"):
        capi.inflate_chunk(enc[:6] + bytes([4]) + enc[7:])  # OpenZL


def _varint(v):
    out = bytearray()
    while v >= 128:
        out.append(128 | (v & 127))
        v >>= 7
    out.append(v)
    return bytes(out)


def test_nested_block_encodings_are_sized_once():
    """A chain of BlockBitPacking encodings nested through the baselines child,
    31 deep: the rewritten size of every varint-sized child is computed by ONE
    walk, so the pre-pass stays linear in the chunk (sizing each child by
    walking it again at every level would take 2^31 walks here)."""
    import time

    small = E.encode(E.Trivial(), np.zeros(1, dtype=np.uint8), W.UINT8)
    offs = E.encode(E.Trivial(), np.zeros(1, dtype=np.uint32), W.UINT32)
    enc = E.encode(E.Trivial(compression="zstd"), np.arange(64, dtype=np.int64), W.INT64)
    plain = E.encode(E.Trivial(), np.arange(64, dtype=np.int64), W.INT64)

    def wrap(child):
        return (E.prefix(W.BLOCK_BIT_PACKING, W.INT64, 1) + bytes([W.UNCOMPRESSED]) + _varint(1) + _varint(1)
                + _varint(len(child)) + child + _varint(len(small)) + small + _varint(len(offs)) + offs + _varint(1) + bytes(8))

    for _ in range(30):
        enc, plain = wrap(enc), wrap(plain)
    t0 = time.perf_counter()
    assert capi.inflate_chunk(enc) == plain
    assert time.perf_counter() - t0 < 1.0


def test_empty_chunk_is_corrupt_not_a_bad_argument():
    need = C.c_uint64(0)
    assert capi.lib().nbl_chunk_inflated_size(None, 0, None, C.byref(need)) == capi.NBL_ERR_CORRUPT
