"""BlockBitPacking / FOR on the CPU side: the oracle against hand-assembled
bytes (layouts: encodings/BlockBitPackingEncoding.h:1207-1223,
encodings/ForEncoding.h:43-52,740-754), encoder/oracle round trips over the
reference tests' shapes (encodings/tests/BlockBitPackingEncodingTest.cpp,
ForEncodingTest.cpp: constant blocks, raw blocks, partial last block, sliced
first block), EncodingView-style range reads, and what the planner makes of them.

Parity pin status: the reference holds no encoded-byte vectors for these two
encodings either; the hand-assembled streams below follow its layout comments."""
import struct

import numpy as np
import pytest

from nimble_amd import capi
from nimble_amd import encoder as E
from nimble_amd import wire as W

from util import NP_OF


def _trivial(dtype, values):
    return struct.pack("<BBI", W.TRIVIAL, dtype, len(values)) + b"\0" + np.asarray(values, dtype=NP_OF[dtype]).tobytes()


def _vs(b):
    assert len(b) < 128
    return bytes([len(b)]) + b


def test_oracle_block_bit_packing_hand_assembled(oracle):
    """Three blocks of block size 4 over 10 uint32 rows: packed (w=3, base 100),
    constant (w=0, base 7: its 7 reserved bytes hold nothing), raw (marker 255)."""
    # residuals 0,1,2,7 at 3 bits, LSB first: v0 in bits 0-2, v1 in 3-5, v2 in 6-8, v3 in 9-11; + 7 slop bytes
    packed = bytes([0b10001000, 0b00001110]) + bytes(7)
    const_block = bytes(7)
    raw = np.array([0xFFFFFFFF, 5], dtype=np.uint32).tobytes()
    payload = packed + const_block + raw
    enc = (struct.pack("<BBI", W.BLOCK_BIT_PACKING, W.UINT32, 10) + b"\0" + bytes([4]) + bytes([3]) +
           _vs(_trivial(W.UINT32, [100, 7, 0])) + _vs(_trivial(W.UINT8, [3, 0, 255])) +
           _vs(_trivial(W.UINT32, [0, len(packed), len(packed) + 7])) + bytes([4]) + payload)
    e = oracle.Encoding(enc)
    assert e.row_count == 10 and e.encoding_type == W.BLOCK_BIT_PACKING
    assert e.materialize(10).tolist() == [100, 101, 102, 107, 7, 7, 7, 7, 0xFFFFFFFF, 5]
    assert e.read(3, 6).tolist() == [107, 7, 7, 7, 7, 0xFFFFFFFF]
    # what the writer produces for the same rows is byte-identical
    ours = E.encode(E.BlockBitPacking(block_size=4), np.array([100, 101, 102, 107, 7, 7, 7, 7, 0xFFFFFFFF, 5], dtype=np.uint32), W.UINT32)
    assert ours == enc


def test_oracle_for_hand_assembled(oracle):
    """Two frames of frame size 4 over 6 uint16 rows: w=2 (reference 10) then w=8
    (reference 1000), the second frame starting at bit 8."""
    frame0 = 0b11_10_01_00  # residuals 0,1,2,3
    payload = bytes([frame0, 0, 255])
    enc = (struct.pack("<BBI", W.FOR, W.UINT16, 6) + b"\0" + bytes([4, 2, 4]) + _vs(_trivial(W.UINT8, [2, 8])) +
           _vs(_trivial(W.UINT16, [10, 1000])) + _vs(_trivial(W.UINT64, [0, 8])) + bytes([len(payload)]) + payload)
    e = oracle.Encoding(enc)
    assert e.materialize(6).tolist() == [10, 11, 12, 13, 1000, 1255]
    assert e.read(2, 3).tolist() == [12, 13, 1000]
    ours = E.encode(E.FOR(frame_size=4), np.array([10, 11, 12, 13, 1000, 1255], dtype=np.uint16), W.UINT16)
    assert ours == enc


def _shapes(rng, dtype, n):
    npt = NP_OF[dtype]
    pt = W.physical_dtype(dtype)
    bits = 8 * np.dtype(npt).itemsize
    yield "narrow", rng.integers(0, 50, n).astype(npt)
    yield "locally-narrow", (np.repeat(rng.integers(0, 1 << (bits - 2), n // 100 + 1, dtype=np.uint64), 100)[:n] +
                             rng.integers(0, 7, n, dtype=np.uint64)).astype(pt).view(npt)
    yield "full-range", (rng.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)).astype(pt).view(npt)
    yield "constant", np.full(n, 42).astype(npt)
    mixed = rng.integers(0, 9, n).astype(npt)
    mixed[n // 2:] = rng.integers(0, 2**63, n - n // 2, dtype=np.uint64).astype(pt).view(npt)
    yield "half-raw", mixed


def block_cases(rng, dtypes=(W.INT8, W.UINT16, W.INT32, W.UINT32, W.INT64, W.UINT64, W.FLOAT, W.DOUBLE),
                sizes=(1, 7, 128, 129, 1024, 1025, 4099)):
    """(label, dtype, values, spec) over both encodings, nested metadata encodings and sliced starts."""
    for dtype in dtypes:
        for n in sizes:
            for shape, v in _shapes(rng, dtype, n):
                specs = [E.BlockBitPacking(), E.BlockBitPacking(block_size=100, first_block_rows=min(37, n)),
                         E.BlockBitPacking(baselines=E.FixedBitWidth(), bit_widths=E.RLE(), offsets=E.Varint())]
                if dtype in W.INTEGER_TYPES:
                    specs += [E.FOR(), E.FOR(exact_bits=True, frame_size=50, first_frame_rows=min(13, n)),
                              E.FOR(bit_widths=E.Dictionary(), references=E.FixedBitWidth(exact_bits=True), bit_offsets=E.Varint())]
                for spec in specs:
                    yield f"{type(spec).__name__}/{W.DATA_TYPE_NAMES[dtype]}/{n}/{shape}/{spec!r}", dtype, v, spec


def test_oracle_round_trip(oracle):
    rng = np.random.default_rng(15)
    for label, dtype, v, spec in block_cases(rng, sizes=(1, 129, 1025)):
        for vr in (False, True):
            e = oracle.Encoding(E.encode(spec, v, dtype, vr), vr)
            want = v.view(W.physical_dtype(dtype))
            assert np.array_equal(e.materialize(v.size), want), label
            e.reset()
            e.skip(v.size // 3)
            assert np.array_equal(e.materialize(v.size - v.size // 3), want[v.size // 3:]), label
            if v.size > 3:  # EncodingView::read is cursor independent
                assert np.array_equal(e.read(2, v.size - 3), want[2:v.size - 1]), label


def test_writer_width_rules():
    """FOR rounds widths up to {1,2,4,8,16,32,64} with 1 for a constant frame
    (ForEncoding.h:227-240); BlockBitPacking keeps exact widths, 0 for constant
    blocks and the raw marker when packing does not pay (:571-593)."""
    v = np.concatenate([np.full(128, 9), np.arange(128) % 5, np.arange(128) * 1000]).astype(np.uint32)
    enc = E.encode(E.FOR(), v, W.UINT32)
    # widths child is the first varint string after compression byte + 3 varints
    pos = 6 + 1 + 2 + 1 + 2  # frame size 128 takes two varint bytes, as does first-frame rows
    size = enc[pos]
    widths = np.frombuffer(enc[pos + 1 + 7:pos + 1 + size], dtype=np.uint8)
    assert widths.tolist() == [1, 4, 32]
    v = np.concatenate([np.full(1024, 9), np.arange(1024) % 5, np.arange(1024) * 4000000]).astype(np.uint32)
    enc = E.encode(E.BlockBitPacking(), v, W.UINT32)
    pos = 6 + 1 + 2 + 1
    pos += 1 + enc[pos]  # baselines
    size = enc[pos]
    widths = np.frombuffer(enc[pos + 1 + 7:pos + 1 + size], dtype=np.uint8)
    assert widths.tolist() == [0, 3, 255]


@pytest.mark.parametrize("spec,dtype", [(E.BlockBitPacking(), W.INT64), (E.FOR(), W.INT32),
                                        (E.BlockBitPacking(baselines=E.RLE(), offsets=E.Varint()), W.UINT16)])
def test_planner_accepts_block_encodings(spec, dtype):
    rng = np.random.default_rng(3)
    v = rng.integers(0, 1000, 5000).astype(NP_OF[dtype])
    enc = E.encode(spec, v, dtype)
    from nimble_amd.decoder import Plan

    for per_kernel in (False, True):
        plan = Plan.from_encodings([enc], per_kernel=per_kernel)
        names = [l[0] for l in plan.launches()]
        # the block unpack keeps a launch of its own in fused plans too
        assert names.count(f"k_frame_unpack<{W.elem_bytes(dtype)}>") == 1, names
        assert names[-1].startswith("k_frame_unpack<"), names
        assert plan.info.total_rows == 5000
        lay = plan.layouts[0]
        assert lay.encoding_type == (W.FOR if isinstance(spec, E.FOR) else W.BLOCK_BIT_PACKING)
        assert lay.values_bytes == 5000 * W.elem_bytes(dtype)
        plan.close()


def test_planner_rejects_bad_geometry():
    from nimble_amd.decoder import Plan

    v = np.arange(300, dtype=np.uint32)
    good = E.encode(E.BlockBitPacking(block_size=100), v, W.UINT32)
    bad = bytearray(good)
    bad[8] = 2  # block count 3 -> 2
    with pytest.raises(capi.NimbleCorruptError):
        Plan.from_encodings([bytes(bad)])
    f = bytearray(E.encode(E.FOR(frame_size=100), v, W.UINT32))
    f[9] = 0  # first frame rows 100 -> 0
    with pytest.raises(capi.NimbleCorruptError):
        Plan.from_encodings([bytes(f)])
    with pytest.raises(capi.NimbleCorruptError):  # FOR is integers only (EncodingFactory.cpp:155-157)
        Plan.from_encodings([bytes([W.FOR, W.FLOAT]) + bytes(f[2:])])
