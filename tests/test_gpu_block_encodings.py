"""GPU parity for the block encodings (BlockBitPacking, FOR): the HIP decode
through the C ABI against the oracle, bit-exact, over the shapes the reference
tests use (encodings/tests/BlockBitPackingEncodingTest.cpp, ForEncodingTest.cpp:
constant / raw / partial blocks, sliced first block), nested under the
production encodings, and with inconsistent block tables."""
import struct

import numpy as np
import pytest

from nimble_amd import capi
from nimble_amd import encoder as E
from nimble_amd import wire as W
from nimble_amd.decoder import BatchDecoder, Plan

from test_block_encodings_cpu import _trivial, _vs, block_cases
from util import NP_OF, check_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("varint_rows", [False, True])
def test_block_encodings_matrix(gpu, oracle, varint_rows):
    rng = np.random.default_rng(151)
    encs, labels = [], []
    for label, dtype, v, spec in block_cases(rng):
        encs.append(E.encode(spec, v, dtype, varint_rows))
        labels.append(label)
    check_parity(oracle, encs, varint_rows, labels)


def test_for_every_width_and_alignment(gpu, oracle):
    """Frames of every width 0..64 (56 at most off byte boundaries, where the
    reference's bit cursor is exact) behind a first frame that leaves the next
    one at every bit alignment."""
    rng = np.random.default_rng(64)
    encs, labels = [], []
    for width in range(0, 65):
        for first in (1, 3, 8):
            n = first + 300
            v = np.zeros(n, dtype=np.uint64)
            v[:first] = rng.integers(0, 2, first)  # 1-bit first frame: the next starts at bit `first`
            if first == 1:
                v[0] = 1
            else:
                v[0], v[1] = 0, 1
            w = width if (first % 8 == 0 or width <= 56) else 56
            body = rng.integers(0, 2**63, 300, dtype=np.uint64)
            body = body & np.uint64((1 << w) - 1) if w < 64 else body * np.uint64(2) + np.uint64(1)
            if w:
                body[0], body[1] = 0, (1 << w) - 1
            else:
                body[:] = 5
            v[first:] = body
            encs.append(E.encode(E.FOR(exact_bits=True, frame_size=300, first_frame_rows=first), v, W.UINT64))
            labels.append(f"w={w} first={first}")
    check_parity(oracle, encs, labels=labels)


def test_block_encodings_nested(gpu, oracle):
    """Block encodings as children of the production encodings."""
    rng = np.random.default_rng(9)
    n = 20000
    base = np.repeat(rng.integers(0, 1 << 40, n // 500 + 1), 500)[:n] + rng.integers(0, 100, n)
    v = base.astype(np.int64)
    mask = rng.random(n) < 0.8
    runs = np.repeat(rng.integers(0, 1 << 20, n // 7 + 1), 7)[:n].astype(np.uint32)
    encs = [
        E.encode_nullable(E.Nullable(data=E.BlockBitPacking(), nulls=E.SparseBool()), v[mask], mask, W.INT64),
        E.encode_nullable(E.Nullable(data=E.FOR(), nulls=E.Trivial()), v[mask], mask, W.INT64),
        E.encode(E.RLE(lengths=E.FOR(), values=E.BlockBitPacking()), runs, W.UINT32),
        E.encode(E.Dictionary(alphabet=E.BlockBitPacking(block_size=16), indices=E.FOR()), runs, W.UINT32),
        E.encode(E.MainlyConstant(is_common=E.SparseBool(indices=E.FOR()), other=E.BlockBitPacking()),
                 np.where(rng.random(n) < 0.9, 7, v).astype(np.int64), W.INT64),
        E.encode(E.BlockBitPacking(baselines=E.FOR(frame_size=4), bit_widths=E.BlockBitPacking(block_size=8),
                                   offsets=E.FOR()), v, W.INT64),
    ]
    check_parity(oracle, encs)


def test_tiles_with_many_segments(gpu, oracle):
    """Segments far smaller than a 16 KiB tile of rows (hundreds per tile, more
    than the kernel keeps parameters for) and just below that limit."""
    rng = np.random.default_rng(160)
    encs, labels = [], []
    for dtype, n, seg in ((W.INT8, 40000, 50), (W.INT8, 40000, 128), (W.UINT16, 20000, 16), (W.UINT16, 20000, 51),
                          (W.UINT32, 9000, 7), (W.INT64, 5000, 3), (W.INT64, 5000, 13)):
        npt = NP_OF[dtype]
        bits = 8 * np.dtype(npt).itemsize
        steps = np.repeat(rng.integers(0, 1 << (bits - 1), n // seg + 1, dtype=np.uint64), seg)[:n]
        v = (steps + rng.integers(0, 1 << min(bits - 2, 11), n, dtype=np.uint64)).astype(W.physical_dtype(dtype)).view(npt)
        specs = [E.BlockBitPacking(block_size=seg, first_block_rows=max(1, seg // 3))]
        if dtype in W.INTEGER_TYPES:
            specs += [E.FOR(frame_size=seg, exact_bits=True), E.FOR(frame_size=seg, first_frame_rows=1)]
        for spec in specs:
            encs.append(E.encode(spec, v, dtype))
            labels.append(f"{W.DATA_TYPE_NAMES[dtype]}/{n}/{spec!r}")
    check_parity(oracle, encs, labels=labels)


def test_blocks_scattered_in_the_payload(gpu, oracle):
    """Block offsets need not ascend: blocks stored in another order, adjacent
    (one tile's bytes still fit the staging area) and far apart (they do not)."""
    rng = np.random.default_rng(161)
    nb, bs = 8, 1024
    encs = []
    for dtype, order in ((W.UINT64, [7, 6, 5, 4, 3, 2, 1, 0]), (W.UINT64, [0, 4, 1, 5, 2, 6, 3, 7]),
                         (W.UINT32, [3, 7, 0, 4, 1, 5, 2, 6])):
        npt = NP_OF[dtype]
        es = np.dtype(npt).itemsize
        v = rng.integers(0, 1 << 30, nb * bs).astype(npt)
        raw = [k % 3 == 0 for k in range(nb)]  # every third block raw, the others packed at 30 bits from baseline 0
        blobs = []
        for k in range(nb):
            part = v[k * bs:(k + 1) * bs]
            blobs.append(part.tobytes() if raw[k] else E.bitpack_variable(part, np.full(bs, 30))[0] + bytes(7))
        offsets, pos = [0] * nb, 0
        payload = b""
        for k in order:  # block k is stored at the position `order` gives it
            offsets[k] = pos
            payload += blobs[k]
            pos += len(blobs[k])
        widths = [255 if r else 30 for r in raw]
        enc = (struct.pack("<BBI", W.BLOCK_BIT_PACKING, dtype, nb * bs) + b"\0" + E._write_varint(bs) + E._write_varint(nb) +
               E._varint_string(_trivial(dtype, [0] * nb)) + E._varint_string(_trivial(W.UINT8, widths)) +
               E._varint_string(_trivial(W.UINT32, offsets)) + E._write_varint(bs) + payload)
        assert np.array_equal(oracle.Encoding(enc).materialize(nb * bs), v)
        encs.append(enc)
    check_parity(oracle, encs)


def _decode_status(enc):
    plan = Plan.from_encodings([enc])
    dec = BatchDecoder(plan)
    dec.stage()
    dec.decode()
    try:
        dec.status()
    finally:
        dec.torch.cuda.synchronize()
        plan.close()


def test_inconsistent_block_tables_are_reported(gpu, oracle):
    """Offsets or widths that point outside the packed payload raise the
    device-data status instead of reading out of bounds; the oracle refuses the
    same streams."""
    payload = np.arange(8, dtype=np.uint32).tobytes()

    def bbp(widths, offsets):
        return (struct.pack("<BBI", W.BLOCK_BIT_PACKING, W.UINT32, 8) + b"\0" + bytes([4, 2]) +
                _vs(_trivial(W.UINT32, [0, 0])) + _vs(_trivial(W.UINT8, widths)) + _vs(_trivial(W.UINT32, offsets)) +
                bytes([4]) + payload)

    good = bbp([255, 255], [0, 16])
    check_parity(oracle, [good])
    for bad in (bbp([255, 255], [0, 20]),       # raw block runs past the payload
                bbp([255, 255], [0, 0xFFFFFFF0]),
                bbp([31, 40], [0, 0]),           # width 40 on a 4-byte type
                bbp([31, 31], [0, 30])):         # packed block runs past the payload
        with pytest.raises(oracle.OracleError):
            e = oracle.Encoding(bad)
            e.materialize(8)
        with pytest.raises(capi.NimbleError) as ei:
            _decode_status(bad)
        assert ei.value.status == capi.NBL_ERR_DEVICE_DATA

    def frames(widths, offsets, packed=payload):
        return (struct.pack("<BBI", W.FOR, W.UINT32, 8) + b"\0" + bytes([4, 2, 4]) + _vs(_trivial(W.UINT8, widths)) +
                _vs(_trivial(W.UINT32, [0, 0])) + _vs(_trivial(W.UINT64, offsets)) + bytes([len(packed)]) + packed)

    check_parity(oracle, [frames([32, 32], [0, 128])])
    for bad in (frames([32, 32], [0, 160]), frames([32, 65], [0, 0]), frames([32, 32], [0, 2**63])):
        with pytest.raises(oracle.OracleError):
            oracle.Encoding(bad).materialize(8)
        with pytest.raises(capi.NimbleError) as ei:
            _decode_status(bad)
        assert ei.value.status == capi.NBL_ERR_DEVICE_DATA
