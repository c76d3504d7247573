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
