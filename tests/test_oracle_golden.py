"""Pins the CPU oracle (and the test-vector encoder) to the hand-assembled
known-answer vectors of tests/golden/golden_v1.json, which restate the byte
layouts and inputs of the reference's own tests (see make_golden.py)."""
import numpy as np
import pytest

from nimble_amd import encoder as E
from nimble_amd import wire as W

from golden_util import assert_matches, load_cases
from util import oracle_chunk

CASES = load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_decodes_golden(oracle, case):
    vals, nulls, nn, n, dt = oracle_chunk(oracle, case["enc"], case["varint_rows"])
    assert dt == case["data_type"]
    assert_matches(case, vals, nulls, nn, case["name"])


def test_prefix_layout():
    """EncodingPrefixTest.cpp:42-83: type bytes, fixed u32 row count, varint row count."""
    p = E.prefix(W.DELTA, W.INT64, 123)
    assert p == bytes([8, 7, 123, 0, 0, 0]) and len(p) == W.FIXED_PREFIX_SIZE
    pv = E.prefix(W.RLE, W.STRING, 1_000_000, varint_rows=True)
    assert pv == bytes([1, 12, 0xC0, 0x84, 0x3D])  # 1,000,000 as LEB128


def test_chunk_header_layout():
    """ChunkHeaderTest / ChunkHeader.h:25-47: 4-byte length + 1-byte compression type = 5 bytes."""
    c = E.chunk(b"abc")
    assert c == b"\x03\x00\x00\x00\x00abc" and len(c) == W.CHUNK_HEADER_SIZE + 3


def test_fbw_header_bit_width_byte():
    """FixedBitWidthEncodingTest.cpp:140-188: {0..52} -> bit-width byte 8 by default, 6 with exact bits."""
    v = np.arange(53, dtype=np.uint32)
    d = E.encode(E.FixedBitWidth(), v, W.UINT32)
    x = E.encode(E.FixedBitWidth(exact_bits=True), v, W.UINT32)
    assert d[6 + 1 + 4] == 8
    assert x[6 + 1 + 4] == 6
    # FixedBitArray::bufferSize: ceil(n*w/8) + 7 bytes of slop
    assert len(d) == 6 + 1 + 4 + 1 + 53 + 7
    assert len(x) == 6 + 1 + 4 + 1 + (53 * 6 + 7) // 8 + 7


def test_encoder_reproduces_hand_assembled_bytes():
    """The numpy encoder and the independent hand assembly must agree byte for byte."""
    by_name = {c["name"]: c for c in CASES}
    r53 = np.arange(53, dtype=np.uint32)
    assert E.encode(E.FixedBitWidth(), r53, W.UINT32) == by_name["fbw_u32_0to52_byte_rounded"]["enc"]
    assert E.encode(E.FixedBitWidth(exact_bits=True), r53, W.UINT32) == by_name["fbw_u32_0to52_exact_6bits"]["enc"]
    assert E.encode(E.FixedBitWidth(), r53, W.UINT32, varint_rows=True) == by_name["fbw_u32_0to52_varint_prefix"]["enc"]
    quad = np.array([(i + i * i) % 1024 for i in range(10000)], dtype=np.uint32)
    assert E.encode(E.FixedBitWidth(bit_width=10), quad, W.UINT32) == by_name["fba_10bit_quadratic"]["enc"]
    tri = np.array([1, 2, 2, 3, 3, 3, 4, 4, 4, 4], dtype=np.int32)
    assert E.encode(E.RLE(lengths=E.Trivial(), values=E.Trivial()), tri, W.INT32) == by_name["rle_i32_triangle"]["enc"]
    dv = np.array([7, 7, 3, 9, 3, 7, 9, 9, 1], dtype=np.int32)
    assert E.encode(E.Dictionary(alphabet=E.Trivial(), indices=E.Trivial()), dv, W.INT32) == by_name["dictionary_i32"]["enc"]
    sb = np.zeros(40, dtype=bool)
    sb[[3, 17, 18, 39]] = True
    assert E.encode(E.SparseBool(indices=E.Trivial()), sb, W.BOOL) == by_name["sparse_bool_few_set"]["enc"]
    assert E.encode(E.SparseBool(indices=E.Trivial()), ~sb, W.BOOL) == by_name["sparse_bool_few_unset"]["enc"]
    vv = np.array([0, 1, 127, 128, 300, 16383, 16384, 2**21, 2**28 - 1, 2**28, 2**32 - 1], dtype=np.uint32)
    assert E.encode(E.Varint(), vv, W.UINT32) == by_name["varint_u32"]["enc"]
    mask = np.array([True, False, True, True, False, False, True, True, True, False])
    nn = np.array([5, 6, 5, 7, 6, 5], dtype=np.int32)
    got = E.encode_nullable(E.Nullable(data=E.Dictionary(alphabet=E.Trivial(), indices=E.Trivial()), nulls=E.Trivial()),
                            nn, mask, W.INT32)
    assert got == by_name["nullable_dictionary_i32"]["enc"]
    mc2 = np.array([1, 1, 2, 1, 3, 1, 1, 9, 1, 1, 1], dtype=np.int64)
    got = E.encode(E.MainlyConstant(is_common=E.Trivial(), other=E.Dictionary(alphabet=E.Trivial(), indices=E.Trivial())),
                   mc2, W.INT64)
    assert got == by_name["mainly_constant_dictionary_others"]["enc"]
    strs = E.StringColumn.from_list([b"", b"a", b"nimble", b"\x00\xff", b"cascading stream encodings", b""])
    assert E.encode(E.Trivial(lengths=E.Trivial()), strs, W.STRING) == by_name["trivial_string"]["enc"]
