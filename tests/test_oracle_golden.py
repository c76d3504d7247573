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


# ---- vectors whose bytes one of the reference's own tests assembles ----------------
def _doc():
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.json")) as f:
        return json.load(f)


def test_every_case_names_its_origin():
    doc = _doc()
    origins = [c["origin"] for c in doc["cases"]]
    assert set(origins) == {"reference-held", "reference-frame", "self-assembled"}
    framed = [c["name"] for c in doc["cases"] if c["origin"] == "reference-frame"]
    assert len(framed) >= 19 and {"ref_mainly_constant_sparse_is_common", "ref_mainly_constant_sparse_is_common_varint_rows",
                                  "ref_nullable_string_dictionary_50", "ref_nullable_string_dictionary_masked_value"} <= set(framed)
    held = [c["name"] for c in doc["cases"] if c["origin"] == "reference-held"]
    assert len(held) >= 10 and all(c["source"] for c in doc["cases"])
    # the reference-held families: Trivial<u32> / Constant<u32> of EncodingUtilitiesTest.cpp, encodeByteRounded in both prefix modes
    assert {"ref_trivial_u32_10_20_30", "ref_constant_u32_42x100", "fbw_u32_0to52_byte_rounded", "fbw_u32_0to52_varint_prefix"} <= set(held)


def test_reference_held_prefixes_and_chunk_headers():
    from nimble_amd import capi

    doc = _doc()
    for p in doc["prefixes"]:  # EncodingPrefix::serialize (EncodingPrefixTest.cpp:42-83)
        assert E.prefix(p["encoding_type"], p["data_type"], p["rows"], p["varint_rows"]).hex() == p["hex"], p["source"]
    for h in doc["chunk_headers"]:  # writeChunkHeader / readChunkHeader (ChunkHeaderTest.cpp:28-62)
        raw = bytes.fromhex(h["hex"])
        assert len(raw) == doc["chunk_header_size"] == W.CHUNK_HEADER_SIZE
        body = bytes(h["length"])
        comp = np.zeros(1, np.uint8)
        offs, sizes = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        import ctypes as C
        n = C.c_uint32(0)
        stream = capi.as_u8(raw + body)
        capi.check(capi.lib().nbl_split_stream_ex(stream.ctypes.data, stream.size, offs.ctypes.data, sizes.ctypes.data,
                                                  comp.ctypes.data, 1, C.byref(n)))
        assert (n.value, int(offs[0]), int(sizes[0]), int(comp[0])) == (1, 5, h["length"], h["compression"]), h["source"]
        if h["compression"] == 0:
            assert E.chunk(body) == raw + body


def test_reference_held_varint_sizes():
    for e in _doc()["varint_sizes"]["entries"]:  # VarintTests.cpp:64-76,95-103
        assert len(E.varint_bytes(e["value"])) == e["size"]


def test_reference_held_block_bit_packing_header_walk():
    """The BlockBitPacking header the reference's tool test assembles (no packed
    data; bit widths carried as a uint32 stream, which traverseEncodings does not
    mind): the host walk that the decompression pre-pass shares with the planner
    must find the three varint-sized children in the reference's order and
    reproduce the stream byte for byte."""
    from nimble_amd import capi

    for h in _doc()["headers"]:
        stream = bytes.fromhex(h["hex"])
        assert capi.inflate_chunk(stream) == stream
        # the children sit where the reference's builder put them, in Baselines, BitWidths, DataOffsets order
        pos = 6 + 1 + 1 + 1  # prefix, compression byte, varint(64), varint(2)
        for child_hex in h["children_hex"]:
            child = bytes.fromhex(child_hex)
            assert stream[pos] == len(child) and stream[pos + 1: pos + 1 + len(child)] == child
            pos += 1 + len(child)
        assert stream[pos] == h["first_block_rows"] and pos + 1 == len(stream)
        # damaging a child's size byte must be caught by the same walk
        bad = bytearray(stream)
        bad[9] = 200
        with pytest.raises(capi.NimbleCorruptError):
            capi.inflate_chunk(bytes(bad))


# ---- frames and answers held by the reference's tests ------------------------------
def _values_list(case, vals):
    if case["data_type"] == W.STRING:
        lengths, blob = vals
        out, pos = [], 0
        for n in lengths.tolist():
            out.append(bytes(blob[pos:pos + n]))
            pos += n
        return out
    return [int(v) for v in np.asarray(vals).astype(np.uint64)]


@pytest.mark.parametrize("case", [c for c in CASES if c["cursor"]], ids=lambda c: c["name"])
def test_oracle_cursor_reads_of_reference_frames(oracle, case):
    """reset, skip(n), materialize(m) as MainlyConstantEncodingTest.cpp:456-462 asserts it."""
    e = oracle.Encoding(case["enc"], case["varint_rows"])
    e.materialize(e.row_count)
    for step in case["cursor"]:
        e.reset()
        e.skip(step["skip"])
        got = e.materialize(step["materialize"])
        assert [int(v) for v in got.astype(np.int64)] == step["expected"], case["name"]


def test_encoder_computes_the_reference_runs(oracle):
    """RLETests.cpp:42-169: data -> (run lengths, run values); an RLE of those runs decodes to the data."""
    import struct

    for e in _doc()["rle_runs"]["entries"]:
        dt = e["data_type"]
        npt = W.physical_dtype(dt)
        data = np.array(e["data"], dtype=np.int64).astype(npt)
        lens = E.encode(E.Trivial(), np.array(e["lengths"], dtype=np.uint32), W.UINT32)
        vals = E.encode(E.Trivial(), np.array(e["values"], dtype=np.int64).astype(npt), dt)
        framed = E.prefix(W.RLE, dt, len(e["data"])) + struct.pack("<I", len(lens)) + lens + vals
        assert E.encode(E.RLE(lengths=E.Trivial(), values=E.Trivial()), data, dt) == framed, e["source"]
        got = oracle.Encoding(framed).materialize(len(e["data"]))
        assert np.array_equal(got, data), e["source"]


def test_sparse_bool_polarity_and_positions(oracle):
    """SparseBoolEncodingTest.cpp:145-216: which value is listed, and where."""
    for e in _doc()["sparse_bool_facts"]["entries"]:
        bools = np.array(e["values"], dtype=bool)
        enc = E.encode(E.SparseBool(indices=E.Trivial()), bools, W.BOOL)
        assert enc[6] == e["sparse_value"], e["source"]
        idx = oracle.Encoding(enc[7:]).materialize(len(e["positions"]) + 1)
        assert idx.tolist() == e["positions"] + [len(e["values"])], e["source"]  # listed positions, then the row-count sentinel
        assert np.array_equal(oracle.Encoding(enc).materialize(bools.size).astype(bool), bools)
