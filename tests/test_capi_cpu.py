"""CPU: the C-ABI library loads, exports every symbol include/nimble_amd.h
declares, plans batches (header walk only, no GPU call) and reports malformed
or unsupported bytes the way the reference's constructors do."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

from nimble_amd import capi
from nimble_amd import encoder as E
from nimble_amd import wire as W
from nimble_amd.decoder import Plan

from golden_util import load_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "nimble_amd.h")).read()
    # function declarations: a return type, then the name, then "("
    declared = set(re.findall(r"^(?:nbl_status|void|const char\*)\s+(nbl_[a-z_]+)\s*\(", header, flags=re.M))
    assert declared, "no declarations found"
    lib = capi.lib()
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} is declared in include/nimble_amd.h but not exported"
    assert declared == set(capi.EXPORTED_SYMBOLS)
    assert lib.nbl_build_arch() == b"gfx950"
    from nimble_amd import build as nb

    assert lib.nbl_build_source_id().decode() == nb.source_id(), "libnimble_amd.so was built from other sources: rebuild"


def test_no_oracle_in_product_path():
    """The product package must not import, link or load anything under oracle/."""
    pkg = os.path.join(ROOT, "nimble_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"(import|from|include|CDLL|dlopen)[^\n]*oracle", text), f"{f} reaches into oracle/"


def test_plan_layout_of_a_batch():
    rng = np.random.default_rng(0)
    v = rng.integers(0, 1000, 100_000).astype(np.int64)
    runs = np.repeat(v[:3000], 30)
    mask = rng.random(5000) < 0.8
    encs = [
        E.encode(E.FixedBitWidth(), v, W.INT64),
        E.encode(E.RLE(), runs, W.INT64),
        E.encode_nullable(E.Nullable(data=E.FixedBitWidth(), nulls=E.SparseBool()), v[: int(mask.sum())], mask, W.INT64),
        E.encode(E.Trivial(), E.StringColumn.from_list([b"ab", b"", b"cde"]), W.STRING),
        E.encode(E.Trivial(), rng.random(77) < 0.5, W.BOOL),
    ]
    p = Plan.from_encodings(encs)
    info = p.info
    assert info.n_chunks == 5 and info.n_nodes == 1 + 3 + 4 + 2 + 1
    assert info.encoded_bytes == sum(len(e) for e in encs)
    assert info.arena_bytes >= info.encoded_bytes + 16
    assert info.total_rows == 100_000 + 90_000 + 5000 + 3 + 77
    lay = p.layouts
    assert (lay[0].rows, lay[0].elem_bytes, lay[0].nullable, lay[0].values_bytes) == (100_000, 8, 0, 800_000)
    assert lay[2].nullable == 1 and lay[2].non_null_rows == int(mask.sum()) and lay[2].nulls_bytes == (5000 + 63) // 64 * 8
    assert lay[3].data_type == W.STRING and lay[3].elem_bytes == 8 and lay[3].string_bytes >= 5
    assert lay[4].data_type == W.BOOL and lay[4].values_bytes == 77
    # outputs do not overlap and stay inside out_bytes
    spans = []
    for l in lay:
        for off, size in ((l.values_offset, l.values_bytes), (l.nulls_offset, l.nulls_bytes), (l.string_offset, l.string_bytes)):
            if size:
                spans.append((off, off + size))
    spans.sort()
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0
    assert spans[-1][1] <= info.out_bytes
    launches = p.launches()
    assert len(launches) == info.n_launches
    assert sum(l[4] for l in launches) == info.encoded_bytes + info.decoded_bytes
    levels = [l[3] for l in launches]
    assert levels == sorted(levels)
    p.close()


def test_every_golden_vector_plans():
    for case in load_cases():
        p = Plan.from_encodings([case["enc"]], case["varint_rows"])
        assert p.layouts[0].data_type == case["data_type"], case["name"]
        exp = case["expected"]
        n = exp[0].size if case["data_type"] == W.STRING else exp.size
        assert p.layouts[0].rows == n
        p.close()


def _expect(exc, enc, match=None):
    with pytest.raises(exc, match=match):
        Plan.from_encodings([enc])


def test_malformed_and_unsupported_inputs():
    ok = E.encode(E.Trivial(), np.arange(4, dtype=np.int32), W.INT32)
    _expect(capi.NimbleCorruptError, bytes([99]) + ok[1:], "invalid EncodingType")        # EncodingFactory.cpp:158-162
    _expect(capi.NimbleCorruptError, ok + b"\x00", "Unexpected trivial encoding end")       # TrivialEncoding.h:273-282
    _expect(capi.NimbleCorruptError, ok[:-1], "Unexpected trivial encoding end")
    _expect(capi.NimbleCorruptError, ok[:4], "prefix")
    _expect(capi.NimbleUnsupportedError, ok[:6] + b"\x01" + ok[7:], "compressed")           # Zstd leaf payload
    _expect(capi.NimbleUnsupportedError, bytes([8]) + ok[1:], "not decoded on the GPU")     # Delta
    bad = bytearray(E.encode(E.SparseBool(), np.zeros(9, dtype=bool), W.BOOL))
    bad[1] = W.INT32
    _expect(capi.NimbleCorruptError, bytes(bad), "non-bool")                                # EncodingFactory.cpp:90-97
    c = E.encode(E.Constant(), np.full(5, 3, dtype=np.int64), W.INT64)
    _expect(capi.NimbleCorruptError, c + b"\x00", "Unexpected constant encoding end")       # ConstantEncoding.h:299-301
    rb = E.encode(E.RLE(), np.array([1, 1, 0], dtype=bool), W.BOOL)
    _expect(capi.NimbleCorruptError, rb + b"\x00", "Unexpected run length encoding end")    # RLEEncoding.cpp:30-35
    f = bytearray(E.encode(E.FixedBitWidth(), np.arange(100, dtype=np.int32), W.INT32))
    f[6 + 1 + 4] = 40
    _expect(capi.NimbleCorruptError, bytes(f), "bit width")                                 # FixedBitArray.cpp:311-313
    _expect(capi.NimbleCorruptError, E.encode(E.FixedBitWidth(), np.arange(100, dtype=np.int32), W.INT32)[:-20], "truncated")
    # Nullable whose nulls child has another row count (NullableEncoding.h:169-170)
    mask = np.array([True, False, True])
    n = bytearray(E.encode_nullable(E.Nullable(), np.array([1, 2], dtype=np.int32), mask, W.INT32))
    n[2:6] = struct.pack("<I", 4)
    _expect(capi.NimbleCorruptError, bytes(n), "Nulls count mismatch")
    # child size field pointing past the parent
    r = bytearray(E.encode(E.RLE(), np.array([5, 5, 6], dtype=np.int32), W.INT32))
    r[6:10] = struct.pack("<I", 10_000)
    _expect(capi.NimbleCorruptError, bytes(r), "beyond end")


def test_data_types_an_encoding_cannot_carry_are_refused_in_the_reference_words():
    """The factory's dispatch by data type (encodings/common/EncodingFactory.cpp:71-160 with the macros of
    EncodingTypeDispatch.h:24-68): which encodings take which data types, and what the refusal says."""
    def bare(encoding_type, data_type):  # a prefix alone: the data type is looked at before anything behind it
        return bytes([encoding_type, data_type]) + struct.pack("<I", 1) + bytes(40)
    cases = [
        (W.FIXED_BIT_WIDTH, W.STRING, "Trying to deserialize a non-numeric stream for a numeric data type String."),
        (W.FIXED_BIT_WIDTH, W.BOOL, "Trying to deserialize a non-numeric stream for a numeric data type Bool."),
        (W.BLOCK_BIT_PACKING, W.STRING, "Trying to deserialize a non-numeric stream for a numeric data type String."),
        (W.VARINT, W.INT16, "Trying to deserialize a varint stream for an incompatible data type Int16."),
        (W.VARINT, W.STRING, "Trying to deserialize a varint stream for an incompatible data type String."),
        (W.DICTIONARY, W.BOOL, "Trying to deserialize a non-bool stream for the bool data type Bool."),
        (W.MAINLY_CONSTANT, W.BOOL, "Trying to deserialize a non-bool stream for the bool data type Bool."),
        (W.FOR, W.FLOAT, "Trying to deserialize an integer stream for an incompatible data type Float."),
        (W.SPARSE_BOOL, W.INT32, "Trying to deserialize a SparseBoolEncoding with a non-bool data type."),
    ]
    for et, dt, message in cases:
        with pytest.raises(capi.NimbleCorruptError) as ei:
            Plan.from_encodings([bare(et, dt)])
        assert message in ei.value.message, (et, dt, ei.value.message)


def test_split_stream():
    e1 = E.encode(E.Trivial(), np.arange(5, dtype=np.int32), W.INT32)
    e2 = E.encode(E.FixedBitWidth(), np.arange(9, dtype=np.int64), W.INT64)
    s = E.stream([e1, e2])
    assert capi.split_stream(s) == [(5, len(e1)), (5 + len(e1) + 5, len(e2))]
    assert capi.split_stream(b"") == []
    with pytest.raises(capi.NimbleCorruptError, match="Read beyond end of stream"):
        capi.split_stream(s[:-1])
    with pytest.raises(capi.NimbleUnsupportedError):
        capi.split_stream(E.chunk(e1, compression=W.ZSTD))
    with pytest.raises(capi.NimbleCorruptError, match="Unexpected stream compression type"):
        capi.split_stream(E.chunk(e1, compression=7))


def test_decode_without_gpu_fails_loudly():
    """No CPU fallback: without a device the decode entry point must report an error."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from nimble_amd.decoder import BatchDecoder

    p = Plan.from_encodings([E.encode(E.Trivial(), np.arange(4, dtype=np.int32), W.INT32)])
    with pytest.raises(RuntimeError):
        BatchDecoder(p)
    p.close()


def test_fused_and_per_operation_plans_agree_on_layout_and_bytes():
    """The two launch shapes of a plan (one launch per cascade level / one per
    operation) describe the same outputs; launches are ordered by level, the
    fused plan has exactly one launch per level and both account the same
    algorithmic bytes."""
    from nimble_amd import workloads as WL

    chunks = WL.mixed_table_stripe(stripe=1, rows=3000)
    buf, offs = WL.pack_chunks(chunks)
    only_fused = dict(four_wave_scatter=True, no_lds_dictionary=True)  # no kernel that is a launch of its own
    fused, per_op = Plan(buf, offs, **only_fused), Plan(buf, offs, per_kernel=True, **only_fused)
    assert fused.info.out_bytes == per_op.info.out_bytes and fused.info.work_bytes == per_op.info.work_bytes
    for a, b in zip(fused.layouts, per_op.layouts):
        assert bytes(a) == bytes(b)
    lf, lp = fused.launches(), per_op.launches()
    levels = [l[3] for l in lf]
    assert levels == sorted(set(levels)) and len(lp) > len(lf)
    assert {l[0] for l in lf} <= {"k_level<false>", "k_level<true>"} and lf[-1][0] == "k_level<true>"
    assert sum(l[4] for l in lf) == sum(l[4] for l in lp) == fused.info.encoded_bytes + fused.info.decoded_bytes
    assert sum(l[2] for l in lf) == sum(l[2] for l in lp)  # same tiles, grouped differently
    assert [l[3] for l in lp] == sorted(l[3] for l in lp)
    # Rank scatters of sparse selections can go to the one-wave kernel, a launch
    # of its own behind the fused launch of its level (by default only where a
    # level has tens of thousands of such tiles): same outputs, same tiles and
    # bytes, one fused launch per level as before.
    small = Plan(buf, offs, no_lds_dictionary=True)
    assert {l[0] for l in small.launches()} <= {"k_level<false>", "k_level<true>"}
    small.close()
    dflt = Plan(buf, offs, wave_scatter_min_tiles=1, no_lds_dictionary=True)
    assert dflt.info.out_bytes == fused.info.out_bytes
    for a, b in zip(dflt.layouts, fused.layouts):
        assert bytes(a) == bytes(b)
    ld = dflt.launches()
    names = {l[0] for l in ld}
    assert names <= {"k_level<false>", "k_level<true>", "k_select_scatter_wave<4>", "k_select_scatter_wave<8>"}
    assert names & {"k_select_scatter_wave<4>", "k_select_scatter_wave<8>"}  # the table has sparse nullable columns
    assert [l[3] for l in ld] == sorted(l[3] for l in ld)
    fused_levels = [l[3] for l in ld if l[0].startswith("k_level")]
    assert fused_levels == sorted(set(fused_levels))
    assert sum(l[4] for l in ld) == sum(l[4] for l in lf) and sum(l[2] for l in ld) == sum(l[2] for l in lf)
    fused.close(); per_op.close(); dflt.close()


def test_leaf_only_chunks_fill_the_preparatory_levels():
    """Chunks that are a single unpack / copy depend on nothing: their tiles are
    dealt over the levels that the deeper cascades need, so no level is left
    with only a handful of small tiles."""
    rng = np.random.default_rng(3)
    n = 400_000
    leaf = [E.encode(E.FixedBitWidth(exact_bits=True), rng.integers(0, 1 << 9, n).astype(np.int64), W.INT64) for _ in range(6)]
    deep = [E.encode(E.RLE(), np.repeat(rng.integers(0, 50, n // 20), 20).astype(np.int64), W.INT64)]
    p = Plan.from_encodings(leaf + deep)
    launches = p.launches()
    assert len(launches) == 2  # run ends (a chained scan over the tiles of runs), expand
    tiles = [l[2] for l in launches]
    assert min(tiles) > 100 and max(tiles) < 3 * min(tiles), tiles
    only_deep = Plan.from_encodings(deep)
    assert [l[2] for l in only_deep.launches()][:1] == [10]  # 20,000 runs: ten tiles of run ends
    p.close(); only_deep.close()


def test_rank_counting_is_placed_at_the_bitmap_level():
    """The rank partials of a Nullable read the null bitmap only: under a values
    child that has steps of its own (RLE: run ends) they run alongside them.
    Both are chained scans, one step each.  The rank scatter expands the runs
    itself, so the RLE has no expand step and the scatter follows directly."""
    rng = np.random.default_rng(5)
    n = 300_000
    mask = rng.random(n) < 0.7
    v = np.repeat(rng.integers(0, 90, n // 10 + 1), 10)[: int(mask.sum())].astype(np.int64)
    enc = E.encode_nullable(E.Nullable(data=E.RLE(), nulls=E.Trivial()), v, mask, W.INT64)
    p = Plan.from_encodings([enc], per_kernel=True)
    level = {}
    for name, _, _, lvl, _ in p.launches():
        level.setdefault(name, lvl)
    assert level["k_reduce_popc"] == level["k_rle_ends"] == 0
    assert "k_rle_expand<8>" not in level and level["k_select_scatter<8>"] == 1
    assert set(level) == {"k_reduce_popc", "k_rle_ends", "k_select_scatter<8>"}
    p.close()
    fused = Plan.from_encodings([enc])
    assert int(fused.info.n_launches) == 2
    fused.close()


def test_values_children_resolved_inside_the_rank_scatter():
    """Nullable / MainlyConstant over a Dictionary, an RLE or a MainlyConstant of
    4- or 8-byte rows: the parent's rank scatter gathers / expands / selects for
    the child, which therefore has no step writing a column of its own.  Narrow
    rows and bool keep the two-pass form."""
    rng = np.random.default_rng(6)
    n = 50_000
    mask = rng.random(n) < 0.6
    nn = int(mask.sum())

    def names(enc):
        # the one-wave kernel takes the scatters that stage few values; what a scatter resolves does not depend on it
        p = Plan.from_encodings([enc], per_kernel=True, wave_scatter_min_tiles=1)
        out = [l[0].replace("k_select_scatter_wave", "k_select_scatter") for l in p.launches()]
        p.close()
        return out

    runs = np.repeat(rng.integers(0, 90, nn // 10 + 1), 10)[:nn]
    got = names(E.encode_nullable(E.Nullable(data=E.RLE(), nulls=E.Trivial()), runs.astype(np.int32), mask, W.INT32))
    assert "k_rle_ends" in got and not any(k.startswith("k_rle_expand") for k in got) and got.count("k_select_scatter<4>") == 1
    got = names(E.encode_nullable(E.Nullable(data=E.RLE(), nulls=E.Trivial()), runs.astype(np.int16), mask, W.INT16))
    assert "k_rle_expand<2>" in got and "k_select_scatter<2>" in got  # narrow rows: two passes
    mc = np.where(rng.random(nn) < 0.9, 7, rng.integers(100, 1000, nn)).astype(np.int64)
    got = names(E.encode_nullable(E.Nullable(data=E.MainlyConstant(), nulls=E.SparseBool()), mc, mask, W.INT64))
    # both nodes count ranks, one scatters; SparseBool selections write them scanned while they build their bitmaps
    assert got == ["k_sparse_bits", "k_select_scatter<8>"]
    got = names(E.encode_nullable(E.Nullable(data=E.MainlyConstant(is_common=E.Trivial()), nulls=E.SparseBool()), mc, mask, W.INT64))
    assert got.count("k_select_scatter<8>") == 1 and "k_reduce_popc" in got  # a plain bitmap is counted by a pass of its own
    # nesting goes one deep: of three MainlyConstants the middle one is resolved by the outermost, the innermost scatters itself
    v = np.where(rng.random(n) < 0.8, 1, np.where(rng.random(n) < 0.7, 2, rng.integers(3, 9, n))).astype(np.int32)
    three = E.encode(E.MainlyConstant(other=E.MainlyConstant(other=E.MainlyConstant())), v, W.INT32)
    assert names(three).count("k_select_scatter<4>") == 2


def test_bool_rle_root_expands_to_bytes_without_a_bitmap_step():
    """An RLE<bool> that is a chunk's root writes one byte per row from its
    expansion; under a Nullable it stays a bitmap for the select step."""
    b = np.repeat(np.arange(500) & 1, 77).astype(bool)
    p = Plan.from_encodings([E.encode(E.RLE(), b, W.BOOL)], per_kernel=True)
    assert [l[0] for l in p.launches()] == ["k_rle_ends", "k_rle_expand_bool"]  # 500 runs: one tile
    p.close()
    many = np.repeat(np.arange(5000) & 1, 7).astype(bool)                      # 5,000 runs: three tiles, chained
    p = Plan.from_encodings([E.encode(E.RLE(), many, W.BOOL)], per_kernel=True)
    assert [(l[0], l[2]) for l in p.launches()][:1] == [("k_rle_ends", 3)] and len(p.launches()) == 2
    p.close()
    mask = np.random.default_rng(9).random(60_000) < 0.5
    enc = E.encode_nullable(E.Nullable(data=E.RLE(), nulls=E.Trivial()), b[: int(mask.sum())], mask, W.BOOL)
    p = Plan.from_encodings([enc], per_kernel=True)
    assert [l[0] for l in p.launches()][-2:] == ["k_select_bits", "k_bits_to_bytes"]
    p.close()


def test_dense_sparse_bool_gets_smaller_bitmap_tiles():
    """A SparseBool whose listed positions are more than 1/32 of its rows is cut
    into 16384-row tiles (a tile's positions are loaded in rounds), a sparser
    one into 65536-row tiles (the search for a tile's range dominates)."""
    n = 200_000
    rng = np.random.default_rng(8)

    def sparse_tiles(frac):
        b = rng.random(n) < frac
        p = Plan.from_encodings([E.encode(E.SparseBool(), b, W.BOOL)], per_kernel=True)
        t = [l[2] for l in p.launches() if l[0] == "k_sparse_bits"]
        p.close()
        return t

    assert sparse_tiles(0.01) == [-(-n // 65536)]
    assert sparse_tiles(0.2) == [-(-n // 16384)]


def _rle_bytes(dtype, rows, lengths, values, varint_rows=False):
    lens = E.encode(E.Trivial(), np.asarray(lengths, dtype=np.uint32), W.UINT32, varint_rows)
    vals = E.encode(E.Trivial(), np.asarray(values).astype(W.physical_dtype(dtype)), dtype, varint_rows)
    return E.prefix(W.RLE, dtype, rows, varint_rows) + struct.pack("<I", len(lens)) + lens + vals


def empty_rle_cases(varint_rows=False):
    """An RLE node whose row count is 0 while its run-length child is not empty:
    as a root, as the values child of an all-null Nullable, and as a bool column."""
    root = _rle_bytes(W.INT64, 0, [5], [7], varint_rows)
    nulls = E.encode(E.Trivial(), np.zeros(3000, dtype=bool), W.BOOL, varint_rows)
    data = _rle_bytes(W.INT32, 0, [5, 9], [7, 8], varint_rows)
    under = E.prefix(W.NULLABLE, W.INT32, 3000, varint_rows) + struct.pack("<I", len(data)) + data + nulls
    lens = E.encode(E.Trivial(), np.asarray([4, 4], dtype=np.uint32), W.UINT32, varint_rows)
    boolean = E.prefix(W.RLE, W.BOOL, 0, varint_rows) + struct.pack("<I", len(lens)) + lens + b"\x01"
    return [root, under, boolean]


@pytest.mark.parametrize("varint_rows", [False, True])
def test_rle_without_rows_schedules_no_run_kernels(varint_rows):
    """rowCount 0 over non-empty run lengths (one damaged byte away under varint
    row counts): nothing is expanded, so no run-end / expand step may be planned
    -- the run-end kernel's per-tile table has no entry for such a node."""
    for per_kernel in (False, True):
        p = Plan.from_encodings(empty_rle_cases(varint_rows), varint_rows, per_kernel=per_kernel)
        names = {l[0] for l in p.launches()}
        assert not any(n.startswith("k_rle") for n in names), names
        assert [l.rows for l in p.layouts] == [0, 3000, 0]
        p.close()


def test_chunk_offsets_that_overflow_are_refused():
    enc = E.encode(E.Trivial(), np.arange(4, dtype=np.int32), W.INT32)
    buf = capi.as_u8(enc)
    offs, sizes = np.array([2**64 - 8], dtype=np.uint64), np.array([len(enc)], dtype=np.uint64)
    plan = C.c_void_p()
    st = capi.lib().nbl_plan_batch(buf.ctypes.data, offs.ctypes.data, sizes.ctypes.data, 1, None, C.byref(plan))
    assert st == capi.NBL_ERR_ARGUMENT and b"overflows" in capi.lib().nbl_last_error()


def test_one_wave_scatter_needs_a_level_full_of_rank_scatters():
    """The one-wave kernel is launched for a level whose rank scatters (of either
    kind) reach wave_scatter_min_tiles and whose eligible tiles of one element
    size are at least an eighth of that; below either bound they stay four-wave
    tiles inside the fused launch."""
    rng = np.random.default_rng(3)
    n = 60_000  # 15 tiles of 4096 float rows per column
    cols = []
    for c in range(8):
        mask = rng.random(n) < 0.05
        vals = rng.integers(0, 1 << 30, int(mask.sum())).astype(np.float32)
        cols.append(E.encode_nullable(E.Nullable(data=E.Trivial(), nulls=E.SparseBool()), vals, mask, W.FLOAT))
    dense_mask = rng.random(n) < 0.9   # a dense selection: never on the one-wave kernel, but it counts for the level
    dense = E.encode_nullable(E.Nullable(data=E.Trivial(), nulls=E.Trivial()),
                              rng.integers(0, 1 << 30, int(dense_mask.sum())).astype(np.float32), dense_mask, W.FLOAT)

    def wave_tiles(encs, min_tiles):
        p = Plan.from_encodings(encs, wave_scatter_min_tiles=min_tiles)
        t = sum(l[2] for l in p.launches() if l[0].startswith("k_select_scatter_wave"))
        p.close()
        return t

    assert wave_tiles(cols, 120) == 8 * 15          # 120 rank-scatter tiles in the level: exactly the bound
    assert wave_tiles(cols, 121) == 0
    assert wave_tiles(cols + [dense], 135) == 8 * 15   # the dense column's 15 tiles count for the level
    assert wave_tiles(cols[:1] + [dense] * 9, 150) == 0   # 15 eligible tiles < 150 / 8
    assert wave_tiles(cols[:2] + [dense] * 8, 150) == 30
