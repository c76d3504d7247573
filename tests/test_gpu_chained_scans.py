"""GPU parity of the chained scans -- RLE run ends (RLEEncoding.h:289-301),
string views (TrivialEncoding.cpp:151-164), varint ordinals (VarintEncoding.h:119-140),
ranks of a bitmap-selected Nullable / MainlyConstant (NullableEncoding.h:239-306) --
whose tiles look back at the tiles in front of them inside one launch: nodes of
one tile (no chain), a few tiles, more tiles than a look-back window (64) and
more than the chip holds at once.  All against the CPU oracle, and decoded
twice by the same plan (look-back words carry the number of the decode; totals,
the zero-length-run flag and the ticket counters are rewritten by every decode)."""
import struct

import numpy as np
import pytest

from nimble_amd import capi
from nimble_amd import encoder as E
from nimble_amd import wire as W
from nimble_amd.decoder import BatchDecoder, Plan

from util import assert_chunk_equal, check_parity, oracle_chunk

pytestmark = pytest.mark.gpu


def _rle(dtype, rows, lengths, values, lengths_spec=None):
    lens = E.encode(lengths_spec or E.FixedBitWidth(), np.asarray(lengths, dtype=np.uint32), W.UINT32)
    vals = E.encode(E.Trivial(), np.asarray(values).astype(W.physical_dtype(dtype)), dtype)
    return E.prefix(W.RLE, dtype, rows) + struct.pack("<I", len(lens)) + lens + vals


def _names(encs):
    p = Plan.from_encodings(encs, per_kernel=True)
    out = [l[0] for l in p.launches()]
    p.close()
    return out


def _decode_twice(oracle, encs):
    plan = Plan.from_encodings(encs)
    dec = BatchDecoder(plan)
    dec.stage()
    for _ in range(2):
        dec.decode()
        dec.status()
        for enc, res in zip(encs, dec.fetch()):
            assert_chunk_equal(res, oracle_chunk(oracle, enc), "second decode of the same plan")
    plan.close()


def test_runs_over_several_tiles_with_zero_length_runs(gpu, oracle):
    """2,049, 5,000 and 70,000 runs (2, 3 and 35 tiles), some of length zero: the
    flag travels down the chain with the sums and the last tile writes it."""
    rng = np.random.default_rng(51)
    encs = []
    for runs in (2049, 5000, 70_000):
        lengths = rng.integers(0, 6, runs)
        lengths[rng.random(runs) < 0.05] = 0
        lengths[-1] = 3
        rows = int(lengths.sum())
        values = rng.integers(0, 1 << 20, runs)
        encs.append(_rle(W.INT32, rows, lengths, values))
        encs.append(_rle(W.INT64, rows, lengths, values, E.Trivial()))
        mask = rng.random(rows * 2) < 0.5
        mask[:] = False
        mask[rng.choice(rows * 2, rows, replace=False)] = True
        data = _rle(W.INT64, rows, lengths, values)
        nulls = E.encode(E.Trivial(), mask, W.BOOL)
        encs.append(E.prefix(W.NULLABLE, W.INT64, rows * 2) + struct.pack("<I", len(data)) + data + nulls)
    assert _names(encs[:1]) == ["k_rle_ends", "k_rle_expand<4>"]
    check_parity(oracle, encs)
    _decode_twice(oracle, encs)


def test_one_tile_of_runs_writes_its_flag_word_itself(gpu, oracle):
    rng = np.random.default_rng(52)
    with_zero = _rle(W.INT32, 12, [3, 0, 0, 4, 0, 5, 0], [10, 11, 12, 13, 14, 15, 16])
    lengths = rng.integers(1, 9, 2048)
    without = _rle(W.INT32, int(lengths.sum()), lengths, rng.integers(0, 99, 2048))
    assert _names([with_zero, without])[0] == "k_rle_ends"
    check_parity(oracle, [with_zero, without])
    # the same plan decodes a column with zero-length runs, then must not remember them
    _decode_twice(oracle, [with_zero, without])
    _decode_twice(oracle, [without, with_zero])


def test_run_lengths_that_do_not_add_up_are_reported_by_the_last_tile(gpu):
    rng = np.random.default_rng(53)
    for runs in (100, 9000):
        lengths = rng.integers(1, 9, runs)
        enc = _rle(W.INT64, int(lengths.sum()) + 5, lengths, np.arange(runs))
        plan = Plan.from_encodings([enc])
        dec = BatchDecoder(plan)
        dec.stage()
        dec.decode()
        with pytest.raises(capi.NimbleError, match="run lengths") as ei:
            dec.status()
        assert ei.value.status == capi.NBL_ERR_DEVICE_DATA
        plan.close()


def test_strings_over_several_tiles(gpu, oracle):
    rng = np.random.default_rng(54)
    encs = []
    for n in (1, 2048, 2049, 10_000, 150_000):
        lens = rng.integers(0, 12, n)
        strings = E.StringColumn(lens, rng.integers(97, 123, int(lens.sum())).astype(np.uint8))
        encs.append(E.encode(E.Trivial(lengths=E.FixedBitWidth()), strings, W.STRING))
        encs.append(E.encode(E.Trivial(lengths=E.Trivial()), strings, W.STRING))
    assert set(_names(encs)) == {"k_string_views", "k_copy_bytes"}
    check_parity(oracle, encs)
    _decode_twice(oracle, encs)


def test_string_lengths_beyond_the_blob_over_several_tiles(gpu):
    lengths = np.full(9000, 10, dtype=np.uint32)
    lens = E.encode(E.FixedBitWidth(), lengths, W.UINT32)
    enc = (E.prefix(W.TRIVIAL, W.STRING, 9000) + bytes([W.UNCOMPRESSED]) + struct.pack("<I", len(lens)) + lens + bytes(5000))
    plan = Plan.from_encodings([enc])
    dec = BatchDecoder(plan)
    dec.stage()
    dec.decode()
    with pytest.raises(capi.NimbleError, match="string lengths") as ei:
        dec.status()
    assert ei.value.status == capi.NBL_ERR_DEVICE_DATA
    plan.close()


def test_chains_longer_than_the_chip_holds(gpu, oracle):
    """5,000,000 runs / strings are 2,442 tiles each: more workgroups than are
    resident at once (2,048), so late tiles find finished prefixes."""
    rng = np.random.default_rng(55)
    runs = 5_000_000
    lengths = np.ones(runs, dtype=np.uint32)
    lengths[rng.integers(0, runs, 1000)] = 0
    lengths[-1] = 1
    rle = _rle(W.INT32, int(lengths.sum()), lengths, rng.integers(0, 1 << 16, runs))
    slens = rng.integers(0, 3, runs)
    strings = E.StringColumn(slens, rng.integers(97, 123, int(slens.sum())).astype(np.uint8))
    strs = E.encode(E.Trivial(lengths=E.FixedBitWidth()), strings, W.STRING)
    check_parity(oracle, [rle, strs])
    _decode_twice(oracle, [rle, strs])


def test_varints_in_one_tile_and_in_chains(gpu, oracle):
    """Varint payloads of one 4 KiB tile (no chain), of several tiles and of
    1,075: the decode tiles are the chain, no counting step runs."""
    rng = np.random.default_rng(56)
    small = E.encode(E.Varint(), rng.integers(0, 1 << 20, 900).astype(np.uint32), W.INT32)
    several = E.encode(E.Varint(), rng.integers(0, 1 << 40, 60_000).astype(np.uint64), W.INT64)
    big = E.encode(E.Varint(), rng.integers(128, 1 << 14, 2_200_000).astype(np.uint32), W.INT32)  # two bytes each: 1,075 tiles
    assert _names([small]) == ["k_varint_decode<4>"]
    assert _names([several]) == ["k_varint_decode<8>"]
    assert _names([big]) == ["k_varint_decode<4>"]
    check_parity(oracle, [small, several, big])
    _decode_twice(oracle, [small, several, big])
    for enc, cut in ((small, 40), (several, 3000)):  # fewer varints than rows: the last tile reports it
        plan = Plan.from_encodings([enc[: len(enc) - cut]])
        dec = BatchDecoder(plan)
        dec.stage()
        dec.decode()
        with pytest.raises(capi.NimbleError, match="fewer varints") as ei:
            dec.status()
        assert ei.value.status == capi.NBL_ERR_DEVICE_DATA
        plan.close()


def _nullable_counts(rng, rows, frac, dt, npt, nulls):
    mask = rng.random(rows) < frac
    vals = rng.integers(1, 1 << 30, int(mask.sum())).astype(npt)
    return E.encode_nullable(E.Nullable(data=E.Trivial(), nulls=nulls), vals, mask, dt)


def test_ranks_of_bitmap_selections_in_chains(gpu, oracle):
    """Null bitmaps stored as bits (Trivial<bool>): counted by a chained scan of
    65,536-row tiles -- 1, 2, 3, 70 and 1,025 of them -- for 4- and 8-byte rows
    (16 / 32 consumer tiles per counting tile) and MainlyConstant's inverted
    selection."""
    rng = np.random.default_rng(57)
    encs = []
    for rows in (131_072, 131_073, 196_608, 70 * 65_536 - 5, 1_025 * 65_536 - 77):
        for dt, npt in ((W.INT32, np.int32), (W.INT64, np.int64)):
            if rows > 10_000_000 and dt == W.INT64:
                continue
            encs.append(_nullable_counts(rng, rows, 0.3 if rows < 10_000_000 else 0.02, dt, npt, E.Trivial()))
    v = np.where(rng.random(400_000) < 0.9, 11, rng.integers(100, 1 << 20, 400_000)).astype(np.int32)
    encs.append(E.encode(E.MainlyConstant(is_common=E.Trivial(), other=E.Trivial()), v, W.INT32))
    names = _names(encs[1:2])
    assert names[0] == "k_reduce_popc" and len(names) == 2
    check_parity(oracle, encs)
    _decode_twice(oracle, encs[:-3] + encs[-1:])


def test_ranks_of_sparse_selections_come_scanned_from_the_bitmap_step(gpu, oracle):
    """SparseBool null lists of columns of 1, 2, 3, 70 and 5,000 bitmap tiles
    (65,536 rows; 16,384 where the list is dense): every tile takes the count in
    front of it from its place in the sorted list, no chain and no scan step."""
    rng = np.random.default_rng(58)
    encs = []
    for rows in (131_073, 196_608, 70 * 65_536 - 5):
        for frac in (0.02, 0.5, 0.97):
            for dt, npt in ((W.FLOAT, np.float32), (W.INT64, np.int64)):
                encs.append(_nullable_counts(rng, rows, frac, dt, npt, E.SparseBool()))
    encs.append(_nullable_counts(rng, 5_000 * 16_384 - 3, 0.9, W.INT32, np.int32, E.SparseBool(indices=E.FixedBitWidth())))
    v = np.where(rng.random(300_000) < 0.93, 11, rng.integers(100, 1 << 20, 300_000)).astype(np.int64)
    encs.append(E.encode(E.MainlyConstant(is_common=E.SparseBool(), other=E.Trivial()), v, W.INT64))
    assert _names(encs[:1])[0] == "k_sparse_bits" and len(_names(encs[:1])) == 2
    check_parity(oracle, encs)
    _decode_twice(oracle, encs[:6] + encs[-1:])


def test_a_sparse_selection_that_is_not_sorted_is_reported(gpu):
    """Positions out of order (or repeated) under a Nullable: the bitmap step's
    tiles find that their range of the list did not put that many bits into them."""
    rows = 200_000
    for idx in ([30_000, 10, 150_000, 5, rows], [7, 7, 9, rows], [5, 70_000, 140_000]):
        ind = E.encode(E.Trivial(), np.asarray(idx, dtype=np.uint32), W.UINT32)
        nulls = E.prefix(W.SPARSE_BOOL, W.BOOL, rows) + bytes([0]) + ind
        nn = rows - (len(idx) - 1)
        data = E.encode(E.Trivial(), np.arange(nn, dtype=np.uint32), W.INT32)
        enc = E.prefix(W.NULLABLE, W.INT32, rows) + struct.pack("<I", len(data)) + data + nulls
        for per_kernel in (False, True):
            plan = Plan.from_encodings([enc], per_kernel=per_kernel)
            dec = BatchDecoder(plan)
            dec.stage()
            dec.decode()
            with pytest.raises(capi.NimbleError, match="selected-row count") as ei:
                dec.status()
            assert ei.value.status == capi.NBL_ERR_DEVICE_DATA
            dec.fetch()
            plan.close()
