"""GPU parity of the two ways a column's tile sums become offsets (RLE run ends,
RLEEncoding.h:289-301; string views, TrivialEncoding.cpp:151-164): up to 1,024
tiles of 2,048 lengths every consumer tile adds up the sums in front of its own
and no scan step runs (one tile: no sums either); above that a scan step does it.
Both against the CPU oracle, decoded twice (the per-node flag word of zero-length
runs and the totals are rewritten by every decode)."""
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
    """5,000 and 70,000 runs (3 and 35 tiles of sums), some of length zero: the
    first tile of sums clears the flag word, run-end tiles raise it."""
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
    names = _names(encs[:1])
    assert "k_reduce_sum" in names and "k_scan_tiles" not in names
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
    assert "k_scan_tiles" not in _names(encs)
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


def test_more_than_1024_tiles_of_sums_keep_the_scan_step(gpu, oracle):
    """2,100,000 runs / strings are 1,026 tiles of sums: a scan step adds them up."""
    rng = np.random.default_rng(55)
    runs = 2_100_000
    lengths = np.ones(runs, dtype=np.uint32)
    lengths[rng.integers(0, runs, 1000)] = 0
    lengths[-1] = 1
    rle = _rle(W.INT32, int(lengths.sum()), lengths, rng.integers(0, 1 << 16, runs))
    slens = rng.integers(0, 3, runs)
    strings = E.StringColumn(slens, rng.integers(97, 123, int(slens.sum())).astype(np.uint8))
    strs = E.encode(E.Trivial(lengths=E.FixedBitWidth()), strings, W.STRING)
    names = _names([rle, strs])
    assert names.count("k_scan_tiles") == 2
    check_parity(oracle, [rle, strs])


def test_varints_in_one_tile_several_tiles_and_behind_a_scan_step(gpu, oracle):
    """Varint payloads of one 4 KiB tile (no counts), of several (the decode tiles
    add up the counts in front of their own) and of more than 1,024 (a scan step)."""
    rng = np.random.default_rng(56)
    small = E.encode(E.Varint(), rng.integers(0, 1 << 20, 900).astype(np.uint32), W.INT32)
    several = E.encode(E.Varint(), rng.integers(0, 1 << 40, 60_000).astype(np.uint64), W.INT64)
    big = E.encode(E.Varint(), rng.integers(128, 1 << 14, 2_200_000).astype(np.uint32), W.INT32)  # two bytes each: 1,075 tiles
    assert _names([small]) == ["k_varint_decode<4>"]
    assert _names([several]) == ["k_reduce_varint", "k_varint_decode<8>"]
    assert _names([big]) == ["k_reduce_varint", "k_scan_tiles", "k_varint_decode<4>"]
    check_parity(oracle, [small, several, big])
    _decode_twice(oracle, [small, several])
    for enc, cut in ((small, 40), (several, 3000)):  # fewer varints than rows: the last tile reports it
        plan = Plan.from_encodings([enc[: len(enc) - cut]])
        dec = BatchDecoder(plan)
        dec.stage()
        dec.decode()
        with pytest.raises(capi.NimbleError, match="fewer varints") as ei:
            dec.status()
        assert ei.value.status == capi.NBL_ERR_DEVICE_DATA
        plan.close()
