"""GPU: payloads whose headers are well formed but whose DATA is inconsistent
(run lengths that do not add up, dictionary indices past the alphabet, ...).
The reference meets these inside its decode loops (NIMBLE_CHECK in
RLEEncoding.h:289-301, DictionaryEncoding.h:227-236, ...); here every kernel
clamps the data-dependent index, finishes, and raises a status bit that
nbl_decode_status reports as NBL_ERR_DEVICE_DATA.  Also: empty chunks."""
import struct

import numpy as np
import pytest

from nimble_amd import capi
from nimble_amd import encoder as E
from nimble_amd import wire as W
from nimble_amd.decoder import BatchDecoder, Plan

from util import check_parity

pytestmark = pytest.mark.gpu


def _decode_expect_device_data(encodings, match):
    # fused and per-operation launches; rank scatters as four-wave tiles and on the one-wave kernel
    for per_kernel, wave_min in ((False, 0), (True, 0), (False, 1), (True, 1)):
        plan = Plan.from_encodings(encodings, per_kernel=per_kernel, wave_scatter_min_tiles=wave_min)
        dec = BatchDecoder(plan)
        dec.stage()
        dec.decode()
        with pytest.raises(capi.NimbleError, match=match) as ei:
            dec.status()
        assert ei.value.status == capi.NBL_ERR_DEVICE_DATA
        dec.fetch()  # outputs are defined memory, whatever they hold
        plan.close()


def _rle(dtype, rows, lengths, values, lengths_spec=None):
    lens = E.encode(lengths_spec or E.FixedBitWidth(), np.asarray(lengths, dtype=np.uint32), W.UINT32)
    vals = E.encode(E.Trivial(), np.asarray(values).astype(W.physical_dtype(dtype)), dtype)
    return E.prefix(W.RLE, dtype, rows) + struct.pack("<I", len(lens)) + lens + vals


def test_rle_lengths_do_not_add_up(gpu):
    rng = np.random.default_rng(1)
    for rows, lengths in ((5000, [1000, 1000, 1000]),            # too short
                          (5000, [4000, 4000, 4000]),            # too long
                          (70_000, rng.integers(1, 9, 3000)),    # many runs, short
                          (100, [2**31, 2**31, 2**31, 7])):      # 32-bit wrap of the run ends
        values = np.arange(len(lengths)) + 5
        good_sum = int(np.sum(np.asarray(lengths, dtype=np.uint64)) & 0xFFFFFFFF) == rows
        assert not good_sum
        _decode_expect_device_data([_rle(W.INT64, rows, lengths, values), _rle(W.INT16, rows, lengths, values)], "run lengths")


def test_rle_zero_length_runs_decode_like_the_oracle(gpu, oracle):
    # zero-length runs are tolerated (they contribute no rows): the slow path resolves them
    lengths = [3, 0, 0, 4, 0, 5, 0]
    values = [10, 11, 12, 13, 14, 15, 16]
    check_parity(oracle, [_rle(W.INT32, 12, lengths, values), _rle(W.INT64, 12, lengths, values)])


def test_dictionary_index_out_of_range(gpu):
    for dtype, npt in ((W.INT64, np.uint64), (W.INT32, np.uint32), (W.INT8, np.uint8)):
        alpha = E.encode(E.Trivial(), np.arange(10, dtype=npt), dtype)
        idx = np.arange(3000, dtype=np.uint32) % 10
        idx[[5, 1700, 2999]] = (10, 4_000_000_000, 77)
        ind = E.encode(E.Trivial(), idx, W.UINT32)
        enc = E.prefix(W.DICTIONARY, dtype, idx.size) + struct.pack("<I", len(alpha)) + alpha + ind
        _decode_expect_device_data([enc], "dictionary index")


def test_nullable_count_mismatch(gpu):
    rows = 9000
    mask = np.random.default_rng(2).random(rows) < 0.5
    nulls = E.encode(E.Trivial(), mask, W.BOOL)
    for delta in (-7, +7, -int(mask.sum())):
        nn = int(mask.sum()) + delta
        if nn <= 0:
            nn = 1
        data = E.encode(E.Trivial(), np.arange(nn, dtype=np.uint32), W.INT32)
        enc = E.prefix(W.NULLABLE, W.INT32, rows) + struct.pack("<I", len(data)) + data + nulls
        _decode_expect_device_data([enc], "selected-row count")


def test_nullable_count_mismatch_over_the_lds_dictionary_kernels(gpu):
    """The same mismatch where the values child is a Dictionary gathered inside the rank scatter by the kernels that
    keep the alphabet in LDS (8-byte numbers; strings in compact form): ranks past the index column are clamped."""
    from nimble_amd import workloads as WL

    rng = np.random.default_rng(3)
    rows = 150_000
    mask = rng.random(rows) < 0.5
    nulls = E.encode(E.Trivial(), mask, W.BOOL)
    strings = E.StringColumn.from_list(WL.random_alphabet(rng, 4000).to_list()[:3000])
    numbers = (rng.permutation(1 << 20)[:3000].astype(np.int64) << 20)
    for delta in (-700, +700):
        nn = int(mask.sum()) + delta
        idx = rng.integers(0, 3000, nn)
        idx[:3000] = np.arange(3000)  # every entry used: the encoder keeps the alphabet whole
        encs = []
        for dt, alpha, spec in ((W.STRING, strings, E.Dictionary(alphabet=E.Trivial(lengths=E.FixedBitWidth()), indices=E.FixedBitWidth())),
                                (W.INT64, numbers, E.Dictionary(alphabet=E.Trivial(), indices=E.FixedBitWidth()))):
            data = E.encode_dictionary_from_indices(spec, alpha, idx, dt)
            encs.append(E.prefix(W.NULLABLE, dt, rows) + struct.pack("<I", len(data)) + data + nulls)
        for options in (dict(lds_dictionary_min_tiles=1), dict(lds_dictionary_min_tiles=1, per_kernel=True), dict(no_lds_dictionary=True)):
            plan = Plan.from_encodings(encs, **options)
            names = [l[0] for l in plan.launches()]
            if options.get("per_kernel"):
                assert "k_dict_str<8>" in names and "k_dict_lds<8>" in names
            dec = BatchDecoder(plan)
            dec.stage()
            dec.decode()
            with pytest.raises(capi.NimbleError, match="selected-row count") as ei:
                dec.status()
            assert ei.value.status == capi.NBL_ERR_DEVICE_DATA
            dec.fetch()
            plan.close()


def test_varint_truncated(gpu):
    v = np.random.default_rng(3).integers(0, 1 << 40, 5000).astype(np.uint64)
    enc = E.encode(E.Varint(), v, W.INT64)
    _decode_expect_device_data([enc[: len(enc) - 300]], "fewer varints")


def test_string_lengths_exceed_blob(gpu):
    lengths = np.full(100, 10, dtype=np.uint32)
    lens = E.encode(E.FixedBitWidth(), lengths, W.UINT32)
    enc = (E.prefix(W.TRIVIAL, W.STRING, 100) + bytes([W.UNCOMPRESSED]) + struct.pack("<I", len(lens)) + lens + bytes(500))
    plan = Plan.from_encodings([enc])
    dec = BatchDecoder(plan)
    dec.stage()
    dec.decode()
    with pytest.raises(capi.NimbleError, match="string lengths") as ei:
        dec.status()
    assert ei.value.status == capi.NBL_ERR_DEVICE_DATA
    plan.close()


def test_sparse_bool_positions_out_of_range_or_unsorted(gpu):
    # positions past the row count / out of order never address outside the bitmap
    rows = 40_000
    for idx in ([5, 9, 50_000, 60_000, rows], [30_000, 10, 20_000, 5, rows], [2**32 - 1] * 50 + [rows]):
        ind = E.encode(E.Trivial(), np.asarray(idx, dtype=np.uint32), W.UINT32)
        enc = E.prefix(W.SPARSE_BOOL, W.BOOL, rows) + bytes([1]) + ind
        plan = Plan.from_encodings([enc])
        dec = BatchDecoder(plan)
        dec.stage()
        dec.decode()
        dec.status()  # no status bit: the reference does not validate these either
        out = dec.fetch()[0].values
        assert out.size == rows and set(np.unique(out)) <= {0, 1}
        plan.close()


def test_empty_chunks(gpu, oracle):
    """rowCount == 0 for every encoding that can be empty (EncodingTest.cpp covers
    empty inputs for all but Constant / MainlyConstant, which cannot be)."""
    empty32 = np.zeros(0, dtype=np.uint32)
    encs = [
        E.encode(E.Trivial(), empty32, W.INT32),
        E.encode(E.FixedBitWidth(), np.zeros(0, dtype=np.uint64), W.INT64),
        E.encode(E.Varint(), empty32, W.UINT32),
        E.encode(E.RLE(), empty32, W.INT32),
        E.encode(E.Dictionary(), np.zeros(0, dtype=np.uint16), W.INT16),
        E.encode(E.Trivial(), np.zeros(0, dtype=bool), W.BOOL),
        E.encode(E.SparseBool(), np.zeros(0, dtype=bool), W.BOOL),
        E.encode(E.RLE(), np.zeros(0, dtype=bool), W.BOOL),
        E.encode(E.Trivial(), E.StringColumn.from_list([]), W.STRING),
        E.encode_nullable(E.Nullable(), empty32, np.zeros(0, dtype=bool), W.INT32),
        E.encode(E.FixedBitWidth(), np.arange(7, dtype=np.uint32), W.INT32),  # a non-empty neighbour in the same batch
    ]
    res = check_parity(oracle, encs)
    assert [r.rows for r in res] == [0] * 10 + [7]


@pytest.mark.parametrize("varint_rows", [False, True])
def test_rle_without_rows_over_non_empty_run_lengths(gpu, oracle, varint_rows):
    """rowCount 0 with run lengths present (root, values child of an all-null
    Nullable, bool): decodes to empty / all-null columns like the oracle, and a
    populated neighbour in the same batch is unaffected."""
    from test_capi_cpu import empty_rle_cases

    neighbour = E.encode(E.RLE(), np.repeat(np.arange(50, dtype=np.int64), 40), W.INT64, varint_rows)
    res = check_parity(oracle, empty_rle_cases(varint_rows) + [neighbour], varint_rows)
    assert [r.rows for r in res] == [0, 3000, 0, 2000]


def test_status_gathers_over_decodes_and_clears_when_reported(gpu, oracle):
    """The flag word lives in the work buffer: it is zeroed by the upload, every
    decode can only add bits, and the call that reports them clears them.  A
    damaged batch therefore reports once per decode (or series of decodes), a
    later report without a decode in between is clean, and the zero-length-run
    flag of an RLE node is rebuilt by every decode (its scan resets it)."""
    bad = _rle(W.INT64, 5000, [1000, 1000, 1000], [1, 2, 3])
    plan = Plan.from_encodings([bad])
    dec = BatchDecoder(plan)
    dec.stage()
    dec.decode()
    dec.decode()
    with pytest.raises(capi.NimbleError, match="run lengths"):
        dec.status()
    dec.status()  # reported and cleared, nothing decoded since
    dec.decode()
    with pytest.raises(capi.NimbleError, match="run lengths"):
        dec.status()
    plan.close()
    # zero-length runs: the same decoder decodes twice, both passes match the oracle
    enc = _rle(W.INT32, 12, [3, 0, 0, 4, 0, 5, 0], [10, 11, 12, 13, 14, 15, 16])
    plan = Plan.from_encodings([enc])
    dec = BatchDecoder(plan)
    dec.stage()
    for _ in range(2):
        dec.decode()
        dec.status()
        got = dec.fetch()[0]
        assert np.asarray(got.values).view(np.int32).tolist() == [10] * 3 + [13] * 4 + [15] * 5
    plan.close()


def test_bool_rle_root_with_zero_length_runs(gpu, oracle):
    """RLE<bool> as a chunk's root writes its bytes itself; zero-length runs (which
    flip the value without a row) take the per-row search, several tiles long."""
    rng = np.random.default_rng(11)
    encs = []
    for rows in (12, 16384, 50_000):
        lengths = []
        left = rows
        while left:
            if rng.random() < 0.2:
                lengths.append(0)
                continue
            n = int(min(left, rng.integers(1, 400)))
            lengths.append(n)
            left -= n
        for first in (0, 1):
            lens = E.encode(E.Trivial(), np.asarray(lengths, dtype=np.uint32), W.UINT32)
            encs.append(E.prefix(W.RLE, W.BOOL, rows) + struct.pack("<I", len(lens)) + lens + bytes([first]))
    check_parity(oracle, encs)
