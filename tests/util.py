"""Shared test helpers: the reference tests' data patterns restated over numpy
(dwio/nimble/common/tests/TestUtils.h:85-247) and GPU-vs-oracle comparison."""
from __future__ import annotations

import numpy as np

from nimble_amd import encoder as E
from nimble_amd import wire as W

NP_OF = {
    W.INT8: np.int8, W.UINT8: np.uint8, W.INT16: np.int16, W.UINT16: np.uint16,
    W.INT32: np.int32, W.UINT32: np.uint32, W.INT64: np.int64, W.UINT64: np.uint64,
    W.FLOAT: np.float32, W.DOUBLE: np.float64,
}
INTEGER_TYPES = (W.INT8, W.UINT8, W.INT16, W.UINT16, W.INT32, W.UINT32, W.INT64, W.UINT64)
ALL_NUMERIC = INTEGER_TYPES + (W.FLOAT, W.DOUBLE)


def random_data(rng, dtype: int, n: int) -> np.ndarray:
    """makeRandomData: integers non-negative over (almost) the full range,
    floats are random integers converted to floating point."""
    if dtype == W.BOOL:
        return rng.integers(0, 2, n).astype(np.bool_)
    npt = NP_OF[dtype]
    if dtype == W.FLOAT:
        return rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.float32)
    if dtype == W.DOUBLE:
        return rng.integers(0, 2**63, n, dtype=np.uint64).astype(np.float64)
    bits = np.dtype(npt).itemsize * 8
    hi = 2 ** (bits - 1) if bits < 64 else 2**63
    raw = rng.integers(0, hi, n, dtype=np.uint64)
    return raw.astype(W.physical_dtype(dtype)).view(npt)


def constant_data(rng, dtype: int, n: int) -> np.ndarray:
    d = random_data(rng, dtype, n)
    d[:] = d[0]
    return d


def rle_data(rng, dtype: int, n: int) -> np.ndarray:
    """makeRLEData: runs of random length covering the array."""
    d = random_data(rng, dtype, n)
    i = 0
    while i < n:
        run = 1 if n - i == 1 else 1 + int(rng.integers(0, n - i - 1)) if n - i - 1 > 0 else 1
        run = min(run, max(1, n // 8))  # keep more than a handful of runs
        d[i:i + run] = d[i]
        i += run
    return d


def small_range_data(rng, dtype: int, n: int) -> np.ndarray:
    """makeHuffmanData: values drawn from a small symbol range."""
    npt = NP_OF[dtype]
    symbols = 1 + int(rng.integers(0, max(n, 1)))
    maxv = symbols if rng.integers(0, 2) else 4096
    if np.issubdtype(npt, np.integer):
        maxv = min(maxv, int(np.iinfo(npt).max))
    return rng.integers(0, max(maxv, 1), n).astype(npt)


def patterns(rng, dtype: int, n: int):
    out = [("random", random_data(rng, dtype, n)), ("constant", constant_data(rng, dtype, n)),
           ("rle", rle_data(rng, dtype, n))]
    if dtype in INTEGER_TYPES:
        out.append(("small", small_range_data(rng, dtype, n)))
    return out


def random_strings(rng, n: int, max_len: int = 50) -> E.StringColumn:
    lens = rng.integers(0, max_len, n).astype(np.uint32)
    blob = rng.integers(0, 256, int(lens.sum()), dtype=np.uint64).astype(np.uint8)
    return E.StringColumn(lens, blob)


def phys(values: np.ndarray, dtype: int) -> np.ndarray:
    if dtype == W.BOOL:
        return np.asarray(values).astype(np.uint8)
    return np.ascontiguousarray(values).view(W.physical_dtype(dtype))


def unpack_bits(bitmap: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(np.asarray(bitmap, dtype=np.uint8), bitorder="little")[:n].astype(bool)


def oracle_chunk(O, enc: bytes, varint_rows: bool = False):
    """Whole-chunk oracle decode in the GPU output convention."""
    e = O.Encoding(enc, varint_rows)
    n = e.row_count
    if e.data_type == W.STRING:
        vals = e.materialize(n)
    else:
        vals = e.materialize(n)
    nulls = None
    nn = n
    if e.is_nullable:
        e2 = O.Encoding(enc, varint_rows)
        # nulls child bitmap via materializeNullable on a scratch buffer (numeric) or
        # by walking the views (string)
        if e.data_type == W.STRING:
            # zero-length placeholder values: use the bool child directly
            nulls = _nullable_bitmap(O, enc, varint_rows)
            nn = int(unpack_bits(nulls, n).sum())
        else:
            buf = np.zeros(n, dtype=vals.dtype)
            bm = np.zeros((n + 7) // 8 + 8, dtype=np.uint8)
            nn, req = e2.materialize_nullable(n, buf, bm)
            if not req:
                bm[:] = 0
                bm[:(n + 7) // 8] = np.packbits(np.ones(n, dtype=np.uint8), bitorder="little")
            nulls = bm[:(n + 7) // 8].copy()
    return vals, nulls, nn, n, e.data_type


def _nullable_bitmap(O, enc: bytes, varint_rows: bool):
    """Bitmap of the nulls child of a Nullable encoding (second child, to the end)."""
    import struct

    pre = 6
    if varint_rows:
        pre = 2
        while enc[pre] & 0x80:
            pre += 1
        pre += 1
    size = struct.unpack_from("<I", enc, pre)[0]
    nulls_enc = enc[pre + 4 + size:]
    e = O.Encoding(nulls_enc, varint_rows)
    b = e.materialize(e.row_count)
    return np.packbits(b.astype(np.uint8), bitorder="little")


def assert_chunk_equal(res, ora, label=""):
    vals, nulls, nn, n, dt = ora
    assert res.rows == n, f"{label}: rows {res.rows} != {n}"
    assert res.data_type == dt, label
    if nulls is not None:
        assert res.nulls is not None, f"{label}: missing nulls"
        assert np.array_equal(res.nulls, nulls), f"{label}: null bitmap differs"
        assert res.non_null == nn, f"{label}: non-null count {res.non_null} != {nn}"
    else:
        assert res.nulls is None, f"{label}: unexpected nulls"
    if dt == W.STRING:
        assert np.array_equal(res.values[0], vals[0]), f"{label}: string lengths differ"
        assert np.array_equal(res.values[1], vals[1]), f"{label}: string bytes differ"
    else:
        if not np.array_equal(res.values, vals):
            bad = np.flatnonzero(res.values != vals)
            raise AssertionError(f"{label}: {bad.size} of {n} values differ, first at {bad[0]}: "
                                 f"gpu={res.values[bad[0]]} oracle={vals[bad[0]]}")


def check_parity(O, encodings, varint_rows: bool = False, labels=None):
    """Decode all encodings in ONE GPU batch and compare each with the oracle."""
    from nimble_amd.decoder import decode_encodings

    # one tile is enough for the one-wave rank scatter and the LDS-alphabet dictionary kernel to get their
    # launches: the test inputs are small
    results = decode_encodings(encodings, varint_rows, wave_scatter_min_tiles=1, lds_dictionary_min_tiles=1)
    assert len(results) == len(encodings)
    # the same batch with one launch per operation instead of one per level
    unfused = decode_encodings(encodings, varint_rows, per_kernel=True)
    # and with every rank scatter on the four-wave tile kernel and every dictionary gathered from memory
    four_wave = decode_encodings(encodings, varint_rows, four_wave_scatter=True, no_lds_dictionary=True)
    for i, (enc, res) in enumerate(zip(encodings, results)):
        label = labels[i] if labels else f"chunk {i}"
        ora = oracle_chunk(O, enc, varint_rows)
        assert_chunk_equal(res, ora, label)
        assert_chunk_equal(unfused[i], ora, label + " (per-operation launches)")
        assert_chunk_equal(four_wave[i], ora, label + " (four-wave rank scatter, dictionaries from memory)")
    return results
