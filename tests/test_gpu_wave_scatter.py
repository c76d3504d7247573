"""GPU parity of the one-wave rank scatter (Nullable / MainlyConstant columns
whose tiles stage few values) against the CPU oracle: every path of the kernel
-- values staged early, staged late from the exact range, read row by row when a
tile holds more than the stage -- for 4- and 8-byte rows, plain and nested
selections.  check_parity plans the same bytes three ways (one-wave kernel from
the first tile on, per-operation launches, four-wave tiles only).
Semantics: encodings/NullableEncoding.h:207-236, MainlyConstantEncoding.h:208-253."""
import numpy as np
import pytest

from nimble_amd import encoder as E
from nimble_amd import wire as W
from nimble_amd.decoder import Plan

from util import check_parity

pytestmark = pytest.mark.gpu


def _uses_wave(enc):
    p = Plan.from_encodings([enc], per_kernel=True, wave_scatter_min_tiles=1)
    names = [l[0] for l in p.launches()]
    p.close()
    return any(n.startswith("k_select_scatter_wave") for n in names)


def _nested(rng, n, outer_frac, inner_other, npt, dt, nulls, dense_inner=None):
    """Nullable{MainlyConstant{SparseBool, Trivial}, nulls}: `inner_other` of the
    non-null rows take an other value; dense_inner = (first, last, fraction)
    raises that share inside a range of non-null ranks."""
    mask = rng.random(n) < outer_frac
    nn = int(mask.sum())
    is_common = rng.random(nn) >= inner_other
    if dense_inner:
        a, b, f = dense_inner
        is_common[a:b] = rng.random(b - a) >= f
    if nn:
        is_common[0] = True
    other = rng.integers(1, 1 << 30, int((~is_common).sum())).astype(npt)
    spec = E.MainlyConstant(is_common=E.SparseBool(indices=E.FixedBitWidth()), other=E.Trivial())
    data = E.mainly_constant_from_parts(spec, is_common, other, npt(7), dt)
    return E.nullable_wrap(data, mask, nulls, dt)


def test_plain_selections_all_sizes(gpu, oracle):
    rng = np.random.default_rng(31)
    encs, labels = [], []
    for dt, npt in ((W.INT32, np.int32), (W.FLOAT, np.float32), (W.INT64, np.int64), (W.DOUBLE, np.float64)):
        for n in (1, 63, 64, 65, 2047, 2048, 4095, 4096, 4097, 20_000, 70_001):
            for frac in (0.0, 0.03, 0.1):
                mask = rng.random(n) < frac
                vals = rng.integers(-(1 << 30), 1 << 30, int(mask.sum())).astype(npt)
                nulls = E.SparseBool() if (n + int(frac * 100)) % 2 else E.Trivial()
                encs.append(E.encode_nullable(E.Nullable(data=E.Trivial(), nulls=nulls), vals, mask, dt))
                labels.append(f"{W.DATA_TYPE_NAMES[dt]} n={n} selected={frac}")
    assert any(_uses_wave(e) for e in encs)
    check_parity(oracle, encs, labels=labels)


def test_plain_bit_packed_values(gpu, oracle):
    """Selected values that are bit-packed (every staged reader kind)."""
    rng = np.random.default_rng(32)
    encs, labels = [], []
    n = 30_000
    for dt, npt, widths in ((W.INT32, np.int32, (1, 7, 13, 25, 26, 31, 32)), (W.INT64, np.int64, (3, 24, 33, 57, 58, 63, 64))):
        for w in widths:
            mask = rng.random(n) < 0.08
            nn = int(mask.sum())
            raw = rng.integers(0, 2**63, nn, dtype=np.uint64)
            if w < 64:
                raw &= np.uint64((1 << w) - 1)
            vals = raw.astype(W.physical_dtype(dt)).view(npt)
            spec = E.Nullable(data=E.FixedBitWidth(bit_width=w, baseline=0), nulls=E.SparseBool())
            encs.append(E.encode_nullable(spec, vals, mask, dt))
            labels.append(f"{W.DATA_TYPE_NAMES[dt]} w={w}")
    check_parity(oracle, encs, labels=labels)


def test_a_tile_denser_than_the_stage(gpu, oracle):
    """A sparse column (so the planner picks the one-wave kernel) with one stretch
    of rows all selected: those tiles hold more values than the stage and read
    them row by row."""
    rng = np.random.default_rng(33)
    encs = []
    for dt, npt in ((W.FLOAT, np.float32), (W.INT64, np.int64)):
        n = 200_000
        mask = rng.random(n) < 0.02
        mask[9_000:15_000] = True
        vals = rng.integers(1, 1 << 30, int(mask.sum())).astype(npt)
        enc = E.encode_nullable(E.Nullable(data=E.Trivial(), nulls=E.SparseBool()), vals, mask, dt)
        assert _uses_wave(enc)
        encs.append(enc)
    check_parity(oracle, encs)


def test_nested_selections(gpu, oracle):
    rng = np.random.default_rng(34)
    encs, labels = [], []
    for dt, npt in ((W.FLOAT, np.float32), (W.DOUBLE, np.float64), (W.INT32, np.int32), (W.INT64, np.int64)):
        for n in (1, 100, 4096, 4097, 9000, 50_000):
            for outer, inner in ((0.01, 0.1), (0.2, 0.1), (0.8, 0.1), (0.5, 0.0), (1.0, 0.3)):
                nulls = E.SparseBool(indices=E.FixedBitWidth()) if outer < 0.5 else E.Trivial()
                encs.append(_nested(rng, n, outer, inner, npt, dt, nulls))
                labels.append(f"{W.DATA_TYPE_NAMES[dt]} n={n} outer={outer} inner={inner}")
    assert any(_uses_wave(e) for e in encs)
    check_parity(oracle, encs, labels=labels)


def test_nested_other_values_staged_late_and_not_at_all(gpu, oracle):
    """The inner tiles a row tile touches hold more other values than the stage:
    the tile stages the exact range it reads (a quarter of the ranks in a
    stretch are other values), or reads row by row where even that is too much
    (a stretch where every rank is one)."""
    rng = np.random.default_rng(35)
    encs = []
    for dt, npt in ((W.FLOAT, np.float32), (W.INT64, np.int64)):
        n = 300_000
        # ~150,000 non-null ranks; 1% other values, but 25% in ranks [20000, 40000) and all in [60000, 66000)
        mask = rng.random(n) < 0.5
        nn = int(mask.sum())
        is_common = rng.random(nn) >= 0.01
        is_common[20_000:40_000] = rng.random(20_000) >= 0.25
        is_common[60_000:66_000] = False
        is_common[0] = True
        other = rng.integers(1, 1 << 30, int((~is_common).sum())).astype(npt)
        spec = E.MainlyConstant(is_common=E.SparseBool(indices=E.FixedBitWidth()), other=E.Trivial())
        data = E.mainly_constant_from_parts(spec, is_common, other, npt(3), dt)
        enc = E.nullable_wrap(data, mask, E.Trivial(), dt)
        assert _uses_wave(enc)
        encs.append(enc)
    check_parity(oracle, encs)


def test_mainly_constant_root(gpu, oracle):
    """MainlyConstant as a chunk's root (inverted selection, fill = the common value)."""
    rng = np.random.default_rng(36)
    encs = []
    for dt, npt in ((W.INT32, np.int32), (W.DOUBLE, np.float64)):
        for n in (5, 5000, 66_000):
            v = np.where(rng.random(n) < 0.93, 11, rng.integers(100, 1 << 20, n)).astype(npt)
            encs.append(E.encode(E.MainlyConstant(is_common=E.SparseBool(), other=E.Trivial()), v, dt))
            encs.append(E.encode(E.MainlyConstant(is_common=E.Trivial(), other=E.FixedBitWidth()), v, dt))
    check_parity(oracle, encs)


def _mask_with_counts(rng, tile_rows, counts):
    """A selection whose tile t (of tile_rows rows) selects exactly counts[t] rows."""
    mask = np.zeros(tile_rows * len(counts), dtype=bool)
    for t, c in enumerate(counts):
        mask[t * tile_rows + rng.choice(tile_rows, c, replace=False)] = True
    return mask


def test_tiles_that_fill_the_stage_exactly_at_every_alignment(gpu, oracle):
    """Tiles whose selected values fill the one-wave kernel's stage to the last
    byte (608 4-byte / 304 8-byte values; 1024 values of 19 bits, also starting
    inside a byte), with the column at every offset modulo 16 in the arena: the
    staged copy starts up to 15 bytes in front of the values and must not reach
    the tile's bitmap words behind the stage."""
    rng = np.random.default_rng(37)
    cols = []
    for dt, npt, per_tile, full in ((W.INT32, np.int32, 4096, 608), (W.INT64, np.int64, 2048, 304)):
        mask = _mask_with_counts(rng, per_tile, [full, full, 5, 0, full, 1, 0, 0])
        vals = rng.integers(1, 1 << 30, int(mask.sum())).astype(npt)
        cols.append(E.encode_nullable(E.Nullable(data=E.Trivial(), nulls=E.Trivial()), vals, mask, dt))
    for counts in ([1024, 0, 1024, 0, 0, 0, 0, 0], [3, 1024, 0, 0, 1, 1024, 0, 0, 0]):
        mask = _mask_with_counts(rng, 4096, counts)
        vals = rng.integers(0, 1 << 19, int(mask.sum())).astype(np.int32)
        spec = E.Nullable(data=E.FixedBitWidth(bit_width=19, baseline=0), nulls=E.SparseBool())
        cols.append(E.encode_nullable(spec, vals, mask, W.INT32))
    for c in cols:
        assert _uses_wave(c)
    encs, labels, pos = [], [], 0
    for r in range(16):
        for i, c in enumerate(cols):
            pad_rows = (r - pos - 7) % 16  # a Trivial<int8> of k rows is 7 + k bytes
            encs.append(E.encode(E.Trivial(), np.arange(pad_rows, dtype=np.int8), W.INT8))
            labels.append(f"pad before column {i} at offset {r} mod 16")
            pos += len(encs[-1])
            assert pos % 16 == r
            encs.append(c)
            labels.append(f"column {i} at offset {r} mod 16")
            pos += len(c)
    check_parity(oracle, encs, labels=labels)
