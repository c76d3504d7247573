"""GPU parity: the HIP decode (through the C ABI) against the CPU oracle on the
same encoded bytes, bit-exact.  Mirrors the reference's master matrix
(dwio/nimble/encodings/tests/EncodingTest.cpp:500-540: encodings x types x
data patterns) and its per-encoding deterministic inputs."""
import numpy as np
import pytest

from nimble_amd import encoder as E
from nimble_amd import wire as W

from util import (ALL_NUMERIC, INTEGER_TYPES, NP_OF, check_parity, patterns, random_data, random_strings,
                  rle_data)

pytestmark = pytest.mark.gpu

SIZES = (1, 2, 63, 64, 65, 1000, 4097)


def _fbw_values(rng, dtype, n, width, baseline):
    npt = NP_OF[dtype] if dtype in NP_OF else None
    raw = rng.integers(0, 2**63, n, dtype=np.uint64)
    if width < 64:
        raw &= np.uint64((1 << width) - 1)
    else:
        raw |= rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63)
    if n:
        raw[rng.integers(0, n)] = 0  # make min the baseline
        raw[rng.integers(0, n)] = (1 << width) - 1 if width < 64 else 2**64 - 1
    pt = W.physical_dtype(dtype)
    vals = (raw + np.uint64(baseline)).astype(pt) if width < 64 else raw.astype(pt)
    return vals.view(npt)


def test_fbw_all_widths_int64(gpu, oracle):
    rng = np.random.default_rng(7)
    encs, labels = [], []
    for width in range(1, 65):
        for n in (1, 77, 1029):
            base = 0 if width >= 63 else int(rng.integers(0, 1 << 20))
            raw = rng.integers(0, 2**63, n, dtype=np.uint64)
            if width < 64:
                raw &= np.uint64((1 << width) - 1)
            else:
                raw = raw | (rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63))
            vals = raw + np.uint64(base)
            encs.append(E.encode(E.FixedBitWidth(bit_width=width, baseline=base), vals.view(np.int64), W.INT64))
            labels.append(f"w={width} n={n}")
    check_parity(oracle, encs, labels=labels)


@pytest.mark.parametrize("dtype", [W.INT8, W.UINT8, W.INT16, W.UINT16, W.INT32, W.UINT32, W.FLOAT])
def test_fbw_narrow_types(gpu, oracle, dtype):
    rng = np.random.default_rng(dtype)
    bits = W.elem_bytes(dtype) * 8
    encs, labels = [], []
    for width in range(1, min(bits, 32) + 1):
        n = int(rng.integers(1, 3000))
        raw = rng.integers(0, 1 << width, n, dtype=np.uint64)
        base = int(rng.integers(0, (1 << bits) - (1 << width) + 1))
        vals = (raw + np.uint64(base)).astype(W.physical_dtype(dtype))
        encs.append(E.encode(E.FixedBitWidth(bit_width=width, baseline=base), vals, dtype))
        labels.append(f"dtype={dtype} w={width} n={n}")
    check_parity(oracle, encs, labels=labels)


def test_fbw_wraparound_baseline(gpu, oracle):
    """baseline + residual wraps modulo 2^bits exactly like the CPU add."""
    encs = []
    for dtype, base in ((W.UINT8, 250), (W.UINT16, 65530), (W.UINT32, 2**32 - 3), (W.UINT64, 2**64 - 3)):
        pt = W.physical_dtype(dtype)
        # hand-assemble: residuals 0..9 with a baseline near the top of the range
        resid = np.arange(10, dtype=np.uint64)
        body = (E.prefix(W.FIXED_BIT_WIDTH, dtype, 10) + bytes([0]) + np.array([base], dtype=pt).tobytes() +
                bytes([8]) + E.bitpack(resid, 8))
        encs.append(body)
    check_parity(oracle, encs)


@pytest.mark.parametrize("dtype", ALL_NUMERIC)
def test_numeric_encodings_by_pattern(gpu, oracle, dtype):
    rng = np.random.default_rng(100 + dtype)
    encs, labels = [], []
    for n in SIZES:
        for pname, data in patterns(rng, dtype, n):
            specs = [("trivial", E.Trivial()), ("fbw", E.FixedBitWidth()), ("fbw-exact", E.FixedBitWidth(exact_bits=True)),
                     ("rle", E.RLE()), ("rle-triv", E.RLE(lengths=E.Trivial(), values=E.Trivial())),
                     ("dict", E.Dictionary()), ("dict-fbw", E.Dictionary(alphabet=E.FixedBitWidth(), indices=E.Trivial())),
                     ("mc", E.MainlyConstant()), ("mc-bitmap", E.MainlyConstant(is_common=E.Trivial(), other=E.FixedBitWidth()))]
            if W.elem_bytes(dtype) >= 4:
                specs.append(("varint", E.Varint()))
            if pname == "constant":
                specs.append(("constant", E.Constant()))
            for sname, spec in specs:
                encs.append(E.encode(spec, data, dtype))
                labels.append(f"{sname}/{pname}/n={n}")
    check_parity(oracle, encs, labels=labels)


def test_bool_encodings(gpu, oracle):
    rng = np.random.default_rng(5)
    encs, labels = [], []
    for n in SIZES + (16384, 16385, 40000):
        for p in (0.0, 0.03, 0.5, 0.97, 1.0):
            b = rng.random(n) < p
            for sname, spec in (("trivial", E.Trivial()), ("sparse", E.SparseBool()), ("sparse-triv", E.SparseBool(indices=E.Trivial())),
                                ("rle", E.RLE()), ("rle-varint", E.RLE(lengths=E.Varint()))):
                encs.append(E.encode(spec, b, W.BOOL))
                labels.append(f"{sname}/p={p}/n={n}")
            if p in (0.0, 1.0):
                encs.append(E.encode(E.Constant(), b, W.BOOL))
                labels.append(f"constant/p={p}/n={n}")
        runs = np.repeat(rng.random(max(1, n // 30)) < 0.5, 30)[:n]
        if runs.size:
            encs.append(E.encode(E.RLE(), runs, W.BOOL))
            labels.append(f"rle-long/n={n}")
    check_parity(oracle, encs, labels=labels)


@pytest.mark.parametrize("dtype", [W.INT32, W.INT64, W.FLOAT, W.DOUBLE, W.UINT8, W.INT16])
def test_nullable(gpu, oracle, dtype):
    rng = np.random.default_rng(200 + dtype)
    encs, labels = [], []
    for n in SIZES + (20000,):
        for p_null in (0.0, 0.05, 0.5, 0.95, 1.0):
            mask = rng.random(n) >= p_null
            nn = int(mask.sum())
            vals = rle_data(rng, dtype, nn) if nn else random_data(rng, dtype, 0)
            for sname, spec in (
                    ("triv/triv", E.Nullable(data=E.Trivial(), nulls=E.Trivial())),
                    ("fbw/sparse", E.Nullable(data=E.FixedBitWidth(), nulls=E.SparseBool())),
                    ("rle/rle", E.Nullable(data=E.RLE(), nulls=E.RLE())),
                    ("dict/sparse", E.Nullable(data=E.Dictionary(), nulls=E.SparseBool()))):
                if nn == 0 and sname in ("dict/sparse",):
                    continue
                encs.append(E.encode_nullable(spec, vals, mask, dtype))
                labels.append(f"{sname}/p={p_null}/n={n}")
            if nn and p_null > 0:
                mcv = vals.copy()
                mcv[rng.random(nn) < 0.9] = mcv[0]
                encs.append(E.encode_nullable(E.Nullable(data=E.MainlyConstant(), nulls=E.SparseBool()), mcv, mask, dtype))
                labels.append(f"mc/sparse/p={p_null}/n={n}")
    check_parity(oracle, encs, labels=labels)


def test_nullable_bool(gpu, oracle):
    rng = np.random.default_rng(9)
    encs, labels = [], []
    for n in (1, 64, 65, 1000, 16385, 50000):
        for p_null in (0.0, 0.1, 0.6, 1.0):
            mask = rng.random(n) >= p_null
            vals = rng.random(int(mask.sum())) < 0.3
            for sname, spec in (("triv/triv", E.Nullable(data=E.Trivial(), nulls=E.Trivial())),
                                ("sparse/sparse", E.Nullable(data=E.SparseBool(), nulls=E.SparseBool())),
                                ("rle/rle", E.Nullable(data=E.RLE(), nulls=E.RLE()))):
                encs.append(E.encode_nullable(spec, vals, mask, W.BOOL))
                labels.append(f"{sname}/p={p_null}/n={n}")
    check_parity(oracle, encs, labels=labels)


def test_strings(gpu, oracle):
    rng = np.random.default_rng(11)
    encs, labels = [], []
    for n in (1, 2, 64, 1000, 5000):
        col = random_strings(rng, n)
        encs.append(E.encode(E.Trivial(), col, W.STRING)); labels.append(f"trivial/n={n}")
        encs.append(E.encode(E.Trivial(lengths=E.Varint()), col, W.STRING)); labels.append(f"trivial-varint/n={n}")
        alpha = random_strings(rng, max(1, min(64, n)))
        picked = alpha.take(rng.integers(0, len(alpha), n))
        encs.append(E.encode(E.Dictionary(), picked, W.STRING)); labels.append(f"dict/n={n}")
        runs = alpha.take(np.sort(rng.integers(0, len(alpha), n)))
        encs.append(E.encode(E.RLE(values=E.Trivial()), runs, W.STRING)); labels.append(f"rle/n={n}")
        encs.append(E.encode(E.RLE(values=E.Dictionary()), runs, W.STRING)); labels.append(f"rle-dict/n={n}")
        mcs = alpha.take(np.where(rng.random(n) < 0.9, 0, rng.integers(0, len(alpha), n)))
        encs.append(E.encode(E.MainlyConstant(other=E.Trivial()), mcs, W.STRING)); labels.append(f"mc/n={n}")
        encs.append(E.encode(E.MainlyConstant(other=E.Dictionary()), mcs, W.STRING)); labels.append(f"mc-dict/n={n}")
        encs.append(E.encode(E.Constant(), alpha.take(np.zeros(n, dtype=np.int64)), W.STRING)); labels.append(f"const/n={n}")
        mask = rng.random(n) < 0.8
        nn = int(mask.sum())
        if nn:
            encs.append(E.encode_nullable(E.Nullable(data=E.Dictionary(), nulls=E.SparseBool()),
                                          alpha.take(rng.integers(0, len(alpha), nn)), mask, W.STRING))
            labels.append(f"nullable-dict/n={n}")
            encs.append(E.encode_nullable(E.Nullable(data=E.Trivial(), nulls=E.Trivial()),
                                          random_strings(rng, nn), mask, W.STRING))
            labels.append(f"nullable-trivial/n={n}")
    check_parity(oracle, encs, labels=labels)


def test_reference_deterministic_inputs(gpu, oracle):
    """Inputs the reference's own tests use: FixedBitWidthEncodingTest.cpp:122-138,332-388,
    RLEEncodingTest.cpp:142-148,184-195, FixedBitArrayTests.cpp:27-39,151-170."""
    encs = []
    for vals in (np.arange(10), np.array([42]), np.zeros(100), np.full(100, 100),
                 np.array([2**31 - 1, 2**30, 0]), np.arange(53)):
        for dt in (W.INT32, W.UINT32, W.INT64, W.UINT64):
            v = vals.astype(NP_OF[dt])
            encs.append(E.encode(E.FixedBitWidth(), v, dt))
            encs.append(E.encode(E.FixedBitWidth(exact_bits=True), v, dt))
    for vals in (np.array([2, 3, 3]), np.array([1, 2, 2, 3, 3, 3, 4, 4, 4, 4])):
        encs.append(E.encode(E.RLE(), vals.astype(np.int32), W.INT32))
        encs.append(E.encode(E.RLE(lengths=E.Trivial(), values=E.Trivial()), vals.astype(np.int64), W.INT64))
    # float bit patterns: +-0.0 and four distinct NaN payloads must survive bit-exactly
    f32 = np.array([0x00000000, 0x80000000, 0x7FC00000, 0x7FA00000, 0x7FC00003, 0x7FC00005, 0x7FC00000], dtype=np.uint32)
    f64 = np.array([0, 1 << 63, 0x7FF8000000000000, 0x7FF4000000000000, 0x7FF8000000000003, 0x7FF8000000000005],
                   dtype=np.uint64)
    for spec in (E.RLE(), E.Trivial(), E.Dictionary(), E.FixedBitWidth(), E.MainlyConstant()):
        encs.append(E.encode(spec, np.repeat(f32, 3).view(np.float32), W.FLOAT))
        encs.append(E.encode(spec, np.repeat(f64, 3).view(np.float64), W.DOUBLE))
    i = np.arange(10000, dtype=np.uint64)
    encs.append(E.encode(E.FixedBitWidth(bit_width=10), ((i + i * i) % 1024).astype(np.uint32), W.UINT32))
    encs.append(E.encode(E.FixedBitWidth(bit_width=7), (100 - np.arange(27)).astype(np.uint32), W.UINT32))
    check_parity(oracle, encs)


def test_varint_row_count_prefix(gpu, oracle):
    rng = np.random.default_rng(3)
    encs = []
    for n in (1, 127, 128, 16384, 70000):
        v = rle_data(rng, W.INT64, n)
        encs.append(E.encode(E.RLE(), v, W.INT64, varint_rows=True))
        encs.append(E.encode(E.Dictionary(), v, W.INT64, varint_rows=True))
        m = rng.random(n) < 0.7
        encs.append(E.encode_nullable(E.Nullable(data=E.FixedBitWidth(), nulls=E.SparseBool()), v[: int(m.sum())], m,
                                      W.INT64, varint_rows=True))
    check_parity(oracle, encs, varint_rows=True)


def test_three_level_cascades(gpu, oracle):
    """Nullable -> RLE -> FBW, Nullable -> MainlyConstant -> {SparseBool -> FBW, Dictionary},
    Dictionary -> RLE indices, RLE -> Dictionary values."""
    rng = np.random.default_rng(21)
    encs, labels = [], []
    for n in (1000, 30000, 100003):
        mask = rng.random(n) < 0.85
        nn = int(mask.sum())
        runs = np.repeat(rng.integers(0, 1000, nn // 20 + 1), 20)[:nn].astype(np.int64)
        encs.append(E.encode_nullable(E.Nullable(data=E.RLE(), nulls=E.SparseBool()), runs, mask, W.INT64))
        labels.append(f"nullable-rle-fbw/n={n}")
        mc = np.where(rng.random(nn) < 0.9, 7, rng.integers(0, 50, nn)).astype(np.int32)
        encs.append(E.encode_nullable(E.Nullable(data=E.MainlyConstant(other=E.Dictionary()), nulls=E.RLE()), mc, mask, W.INT32))
        labels.append(f"nullable-mc-dict/n={n}")
        encs.append(E.encode(E.Dictionary(alphabet=E.FixedBitWidth(), indices=E.RLE()), np.sort(mc), W.INT32))
        labels.append(f"dict-rle/n={n}")
        encs.append(E.encode(E.RLE(lengths=E.Varint(), values=E.Dictionary()), runs, W.INT64))
        labels.append(f"rle-varint-dict/n={n}")
        fl = np.where(rng.random(nn) < 0.9, np.float32(1.5), rng.standard_normal(nn).astype(np.float32))
        encs.append(E.encode_nullable(E.Nullable(data=E.MainlyConstant(is_common=E.SparseBool(), other=E.Trivial()),
                                                 nulls=E.SparseBool()), fl, mask, W.FLOAT))
        labels.append(f"nullable-mc-float/n={n}")
    check_parity(oracle, encs, labels=labels)


def test_unaligned_chunk_starts(gpu, oracle):
    """Encodings packed back to back at every byte alignment."""
    rng = np.random.default_rng(33)
    encs = []
    for k in range(33):
        encs.append(E.encode(E.Trivial(), rng.integers(0, 255, k + 1).astype(np.uint8), W.UINT8))  # odd sizes shift the next start
        encs.append(E.encode(E.FixedBitWidth(exact_bits=True), rng.integers(0, 1 << 13, 777).astype(np.int64), W.INT64))
        encs.append(E.encode(E.Trivial(), rng.integers(0, 2**62, 333).astype(np.int64), W.INT64))
        encs.append(E.encode(E.Trivial(), rng.random(1001) < 0.5, W.BOOL))
    check_parity(oracle, encs)
