"""Synthetic tables of the benchmark configurations (encoded with
nimble_amd.encoder; compression off, as the reference's benchmark files are
written).  Every generator returns a list of (label, encoding bytes) chunks:
one column stream is split into chunks of at most 20 MiB of raw data, the cap
the reference writer applies (dwio/nimble/writer/VeloxWriterOptions.h:255-267).
"""
from __future__ import annotations

import numpy as np

from . import encoder as E
from . import wire as W

RAW_CHUNK_BYTES = 20 << 20

FBW_WIDTHS_32 = (1, 3, 7, 8, 13, 16, 24, 32)
FBW_WIDTHS_64 = FBW_WIDTHS_32 + (33, 40, 48, 57, 58, 59, 63, 64)


def _chunks(rows: int, elem_bytes: int):
    per = max(1, RAW_CHUNK_BYTES // elem_bytes)
    pos = 0
    while pos < rows:
        n = min(per, rows - pos)
        yield pos, n
        pos += n


def _uniform(rng, n: int, width: int) -> np.ndarray:
    if width >= 63:
        v = rng.integers(0, 2**63, n, dtype=np.uint64)
        if width == 64:
            v |= rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63)
        return v
    return rng.integers(0, 1 << width, n, dtype=np.uint64)


def run_length_values(rng, n: int, lo: int = 10, hi: int = 60, card: int = 1000, base: int = 12345) -> np.ndarray:
    """Runs of length U[lo,hi] of values U[0,card) + base: the RunLength pattern of
    dwio/nimble/encodings/benchmarks/BenchmarkUtils.h:113-131 with the baseline of
    FixedBitWidthBenchmark.cpp:40."""
    n_runs = n // lo + 1
    lens = rng.integers(lo, hi + 1, n_runs)
    vals = rng.integers(0, card, n_runs).astype(np.uint64) + np.uint64(base)
    return np.repeat(vals, lens)[:n]


def fbw_rle_table(rows: int = 10_000_000, seed: int = 42, compression=None):
    """Benchmark table "10M-row int32/int64 columns, fixed-bit-width + RLE cascades".
    compression = "zstd" | "lz4": every bit-packed payload is written compressed, the
    reference writer's default for leaves (compression/CompressionPolicy.h:115-143;
    here unconditionally, without its 2% gain test) -- the host layer's input only,
    the planner itself takes uncompressed payloads."""
    rng = np.random.default_rng(seed)
    chunks = []
    for dtype, widths, npt in ((W.INT32, FBW_WIDTHS_32, np.uint32), (W.INT64, FBW_WIDTHS_64, np.uint64)):
        es = W.elem_bytes(dtype)
        for w in widths:
            top = (1 << (8 * es)) - (1 << w) if w < 8 * es else 0
            base = int(rng.integers(0, min(top, 1 << 20) + 1)) if top > 0 else 0
            for pos, n in _chunks(rows, es):
                vals = (_uniform(rng, n, w) + np.uint64(base)).astype(npt)
                if n > 1:  # pin min/max so the header carries exactly this width and baseline
                    vals[0] = base
                    vals[1] = (base + ((1 << w) - 1)) & ((1 << (8 * es)) - 1)
                chunks.append((f"fbw/{W.DATA_TYPE_NAMES[dtype]}/w{w}/r{pos}",
                               E.encode(E.FixedBitWidth(bit_width=w, baseline=base, compression=compression), vals, dtype)))
        for exact in (False, True):
            for pos, n in _chunks(rows, es):
                vals = run_length_values(rng, n).astype(npt)
                spec = E.RLE(lengths=E.FixedBitWidth(exact_bits=exact, compression=compression),
                             values=E.FixedBitWidth(exact_bits=exact, compression=compression))
                chunks.append((f"rle/{W.DATA_TYPE_NAMES[dtype]}/{'exact' if exact else 'bytes'}/r{pos}",
                               E.encode(spec, vals, dtype)))
    return chunks


def cpu_reference_column(rows: int = 1_000_000, seed: int = 42):
    """Benchmark column "1M-row int64 column (bit-packed + RLE)": one chunk, byte-rounded and exact widths."""
    rng = np.random.default_rng(seed)
    vals = run_length_values(rng, rows).astype(np.uint64)
    out = []
    for exact in (False, True):
        spec = E.RLE(lengths=E.FixedBitWidth(exact_bits=exact), values=E.FixedBitWidth(exact_bits=exact))
        out.append((f"rle/int64/{'exact' if exact else 'bytes'}", E.encode(spec, vals, W.INT64)))
    return out, vals


def pack_chunks(chunks):
    """Concatenates chunk encodings into one host buffer; returns (buffer, [(offset,size)])."""
    offs, pos = [], 0
    for _, enc in chunks:
        offs.append((pos, len(enc)))
        pos += len(enc)
    buf = np.empty(pos, dtype=np.uint8)
    for (o, s), (_, enc) in zip(offs, chunks):
        buf[o:o + s] = np.frombuffer(enc, dtype=np.uint8)
    return buf, offs


# --------------------------------------------------------------------------
# Dictionary-encoded string column with a nullable wrapper
# --------------------------------------------------------------------------
def random_alphabet(rng, size: int, max_len: int = 50) -> E.StringColumn:
    """`size` distinct strings of length U[0,max_len) of random bytes (the string
    pattern of dwio/nimble/common/tests/TestUtils.h:147-163); duplicates are
    dropped, so small alphabets may come out slightly smaller."""
    lengths = rng.integers(0, max_len, size).astype(np.uint32)
    blob = rng.integers(0, 256, int(lengths.sum()), dtype=np.uint8)
    items = E.StringColumn(lengths, blob).to_list()
    return E.StringColumn.from_list(list(dict.fromkeys(items)))


def dict_string_table(rows: int = 10_000_000, seed: int = 7, alphabets=(64, 4096, 65536), null_fracs=(0.0, 0.05, 0.5),
                      rows_per_chunk: int = 700_000):
    """Benchmark table "Dictionary-encoded string column (offsets + bytes) with
    nullable wrapper": Nullable{Dictionary{Trivial<string>{FBW lengths}, FBW
    indices}, SparseBool{FBW} | Trivial<bool>} -- the shape of docs/Encodings.md:39-49
    wrapped in Nullable.  ~20 MiB of raw strings per chunk."""
    rng = np.random.default_rng(seed)
    chunks = []
    dspec = E.Dictionary(alphabet=E.Trivial(lengths=E.FixedBitWidth()), indices=E.FixedBitWidth())
    for a in alphabets:
        alpha = random_alphabet(rng, a)
        for nf in null_fracs:
            for pos in range(0, rows, rows_per_chunk):
                n = min(rows_per_chunk, rows - pos)
                mask = rng.random(n) >= nf if nf > 0 else np.ones(n, dtype=bool)
                idx = rng.integers(0, len(alpha), int(mask.sum()))
                enc = E.encode_dictionary_from_indices(dspec, alpha, idx, W.STRING)
                if nf > 0:
                    nulls = E.SparseBool() if nf < 0.25 else E.Trivial()
                    enc = E.nullable_wrap(enc, mask, nulls, W.STRING)
                chunks.append((f"dict/string/a{a}/null{nf}/r{pos}", enc))
    return chunks


# --------------------------------------------------------------------------
# Mixed-type 100-column table, 3-level cascades
# --------------------------------------------------------------------------
def _nullable_mask(rng, n, frac):
    return rng.random(n) >= frac


def mixed_table_stripe(stripe: int = 0, rows: int = 1_250_000, scale: float = 1.0):
    """One stripe of the "Mixed-type 100-column table, 3-level cascaded encodings":
    25 int32 FBW, 25 int64 Nullable>RLE>FBW, 20 float32/float64
    Nullable>MainlyConstant>{SparseBool>FBW, Trivial}, 15 string
    Nullable>Dictionary>..., 10 bool RLE/SparseBool, 5 Constant.  Seeds are
    1000*stripe + column.  `scale` < 1 keeps the mix but fewer columns."""
    chunks = []
    col = 0

    def count(k):
        return max(1, int(round(k * scale)))

    def rng_for():
        return np.random.default_rng(1000 * stripe + col)

    for i in range(count(25)):
        rng = rng_for()
        w = 1 + (i * 5) % 31
        vals = rng.integers(0, 1 << w, rows, dtype=np.uint64).astype(np.uint32)
        chunks.append((f"s{stripe}/c{col}/int32/fbw{w}", E.encode(E.FixedBitWidth(exact_bits=bool(i & 1)), vals, W.INT32)))
        col += 1
    for i in range(count(25)):
        rng = rng_for()
        mask = _nullable_mask(rng, rows, (0.02, 0.2, 0.5)[i % 3])
        vals = run_length_values(rng, int(mask.sum()), card=1 << (8 + i % 20)).astype(np.uint64)
        spec = E.Nullable(data=E.RLE(lengths=E.FixedBitWidth(), values=E.FixedBitWidth(exact_bits=True)),
                          nulls=E.SparseBool() if i % 3 == 0 else E.Trivial())
        chunks.append((f"s{stripe}/c{col}/int64/nullable-rle-fbw", E.encode_nullable(spec, vals, mask, W.INT64)))
        col += 1
    for i in range(count(20)):
        rng = rng_for()
        dt, npt = ((W.FLOAT, np.float32), (W.DOUBLE, np.float64))[i & 1]
        mask = _nullable_mask(rng, rows, (0.0, 0.1, 0.4)[i % 3])
        nn = int(mask.sum())
        is_common = rng.random(nn) < (0.95, 0.7)[(i >> 1) & 1]
        other = rng.standard_normal(int((~is_common).sum())).astype(npt)
        spec = E.MainlyConstant(is_common=E.SparseBool(indices=E.FixedBitWidth()) if (i >> 1) & 1 == 0 else E.Trivial(), other=E.Trivial())
        data = E.mainly_constant_from_parts(spec, is_common, other, npt(0.0), dt)
        chunks.append((f"s{stripe}/c{col}/{W.DATA_TYPE_NAMES[dt]}/nullable-mainlyconstant",
                       E.nullable_wrap(data, mask, E.SparseBool() if i % 3 == 1 else E.Trivial(), dt)))
        col += 1
    for i in range(count(15)):
        rng = rng_for()
        alpha = random_alphabet(rng, (16, 300, 5000)[i % 3])
        mask = _nullable_mask(rng, rows, (0.05, 0.3)[i & 1])
        idx = rng.integers(0, len(alpha), int(mask.sum()))
        dspec = E.Dictionary(alphabet=E.Trivial(lengths=E.FixedBitWidth()), indices=E.FixedBitWidth(exact_bits=bool(i & 1)))
        data = E.encode_dictionary_from_indices(dspec, alpha, idx, W.STRING)
        chunks.append((f"s{stripe}/c{col}/string/nullable-dictionary", E.nullable_wrap(data, mask, E.SparseBool(), W.STRING)))
        col += 1
    for i in range(count(10)):
        rng = rng_for()
        if i & 1:
            b = rng.random(rows) < 0.03
            chunks.append((f"s{stripe}/c{col}/bool/sparse", E.encode(E.SparseBool(), b, W.BOOL)))
        else:
            b = np.repeat(np.arange(rows // 20 + 1) & 1, rng.integers(1, 40, rows // 20 + 1))[:rows].astype(bool)
            b = np.resize(b, rows)
            chunks.append((f"s{stripe}/c{col}/bool/rle", E.encode(E.RLE(lengths=E.FixedBitWidth()), b, W.BOOL)))
        col += 1
    for i in range(count(5)):
        rng = rng_for()
        dt = (W.INT32, W.INT64, W.DOUBLE, W.FLOAT, W.INT16)[i % 5]
        v = np.full(rows, rng.integers(0, 100), dtype=W.physical_dtype(dt))
        chunks.append((f"s{stripe}/c{col}/{W.DATA_TYPE_NAMES[dt]}/constant", E.encode(E.Constant(), v, dt)))
        col += 1
    return chunks


# --------------------------------------------------------------------------
# Wide ML training table: sparse / nullable float features
# --------------------------------------------------------------------------
def wide_table_stripe(stripe: int = 0, columns: int = 5000, rows: int = 100_000):
    """One stripe of the "Wide ML training table (5,000 feature columns,
    sparse/nullable floats)": per column a non-null fraction from {1,5,20,80}%;
    non-null values 90% one common value, else random bits; cascade
    Nullable<float>{MainlyConstant<float> | Trivial<float>, SparseBool | Trivial<bool>}.
    One chunk per column (2 MiB chunk cap of wide schemas,
    dwio/nimble/writer/VeloxWriterOptions.h:263-267, is not reached at 100K rows)."""
    chunks = []
    for c in range(columns):
        rng = np.random.default_rng(1_000_003 * stripe + c)
        frac = (0.01, 0.05, 0.2, 0.8)[int(rng.integers(0, 4))]
        mask = rng.random(rows) < frac
        nn = int(mask.sum())
        if c % 4 == 3:
            vals = rng.integers(0, 1 << 32, nn, dtype=np.uint64).astype(np.uint32)
            data = E.encode(E.Trivial(), vals, W.FLOAT)
        else:
            is_common = rng.random(nn) < 0.9
            if nn:
                is_common[0] = True
            other = rng.integers(0, 1 << 32, int((~is_common).sum()), dtype=np.uint64).astype(np.uint32)
            spec = E.MainlyConstant(is_common=E.SparseBool(indices=E.FixedBitWidth()), other=E.Trivial())
            data = E.mainly_constant_from_parts(spec, is_common, other, np.float32(1.0), W.FLOAT) if nn else \
                E.encode(E.Trivial(), other, W.FLOAT)
        nulls = E.SparseBool(indices=E.FixedBitWidth()) if frac < 0.5 else E.Trivial()
        chunks.append((f"s{stripe}/f{c}", E.nullable_wrap(data, mask, nulls, W.FLOAT)))
    return chunks


WORKLOADS = {
    "fbw_rle": lambda rows, seed: fbw_rle_table(rows, seed),
    "fbw_rle_zstd": lambda rows, seed: fbw_rle_table(rows, seed, compression="zstd"),
    "dict_string": lambda rows, seed: dict_string_table(rows, seed),
    "mixed": lambda rows, seed: mixed_table_stripe(seed, rows),
    "wide": lambda rows, seed: wide_table_stripe(seed, 5000, rows),
}
