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


def fbw_rle_table(rows: int = 10_000_000, seed: int = 42):
    """Benchmark table "10M-row int32/int64 columns, fixed-bit-width + RLE cascades"."""
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
                               E.encode(E.FixedBitWidth(bit_width=w, baseline=base), vals, dtype)))
        for exact in (False, True):
            for pos, n in _chunks(rows, es):
                vals = run_length_values(rng, n).astype(npt)
                spec = E.RLE(lengths=E.FixedBitWidth(exact_bits=exact), values=E.FixedBitWidth(exact_bits=exact))
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
