"""Checks shared by the oracle and the GPU host layer for the selective
reader's ChunkedDecoder bulk entry points (velox/selective/ChunkedDecoder.h:78-90,
206-268; tests: velox/selective/tests/ChunkedDecoderTest.cpp): both objects run
the same script of reads over the same multi-chunk streams and must leave the
same bytes in the caller's buffers."""
from __future__ import annotations

import numpy as np

from nimble_amd import encoder as E
from nimble_amd import wire as W

from util import phys, random_data


def _bitmap(mask):
    """uint64 words (as the reader passes them), LSB first, with a spare word."""
    n = len(mask)
    b = np.zeros((n + 63) // 64 * 8 + 8, dtype=np.uint8)
    packed = np.packbits(np.asarray(mask, dtype=np.uint8), bitorder="little")
    b[: packed.size] = packed
    return b


def value_stream(rng, dtype, nullable_chunks=True, max_chunks=6, max_rows=400):
    """(stream bytes, values incl. zeros at nulls, non-null mask)"""
    chunks, values, masks = [], [], []
    for _ in range(int(rng.integers(1, max_chunks + 1))):
        n = int(rng.integers(1, max_rows + 1))
        nullable = nullable_chunks and bool(rng.integers(0, 2))
        mask = (rng.random(n) < 0.7) if nullable else np.ones(n, dtype=bool)
        v = random_data(rng, dtype, int(mask.sum()))
        if nullable:
            chunks.append(E.encode_nullable(E.Nullable(data=E.FixedBitWidth(), nulls=E.Trivial()), v, mask, dtype))
        else:
            chunks.append(E.encode([E.FixedBitWidth(), E.Trivial(), E.RLE()][int(rng.integers(0, 3))], v, dtype))
        full = np.zeros(n, dtype=W.physical_dtype(dtype))
        full[mask] = phys(v, dtype)
        values.append(full)
        masks.append(mask)
    return E.stream(chunks), np.concatenate(values), np.concatenate(masks)


def bool_stream(rng, max_chunks=6, max_rows=500):
    chunks, values = [], []
    for _ in range(int(rng.integers(1, max_chunks + 1))):
        n = int(rng.integers(1, max_rows + 1))
        b = rng.random(n) < (0.05 if rng.integers(0, 2) else 0.5)
        chunks.append(E.encode([E.Trivial(), E.SparseBool(), E.RLE()][int(rng.integers(0, 3))], b, W.BOOL))
        values.append(b)
    return E.stream(chunks), np.concatenate(values)


def script(rng, total):
    """Reads of random sizes covering `total` stream values, each with or
    without incoming nulls: [(count, incoming mask or None)]."""
    out, left = [], total
    while left > 0:
        n = int(rng.integers(1, left + 1))
        if rng.random() < 0.5:
            out.append((n, None))
        else:  # n stream values spread over a longer read range
            gaps = int(rng.integers(0, n + 5))
            mask = np.zeros(n + gaps, dtype=bool)
            mask[rng.choice(n + gaps, size=n, replace=False)] = True
            out.append((n + gaps, mask))
        left -= n
    return out


def run_decode_nullable(dec, dtype, reads):
    """Outputs of every read: (nulls bitmap bytes, data) with data zero-initialised."""
    results = []
    for count, mask in reads:
        nulls = np.zeros((count + 63) // 64 * 8 + 8, dtype=np.uint8)
        data = np.zeros(count, dtype=W.physical_dtype(dtype))
        dec.decode_nullable(nulls, data, count, _bitmap(mask) if mask is not None else None)
        results.append((nulls[: (count + 7) // 8].copy(), data))
    return results


def run_next_bools(dec, reads):
    results = []
    for count, mask in reads:
        words = np.zeros((count + 63) // 64 + 1, dtype=np.uint64)
        dec.next_bools(words, count, _bitmap(mask) if mask is not None else None)
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:count].copy()
        results.append((bits, None if dec.preloaded_values is None else dec.preloaded_values.copy()))
    return results


def run_next_indices(dec, reads):
    results = []
    for count, mask in reads:
        data = np.zeros(count, dtype=np.int32)
        dec.next_indices(data, count, _bitmap(mask) if mask is not None else None)
        results.append(data)
    return results


def expected_decode_nullable(values, nn_mask, reads):
    """What the contract says, from the plain column: stream value k lands on the
    k-th set position of the incoming mask; a slot is non-null when the mask has it
    AND the stream value is non-null."""
    out, pos = [], 0
    for count, mask in reads:
        m = np.ones(count, dtype=bool) if mask is None else mask
        k = int(m.sum())
        v, nn = values[pos:pos + k], nn_mask[pos:pos + k]
        pos += k
        slots = np.flatnonzero(m)
        valid = np.zeros(count, dtype=bool)
        valid[slots] = nn
        out.append((valid, slots[nn], v[nn]))
    return out
