#!/usr/bin/env python3
"""Writes the seed corpus for tools/plan_fuzz.cpp (host-only, no GPU):
  u32 count, then per entry: u8 kind, u8 varint_rows, u32 size, bytes
kind 0 = one top-level encoding (random cascades of every encoding the planner
knows, some with Zstd / LZ4 leaves), kind 1 = a whole stream (chunk headers,
some chunks compressed), kind 2 = a Nimble file.
Usage: python tools/make_corpus.py OUT [n_chunks]"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from nimble_amd import encoder as E, wire as W  # noqa: E402
from fuzz_util import random_chunk  # noqa: E402
from tablet_writer import write_tablet  # noqa: E402


def main():
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    rng = np.random.default_rng(2024)
    entries = []
    chunks = []
    for _ in range(n):
        _, enc = random_chunk(rng, max_rows=400)
        chunks.append(enc)
        entries.append((0, 0, enc))
    # varint row counts and compressed leaves
    ints = np.repeat(rng.integers(0, 50, 40), 10).astype(np.int64)
    bools = rng.random(400) < 0.1
    try:
        import test_inflate_cpu as TI

        for codec in ("zstd", "lz4"):
            for spec, values, dtype in TI._cases(rng):
                for vr in (0, 1):
                    entries.append((0, vr, E.encode(TI._with_compression(spec, codec), values, dtype, bool(vr))))
    except Exception as e:  # pyarrow (the test's compressor) missing: plain cascades only
        print("no compressed seeds:", e)
    for vr in (0, 1):
        entries.append((0, vr, E.encode(E.RLE(), ints, W.INT64, bool(vr))))
        entries.append((0, vr, E.encode(E.SparseBool(), bools, W.BOOL, bool(vr))))
    # streams
    for k in range(0, min(len(chunks), 60), 3):
        entries.append((1, 0, E.stream(chunks[k:k + 3])))
    # files
    for s in range(3):
        stripes = []
        for i in range(1 + s):
            m = 50 + 7 * i
            a = rng.integers(0, 9, m).astype(np.int32)
            stripes.append((m, [E.stream([E.encode(E.FixedBitWidth(), a, W.INT32)]), b"", E.stream([E.encode(E.RLE(), a, W.INT32)])]))
        entries.append((2, 0, write_tablet(stripes, stripes_per_group=2, optional={"columnar.schema": b"abc"})))
    with open(out, "wb") as f:
        f.write(struct.pack("<I", len(entries)))
        for kind, vr, data in entries:
            f.write(struct.pack("<BBI", kind, vr, len(data)))
            f.write(data)
    print(f"{len(entries)} entries, {sum(len(d) for _, _, d in entries)} bytes -> {out}")


if __name__ == "__main__":
    main()
