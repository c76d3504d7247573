#!/usr/bin/env python3
"""Writes a benchmark table as <prefix>.bin (chunk encodings back to back) and
<prefix>.off (uint64 offset,size pairs) for tools/host_bench.cpp.
usage: dump_workload.py fbw_rle|fbw_rle_zstd|dict_string|mixed|wide <prefix> [rows]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nimble_amd import workloads as WL  # noqa: E402

name, prefix = sys.argv[1], sys.argv[2]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else {"fbw_rle": 10_000_000, "fbw_rle_zstd": 10_000_000, "dict_string": 10_000_000, "mixed": 1_250_000, "wide": 100_000}[name]
chunks = WL.WORKLOADS[name](rows, 42)
buf, offs = WL.pack_chunks(chunks)
buf.tofile(prefix + ".bin")
np.array(offs, dtype=np.uint64).tofile(prefix + ".off")
print(f"{name}: {len(chunks)} chunks, {buf.size} bytes")
