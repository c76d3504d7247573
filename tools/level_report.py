#!/usr/bin/env python3
"""Per-level view of a benchmark table (tuning aid, needs a GPU): the duration of
every fused level launch next to the operations that level holds, timed one
launch per (level, operation).  Usage: python tools/level_report.py <workload> [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from nimble_amd import workloads  # noqa: E402
from nimble_amd.decoder import BatchDecoder, Plan  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mixed"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else {"fbw_rle": 10_000_000, "dict_string": 10_000_000, "mixed": 1_250_000, "wide": 100_000}[name]
chunks = workloads.WORKLOADS[name](rows, 42)
buf, offs = workloads.pack_chunks(chunks)


def timed(per_kernel):
    plan = Plan(buf, offs, per_kernel=per_kernel)
    dec = BatchDecoder(plan)
    dec.stage()
    for _ in range(3):
        dec.decode()
    dec.status()
    steps = 10
    dec.profile_begin(steps)
    for s in range(steps):
        dec.decode_profiled(s)
    ms = dec.profile_end()
    out = list(zip(plan.launches(), ms))
    plan.close()
    return out


fused = timed(False)
split = timed(True)
for (lname, es, nt, lvl, alg), ms in fused:
    print(f"level {lvl}: {lname:16s} tiles {nt:7d}  {ms * 1e3:7.1f} us")
    for (kname, kes, knt, klvl, kalg), kms in split:
        if klvl == lvl:
            print(f"      {kname:24s} tiles {knt:7d}  {kms * 1e3:7.1f} us")
print(f"fused sum {sum(m for _, m in fused) * 1e3:.1f} us, per-operation sum {sum(m for _, m in split) * 1e3:.1f} us")
