#!/usr/bin/env python3
"""Per-width throughput of the root FixedBitWidth / Trivial kernels (tuning aid).
Usage: python tools/tune_fbw.py [rows]   (needs a GPU)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nimble_amd import encoder as E, wire as W  # noqa: E402
from nimble_amd.decoder import BatchDecoder, Plan  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
misalign = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(0)
per_chunk = {4: 5_000_000, 8: 2_500_000}
print(f"rows={rows} misalign={misalign}")
for dtype, widths in ((W.INT32, (1, 3, 7, 8, 13, 16, 24, 32)), (W.INT64, (1, 3, 7, 8, 13, 16, 24, 32, 33, 40, 48, 57, 58, 59, 63, 64))):
    es = W.elem_bytes(dtype)
    npt = np.uint32 if es == 4 else np.uint64
    for w in widths + ("raw", "const"):
        n = per_chunk[es]
        if w == "const":
            enc = E.encode(E.Constant(), np.full(n, 7, dtype=npt), dtype)
        elif w == "raw":
            enc = E.encode(E.Trivial(), rng.integers(0, 2**31, n, dtype=np.uint64).astype(npt), dtype)
        else:
            v = rng.integers(0, 1 << min(w, 63), n, dtype=np.uint64).astype(npt)
            enc = E.encode(E.FixedBitWidth(bit_width=w, baseline=0), v, dtype)
        k = rows // n
        pad = (-len(enc)) % 16
        blob = (enc + bytes(pad)) if misalign == 0 else (enc + bytes(pad + misalign))
        buf = np.frombuffer(blob * k, dtype=np.uint8)
        offs = [(i * len(blob), len(enc)) for i in range(k)]
        plan = Plan(buf, offs, per_kernel=True)
        dec = BatchDecoder(plan)
        dec.stage()
        for _ in range(3):
            dec.decode()
        steps = 10
        dec.profile_begin(steps)
        for s in range(steps):
            dec.decode_profiled(s)
        ms = dec.profile_end()
        alg = plan.info.encoded_bytes + plan.info.decoded_bytes
        t = sum(ms) * 1e-3
        print(f"es={es} w={w!s:>3}  {alg / t / 1e9:8.1f} GB/s alg   {plan.info.decoded_bytes / t / 1e9:8.1f} GB/s out   {sum(ms)*1e3:8.1f} us")
        plan.close()
        del dec
