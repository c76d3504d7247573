#!/usr/bin/env python3
"""Last-level time of wide-table columns per (non-null fraction, values kind)
group: 1250 columns of 100,000 rows each (tuning aid, needs a GPU).
Usage: python tools/wide_breakdown.py [columns] [nested|plain] [fraction]
(the last two restrict the run to one group, e.g. under a counter pass)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from nimble_amd import encoder as E  # noqa: E402
from nimble_amd import wire as W  # noqa: E402
from nimble_amd import workloads  # noqa: E402
from nimble_amd.decoder import BatchDecoder, Plan  # noqa: E402

columns = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
only_kind = sys.argv[2] if len(sys.argv) > 2 else None
only_frac = float(sys.argv[3]) if len(sys.argv) > 3 else None
rows = 100_000


def column(c, frac, nested):
    rng = np.random.default_rng(c)
    mask = rng.random(rows) < frac
    nn = int(mask.sum())
    if not nested:
        vals = rng.integers(0, 1 << 32, nn, dtype=np.uint64).astype(np.uint32)
        data = E.encode(E.Trivial(), vals, W.FLOAT)
    else:
        is_common = rng.random(nn) < 0.9
        is_common[0] = True
        other = rng.integers(0, 1 << 32, int((~is_common).sum()), dtype=np.uint64).astype(np.uint32)
        spec = E.MainlyConstant(is_common=E.SparseBool(indices=E.FixedBitWidth()), other=E.Trivial())
        data = E.mainly_constant_from_parts(spec, is_common, other, np.float32(1.0), W.FLOAT)
    nulls = E.SparseBool(indices=E.FixedBitWidth()) if frac < 0.5 else E.Trivial()
    return (f"f{c}", E.nullable_wrap(data, mask, nulls, W.FLOAT))


for nested in (False, True):
    for frac in (0.01, 0.05, 0.2, 0.8):
        if (only_kind and only_kind != ("nested" if nested else "plain")) or (only_frac is not None and only_frac != frac):
            continue
        # a few distinct columns repeated: generation time, not decode time, is what this saves
        distinct = [column(c, frac, nested) for c in range(25)]
        chunks = [distinct[i % 25] for i in range(columns)]
        buf, offs = workloads.pack_chunks(chunks)
        plan = Plan(buf, offs)
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
        ll = plan.launches()
        alg = ll[-1][4]
        print("%-6s non-null %.2f: last level %7.1f us  tiles %6d  %5.2f ns/tile  %6.0f GB/s alg (%.3f of peak)   earlier levels %5.1f us" %
              ("nested" if nested else "plain", frac, ms[-1] * 1e3, ll[-1][2], ms[-1] * 1e6 / ll[-1][2], alg / ms[-1] / 1e6,
               alg / ms[-1] / 1e6 / 8000, sum(ms[:-1]) * 1e3), flush=True)
        del dec
        plan.close()
