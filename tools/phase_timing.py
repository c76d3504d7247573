#!/usr/bin/env python3
"""Where a rank-scatter tile's lifetime goes (tuning aid, needs a GPU and a
library built with -DNBL_PHASE_TIMING, see tools/build_phase_lib.sh):
  NIMBLE_AMD_LIB=tools/bin/libnimble_amd_phase.so python tools/phase_timing.py [nested|plain] [fraction] [columns]
Decodes one group of wide-table columns with one launch per operation and
prints the mean cycles between the marks of thread 0 of every tile."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from nimble_amd import capi, encoder as E, wire as W, workloads  # noqa: E402
from nimble_amd.decoder import BatchDecoder, Plan  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "nested"
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
columns = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
rows = 100_000


def column(c):
    rng = np.random.default_rng(c)
    mask = rng.random(rows) < frac
    if kind == "sparse":  # a root SparseBool column: marks 1-3 of the bitmap step survive the bit -> byte launch behind it
        return (f"b{c}", E.encode(E.SparseBool(indices=E.FixedBitWidth()), mask, W.BOOL))
    nn = int(mask.sum())
    if kind != "nested":
        data = E.encode(E.Trivial(), rng.integers(0, 1 << 32, nn, dtype=np.uint64).astype(np.uint32), W.FLOAT)
    else:
        is_common = rng.random(nn) < 0.9
        is_common[0] = True
        other = rng.integers(0, 1 << 32, int((~is_common).sum()), dtype=np.uint64).astype(np.uint32)
        spec = E.MainlyConstant(is_common=E.SparseBool(indices=E.FixedBitWidth()), other=E.Trivial())
        data = E.mainly_constant_from_parts(spec, is_common, other, np.float32(1.0), W.FLOAT)
    nulls = E.SparseBool(indices=E.FixedBitWidth()) if frac < 0.5 else E.Trivial()
    return (f"f{c}", E.nullable_wrap(data, mask, nulls, W.FLOAT))


if kind == "wide":  # the benchmark table itself: read the bitmap step's records with PHASE_LAUNCH=<index of k_sparse_bits>
    buf, offs = workloads.pack_chunks(workloads.wide_table_stripe(0, columns, rows))
else:
    distinct = [column(c) for c in range(25)]
    buf, offs = workloads.pack_chunks([distinct[i % 25] for i in range(columns)])
plan = Plan(buf, offs, per_kernel=True)
dec = BatchDecoder(plan)
dec.stage()
for _ in range(2):
    dec.decode()
dec.status()
lib = capi.lib()
lib.nbl_phase_read.argtypes = [C.c_void_p, C.c_uint32]
dec.profile_begin(1)
dec.decode_profiled(0)
ms = dec.profile_end()
dec.status()
ll = plan.launches()
which = int(os.environ.get("PHASE_LAUNCH", "-1"))  # launch whose records are read (the last one by default)
tiles = min(ll[which][2], 1 << 18)
rec = np.zeros((tiles, 8), dtype=np.uint32)
assert lib.nbl_phase_read(rec.ctypes.data, tiles) == 0
out = rec.astype(np.float64).sum(axis=0)
names = ["tile entry", "descriptor + partials (sparse: desc, search, zeroed tile)", "outer word + inner partials", "own staged values issued+arrived",
         "scan + LDS writes", "barrier", "row loop (stores issued)", "stores acknowledged"]
for i, (l, m) in enumerate(zip(ll, ms)):
    print(f"  launch {i}: {l[0]:24s} {l[2]:7d} tiles {m * 1e3:8.1f} us")
print(f"{kind} {frac}: launch {ll[which][0]} {ms[which] * 1e3:.1f} us, {ll[which][2]} tiles, {ms[which] * 1e6 / ll[which][2]:.2f} ns/tile")
tot = 0
for i, n in enumerate(names):
    c = out[i] / tiles
    tot += c
    print(f"  {n:36s} {c:9.0f} cycles")
print(f"  {'sum':36s} {tot:9.0f} cycles")
life = rec.astype(np.float64).sum(axis=1)
print(f"  tile lifetimes: sum {life.sum() / 2.4e3:.0f} us at 2.4 GHz = {life.sum() / 2.4e3 / 2048:.1f} us over 2,048 workgroup slots; longest {life.max() / 2.4e3:.1f} us; quartiles "
      + " / ".join(f"{np.percentile(life, q) / 2.4e3:.1f}" for q in (25, 50, 75)) + " us")
