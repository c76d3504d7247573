#!/usr/bin/env python3
"""Last-level time of the dictionary-string table per (alphabet size, null
fraction) group (tuning aid, needs a GPU).  Usage: python tools/dict_breakdown.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from nimble_amd import workloads  # noqa: E402
from nimble_amd.decoder import BatchDecoder, Plan  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for a in (64, 4096, 65536):
    for nf in (0.0, 0.05, 0.5):
        chunks = workloads.dict_string_table(rows, 7, alphabets=(a,), null_fracs=(nf,))
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
        print("alphabet %6d nulls %.2f: last level %7.1f us  tiles %6d  %6.0f GB/s alg (%.3f of peak)   earlier levels %5.1f us" %
              (a, nf, ms[-1] * 1e3, ll[-1][2], alg / ms[-1] / 1e6, alg / ms[-1] / 1e6 / 8000, sum(ms[:-1]) * 1e3), flush=True)
        del dec
        plan.close()
