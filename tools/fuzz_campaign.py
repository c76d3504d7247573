#!/usr/bin/env python3
"""Longer run of the random-cascade parity check (needs a GPU): seeds
[first, first + count), each a batch of random cascades decoded on the GPU and
compared with the oracle.  Usage: python tools/fuzz_campaign.py [first] [count]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fuzz_util import random_chunk  # noqa: E402
from oracle import oracle as O  # noqa: E402
from util import check_parity  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
chunks = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    big = seed % 10 == 0
    made = [random_chunk(rng, max_rows=(300_000 if big else 8000), max_depth=4) for _ in range(10 if big else 60)]
    try:
        check_parity(O, [e for _, e in made], labels=[l for l, _ in made])
    except Exception as e:  # report and go on: one run should list every failing seed
        print(f"seed {seed}: FAILED {type(e).__name__}: {str(e)[:400]}", flush=True)
        continue
    chunks += len(made)
    if seed % 10 == 9:
        print(f"seed {seed} ok, {chunks} chunks, {time.time() - t0:.0f} s", flush=True)
print(f"done: {chunks} chunks matched in {time.time() - t0:.0f} s")
