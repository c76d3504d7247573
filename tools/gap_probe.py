#!/usr/bin/env python3
"""Step time of the headline batch with and without the per-launch events (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nimble_amd import workloads
from nimble_amd.decoder import BatchDecoder, Plan

chunks = workloads.fbw_rle_table(10_000_000, 42)
buf, offs = workloads.pack_chunks(chunks)
plan = Plan(buf, offs)
dec = BatchDecoder(plan)
dec.stage()
for _ in range(5):
    dec.decode()
torch.cuda.synchronize()
for name, fn in (("plain", lambda s: dec.decode()), ("events", None)):
    steps = 50
    if fn is None:
        dec.profile_begin(steps)
        fn = lambda s: dec.decode_profiled(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        fn(s)
    torch.cuda.synchronize()
    print(name, "ms/step", (time.perf_counter() - t0) / steps * 1e3)
# graph
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    dec.stream = s
    dec.decode()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        dec.decode()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) / 50 * 1e3)
