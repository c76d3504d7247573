#!/usr/bin/env python3
"""Per-stripe timeline of a traced run of tools/host_bench.cpp: when the copies in of
each batch ran and when its copy back did (copy-engine copies from
<prefix>_memory_copy_trace.csv, copy-back kernels from <prefix>_kernel_trace.csv), as
written by  rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d <dir> -- tools/bin/host_bench ...
usage: copy_timeline.py <dir>/<host>/<pid>   (the common prefix of the two files)"""
import csv
import sys

prefix = sys.argv[1]
copies = list(csv.DictReader(open(prefix + "_memory_copy_trace.csv")))
kernels = list(csv.DictReader(open(prefix + "_kernel_trace.csv")))
t0 = min(int(r["Start_Timestamp"]) for r in copies)


def ms(v):
    return (int(v) - t0) / 1e6


events = []
for r in copies:
    events.append((ms(r["Start_Timestamp"]), ms(r["End_Timestamp"]), "in " if "HOST_TO_DEVICE" in r["Direction"] else "out", r["Stream_Id"]))
for r in kernels:
    if "copy_to_host" in r["Kernel_Name"]:
        events.append((ms(r["Start_Timestamp"]), ms(r["End_Timestamp"]), "out", r["Stream_Id"] + " (kernel)"))
events.sort()
runs = []  # consecutive transfers of one direction on one stream, gaps under 3 ms
for s, e, d, st in events:
    if e - s < 0.05:  # descriptor tables, status words
        continue
    last = next((r for r in reversed(runs) if r[2] == d), None)
    if last is not None and last[3] == st and s - last[1] < 3.0:
        last[1] = max(last[1], e)
        last[4] += 1
        last[5] += e - s
    else:
        runs.append([s, e, d, st, 1, e - s])
print("dir  stream            start ms    end ms   busy ms  transfers")
for s, e, d, st, n, busy in runs:
    print(f"{d}  {st:<16} {s:9.2f} {e:9.2f} {busy:9.2f}  {n}")
