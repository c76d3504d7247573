#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV and prints, for the repeating sequence of
k_level launches (one decode step), the mean duration of every launch of the
step and the mean gap in front of it.  Usage: trace_levels.py <kernel_trace.csv>"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
lv = [r for r in rows if "k_level" in r[2]]
# a step ends with k_level<true>
steps, cur = [], []
for r in lv:
    cur.append(r)
    if "true" in r[2] or "Lb1" in r[2]:
        steps.append(cur)
        cur = []
n = max(set(len(s) for s in steps), key=[len(s) for s in steps].count)
steps = [s for s in steps if len(s) == n][-6:]
print("launches per step %d, steps averaged %d" % (n, len(steps)))
for i in range(n):
    dur = sum(s[i][1] - s[i][0] for s in steps) / len(steps) / 1000
    gap = sum((s[i][0] - s[i - 1][1]) for s in steps) / len(steps) / 1000 if i else 0.0
    print("launch %d: grid %8d  %8.1f us   gap before %6.1f us" % (i, steps[0][i][3], dur, gap))
print("step span %.1f us" % (sum(s[-1][1] - s[0][0] for s in steps) / len(steps) / 1000))
