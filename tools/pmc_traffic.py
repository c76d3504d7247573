#!/usr/bin/env python3
"""Turns two rocprofv3 counter passes over bench.py (one `--pmc FETCH_SIZE`, one
`--pmc WRITE_SIZE`, each with `--kernel-trace --output-format csv`) into
per-kernel HBM traffic per launch.

Corrections applied (MI355X / gfx950):
  * both counters are reported in KiB;
  * FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes,
    so it is doubled.  The pure-copy kernel of the same run (k_materialize_raw,
    which reads exactly as many bytes as it writes) is the calibration: the
    script prints fetch*2/write for it, which must be close to 1;
  * WRITE_SIZE is exact for 16-byte-per-lane streaming stores.

usage: pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json [rows] [workload]
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"nbl::(k_\w+(?:<\w+>)?)", r["Kernel_Name"])
        if not m:
            continue
        agg[m.group(1)].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    rows = int(sys.argv[4]) if len(sys.argv) > 4 else 10_000_000
    workload = sys.argv[5] if len(sys.argv) > 5 else "fbw_rle"
    out = {"workload": workload, "rows": rows, "unit": "bytes per launch", "fetch_correction": 2.0, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, (0.0, 0))
        w = write.get(k, (0.0, 0))
        rd = f[0] * 1024 * 2.0
        wr = w[0] * 1024
        out["kernels"][k] = {"read_bytes": round(rd), "write_bytes": round(wr), "traffic_bytes": round(rd + wr),
                             "launches_sampled": min(f[1], w[1])}
    for k, v in out["kernels"].items():
        if k.startswith("k_materialize_raw") and v["write_bytes"]:  # present in per-operation (unfused) runs
            v["calibration_read_over_write"] = round(v["read_bytes"] / v["write_bytes"], 4)
            print(f"calibration {k}: corrected read / write = {v['calibration_read_over_write']}")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
