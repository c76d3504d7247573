#!/bin/bash
# L2 <-> memory counter passes (one rocprofv3 run each) over one group of wide-table
# columns decoded alone: requests, stalls and in-flight levels (level / requests =
# mean latency in L2 cycles).  Usage: tools/pmc_tcc_variant.sh <nested|plain> <fraction> <outdir> [columns]
set -e
KIND=$1; FRAC=$2; OUT=$3; COLS=${4:-5000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
P1="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum"
P2="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_CYCLE_sum"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -s KILL 150 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/tools/wide_breakdown.py" "$COLS" "$KIND" "$FRAC" > "$OUT/p$i.log" 2>&1
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "nbl::" not in k: continue
        k = k.split("nbl::")[1].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, cs in agg.items():
        fh.write(k + "\n")
        for c, v in sorted(cs.items()):
            fh.write(f"   {c:36s} mean {sum(v)/len(v):16.1f}  n={len(v)}\n")
print(open(out + "/summary.txt").read())
PY
