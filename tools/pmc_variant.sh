#!/bin/bash
# SQ counter passes (one rocprofv3 run each) over one group of wide-table columns
# decoded alone.  Usage: tools/pmc_variant.sh <nested|plain> <fraction> <outdir> [columns]
set -e
KIND=$1; FRAC=$2; OUT=$3; COLS=${4:-5000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES_EQ_64"
P4="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES"
P5="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
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
            fh.write(f"   {c:32s} mean {sum(v)/len(v):16.1f}  n={len(v)}\n")
print(open(out + "/summary.txt").read())
PY
