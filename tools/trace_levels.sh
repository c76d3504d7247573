#!/bin/bash
# Exact duration of every launch of a step and the gaps between them, from a
# rocprofv3 kernel trace of a short bench.py run (tuning aid, GPU box).
# Usage: tools/trace_levels.sh <outdir> <workload>...
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(cd "$ROOT" && mkdir -p "$1" && cd "$1" && pwd); shift
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  rm -rf "/tmp/trace_$W"
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "/tmp/trace_$W" -- python3 "$ROOT/bench.py" --workload $W \
      --no-cpu-baseline --no-pcie --no-breakdown --no-other-workloads --steps 6 --warmup 2 > "$OUT/trace_$W.log" 2>&1
  python3 "$ROOT/tools/trace_levels.py" "$(ls /tmp/trace_$W/*/*kernel_trace.csv | head -1)" > "$OUT/trace_$W.txt"
done
