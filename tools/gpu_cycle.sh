#!/bin/bash
# One tuning cycle on the GPU box: parity suite, the bench line without the CPU
# and link legs, the per-level reports.  Usage: tools/gpu_cycle.sh <tag> [level tables...]
set -e -o pipefail
tag=$1; shift || true
out=gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tee "$out/pytest.log" | tail -3
timeout -k 10 300 python bench.py --no-cpu-baseline --no-pcie > "$out/bench.json" 2> "$out/bench.err"
for t in "$@"; do
  timeout -k 10 200 python tools/level_report.py "$t" > "$out/level_$t.txt" 2>&1
done
echo done
