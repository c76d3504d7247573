#!/bin/bash
# The four benchmark tables, decode only (no CPU baseline, link or per-operation legs): a minute of GPU time.
# usage: tools/quick_bench.sh <tag>   -> gpurun_out/qb_<tag>.json
set -e
tag=${1:-x}
timeout -k 10 200 python bench.py --no-cpu-baseline --no-pcie --no-breakdown > gpurun_out/qb_${tag}.json 2> gpurun_out/qb_${tag}.err
