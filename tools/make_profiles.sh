#!/bin/bash
# Regenerates profiles/<tag>_* on the GPU box: for every benchmark table the
# rocprofv3 kernel statistics of a bench.py run, and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE: one rocprofv3 run each) turned into HBM bytes per
# launch by tools/pmc_traffic.py.  Usage: tools/make_profiles.sh <tag> [workloads...]
set -e
TAG=$1; shift
WL=${@:-fbw_rle dict_string mixed wide}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$ROOT/profiles"
cd /tmp && export TMPDIR=/tmp
declare -A ROWS=([fbw_rle]=10000000 [dict_string]=10000000 [mixed]=1250000 [wide]=100000)
for W in $WL; do
  ARGS="--workload $W --no-cpu-baseline --no-pcie --no-breakdown --no-other-workloads --no-cold"
  rm -rf "$OUT/$W"
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$W/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$W.stats.json" 2> "$OUT/$W.stats.err"
  cp "$(ls "$OUT/$W"/stats/*/*kernel_stats.csv | head -1)" "$ROOT/profiles/${TAG}_${W}_kernel_stats.csv"
  tail -1 "$OUT/$W.stats.json" > "$ROOT/profiles/${TAG}_${W}_bench_under_rocprof.json"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$W/$C" -- python3 "$ROOT/bench.py" $ARGS --steps 3 --warmup 1 > "$OUT/$W.$C.log" 2>&1
    cp "$(ls "$OUT/$W/$C"/*/*counter_collection.csv | head -1)" "$ROOT/profiles/${TAG}_${W}_pmc_$(echo $C | tr A-Z a-z).csv"
  done
  python3 "$ROOT/tools/pmc_traffic.py" "$ROOT/profiles/${TAG}_${W}_pmc_fetch_size.csv" "$ROOT/profiles/${TAG}_${W}_pmc_write_size.csv" \
      "$ROOT/profiles/${TAG}_${W}_pmc_traffic.json" ${ROWS[$W]} $W > "$OUT/$W.traffic.log"
  echo "$W done"
done
# profiles/ is tracked; gpurun only merges gpurun_out/ back
mkdir -p "$ROOT/gpurun_out/profiles_out" && cp "$ROOT"/profiles/${TAG}_* "$ROOT/gpurun_out/profiles_out/"
