#!/bin/bash
# HBM traffic of the bench step's kernels: two separate counter passes (FETCH_SIZE, WRITE_SIZE), eager launches.
#   tools/profile_pmc.sh <tag> [steps]      -> gpurun_out/pmc_<tag>/pmc_traffic.json
set -e
TAG=${1:-x}; STEPS=${2:-3}; WL=${3:-yelp_6l_d768_b256}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the GEMM builds are measured per shape on their first eager call: do that in an untraced run, the traced run reads the choices
export CST_GEMM_PP_CACHE=/tmp/cst_gemm_pp_cache_$$.txt
python3 "$ROOT/bench.py" --no-graph --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-stage-split --no-other-workloads --no-f32 ${WL:+--workload $WL} > /dev/null 2>&1 || true
for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o run -- python3 "$ROOT/bench.py" --no-graph --steps "$STEPS" --warmup 0 --no-roofline --no-cpu-baseline --workload $WL > "$OUT/$C.log" 2>&1
    echo "$C pass done"
done
F=$(find "$OUT/FETCH_SIZE" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/WRITE_SIZE" -name "*counter_collection.csv" | head -1)
python3 "$ROOT/tools/profile_summary.py" pmc "$F" "$W" "$OUT/pmc_traffic.json" "$WL"
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
head -12 "$OUT/pmc_traffic.json"
