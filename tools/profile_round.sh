#!/bin/bash
# Kernel-trace stats of the bench step (eager launches, so every kernel is visible to the tracer).
#   tools/profile_round.sh <tag> [steps] ["--only-stage optimize"]   -> gpurun_out/prof_<tag>/summary.md (+ kernel_stats.csv)
set -e
TAG=${1:-x}; STEPS=${2:-6}; EXTRA=${3:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the GEMM builds are measured per shape on their first eager call: do that in an untraced run, the traced run reads the choices
export CST_GEMM_PP_CACHE=/tmp/cst_gemm_pp_cache_$$.txt
python3 "$ROOT/bench.py" --no-graph --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-stage-split --no-other-workloads --no-f32 > /dev/null 2>&1 || true
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -o run -- python3 "$ROOT/bench.py" --no-graph --steps "$STEPS" --warmup 0 --no-roofline --no-cpu-baseline --no-stage-split $EXTRA > "$OUT/bench.log" 2>&1
CSV=$(find "$OUT/raw" -name "*kernel_stats.csv" | head -1)
cp "$CSV" "$OUT/kernel_stats.csv"
python3 "$ROOT/tools/profile_summary.py" stats "$OUT/kernel_stats.csv" "$STEPS" "$OUT/summary.md"
rm -rf "$OUT/raw"
tail -1 "$OUT/bench.log" | cut -c1-200
