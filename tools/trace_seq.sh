#!/bin/bash
# Per-launch kernel sequence of one stage's step (eager launches under rocprofv3 --kernel-trace): which kernels run, in what
# order, how long each takes -- the view the per-family --stats table cannot give for the decode loop's dependent chain.
#   tools/trace_seq.sh <tag> <stage: pretrain|warmup|optimize> [steps]   -> gpurun_out/seq_<tag>/{seq.txt,by_shape.txt}
set -e
TAG=${1:-x}; STAGE=${2:-optimize}; STEPS=${3:-3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/seq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -o run -- python3 "$ROOT/bench.py" --no-graph --steps "$STEPS" --warmup 0 --only-stage "$STAGE" > "$OUT/bench.log" 2>&1
CSV=$(find "$OUT/raw" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/trace_seq.py" "$CSV" "$STEPS" "$OUT"
rm -rf "$OUT/raw"
head -60 "$OUT/by_shape.txt"
