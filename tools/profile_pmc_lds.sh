#!/bin/bash
# LDS / MFMA counters of the bench step's kernels (separate --pmc passes, eager launches).
#   tools/profile_pmc_lds.sh <tag> [steps]   -> gpurun_out/pmclds_<tag>/<COUNTER>.csv (per-kernel averages)
set -e
TAG=${1:-x}; STEPS=${2:-2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmclds_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in ${COUNTERS:-SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS}; do
    timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o run -- python3 "$ROOT/bench.py" --no-graph --steps "$STEPS" --warmup 0 --no-roofline --no-cpu-baseline --no-stage-split --only-stage pretrain > "$OUT/$C.log" 2>&1 || { echo "$C failed"; tail -3 "$OUT/$C.log"; continue; }
    F=$(find "$OUT/$C" -name "*counter_collection.csv" 2>/dev/null | head -1)
    if [ -z "$F" ]; then echo "== $C: not available on this agent"; continue; fi
    python3 - "$F" "$C" > "$OUT/$C.txt" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") != sys.argv[2]:
        continue
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{k:60s} n={n:6d} avg={v / n:14.1f}")
PY
    rm -rf "$OUT/$C"
    echo "== $C"; head -6 "$OUT/$C.txt"
done
