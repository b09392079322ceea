#!/bin/bash
# SQ counters of the register-staged experiment next to the dispatched kernel (bench build of the library needed).
#   tools/gemm_rs_pmc.sh <tag> [M N K]   -> gpurun_out/rspmc_<tag>/summary.txt
set -e
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/rspmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
    local name=$1; shift
    timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/raw_$name" -o run -- python3 "$ROOT/tools/gemm_rs_pmc.py" $ARGS > "$OUT/$name.log" 2>&1
    cp "$(find "$OUT/raw_$name" -name "*counter_collection.csv" | head -1)" "$OUT/$name.csv"
    rm -rf "$OUT/raw_$name"
}
ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
python3 - "$OUT" > "$OUT/summary.txt" <<'PY'
import csv, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in ("sq1", "sq2"):
    for r in csv.DictReader(open(f"{out}/{f}.csv")):
        n = r["Kernel_Name"]
        k = "rs" if "gemm_rs" in n else ("dispatched" if "cst_gemm_bf16_kernel" in n else None)
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, "  ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(d.items())))
PY
cat "$OUT/summary.txt"
