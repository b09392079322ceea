#!/bin/bash
# Hardware counters of the ping-pong GEMM (csrc/gemm_pp.hip) next to the tile kernels on the encoder-layer shapes (tools/gemm_pp_pmc.py).
#   tools/profile_gemm_pp_pmc.sh <tag>      -> gpurun_out/pppmc_<tag>/summary.txt
# Separate --pmc passes (SQ: 8 slots; TCC: FETCH_SIZE 3 + 1, WRITE_SIZE 2 + 2), kernel-trace only.
set -e
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pppmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
    local name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/raw_$name" -o run -- python3 "$ROOT/tools/gemm_pp_pmc.py" > "$OUT/$name.log" 2>&1
    cp "$(find "$OUT/raw_$name" -name "*counter_collection.csv" | head -1)" "$OUT/$name.csv"
    rm -rf "$OUT/raw_$name"
    echo "$name pass done"
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run mem1 FETCH_SIZE GRBM_GUI_ACTIVE
run mem2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 "$ROOT/tools/gemm_pp_pmc.py" summary "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
