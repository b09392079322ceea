#!/bin/bash
# In-step times of the encoder-layer products under ONE forced build of the ping-pong kernel at a time (CST_GEMM_PP_FORCE), next to the tile
# kernels (CST_GEMM_PP=0) and the measured per-shape choice: bench.py's roofline leg (HIP events bound to every GEMM dispatch of eager steps).
#   tools/gemm_pp_instep.sh <tag> [cfg ...]     -> gpurun_out/pp_instep_<tag>.txt
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pp_instep_$TAG.txt
: > "$OUT"
CFGS=${*:-"tiles auto 504 703 702 804 404 30602 30502 10403 30403 10402"}
for c in $CFGS; do
    case $c in
        tiles) export CST_GEMM_PP=0; unset CST_GEMM_PP_FORCE;;
        auto) unset CST_GEMM_PP; unset CST_GEMM_PP_FORCE;;
        *) unset CST_GEMM_PP; export CST_GEMM_PP_FORCE=$c;;
    esac
    timeout -k 10 240 python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-stage-split --no-other-workloads --no-f32 --no-cpu-baseline > /tmp/instep.json 2> /tmp/instep.err || { echo "$c FAILED" >> "$OUT"; tail -3 /tmp/instep.err >> "$OUT"; continue; }
    python3 - "$c" >> "$OUT" <<'PY'
import json, sys
d = json.loads(open("/tmp/instep.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(f"== {sys.argv[1]}: {d['ms_per_step']:.2f} ms/step; dominant {r['kernel']} {r['kernel_ms_per_step']:.2f} ms/step, frac {r['frac']:.3f}")
print("   " + "  ".join(f"{k.replace('.nt','')}:{v['avg_us']:.1f}" for k, v in r["by_shape"].items() if k.endswith(".nt")))
PY
done
cat "$OUT"
