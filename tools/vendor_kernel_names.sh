#!/bin/bash
# Calibration only: names (tile configuration), grids, LDS and durations of the vendor GEMM kernels torch.matmul dispatches on the step's shapes.
#   tools/vendor_kernel_names.sh <tag>   -> gpurun_out/vendor_<tag>/names.txt
set -e
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/vendor_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -o run -- python3 "$ROOT/tools/blas_reference.py" > "$OUT/bench.log" 2>&1
CSV=$(find "$OUT/raw" -name "*kernel_trace.csv" | head -1)
python3 - "$CSV" > "$OUT/names.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if not n.startswith("Cijk"): continue
    k = (n, r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
for k, (c, t) in agg.items():
    print(f"{c:4d} x {t / c:7.1f} us  grid {k[1]} wg {k[2]} lds {k[3]} vgpr {k[4]} agpr {k[5]}\n      {k[0]}")
PY
rm -rf "$OUT/raw"
cat "$OUT/names.txt"
