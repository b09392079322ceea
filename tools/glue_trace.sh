#!/bin/bash
# Per-launch durations of the torch / runtime glue kernels (fills, copies, elementwise adds) in the bench step.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/glue
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -o run -- python3 "$ROOT/bench.py" --no-graph --steps 4 --warmup 0 --no-roofline --no-cpu-baseline > "$OUT/bench.log" 2>&1
CSV=$(find "$OUT/raw" -name "*kernel_trace.csv" | head -1)
python3 - "$CSV" > "$OUT/glue.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in rows]
n = len(rows)
# last quarter = the last step (it=3, no D update) ; take the step before (it=... ) whichever: use last 1/4 of launches
lo = n * 3 // 4
agg = collections.defaultdict(list)
for i in range(lo, n):
    nm = names[i]
    if any(k in nm for k in ("elementwise", "fillBuffer", "copyBuffer", "reduce_kernel", "CatArray", "index")):
        prev = names[i - 1].split("(")[0][:40]
        nxt = names[i + 1].split("(")[0][:40] if i + 1 < n else ""
        key = ("add" if "CUDAFunctor_add" in nm else "fill" if "Fill" in nm or "fillBuffer" in nm else "copy" if "copy" in nm.lower() else nm[:30])
        agg[(key, prev, nxt)].append(dur[i])
tot = 0
for (key, prev, nxt), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print(f"{sum(v):8.1f} us  n={len(v):3d} max={max(v):6.1f}  {key:6s} after[{prev}] before[{nxt}]")
print("total", tot)
PY
rm -rf "$OUT/raw"
head -45 "$OUT/glue.txt"
