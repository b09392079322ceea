#!/usr/bin/env python3
"""Condense a rocprofv3 kernel_trace.csv of `bench.py --no-graph --only-stage X` into
  seq.txt       the launches of the LAST step in order: short kernel name, grid, workgroup, LDS, duration
  by_shape.txt  (kernel, grid, workgroup) -> launches per step, average duration, total per step

    python tools/trace_seq.py <kernel_trace.csv> <steps> <out_dir>
"""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("at::native::", "")
    return name[:70]


def main():
    path, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(rows)
    per = n // steps
    gx = lambda r: int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    gy = lambda r: int(r.get("Grid_Size_Y", 1) or 1)
    gz = lambda r: int(r.get("Grid_Size_Z", 1) or 1)
    wx = lambda r: int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    last = rows[n - per:]
    with open(out + "/seq.txt", "w") as f:
        t0 = int(last[0]["Start_Timestamp"])
        f.write(f"# {n} launches in {steps} steps; the last {per} (one step) in start order: start_us dur_us gap_us grid wg lds name\n")
        prev_end = t0
        for r in last:
            s = int(r["Start_Timestamp"])
            f.write(f"{(s - t0) / 1e3:10.1f} {dur(r):8.1f} {(s - prev_end) / 1e3:7.1f}  {gx(r) // max(1, wx(r)):6d}x{gy(r)}x{gz(r)} {wx(r):5d} {int(r.get('LDS_Block_Size', 0) or 0):7d}  {short(r['Kernel_Name'])}\n")
            prev_end = int(r["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[per:]:                      # skip the first (cold) step
        k = (short(r["Kernel_Name"]), gx(r) // max(1, wx(r)), gy(r), gz(r), wx(r))
        agg[k][0] += 1
        agg[k][1] += dur(r)
    ns = max(1, steps - 1)
    tot = sum(v[1] for v in agg.values())
    with open(out + "/by_shape.txt", "w") as f:
        f.write(f"# steps 2..{steps}: total kernel time {tot / ns / 1e3:.3f} ms/step, {sum(v[0] for v in agg.values()) / ns:.0f} launches/step\n")
        f.write("# ms/step  calls/step  avg_us   grid(wg)xYxZ  threads  kernel\n")
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{t / ns / 1e3:8.3f} {c / ns:8.1f} {t / c:8.1f}   {k[1]:6d}x{k[2]}x{k[3]} {k[4]:5d}  {k[0]}\n")


if __name__ == "__main__":
    main()
