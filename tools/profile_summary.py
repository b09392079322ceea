#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into the small
summaries committed under profiles/.

  python tools/profile_summary.py stats  <kernel_stats.csv> <steps> <out.md>
  python tools/profile_summary.py pmc    <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def stats(path, steps, out):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    fam = collections.defaultdict(lambda: [0, 0.0])
    BOTH = "`cst_gemm_bf16_kernel<*>` + `cst_gemm_bf16_tt_group_kernel` + `cst_gemm_bf16_pp_kernel<*>` (the bf16 GEMM of cst_gemm_bf16 / _tt: its kernels together)"
    for r in rows:
        n = short(r["Name"])
        f = "cst_gemm_kernel<*>" if n.startswith("cst_gemm_kernel") else ("cst_gemm_bf16_kernel<*>" if n.startswith("cst_gemm_bf16_kernel") else n)
        if "cst_gemm_bf16_pp_kernel" in n:
            f = "cst_gemm_bf16_pp_kernel<*>"
        fam[f][0] += int(r["Calls"])
        fam[f][1] += float(r["TotalDurationNs"])
        if f in ("cst_gemm_bf16_kernel<*>", "cst_gemm_bf16_pp_kernel<*>", "cst_gemm_bf16_tt_group_kernel"):
            fam[BOTH][0] += int(r["Calls"])
            fam[BOTH][1] += float(r["TotalDurationNs"])
    with open(out, "w") as f:
        f.write(f"rocprofv3 --kernel-trace --stats, {steps} steps: total kernel time {tot / 1e6:.2f} ms = {tot / 1e6 / steps:.2f} ms/step\n")
        f.write("(the first row is the SUM of the rows of the bf16 GEMM's kernels below it -- the big-tile ping-pong kernel of csrc/gemm_pp.hip takes the shapes it wins,\n"
                " the LDS-DMA tile kernels of csrc/gemm_bf16.hip the rest, the grouped launch of the same tile body the weight gradients of a layer / of the generator;\n"
                " bench.py's roofline.kernel is this family, its member cst_gemm_bf16_kernel = the tile kernels + the grouped launches)\n\n")
        f.write("| kernel (family) | calls/step | total ms/step | avg us | % |\n|---|---|---|---|---|\n")
        for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:29]:
            name = k if k == BOTH else f"`{k}`"
            f.write(f"| {name} | {c / steps:.1f} | {t / 1e6 / steps:.3f} | {t / c / 1e3:.1f} | {100 * t / tot:.1f} |\n")
        f.write("\nPer template instantiation of the GEMM:\n\n| kernel | calls/step | avg us | % |\n|---|---|---|---|\n")
        for r in rows:
            if "cst_gemm" in r["Name"]:
                f.write(f"| `{short(r['Name'])}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")


def pmc(fetch, write, out, workload="yelp_6l_d768_b256"):
    def agg(path, name):
        d = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != name:
                continue
            k = short(r["Kernel_Name"])
            k = "cst_gemm_kernel" if k.startswith("cst_gemm_kernel") else k
            k = "cst_gemm_bf16_kernel" if (k.startswith("cst_gemm_bf16_kernel") or k.startswith("cst_gemm_bf16_tt_group_kernel")) else k
            k = "cst_gemm_bf16_pp_kernel" if "cst_gemm_bf16_pp_kernel" in k else k
            d[k][0] += 1
            d[k][1] += float(r["Counter_Value"])
            d[k][2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        return d
    f, w = agg(fetch, "FETCH_SIZE"), agg(write, "WRITE_SIZE")
    res = {"workload": workload,
           "note": "FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced "
                   "reads (MI355X_MICROARCH.md, HBM) so it is doubled here; separate --pmc passes", "kernels": {}}
    for k in sorted(f, key=lambda k: -f[k][1])[:12]:
        n = f[k][0]
        res["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": 2 * f[k][1] * 1024 / n,
                             "write_bytes_per_launch": w[k][1] * 1024 / max(1, w[k][0]),
                             "avg_us_under_pmc": f[k][2] / n / 1e3}
    json.dump(res, open(out, "w"), indent=1)


def gemmpmc(out_dir):
    """Per (kernel, grid) averages of every counter in sq.csv / mem1.csv / mem2.csv, plus dispatch duration."""
    import os
    table = collections.defaultdict(lambda: collections.defaultdict(list))
    for name in ("sq", "mem1", "mem2"):
        path = os.path.join(out_dir, name + ".csv")
        if not os.path.exists(path):
            continue
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if "gemm_bf16" not in k or "reduce" in k:
                continue
            key = (k.replace("cst_gemm_bf16_", ""), r.get("Grid_Size", "?"))
            table[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if name == "sq" and r["Counter_Name"] == "SQ_WAVE_CYCLES":
                table[key]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for key in sorted(table):
        t = table[key]
        avg = {c: sum(v[2:]) / max(1, len(v[2:])) for c, v in t.items()}      # skip the first two (cold) launches
        print(f"{key[0]:40s} grid {key[1]:>8s}  " + "  ".join(f"{c}={avg[c]:.4g}" for c in sorted(avg)))


if __name__ == "__main__":
    if sys.argv[1] == "gemmpmc":
        gemmpmc(sys.argv[2])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], *(sys.argv[5:6]))
