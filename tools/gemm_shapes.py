#!/usr/bin/env python3
"""Census of the GEMM shapes one bench step launches, with each unique shape timed in isolation.

Records (kernel, M, N, K, flags) of every cst_gemm / cst_gemm_bf16 call of one eager pretrain + warmup +
optimize step, then replays each unique call 20x back to back inside a hipGraph (hot L2) and prints
count x time, sorted by share.  Use it to decide which shapes deserve a tile / split-K rule."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from consistent__style_transfer_amd import _lib, ops  # noqa: E402

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "yelp_4l_d512_b256"]
bench.ONLY = sys.argv[2] if len(sys.argv) > 2 else None          # optional: one stage only
dev = torch.device("cuda:0")
stages_ = bench.build_stages(w, dev)
batches = bench.make_batches(w, 0, dev)
for it in range(2):
    bench.run_step(stages_, batches, it, None)
torch.cuda.synchronize()

calls = collections.OrderedDict()
orig_call = _lib.call


def spy(name, *args):
    if name in ("cst_gemm", "cst_gemm_bf16"):
        if name == "cst_gemm":
            M, N, K = args[8], args[9], args[10]
            key = ("f32stage", M, N, K, f"a{args[2]}b{args[5]}", "acc" if args[18] else "")
        else:
            M, N, K = args[8], args[9], args[10]
            key = ("bf16", M, N, K, "Cb" if args[6] is not None else "C", "acc" if args[19] else "")
        ent = calls.setdefault(key, [0, None])
        ent[0] += 1
        if ent[1] is None:
            ent[1] = (name, args)
    return orig_call(name, *args)


_lib.call = spy
ops.call = spy
for m in list(sys.modules.values()):
    if m is not None and getattr(m, "__name__", "").startswith("consistent__style_transfer_amd") and getattr(m, "call", None) is orig_call:
        m.call = spy
bench.run_step(stages_, batches, 4, None)          # it % 4 == 0: includes the discriminator update
torch.cuda.synchronize()
for m in list(sys.modules.values()):
    if m is not None and getattr(m, "call", None) is spy:
        m.call = orig_call

rows = []
for key, (cnt, (name, args)) in calls.items():
    for _ in range(3):
        orig_call(name, *args)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    n = 20
    with torch.cuda.graph(g):
        for _ in range(n):
            orig_call(name, *args)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1000 / n
    rows.append((cnt * us, cnt, us, key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{len(rows)} unique GEMM calls, {sum(r[1] for r in rows)} launches/step, {tot / 1000:.2f} ms/step if each ran hot")
print(f"{'kernel':9s} {'M':>6s} {'N':>6s} {'K':>6s} {'flags':10s} {'n':>4s} {'us':>7s} {'ms/step':>8s} {'TF/s':>7s} {'cum%':>5s}")
cum = 0.0
for t, cnt, us, key in rows:
    cum += t
    k, M, N, K = key[:4]
    print(f"{k:9s} {M:6d} {N:6d} {K:6d} {' '.join(key[4:]):10s} {cnt:4d} {us:7.1f} {t / 1000:8.3f} {2.0 * M * N * K / us / 1e6:7.1f} {100 * cum / tot:5.1f}")
