#!/usr/bin/env python3
"""Does a grouped weight-gradient launch (336 workgroups on 512 slots) overlap with the next layer's backward products when it is issued
on a second stream?  Times, inside one hipGraph: the group + four NT products back to back on one stream, against the group forked onto
a side stream and joined after the four products.    python tools/overlap_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import ops
ops.set_precision("bf16")


def bf(r, c):
    return torch.randn(r, c, device="cuda").to(torch.bfloat16).view(torch.int16)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n


side = torch.cuda.Stream()
for T in (9216, 4608):
    d, F = 768, 2048
    dfb, wh, dhb, wy1, dob, watt, dqb, wx = bf(T, d), bf(T, F), bf(T, F), bf(T, d), bf(T, d), bf(T, d), bf(T, 3 * d), bf(T, d)
    outs = [torch.empty(d, F, device="cuda"), torch.empty(F, d, device="cuda"), torch.empty(d, d, device="cuda"), torch.empty(3 * d, d, device="cuda")]
    w2t, w1t, wot, wqt = bf(F, d), bf(d, F), bf(d, d), bf(d, 3 * d)          # dgrad B operands [N, K]
    c_f, c_d, c_d2, c_d3 = torch.empty(T, F, device="cuda", dtype=torch.int16), torch.empty(T, d, device="cuda"), torch.empty(T, d, device="cuda", dtype=torch.int16), torch.empty(T, d, device="cuda")

    def group():
        with ops.tt_group():
            ops.gemm_bf16_tt(dfb, wh, d, F, C=outs[0])
            ops.gemm_bf16_tt(dhb, wy1, F, d, C=outs[1])
            ops.gemm_bf16_tt(dob, watt, d, d, C=outs[2])
            ops.gemm_bf16_tt(dqb, wx, 3 * d, d, C=outs[3])

    def nts():                                  # the next layer's four dgrad products
        ops.gemm_bf16(dfb, w2t, T, F, Cb=c_f)
        ops.gemm_bf16(dhb, w1t, T, d, C=c_d)
        ops.gemm_bf16(dob, wot, T, d, Cb=c_d2)
        ops.gemm_bf16(dqb, wqt, T, d, C=c_d3)

    def serial():
        group()
        nts()

    def forked():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            group()
        nts()
        main.wait_stream(side)

    tg, tn, ts, tf = timeit(group), timeit(nts), timeit(serial), timeit(forked)
    print(f"T={T}: group alone {tg:.1f} us, four NT products alone {tn:.1f} us, back to back {ts:.1f} us, group on a side stream {tf:.1f} us", flush=True)
