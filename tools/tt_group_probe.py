#!/usr/bin/env python3
"""The weight gradients of one encoder layer (dW of FFN2, FFN1, out-projection, QKV: C = A^T B over the token count), launched one by one
(split-K slabs + a reduce each) against ONE grouped launch of whole-K workgroups (cst_gemm_bf16_tt_group_*, ops.tt_group).
    python tools/tt_group_probe.py            (CST_TT_GROUP_MIN=1 to force grouping of the small-d layers too, CST_TT_GROUP_SPLITS=S to force S workgroups per tile)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import ops
ops.set_precision("bf16")


def bf(r, c):
    return torch.randn(r, c, device="cuda").to(torch.bfloat16).view(torch.int16)


def timeit(fn, n=20):
    for _ in range(3):
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


for T, d, F, note in ((9216, 768, 2048, "Matcher d768"), (4608, 768, 2048, "MLM d768"), (9216, 512, 2048, "Matcher d512"), (4608, 512, 2048, "MLM d512"),
                      (30720, 512, 2048, "Matcher book"), (15360, 512, 2048, "MLM book"), (73728, 768, 2048, "Matcher B=2048")):
    dfb, wh, dhb, wy1, dob, watt, dqb, wx = bf(T, d), bf(T, F), bf(T, F), bf(T, d), bf(T, d), bf(T, d), bf(T, 3 * d), bf(T, d)
    outs = [torch.empty(d, F, device="cuda"), torch.empty(F, d, device="cuda"), torch.empty(d, d, device="cuda"), torch.empty(3 * d, d, device="cuda")]

    def four():
        ops.gemm_bf16_tt(dfb, wh, d, F, C=outs[0])
        ops.gemm_bf16_tt(dhb, wy1, F, d, C=outs[1])
        ops.gemm_bf16_tt(dob, watt, d, d, C=outs[2])
        ops.gemm_bf16_tt(dqb, wx, 3 * d, d, C=outs[3])

    def grouped():
        with ops.tt_group():
            four()

    def two_layers():                      # what a backward pass under ops.tt_deferred does: eight problems per launch
        with ops.tt_deferred():
            for _ in range(2):
                with ops.tt_group(deferrable=True):
                    four()

    from consistent__style_transfer_amd._lib import call_plain
    t4, tg = timeit(four), timeit(grouped)
    S = call_plain("cst_gemm_bf16_tt_group_last_splits")          # of the one-layer group (0: launched product by product)
    t2 = timeit(two_layers) / 2
    fl = 2.0 * T * (2 * d * F + 4 * d * d)
    tiles = 2 * (-(-d // 128)) * (F // 128) + (-(-d // 128)) ** 2 + (-(-3 * d // 128)) * (-(-d // 128))
    print(f"{note:16s} T={T:6d} d={d} ({tiles} tiles): one by one {t4:7.1f} us ({fl / t4 / 1e6:4.0f} TF/s)   grouped {tg:7.1f} us ({fl / tg / 1e6:4.0f} TF/s, {S} split{'s' if S != 1 else ''})   two layers per launch {t2:7.1f} us a layer ({fl / t2 / 1e6:4.0f} TF/s)", flush=True)
