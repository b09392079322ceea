#!/usr/bin/env python3
"""LayerNorm forward / backward kernels at the encoder-layer shapes, hot loop inside a hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import ops
from consistent__style_transfer_amd.ops import _ln_bwd, _ln_fwd


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n


for T, d in [(9216, 512), (4608, 512)]:
    x, res = torch.randn(T, d, device="cuda"), torch.randn(T, d, device="cuda")
    g, b = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
    z, y = torch.empty(T, d, device="cuda"), torch.empty(T, d, device="cuda")
    yb = torch.empty(T, d, device="cuda", dtype=torch.int16)
    mean, rstd = torch.empty(T, device="cuda"), torch.empty(T, device="cuda")
    drop = ops.Drop(0.1, 5, 1001)
    print(T, d, "fwd+twin+drop us:", timeit(lambda: _ln_fwd(x, res, g, b, drop, z, y, mean, rstd, yb=yb)))
    print(T, d, "fwd plain us:", timeit(lambda: _ln_fwd(x, res, g, b, ops.NO_DROP, z, y, mean, rstd)))
    dy = torch.randn(T, d, device="cuda")
    print(T, d, "bwd fused(twin+drop+3 partials) us:", timeit(lambda: _ln_bwd(dy, z, mean, rstd, g, True, dzb_drop=drop)))
    print(T, d, "bwd plain us:", timeit(lambda: _ln_bwd(dy, z, mean, rstd, g, True)))
