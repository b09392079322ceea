#!/usr/bin/env python3
"""cst_colsum at the bias-gradient shapes of the bench step, hot loop inside a hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import ops
for M, N in [(9216, 1536), (4608, 1536), (4608, 1024), (4608, 2048), (4608, 512), (1024, 1536), (4608, 10000)]:
    x = torch.randn(M, N, device="cuda")
    for _ in range(3): ops.colsum(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): ops.colsum(x)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    err = (ops.colsum(x) - x.sum(0)).abs().max().item()
    print(M, N, "us:", round(a.elapsed_time(b) * 1000 / 20, 2), "GB/s:", round(M * N * 4 / (a.elapsed_time(b) / 20 * 1e-3) / 1e9), "err", err)
