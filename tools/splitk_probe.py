#!/usr/bin/env python3
"""Long-K bf16 NT products of the step (vocabulary-side dgrads, K = 10 048): time per (tile, split-K) choice.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import ops

SHAPES = [(4608, 512, 10048), (4608, 1024, 10048), (4608, 768, 2304), (9216, 768, 2048), (4608, 768, 2048)]
for M, N, K in SHAPES:
    A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    Ab, _ = ops.cast_bf16(A, want_t=False)
    Bb, _ = ops.cast_bf16(B, want_t=False)
    C = torch.empty(M, N, device="cuda")
    row = []
    for tile in (64, 136):
        for sk in (1, 2, 3, 4, 6):
            for _ in range(3):
                ops.gemm_bf16(Ab, Bb, M, N, C=C, tile=tile, splitk=sk)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    ops.gemm_bf16(Ab, Bb, M, N, C=C, tile=tile, splitk=sk)
            g.replay(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record(); torch.cuda.synchronize()
            row.append(f"t{tile}/k{sk}:{a.elapsed_time(b) * 50:6.1f}")
    print(f"{M}x{N}x{K}  " + "  ".join(row), flush=True)
