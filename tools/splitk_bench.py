#!/usr/bin/env python3
"""Split-K on/off for the small recurrent GEMM shapes, timed as a dependent chain in a hipGraph."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import ops
SH = [(256, 256, 1024, 1, 0, "enc dh"), (256, 640, 2048, 1, 0, "dec dXH"), (256, 512, 1024, 1, 1, "fn_1 step"),
      (256, 512, 10000, 1, 0, "fn_2 dgrad step"), (256, 1024, 256, 1, 1, "enc gates"), (256, 2048, 640, 1, 1, "dec gates"),
      (300, 24, 65536, 0, 0, "disc conv wgrad"), (100, 1200, 4096, 0, 0, "f2o wgrad"), (1200, 1200, 4096, 0, 0, "highway wgrad"),
      (2048, 512, 9216, 0, 0, "FFN1 wgrad"), (1536, 512, 9216, 0, 0, "QKV wgrad"), (512, 512, 9216, 0, 0, "out wgrad")]
for M, N, K, akm, bkm, note in SH:
    A = torch.randn((M, K) if akm else (K, M), device="cuda")
    B = torch.randn((N, K) if bkm else (K, N), device="cuda")
    C = torch.empty(M, N, device="cuda")
    res = []
    for sk in (0, 1, 2, 4, 8):
        for _ in range(2):
            ops.gemm(A, akm, B, bkm, C, M, N, K, splitk=sk)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        n = 20
        with torch.cuda.graph(g):
            for _ in range(n):
                ops.gemm(A, akm, B, bkm, C, M, N, K, splitk=sk)
                ops.gemm(A, akm, B, bkm, C, M, N, K, splitk=sk)   # same C: serialised by the WAW dependency of the stream
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) * 1000 / (2 * n))
    print(f"{note:18s} {M:5d}x{N:5d}x{K:6d} a{akm}b{bkm}  " + "  ".join(f"sk{s}:{u:7.1f}us" for s, u in zip((0, 1, 2, 4, 8), res)))
