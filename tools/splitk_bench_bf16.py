#!/usr/bin/env python3
"""Split-K on/off for the small recurrent shapes on the bf16 direct-to-LDS GEMM (dependent chain in a hipGraph)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import ops
SH = [(256, 2048, 640, "dec gates"), (256, 1024, 256, "enc gates"), (256, 512, 1024, "fn_1 step"), (256, 10000, 512, "fn_2 step"),
      (256, 640, 2048, "dec dXH"), (256, 256, 1024, "enc dh"), (256, 512, 10048, "fn_2 dgrad step"), (256, 1024, 512, "fn_1 dgrad step")]
for M, N, K, note in SH:
    A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    Ab, Bb = ops.cast_bf16(A, want_t=False)[0], ops.cast_bf16(B, want_t=False)[0]
    C = torch.empty(M, N, device="cuda")
    junk = torch.empty(64 * 1024 * 1024, device="cuda")          # 256 MB stream between GEMMs: evicts L2 like the real step
    res = []
    TILE = int(os.environ.get("TILE", "0"))
    for sk in (0, 1, 2, 4, 8):
        row = []
        for evict in (False, True):
            for _ in range(2):
                ops.gemm_bf16(Ab, Bb, M, N, C=C, splitk=sk, tile=TILE)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            n = 10
            with torch.cuda.graph(g):
                for _ in range(n):
                    if evict:
                        junk[:8 * 1024 * 1024].zero_()
                    ops.gemm_bf16(Ab, Bb, M, N, C=C, splitk=sk, tile=TILE)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                for _ in range(n):
                    if evict:
                        junk[:8 * 1024 * 1024].zero_()
            def t(gr):
                gr.replay(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
                return a.elapsed_time(b) * 1000 / n
            row.append(t(g) - (t(g2) if evict else 0.0))
        res.append(row)
    print(f"{note:16s} {M}x{N}x{K}  " + "  ".join(f"sk{s}: {r[0]:5.1f}/{r[1]:5.1f}us" for s, r in zip((0, 1, 2, 4, 8), res)) + "   (hot / after a 32 MB memset)")
