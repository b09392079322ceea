#!/usr/bin/env python3
"""Calibration only (never on the product path): what the vendor GEMM (torch.matmul on bf16 -> hipBLASLt / rocBLAS) reaches on the
encoder-layer shapes of the step, next to this library's kernels, same random data, hipGraph of 20 launches each.  A known-good reference
on the same hardware bounds what these shapes allow (cdna_hip_programming.md 5.4 rule 10)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import ops

SHAPES = [(9216, 2048, 768), (9216, 768, 2048), (9216, 2304, 768), (9216, 768, 2304), (9216, 768, 768), (4608, 2048, 768), (4608, 768, 2048),
          (4608, 2304, 768), (4608, 768, 768), (4608, 512, 10048), (4608, 10000, 768), (256, 10000, 512), (256, 2048, 640), (4096, 4096, 4096)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n


for M, N, K in SHAPES:
    A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    Ab16, Bb16 = A.bfloat16(), B.bfloat16()
    out16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_blas = timed(lambda: torch.matmul(Ab16, Bb16.t(), out=out16))
    Ab, _ = ops.cast_bf16(A, want_t=False)
    Bb, _ = ops.cast_bf16(B, want_t=False)
    Cb = torch.empty(M, (N + 63) // 64 * 64, device="cuda", dtype=torch.int16)
    C = torch.empty(M, N, device="cuda")
    t_ours_b = timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb))
    t_ours_f = timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, C=C))
    fl = 2.0 * M * N * K
    print(f"{M:6d}x{N:5d}x{K:5d}  vendor bf16-out {t_blas:7.1f} us {fl / t_blas / 1e6:7.1f} TF/s | ours bf16-out {t_ours_b:7.1f} us {fl / t_ours_b / 1e6:7.1f} TF/s"
          f" | ours fp32-out {t_ours_f:7.1f} us {fl / t_ours_f / 1e6:7.1f} TF/s", flush=True)
