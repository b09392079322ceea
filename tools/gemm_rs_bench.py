#!/usr/bin/env python3
"""The register-staged 256 x 128 experiment (csrc/gemm_rs.hip) against the dispatched kernel and the vendor library on the encoder-layer
shapes it can take; checks the product first.  GPU box only; needs the bench build of the library:
    CST_BENCH_VARIANTS=1 python -m consistent__style_transfer_amd.build --force && python tools/gemm_rs_bench.py
(the experiment's entry point is not part of include/cst_hip.h: it is bound here by hand)."""
import ctypes
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import ops
from consistent__style_transfer_amd import build

_so = ctypes.CDLL(build.LIB)
if not hasattr(_so, "cst_gemm_bf16_rs"):
    sys.exit("libcst_hip.so was built without CST_BENCH_VARIANTS=1")
_so.cst_gemm_bf16_rs.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


def call(name, A, lda, B, ldb, C, ldc, M, N, K, pf):
    rc = _so.cst_gemm_bf16_rs(A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), ldc, M, N, K, pf, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


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
    a.record(); g.replay(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / (2 * n)


for M, N, K in ((9216, 768, 2048), (9216, 768, 2304), (9216, 768, 768), (9216, 2304, 768), (9216, 2048, 768), (4608, 768, 2048), (4608, 2304, 768), (4096, 4096, 4096)):
    A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    Ab, _ = ops.cast_bf16(A, want_t=False)
    Bb, _ = ops.cast_bf16(B, want_t=False)
    Cb = torch.zeros(M, N, device="cuda", dtype=torch.int16)
    Cr = torch.zeros(M, N, device="cuda", dtype=torch.int16)
    ops.gemm_bf16(Ab, Bb, M, N, Cb=Cr)
    res = {}
    for pf in (2, 3):
        Cb.zero_()
        call("cst_gemm_bf16_rs", Ab, Ab.stride(0), Bb, Bb.stride(0), Cb, N, M, N, K, pf)
        torch.cuda.synchronize()
        d = (Cb.view(torch.bfloat16).float() - Cr.view(torch.bfloat16).float()).abs().max().item()
        ref = Cr.view(torch.bfloat16).float().abs().max().item()
        assert d <= 2e-2 * ref, (M, N, K, pf, d, ref)
        res[pf] = timed(lambda: call("cst_gemm_bf16_rs", Ab, Ab.stride(0), Bb, Bb.stride(0), Cb, N, M, N, K, pf))
    t_ours = timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, Cb=Cr))
    A16, B16 = A.bfloat16(), B.bfloat16()
    out16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_v = timed(lambda: torch.matmul(A16, B16.t(), out=out16))
    fl = 2.0 * M * N * K / 1e6
    print(f"{M:5d}x{N:5d}x{K:5d}: rs pf2 {res[2]:6.1f} us ({fl / res[2]:6.0f} TF/s)  pf3 {res[3]:6.1f} ({fl / res[3]:6.0f})  | dispatched {t_ours:6.1f} ({fl / t_ours:6.0f}) | vendor {t_v:6.1f} ({fl / t_v:6.0f})", flush=True)
