#!/usr/bin/env python3
"""A few launches of the register-staged experiment and of the dispatched kernel on one shape, for rocprofv3 --pmc (bench build)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import build, ops
_so = ctypes.CDLL(build.LIB)
_so.cst_gemm_bf16_rs.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 4096, 4096)))
A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
Ab, _ = ops.cast_bf16(A, want_t=False)
Bb, _ = ops.cast_bf16(B, want_t=False)
Cb = torch.zeros(M, N, device="cuda", dtype=torch.int16)
for _ in range(5):
    _so.cst_gemm_bf16_rs(Ab.data_ptr(), Ab.stride(0), Bb.data_ptr(), Bb.stride(0), Cb.data_ptr(), N, M, N, K, 3, torch.cuda.current_stream().cuda_stream)
    ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb)
torch.cuda.synchronize()
