import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistent__style_transfer_amd import ops
def t_graph(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n
for M, N, K in [(9216, 2048, 768), (9216, 2304, 768), (9216, 768, 2048)]:
    A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    Ab, _ = ops.cast_bf16(A, want_t=False); Bb, _ = ops.cast_bf16(B, want_t=False)
    C = torch.empty(M, N, device="cuda"); Cb = torch.empty(M, (N + 63) // 64 * 64, device="cuda", dtype=torch.int16)
    bias = torch.randn(N, device="cuda"); add = torch.randn(M, N, device="cuda")
    d = ops.Drop(0.1, 3, 1001)
    r = {}
    r["fp32 C"] = t_graph(lambda: ops.gemm_bf16(Ab, Bb, M, N, C=C))
    r["Cb only"] = t_graph(lambda: ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb))
    r["Cb+bias+relu"] = t_graph(lambda: ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb, bias=bias, act=1))
    r["Cb+bias+relu+drop"] = t_graph(lambda: ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb, bias=bias, act=1, drop=d))
    r["C+bias"] = t_graph(lambda: ops.gemm_bf16(Ab, Bb, M, N, C=C, bias=bias))
    r["C+addend"] = t_graph(lambda: ops.gemm_bf16(Ab, Bb, M, N, C=C, addend=add))
    print(M, N, K, "  ".join(f"{k}: {v:.1f}" for k, v in r.items()))
