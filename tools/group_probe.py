#!/usr/bin/env python3
"""Would ONE grouped launch of the Matcher's (M = 9216) and the MLM's (M = 4608) encoder-layer products beat two launches?  A grouped launch
of two problems with equal N, K costs what a single (M1 + M2) x N x K product costs (same tiles, other pointers), so this times exactly that
against the two separate launches, 20 dependent launches per hipGraph, for every NT product of a layer.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import ops

M1, M2 = 9216, 4608


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


tot_sep = tot_grp = 0.0
for N, K, out in ((2304, 768, "Cb"), (768, 768, "C"), (2048, 768, "Cb"), (768, 2048, "C"), (768, 2304, "C")):
    A = torch.randn(M1 + M2, K, device="cuda")
    Ab, _ = ops.cast_bf16(A, want_t=False)
    Bb, _ = ops.cast_bf16(torch.randn(N, K, device="cuda"), want_t=False)
    Bb2, _ = ops.cast_bf16(torch.randn(N, K, device="cuda"), want_t=False)
    C = torch.empty(M1 + M2, N, device="cuda")
    Cb = torch.empty(M1 + M2, (N + 63) // 64 * 64, device="cuda", dtype=torch.int16)
    kw = lambda lo, hi: dict(Cb=Cb[lo:hi]) if out == "Cb" else dict(C=C[lo:hi])

    def sep():
        ops.gemm_bf16(Ab[:M1], Bb, M1, N, **kw(0, M1))
        ops.gemm_bf16(Ab[M1:], Bb2, M2, N, **kw(M1, M1 + M2))

    def grp():
        ops.gemm_bf16(Ab, Bb, M1 + M2, N, **kw(0, M1 + M2))

    res = {}
    for tile in (0, 64, 128):
        def grp_t(tile=tile):
            ops.gemm_bf16(Ab, Bb, M1 + M2, N, tile=tile, **kw(0, M1 + M2))
        res[tile] = timed(grp_t)
    t_sep = timed(sep)
    best = min(res.values())
    tot_sep += t_sep; tot_grp += best
    print(f"N={N:5d} K={K:5d} {out:2s}: separate {t_sep:6.1f} us | grouped auto {res[0]:6.1f}  64x128 {res[64]:6.1f}  128x128 {res[128]:6.1f} | best saves {t_sep - best:5.1f} us", flush=True)
print(f"per layer (5 NT products of the forward; the 4 dgrads have the same shapes): separate {tot_sep:.1f} us, grouped {tot_grp:.1f} us")
