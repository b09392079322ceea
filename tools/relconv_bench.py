#!/usr/bin/env python3
"""RelGAN_D convolution bank at the bench shape (B=256, L=18, E=128, R=16, F=300): fused kernels per filter size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import ops
from consistent__style_transfer_amd._lib import call

def t_graph(fn, n=10):
    for _ in range(2):
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

B, L, E, R, F = 256, 18, 128, 16, 300
G = B * R
e = torch.randn(B, L, E, device="cuda")
ws = ops._workspace(e.device)
for k in (2, 3, 4, 5):
    KE = k * E // R
    w, b = torch.randn(F, KE, device="cuda") * 0.2, torch.randn(F, device="cuda") * 0.1
    feats, arg = torch.empty(G, F, device="cuda"), torch.empty(G, F, device="cuda", dtype=torch.int32)
    df = torch.randn(G, F, device="cuda")
    de, dw, db = torch.empty_like(e), torch.empty_like(w), torch.empty_like(b)
    f = t_graph(lambda: call("cst_relconv_fwd", e, B, L, E, R, k, w, b, F, feats, F, arg))
    bi = t_graph(lambda: call("cst_relconv_bwd_input", df, F, feats, F, arg, w, B, L, E, R, k, F, de, 0))
    bw = t_graph(lambda: call("cst_relconv_bwd_weight", df, F, feats, F, arg, e, B, L, E, R, k, F, dw, db, ws, ops.WS_FLOATS))
    print(f"k={k} KE={KE} T={L - k + 1}: fwd {f:6.1f} us   bwd_input {bi:6.1f} us   bwd_weight {bw:6.1f} us")
