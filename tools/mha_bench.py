#!/usr/bin/env python3
"""Encoder self-attention core (cst_mha_fwd / cst_mha_bwd) at the bench shapes: time per launch and the
HBM-traffic floor (q,k,v,dO in + dq,dk,dv out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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

for B, S, H, hd, p in [(256, 18, 8, 64, 0.1), (256, 36, 8, 64, 0.1), (256, 18, 8, 64, 0.0), (512, 60, 8, 64, 0.1), (256, 18, 8, 96, 0.1)]:
    d = H * hd
    qkv = torch.randn(B, S, 3 * d, device="cuda")
    out = torch.empty(B, S, d, device="cuda"); lse = torch.empty(B, H, S, device="cuda")
    dout = torch.randn(B, S, d, device="cuda"); dqkv = torch.empty_like(qkv)
    f = t_graph(lambda: call("cst_mha_fwd", qkv, out, lse, B, S, H, hd, p, 1, 2, None))
    bw = t_graph(lambda: call("cst_mha_bwd", qkv, dout, lse, dqkv, B, S, H, hd, p, 1, 2, None))
    fb, bb = 4 * (qkv.numel() + out.numel()), 4 * (2 * qkv.numel() + dout.numel())
    print(f"B={B} S={S} H={H} hd={hd} p={p}: fwd {f:6.1f} us ({fb / f / 1e3:6.0f} GB/s)   bwd {bw:6.1f} us ({bb / bw / 1e3:6.0f} GB/s)")

# bf16-I/O forms (the ones the bf16 precision mode runs): cst_mha_fwd_h / cst_mha_bwd_h; CST_MHA_HB_OFF=1 keeps fp32 LDS images
print("---- bf16 I/O (cst_mha_fwd_h / cst_mha_bwd_h)" + (" [fp32 LDS images]" if os.environ.get("CST_MHA_HB_OFF") else " [bf16 LDS images in the backward]"))
for B, S, H, hd, p in [(256, 18, 8, 96, 0.1), (256, 36, 8, 96, 0.1), (256, 18, 8, 64, 0.1), (256, 36, 8, 64, 0.1), (512, 60, 8, 64, 0.1)]:
    d = H * hd
    qb = torch.randn(B * S, 3 * d, device="cuda").to(torch.bfloat16).view(torch.int16)
    wb = torch.randn(B * S, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    lse = torch.empty(B * H * S, device="cuda")
    ob = torch.empty(B * S, d, device="cuda", dtype=torch.int16)
    dqb = torch.empty(B * S, 3 * d, device="cuda", dtype=torch.int16)
    f = t_graph(lambda: call("cst_mha_fwd_h", qb, None, lse, B, S, H, hd, p, 1, 2, None, ob, d))
    bw = t_graph(lambda: call("cst_mha_bwd_h", qb, wb, lse, None, B, S, H, hd, p, 1, 2, None, dqb, 3 * d))
    fb, bb = 2 * (qb.numel() + ob.numel()), 2 * (2 * qb.numel() + wb.numel())
    print(f"B={B} S={S} H={H} hd={hd} p={p}: fwd {f:6.1f} us ({fb / f / 1e3:6.0f} GB/s)   bwd {bw:6.1f} us ({bb / bw / 1e3:6.0f} GB/s)")
