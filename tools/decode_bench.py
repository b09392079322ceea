#!/usr/bin/env python3
"""The decode step's kernels one by one, each as a chain of 50 dependent launches inside a hipGraph (what the training step replays):
time per launch, next to a one-block fill (the fixed cost of a dependent launch).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import ops
from consistent__style_transfer_amd._lib import call, call_plain

B, E, Hd, V, L, T = 256, 128, 512, 10000, 18, 18
dev = "cuda"
W_ = 2 * Hd
f32 = lambda *s: torch.randn(*s, device=dev)
i16 = lambda *s: torch.zeros(*s, device=dev, dtype=torch.int16)
XH = f32(B, E + Hd)
XHb, _ = ops.cast_bf16(XH, want_t=False)
wcat_b, _ = ops.cast_bf16(f32(4 * Hd, E + Hd) * 0.05, want_t=False)
fn1_b, _ = ops.cast_bf16(f32(Hd, W_) * 0.05, want_t=False)
fn2_b, _ = ops.cast_bf16(f32(V, Hd) * 0.05, want_t=False)
table, bias, c_prev = f32(V, E), f32(4 * Hd), f32(B, Hd)
gates, c_out, h_out = f32(B, 4 * Hd), f32(B, Hd), f32(B, W_)
hb, xb = i16(B, Hd), i16(B, E)
NG = call_plain("cst_argmax_groups")
amax = torch.zeros(NG, B, device=dev, dtype=torch.int64)
mem, patt = f32(B, L, Hd), f32(B, L)
idb, r1, r1b = i16(B, W_), f32(B, Hd), i16(B, Hd)
out = f32(B, V)
small = torch.zeros(64, device=dev)
d = ops.Drop(0.1, 3, 7)


def k_fill():
    call("cst_zero", small, 256)


def k_gates():
    call("cst_dec_gates", XHb, XHb.stride(0), wcat_b, wcat_b.stride(0), amax, None, 0, None, table, E, V, *d.args(), xb, E, bias, c_prev, Hd,
         gates, 4 * Hd, c_out, Hd, h_out, W_, hb, Hd, B, E, Hd)


def k_attn():
    call("cst_dec_attn", h_out, W_, mem, h_out[:, Hd:], W_, patt, B, L, Hd, idb, W_, *d.args())


def k_fn1():
    call("cst_gemm_bf16_skinny", idb, W_, fn1_b, fn1_b.stride(0), r1, Hd, r1b, Hd, B, Hd, W_, bias[:Hd], 2, *ops.NO_DROP.args())


def k_fn2():
    call("cst_dec_fn2", r1b, Hd, fn2_b, fn2_b.stride(0), out, V, B, V, Hd, amax)


def k_fn2_generic():
    call("cst_gemm_bf16_argmax", r1b, Hd, fn2_b, fn2_b.stride(0), out, V, B, V, Hd, amax)


def k_step():
    k_gates(); k_attn(); k_fn1(); k_fn2()


for name, fn, per in (("one-block fill", k_fill, 1), ("dec_gates", k_gates, 1), ("dec_attn", k_attn, 1), ("skinny fn_1", k_fn1, 1), ("dec_fn2", k_fn2, 1),
                      ("fn_2 generic + argmax", k_fn2_generic, 1), ("whole step (4 launches)", k_step, 4)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    n = 50
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record(); torch.cuda.synchronize()
    print(f"{name:28s} {a.elapsed_time(b) * 1000 / (5 * n):7.2f} us per call ({per} launch{'es' if per > 1 else ''})", flush=True)
