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


# ---- the soft decode's backward step ------------------------------------------------------------------------------------------
T = 21
dXH, dxe = f32(B, E + Hd), f32(B, E)
etok_b, _ = ops.cast_bf16(table, want_t=False)
dout = f32(B, T * V)
probs = torch.softmax(f32(B, V), -1)
dlb = i16(B, (V + 63) // 64 * 64)
fn2_t = ops.cast_bf16(f32(V, Hd) * 0.05)[1]
fn1_t = ops.cast_bf16(f32(Hd, W_) * 0.05)[1]
wcat_t = ops.cast_bf16(f32(4 * Hd, E + Hd) * 0.05)[1]
dp1, dp1b = f32(B, Hd), i16(B, Hd)
diffn, ds_all, dgd, dgb, dc = f32(B, W_), f32(B, L), f32(B, 4 * Hd), i16(B, 4 * Hd), f32(B, Hd)
acts, dmem = torch.sigmoid(f32(B, 4 * Hd)), f32(B, L, Hd)


def k_dxe():
    call("cst_dec_dxe", dXH, E + Hd, dxe, E, etok_b, etok_b.stride(0), dout, T * V, B, V, E, *d.args())


def k_dxe_old():
    ops.dropout2d(dXH[:, :E], d, out=dxe)
    ops.gemm(dxe, True, table, True, dout[:, :V], B, V, E, accumulate=True)


def k_smbwd():
    ops.softmax_tau_bwd(probs, dout[:, :V], 10.0, dout[:, :V], dx_b=dlb)


def k_fn2t():
    ops.gemm_bf16(dlb, fn2_t, B, Hd, C=dp1, Cb=dp1b, aux=r1b, act=4, tile=int(os.environ.get("DB_TILE", 0)), splitk=int(os.environ.get("DB_SPLIT", 0)))


def k_fn1t():
    call("cst_gemm_bf16_skinny", dp1b, Hd, fn1_t, fn1_t.stride(0), diffn, W_, None, 0, B, W_, Hd, None, 0, *d.args())


def k_attn_cell_bwd():
    call("cst_dec_attn_cell_bwd", diffn, W_, mem, patt, ds_all, B, L, Hd, acts, 4 * Hd, c_prev, Hd, c_out, Hd, dXH[:, E:], E + Hd, dc, Hd,
         dgd, 4 * Hd, dc, Hd, dgb, 4 * Hd)


def k_attn_cell_bwd_old():
    call("cst_dot_attn_bwd", diffn[:, Hd:], W_, h_out, W_, mem, patt, diffn[:, :Hd], W_, 1, dmem, B, L, Hd)
    call("cst_lstm_cell_bwd", acts, 4 * Hd, c_prev, Hd, c_out, Hd, diffn, W_, dXH[:, E:], E + Hd, dc, Hd, dgd, 4 * Hd, dc, Hd, dgb, 4 * Hd, B, Hd)


def k_wcat_t():
    ops.gemm_bf16(dgb, wcat_t, B, E + Hd, C=dXH, tile=int(os.environ.get("DB_TILE", 0)), splitk=int(os.environ.get("DB_SPLIT", 0)))


def k_bwd_step():
    k_dxe(); k_smbwd(); k_fn2t(); k_fn1t(); k_attn_cell_bwd(); k_wcat_t()


ONLY = os.environ.get("DB_ONLY")
for name, fn, per in (("one-block fill", k_fill, 1), ("dec_gates", k_gates, 1), ("dec_attn", k_attn, 1), ("skinny fn_1", k_fn1, 1), ("dec_fn2", k_fn2, 1),
                      ("fn_2 generic + argmax", k_fn2_generic, 1), ("whole step (4 launches)", k_step, 4),
                      ("bwd: dxe (dropout + product + accumulate)", k_dxe, 1), ("bwd: dropout + fp32-staged dx E^T (old)", k_dxe_old, 2),
                      ("bwd: softmax_tau_bwd", k_smbwd, 1), ("bwd: fn_2^T dgrad (split-K + reduce)", k_fn2t, 2), ("bwd: fn_1^T dgrad skinny", k_fn1t, 1),
                      ("bwd: attention + cell", k_attn_cell_bwd, 1), ("bwd: attention, cell (old)", k_attn_cell_bwd_old, 2),
                      ("bwd: dgates [W_ih | W_hh] (split-K + reduce)", k_wcat_t, 2), ("bwd: whole soft step", k_bwd_step, 8)):
    if ONLY and ONLY not in name:
        continue
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
    print(f"{name:46s} {a.elapsed_time(b) * 1000 / (5 * n):7.2f} us per call ({per} launch{'es' if per > 1 else ''})", flush=True)
