"""DenoiseLSTM forward / backward as ONE autograd node over the HIP kernels.

Restates reference src/model/rnn.py:55-98 (BiLSTM encoder, attention-LSTM decoder loop,
temperature-softmax straight-through sampling, scheduled-sampling teacher forcing) with explicit
back-propagation through time.  The per-step work is: one gate GEMM over [x_t | h_{t-1}], the
fused LSTM cell, single-query attention, fn_1 (+LeakyReLU), fn_2 straight into the stacked
(B,T,V) output, then softmax/argmax and the embedding gather for the next input.  Weight
gradients are NOT accumulated step by step: per-step activations and gate gradients are kept and
each weight gets one large wgrad GEMM after the loop.
"""
import os

import torch

from . import ops
from ._lib import call, call_plain
from .ops import (NO_DROP, STREAM_G_EMB_IN, STREAM_G_FFN, STREAM_G_XT, act_bwd, argmax_rows, axpby, cast_bf16, colsum,
                  dgrad, dropout2d, embed_gather, embed_scatter_add, gemm, gemm_bf16, linear_fwd, softmax_tau,
                  softmax_tau_bwd, weight_bf16, wgrad)

PARAM_KEYS = (
    "start_embedding.weight", "token_embedding.weight", "enc_style_embedding.weight", "style_embedding.weight",
    "encoder.weight_ih_l0", "encoder.weight_hh_l0", "encoder.bias_ih_l0", "encoder.bias_hh_l0",
    "encoder.weight_ih_l0_reverse", "encoder.weight_hh_l0_reverse", "encoder.bias_ih_l0_reverse",
    "encoder.bias_hh_l0_reverse",
    "decoder.weight_ih_l0", "decoder.weight_hh_l0", "decoder.bias_ih_l0", "decoder.bias_hh_l0",
    "transfer.weight", "fn_1.weight", "fn_1.bias", "fn_2.weight",
)


def _new(dev, *shape, dtype=torch.float32):
    return torch.empty(*shape, device=dev, dtype=dtype)


def _st(t):
    return t.stride(0) if t is not None else 0


def _cell_fwd(gates, c_prev, h_out, c_out, h_out2, B, H, hb=None, hb2=None):
    call("cst_lstm_cell_fwd", gates, gates.stride(0), c_prev, c_prev.stride(0), h_out, h_out.stride(0),
         c_out, c_out.stride(0), h_out2, _st(h_out2), hb, _st(hb), hb2, _st(hb2), B, H)


def _gemm_cell_fwd(probs, B, H):
    """Gate product + cell in two launches for one or two problems (dicts with Ab, Bb, gates, c_prev, h_out,
    c_out and optional bias, addend, h_out2, hb, hb2); both problems share shapes and leading dimensions."""
    p, q = probs[0], (probs[1] if len(probs) > 1 else {})
    g = lambda d, k: d.get(k)
    call("cst_gemm_bf16_lstm", p["Ab"], p["Ab"].stride(0), p["Bb"], p["Bb"].stride(0), B, H, p["Ab"].shape[1],
         g(p, "bias"), g(p, "addend"), _st(g(p, "addend")),
         p["gates"], p["gates"].stride(0), p["c_prev"], p["c_prev"].stride(0), p["h_out"], p["h_out"].stride(0),
         p["c_out"], p["c_out"].stride(0), g(p, "h_out2"), _st(g(p, "h_out2")), g(p, "hb"), _st(g(p, "hb")), g(p, "hb2"), _st(g(p, "hb2")),
         g(q, "Ab"), g(q, "Bb"), g(q, "bias"), g(q, "addend"), g(q, "gates"), g(q, "c_prev"), g(q, "h_out"), g(q, "c_out"),
         g(q, "h_out2"), g(q, "hb"), g(q, "hb2"),
         ops.LSTM_SPLITK, ops._workspace(p["gates"].device), ops.WS_FLOATS)


def _gemm_cell_bwd(probs, B, H):
    """dh = dgates_next W_hh (+ dh_extra) and the cell backward of this step, two launches, one or two problems
    (dicts with Ab, Bb, gates, c_prev, c_new, dgates, dc_prev and optional dh_extra, dc_in, dgb)."""
    p, q = probs[0], (probs[1] if len(probs) > 1 else {})
    g = lambda d, k: d.get(k)
    call("cst_gemm_bf16_lstm_bwd", p["Ab"], p["Ab"].stride(0), p["Bb"], p["Bb"].stride(0), B, H, p["Ab"].shape[1],
         p["gates"], p["gates"].stride(0), p["c_prev"], p["c_prev"].stride(0), p["c_new"], p["c_new"].stride(0),
         g(p, "dh_extra"), _st(g(p, "dh_extra")), g(p, "dc_in"), _st(g(p, "dc_in")),
         p["dgates"], p["dgates"].stride(0), p["dc_prev"], p["dc_prev"].stride(0), g(p, "dgb"), _st(g(p, "dgb")),
         g(q, "Ab"), g(q, "Bb"), g(q, "gates"), g(q, "c_prev"), g(q, "c_new"), g(q, "dh_extra"), g(q, "dc_in"),
         g(q, "dgates"), g(q, "dc_prev"), g(q, "dgb"),
         p.get("n_extra", 0), g(p, "extra_out"), g(q, "extra_out"), _st(g(p, "extra_out")),
         ops.LSTM_SPLITK, ops._workspace(p["gates"].device), ops.WS_FLOATS)


def _cell_bwd(gates, c_prev, c_new, dh, dh2, dc, dgates, dc_prev, B, H, dgb=None):
    call("cst_lstm_cell_bwd", gates, gates.stride(0), c_prev, c_prev.stride(0), c_new, c_new.stride(0),
         dh, _st(dh), dh2, _st(dh2), dc, _st(dc), dgates, dgates.stride(0), dc_prev, dc_prev.stride(0),
         dgb, _st(dgb), B, H)


_XCHG = {}
_SPLIT = {"disabled": False, "why": None}


def disable_split(why):
    """Use the one-workgroup encoder kernel (cst_lstm_seq_fwd) for the rest of the process: called when the split kernel's workgroups
    were ever not resident together (Trainer's probe, bench.py's probe)."""
    _SPLIT["disabled"], _SPLIT["why"] = True, why


def split_enabled(B, dev=None):
    """May the encoder forward of batch B run as cst_lstm_seq_fwd_split (two workgroups per row group exchanging h_t inside the launch)?
    Its 4 B / 16 workgroups need a whole CU each and must be resident TOGETHER, so: never after a timeout was seen in this process;
    never beyond the device's CU count (asked from the library: hipDeviceAttributeMultiprocessorCount); and with other ranks' collectives
    possibly resident on the same device (world > 1: RCCL kernels hold CUs while a stage's all-reduces run), only up to half of the
    CUs.  CST_LSTM_SPLIT=0 switches it off by hand."""
    if _SPLIT["disabled"] or os.environ.get("CST_LSTM_SPLIT", "1") == "0":
        return False
    cap = call_plain("cst_lstm_seq_split_capacity")
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        cap //= 2
    return 0 < call_plain("cst_lstm_seq_split_workgroups", B) <= cap


def _xchg_key(dev, B, sid):
    return (dev.index if dev.index is not None else torch.cuda.current_device(), B, sid)


def _xchg_alloc(key, dev):
    nb = call_plain("cst_lstm_seq_xchg_bytes", key[1])
    ws = _XCHG[key] = torch.empty(nb, device=dev, dtype=torch.uint8)
    call("cst_zero", ws, nb)
    return ws


def _xchg_workspace(dev, B):
    """Exchange workspace of cst_lstm_seq_fwd_split for batch B on `dev`: one persistent buffer per (device, B, stream) -- launches on a stream
    are ordered and the entry point zeroes it in front of each.  Its last 16 bytes are the timeout word check_exchange_timeouts() reads."""
    # one buffer per (device, B, stream): launches on ONE stream are ordered, two eager streams (stages.Fork) must not share granules.  A
    # capture uses the (device, B, 0) buffer, which must exist BEFORE the capture begins (reserve_capture_workspaces, called by
    # graphs.GraphedStep after its eager pass): a buffer created INSIDE a capture would have its zero fill -- sticky timeout word
    # included -- replayed with the graph, erasing earlier reports, and would live in the graph's private pool.
    if torch.cuda.is_current_stream_capturing():
        ws = _XCHG.get(_xchg_key(dev, B, 0))
        if ws is None:
            raise RuntimeError(f"cst_lstm_seq_fwd_split: no exchange workspace for batch {B} was reserved before this capture "
                               "(gen_fn.reserve_capture_workspaces() after one eager pass of the step)")
        return ws
    key = _xchg_key(dev, B, int(torch.cuda.current_stream(dev).cuda_stream))
    ws = _XCHG.get(key)
    return ws if ws is not None else _xchg_alloc(key, dev)


def reserve_capture_workspaces():
    """Outside any capture: make sure every (device, batch) the eager passes have used also has the buffer captured launches take."""
    assert not torch.cuda.is_current_stream_capturing()
    for (idx, B, sid), ws in list(_XCHG.items()):
        if sid != 0 and (idx, B, 0) not in _XCHG:
            _xchg_alloc((idx, B, 0), ws.device)


def exchange_timed_out(clear=False):
    """(device, batch) keys of the split-encoder workspaces whose sticky timeout word is set (reads the device: synchronises)."""
    bad = [k[:2] for k, ws in _XCHG.items() if int(ws[-16:].view(torch.int32)[0].item()) != 0]
    if clear:
        for ws in _XCHG.values():
            ws[-16:].zero_()
    return bad


def check_exchange_timeouts():
    """Raise if a workgroup of the split encoder kernel ever gave up waiting for its partner (the kernel then poisoned that row group's
    encoder states with NaN, so the step's loss is NaN as well).  Reads the device: call at synchronisation points only (every loss
    readback, validation, the end of the transfer writer, the end of bench.py).  The split kernel is switched off for the rest of
    the process before raising, so a caller that catches the error continues on the one-workgroup kernel."""
    bad = exchange_timed_out()
    if bad:
        idx, B = bad[0]
        disable_split(f"timeout on device {idx}, batch {B}")
        raise RuntimeError(f"cst_lstm_seq_fwd_split (device {idx}, batch {B}): a workgroup timed out waiting for its partner's hidden states "
                           "(its encoder states were set to NaN); the split kernel is now disabled in this process")


def probe_split():
    """After ONE eager pass through the generator (a sanity validation, a first step): if a workgroup of the split kernel gave up, clear
    the report, fall back to the one-workgroup kernel for the rest of the process and say so.  -> True when the split kernel stays on."""
    bad = exchange_timed_out(clear=True)
    if bad:
        disable_split(f"probe: timeout on (device, batch) {bad}")
        import sys
        print(f"[cst] cst_lstm_seq_fwd_split timed out in the probe pass {bad}: using cst_lstm_seq_fwd (one workgroup per row group) "
              "for the rest of this process", file=sys.stderr, flush=True)
        return False
    return True


def _lstm_frag_order(wb, H):
    """bf16 W_hh [4H, H] (K contiguous) -> the MFMA-fragment order cst_lstm_seq_fwd streams:
    [wave][gate][tile][k step][lane = 16 lq + lr][8], element = W_hh[q*H + 64w + 16j + lr][32kk + 8lq + e]."""
    hit = getattr(wb, "_cst_frag", None)          # parked on the bf16 twin: ops.weight_bf16 drops every _cst_* attribute when it rewrites the twin
    if hit is None:
        v = wb[:, :H].reshape(4, 4, H // 64, 16, H // 32, 4, 8)       # (q, w, j, lr, kk, lq, e)
        hit = wb._cst_frag = v.permute(1, 0, 2, 4, 5, 3, 6).contiguous()   # (w, q, j, kk, lq, lr, e)
    return hit


def _lstm_frag_order_t(wt, H):
    """bf16 W_hh^T [H, 4H] (K = 4H contiguous) -> the fragment order cst_lstm_seq_bwd streams:
    [wave][k step][tile][lane = 16 lq + lr][8], element = W_hh^T[64w + 16j + lr][32kk + 8lq + e]."""
    hit = getattr(wt, "_cst_frag", None)
    if hit is None:
        v = wt[:, :4 * H].reshape(4, H // 64, 16, 4 * H // 32, 4, 8)  # (w, j, lr, kk, lq, e)
        hit = wt._cst_frag = v.permute(0, 3, 1, 4, 2, 5).contiguous()      # (w, kk, j, lq, lr, e)
    return hit


def _bf16_ok(*dims):
    """The direct-to-LDS GEMM needs every reduction length to be a multiple of 64 (one 128-byte LDS
    row of bf16); the reference's module constants are, toy test sizes are not."""
    return not ops._STATE["f32"] and all(d % 64 == 0 for d in dims)


def _i16(dev, *shape):
    return torch.empty(*shape, device=dev, dtype=torch.int16)


_GRAD_SHARE = [None]


class shared_param_grads:
    """`with shared_param_grads():` around SEVERAL generator calls whose outputs feed ONE backward pass (the optimize stage's generator
    step decodes twice with the same parameters, main_optimize.py:97 and :104): the backward of the call that runs first keeps its parameter
    gradients, the backward of every other call adds its own into those tensors with one multi-tensor launch and hands autograd nothing
    -- instead of autograd summing the two sets parameter by parameter (44 small add launches per step).  Same sums,
    same order of the two addends."""

    def __enter__(self):
        self.prev = _GRAD_SHARE[0]
        _GRAD_SHARE[0] = {}
        return self

    def __exit__(self, *exc):
        _GRAD_SHARE[0] = self.prev
        return False


class GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, label_i, x, label, coins, cfg, *params):
        ctx.share = _GRAD_SHARE[0]
        """inp: (B,L') int64 ids or (B,L',V) fp32 probabilities; x: (B,T) int64 or None;
        coins: (T,) int32 device tensor (1 = feed back argmax) or None;
        cfg: dict(mode 'none'|'softmax', tau, max_len, drop Drop or NO_DROP)."""
        P = dict(zip(PARAM_KEYS, params))
        E_tok = P["token_embedding.weight"]
        V, E = E_tok.shape
        H = P["encoder.weight_hh_l0"].shape[1]
        Hd = P["decoder.weight_hh_l0"].shape[1]
        assert Hd == 2 * H, "decoder width must equal the BiLSTM memory width (rnn.py:46-50 dot attention)"
        dev = E_tok.device
        B, Lp = inp.shape[0], inp.shape[1]
        mode, tau, drop = cfg["mode"], float(cfg["tau"]), cfg["drop"]
        soft = mode == "softmax"
        T = int(cfg["max_len"]) if x is None else x.shape[1]
        inv_tau = 1.0 / tau
        label_i = label_i.contiguous()
        label = label.contiguous()

        # ---- encoder input embedding (rnn.py:58-61) ------------------------------------------
        emb = _new(dev, B * Lp, E)
        use_b = _bf16_ok(H, 4 * H, E + Hd, 4 * Hd, Hd + 2 * H, Hd)      # bf16-operand GEMMs for the recurrent products
        seq_fused = use_b and H == 256 and B % 16 == 0                  # whole-sequence encoder kernels
        # the encoder's input-side products (x W_ih^T, d emb, dW_ih, dW_hh) on bf16 twins written by their producers
        enc_b = seq_fused and E % 64 == 0 and (B * Lp) % 64 == 0
        embb = _i16(dev, B * Lp, E) if enc_b else None
        if inp.dim() == 2:
            ids_in = inp.contiguous().reshape(-1)
            embed_gather(E_tok, emb, ids_a=ids_in, drop=drop.at(STREAM_G_EMB_IN), out_b=embb)
            in_drop = drop.at(STREAM_G_EMB_IN)
        else:
            ids_in = argmax_rows(inp.reshape(B * Lp, V))        # hard_sample(inp) @ E == row gather
            embed_gather(E_tok, emb, ids_a=ids_in, out_b=embb)
            in_drop = NO_DROP

        # ---- BiLSTM encoder (rnn.py:57,62) ---------------------------------------------------
        h0cat = _new(dev, B, 2 * H)
        embed_gather(P["enc_style_embedding.weight"], h0cat, ids_a=label_i)
        memory = _new(dev, B, Lp, 2 * H)
        hprev = _new(dev, 2, B, Lp, H)                # h_{t_prev} per (dir, b, t): B operand of dW_hh
        genc = _new(dev, 2, Lp, B, 4 * H)             # gate activations
        cenc = _new(dev, 2, Lp, B, H)
        c_cat = _new(dev, B, 2 * H)
        zeros_c = ops.zeros(B, H, device=dev)
        mem2 = memory.view(B, Lp * 2 * H)
        W_ = Hd + 2 * H
        hprevb = _i16(dev, 2, B, Lp, H) if enc_b else None
        memb = _i16(dev, B, Lp * 2 * H) if use_b else None     # bf16 copy of the encoder states (A operand of h W_hh^T)
        enc = []
        for d, suf in enumerate(("", "_reverse")):
            w_ih, w_hh = P["encoder.weight_ih_l0" + suf], P["encoder.weight_hh_l0" + suf]
            bsum = axpby(P["encoder.bias_ih_l0" + suf].view(1, -1), 1.0, P["encoder.bias_hh_l0" + suf].view(1, -1), 1.0).view(-1)
            if enc_b:
                xp = gemm_bf16(embb, weight_bf16(w_ih)[0], B * Lp, 4 * H, C=_new(dev, B * Lp, 4 * H), bias=bsum).view(B, Lp * 4 * H)
            else:
                xp = linear_fwd(emb, w_ih, bsum).view(B, Lp * 4 * H)        # (B, L', 4H)
            order = list(range(Lp)) if d == 0 else list(range(Lp - 1, -1, -1))
            enc.append((w_hh, weight_bf16(w_hh)[0] if use_b else None, xp, order, hprev[d].view(B, Lp * H)))
        if seq_fused:
            # both directions, all L' steps, one launch: 16 batch rows per workgroup (cst_lstm_seq_fwd: no inter-workgroup dependencies;
            # cst_lstm_seq_fwd_split: two workgroups per row group that exchange their halves of h_t every step)
            (_, wb0, xp0, _, _), (_, wb1, xp1, _, _) = enc
            seq_args = (_lstm_frag_order(wb0, H), _lstm_frag_order(wb1, H), xp0, xp1, h0cat, 2 * H, genc[0], genc[1], cenc[0], cenc[1],
                        hprev[0], hprev[1], hprevb[0] if enc_b else None, hprevb[1] if enc_b else None, c_cat, 2 * H, memory, memb, B, Lp, H)
            if split_enabled(B, dev):
                # two workgroups per row group, W_hh resident on chip, one 4 KB exchange of h_t per step (lstm_seq.hip)
                xchg = _xchg_workspace(dev, B)
                call("cst_lstm_seq_fwd_split", *seq_args, xchg, xchg.numel())
            else:
                call("cst_lstm_seq_fwd", *seq_args)
        for n in (() if seq_fused else range(Lp)):
            probs = []
            for d, (w_hh, whh_b, xp, order, hp2) in enumerate(enc):
                t = order[n]
                if n == 0:
                    h_in = h0cat[:, d * H:(d + 1) * H]
                    c_in = zeros_c
                    axpby(h_in, 1.0, out=hp2[:, t * H:(t + 1) * H])
                    h_in_b = cast_bf16(h_in, want_t=False)[0] if use_b else None
                else:
                    tp = order[n - 1]
                    h_in = mem2[:, tp * 2 * H + d * H: tp * 2 * H + (d + 1) * H]
                    c_in = cenc[d, tp]
                    h_in_b = memb[:, tp * 2 * H + d * H: tp * 2 * H + (d + 1) * H] if use_b else None
                g = genc[d, t]
                last = n == Lp - 1
                c_out = c_cat[:, d * H:(d + 1) * H] if last else cenc[d, t]
                h_next = None if last else hp2[:, order[n + 1] * H:(order[n + 1] + 1) * H]
                h_t = mem2[:, t * 2 * H + d * H: t * 2 * H + (d + 1) * H]
                xa = xp[:, t * 4 * H:(t + 1) * 4 * H]
                if use_b:
                    probs.append(dict(Ab=h_in_b, Bb=whh_b, gates=g, c_prev=c_in, h_out=h_t, c_out=c_out, h_out2=h_next, addend=xa,
                                      hb=memb[:, t * 2 * H + d * H: t * 2 * H + (d + 1) * H]))
                else:
                    gemm(h_in, True, w_hh, True, g, B, 4 * H, H, addend=xa)
                    _cell_fwd(g, c_in, h_t, c_out, h_next, B, H)
            if use_b:
                _gemm_cell_fwd(probs, B, H)            # both directions in one pair of launches (same shapes and strides)

        # ---- decoder initial state (rnn.py:67-69) --------------------------------------------
        c0 = linear_fwd(c_cat, P["transfer.weight"], None, act=2)
        h0d = _new(dev, B, Hd)
        embed_gather(P["style_embedding.weight"], h0d, ids_a=label)
        wcat = torch.cat([P["decoder.weight_ih_l0"], P["decoder.weight_hh_l0"]], dim=1).contiguous()   # (4Hd, E+Hd)
        bdec = axpby(P["decoder.bias_ih_l0"].view(1, -1), 1.0, P["decoder.bias_hh_l0"].view(1, -1), 1.0).view(-1)
        XH = _new(dev, T, B, E + Hd)                  # [x_t | h_{t-1}]
        zero_ids = ops.zeros(B, device=dev, dtype=torch.int64)
        embed_gather(P["start_embedding.weight"], XH[0][:, :E], ids_a=zero_ids)
        axpby(h0d, 1.0, out=XH[0][:, E:])
        gdec = _new(dev, T, B, 4 * Hd)
        cdec = _new(dev, T, B, Hd)
        if use_b:
            wcat_b, wcat_t = cast_bf16(wcat)          # [4Hd, E+Hd] and its transpose for the dgrad
            fn1_b = weight_bf16(P["fn_1.weight"])[0]
            fn2_b = weight_bf16(P["fn_2.weight"])[0]
            XHb = _i16(dev, T, B, E + Hd)
            call("cst_cast_bf16", XH[0], 0, E + Hd, B, E + Hd, XHb[0], E + Hd, None, 0, 0.0, 0, 0, None)
            ifdb = _i16(dev, B, T * W_)               # bf16 dropout(i_ffn), written by the attention kernel
            r1b = _i16(dev, B, T * Hd)
        else:
            wcat_t = XHb = ifdb = r1b = None
        iffn = _new(dev, B, T, W_)                    # [h_t | a_t], batch-major like `out`
        iffn_d = _new(dev, B, T, W_) if drop.p > 0 else iffn
        if2, ifd2 = iffn.view(B, T * W_), iffn_d.view(B, T * W_)
        patt = _new(dev, T, B, Lp)
        r1 = _new(dev, B, T, Hd)
        out = _new(dev, B, T, V)
        ids_fb = _new(dev, T, B, dtype=torch.int64)
        out2 = out.view(B, T * V)
        r12 = r1.view(B, T * Hd)
        x_c = x.contiguous() if x is not None else None
        # soft decode: the softmax kernel also writes the bf16 twin of the distributions, the operand of the critics'
        # shared soft-embedding product over all steps (ops.SharedSoftEmbedFn)
        Vp = (V + 63) // 64 * 64
        outb = _i16(dev, B, T * Vp) if (soft and use_b and V % 4 == 0 and V <= 16384) else None
        # Fast decode loop (csrc/decode.hip): four launches per step, none a split-K reduce -- cst_dec_gates (token choice + embedding +
        # gates + cell), cst_dot_attn_fwd, cst_gemm_bf16_skinny (fn_1), cst_gemm_bf16_argmax (fn_2 + the row arg-max the next step feeds
        # back).  The soft decode's softmax runs ONCE after the loop over all T steps (the recurrence only needs the arg-max).
        fast = (use_b and E == 128 and Hd % 16 == 0 and (B * T) % 64 == 0 and W_ % 64 == 0 and W_ <= 1280 and Hd % 32 == 0
                and V % 4 == 0 and V <= 16384 and call_plain("cst_dec_gates_lds_bytes", E, Hd) <= 160 * 1024 and os.environ.get("CST_DECODE_SLOW") != "1")
        if fast:
            NG = call_plain("cst_argmax_groups")
            amax = ops.zeros(T, NG, B, device=dev, dtype=torch.int64)      # packed (logit, first index) words: NG per row and step
            for s in range(T):
                c_in = c0 if s == 0 else cdec[s - 1]
                i_s = if2[:, s * W_:(s + 1) * W_]
                idb_s = ifdb[:, s * W_:(s + 1) * W_]
                hb_next = XHb[s + 1][:, E:] if s + 1 < T else None
                teacher = x_c[:, s - 1] if (s > 0 and not soft and x_c is not None) else None
                xd = drop.at(STREAM_G_XT + s - 1) if s > 0 else NO_DROP
                call("cst_dec_gates", XHb[s], XHb[s].stride(0), wcat_b, wcat_b.stride(0),
                     amax[s - 1] if s > 0 else None, teacher, T if teacher is not None else 0,
                     coins[s - 1:s] if teacher is not None else None, E_tok, E_tok.stride(0), V, *xd.args(),
                     XHb[s] if s > 0 else None, XHb[s].stride(0), bdec, c_in, c_in.stride(0),
                     gdec[s], gdec[s].stride(0), cdec[s], cdec[s].stride(0), i_s[:, :Hd], T * W_, hb_next, _st(hb_next), B, E, Hd)
                if Hd == 512 and Lp <= 64:
                    call("cst_dec_attn", i_s[:, :Hd], T * W_, memory, i_s[:, Hd:], T * W_, patt[s], B, Lp, Hd,
                         idb_s, T * W_, *drop.at(STREAM_G_FFN + s).args())
                else:
                    call("cst_dot_attn_fwd", i_s[:, :Hd], T * W_, memory, i_s[:, Hd:], T * W_, patt[s], B, Lp, Hd,
                         None, 0, idb_s, T * W_, *drop.at(STREAM_G_FFN + s).args())
                r1b_s = r1b[:, s * Hd:(s + 1) * Hd]
                call("cst_gemm_bf16_skinny", idb_s, T * W_, fn1_b, fn1_b.stride(0), r12[:, s * Hd:(s + 1) * Hd], T * Hd, r1b_s, T * Hd,
                     B, Hd, W_, P["fn_1.bias"], 2, *NO_DROP.args())
                if Hd == 512 and os.environ.get("CST_FN2_GENERIC") != "1":
                    call("cst_dec_fn2", r1b_s, T * Hd, fn2_b, fn2_b.stride(0), out2[:, s * V:(s + 1) * V], T * V, B, V, Hd, amax[s])
                else:
                    call("cst_gemm_bf16_argmax", r1b_s, T * Hd, fn2_b, fn2_b.stride(0), out2[:, s * V:(s + 1) * V], T * V, B, V, Hd, amax[s])
            call("cst_unpack_argmax", amax, ids_fb, B, T)
            if soft:
                o2 = out.view(B * T, V)
                softmax_tau(o2, inv_tau, o2, None, gather=None, p_b=outb.view(B * T, Vp) if outb is not None else None)
        for s in (() if fast else range(T)):
            c_in = c0 if s == 0 else cdec[s - 1]
            h_next = XH[s + 1][:, E:] if s + 1 < T else None
            i_s = if2[:, s * W_:(s + 1) * W_]
            id_s = ifd2[:, s * W_:(s + 1) * W_]
            fd = drop.at(STREAM_G_FFN + s)
            idb_s = ifdb[:, s * W_:(s + 1) * W_] if use_b else None
            if use_b and Hd == 2 * H and Hd <= 1024:
                # gates product, then ONE kernel: split-K sum + cell + attention over the encoder states + dropout(i_ffn)
                hb2 = XHb[s + 1][:, E:] if s + 1 < T else None
                call("cst_gemm_bf16_lstm_attn", XHb[s], XHb[s].stride(0), wcat_b, wcat_b.stride(0), B, Hd, XHb[s].shape[1],
                     bdec, gdec[s], gdec[s].stride(0), c_in, c_in.stride(0), i_s[:, :Hd], T * W_, cdec[s], cdec[s].stride(0),
                     h_next, _st(h_next), hb2, _st(hb2), memory, Lp, i_s[:, Hd:], T * W_, patt[s],
                     id_s if drop.p > 0 else None, T * W_, idb_s, T * W_, *fd.args(),
                     ops.LSTM_SPLITK, ops._workspace(dev), ops.WS_FLOATS)
            else:
                if use_b:
                    _gemm_cell_fwd([dict(Ab=XHb[s], Bb=wcat_b, gates=gdec[s], c_prev=c_in, h_out=i_s[:, :Hd], c_out=cdec[s], h_out2=h_next,
                                         bias=bdec, hb2=XHb[s + 1][:, E:] if s + 1 < T else None)], B, Hd)
                else:
                    gemm(XH[s], True, wcat, True, gdec[s], B, 4 * Hd, E + Hd, bias=bdec)
                    _cell_fwd(gdec[s], c_in, i_s[:, :Hd], cdec[s], h_next, B, Hd)
                call("cst_dot_attn_fwd", i_s[:, :Hd], T * W_, memory, i_s[:, Hd:], T * W_, patt[s], B, Lp, Hd,
                     id_s if drop.p > 0 else None, T * W_, idb_s, T * W_, *fd.args())   # also writes dropout(i_ffn) (+ bf16)
            r1s = r12[:, s * Hd:(s + 1) * Hd]
            o_s = out2[:, s * V:(s + 1) * V]
            if use_b:
                r1b_s = r1b[:, s * Hd:(s + 1) * Hd]
                gemm_bf16(idb_s, fn1_b, B, Hd, C=r1s, Cb=r1b_s, bias=P["fn_1.bias"], act=2)
                gemm_bf16(r1b_s, fn2_b, B, V, C=o_s)
            else:
                linear_fwd(id_s, P["fn_1.weight"], P["fn_1.bias"], act=2, out=r1s)
                linear_fwd(r1s, P["fn_2.weight"], None, out=o_s)
            # the row kernel that finds the argmax also embeds the token fed to step s+1 (rnn.py:88-96)
            fb = None
            if s + 1 < T:
                fb = dict(table=E_tok, out=XH[s + 1][:, :E], out_b=XHb[s + 1][:, :E] if use_b else None, drop=drop.at(STREAM_G_XT + s))
                if not (soft or x_c is None):
                    fb.update(ids_b=x_c[:, s], ldb=T, coin=coins[s:s + 1])
            if soft:
                softmax_tau(o_s, inv_tau, o_s, ids_fb[s], gather=fb, p_b=outb[:, s * Vp:(s + 1) * Vp] if outb is not None else None)
            else:
                argmax_rows(o_s, ids_fb[s], gather=fb)

        if outb is not None:
            ops._side_put(out, outb.view(B * T, Vp))
        ctx.cfg = (B, Lp, T, V, E, H, Hd, soft, inv_tau, drop, in_drop, inp.dim() == 3, use_b)
        ctx.save_for_backward(*params, emb, ids_in, label_i, label, h0cat, memory, hprev, genc, cenc, c_cat, c0,
                              wcat, XH, gdec, cdec, iffn, iffn_d, patt, r1, out, ids_fb, x_c, coins, zeros_c,
                              inp if inp.dim() == 3 else None, wcat_t, r1b, XHb, ifdb, embb, hprevb)
        ctx.mark_non_differentiable(ids_fb)
        return out, ids_fb

    @staticmethod
    def backward(ctx, dout, _dids):
        B, Lp, T, V, E, H, Hd, soft, inv_tau, drop, in_drop, soft_in, use_b = ctx.cfg
        sv = ctx.saved_tensors
        n = len(PARAM_KEYS)
        P = dict(zip(PARAM_KEYS, sv[:n]))
        (emb, ids_in, label_i, label, h0cat, memory, hprev, genc, cenc, c_cat, c0, wcat, XH, gdec, cdec, iffn,
         iffn_d, patt, r1, out, ids_fb, x_c, coins, zeros_c, inp3, wcat_t, r1b, XHb, ifdb, embb, hprevb) = sv[n:]
        dev = dout.device
        E_tok = P["token_embedding.weight"]
        dout = dout.contiguous()                      # (B,T,V); in softmax mode rewritten in place to dlogits
        dout2 = dout.view(B, T * V)
        out2 = out.view(B, T * V)
        r12 = r1.view(B, T * Hd)
        G = {k: None for k in PARAM_KEYS}
        dE = ops.zeros(V, E, device=dev)
        dmem = ops.zeros(B, Lp, 2 * H, device=dev)
        W_ = Hd + 2 * H
        if2 = iffn.view(B, T * W_)
        dpre1 = _new(dev, B, T, Hd)
        dp12 = dpre1.view(B, T * Hd)
        dgd = _new(dev, T, B, 4 * Hd)
        diffn_all = _new(dev, B, T, W_)
        df2 = diffn_all.view(B, T * W_)
        Vp = (V + 63) // 64 * 64
        fn1_t = fn2_t = dgdb_all = dp1b_all = dlb_all = None
        if use_b:
            fn1_t = weight_bf16(P["fn_1.weight"])[1]                # [W_, Hd]
            fn2_t = weight_bf16(P["fn_2.weight"])[1]                # [Hd, up64(V)]
            dgdb_all = _i16(dev, T, B, 4 * Hd)                      # bf16 dgates of every step: dgrad operand now, wgrad operand later
            dp1b_all = _i16(dev, B, T * Hd)                         # bf16 d(fn_1 pre-activation), same role
            if soft:
                # bf16 dlogits of ALL steps (softmax backward writes slice s): per-step dgrad operand now, operand of
                # the fn_2 weight gradient after the loop; only the K-padding columns need zeroing
                dlb_all = _i16(dev, B, T * Vp)
                if Vp != V:
                    dlb_all.view(B, T, Vp)[:, :, V:].zero_()
            else:
                dlb_all = ops._side_take(dout)                       # written by the token-CE kernel next to dlogits
                if dlb_all is not None:
                    dlb_all = dlb_all.view(B, T * Vp)
        dp1b_ok = soft and use_b                          # dp1b_all holds every step's bf16 d(fn_1 pre-activation) after the loop
        if not soft:
            # all dlogits are known up front (token CE): one large dgrad through fn_2 (+LeakyReLU gate)
            # and one through fn_1 instead of T small ones
            if dlb_all is not None:
                gemm_bf16(dlb_all.view(B * T, Vp), fn2_t, B * T, Hd, C=dpre1.view(B * T, Hd), Cb=dp1b_all.view(B * T, Hd),
                          aux=r1b.view(B * T, Hd), act=4)
                gemm_bf16(dp1b_all.view(B * T, Hd), fn1_t, B * T, W_, C=diffn_all.view(B * T, W_))
                dp1b_ok = True
            else:
                dgrad(dout.view(B * T, V), P["fn_2.weight"], out=dpre1.view(B * T, Hd), aux=r1.view(B * T, Hd), act=4)
                dgrad(dpre1.view(B * T, Hd), P["fn_1.weight"], out=diffn_all.view(B * T, W_))
                if ifdb is not None and (B * T) % 64 == 0 and E % 8 == 0 and Hd % 64 == 0:
                    # dout came without bf16 dlogits (any loss other than token_ce(unit_grad=True)): the fn_1 weight gradient below must
                    # still take the bf16 dropout(i_ffn) the forward wrote -- the fast decode loop writes NO fp32 copy of it (iffn_d is
                    # then an unwritten buffer when drop.p > 0) -- so give it the bf16 twin of d(fn_1 pre-activation) here
                    dp1b_all = cast_bf16(dpre1.view(B * T, Hd), want_t=False)[0].view(B, T * Hd)
                    dp1b_ok = True
        dXH_all = _new(dev, T, B, E + Hd)              # step s writes [d x_s | d h_{s-1}] into slice s
        etok_b = weight_bf16(E_tok)[0] if (soft and use_b) else None
        dxe_all = _new(dev, max(T - 1, 1), B, E) if (soft and drop.p > 0) else None
        dc = _new(dev, B, Hd)
        # Teacher-forced / free-running decodes on the bf16 path: every step's FFN gradient is already in
        # diffn_all, so the attention backward of all steps runs as ONE launch (memory tile staged once per
        # batch row, FFN-input dropout applied on the way in) and each step of the recurrence is one launch
        # pair: dgates_{s+1} [W_ih | W_hh] with the cell backward of step s inside its split-K reduce.
        steps_fused = (not soft) and use_b and 2 * Hd <= 1024 and Hd == 2 * H and E % 4 == 0
        if steps_fused:
            call("cst_dot_attn_bwd_steps", df2, T * W_, W_, if2, T * W_, W_, memory, patt, dmem, B, T, Lp, Hd,
                 *drop.at(STREAM_G_FFN).args())
            for s in range(T - 1, -1, -1):
                c_prev = c0 if s == 0 else cdec[s - 1]
                diffn = df2[:, s * W_:(s + 1) * W_]
                if s == T - 1:
                    _cell_bwd(gdec[s], c_prev, cdec[s], diffn[:, :Hd], None, None, dgd[s], dc, B, Hd, dgb=dgdb_all[s])
                else:
                    _gemm_cell_bwd([dict(Ab=dgdb_all[s + 1], Bb=wcat_t, gates=gdec[s], c_prev=c_prev, c_new=cdec[s], dh_extra=diffn[:, :Hd],
                                         dc_in=dc, dgates=dgd[s], dc_prev=dc, dgb=dgdb_all[s], n_extra=E, extra_out=dXH_all[s + 1][:, :E])], B, Hd)
            gemm_bf16(dgdb_all[0], wcat_t, B, E + Hd, C=dXH_all[0])
        # soft decode: attention backward + cell backward of a step in one launch, the steps' d memory in one launch after the loop
        attn_cell = (not steps_fused) and Hd == 512 and Lp <= 64
        ds_all = _new(dev, T, B, Lp) if attn_cell else None
        for s in (() if steps_fused else range(T - 1, -1, -1)):
            dl = dout2[:, s * V:(s + 1) * V]
            dXH = dXH_all[s + 1] if s + 1 < T else None        # written by step s+1's dgrad below
            if s + 1 < T and soft:
                # gradient of the embedding that fed step s+1 (dropout STREAM_G_XT+s was applied to it);
                # straight-through: d p_s += dx @ E^T (rnn.py:84-85).  The scatter into dE waits for the end of the loop.
                xd = drop.at(STREAM_G_XT + s)
                if use_b and E == 128 and V % 4 == 0:
                    # dropout on the operand load, bf16 product, accumulate: one launch (the dropped rows go to dxe_all for the scatter)
                    g_x = dxe_all[s] if xd.p > 0 else None
                    call("cst_dec_dxe", dXH, E + Hd, g_x, E, etok_b, etok_b.stride(0), dl, T * V, B, V, E, *xd.args())
                else:
                    if xd.p > 0:
                        g_x = dxe_all[s]
                        dropout2d(dXH[:, :E], xd, out=g_x)
                    else:
                        g_x = dXH[:, :E]
                    gemm(g_x, True, E_tok, True, dl, B, V, E, accumulate=True)
            if soft:
                dlb = dlb_all[:, s * Vp:(s + 1) * Vp] if use_b else None
                softmax_tau_bwd(out2[:, s * V:(s + 1) * V], dl, inv_tau, dl, dx_b=dlb)
            diffn = df2[:, s * W_:(s + 1) * W_]
            fd = drop.at(STREAM_G_FFN + s)
            if soft:
                r1s = r12[:, s * Hd:(s + 1) * Hd]
                dp1s = dp12[:, s * Hd:(s + 1) * Hd]
                if use_b:
                    dp1b = dp1b_all[:, s * Hd:(s + 1) * Hd]
                    gemm_bf16(dlb, fn2_t, B, Hd, C=dp1s, Cb=dp1b, aux=r1b[:, s * Hd:(s + 1) * Hd], act=4)   # through LeakyReLU
                    if Hd % 64 == 0 and Hd <= 1280 and W_ % 32 == 0:                                         # whole-K product, no split, no reduce launch
                        call("cst_gemm_bf16_skinny", dp1b, T * Hd, fn1_t, fn1_t.stride(0), diffn, T * W_, None, 0, B, W_, Hd, None, 0, *fd.args())
                    else:
                        gemm_bf16(dp1b, fn1_t, B, W_, C=diffn, drop=fd)                                      # through dropout(i_ffn)
                else:
                    dgrad(dl, P["fn_2.weight"], out=dp1s, aux=r1s, act=4)                   # through LeakyReLU
                    dgrad(dp1s, P["fn_1.weight"], out=diffn, drop=fd)                         # through dropout(i_ffn)
            elif fd.p > 0:
                dropout2d(diffn, fd, out=diffn)
            c_prev = c0 if s == 0 else cdec[s - 1]
            last = s == T - 1
            dh2 = None if last else dXH[:, E:]
            dgb = dgdb_all[s] if use_b else None
            if attn_cell:
                call("cst_dec_attn_cell_bwd", diffn, T * W_, memory, patt[s], ds_all[s], B, Lp, Hd, gdec[s], gdec[s].stride(0),
                     c_prev, c_prev.stride(0), cdec[s], cdec[s].stride(0), dh2, _st(dh2), None if last else dc, Hd,
                     dgd[s], dgd[s].stride(0), dc, Hd, dgb, _st(dgb))
            else:
                call("cst_dot_attn_bwd", diffn[:, Hd:], T * W_, if2[:, s * W_:s * W_ + Hd], T * W_, memory, patt[s],
                     diffn[:, :Hd], T * W_, 1, dmem, B, Lp, Hd)
                _cell_bwd(gdec[s], c_prev, cdec[s], diffn[:, :Hd], dh2, None if last else dc, dgd[s], dc, B, Hd, dgb=dgb)
            if use_b:
                gemm_bf16(dgdb_all[s], wcat_t, B, E + Hd, C=dXH_all[s])
            else:
                dgrad(dgd[s], wcat, out=dXH_all[s])
        if attn_cell:
            call("cst_dec_attn_dmem", df2[:, Hd:], T * W_, W_, if2, T * W_, W_, patt, ds_all, dmem, B, T, Lp, Hd)
        # embedding gradients of the tokens fed to steps 1..T-1, all steps in one scatter
        if T > 1:
            S_ = T - 1
            if soft:
                src = dxe_all.view(S_ * B, E) if drop.p > 0 else dXH_all[1:].view(S_ * B, E + Hd)[:, :E]
                call("cst_embed_scatter_add_steps", ids_fb.view(-1), None, 0, None, src, src.stride(0), dE, E, S_, B, E, V,
                     *NO_DROP.args())
            else:
                src = dXH_all[1:].view(S_ * B, E + Hd)[:, :E]
                call("cst_embed_scatter_add_steps", ids_fb.view(-1), x_c, T if x_c is not None else 0,
                     coins if x_c is not None else None, src, E + Hd, dE, E, S_, B, E, V, *drop.at(STREAM_G_XT).args())
        dXH = dXH_all[0]
        # step 0 input was the start embedding (no dropout), h_{-1} the style embedding
        G["start_embedding.weight"] = colsum(dXH[:, :E]).view(1, E)
        dstyle = ops.zeros_like(P["style_embedding.weight"])
        embed_scatter_add(dstyle, dXH[:, E:], ids_a=label)
        G["style_embedding.weight"] = dstyle
        # transfer (rnn.py:68)
        dpre_t = act_bwd(dc, c0, 0.1)
        G["transfer.weight"] = wgrad(dpre_t, c_cat)
        dc_cat = dgrad(dpre_t, P["transfer.weight"])
        # every weight gradient taken straight from bf16 row-major twins (cst_gemm_bf16_tt: decoder fn_2 / fn_1 / gates, encoder W_hh / W_ih of both
        # directions) is recorded here and launched as ONE grouped kernel when the block ends -- their outputs are defined only after it
        with ops.tt_group():
            # batched weight gradients of the decoder
            if dlb_all is not None and (B * T) % 64 == 0 and V % 8 == 0 and Hd % 8 == 0:
                G["fn_2.weight"] = ops.gemm_bf16_tt(dlb_all.view(B * T, Vp), r1b.view(B * T, Hd), V, Hd)   # dlogits^T r1, no transposes
            else:
                G["fn_2.weight"] = wgrad(dout.view(B * T, V), r1.view(B * T, Hd))
            dp1 = dpre1.view(B * T, Hd)
            tt_ok = use_b and (B * T) % 64 == 0 and E % 8 == 0            # weight gradients straight from the bf16 row-major copies
            if tt_ok and dp1b_ok and ifdb is not None:
                G["fn_1.weight"] = ops.gemm_bf16_tt(dp1b_all.view(B * T, Hd), ifdb.view(B * T, W_), Hd, W_)
            else:
                # (fp32 dropout(i_ffn): written by the per-step decode loop only -- every configuration the fast loop accepts takes the branch above)
                G["fn_1.weight"] = wgrad(dp1, iffn_d.view(B * T, W_))
            G["fn_1.bias"] = colsum(dp1)
            dg2 = dgd.view(T * B, 4 * Hd)
            if tt_ok and XHb is not None:
                dwcat = ops.gemm_bf16_tt(dgdb_all.view(T * B, 4 * Hd), XHb.view(T * B, E + Hd), 4 * Hd, E + Hd)
            else:
                dwcat = wgrad(dg2, XH.view(T * B, E + Hd))
            db = colsum(dg2)
            G["decoder.bias_ih_l0"] = db
            G["decoder.bias_hh_l0"] = db.clone()

            # ---- encoder BPTT --------------------------------------------------------------------
            dh0cat = _new(dev, B, 2 * H)
            demb = _new(dev, B * Lp, E)
            dmem2 = dmem.view(B, Lp * 2 * H)
            dge = _new(dev, 2, B, Lp, 4 * H)
            dhr2 = _new(dev, 2, B, H)
            dce2 = _new(dev, 2, B, H)
            dgtb2 = _i16(dev, 2, B, 4 * H) if use_b else None
            encb = []
            for d, suf in enumerate(("", "_reverse")):
                w_ih, w_hh = P["encoder.weight_ih_l0" + suf], P["encoder.weight_hh_l0" + suf]
                order = list(range(Lp)) if d == 0 else list(range(Lp - 1, -1, -1))
                encb.append((w_ih, w_hh, weight_bf16(w_hh)[1] if use_b else None, order, dge[d].view(B, Lp * 4 * H)))   # whh_t [H, 4H]
            seq_bwd = use_b and H == 256 and B % 16 == 0
            enc_b = seq_bwd and embb is not None
            dgeb = _i16(dev, 2, B * Lp, 4 * H) if enc_b else None
            if seq_bwd:
                # both directions, all steps, one launch (mirror of cst_lstm_seq_fwd)
                call("cst_lstm_seq_bwd", _lstm_frag_order_t(encb[0][2], H), _lstm_frag_order_t(encb[1][2], H), genc[0], genc[1],
                     cenc[0], cenc[1], c_cat, 2 * H, dc_cat, dc_cat.stride(0), dmem, dge[0], dge[1],
                     dgeb[0] if enc_b else None, dgeb[1] if enc_b else None, dh0cat, 2 * H, B, Lp, H)
            for n_ in (() if seq_bwd else range(Lp - 1, -1, -1)):
                probs = []
                for d, (w_ih, w_hh, whh_t, order, dg2d) in enumerate(encb):
                    t = order[n_]
                    lastf = n_ == Lp - 1
                    c_new = c_cat[:, d * H:(d + 1) * H] if lastf else cenc[d, t]
                    c_prev = zeros_c if n_ == 0 else cenc[d, order[n_ - 1]]
                    dgt = dg2d[:, t * 4 * H:(t + 1) * 4 * H]
                    dmt = dmem2[:, t * 2 * H + d * H: t * 2 * H + (d + 1) * H]
                    dce = dce2[d]
                    if use_b and not lastf:
                        # dgtb2[d] holds dgates of the step after; the fused reduce overwrites it with this step's
                        probs.append(dict(Ab=dgtb2[d], Bb=whh_t, gates=genc[d, t], c_prev=c_prev, c_new=c_new, dh_extra=dmt, dc_in=dce,
                                          dgates=dgt, dc_prev=dce, dgb=dgtb2[d]))
                        continue
                    _cell_bwd(genc[d, t], c_prev, c_new, dmt, None if lastf else dhr2[d], dc_cat[:, d * H:(d + 1) * H] if lastf else dce,
                              dgt, dce, B, H, dgb=dgtb2[d] if use_b else None)
                    if not use_b:
                        dgrad(dgt, w_hh, out=dh0cat[:, d * H:(d + 1) * H] if n_ == 0 else dhr2[d])
                if probs:
                    _gemm_cell_bwd(probs, B, H)
            for d, (w_ih, w_hh, whh_t, order, dg2d) in enumerate(encb):
                if use_b and not seq_bwd:
                    gemm_bf16(dgtb2[d], whh_t, B, H, C=dh0cat[:, d * H:(d + 1) * H])
                dgf = dge[d].view(B * Lp, 4 * H)
                suf = "" if d == 0 else "_reverse"
                if enc_b:
                    # dgates^T [h_prev], dgates^T emb through transposed LDS reads of the row-major bf16 twins; d emb on the bf16 GEMM
                    G["encoder.weight_hh_l0" + suf] = ops.gemm_bf16_tt(dgeb[d], hprevb[d].view(B * Lp, H), 4 * H, H)
                    G["encoder.weight_ih_l0" + suf] = ops.gemm_bf16_tt(dgeb[d], embb, 4 * H, E)
                    gemm_bf16(dgeb[d], weight_bf16(w_ih)[1], B * Lp, E, C=demb, accumulate=d == 1)
                else:
                    G["encoder.weight_hh_l0" + suf] = wgrad(dgf, hprev[d].view(B * Lp, H))
                    G["encoder.weight_ih_l0" + suf] = wgrad(dgf, emb)
                    dgrad(dgf, w_ih, out=demb, accumulate=d == 1)
                dbe = colsum(dgf)
                G["encoder.bias_ih_l0" + suf] = dbe
                G["encoder.bias_hh_l0" + suf] = dbe.clone()
        G["decoder.weight_ih_l0"] = dwcat[:, :E].contiguous()
        G["decoder.weight_hh_l0"] = dwcat[:, E:].contiguous()
        dstyle_e = ops.zeros_like(P["enc_style_embedding.weight"])
        embed_scatter_add(dstyle_e, dh0cat, ids_a=label_i)
        G["enc_style_embedding.weight"] = dstyle_e
        embed_scatter_add(dE, demb, ids_a=ids_in, drop=in_drop)
        G["token_embedding.weight"] = dE
        dinp = None
        if soft_in and ctx.needs_input_grad[0]:
            dinp = _new(dev, B * Lp, V)
            gemm(demb, True, E_tok, True, dinp, B * Lp, V, E)            # straight-through of rnn.py:61
            dinp = dinp.view(B, Lp, V)
        grads = [G[k] for k in PARAM_KEYS]
        share = ctx.share
        if share is not None:
            prev = share.get("grads")
            here = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
            if prev is not None and share.get("stream") != here:
                prev = None                                   # (CST_FORK=1: the two decodes ran on different streams -- autograd's own sums)
            if prev is None:
                share["grads"], share["stream"] = grads, here # the first backward of the group: autograd gets these tensors ...
            else:
                both = [(a, b) for a, b in zip(prev, grads) if a is not None and b is not None]
                if both:
                    torch._foreach_add_([a for a, _ in both], [b for _, b in both])       # ... and every later one adds into them
                grads = [b if a is None else None for a, b in zip(prev, grads)]
                share["grads"] = None                         # (drop the references: AccumulateGrad may take the tensors as they are)
        return (dinp, None, None, None, None, None, *grads)
