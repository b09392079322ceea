"""Plain-torch CPU restatement of the five model modules of the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function takes a ``state_dict``-shaped
mapping ``P`` (the reference's parameter names) and spells the
arithmetic out with elementary tensor ops -- no nn.LSTM / nn.TransformerEncoder / nn.Conv2d --
so that the HIP kernels have an unambiguous statement of what to compute and autograd can
supply reference gradients.  Pinned by tests/golden/*.npz (generated from the imported
reference modules).

Dropout: ``drop`` is ``None`` (evaluation / p forced to 0) or a ``DropSpec(seed, p_scale)``;
masks then come from oracle.rng with the stream ids below -- the same ids the HIP path uses.
"""
import math
from collections import namedtuple

import numpy as np
import torch
import torch.nn.functional as F

from . import rng as _rng

# ---- dropout call-site stream ids (must match consistent__style_transfer_amd/ops.py) -------
STREAM_G_EMB_IN = 1          # rnn.py:59   (B, L', d_embed)
STREAM_G_FFN = 100           # rnn.py:79   + step, (B, 2*d_enc + d_dec)
STREAM_G_XT = 200            # rnn.py:96   + step, (B, d_embed)
STREAM_TFM = 1000            # + 10*layer + {0 attn probs, 1 post-attn, 2 ffn hidden, 3 post-ffn}
STREAM_CLS = 2000            # classifier.py:37 (B, 384)
STREAM_DISC = 3000           # discriminator.py:96 (B*16, 1200)

# reference dropout rates (module constants)
P_DROP_G = 0.1               # rnn.py:14
P_DROP_TFM = 0.1             # nn.TransformerEncoderLayer default
P_DROP_CLS = 0.5             # classifier.py:7
P_DROP_DISC = 0.25           # discriminator.py:13

DropSpec = namedtuple("DropSpec", ["seed"])


def _drop(x, drop, stream, p):
    if drop is None or p <= 0.0:
        return x
    m = torch.from_numpy(_rng.dropout_mask(drop.seed, stream, tuple(x.shape), p))
    return x * m.to(x.dtype)


def leaky(x, slope=0.1):
    return torch.where(x > 0, x, x * slope)


# =============================================================================================
# DenoiseLSTM  (reference: src/model/rnn.py:16-98)
# =============================================================================================
def lstm_cell(x_proj, h, c, w_hh):
    """One LSTM step.  x_proj already holds x W_ih^T + b_ih + b_hh.  Gate order i,f,g,o
    (torch.nn.LSTM convention, which rnn.py:25-33 instantiates)."""
    H = h.shape[-1]
    g = x_proj + h @ w_hh.t()
    i = torch.sigmoid(g[:, 0 * H:1 * H])
    f = torch.sigmoid(g[:, 1 * H:2 * H])
    gg = torch.tanh(g[:, 2 * H:3 * H])
    o = torch.sigmoid(g[:, 3 * H:4 * H])
    c2 = f * c + i * gg
    h2 = o * torch.tanh(c2)
    return h2, c2


def hard_sample_value(p):
    """rnn.py:52-53: one_hot(argmax) - p.detach() + p  (value ~ one-hot, gradient = identity)."""
    oh = F.one_hot(p.argmax(-1), p.size(-1)).to(p.dtype)
    return oh - p.detach() + p


def bilstm_encode(P, emb, h0):
    """rnn.py:62.  emb (B,L,E); h0 (2,B,H).  Runs over every position, PAD included."""
    B, L, _ = emb.shape
    outs = []
    c_end = []
    for d, suf in enumerate(("", "_reverse")):
        w_ih = P["encoder.weight_ih_l0" + suf]
        w_hh = P["encoder.weight_hh_l0" + suf]
        b = P["encoder.bias_ih_l0" + suf] + P["encoder.bias_hh_l0" + suf]
        xp = emb @ w_ih.t() + b                       # (B,L,4H)
        h = h0[d]
        c = torch.zeros_like(h)
        hs = [None] * L
        order = range(L) if d == 0 else range(L - 1, -1, -1)
        for t in order:
            h, c = lstm_cell(xp[:, t], h, c, w_hh)
            hs[t] = h
        outs.append(torch.stack(hs, dim=1))
        c_end.append(c)
    memory = torch.cat(outs, dim=-1)                  # (B,L,2H)
    return memory, torch.cat(c_end, dim=-1)           # (B,2H): [c_fwd, c_bwd] per row (rnn.py:68)


def dot_attn(q, mem):
    """rnn.py:46-50 single query, unmasked.  q (B,D), mem (B,L,D) -> (B,D)."""
    a = torch.einsum("bd,bld->bl", q, mem) / math.sqrt(mem.size(-1))
    w = torch.softmax(a, dim=-1)
    return torch.einsum("bl,bld->bd", w, mem)


def denoise_lstm(P, inp, label_i, x, label, res_type="none", tau=1.0, max_len=None,
                 coins=None, drop=None):
    """DenoiseLSTM.forward (rnn.py:55-98).

    coins: sequence of booleans, one per decode step, replacing ``random.random() < 1/2``
    (rnn.py:91); True = feed back argmax.  Ignored when x is None (always argmax) and in
    softmax mode.  Returns (B,T,V): raw logits ("none") or probabilities ("softmax").
    """
    E = P["token_embedding.weight"]
    d_enc = P["encoder.weight_hh_l0"].shape[1]
    B = inp.size(0)
    h0 = P["enc_style_embedding.weight"][label_i].reshape(B, 2, d_enc).transpose(0, 1)
    if inp.dim() == 2:
        emb = _drop(E[inp], drop, STREAM_G_EMB_IN, P_DROP_G)
    else:
        emb = hard_sample_value(inp) @ E
    memory, c_cat = bilstm_encode(P, emb, h0)

    T = max_len if x is None else x.size(1)
    x_t = P["start_embedding.weight"][0].unsqueeze(0).expand(B, -1)
    c_t = leaky(c_cat @ P["transfer.weight"].t())
    h_t = P["style_embedding.weight"][label]
    w_ih, w_hh = P["decoder.weight_ih_l0"], P["decoder.weight_hh_l0"]
    b_dec = P["decoder.bias_ih_l0"] + P["decoder.bias_hh_l0"]
    outs = []
    for step in range(T):
        h_t, c_t = lstm_cell(x_t @ w_ih.t() + b_dec, h_t, c_t, w_hh)
        a_t = dot_attn(h_t, memory)
        i_ffn = _drop(torch.cat([h_t, a_t], dim=-1), drop, STREAM_G_FFN + step, P_DROP_G)
        o_f1 = i_ffn @ P["fn_1.weight"].t() + P["fn_1.bias"]
        logits_t = leaky(o_f1) @ P["fn_2.weight"].t()
        if res_type == "softmax":
            logits_t = torch.softmax(logits_t / tau, dim=-1)
            x_t = hard_sample_value(logits_t) @ E
        else:
            feed_argmax = True if x is None else bool(coins[step])
            tok = logits_t.argmax(-1) if feed_argmax else x[:, step]
            x_t = E[tok]
        x_t = _drop(x_t, drop, STREAM_G_XT + step, P_DROP_G)
        outs.append(logits_t)
    return torch.stack(outs, dim=1)


# =============================================================================================
# nn.TransformerEncoderLayer as configured at mlm.py:20-22 / match.py:18-20
# (post-LN, ReLU, eps 1e-5, dropout 0.1)
# =============================================================================================
def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def encoder_layer(P, pre, x, n_head, drop=None, layer=0):
    """x (B,S,d) batch-first (the reference transposes to seq-first; the math is per batch row)."""
    B, S, d = x.shape
    hd = d // n_head
    qkv = x @ P[pre + "self_attn.in_proj_weight"].t() + P[pre + "self_attn.in_proj_bias"]
    q, k, v = qkv.split(d, dim=-1)

    def heads(t):
        return t.reshape(B, S, n_head, hd).permute(0, 2, 1, 3)      # (B,H,S,hd)

    q, k, v = heads(q), heads(k), heads(v)
    att = torch.softmax((q / math.sqrt(hd)) @ k.transpose(-1, -2), dim=-1)
    att = _drop(att, drop, STREAM_TFM + 10 * layer + 0, P_DROP_TFM)
    o = (att @ v).permute(0, 2, 1, 3).reshape(B, S, d)
    o = o @ P[pre + "self_attn.out_proj.weight"].t() + P[pre + "self_attn.out_proj.bias"]
    o = _drop(o, drop, STREAM_TFM + 10 * layer + 1, P_DROP_TFM)
    x = layer_norm(x + o, P[pre + "norm1.weight"], P[pre + "norm1.bias"])
    hdn = torch.relu(x @ P[pre + "linear1.weight"].t() + P[pre + "linear1.bias"])
    hdn = _drop(hdn, drop, STREAM_TFM + 10 * layer + 2, P_DROP_TFM)
    f = hdn @ P[pre + "linear2.weight"].t() + P[pre + "linear2.bias"]
    f = _drop(f, drop, STREAM_TFM + 10 * layer + 3, P_DROP_TFM)
    return layer_norm(x + f, P[pre + "norm2.weight"], P[pre + "norm2.bias"])


def encoder_stack(P, pre, x, n_head, drop=None):
    n_layer = 0
    while (pre + f"layers.{n_layer}.norm1.weight") in P:
        n_layer += 1
    for i in range(n_layer):
        x = encoder_layer(P, pre + f"layers.{i}.", x, n_head, drop, i)
    return x


def _tok_embed(table, t):
    """index path or soft path (mlm.py:28-33, match.py:25-30, classifier.py:24-27)."""
    if t.dim() == 2:
        return table[t]
    if t.dim() == 3:
        return t @ table
    raise Exception


# =============================================================================================
# MLM  (src/model/mlm.py:27-46)
# =============================================================================================
def mlm(P, inputs, n_head=8, drop=None):
    L = inputs.size(1)
    x = _tok_embed(P["token_embedding.weight"], inputs) + P["posit_embedding.weight"][:L].unsqueeze(0)
    x = encoder_stack(P, "lm.", x, n_head, drop)
    return x @ P["fwd.weight"].t() + P["fwd.bias"]


# =============================================================================================
# Matcher  (src/model/match.py:24-42)
# =============================================================================================
def matcher(P, x1, x2, n_head=8, drop=None):
    def emb(t, seg):
        L = t.size(1)
        return (_tok_embed(P["token_embedding.weight"], t)
                + P["posit_embedding.weight"][:L].unsqueeze(0)
                + P["segment_embedding.weight"][seg].reshape(1, 1, -1))

    x = torch.cat([emb(x1, 0), emb(x2, 1)], dim=1)
    x = encoder_stack(P, "matcher.", x, n_head, drop)
    pooled = x.max(dim=1).values
    return (pooled @ P["hidden2logits.weight"].t() + P["hidden2logits.bias"]).squeeze(1)


# =============================================================================================
# TextCNN  (src/model/classifier.py:23-41)
# =============================================================================================
def textcnn(P, x, drop=None):
    e = _tok_embed(P["embedding.weight"], x)                       # (B,L,E)
    B, L, E = e.shape
    feats = []
    i = 0
    while f"convs.{i}.weight" in P:
        w = P[f"convs.{i}.weight"]                                 # (F,1,k,E)
        k = w.shape[2]
        ep = F.pad(e, (0, 0, k - 1, k - 1))                        # zero rows both ends
        win = ep.unfold(1, k, 1).permute(0, 1, 3, 2).reshape(B, L + k - 1, k * E)
        y = torch.relu(win @ w.reshape(w.shape[0], -1).t() + P[f"convs.{i}.bias"])
        feats.append(y.max(dim=1).values)                          # global max over time
        i += 1
    h = _drop(torch.cat(feats, dim=1), drop, STREAM_CLS, P_DROP_CLS)
    return h @ P["out.weight"].t() + P["out.bias"]


# =============================================================================================
# RelGAN_D  (src/model/discriminator.py:33-57)
# =============================================================================================
def relgan_d(P, inp, drop=None, num_rep=None):
    """inp (B,L,V) dense probabilities / one-hot, or (B,L) ids == one-hot fast path."""
    W = P["embeddings.weight"]                                     # (E,V)
    e = W.t()[inp] if inp.dim() == 2 else inp @ W.t()              # (B,L,E)
    B, L, E = e.shape
    es = P["convs.0.weight"].shape[3]                              # emb_dim_single (discriminator.py:18)
    num_rep = E // es
    er = e.reshape(B, L, num_rep, es)
    pools = []
    i = 0
    while f"convs.{i}.weight" in P:
        w = P[f"convs.{i}.weight"]                                 # (N,1,f,es)
        f = w.shape[2]
        win = er.unfold(1, f, 1)                                   # (B,L-f+1,R,es,f)
        win = win.permute(0, 2, 1, 4, 3).reshape(B, num_rep, L - f + 1, f * es)
        y = torch.relu(win @ w.reshape(w.shape[0], -1).t() + P[f"convs.{i}.bias"])
        pools.append(y.max(dim=2).values)                          # (B,R,N)
        i += 1
    pred = torch.cat(pools, dim=-1).reshape(B * num_rep, -1)       # (B*R, 1200)
    hw = pred @ P["highway.weight"].t() + P["highway.bias"]
    sg = torch.sigmoid(hw)
    pred = sg * torch.relu(hw) + (1.0 - sg) * pred
    pred = _drop(pred, drop, STREAM_DISC, P_DROP_DISC)
    pred = pred @ P["feature2out.weight"].t() + P["feature2out.bias"]
    return (pred @ P["out2logits.weight"].t() + P["out2logits.bias"]).squeeze(1)
