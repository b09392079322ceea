"""Shared plumbing of the model modules: parameter holders with the reference's state_dict key
layout, and the per-module dropout seed state."""
import math

import torch
import torch.nn as nn

from ..ops import Drop, NO_DROP


class SeedState:
    """Dropout seeds: `seed` (host int) + optional device word that a captured graph bumps on
    replay.  Modules draw one seed per forward call unless the caller passes `seed=`."""

    def __init__(self, base=0x5EED):
        self.base = base
        self.calls = 0
        self.seed_dev = None

    def next(self):
        self.calls += 1
        return (self.base + 0x9E37 * self.calls) & 0x7FFFFFFF


def make_drop(module, p, seed):
    if not module.training or p <= 0:
        return NO_DROP
    st = module._seed_state
    return Drop(p, st.next() if seed is None else seed, 0, st.seed_dev)


class _SelfAttnParams(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class EncoderLayerParams(nn.Module):
    """Parameter set of nn.TransformerEncoderLayer(d_model, nhead) with PyTorch's defaults
    (dim_feedforward 2048, post-LN) -- keys self_attn.in_proj_weight/.in_proj_bias,
    self_attn.out_proj.{weight,bias}, linear1.*, linear2.*, norm1.*, norm2.* (SURVEY 8a row 8)."""

    def __init__(self, d, dim_ff=2048):
        super().__init__()
        self.self_attn = _SelfAttnParams(d)
        self.linear1 = nn.Linear(d, dim_ff)
        self.linear2 = nn.Linear(dim_ff, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)

    def flat(self):
        a = self.self_attn
        return (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias)


class EncoderStack(nn.Module):
    """nn.TransformerEncoder(layer, num_layers): `layers.{i}.*`; all layers start as copies of one
    initialisation, as copy.deepcopy does in torch."""

    def __init__(self, d, n_head, n_layer, dim_ff=2048):
        super().__init__()
        first = EncoderLayerParams(d, dim_ff)
        self.layers = nn.ModuleList([first] + [EncoderLayerParams(d, dim_ff) for _ in range(n_layer - 1)])
        for l in self.layers[1:]:
            l.load_state_dict(first.state_dict())
        self.n_head = n_head
        self.taps = None           # set to a list to record [input of layer 0, output of layer 0, ..., output of the last layer]

    def run(self, x2d, B, S, drop):
        """`taps` (stages.bucketed_backward): the tensors at the layer seams, recorded so that the backward pass can be driven
        one encoder layer at a time from Python -- each layer's gradients are complete, and their all-reduce can start, while
        the layers below are still being back-propagated."""
        from ..ops import EncoderLayerBf16Fn, EncoderLayerFn, get_precision
        fn = EncoderLayerFn if get_precision() == "f32" else EncoderLayerBf16Fn
        if self.taps is not None:
            self.taps.clear()
            self.taps.append(x2d)
        for i, l in enumerate(self.layers):
            x2d = fn.apply(x2d, *l.flat(), B, S, self.n_head, drop, i)
            if self.taps is not None:
                self.taps.append(x2d)
        return x2d
