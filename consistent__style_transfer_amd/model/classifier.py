"""TextCNN style classifier (reference: src/model/classifier.py)."""
import torch
import torch.nn as nn

from .. import ops
from ._common import SeedState, make_drop

d_embed = 128
p_drop = 0.5
kernels = [3, 4, 5]
kernel_number = [128, 128, 128]


class TextCNN(nn.Module):
    def __init__(self, n_vocab, n_class):
        super().__init__()
        self.embedding = nn.Embedding(n_vocab, d_embed)
        self.convs = nn.ModuleList(
            [nn.Conv2d(1, number, (size, d_embed), padding=(size - 1, 0)) for (size, number) in zip(kernels, kernel_number)]
        )                                                      # parameter holders only (keys convs.{i}.weight/.bias)
        self.out = nn.Linear(sum(kernel_number), n_class)
        self._seed_state = SeedState(0xC1A5)

    def forward(self, x, seed=None):
        if len(x.shape) == 2:
            B, L = x.shape
            e = ops.EmbedFn.apply(x, self.embedding.weight, False)
        elif len(x.shape) == 3:
            B, L, V = x.shape
            e = ops.soft_embed(x.reshape(B * L, V), self.embedding.weight)
        else:
            raise Exception
        wb = []
        for c in self.convs:
            wb += [c.weight, c.bias]
        feats = ops.ConvBankFn.apply(e.view(B, L, -1), 0, 1, *wb)
        drop = make_drop(self, p_drop, seed)
        return ops.linear(ops.dropout(feats, drop.at(ops.STREAM_CLS)), self.out.weight, self.out.bias)
