"""RelGAN discriminator (reference: src/model/discriminator.py)."""
import math

import torch
import torch.nn as nn

from .. import ops
from ._common import SeedState, make_drop

embed_dim = 128
num_rep = 16
dis_filter_sizes = [2, 3, 4, 5]
dis_num_filters = [300, 300, 300, 300]


class RelGAN_D(nn.Module):
    def __init__(self, vocab_size, dropout=0.25):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_rep = num_rep
        self.feature_dim = sum(dis_num_filters)
        self.emb_dim_single = int(embed_dim / num_rep)
        self.embeddings = nn.Linear(vocab_size, embed_dim, bias=False)
        self.convs = nn.ModuleList([
            nn.Conv2d(1, n, (f, self.emb_dim_single), stride=(1, self.emb_dim_single))
            for (n, f) in zip(dis_num_filters, dis_filter_sizes)
        ])                                                      # parameter holders only
        self.highway = nn.Linear(self.feature_dim, self.feature_dim)
        self.feature2out = nn.Linear(self.feature_dim, 100)
        self.out2logits = nn.Linear(100, 1)
        self.p_drop = dropout
        self._seed_state = SeedState(0xD15C)
        self.init_params()

    def forward(self, inp, seed=None):
        """inp: (B,L,V) probabilities / one-hot rows (discriminator.py:33-38), or -- additive fast
        path -- (B,L) int64 ids, equal to feeding F.one_hot(ids).float() (main_optimize.py:117)
        without materialising the dense tensor.  Returns (B*num_rep,) logits."""
        W = self.embeddings.weight                              # (E, V)
        if inp.dim() == 3:
            B, L, V = inp.shape
            e = ops.soft_embed(inp.reshape(B * L, V), W, True)
        elif inp.dim() == 2:
            B, L = inp.shape
            e = ops.EmbedFn.apply(inp, W, True)
        else:
            raise Exception
        wb = []
        for c in self.convs:
            wb += [c.weight, c.bias]
        pred = ops.ConvBankFn.apply(e.view(B, L, -1), 1, self.num_rep, *wb)          # (B*R, feature_dim)
        hw = ops.linear(pred, self.highway.weight, self.highway.bias)
        pred = ops.HighwayFn.apply(hw, pred)
        drop = make_drop(self, self.p_drop, seed)
        pred = ops.linear(ops.dropout(pred, drop.at(ops.STREAM_DISC)), self.feature2out.weight, self.feature2out.bias)
        return ops.linear(pred, self.out2logits.weight, self.out2logits.bias).view(-1)

    def init_params(self):
        """discriminator.py:53-57: every tensor (biases too) ~ N(0, 1/sqrt(shape[0]))."""
        for param in self.parameters():
            if param.requires_grad and len(param.shape) > 0:
                torch.nn.init.normal_(param, std=1 / math.sqrt(param.shape[0]))
