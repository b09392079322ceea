"""Content matcher (reference: src/model/match.py)."""
import torch
import torch.nn as nn

from .. import ops
from ._common import EncoderStack, SeedState, make_drop

d_model = 512
n_head = 8
n_layer = 6
p_drop = 0.1


class Matcher(nn.Module):
    def __init__(self, n_vocab):
        super().__init__()
        self.token_embedding = nn.Embedding(n_vocab, d_model)
        self.segment_embedding = nn.Embedding(2, d_model)
        self.posit_embedding = nn.Embedding(100, d_model)
        self.matcher = EncoderStack(d_model, n_head, n_layer)
        self.hidden2logits = nn.Linear(d_model, 1)
        self._seed_state = SeedState(0x3A7C)

    def embedding(self, tensor, seg_id):
        """match.py:24-34 for one segment (positions restart at 0 per segment)."""
        seg = self.segment_embedding.weight
        if seg_id == 1:
            seg = seg.flip(0)
        return ops.TpsEmbedFn.apply(tensor, None, self.token_embedding.weight, self.posit_embedding.weight, seg)

    def forward(self, x1, x2, seed=None):
        pre1 = ops.shared_lookup(x1, self.token_embedding.weight) if x1.dim() == 3 else None
        x = ops.TpsEmbedFn.apply(x1, x2, self.token_embedding.weight, self.posit_embedding.weight,
                                 self.segment_embedding.weight, pre1)
        B, S, d = x.shape
        drop = make_drop(self, p_drop, seed)
        x = self.matcher.run(x.view(B * S, d), B, S, drop)
        pooled = ops.SeqMaxFn.apply(x.view(B, S, d))
        return ops.linear(pooled, self.hidden2logits.weight, self.hidden2logits.bias).view(B)
