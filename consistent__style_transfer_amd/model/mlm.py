"""MLM denoiser / naturalness checker (reference: src/model/mlm.py)."""
import torch
import torch.nn as nn

from .. import ops
from ._common import EncoderStack, SeedState, make_drop

d_model = 512
n_head = 8
n_layer = 6
p_drop = 0.1          # nn.TransformerEncoderLayer default (mlm.py:20-22 passes none)


class MLM(nn.Module):
    def __init__(self, n_vocab, n_class):
        super().__init__()
        self.token_embedding = nn.Embedding(n_vocab, d_model)
        self.posit_embedding = nn.Embedding(100, d_model)        # mlm.py:14: max sequence length 100
        nn.init.xavier_uniform_(self.posit_embedding.weight)     # mlm.py:17
        self.lm = EncoderStack(d_model, n_head, n_layer)
        self.fwd = nn.Linear(d_model, n_vocab)
        self._seed_state = SeedState(0x31A0)

    def embedding(self, tensor):
        """mlm.py:27-38: (B,L) ids or (B,L,V) probabilities -> (B,L,d)."""
        return ops.TpsEmbedFn.apply(tensor, None, self.token_embedding.weight, self.posit_embedding.weight, None)

    def forward(self, inputs, seed=None):
        x = self.embedding(inputs)
        B, L, d = x.shape
        drop = make_drop(self, p_drop, seed)
        x = self.lm.run(x.view(B * L, d), B, L, drop)
        logits = ops.vocab_proj(x, self.fwd.weight, self.fwd.bias)
        return logits.view(B, L, -1)
