"""DenoiseLSTM generator (reference: src/model/rnn.py)."""
import random

import torch
import torch.nn as nn

from .. import ops
from ..gen_fn import PARAM_KEYS, GeneratorFn
from ._common import SeedState, make_drop

d_embed = 128
d_enc = 256
d_dec = 512
p_drop = 0.1


class _LSTMParams(nn.Module):
    """Holder with nn.LSTM's parameter names (weight_ih_l0, ..., bias_hh_l0[_reverse]) and init."""

    def __init__(self, input_size, hidden_size, bidirectional):
        super().__init__()
        k = 1.0 / hidden_size ** 0.5
        for suf in ([""] + (["_reverse"] if bidirectional else [])):
            for name, shape in (("weight_ih_l0", (4 * hidden_size, input_size)), ("weight_hh_l0", (4 * hidden_size, hidden_size)),
                                ("bias_ih_l0", (4 * hidden_size,)), ("bias_hh_l0", (4 * hidden_size,))):
                self.register_parameter(name + suf, nn.Parameter(torch.empty(*shape).uniform_(-k, k)))


class DenoiseLSTM(nn.Module):
    def __init__(self, n_vocab, n_class, max_len):
        super().__init__()
        self.start_embedding = nn.Embedding(1, d_embed)
        self.token_embedding = nn.Embedding(n_vocab, d_embed)
        self.enc_style_embedding = nn.Embedding(n_class, 2 * d_enc)
        self.style_embedding = nn.Embedding(n_class, d_dec)
        self.encoder = _LSTMParams(d_embed, d_enc, True)
        self.decoder = _LSTMParams(d_embed, d_dec, False)
        self.transfer = nn.Linear(2 * d_enc, d_dec, bias=False)
        self.fn_1 = nn.Linear(2 * d_enc + d_dec, d_dec)
        self.fn_2 = nn.Linear(d_dec, n_vocab, bias=False)
        self.max_len = max_len
        self._seed_state = SeedState(0x6E4E)
        self.last_ids = None

    def _params(self):
        sd = dict(self.named_parameters())
        return [sd[k] for k in PARAM_KEYS]

    def forward(self, inp, label_i, x, label, res_type="none", tau=1.0, coins=None, seed=None):
        """rnn.py:55-98.  `coins` (additive): the per-step scheduled-sampling draws, True = feed the
        argmax back; default draws random.random() < 1/2 per step exactly as rnn.py:91 does (one
        draw per step for the whole batch, none when x is None).  After the call `self.last_ids`
        holds the (T,B) argmax ids that were computed on the way."""
        mode = "softmax" if res_type == "softmax" else "none"     # any other string falls through (rnn.py:90)
        coins_dev = None
        if mode == "none" and x is not None:
            if coins is None:
                coins = [random.random() < 1 / 2 for _ in range(x.size(1))]
            if not isinstance(coins, torch.Tensor):
                coins = torch.tensor([int(bool(c)) for c in coins], dtype=torch.int32)
            if coins.device != inp.device or coins.dtype != torch.int32:
                coins = coins.to(device=inp.device, dtype=torch.int32)
            coins_dev = coins.contiguous()
        cfg = {"mode": mode, "tau": tau, "max_len": self.max_len, "drop": make_drop(self, p_drop, seed)}
        out, ids = GeneratorFn.apply(inp, label_i, x, label, coins_dev, cfg, *self._params())
        self.last_ids = ids
        return out
