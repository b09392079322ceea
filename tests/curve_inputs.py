"""Seeded batches of the loss-curve fixtures (same generators as tests/golden/make_golden.py::curve_goldens)."""
import numpy as np
import torch

from oracle.configs import noised_len, seg2_len
from oracle.detinit import det_tokens


def opt_batch(c, it):
    x = det_tokens(c["B"], c["L"], c["V"], 100 + it)
    labels = torch.tensor([(i + it) % 2 for i in range(c["B"])], dtype=torch.long)
    return x, labels


def warm_batch(c, it):
    x, labels = opt_batch(c, it)
    return det_tokens(c["B"], noised_len(c), c["V"], 300 + it), x, labels


def pre_batch(c, it):
    x, labels = opt_batch(c, it)
    B, L, V = c["B"], c["L"], c["V"]
    c_label = torch.from_numpy(np.random.RandomState(700 + it).uniform(0, 1.5, size=(B,)).astype(np.float32))
    return (x, det_tokens(B, L, V, 400 + it), det_tokens(B, seg2_len(c), V, 500 + it), det_tokens(B, L, V, 600 + it), labels, c_label)


HP = dict(w_s=0.1, w_c=0.5, w_adv=1.0, w_bt=1.0, tau=0.1, gap=0.0)

CURVE_LR = {"tiny": 1e-3, "ref": 1e-5, "long": 1e-3, "b16": {"optimize": 1e-4, "warmup": 1e-4, "pretrain": 1e-5}}      # as tests/golden/make_golden.py


def curve_lr(name, stage):
    """Learning rate of the `stage` ("optimize" | "warmup" | "pretrain") curve fixture of configuration `name`."""
    v = CURVE_LR[name]
    return v[stage] if isinstance(v, dict) else v
