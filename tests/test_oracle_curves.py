"""CPU: the oracle training loops (oracle/train.py) reproduce the multi-step loss curves recorded
from the reference modules -- this pins optimiser / clipping / D-every-4th-batch semantics."""
import numpy as np
import pytest
import torch

from curve_inputs import HP, curve_lr, opt_batch, pre_batch, warm_batch
from helpers import CONFIGS, det_params, load_golden
from oracle import train as T

torch.set_num_threads(4)


@pytest.mark.parametrize("name", ["tiny", "ref", "b16", "long"])
def test_optimize_curve(name):
    c, G = CONFIGS[name], load_golden("curves", name)
    P = {k: det_params(name, k) for k in ("G", "cls", "mat", "dn", "disc")}
    tr = T.OracleOptimize(P["G"], P["cls"], P["mat"], P["dn"], P["disc"], HP, c["n_head"], c["max_len"], lr=curve_lr(name, "optimize"))
    steps = G["optimize.curve"].shape[0] if name in ("tiny", "long") else 3
    rows = [tr.step(opt_batch(c, it), it, G["optimize.coins"][it]) for it in range(steps)]
    np.testing.assert_allclose(np.array(rows), G["optimize.curve"][:steps], rtol=2e-3, atol=1e-3)


@pytest.mark.parametrize("name", ["tiny", "long"])
def test_warmup_and_pretrain_curves(name):
    c, G = CONFIGS[name], load_golden("curves", name)
    tw = T.OracleWarmup(det_params(name, "G"), lr=1e-3)
    rows = [tw.step(warm_batch(c, it), G["warmup.coins"][it]) for it in range(G["warmup.curve"].shape[0])]
    np.testing.assert_allclose(rows, G["warmup.curve"], rtol=1e-3, atol=1e-3)
    tp = T.OraclePretrain(det_params(name, "cls"), det_params(name, "mat"), det_params(name, "dn"), c["n_head"], lr=1e-3)
    rows = [tp.step(pre_batch(c, it)) for it in range(G["pretrain.curve"].shape[0])]
    np.testing.assert_allclose(np.array(rows), G["pretrain.curve"], rtol=2e-3, atol=1e-3)
