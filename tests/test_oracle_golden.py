"""CPU: the oracle (oracle/) reproduces the golden vectors generated from the imported
reference modules (tests/golden/make_golden.py).  This is the pin of the oracle."""
import numpy as np
import pytest
import torch

from oracle import modules as M
from oracle import steps as S
from helpers import CONFIGS, check_grads, det_params, load_golden, lossw, soft_input

RTOL, ATOL = 2e-4, 2e-5
torch.set_num_threads(4)


def _t(a):
    return torch.from_numpy(a)


@pytest.mark.parametrize("name", ["tiny", "ref", "b16", "long"])
def test_textcnn(name):
    c, G = CONFIGS[name], load_golden("modules", name)
    P = det_params(name, "cls", True)
    y = M.textcnn(P, _t(G["x"]))
    np.testing.assert_allclose(y.detach().numpy(), G["cls.ids.out"], rtol=RTOL, atol=ATOL)
    gs = torch.autograd.grad(lossw("cls.ids", y), list(P.values()), allow_unused=True)
    check_grads(G, "cls.ids", {k: g for k, g in zip(P, gs) if g is not None}, 1e-3, 1e-4)
    sp = soft_input(c["B"], c["L"], c["V"], 11)
    y = M.textcnn(P, sp)
    np.testing.assert_allclose(y.detach().numpy(), G["cls.soft.out"], rtol=RTOL, atol=ATOL)
    gs = torch.autograd.grad(lossw("cls.soft", y), list(P.values()) + [sp])
    check_grads(G, "cls.soft", dict(zip(P, gs[:-1])), 1e-3, 1e-4, gs[-1])


@pytest.mark.parametrize("name", ["tiny", "ref", "b16", "long"])
def test_mlm(name):
    c, G = CONFIGS[name], load_golden("modules", name)
    P = det_params(name, "dn", True)
    y = M.mlm(P, _t(G["x"]), c["n_head"])
    np.testing.assert_allclose(y.detach().numpy(), G["mlm.ids.out"], rtol=1e-3, atol=1e-4)
    gs = torch.autograd.grad(lossw("mlm.ids", y), list(P.values()))
    check_grads(G, "mlm.ids", dict(zip(P, gs)), 2e-3, 2e-3)
    sp = soft_input(c["B"], c["L"], c["V"], 12)
    y = M.mlm(P, sp, c["n_head"])
    np.testing.assert_allclose(y.detach().numpy(), G["mlm.soft.out"], rtol=1e-3, atol=1e-4)
    gs = torch.autograd.grad(lossw("mlm.soft", y), list(P.values()) + [sp])
    check_grads(G, "mlm.soft", dict(zip(P, gs[:-1])), 2e-3, 2e-3, gs[-1])


@pytest.mark.parametrize("name", ["tiny", "ref", "b16", "long"])
def test_matcher(name):
    c, G = CONFIGS[name], load_golden("modules", name)
    P = det_params(name, "mat", True)
    y = M.matcher(P, _t(G["x"]), _t(G["x2"]), c["n_head"])
    np.testing.assert_allclose(y.detach().numpy(), G["mat.ids.out"], rtol=1e-3, atol=1e-4)
    gs = torch.autograd.grad(lossw("mat.ids", y), list(P.values()))
    # b16 (d = 768): one of 1559 sampled elements sits 2.3e-3 off -- a near-tie in the max over the sequence (match.py:41)
    # resolved differently by the two summation orders moves that position's gradient
    ga = 4e-3 if name == "b16" else 2e-3
    check_grads(G, "mat.ids", dict(zip(P, gs)), 2e-3, ga)
    sp = soft_input(c["B"], c["L"], c["V"], 13)
    y = M.matcher(P, sp, _t(G["x"]), c["n_head"])
    np.testing.assert_allclose(y.detach().numpy(), G["mat.soft.out"], rtol=1e-3, atol=1e-4)
    gs = torch.autograd.grad(lossw("mat.soft", y), list(P.values()) + [sp])
    check_grads(G, "mat.soft", dict(zip(P, gs[:-1])), 2e-3, ga, gs[-1])


@pytest.mark.parametrize("name", ["tiny", "ref", "b16", "long"])
def test_relgan_d(name):
    c, G = CONFIGS[name], load_golden("modules", name)
    P = det_params(name, "disc", True)
    sp = soft_input(c["B"], c["L"], c["V"], 14)
    y = M.relgan_d(P, sp, num_rep=c["d_rep"])
    np.testing.assert_allclose(y.detach().numpy(), G["disc.soft.out"], rtol=RTOL, atol=ATOL)
    gs = torch.autograd.grad(lossw("disc.soft", y), list(P.values()) + [sp])
    check_grads(G, "disc.soft", dict(zip(P, gs[:-1])), 1e-3, 1e-4, gs[-1])
    # ids fast path == dense one-hot input (main_optimize.py:117)
    y = M.relgan_d(P, _t(G["x"]), num_rep=c["d_rep"])
    np.testing.assert_allclose(y.detach().numpy(), G["disc.onehot.out"], rtol=RTOL, atol=ATOL)
    gs = torch.autograd.grad(lossw("disc.onehot", y), list(P.values()))
    check_grads(G, "disc.onehot", dict(zip(P, gs)), 1e-3, 1e-4)


@pytest.mark.parametrize("name", ["tiny", "ref", "b16", "long"])
def test_generator(name):
    c, G = CONFIGS[name], load_golden("modules", name)
    P = det_params(name, "G", True)
    x, nx, labels = _t(G["x"]), _t(G["nx"]), _t(G["labels"])
    # teacher forcing with recorded coins
    y = M.denoise_lstm(P, nx, labels, x, labels, coins=G["gen.tf.coins"])
    np.testing.assert_allclose(y.detach().numpy(), G["gen.tf.out"], rtol=1e-3, atol=1e-4)
    gs = torch.autograd.grad(lossw("gen.tf", y), list(P.values()), allow_unused=True)
    check_grads(G, "gen.tf", {k: g for k, g in zip(P, gs) if g is not None}, 2e-3, 1e-3)
    # pure teacher forcing (every coin False)
    y = M.denoise_lstm(P, nx, labels, x, labels, coins=[False] * x.shape[1])
    np.testing.assert_allclose(y.detach().numpy(), G["gen.tf0.out"], rtol=1e-3, atol=1e-4)
    gs = torch.autograd.grad(lossw("gen.tf0", y), list(P.values()), allow_unused=True)
    check_grads(G, "gen.tf0", {k: g for k, g in zip(P, gs) if g is not None}, 2e-3, 1e-3)
    # softmax / straight-through
    for tag, tau in (("gen.soft", 0.1), ("gen.soft1", 1.0)):
        y = M.denoise_lstm(P, x, labels, None, 1 - labels, "softmax", tau, max_len=c["max_len"])
        np.testing.assert_allclose(y.detach().numpy(), G[tag + ".out"], rtol=2e-3, atol=1e-5)
        gs = torch.autograd.grad(lossw(tag, y), list(P.values()), allow_unused=True)
        check_grads(G, tag, {k: g for k, g in zip(P, gs) if g is not None}, 5e-3, 2e-3)
    # greedy ids: bit exact
    with torch.no_grad():
        y = M.denoise_lstm(P, x, labels, None, 1 - labels, max_len=c["max_len"])
    assert np.array_equal(y.argmax(-1).numpy(), G["gen.greedy.ids"])
    np.testing.assert_allclose(y.numpy(), G["gen.greedy.out"], rtol=1e-3, atol=1e-4)
    # 3-D encoder input
    sp = soft_input(c["B"], c["L"], c["V"], 15)
    y = M.denoise_lstm(P, sp, labels, x, labels, coins=G["gen.soft_in.coins"])
    np.testing.assert_allclose(y.detach().numpy(), G["gen.soft_in.out"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("name", ["tiny", "ref", "b16", "long"])
def test_stage_steps(name):
    c, G = CONFIGS[name], load_golden("steps", name)
    Pg, Pc, Pm, Pd, Pdisc = (det_params(name, k, True) for k in ("G", "cls", "mat", "dn", "disc"))
    x, nx1, nx2, nx3 = (_t(G[k]) for k in ("x", "nx1", "nx2", "nx3"))
    labels, c_label = _t(G["labels"]), _t(G["c_label"])
    s, cl, dn = S.pretrain_losses(Pc, Pm, Pd, (x, nx1, nx2, nx3, labels, c_label), c["n_head"])
    np.testing.assert_allclose([s.item(), cl.item(), dn.item()], G["pretrain.losses"], rtol=2e-4)
    w = S.warmup_loss(Pg, (nx2, x, labels), G["warmup.coins"])
    np.testing.assert_allclose(w.item(), G["warmup.loss"][0], rtol=2e-4)
    gs = torch.autograd.grad(w, list(Pg.values()), allow_unused=True)
    gn = torch.sqrt(sum((g ** 2).sum() for g in gs if g is not None)).item()
    np.testing.assert_allclose(gn, G["warmup.gnorm"][0], rtol=1e-3)
    hp = dict(w_s=0.1, w_c=0.5, w_adv=1.0, w_bt=1.0, tau=0.1, gap=0.0)
    r = S.optimize_g_losses(Pg, Pc, Pm, Pdisc, (x, labels), G["optimize.coins"], hp, c["n_head"], c["max_len"])
    got = [r["loss"].item(), r["G"].item(), r["STI"].item(), r["CP"].item(), r["BK"].item()]
    np.testing.assert_allclose(got, G["optimize.g.losses"], rtol=5e-4, atol=1e-5)
    assert np.array_equal(r["sample_p"].argmax(-1).numpy(), G["optimize.g.sample_ids"])
    gs = torch.autograd.grad(r["loss"], list(Pg.values()), allow_unused=True)
    gn = torch.sqrt(sum((g ** 2).sum() for g in gs if g is not None)).item()
    np.testing.assert_allclose(gn, G["optimize.g.gnorm"][0], rtol=5e-3)
    d = S.optimize_d_losses(Pg, Pdisc, (x, labels), hp, c["max_len"])
    np.testing.assert_allclose(d["D"].item(), G["optimize.d.losses"][0], rtol=2e-4)
    gs = torch.autograd.grad(d["loss"], list(Pdisc.values()))
    gn = torch.sqrt(sum((g ** 2).sum() for g in gs)).item()
    np.testing.assert_allclose(gn, G["optimize.d.gnorm"][0], rtol=1e-3)
    with torch.no_grad():
        v = S.optimize_val_loss(Pg, Pc, Pm, Pd, (x, labels), hp, c["n_head"], c["max_len"])
    np.testing.assert_allclose(v.item(), G["optimize.val"][0], rtol=5e-4)
