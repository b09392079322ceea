"""Loss composition of the three stages, restated on top of oracle.modules.

TEST INFRASTRUCTURE (see oracle/__init__.py).
  pretrain : src/main_pretrain.py:66-77   CE(cls) + MSE(matcher, wmd label) + token CE(mlm)
  warmup   : src/main_warmup.py:45-58     token CE(generator(nx -> x))
  optimize : src/main_optimize.py:93-124  generator step (idx 0) and discriminator step (idx 1)
All reductions are means; token CE counts PAD targets (nn.CrossEntropyLoss default
ignore_index=-100, main_pretrain.py:41).
"""
import torch
import torch.nn.functional as F

from . import modules as M


def token_ce(logits, target):
    return F.cross_entropy(logits.reshape(-1, logits.size(-1)), target.reshape(-1))


def bce_logits(logits, value):
    return F.binary_cross_entropy_with_logits(logits, torch.full_like(logits, value))


def pretrain_losses(Pc, Pm, Pd, batch, n_head=8, drop=None):
    """Returns (s_loss, c_loss, dn_loss); total = their sum (main_pretrain.py:77)."""
    x, nx1, nx2, nx, label, c_label = batch
    s_loss = F.cross_entropy(M.textcnn(Pc, x, drop), label)
    c_loss = F.mse_loss(M.matcher(Pm, nx1, nx2, n_head, drop), c_label)
    dn_loss = token_ce(M.mlm(Pd, nx, n_head, drop), x)
    return s_loss, c_loss, dn_loss


def warmup_loss(Pg, batch, coins, drop=None):
    nx, x, labels = batch
    logits = M.denoise_lstm(Pg, nx, labels, x, labels, coins=coins, drop=drop)
    return token_ce(logits, x)


def optimize_g_losses(Pg, Pc, Pm, Pdisc, batch, coins, hp, n_head=8, max_len=None,
                      drop_g=None, drop_g2=None, drop_c=None):
    """Generator step (main_optimize.py:96-113).  hp: dict with w_s,w_c,w_adv,w_bt,tau,gap.
    disc runs in eval mode (``self.disc.eval()`` at :102) -> no dropout there.
    Returns dict(loss, G, STI, CP, BK, sample_p)."""
    x, labels = batch
    sample_p = M.denoise_lstm(Pg, x, labels, None, 1 - labels, "softmax", hp["tau"],
                              max_len=max_len, drop=drop_g)
    s_logits = M.textcnn(Pc, sample_p, drop_c)
    c_logits = M.matcher(Pm, sample_p, x, n_head, drop_c)
    adv_logits = M.relgan_d(Pdisc, sample_p, None)
    bk_logits = M.denoise_lstm(Pg, sample_p.argmax(-1), 1 - labels, x, labels, coins=coins,
                               drop=drop_g2)
    s_loss = F.cross_entropy(s_logits, 1 - labels)
    c_loss = F.mse_loss(c_logits, torch.full_like(c_logits, hp["gap"]))
    g_loss = bce_logits(adv_logits, 1.0)
    bk_loss = token_ce(bk_logits, x)
    loss = hp["w_bt"] * bk_loss + hp["w_c"] * c_loss + hp["w_adv"] * g_loss + hp["w_s"] * s_loss
    return {"loss": loss, "G": g_loss, "STI": s_loss, "CP": c_logits.mean(), "BK": bk_loss,
            "sample_p": sample_p}


def optimize_d_losses(Pg, Pdisc, batch, hp, max_len=None, drop_g=None, drop_d_real=None,
                      drop_d_fake=None):
    """Discriminator step (main_optimize.py:115-124)."""
    x, labels = batch
    V = Pg["fn_2.weight"].shape[0]
    t_logits = M.relgan_d(Pdisc, F.one_hot(x, V).float(), drop_d_real)
    with torch.no_grad():
        x_ = M.denoise_lstm(Pg, x, labels, None, 1 - labels, "softmax", hp["tau"],
                            max_len=max_len, drop=drop_g)
    f_logits = M.relgan_d(Pdisc, x_, drop_d_fake)
    d_loss = 0.5 * (bce_logits(t_logits, 1.0) + bce_logits(f_logits, 0.0))
    return {"loss": hp["w_adv"] * d_loss, "D": d_loss}


def optimize_val_loss(Pg, Pc, Pm, Pnt, batch, hp, n_head=8, max_len=None):
    """validation_step (main_optimize.py:127-141), eval mode."""
    x, labels = batch
    sample_p = M.denoise_lstm(Pg, x, labels, None, 1 - labels, "softmax", hp["tau"], max_len=max_len)
    tokens = sample_p.argmax(-1)
    s_loss = F.cross_entropy(M.textcnn(Pc, tokens), 1 - labels)
    c_logits = M.matcher(Pm, tokens, x, n_head)
    nt_loss = token_ce(M.mlm(Pnt, tokens, n_head), tokens)
    return nt_loss + s_loss + c_logits.mean()


def greedy_decode(Pg, x, labels, max_len):
    """test_step (main_optimize.py:157-164): free-running argmax ids, max_len steps."""
    logits = M.denoise_lstm(Pg, x, labels, None, 1 - labels, max_len=max_len)
    return logits.argmax(-1)
