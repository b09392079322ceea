"""GPU parity of the five HIP-backed modules against (a) the golden vectors generated from the
imported reference modules and (b) the CPU oracle on the same seeded inputs.

Tolerances: 'f32' mode (exact-fp32 MFMA) must match the fp32 reference to rtol 2e-3 on outputs and
5e-3 on gradients (different summation orders through 6 layers / 18 recurrent steps); greedy-decode
token ids must be bit-exact in f32 mode.

'bf16' mode (bf16 operands, fp32 accumulate) -- the mode the benchmark runs in -- is held to the SAME
reference vectors through relative L2 deviations, ||got - ref|| / ||ref||: BF16_OUT for outputs,
BF16_GRAD for every parameter / input gradient (for gradients stored as norm + strided sample: RMS
deviation of the sample over the RMS of the whole gradient); modules whose gradient is ROUTED by an
arg-max (TextCNN, RelGAN_D: max over time; Matcher: max over the sequence) get BF16_GRAD_ROUTED for the
worst tensor and BF16_GRAD_ROUTED_MEDIAN for the median over their tensors, because a near-tie resolved
the other way under bf16 rounding moves one feature's whole gradient to another position (at the B = 2 of
the ref / long configurations one flip is 1/256 of a filter bank).  Decodes that feed their own argmax back
(soft / greedy / scheduled-sampling coins) can legitimately flip a near-tie under bf16 rounding and then
follow a different trajectory for that sentence: for those, at least BF16_ROWS of the sentences must
reproduce the reference's token ids at EVERY step, and the outputs of exactly those sentences are held to
BF16_OUT; the pure teacher-forced vector (gen.tf0, no feedback) is compared whole, gradients included.
The measured deviations are appended to gpurun_out/parity_report.jsonl and quoted in DESIGN.md."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import CONFIGS, check_grads, det_params, grad_rel_l2, load_golden, lossw, rel_l2, report, sd_shapes, soft_input  # noqa: E402
from oracle.configs import noised_len, seg2_len  # noqa: E402
from oracle.detinit import det_state_dict  # noqa: E402
from helpers import SEEDS  # noqa: E402


@pytest.fixture(scope="module")
def cst():
    assert torch.cuda.is_available()
    import consistent__style_transfer_amd as pkg
    from consistent__style_transfer_amd import model, ops
    return pkg, model, ops


def set_constants(model, c):
    from consistent__style_transfer_amd.model import classifier, discriminator, match, mlm, rnn
    mlm.d_model = match.d_model = c["d_model"]
    mlm.n_head = match.n_head = c["n_head"]
    mlm.n_layer = match.n_layer = c["n_layer"]
    rnn.d_embed, rnn.d_enc, rnn.d_dec = c["g_embed"], c["g_enc"], c["g_dec"]
    classifier.d_embed, classifier.kernel_number = c["c_embed"], c["c_filters"]
    discriminator.embed_dim, discriminator.num_rep, discriminator.dis_num_filters = c["d_embed"], c["d_rep"], c["d_filters"]


def build(model, name, which):
    c = CONFIGS[name]
    set_constants(model, c)
    V = c["V"]
    m = {"G": lambda: model.DenoiseLSTM(V, 2, c["max_len"]), "cls": lambda: model.TextCNN(V, 2),
         "mat": lambda: model.Matcher(V), "dn": lambda: model.MLM(V, 2), "disc": lambda: model.RelGAN_D(V)}[which]()
    sd = det_state_dict({k: v.shape for k, v in m.state_dict().items()}, SEEDS[which])
    m.load_state_dict(sd)
    m = m.cuda()
    m.eval()                      # dropout off: the golden vectors were taken with p = 0
    return m


NAMES = ["tiny", "ref", "b16", "long"]
BF16_OUT = 2e-2           # relative L2 of an output tensor
# measured on MI355X, round 2 (gpurun_out/parity_report.jsonl): outputs <= 1.3e-2; gradients without arg-max routing <= 7.1e-2
# (generator transfer.weight, MLM linear1.weight of the first layers), median over tensors 1-5e-2; routed <= 0.22 (TextCNN
# convs.2.weight at B = 2), median <= 5e-2
BF16_GRAD = 0.10          # relative L2 of a gradient (6 encoder layers / 8-40 recurrent steps of bf16 products compound)
BF16_GRAD_LONG = 0.40     # generator of the `long` configuration: 40 recurrent steps at hidden width 16 / 32 through the tau = 0.1
                          # straight-through softmax (measured 0.29 on encoder.bias_ih_l0_reverse with identical token trajectories)
BF16_GRAD_ROUTED = 0.30   # worst tensor of a module whose gradients are routed by an arg-max
BF16_GRAD_ROUTED_MEDIAN = 0.08
BF16_ROWS = 0.75          # fraction of sentences whose fed-back token ids all agree with the reference


def tols(prec):
    return (2e-3, 2e-4, 5e-3, 2e-3) if prec == "f32" else (5e-2, 5e-2, None, None)


def cmp_out(y, ref, prec, scale_atol=True, tag=""):
    ref = np.asarray(ref)
    if prec == "bf16":
        dev = rel_l2(y, ref)
        report("modules.out", tag=tag, dev=dev)
        assert dev <= BF16_OUT, (tag, dev)
        return
    rt, at, _, _ = tols(prec)
    at = at * max(1.0, float(np.abs(ref).max())) if scale_atol else at
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=rt, atol=at)


def cmp_rows(y, ref, tag, min_rows=BF16_ROWS):
    """bf16 mode, decodes with argmax feedback: sentences whose ids agree with the reference at every step."""
    y = y.detach().float().cpu().numpy()
    ref = np.asarray(ref)
    ok = (y.argmax(-1) == ref.argmax(-1)).all(1)
    frac = float(ok.mean())
    dev = rel_l2(y[ok], ref[ok]) if ok.any() else float("nan")
    report("modules.rows", tag=tag, rows_agree=frac, dev=dev, batch=int(y.shape[0]))
    if y.shape[0] >= 8:                                    # a fraction of 2 or 3 sentences says nothing (40 feedback steps at B = 2)
        assert frac >= min_rows, (tag, frac)
    if ok.any():
        assert dev <= BF16_OUT, (tag, dev)
    return bool(ok.all())


def named_grads(m):
    return {k: p.grad for k, p in m.named_parameters() if p.grad is not None}


def grad_q90(G, prefix, named, input_grad=None):
    out = {}
    items = dict(named)
    if input_grad is not None:
        items["__input__"] = input_grad
    for k, g in items.items():
        g = g.detach().cpu().numpy().astype(np.float64)
        full, nrm, smp = f"{prefix}.grad.{k}", f"{prefix}.gradnorm.{k}", f"{prefix}.gradsample.{k}"
        if full in G and np.linalg.norm(G[full]) > 1e-6:
            ref = G[full].astype(np.float64)
            out[k] = float(np.quantile(np.abs(g - ref), 0.9) / np.sqrt(np.mean(ref ** 2)))
        elif nrm in G and G[nrm][0] > 1e-6:
            stride = next(st for st in (97, 1009) if g.reshape(-1)[::st].size == G[smp].size)
            out[k] = float(np.quantile(np.abs(g.reshape(-1)[::stride] - G[smp]), 0.9) / (G[nrm][0] / np.sqrt(g.size)))
    return out


def run_grads(G, prefix, m, loss, prec, inp=None, same_trajectory=True, routed=False, batch=8, grad_tol=None):
    grad_tol = BF16_GRAD if grad_tol is None else grad_tol
    m.zero_grad()
    loss.backward()
    if prec == "f32":
        check_grads(G, prefix, named_grads(m), 5e-3, 2e-3, None if inp is None else inp.grad)
        return
    devs = grad_rel_l2(G, prefix, named_grads(m), None if inp is None else inp.grad)
    assert devs, f"no gradient of {prefix} was checked"
    worst = max(devs, key=devs.get)
    report("modules.grad", tag=prefix, worst=worst, dev=devs[worst], median=float(np.median(list(devs.values()))),
           same_trajectory=same_trajectory)
    if same_trajectory and routed:
        # TextCNN / RelGAN_D (max over time, classifier.py:32, discriminator.py:42) and the Matcher (max over the sequence,
        # match.py:41): bf16 rounding can flip a near-tie of the arg-max, which moves that feature's gradient to another
        # position wholesale (at B = 2 one flip is 1/256 of a filter bank).  Most elements must still agree closely.
        med = float(np.median(list(devs.values())))
        assert med <= BF16_GRAD_ROUTED_MEDIAN, (prefix, "median", med)
        assert devs[worst] <= BF16_GRAD_ROUTED, (prefix, worst, devs[worst])
    elif same_trajectory:
        assert devs[worst] <= grad_tol, (prefix, worst, devs[worst])
    elif batch >= 8:
        # some sentence followed a different token trajectory than the reference: element-wise comparison is void, the
        # gradient norms still have to agree (with 2-3 sentences one different trajectory IS a different batch: report only)
        for k, g in named_grads(m).items():
            full, nrm = f"{prefix}.grad.{k}", f"{prefix}.gradnorm.{k}"
            ref = np.linalg.norm(G[full].astype(np.float64)) if full in G else (G[nrm][0] if nrm in G else None)
            if ref is not None and ref > 1e-3:
                np.testing.assert_allclose(float(g.double().norm()), ref, rtol=1e-1, err_msg=k)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("name", NAMES)
def test_textcnn(cst, name, prec):
    pkg, model, ops = cst
    ops.set_precision(prec)
    c, G = CONFIGS[name], load_golden("modules", name)
    m = build(model, name, "cls")
    x = torch.from_numpy(G["x"]).cuda()
    y = m(x)
    cmp_out(y, G["cls.ids.out"], prec, tag=f"{name}.cls.ids.out")
    run_grads(G, "cls.ids", m, lossw("cls.ids", y), prec, routed=True)
    sp = soft_input(c["B"], c["L"], c["V"], 11, "cuda")
    y = m(sp)
    cmp_out(y, G["cls.soft.out"], prec, tag=f"{name}.cls.soft.out")
    run_grads(G, "cls.soft", m, lossw("cls.soft", y), prec, sp, routed=True)
    ops.set_precision("bf16")


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("name", NAMES)
def test_mlm(cst, name, prec):
    pkg, model, ops = cst
    ops.set_precision(prec)
    c, G = CONFIGS[name], load_golden("modules", name)
    m = build(model, name, "dn")
    y = m(torch.from_numpy(G["x"]).cuda())
    cmp_out(y, G["mlm.ids.out"], prec, tag=f"{name}.mlm.ids.out")
    run_grads(G, "mlm.ids", m, lossw("mlm.ids", y), prec)
    sp = soft_input(c["B"], c["L"], c["V"], 12, "cuda")
    y = m(sp)
    cmp_out(y, G["mlm.soft.out"], prec, tag=f"{name}.mlm.soft.out")
    run_grads(G, "mlm.soft", m, lossw("mlm.soft", y), prec, sp)
    ops.set_precision("bf16")


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("name", NAMES)
def test_matcher(cst, name, prec):
    pkg, model, ops = cst
    ops.set_precision(prec)
    c, G = CONFIGS[name], load_golden("modules", name)
    m = build(model, name, "mat")
    x, x2 = torch.from_numpy(G["x"]).cuda(), torch.from_numpy(G["x2"]).cuda()
    y = m(x, x2)
    cmp_out(y, G["mat.ids.out"], prec, tag=f"{name}.mat.ids.out")
    run_grads(G, "mat.ids", m, lossw("mat.ids", y), prec, routed=True)
    sp = soft_input(c["B"], c["L"], c["V"], 13, "cuda")
    y = m(sp, x)
    cmp_out(y, G["mat.soft.out"], prec, tag=f"{name}.mat.soft.out")
    run_grads(G, "mat.soft", m, lossw("mat.soft", y), prec, sp, routed=True)
    ops.set_precision("bf16")


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("name", NAMES)
def test_relgan_d(cst, name, prec):
    pkg, model, ops = cst
    ops.set_precision(prec)
    c, G = CONFIGS[name], load_golden("modules", name)
    m = build(model, name, "disc")
    sp = soft_input(c["B"], c["L"], c["V"], 14, "cuda")
    y = m(sp)
    cmp_out(y, G["disc.soft.out"], prec, tag=f"{name}.disc.soft.out")
    run_grads(G, "disc.soft", m, lossw("disc.soft", y), prec, sp, routed=True)
    x = torch.from_numpy(G["x"]).cuda()
    y = m(x)                                                    # ids fast path == dense one-hot
    cmp_out(y, G["disc.onehot.out"], prec, tag=f"{name}.disc.onehot.out")
    run_grads(G, "disc.onehot", m, lossw("disc.onehot", y), prec, routed=True)
    y = m(torch.nn.functional.one_hot(x, c["V"]).float())       # the reference's own calling convention
    cmp_out(y, G["disc.onehot.out"], prec, tag=f"{name}.disc.onehot.out")
    ops.set_precision("bf16")


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("name", NAMES)
def test_generator(cst, name, prec):
    """b16 in bf16 mode is the case that runs the whole-sequence encoder kernels (cst_lstm_seq_fwd/bwd), the 16-row
    MFMA recurrent products, the fused cell + attention decoder step and the transposed-read weight gradients against
    vectors recorded from the reference (rnn.py:55-98)."""
    pkg, model, ops = cst
    ops.set_precision(prec)
    c, G = CONFIGS[name], load_golden("modules", name)
    m = build(model, name, "G")
    x, nx, labels = (torch.from_numpy(G[k]).cuda() for k in ("x", "nx", "labels"))
    bf = prec == "bf16"
    gt = BF16_GRAD_LONG if name == "long" else BF16_GRAD
    # (a') pure teacher forcing: no feedback, compared whole in both modes (outputs and every gradient)
    y = m(nx, labels, x, labels, coins=[0] * x.shape[1])
    cmp_out(y, G["gen.tf0.out"], prec, tag=f"{name}.gen.tf0.out")
    run_grads(G, "gen.tf0", m, lossw("gen.tf0", y), prec, grad_tol=gt)
    # (a) teacher forcing with the recorded coins (argmax fed back where the coin says so)
    y = m(nx, labels, x, labels, coins=G["gen.tf.coins"])
    if bf:
        same = cmp_rows(y, G["gen.tf.out"], f"{name}.gen.tf")
        run_grads(G, "gen.tf", m, lossw("gen.tf", y), prec, same_trajectory=same, batch=c["B"], grad_tol=gt)
    else:
        cmp_out(y, G["gen.tf.out"], prec)
        run_grads(G, "gen.tf", m, lossw("gen.tf", y), prec)
    # (b) softmax / straight-through, free running
    for tag, tau in (("gen.soft", 0.1), ("gen.soft1", 1.0)):
        y = m(x, labels, None, 1 - labels, res_type="softmax", tau=tau)
        np.testing.assert_allclose(y.detach().sum(-1).cpu().numpy(), 1.0, rtol=1e-4)
        assert torch.equal(y.detach().argmax(-1), m.last_ids.t())          # the ids fed back are the argmax of what is returned
        if bf:
            same = cmp_rows(y, G[tag + ".out"], f"{name}.{tag}")
            run_grads(G, tag, m, lossw(tag, y), prec, same_trajectory=same, batch=c["B"], grad_tol=gt)
        else:
            np.testing.assert_allclose(y.detach().cpu().numpy(), G[tag + ".out"], rtol=5e-3, atol=2e-5)
            run_grads(G, tag, m, lossw(tag, y), prec)
    # (c) greedy decode: exact ids in f32 mode
    with torch.no_grad():
        y = m(x, labels, None, 1 - labels)
    if not bf:
        assert np.array_equal(y.argmax(-1).cpu().numpy(), G["gen.greedy.ids"])
        assert np.array_equal(m.last_ids.t().cpu().numpy(), G["gen.greedy.ids"])
        cmp_out(y, G["gen.greedy.out"], prec)
    else:
        cmp_rows(y, G["gen.greedy.out"], f"{name}.gen.greedy")
        agree = float((y.argmax(-1).cpu().numpy() == G["gen.greedy.ids"]).mean())
        report("modules.greedy", tag=name, token_agreement=agree)
        assert agree >= 0.9, agree         # ids are pinned bit-exactly in f32 mode; bf16 logits may flip near-ties
    # (d) 3-D (soft) encoder input
    sp = soft_input(c["B"], c["L"], c["V"], 15, "cuda")
    y = m(sp, labels, x, labels, coins=G["gen.soft_in.coins"])
    if bf:
        same = cmp_rows(y, G["gen.soft_in.out"], f"{name}.gen.soft_in")
        run_grads(G, "gen.soft_in", m, lossw("gen.soft_in", y), prec, sp, same_trajectory=same, batch=c["B"], grad_tol=gt)
    else:
        cmp_out(y, G["gen.soft_in.out"], prec)
        run_grads(G, "gen.soft_in", m, lossw("gen.soft_in", y), prec, sp)
    ops.set_precision("bf16")


def test_dropout_train_mode_matches_oracle(cst):
    """Train-mode dropout follows the shared counter-based RNG contract: the HIP modules and the
    oracle (fed the same seed) agree element-wise."""
    pkg, model, ops = cst
    from oracle import modules as M
    ops.set_precision("f32")
    name = "tiny"
    c, G = CONFIGS[name], load_golden("modules", name)
    x, x2 = torch.from_numpy(G["x"]), torch.from_numpy(G["x2"])
    labels = torch.from_numpy(G["labels"])
    seed = 4242
    drop = M.DropSpec(seed)
    for which, run_hip, run_or in (
        ("cls", lambda m: m(x.cuda(), seed=seed), lambda P: M.textcnn(P, x, drop)),
        ("dn", lambda m: m(x.cuda(), seed=seed), lambda P: M.mlm(P, x, c["n_head"], drop)),
        ("mat", lambda m: m(x.cuda(), x2.cuda(), seed=seed), lambda P: M.matcher(P, x, x2, c["n_head"], drop)),
        ("disc", lambda m: m(x.cuda(), seed=seed), lambda P: M.relgan_d(P, x, drop)),
        ("G", lambda m: m(x2.cuda(), labels.cuda(), x.cuda(), labels.cuda(), coins=G["gen.tf.coins"], seed=seed),
         lambda P: M.denoise_lstm(P, x2, labels, x, labels, coins=G["gen.tf.coins"], drop=drop)),
    ):
        m = build(model, name, which)
        m.train()
        P = det_params(name, which)
        with torch.no_grad():
            y = run_hip(m)
            ref = run_or(P)
        np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=2e-3, atol=2e-4, err_msg=which)
    ops.set_precision("bf16")


def test_generator_bf16_path_matches_f32_path(cst):
    """Reference-size generator: the bf16-operand GEMM path (cst_gemm_bf16 with fused bf16 producers)
    against the exact-fp32 path on the same weights -- single decode step (no argmax feedback yet, so
    no token can flip), logits and all parameter gradients within the bf16 tolerance."""
    pkg, model, ops = cst
    name = "ref"
    c, G = CONFIGS[name], load_golden("modules", name)
    x1 = torch.from_numpy(G["x"]).cuda()[:, :1].contiguous()
    nx, labels = torch.from_numpy(G["nx"]).cuda(), torch.from_numpy(G["labels"]).cuda()
    res = {}
    for prec in ("f32", "bf16"):
        ops.set_precision(prec)
        m = build(model, name, "G")
        y = m(nx, labels, x1, labels, coins=[0])
        m.zero_grad()
        lossw("gen.one", y).backward()
        res[prec] = (y.detach().float().cpu().numpy(), {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None})
    ops.set_precision("bf16")
    yf, gf = res["f32"]
    yb, gb = res["bf16"]
    np.testing.assert_allclose(yb, yf, rtol=3e-2, atol=3e-2 * float(np.abs(yf).max()))
    for k in gf:
        nf, nb = np.linalg.norm(gf[k]), np.linalg.norm(gb[k] - gf[k])
        assert nb <= 1e-1 * max(nf, 1e-6), (k, nb, nf)      # bf16 rounding compounds through fn_2, fn_1, the cell and transfer


def test_generator_generic_loss_fast_decode_equals_per_step_decode(cst, monkeypatch):
    """ADVICE r3 (gen_fn.py): a hard, teacher-forced decode in TRAIN mode (dropout on) under a loss that is NOT token_ce(unit_grad=True) --
    here sum(logits * w) -- reaches the backward without bf16 dlogits.  The fast decode loop writes only the bf16 dropout(i_ffn), so
    the fn_1 weight gradient must be taken from that copy; it used to read an unwritten fp32 buffer.  Every parameter gradient of the
    fast loop must equal the per-step loop's (CST_DECODE_SLOW=1: same masks, same products up to bf16 rounding of the operands)."""
    pkg, model, ops = cst
    name = "b16"
    c, G = CONFIGS[name], load_golden("modules", name)
    x, nx, labels = torch.from_numpy(G["x"]).cuda(), torch.from_numpy(G["nx"]).cuda(), torch.from_numpy(G["labels"]).cuda()
    coins = [1, 0] * (x.shape[1] // 2) + [1] * (x.shape[1] % 2)
    ops.set_precision("bf16")
    res = {}
    for slow in ("0", "1"):
        monkeypatch.setenv("CST_DECODE_SLOW", slow)
        m = build(model, name, "G")
        m.train()
        y = m(nx, labels, x, labels, coins=coins, seed=977)
        w = torch.from_numpy(np.random.RandomState(5).standard_normal(tuple(y.shape)).astype(np.float32)).cuda()
        m.zero_grad()
        (y * w).sum().backward()
        res[slow] = (y.detach().float().cpu().numpy(), {k: p.grad.detach().float().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None})
    (yf, gf), (ys, gs) = res["0"], res["1"]
    assert np.isfinite(yf).all() and rel_l2(yf, ys) < 2e-2
    worst = {}
    for k in gs:
        assert np.isfinite(gf[k]).all(), k
        worst[k] = float(np.linalg.norm(gf[k] - gs[k]) / max(np.linalg.norm(gs[k]), 1e-12))
    report("gen.generic_loss.fast_vs_slow", config=name, mode="bf16", **worst)
    assert worst["fn_1.weight"] < 5e-2, worst
    assert max(worst.values()) < 0.1, worst
