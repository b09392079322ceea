"""The three training stages as step functions over the HIP-backed modules, with the optimiser and
Trainer semantics the reference obtains from pytorch_lightning restated as this build's own
contract:

  * per batch, per optimizer in order: only that optimizer's parameters require grad (critics stay
    frozen in the generator step yet gradient flows through them); training_step -> backward ->
    clip_grad_norm_ over every parameter that currently holds a gradient (including the
    discriminator's accumulating gradients during the generator step) -> optimizer_step hook;
  * optimize stage: generator steps + zero_grad every batch, discriminator only when
    batch_idx % 4 == 0, its gradients accumulating in between (main_optimize.py:78-88);
  * all losses are means; token CE counts PAD targets.

Reference: src/main_pretrain.py:49-77, src/main_warmup.py:36-58, src/main_optimize.py:69-141.
"""
import random

import torch
import torch.nn as nn

from . import gen_fn, ops
from .model import MLM, DenoiseLSTM, Matcher, RelGAN_D, TextCNN
from .optim import FlatGroup, FlatSlice, clip_groups



def _arena_tag(stage, batch, extra=None):
    """Key of ops.zero_arena's high-water mark: the zero-initialised scratch a step needs depends on the stage and the batch shape."""
    return (type(stage).__name__, id(stage), tuple(tuple(t.shape) for t in batch if hasattr(t, "shape")), extra)

class Fork:
    """Run independent branches of a step on forked HIP streams and join them.  Branch 0 stays on
    the current stream.  Autograd replays each branch's backward on the stream its forward ran on,
    so the backward pass forks and joins the same way; inside a hipGraph capture the branches
    become parallel sub-graphs.  OFF by default: measured on MI355X (round 1) the branches do not
    overlap -- eager and hipGraph replay both take the sum of the kernel durations (38.2 ms/step
    either way, 40.6 ms with forked streams in eager mode) -- so the step stays serial unless
    CST_FORK=1."""

    def __init__(self, n, stage=""):
        import os
        want = os.environ.get("CST_FORK", "0")                 # "1": every stage, or a stage name ("pretrain" / "optimize")
        self.enabled = torch.cuda.is_available() and (want == "1" or (stage and want == stage))
        self.streams = [torch.cuda.Stream() for _ in range(n - 1)] if self.enabled else []

    def run(self, fns):
        if not self.enabled:
            return [f() for f in fns]
        main = torch.cuda.current_stream()
        outs = [None] * len(fns)
        for s, (i, f) in zip(self.streams, list(enumerate(fns))[1:]):
            s.wait_stream(main)
            ops.set_lane(i)
            with torch.cuda.stream(s):
                outs[i] = f()
        ops.set_lane(0)
        outs[0] = fns[0]()
        for s in self.streams[:len(fns) - 1]:
            main.wait_stream(s)
        return outs


def bucketed_backward(loss, stack, group, head_params, embed_params, reducer, last=False, tag=""):
    """Backward pass of a transformer critic (MLM / Matcher) driven one encoder layer at a time, so that the all-reduce of a
    layer's gradients runs while the layers below it are still being back-propagated (north_star: "gradient RCCL all-reduce
    over xGMI overlapped with backward"; the reference has no distributed code).

    `stack.taps` holds the tensors at the layer seams of this forward pass.  Buckets, in the order their gradients
    complete: the head (everything after the last layer), layer n-1, ..., layer 0, the embeddings -- each a contiguous
    range of the group's flat gradient buffer (optim.FlatGroup.span), handed to `reducer(bucket, defer=True)` as soon as
    torch.autograd.grad has returned its gradients; only the last bucket of the last critic is reduced with defer=False,
    which makes the stream wait for every collective outstanding.  Under hipGraph capture every reducer call is a
    segment boundary (graphs.GraphedStep), so the collectives run on RCCL's stream between segment replays."""
    taps = stack.taps
    n = len(stack.layers)
    assert taps is not None and len(taps) == n + 1, "EncoderStack.taps was not recorded for this forward pass"
    # the buckets are picked by convention (head, layers, everything named *embedding*): make sure they are a partition of the group --
    # a parameter outside them would get neither a gradient nor an all-reduce on this path only (checked once per group)
    if not getattr(group, "_buckets_checked", False):
        covered = [id(p) for p in head_params] + [id(p) for l in stack.layers for p in l.parameters()] + [id(p) for p in embed_params]
        assert len(covered) == len(set(covered)) and set(covered) == {id(p) for p in group.params}, \
            f"bucketed_backward({tag}): head + layers + embeddings do not partition the optimizer group"
        spans = sorted([group.span(head_params)] + [group.span(list(l.parameters())) for l in stack.layers] + [group.span(embed_params)])
        assert spans[0][0] == 0 and spans[-1][1] == group.total and all(a[1] == b[0] for a, b in zip(spans, spans[1:])), \
            f"bucketed_backward({tag}): the bucket spans do not tile the flat gradient buffer"
        group._buckets_checked = True
    group.direct = True
    try:
        grads = torch.autograd.grad(loss, [taps[n]] + head_params)
        g = grads[0]
        group.put_grads(head_params, grads[1:])
        reducer([FlatSlice(group, *group.span(head_params), tag=f"{tag}.head")], True)
        for i in range(n - 1, -1, -1):
            lp = list(stack.layers[i].parameters())
            grads = torch.autograd.grad(taps[i + 1], [taps[i]] + lp, grad_outputs=g)
            g = grads[0]
            group.put_grads(lp, grads[1:])
            reducer([FlatSlice(group, *group.span(lp), tag=f"{tag}.layer{i}")], True)
        grads = torch.autograd.grad(taps[0], embed_params, grad_outputs=g, allow_unused=True)
        group.put_grads(embed_params, grads)
        if last:
            reducer([FlatSlice(group, *group.span(embed_params), tag=f"{tag}.embed")])
        else:
            reducer([FlatSlice(group, *group.span(embed_params), tag=f"{tag}.embed")], True)
    finally:
        group.direct = False
        taps.clear()


def _set_requires_grad(params, flag):
    for p in params:
        p.requires_grad_(flag)


def _scalar(t):
    return t.reshape(())


class PretrainStage(nn.Module):
    """main_pretrain.py PretrainModel: joint training of TextCNN (CE), Matcher (MSE against the WMD
    label) and MLM (token CE) with Adam(lr 1e-4), clip 5.0; per-model early-freeze flags."""
    clip = 5.0

    def __init__(self, n_vocab, n_class=2, lr=1e-4):
        super().__init__()
        self.classifier = TextCNN(n_vocab, n_class=n_class)
        self.matcher = Matcher(n_vocab)
        self.denoiser = MLM(n_vocab, n_class=n_class)
        self.flags = {"cls": True, "mat": True, "dn": True}
        self.named_models = {"cls": self.classifier, "mat": self.matcher, "dn": self.denoiser}
        self.best_eval = {name: float("inf") for name in self.flags}
        self.lr = lr
        self.groups = None
        self.bucketed = True          # data parallel: per-layer gradient buckets, all-reduce overlapped with the backward pass

    def setup_optim(self):
        dev = next(self.parameters()).device
        self.groups = {k: FlatGroup(m.parameters(), self.lr) for k, m in self.named_models.items()}
        self._scratch = torch.zeros(1, device=dev, dtype=torch.float32)

    def optim_groups(self):
        return list(self.groups.values())

    def replicated_tensors(self):
        return [g.flat_p for g in self.groups.values()]

    def losses(self, batch, seed=None):
        """(s_loss, c_loss, dn_loss) -- main_pretrain.py:66-77; a frozen model contributes None."""
        x, nx_1, nx_2, nx, label, c_label = batch

        def f_dn():
            if not self.flags["dn"]:
                return None
            lg = self.denoiser(nx, seed=seed)
            return ops.token_ce(lg.view(-1, lg.size(-1)), x.reshape(-1), unit_grad=True)

        def f_mat():
            return ops.mse_loss(self.matcher(nx_1, nx_2, seed=seed), c_label) if self.flags["mat"] else None

        def f_cls():
            return ops.token_ce(self.classifier(x, seed=seed), label) if self.flags["cls"] else None

        if not hasattr(self, "_fork"):
            self._fork = Fork(3, "pretrain")
        dn, c, s = self._fork.run([f_dn, f_mat, f_cls])          # three independent critics, three streams
        return s, c, dn

    def train_step(self, batch, seed=None, reducer=None):
        with ops.zero_arena(_arena_tag(self, batch), batch[0].device):
            return self._train_step(batch, seed, reducer)

    def _train_step(self, batch, seed=None, reducer=None):
        """The three critics share nothing, so each runs forward + backward + gradient gather on its own and, under data
        parallelism, its all-reduce is started right away (deferred) while the next critic computes; only the last
        reducer call waits for all of them before the global-norm clip."""
        x, nx_1, nx_2, nx, label, c_label = batch
        # data parallel: the small classifier first (its all-reduce hides under the MLM), then the two transformer critics
        # with their backward passes cut into per-layer buckets (bucketed_backward); single GPU: largest first, one backward each
        bucketed = reducer is not None and self.bucketed
        order = [k for k in (("cls", "dn", "mat") if bucketed else ("dn", "mat", "cls")) if self.flags[k]]
        vals = {"dn": None, "mat": None, "cls": None}
        for i, k in enumerate(order):
            last = i + 1 == len(order)
            stack = {"dn": self.denoiser.lm, "mat": self.matcher.matcher}.get(k)
            if bucketed and stack is not None:
                stack.taps = []
            if k == "dn":
                lg = self.denoiser(nx, seed=seed)
                loss = ops.token_ce(lg.view(-1, lg.size(-1)), x.reshape(-1), unit_grad=True)
            elif k == "mat":
                loss = ops.mse_loss(self.matcher(nx_1, nx_2, seed=seed), c_label)
            else:
                loss = ops.token_ce(self.classifier(x, seed=seed), label)
            vals[k] = loss.detach()
            if bucketed and stack is not None:
                m = self.named_models[k]
                head = list((m.fwd if k == "dn" else m.hidden2logits).parameters())
                embed = [p for n_, p in m.named_parameters() if "embedding" in n_]
                bucketed_backward(loss, stack, self.groups[k], head, embed, reducer, last=last, tag=k)
                stack.taps = None
                continue
            with ops.tt_deferred():                       # the encoder layers' weight gradients go out two layers per launch
                loss.backward()
            self.groups[k].gather_grads(False)
            if reducer is not None:
                if not last:
                    reducer([self.groups[k]], True)
                else:
                    reducer([self.groups[k]])
        live = [self.groups[k] for k in self.flags if self.flags[k]]
        clip_groups(live, self.clip, self._scratch, stepping=live)
        for g in live:
            g.step()
            g.zero_grad()
        total = None
        for k in ("cls", "mat", "dn"):
            if vals[k] is not None:
                total = vals[k] if total is None else total + vals[k]
        return {"s_loss": vals["cls"], "c_loss": vals["mat"], "dn_loss": vals["dn"], "loss": total}


class WarmupStage(nn.Module):
    """main_warmup.py WarmupModel: generator denoising auto-encoding, Adam(lr 1e-3), clip 1.0."""
    clip = 1.0

    def __init__(self, n_vocab, n_class=2, max_len=18, lr=1e-3):
        super().__init__()
        self.generator = DenoiseLSTM(n_vocab, n_class, max_len)
        self.lr = lr
        self.group = None

    def setup_optim(self):
        self.group = FlatGroup(self.generator.parameters(), self.lr)
        self._scratch = torch.zeros(1, device=self.group.flat_p.device, dtype=torch.float32)

    def optim_groups(self):
        return [self.group]

    def replicated_tensors(self):
        return [self.group.flat_p]

    def loss(self, batch, coins=None, seed=None):
        nx, x, labels = batch
        lg = self.generator(nx, labels, x, labels, coins=coins, seed=seed)      # main_warmup.py:47-48
        return ops.token_ce(lg.view(-1, lg.size(-1)), x.reshape(-1), unit_grad=True)

    def train_step(self, batch, coins=None, seed=None, reducer=None):
        with ops.zero_arena(_arena_tag(self, batch), batch[0].device):
            return self._train_step(batch, coins, seed, reducer)

    def _train_step(self, batch, coins=None, seed=None, reducer=None):
        loss = self.loss(batch, coins, seed)
        loss.backward()
        self.group.gather_grads(False)
        if reducer is not None:
            reducer([self.group])
        clip_groups([self.group], self.clip, self._scratch, stepping=[self.group])
        self.group.step()
        self.group.zero_grad()
        return {"dn_loss": loss, "loss": loss}


class OptimizeStage(nn.Module):
    """main_optimize.py GenerationTuner: two-optimizer adversarial fine-tuning."""
    clip = 1.0

    def __init__(self, n_vocab, n_class=2, max_len=18, w_s=0.1, w_c=0.5, w_adv=1.0, w_bt=1.0, tau=0.1, gap=0.0,
                 lr=1e-5):
        super().__init__()
        self.classifier = TextCNN(n_vocab, n_class=n_class)
        self.matcher = Matcher(n_vocab)
        self.nt_checker = MLM(n_vocab, n_class)
        self.disc = RelGAN_D(n_vocab)
        self.generator = DenoiseLSTM(n_vocab, n_class, max_len)
        self.ws, self.wc, self.w_adv, self.w_bt, self.tau, self.gap = w_s, w_c, w_adv, w_bt, tau, gap
        self.lr = lr
        self.n_vocab = n_vocab

    def setup_optim(self):
        self.g_group = FlatGroup(self.generator.parameters(), self.lr)
        self.d_group = FlatGroup(self.disc.parameters(), self.lr)
        self._scratch = torch.zeros(1, device=self.g_group.flat_p.device, dtype=torch.float32)
        self._all = list(self.parameters())

    def optim_groups(self):
        return [self.g_group, self.d_group]

    def replicated_tensors(self):
        """Everything that must be identical on every data-parallel rank: the trained groups and the frozen critics."""
        frozen = [p.data for m in (self.classifier, self.matcher, self.nt_checker) for p in m.parameters()]
        return [self.g_group.flat_p, self.d_group.flat_p] + frozen

    def forward(self, x, src_labels, tgt_labels, tau, seed=None):
        return self.generator(x, src_labels, None, tgt_labels, res_type="softmax", tau=tau, seed=seed)

    # ---- optimizer_idx 0 (main_optimize.py:96-113) ------------------------------------------
    def g_losses(self, batch, coins=None, seed=None):
        x, labels = batch
        with gen_fn.shared_param_grads():                                     # both decodes of this step feed one backward pass
            return self._g_losses(x, labels, coins, seed)

    def _g_losses(self, x, labels, coins, seed):
        sample_p = self.forward(x, labels, 1 - labels, self.tau, seed=seed)
        with torch.no_grad():
            tokens = self.generator.last_ids.t().contiguous()                 # == sample_p.argmax(-1)
        inv_labels = 1 - labels
        was_training = self.disc.training
        self.disc.eval()                                                      # main_optimize.py:102

        # four independent consumers of sample_p: the back-translation decode (latency-bound small
        # kernels) overlaps with the critics' large GEMMs
        def f_bk():
            lg = self.generator(tokens, inv_labels, x, labels, coins=coins, seed=None if seed is None else seed + 1)
            return ops.token_ce(lg.view(-1, lg.size(-1)), x.reshape(-1), weight=self.w_bt, unit_grad=True)

        def f_mat():
            cl = self.matcher(sample_p, x, seed=seed)
            return cl, ops.mse_loss(cl, None, self.gap)

        def f_cls():
            return ops.token_ce(self.classifier(sample_p, seed=seed), inv_labels, weight=1.0)

        def f_adv():
            return ops.bce_logits_loss(self.disc(sample_p), 1.0)

        if not hasattr(self, "_fork"):
            self._fork = Fork(4, "optimize")
        # the three critics embed the same sample_p: one shared product (and one shared d sample_p) instead of three
        with ops.shared_soft_embed(sample_p, [(self.matcher.token_embedding.weight, False), (self.classifier.embedding.weight, False),
                                              (self.disc.embeddings.weight, True)]):
            bk_loss, (c_logits, c_loss), s_loss, g_loss = self._fork.run([f_bk, f_mat, f_cls, f_adv])
        self.disc.train(was_training)
        loss = bk_loss + self.wc * c_loss + self.w_adv * g_loss + self.ws * s_loss
        return {"loss": loss, "G": g_loss, "STI": s_loss, "CP_logits": c_logits, "BK": bk_loss / self.w_bt if self.w_bt else bk_loss,
                "sample_ids": tokens}

    # ---- optimizer_idx 1 (main_optimize.py:115-124) -----------------------------------------
    def d_losses(self, batch, seed=None):
        x, labels = batch
        t_logits = self.disc(x, seed=seed)                                    # ids fast path == one_hot(x).float()
        with torch.no_grad():
            x_ = self.forward(x, labels, 1 - labels, self.tau, seed=None if seed is None else seed + 2)
        f_logits = self.disc(x_, seed=None if seed is None else seed + 3)
        d_loss = 0.5 * (ops.bce_logits_loss(t_logits, 1.0) + ops.bce_logits_loss(f_logits, 0.0))
        return {"loss": self.w_adv * d_loss, "D": d_loss}

    def train_step(self, batch, batch_idx, coins=None, seed=None, reducer=None):
        with ops.zero_arena(_arena_tag(self, batch, batch_idx % 4 == 0), batch[0].device):
            return self._train_step(batch, batch_idx, coins, seed, reducer)

    def _train_step(self, batch, batch_idx, coins=None, seed=None, reducer=None):
        logs = {}
        # generator step
        _set_requires_grad(self._all, False)
        _set_requires_grad(self.g_group.params, True)
        r = self.g_losses(batch, coins, seed)
        r["loss"].backward()
        self.g_group.gather_grads(False)
        if reducer is not None:
            reducer([self.g_group])
        self.d_group.has_grad = True                      # its (possibly all-zero) accumulated grads join the norm
        clip_groups([self.g_group, self.d_group], self.clip, self._scratch, stepping=[self.g_group])
        self.g_group.step()
        self.g_group.zero_grad()
        logs.update(G=r["G"], STI=r["STI"], BK=r["BK"], CP_logits=r["CP_logits"], g_total=r["loss"], sample_ids=r["sample_ids"])
        # discriminator step
        _set_requires_grad(self._all, False)
        _set_requires_grad(self.d_group.params, True)
        self.disc.train(self.training)                                        # main_optimize.py:116
        d = self.d_losses(batch, seed)
        d["loss"].backward()
        self.d_group.gather_grads(True)                   # += : accumulates until zero_grad (flat_g is 0 after it)
        if reducer is not None:
            # EVERY batch, not only the stepping ones: the accumulating buffer enters both clips of every batch (its
            # sum of squares scales the generator's gradients above and is rescaled in place here), so it has to be
            # rank-identical at all times.  Induction: all ranks hold the same accumulated buffer A before this batch's
            # backward; afterwards rank r holds A + g_r, and AVG over ranks gives A + mean_r(g_r) = A + the global-batch
            # gradient on every rank (all D losses are batch means over equal shards) -- exactly the one-process buffer.
            reducer([self.d_group])
        clip_groups([self.g_group, self.d_group], self.clip, self._scratch, stepping=[self.d_group] if batch_idx % 4 == 0 else ())
        if batch_idx % 4 == 0:                                                # main_optimize.py:85-88
            self.d_group.step()
            self.d_group.zero_grad()
        logs.update(D=d["D"])
        return logs

    # ---- validation_step (main_optimize.py:127-141) -----------------------------------------
    @torch.no_grad()
    def val_loss(self, batch):
        x, labels = batch
        self.forward(x, labels, 1 - labels, self.tau)
        tokens = self.generator.last_ids.t().contiguous()
        s_loss = ops.token_ce(self.classifier(tokens), 1 - labels)
        c_logits = self.matcher(tokens, x)
        nt = self.nt_checker(tokens)
        nt_loss = ops.token_ce(nt.view(-1, nt.size(-1)), tokens.reshape(-1))
        return _scalar(nt_loss) + _scalar(s_loss) + c_logits.mean()

    # ---- test_step (main_optimize.py:157-164) -----------------------------------------------
    @torch.no_grad()
    def transfer(self, batch):
        x, labels = batch
        self.generator(x, labels, None, 1 - labels)
        return self.generator.last_ids.t().contiguous()                       # (B, max_len) greedy ids
