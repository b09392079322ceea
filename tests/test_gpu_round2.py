"""GPU: what round 1 left untested ("What's weak" 3-5):

  * every BASELINE config at full size -- book (6-layer d=512, B=512, L=30: Matcher over 60 positions) and the d=768
    shard (6-layer, head dim 96, B=256): all three stage steps run, losses sit where a random-init model must put them,
    every parameter moves and stays finite;
  * hipGraph step cache over MANY batch shapes (real batches are padded to the per-batch maximum and transfer_noise changes
    lengths: loader.py:46-70), with evictions, against the eager loop;
  * an EAGER forward (validation) after a run of graph replays sees the weights the replays trained;
  * >= 200 real pretrain batches of the dev-sample corpus in graph mode;
  * two data-parallel ranks driving the PRODUCT OptimizeStage (gloo on one GPU): replicas bit-identical after 8 batches and
    equal to the one-rank global-batch run.
"""
import math
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from curve_inputs import HP, opt_batch, warm_batch  # noqa: E402
from helpers import CONFIGS, load_golden  # noqa: E402
from test_gpu_modules import set_constants  # noqa: E402
from test_gpu_stages import _load, cu, make_opt  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# --------------------------------------------------------------------------------- full-size configs
@pytest.mark.parametrize("workload", ["yelp_4l_d512_b256", "yelp_6l_d768_b256", "book_6l_d512_b512", "yelp_6l_d512_b256"])
def test_full_size_stage_steps(workload):
    import bench
    from consistent__style_transfer_amd import model, ops
    from consistent__style_transfer_amd import synthetic as syn
    set_constants(model, CONFIGS["ref"])
    ops.set_precision("bf16")
    w = bench.WORKLOADS[workload]
    B, L, V = w["B"], w["L"], w["V"]
    dev = torch.device("cuda")
    pre, wu, opt = bench.build_stages(w, dev)
    try:
        lnV = math.log(V)
        # ---- pretrain (main_pretrain.py:66-77): three critics, one clip, one Adam step each
        before = {k: g.flat_p.clone() for k, g in pre.groups.items()}
        r = pre.train_step(tuple(t.to(dev) for t in syn.pretrain_batch(B, L, V, 11)), seed=1)
        assert abs(r["dn_loss"].item() - lnV) < 1.5, r["dn_loss"].item()           # ~ln(10 000) at random init
        assert 0.3 < r["s_loss"].item() < 2.0 and math.isfinite(r["c_loss"].item())
        for k, g in pre.groups.items():
            assert torch.isfinite(g.flat_p).all(), k
            assert (g.flat_p != before[k]).float().mean().item() > 0.3, k           # embedding rows of unseen tokens stay (TextCNN is 87 % embedding)
        # ---- warmup (main_warmup.py:45-58)
        b0 = wu.group.flat_p.clone()
        coins = torch.randint(0, 2, (L,), dtype=torch.int32, device=dev)
        r = wu.train_step(tuple(t.to(dev) for t in syn.warmup_batch(B, L, V, 12)), coins=coins, seed=2)
        assert abs(r["loss"].item() - lnV) < 1.5
        assert torch.isfinite(wu.group.flat_p).all() and (wu.group.flat_p != b0).float().mean().item() > 0.8
        # ---- optimize (main_optimize.py:93-124): G step + D step, batch 0 updates the discriminator
        bg, bd = opt.g_group.flat_p.clone(), opt.d_group.flat_p.clone()
        critic = opt.matcher.hidden2logits.weight.clone()
        batch = tuple(t.to(dev) for t in syn.optimize_batch(B, L, V, 13))
        logs = opt.train_step(batch, 0, coins=coins, seed=3)
        for k in ("G", "STI", "BK", "D"):
            assert math.isfinite(logs[k].item()), k
        assert abs(logs["BK"].item() - lnV) < 1.5
        assert torch.isfinite(opt.g_group.flat_p).all() and torch.isfinite(opt.d_group.flat_p).all()
        assert (opt.g_group.flat_p != bg).float().mean().item() > 0.8
        assert (opt.d_group.flat_p != bd).float().mean().item() > 0.9
        assert torch.equal(opt.matcher.hidden2logits.weight, critic)               # critics stay frozen
        # the soft decode returns probabilities whose argmax is what was fed back (rnn.py:82-89)
        with torch.no_grad():
            p = opt.forward(batch[0], batch[1], 1 - batch[1], opt.tau)
        assert p.shape == (B, L, V)
        np.testing.assert_allclose(p.sum(-1).cpu().numpy(), 1.0, rtol=2e-4)
        assert torch.equal(p.argmax(-1), opt.generator.last_ids.t())
        ids = opt.transfer(batch)
        assert ids.shape == (B, L) and int(ids.min()) >= 0 and int(ids.max()) < V
        v = opt.val_loss(batch)
        assert math.isfinite(v.item())
    finally:
        set_constants(model, CONFIGS["ref"])
        del pre, wu, opt
        torch.cuda.empty_cache()


def test_book_lengths_after_noise_run_through_the_matcher():
    """Real book batches: max_len 30 (arguments.py:43), transfer_noise lengthens sentences, so the Matcher sees more than 64
    positions.  Reference widths, 2 layers, L1 = 41, L2 = 38 -> S = 79: forward + backward, bf16 and f32 agree."""
    from consistent__style_transfer_amd import model, ops
    from consistent__style_transfer_amd.model import match
    set_constants(model, CONFIGS["ref"])
    match.n_layer = 2
    try:
        torch.manual_seed(3)
        m = model.Matcher(1000).cuda().eval()
        x1 = torch.randint(4, 1000, (8, 41), device="cuda")
        x2 = torch.randint(4, 1000, (8, 38), device="cuda")
        outs = {}
        for prec in ("f32", "bf16"):
            ops.set_precision(prec)
            m.zero_grad()
            y = m(x1, x2)
            y.sum().backward()
            outs[prec] = (y.detach().cpu().numpy(), m.matcher.layers[0].self_attn.in_proj_weight.grad.detach().cpu().numpy())
        np.testing.assert_allclose(outs["bf16"][0], outs["f32"][0], rtol=3e-2, atol=3e-2)
        nf = np.linalg.norm(outs["f32"][1])
        assert np.linalg.norm(outs["bf16"][1] - outs["f32"][1]) <= 6e-2 * nf
    finally:
        ops.set_precision("bf16")
        set_constants(model, CONFIGS["ref"])


# --------------------------------------------------------------------------------- graph cache over many shapes
def _warm(name, lr=1e-3):
    from consistent__style_transfer_amd import model, stages
    c = CONFIGS[name]
    set_constants(model, c)
    wu = stages.WarmupStage(c["V"], 2, c["max_len"], lr=lr)
    _load(wu.generator, "G")
    wu = wu.cuda().eval()
    wu.setup_optim()
    return wu


def test_step_cache_many_shapes_with_evictions_matches_eager():
    """14 distinct (L', L) shapes through a cache that keeps 5 graphs: every shape is captured (no fixed number of pointer
    tables any more), shapes come back after their graph was evicted and are captured again, and the parameters follow
    the eager loop."""
    from consistent__style_transfer_amd import ops
    from consistent__style_transfer_amd.trainer import StepCache
    from oracle.detinit import det_tokens
    ops.set_precision("f32")
    name = "tiny"
    c = CONFIGS[name]
    shapes = [(lp, l) for l in (5, 6, 7) for lp in (4, 5, 6, 7, 8)][:14]
    order = shapes + shapes[:6] + shapes[::-1]               # 34 steps: revisits after evictions

    def run(graphed):
        wu = _warm(name)
        cache = StepCache(graphed, [wu], capacity=5)
        losses = []
        for it, (lp, l) in enumerate(order):
            nx, x = det_tokens(c["B"], lp, c["V"], 900 + it).cuda(), det_tokens(c["B"], l, c["V"], 950 + it).cuda()
            lab = torch.tensor([(i + it) % 2 for i in range(c["B"])], device="cuda")
            coins = torch.tensor([(it + k) % 2 for k in range(l)], dtype=torch.int32, device="cuda")
            out = cache.run("w", lambda nx, x, lab, cc: wu.train_step((nx, x, lab), coins=cc), [nx, x, lab, coins])
            losses.append(out["loss"].item())
        return np.array(losses), wu.group.flat_p.detach().clone(), cache

    le, pe, _ = run(False)
    lg, pg, cache = run(True)
    assert cache.captures >= 14 + 6 and cache.evictions >= 9 and len(cache.graphs) <= 5, (cache.captures, cache.evictions)
    np.testing.assert_allclose(lg, le, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(pg.cpu().numpy(), pe.cpu().numpy(), rtol=1e-4, atol=1e-6)
    # tables of evicted graphs went back to the group's pool: the pool did not grow with the number of captures
    assert len(_pool_of(cache)) <= 16
    ops.set_precision("bf16")


def test_step_cache_capacity_one_alternating_shapes():
    """with room for ONE graph, two alternating shapes evict each other on every step -- the evicted graph's last replay is
    only one asynchronous replay old when its pointer tables go back to the pool and into the next capture.  StepCache waits for the
    stream (and the reducer) before releasing; the parameters must follow the eager loop exactly as with a large cache."""
    from consistent__style_transfer_amd import ops
    from consistent__style_transfer_amd.trainer import StepCache
    from oracle.detinit import det_tokens
    ops.set_precision("f32")
    name = "tiny"
    c = CONFIGS[name]
    order = [(6, 5), (8, 7)] * 6

    def run(graphed):
        wu = _warm(name)
        cache = StepCache(graphed, [wu], capacity=1)
        losses = []
        for it, (lp, l) in enumerate(order):
            nx, x = det_tokens(c["B"], lp, c["V"], 700 + it).cuda(), det_tokens(c["B"], l, c["V"], 750 + it).cuda()
            lab = torch.tensor([(i + it) % 2 for i in range(c["B"])], device="cuda")
            coins = torch.tensor([(it + k) % 2 for k in range(l)], dtype=torch.int32, device="cuda")
            out = cache.run("w", lambda nx, x, lab, cc: wu.train_step((nx, x, lab), coins=cc), [nx, x, lab, coins])
            losses.append(out["loss"].clone())           # no .item(): nothing synchronises between the steps but the cache itself
        return np.array([v.item() for v in losses]), wu.group.flat_p.detach().clone(), cache

    le, pe, _ = run(False)
    lg, pg, cache = run(True)
    assert cache.evictions >= len(order) - 2 and len(cache.graphs) <= 1, (cache.captures, cache.evictions)
    np.testing.assert_allclose(lg, le, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(pg.cpu().numpy(), pe.cpu().numpy(), rtol=1e-4, atol=1e-6)
    ops.set_precision("bf16")


def _pool_of(cache):
    grp = next(iter(cache.graphs.values())).record.tables[0][0]
    return grp._free_tables


@pytest.mark.parametrize("prec", ["bf16", "f32"])
def test_eager_validation_after_graph_replays_sees_trained_weights(prec):
    """ops.weight_bf16 cached a trained weight's bf16 copy keyed on versions that graph replays did not
    move, so an eager forward after replays (validation, the warm-up pass of a new shape) read stale matrices.  Train
    through graphs, validate eagerly, train more, validate again: both validation losses must equal the all-eager run's."""
    from consistent__style_transfer_amd import ops
    from consistent__style_transfer_amd.trainer import StepCache
    ops.set_precision(prec)
    name = "b16" if prec == "bf16" else "tiny"               # b16: the shapes that take the bf16 GEMM paths
    c, G = CONFIGS[name], load_golden("curves", name)
    lr = 1e-3 if prec == "bf16" else 1e-2                 # tiny in f32: a larger step so that five batches visibly move the losses
    n = 5

    def run(graphed):
        st = make_opt(name, lr=lr)
        cache = StepCache(graphed, [st])
        vals = []
        xv, lv = cu(opt_batch(c, 50))
        for rnd in range(2):
            for k in range(n):
                it = rnd * n + k
                upd = it % 4 == 0
                x, lab = cu(opt_batch(c, it))
                coins = torch.from_numpy(np.asarray(G["optimize.coins"][it % len(G["optimize.coins"])]).astype(np.int32)).cuda()
                cache.run(("o", upd), lambda x, lab, cc: st.train_step((x, lab), 0 if upd else 1, coins=cc), [x, lab, coins])
            vals.append(st.val_loss((xv, lv)).item())                              # eager, eval mode, no_grad
            # and an eager teacher-forced forward through the TRAINED generator's bf16 weight copies
            with torch.no_grad():
                lg = st.generator(xv, lv, xv, lv, coins=[0] * xv.shape[1])
            vals.append(ops.token_ce(lg.view(-1, lg.size(-1)), xv.reshape(-1)).item())
        return np.array(vals)

    ve, vg = run(False), run(True)
    tol = 2e-3 if prec == "bf16" else 1e-4
    assert abs(ve[1] - ve[3]) > 5 * tol, "the second round of training must visibly move the teacher-forced loss"
    np.testing.assert_allclose(vg, ve, rtol=tol, atol=tol)
    ops.set_precision("bf16")


# --------------------------------------------------------------------------------- 200 real pretrain batches, graphs on
def test_pretrain_200_real_batches_in_graph_mode(tmp_path):
    from consistent__style_transfer_amd import model, ops, stages
    from consistent__style_transfer_amd.loader import GlobalBatchSampler, StyleDataset, collate_pretrain, iterate_batches, load_s2l
    from consistent__style_transfer_amd.model import match, mlm
    from consistent__style_transfer_amd.trainer import StepCache
    from consistent__style_transfer_amd.vocab import BPETokenizer
    from test_gpu_e2e import _restore_constants, _small_constants
    ops.set_precision("bf16")
    _small_constants()
    mlm.d_model = match.d_model = 64
    mlm.n_head = match.n_head = 4
    mlm.n_layer = match.n_layer = 1
    try:
        vocab = BPETokenizer.load(os.path.join(GOLD, "yelp_sample-vocab.json"), os.path.join(GOLD, "yelp_sample-merges.txt"))
        ds = StyleDataset([os.path.join(GOLD, "yelp_dev_sample.0"), os.path.join(GOLD, "yelp_dev_sample.1")], vocab, 18, load_s2l)
        torch.manual_seed(0)
        pre = stages.PretrainStage(len(vocab), 2).cuda().train()
        pre.setup_optim()
        cache = StepCache(True, [pre], capacity=12)
        sampler = GlobalBatchSampler(len(ds), 32, shuffle=True, seed=0)
        collate = collate_pretrain(vocab)
        n, shapes, last = 0, set(), None
        epoch = 0
        while n < 200:
            sampler.set_epoch(epoch)
            for _, batch in iterate_batches(ds, sampler, collate, seed=epoch):
                b = [t.cuda() for t in batch]
                shapes.add(tuple(tuple(t.shape) for t in b))
                last = cache.run("p", lambda *bb: pre.train_step(bb), b)
                n += 1
                if n >= 200:
                    break
            epoch += 1
        torch.cuda.synchronize()
        assert len(shapes) > 8, len(shapes)                    # more distinct shapes than round 1's fixed 8 pointer tables
        assert cache.captures >= len(shapes) and len(cache.graphs) <= 12
        assert all(math.isfinite(last[k].item()) for k in ("s_loss", "c_loss", "dn_loss"))
        assert all(torch.isfinite(g.flat_p).all() for g in pre.groups.values())
        assert last["dn_loss"].item() < math.log(len(vocab))   # 200 steps of training went somewhere
    finally:
        _restore_constants()


# --------------------------------------------------------------------------------- two ranks, product OptimizeStage
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out, steps, train_mode=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      CST_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from consistent__style_transfer_amd import ops
    from consistent__style_transfer_amd.parallel import GradReducer, broadcast_tensors, check_replicas, init_distributed, shard_batch
    from oracle.detinit import det_tokens
    init_distributed("gloo")
    torch.cuda.set_device(0)                                   # both ranks share the one GPU (gloo moves gradients through the host)
    ops.set_precision("f32")
    name = "tiny"
    c = CONFIGS[name]
    B = 4
    from consistent__style_transfer_amd._lib import call_plain
    st = make_opt(name, lr=1e-3)
    broadcast_tensors(st.replicated_tensors())
    reducer = GradReducer(world)
    full = make_opt(name, lr=1e-3) if rank == 0 else None
    if train_mode:                       # dropout ON everywhere (generator 0.1, critics 0.1 / 0.5, discriminator 0.25), seeded per batch
        st.train()
        if full is not None:
            full.train()
    try:
        for it in range(steps):
            x = det_tokens(B, c["L"], c["V"], 100 + it).cuda()
            lab = torch.tensor([0, 1, 1, 0], device="cuda")
            coins = torch.tensor([(it + k) % 2 for k in range(c["L"])], dtype=torch.int32, device="cuda")
            seed = 7000 + 10 * it if train_mode else None
            st.train_step(shard_batch((x, lab), rank, world), it, coins=coins, seed=seed, reducer=reducer)
            if full is not None:
                call_plain("cst_set_drop_shard", 0)            # the one-process reference run draws the global batch's masks
                full.train_step((x, lab), it, coins=coins, seed=seed)
                call_plain("cst_set_drop_shard", rank)
        check_replicas(st.replicated_tensors(), "product optimize stage")     # raises on every rank if any bit differs
        if rank == 0:
            eg = float((st.g_group.flat_p - full.g_group.flat_p).abs().max())
            ed = float((st.d_group.flat_p - full.d_group.flat_p).abs().max())
            moved = float((st.d_group.flat_p - make_opt(name).d_group.flat_p).abs().max())
            out.put((eg, ed, moved))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("train_mode", [False, True])
def test_dp2_product_optimize_stage_replicas_identical_and_match_global_batch(train_mode):
    """8 batches: 0 and 4 step the discriminator, in between its gradients accumulate and pass through both clips of every
    batch (stages.OptimizeStage.train_step reduces them every batch for that reason).
    train_mode: the same with dropout ON in every module -- each rank registers its data-parallel rank with the library
    (parallel.init_distributed -> cst_set_drop_shard), so its kernels index the masks by GLOBAL batch row and the two shards
    together draw exactly the masks of the one-process global-batch run (round-2 verdict: every rank used to draw the same
    masks for its local rows)."""
    import torch.multiprocessing as mp
    from test_dp_gloo_cpu import _collect
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, 8, train_mode)) for r in range(2)]
    for p in procs:
        p.start()
    (eg, ed, moved), = _collect(procs, q, 1, timeout=600)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert moved > 1e-3, moved                                  # the discriminator did train
    assert eg <= 1e-5 and ed <= 1e-5, (eg, ed)


# --------------------------------------------------------------------------------- the RCCL branch, one rank
def _rccl_worker(port, out):
    """A one-rank `nccl` (= RCCL) group on the one GPU: AVG over one rank is the identity, so the reduced run must equal the plain
    one -- what is exercised is everything no multi-GPU node was available for: communicator setup, async all_reduce(AVG) on slices
    of the flat gradient buffers, deferred collectives next to the bucketed backward, and segmented hipGraph capture / replay with
    RCCL's helper threads alive."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    from consistent__style_transfer_amd import model, ops, stages
    from consistent__style_transfer_amd.parallel import GradReducer, check_replicas, max_over_ranks
    from consistent__style_transfer_amd.trainer import StepCache
    from test_gpu_stages import curve_lr, pre_batch
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        ops.set_precision("f32")
        name = "tiny"
        c = CONFIGS[name]

        def build():
            set_constants(model, c)
            pre = stages.PretrainStage(c["V"], 2, lr=curve_lr(name, "pretrain"))
            for attr, which in (("classifier", "cls"), ("matcher", "mat"), ("denoiser", "dn")):
                _load(getattr(pre, attr), which)
            pre = pre.cuda().eval()
            pre.setup_optim()
            return pre

        def run(reducer, graphed, bucketed):
            pre = build()
            pre.bucketed = bucketed
            cache = StepCache(graphed, [pre], reducer)
            for it in range(5):
                cache.run("p", lambda *b, reducer=None: pre.train_step(b, reducer=reducer), list(cu(pre_batch(c, it))))
            torch.cuda.synchronize()
            return pre, cache

        red = GradReducer(1, bucket_elems=4096, force=True)        # small buckets: several collectives in flight per call
        assert red.avg
        plain, _ = run(None, False, False)
        dp, cache = run(red, True, True)
        assert all(len(g.graphs) > 1 for g in cache.graphs.values())     # split at the reduce points
        # (max deviation, largest fraction of a group's weights further apart than 2e-4): see the assertion in the test
        diffs = {k: (dp.groups[k].flat_p - plain.groups[k].flat_p).abs() for k in plain.groups}
        err = (max(float(d.max()) for d in diffs.values()), max(float((d > 2e-4).float().mean()) for d in diffs.values()))
        check_replicas([g.flat_p for g in dp.groups.values()], "one rank")
        assert max_over_ranks(1.5, torch.device("cuda")) == 1.5
        out.put(err)
    finally:
        dist.destroy_process_group()


def test_rccl_branch_single_rank_segmented_replay():
    import torch.multiprocessing as mp
    from test_dp_gloo_cpu import _collect
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    (err,) = _collect([p], q, 1, timeout=600)
    p.join(120)
    assert p.exitcode == 0
    # Adam turns rounding-level gradient differences into lr-sized steps on the few weights whose gradient is noise (the bias / embedding
    # gradients are float-atomic sums: two runs of the SAME path already differ in the last bit of a handful of gradient elements, see
    # test_pretrain_bucketed_backward_reduce_points_and_segmented_replay): at most 1e-4 of a group's weights beyond 2e-4, none beyond 2 lr
    worst, frac = err
    assert frac <= 1e-4 and worst <= 2e-3, err
