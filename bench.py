#!/usr/bin/env python3
"""Throughput benchmark of the three-stage training path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One "step" = one pretrain step (TextCNN + Matcher + MLM) + one warmup step (DenoiseLSTM
auto-encoding) + one optimize step (generator step + discriminator step) on one synthetic
Yelp-shaped batch per stage (B sentences per GPU, seq <= 18, |V| = 10 000): the whole
`main_pretrain -> main_warmup -> main_optimize` hot path including backward, gradient clipping and
Adam.  `value` = sentences carried through all three stages per second, whole job (B*N / step time).
Rank 0 prints ONE JSON line (see README/DESIGN for the field contract).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: Yelp, 4-layer d=512 critics, batch 256, pretrain -> warmup -> optimize
    "yelp_4l_d512_b256": dict(n_layer=4, d_model=512, n_head=8, B=256, L=18, V=10000),
    # the reference's own constants (6-layer d=512)
    "yelp_6l_d512_b256": dict(n_layer=6, d_model=512, n_head=8, B=256, L=18, V=10000),
    # BASELINE.json configs[3] per-GPU shard: 6-layer d=768, 256 sentences per GPU
    "yelp_6l_d768_b256": dict(n_layer=6, d_model=768, n_head=8, B=256, L=18, V=10000),
    # BASELINE.json configs[2]: book corpus, 6-layer d=512, batch 512, max_len 30 (Matcher S = 60)
    "book_6l_d512_b512": dict(n_layer=6, d_model=512, n_head=8, B=512, L=30, V=10000),
    # BASELINE.json configs[0]-shaped small case
    "yelp_2l_d256_b32": dict(n_layer=2, d_model=256, n_head=8, B=32, L=16, V=10000),
    # the WHOLE global batch of configs[3] on one GPU: the strong-scaling denominator of north_star's ">= 6x at 8 GPUs on batch 2048"
    "yelp_6l_d768_b2048": dict(n_layer=6, d_model=768, n_head=8, B=2048, L=18, V=10000),
}
HEADLINE = "yelp_6l_d768_b256"   # the per-GPU shard of BASELINE configs[3] (global batch 2048 at 8 GPUs): what the 1 -> 8 curve runs
OTHER_WORKLOADS = ("yelp_4l_d512_b256", "book_6l_d512_b512", "yelp_6l_d512_b256", "yelp_6l_d768_b2048")      # configs[1], configs[2], the reference's own sizes, configs[3]'s global batch on one GPU
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0        # dense bf16 MFMA peak (not the 2:1 sparse figure)


def build_stages(w, device, lr_scale=1.0):
    from consistent__style_transfer_amd import stages
    from consistent__style_transfer_amd.model import match, mlm
    mlm.d_model = match.d_model = w["d_model"]
    mlm.n_head = match.n_head = w["n_head"]
    mlm.n_layer = match.n_layer = w["n_layer"]
    torch.manual_seed(1234)                      # identical random-init weights on every rank
    pre = stages.PretrainStage(w["V"], 2).to(device)
    wu = stages.WarmupStage(w["V"], 2, w["L"]).to(device)
    opt = stages.OptimizeStage(w["V"], 2, w["L"]).to(device)
    for s in (pre, wu, opt):
        s.train()
        s.setup_optim()
    return pre, wu, opt


def make_batches(w, rank, device, n=4):
    from consistent__style_transfer_amd import synthetic as syn
    B, L, V = w["B"], w["L"], w["V"]
    out = []
    for i in range(n):
        seed = 1000 * (rank + 1) + i                   # rank offset for data only, never for coins
        out.append((tuple(t.to(device) for t in syn.pretrain_batch(B, L, V, seed)),
                    tuple(t.to(device) for t in syn.warmup_batch(B, L, V, seed)),
                    tuple(t.to(device) for t in syn.optimize_batch(B, L, V, seed))))
    return out


def coins_for(step, L):
    import random
    r = random.Random(777 + step)                      # same sequence on every rank
    return [r.random() < 0.5 for _ in range(L)]


def coins_tensor(step, L, device):
    return torch.tensor([int(c) for c in coins_for(step, L)], dtype=torch.int32).to(device, non_blocking=True)


ONLY = None        # profiling aid (--only-stage): restrict the eager step to one stage


def run_step(stages_, batches, it, reducer):
    """Eager step (used with --no-graph, for N > 1 and for the per-kernel timing leg)."""
    pre, wu, opt = stages_
    bp, bw, bo = batches[it % len(batches)]
    L = bw[1].shape[1]
    dev = bw[1].device
    if ONLY in (None, "pretrain"):
        pre.train_step(bp, seed=3 * it, reducer=reducer)
    if ONLY in (None, "warmup"):
        wu.train_step(bw, coins=coins_tensor(2 * it, L, dev), seed=3 * it + 1, reducer=reducer)
    if ONLY in (None, "optimize"):
        opt.train_step(bo, it, coins=coins_tensor(2 * it + 1, L, dev), seed=3 * it + 2, reducer=reducer)


class GraphedPipeline:
    """The three stage steps captured as hipGraphs (the optimize stage twice: with and without
    the every-4th-batch discriminator update)."""

    def __init__(self, stages_, batches, reducer=None):
        """With a gradient reducer (N > 1) every stage step is captured as graph segments split at its
        all-reduce points; the collectives run eagerly on the stream between the segments."""
        from consistent__style_transfer_amd.graphs import GraphedStep
        pre, wu, opt = stages_
        bp, bw, bo = batches[0]
        L = bw[1].shape[1]
        dev = bw[1].device
        c0 = coins_tensor(0, L, dev)
        self.pre = GraphedStep(lambda *b, reducer=None: pre.train_step(b, reducer=reducer), list(bp), [pre], reducer=reducer)
        self.wu = GraphedStep(lambda nx, x, lab, coins, reducer=None: wu.train_step((nx, x, lab), coins=coins, reducer=reducer),
                              list(bw) + [c0], [wu], reducer=reducer)
        self.opt_d = GraphedStep(lambda x, lab, coins, reducer=None: opt.train_step((x, lab), 0, coins=coins, reducer=reducer),
                                 list(bo) + [c0], [opt], reducer=reducer)
        self.opt_nd = GraphedStep(lambda x, lab, coins, reducer=None: opt.train_step((x, lab), 1, coins=coins, reducer=reducer),
                                  list(bo) + [c0], [opt], reducer=reducer)
        self.L, self.dev = L, dev

    def step(self, batches, it):
        bp, bw, bo = batches[it % len(batches)]
        self.pre(*bp)
        self.wu(*bw, coins_tensor(2 * it, self.L, self.dev))
        (self.opt_d if it % 4 == 0 else self.opt_nd)(*bo, coins_tensor(2 * it + 1, self.L, self.dev))


def _cgroup_cpu_quota():
    """CPUs this process may use at once according to its cgroup (v2 cpu.max, v1 cfs quota), or None."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // per)
    except (OSError, ValueError):
        pass
    return None


CORE_INFO = {}          # what physical_cores() found: host cores, affinity, cgroup quota, the CST_CPU_THREADS cap, threads used


def physical_cores():
    """Host cores this job may actually use: distinct (physical id, core id) pairs of /proc/cpuinfo (SMT siblings counted
    once), capped by the CPU affinity mask, by the cgroup CPU quota and by CST_CPU_THREADS (default 16: the CPU share of a
    one-GPU box in this pool -- more threads than that share made the round-2 baseline 2-3x SLOWER, not faster)."""
    seen, phys, core = set(), None, None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if core is not None:
                    seen.add((phys, core))
                phys = core = None
    except OSError:
        pass
    n = len(seen) or (os.cpu_count() or 1)
    CORE_INFO["physical_cores_of_host"] = n
    try:
        aff = len(os.sched_getaffinity(0))
        CORE_INFO["affinity_cpus"] = aff
        n = min(n, aff)
    except (AttributeError, OSError):
        pass
    q = _cgroup_cpu_quota()
    CORE_INFO["cgroup_cpu_quota"] = q
    if q:
        n = min(n, q)
    cap = int(os.environ.get("CST_CPU_THREADS", "16"))
    CORE_INFO["thread_cap"] = cap
    n = min(n, cap)
    CORE_INFO["threads_used"] = max(1, n)
    return max(1, n)


def _cpu_stage_times(w, Bc, nthreads, warm=3, timed=10, budget_s=30.0):
    """Per-stage step times of the oracle's training loops (plain torch fp32, oracle/train.py) at batch Bc: `warm` untimed +
    `timed` timed steps per stage (fewer only if one stage alone would exceed budget_s)."""
    from consistent__style_transfer_amd import model
    from consistent__style_transfer_amd import synthetic as syn
    from consistent__style_transfer_amd.model import match, mlm
    from oracle import train as OT
    torch.set_num_threads(nthreads)
    L, V, d, nl, nh = w["L"], w["V"], w["d_model"], w["n_layer"], w["n_head"]
    torch.manual_seed(0)

    def P(shapes):
        return {k: (torch.randn(*s) * 0.05) for k, s in shapes.items()}

    # parameter shapes from the product modules (CPU construction, no kernels involved)
    keep = (mlm.d_model, mlm.n_layer, mlm.n_head)
    mlm.d_model = match.d_model = d
    mlm.n_layer = match.n_layer = nl
    mlm.n_head = match.n_head = nh
    try:
        shp = lambda m: {k: tuple(v.shape) for k, v in m.state_dict().items()}
        Pg, Pc, Pm, Pd, Pdisc = (P(shp(model.DenoiseLSTM(V, 2, L))), P(shp(model.TextCNN(V, 2))), P(shp(model.Matcher(V))),
                                 P(shp(model.MLM(V, 2))), P(shp(model.RelGAN_D(V))))
    finally:
        mlm.d_model = match.d_model = keep[0]
        mlm.n_layer = match.n_layer = keep[1]
        mlm.n_head = match.n_head = keep[2]
    for q in (Pm, Pd):
        for k in q:
            if k.endswith("norm1.weight") or k.endswith("norm2.weight"):
                q[k] = torch.ones_like(q[k])
    hp = dict(w_s=0.1, w_c=0.5, w_adv=1.0, w_bt=1.0, tau=0.1, gap=0.0)
    dc = lambda q: {k: v.detach().clone() for k, v in q.items()}
    tp = OT.OraclePretrain(Pc, Pm, Pd, nh)
    tw = OT.OracleWarmup(dc(Pg))
    to = OT.OracleOptimize(Pg, dc(Pc), dc(Pm), dc(Pd), Pdisc, hp, nh, L)
    bp, bw, bo = syn.pretrain_batch(Bc, L, V, 1), syn.warmup_batch(Bc, L, V, 1), syn.optimize_batch(Bc, L, V, 1)
    coins = coins_for(0, L)
    out = {}
    for name, fn in (("pretrain", lambda it: tp.step(bp)), ("warmup", lambda it: tw.step(bw, coins)),
                     ("optimize", lambda it: to.step(bo, it, coins))):
        for it in range(warm):
            fn(it)
        t0, n = time.time(), 0
        while n < timed:
            fn(warm + n)
            n += 1
            if time.time() - t0 > budget_s and n >= 3:
                break
        out[name] = {"ms_per_step": 1000.0 * (time.time() - t0) / n, "timed_steps": n, "warmup_steps": warm}
    return out


def cpu_baseline(w, workload):
    """The oracle's training loops (plain torch fp32 restatement of the reference's modules and stage steps, pinned to the
    reference by tests/golden) timed on this box's host cores: >= 3 warm-up + 10 timed steps PER STAGE at batch 64 on the
    headline workload, and at BASELINE configs[0] as is (2-layer d=256, B=32, L=16).  Threads = physical cores.  `value` =
    sentences carried through all three stages per second = B / (sum of the three mean step times)."""
    cores = physical_cores()
    Bc = 64
    head = _cpu_stage_times(w, Bc, cores)
    tot = sum(v["ms_per_step"] for v in head.values()) * 1e-3
    w0 = WORKLOADS["yelp_2l_d256_b32"]
    c0 = _cpu_stage_times(w0, w0["B"], cores)
    tot0 = sum(v["ms_per_step"] for v in c0.values()) * 1e-3
    return {"value": Bc / tot, "unit": "sentences/s", "cores": cores, "kind": "port", "core_accounting": dict(CORE_INFO),
            "sample": f"oracle (plain torch fp32, torch.set_num_threads({cores}): min of the host's physical cores, the affinity mask, the cgroup CPU "
                      f"quota and the CST_CPU_THREADS cap -- see core_accounting) on {workload} at batch {Bc}: "
                      f"3 warm-up + {min(v['timed_steps'] for v in head.values())} timed steps per stage (pretrain, warmup, optimize G+D)",
            "per_stage": {k: {**v, "sentences_per_s": Bc / (v["ms_per_step"] * 1e-3)} for k, v in head.items()},
            "configs0": {"workload": "yelp_2l_d256_b32", "batch": w0["B"], "value": w0["B"] / tot0, "unit": "sentences/s",
                         "per_stage": {k: {**v, "sentences_per_s": w0["B"] / (v["ms_per_step"] * 1e-3)} for k, v in c0.items()}}}


def host_path(w, pretrain_sentences_per_s):
    """the pretrain stage's host side.  collate_pretrain needs one Word Mover's Distance per sentence
    (src/loader.py:60 -> src/wmd.py:31-45); libcst_host.so solves a whole batch per call (csrc/host_wmd.cpp).  Measured here on
    Yelp-shaped synthetic pairs (two independently noised copies of a batch, 100-dimensional unit word vectors for every id):
    labels / s on one core and on all cores of this box, next to what the GPU's pretrain step consumes."""
    import numpy as np
    from consistent__style_transfer_amd import synthetic as syn
    from consistent__style_transfer_amd.wmd import WMDdistance, WordVectors
    B, L, V = w["B"], w["L"], w["V"]
    rs = np.random.RandomState(0)
    wv = WordVectors([str(i) for i in range(V)], rs.randn(V, 100))

    class Tok:
        def ids_to_tokens(self, ids):
            return [str(i) for i in ids]

        def __len__(self):
            return V

    wm, tok = WMDdistance(wv), Tok()
    bp = syn.pretrain_batch(B, L, V, 5)
    strip = lambda t: [[int(v) for v in row if v != 0] for row in t.tolist()]
    n1, n2 = strip(bp[1]), strip(bp[2])
    cores = physical_cores()
    out = {}
    for name, nt in (("one_core", 1), ("box", cores)):
        wm.cal_wmd_label(n1, n2, tok, nthreads=nt)
        t0, n = time.time(), 0
        while time.time() - t0 < 1.5:
            wm.cal_wmd_label(n1, n2, tok, nthreads=nt)
            n += B
        out[name] = n / (time.time() - t0)
    return {"labels_per_s_per_core": out["one_core"], "labels_per_s_box": out["box"], "threads": cores,
            "pretrain_consumes_sentences_per_s": pretrain_sentences_per_s,
            "cores_needed_at_this_rate": pretrain_sentences_per_s / out["one_core"] if pretrain_sentences_per_s else None,
            "note": "exact transportation solves in C++ (libcst_host.so); each data-parallel rank solves only its own rows; round 2: 281 labels/s/core (scipy LP)"}


def time_pipeline(w, device, rank, steps, warmup, reducer=None):
    """Build the three stages of workload `w`, capture their steps, time `steps` replays.  -> (ms_per_step, stages, batches, pipe)"""
    stages_ = build_stages(w, device)
    batches = make_batches(w, rank, device)
    pipe = GraphedPipeline(stages_, batches, reducer)
    for it in range(warmup):
        pipe.step(batches, it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        pipe.step(batches, warmup + it)
    torch.cuda.synchronize()
    return 1000.0 * (time.perf_counter() - t0) / steps, stages_, batches, pipe


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-stage-split", action="store_true", help="skip the per-stage timing leg")
    ap.add_argument("--only-stage", choices=["pretrain", "warmup", "optimize"], default=None,
                    help="profiling aid: with --no-graph run only this stage's step (the JSON line is then NOT the benchmark metric)")
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS))
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the `workloads` leg (BASELINE configs[1], configs[2], reference sizes)")
    ap.add_argument("--no-f32", action="store_true", help="skip the exact-fp32-mode throughput figure")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32", "fp8w"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying hipGraphs")
    ap.add_argument("--breakdown", action="store_true", help="print the per-entry-point time table to stderr")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` typed by hand (no torch.distributed.run around it): start the N ranks as CHILD processes before
        # this process has touched the GPU (nothing above initialises HIP), relay their output -- rank 0 prints the JSON line -- and
        # leave with their status.  Never an exec: a process that has initialised the GPU must not be replaced.
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"[bench] WORLD_SIZE unset: launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

    from consistent__style_transfer_amd import _lib, ops
    from consistent__style_transfer_amd.parallel import GradReducer, init_distributed, max_over_ranks
    rank, local, world = init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product path has no CPU fallback"
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    ops.set_precision(args.precision)
    global ONLY
    ONLY = args.only_stage
    if ONLY:
        args.no_graph, args.no_stage_split, args.no_roofline, args.no_cpu_baseline = True, True, True, True
        args.no_other_workloads = args.no_f32 = True
    w = WORKLOADS[args.workload]
    stages_ = build_stages(w, device)
    batches = make_batches(w, rank, device)
    reducer = GradReducer(world) if world > 1 else None
    rccl = None
    if world > 1:
        # self-check of the first real multi-GPU run: every rank reports where it sits and what it talks through; rank 0 keeps the table
        import socket
        me = {"rank": rank, "local_rank": local, "device": str(device), "gpu": torch.cuda.get_device_name(device), "host": socket.gethostname(),
              "backend": torch.distributed.get_backend(), "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if torch.distributed.get_backend() == "nccl" else None}
        table = [None] * world
        torch.distributed.all_gather_object(table, me)
        probe = torch.ones(1, device=device if me["backend"] == "nccl" else "cpu")
        torch.distributed.all_reduce(probe)                          # SUM of ones = the number of ranks the collective really reached
        rccl = {"ranks_seen": int(probe.item()), "world": world, "backend": me["backend"], "rccl_version": me["rccl_version"],
                "devices": sorted({(t["host"], t["device"]) for t in table}).__len__(), "table": table}
        print(f"[bench] rank {rank}/{world}: {me['gpu']} {me['device']} on {me['host']}, backend {me['backend']} (RCCL {me['rccl_version']}), "
              f"all_reduce reached {rccl['ranks_seen']} ranks", file=sys.stderr, flush=True)
    if world == 1 and os.environ.get("CST_RCCL_REHEARSAL"):
        # one-rank RCCL group on the one GPU: the N > 1 launch structure (hipGraph segments, bucketed backward, async all_reduce(AVG) on
        # RCCL's stream) with every collective really issued -- what it costs next to the single-graph step, minus the wire time
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)
        reducer = GradReducer(1, force=True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # The split encoder kernel (two workgroups per row group exchanging h_t inside the launch) relies on its workgroups being resident
    # together; its spins are bounded and report through a sticky word.  Probe it with one eager step before anything is captured: if a
    # workgroup ever gave up here, the one-workgroup kernel is used for the whole run (and the JSON line says so) instead of timing garbage.
    from consistent__style_transfer_amd import gen_fn
    split_ok = True
    if not args.no_graph:                                # eager runs (profiles: exact launch counts per step) are checked at the end instead
        run_step(stages_, batches, 0, reducer)           # with the run's own reducer: replicas stay identical
        torch.cuda.synchronize()
        split_ok = gen_fn.probe_split()                  # a timeout here switches the process to cst_lstm_seq_fwd (and says so)
    if os.environ.get("CST_FORCE_SEGMENTS") and reducer is None:
        reducer = lambda groups, defer=False: None     # single-GPU rehearsal of the segmented (N > 1) launch path
    use_graph = not args.no_graph
    if use_graph and (world > 1 or isinstance(reducer, GradReducer)):
        # the segmented capture (graph | eager all-reduce | graph ...) is rehearsed on one GPU only; if it cannot be
        # built on this node every rank falls back to eager launches together rather than losing the measurement
        failed = 0.0
        try:
            pipe = GraphedPipeline(stages_, batches, reducer)
        except Exception as e:                                     # noqa: BLE001
            import traceback
            print("\n" + "!" * 100 + f"\n[bench] rank {rank}: SEGMENTED hipGraph CAPTURE FAILED ({type(e).__name__}: {e});\n"
                  "[bench] every rank falls back to EAGER launches -- the number below is NOT the graph-replay number "
                  "(config.launch says 'eager').\n" + "!" * 100, file=sys.stderr, flush=True)
            traceback.print_exc()
            failed = 1.0
        use_graph = max_over_ranks(failed, device) == 0.0
    elif use_graph:
        pipe = GraphedPipeline(stages_, batches, reducer)
    if use_graph:
        step = lambda it: pipe.step(batches, it)
    else:
        step = lambda it: run_step(stages_, batches, it, reducer)
    for it in range(args.warmup):
        step(it)
    barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        step(args.warmup + it)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, device)
    ms_per_step = 1000.0 * dt / args.steps
    value = w["B"] * world * args.steps / dt

    # per-stage split of the step (the metric is reported per stage as well): the same graphs / eager
    # calls timed one stage at a time, outside the timed region above
    per_stage = None
    if not args.no_stage_split:
        from consistent__style_transfer_amd.flops import stage_gflop_per_sentence
        gf = stage_gflop_per_sentence(w["n_layer"], w["d_model"], w["L"], w["V"])
        pre, wu, opt = stages_
        L, nst = w["L"], max(4, min(args.steps, 8))

        def stage_calls(it):
            bp, bw, bo = batches[it % len(batches)]
            if use_graph:
                return {"pretrain": lambda: pipe.pre(*bp), "warmup": lambda: pipe.wu(*bw, coins_tensor(2 * it, L, device)),
                        "optimize": lambda: (pipe.opt_d if it % 4 == 0 else pipe.opt_nd)(*bo, coins_tensor(2 * it + 1, L, device))}
            return {"pretrain": lambda: pre.train_step(bp, seed=3 * it, reducer=reducer),
                    "warmup": lambda: wu.train_step(bw, coins=coins_tensor(2 * it, L, device), seed=3 * it + 1, reducer=reducer),
                    "optimize": lambda: opt.train_step(bo, it, coins=coins_tensor(2 * it + 1, L, device), seed=3 * it + 2, reducer=reducer)}

        per_stage = {}
        base = args.warmup + args.steps
        for name in ("pretrain", "warmup", "optimize"):
            barrier()
            t1 = time.perf_counter()
            for it in range(nst):
                stage_calls(base + it)[name]()
            barrier()
            ms = 1000.0 * max_over_ranks(time.perf_counter() - t1, device) / nst
            g = gf[name] if name != "optimize" else gf["optimize_g"] + gf["optimize_d"]
            per_stage[name] = {"ms_per_step": ms, "sentences_per_s": w["B"] * world / (ms * 1e-3), "algorithmic_gflop_per_sentence": g,
                               "model_tflops": g * w["B"] * world / (ms * 1e-3) / 1e3}

    roofline = None
    if rank == 0 and not args.no_roofline:
        # dominant kernel = cst_gemm_kernel (55-60 % of the step's kernel time, profiles/): its launches
        # are timed live with start/stop HIP events bound to each dispatch on the launch stream
        # (hipExtLaunchKernelGGL), over extra untimed-for-throughput steps of the same workload.
        nprof = min(3, args.steps)
        torch.cuda.synchronize()
        _lib.gemm_profile(True)
        for it in range(nprof):
            run_step(stages_, batches, args.warmup + args.steps + it, None)
        torch.cuda.synchronize()
        _lib.gemm_profile(False)
        shape_recs = _lib.gemm_profile_shapes()
        prof = _lib.gemm_profile_read()
        # HBM traffic per launch cannot be measured inside this process (PMC counters need rocprofv3 around it): it is
        # read from the committed summary of the same workload (tools/profile_pmc.sh) and the JSON line says so
        pmc, traffic_source = {}, None
        for cand in ("round4_pmc_traffic.json", "round3_pmc_traffic.json", "round2_pmc_traffic.json", "round1_pmc_traffic.json"):
            pmc_path = os.path.join(REPO, "profiles", cand)
            if os.path.exists(pmc_path):
                blob = json.load(open(pmc_path))
                if blob.get("workload", "yelp_4l_d512_b256") == args.workload:
                    pmc = blob["kernels"]
                    traffic_source = f"profiles/{cand} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, not measured in this run)"
                    break
        # the bf16 GEMM (entry points cst_gemm_bf16 / cst_gemm_bf16_tt) runs on two kernels: the big-tile ping-pong kernel (csrc/gemm_pp.hip) for
        # the shapes it wins, the LDS-DMA tile kernels for the rest and all TT products.  They are ONE family here, as the template
        # instantiations of the tile kernels always were (profiles/*kernel_stats*: the combined row); `members` keeps them apart.
        FAMILY = "cst_gemm_bf16_kernel+cst_gemm_bf16_pp_kernel"
        members = {k: prof[k] for k in ("cst_gemm_bf16_kernel", "cst_gemm_bf16_pp_kernel") if k in prof}
        prof = {k: v for k, v in prof.items() if k not in members}
        prof[FAMILY] = tuple(sum(v[i] for v in members.values()) for i in range(4))

        def entry(name, ms, flops, min_bytes, n):
            ach = flops / (ms * 1e-3) / 1e12
            ks = [pmc[m] for m in (members if name == FAMILY else (name,)) if m in pmc]
            traffic = None
            if ks:                                      # launch-weighted HBM-side bytes per launch of the family's kernels
                traffic = sum((k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]) * k["launches"] for k in ks) / sum(k["launches"] for k in ks)
            return {"achieved": ach, "frac": ach / MFMA_PEAK_TFLOPS, "launches_per_step": n / nprof,
                    "avg_launch_us": 1000.0 * ms / n, "flops_per_launch": flops / n,
                    "min_operand_bytes_per_launch": min_bytes / n, "kernel_ms_per_step": ms / nprof, "traffic": traffic}

        kernels = {name: entry(name, *v) for name, v in prof.items() if v[3] and v[0] > 0}
        if kernels:
            # the dominant kernel = the one with the most device time per step
            dom = max(kernels, key=lambda k: kernels[k]["kernel_ms_per_step"])
            d = kernels[dom]
            roofline = {"bound": "mfma", "kernel": dom, "achieved": d["achieved"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": d["frac"], "traffic": d["traffic"], "traffic_source": traffic_source if d["traffic"] is not None else None,
                        "launches_per_step": d["launches_per_step"],
                        "avg_launch_us": d["avg_launch_us"], "flops_per_launch": d["flops_per_launch"],
                        "min_operand_bytes_per_launch": d["min_operand_bytes_per_launch"],
                        "kernel_ms_per_step": d["kernel_ms_per_step"],
                        "members": {m: entry(m, *v) for m, v in members.items() if v[3] and v[0] > 0} if dom == FAMILY else None,
                        "other_mfma_kernels": {k: v for k, v in kernels.items() if k != dom}}
        if roofline is not None:
            # per-shape rates of the encoder-layer products (T = B*L tokens of the MLM, 2*B*L of the Matcher): forward / dgrad
            # (NT kernel, K > 0) and weight gradients (transposed-read kernel, recorded with K < 0)
            agg = {}
            for which, M, N, K, ms in shape_recs:
                if which not in (1, 2) or M < 1024 or ms <= 0:
                    continue
                a = agg.setdefault((M, N, K), [0, 0.0, 0])
                a[0] += 1
                a[1] += ms
                a[2] += which == 2
            by_shape = {}
            for (M, N, K), (n, ms, npp) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
                kind = "nt" if K > 0 else ("tt-group" if N == 128 else "tt")       # tt-group: the dW of one encoder layer in one launch, all outputs as rows of 128
                tf = 2.0 * M * N * abs(K) * n / (ms * 1e-3) / 1e12
                # `pp_launches_per_step`: how many of the shape's launches ran on the ping-pong kernel (csrc/gemm_pp.hip), the rest on the tile kernels
                by_shape[f"{M}x{N}x{abs(K)}.{kind}"] = {"launches_per_step": n / nprof, "pp_launches_per_step": npp / nprof, "avg_us": 1000.0 * ms / n,
                                                        "tflops": tf, "frac": tf / MFMA_PEAK_TFLOPS}
            roofline["by_shape"] = by_shape
        if args.breakdown:
            timer = _lib.KernelTimer(by_shape=True)
            _lib.set_timer(timer)
            for it in range(nprof):
                run_step(stages_, batches, args.warmup + args.steps + nprof + it, None)
            _lib.set_timer(None)
            table = timer.summary()
            tot = sum(v["ms"] for v in table.values())
            for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
                tf = f"  {v['work'] / (v['ms'] * 1e-3) / 1e12:7.1f} TF/s" if v["work"] else ""
                print(f"{k:44s} calls/step {v['calls'] / nprof:7.1f}  ms/step {v['ms'] / nprof:9.3f}  {100 * v['ms'] / tot:5.1f}%{tf}",
                      file=sys.stderr)
    if world > 1:
        torch.distributed.barrier()

    # exact-fp32 mode (v_mfma_f32_16x16x4_f32 everywhere): the mode that meets the 1e-3 loss-curve bar gets a number too
    f32_mode = None
    if rank == 0 and world == 1 and use_graph and args.precision == "bf16" and not args.no_f32:
        del pipe
        torch.cuda.empty_cache()
        ops.set_precision("f32")
        try:
            ms32, s32, b32, p32 = time_pipeline(w, device, rank, 3, 1)
            f32_mode = {"ms_per_step": ms32, "value": w["B"] / (ms32 * 1e-3), "unit": "sentences/s", "steps": 3, "warmup": 1,
                        "note": "same workload with --precision f32: every product on the exact-fp32 matrix pipe"}
            del s32, b32, p32
        finally:
            ops.set_precision(args.precision)
        torch.cuda.empty_cache()

    # BASELINE configs[4]: the same workload with fp8 (e4m3) encoder-layer weights, bf16 activations
    fp8w_mode = None
    if rank == 0 and world == 1 and use_graph and args.precision == "bf16" and not args.no_f32:
        ops.set_precision("fp8w")
        try:
            ms8, s8, b8, p8 = time_pipeline(w, device, rank, 5, 2)
            fp8w_mode = {"ms_per_step": ms8, "value": w["B"] / (ms8 * 1e-3), "unit": "sentences/s", "steps": 5, "warmup": 2,
                         "note": "same workload with --precision fp8w: QKV / out-projection / FFN weights of the critics in fp8 e4m3 with "
                                 "per-output-channel scales, widened to bf16 in registers (W8A16)"}
            del s8, b8, p8
        finally:
            ops.set_precision(args.precision)
        torch.cuda.empty_cache()

    # the other BASELINE configs, same step definition, fewer steps (outside the timed region of the headline)
    others = None
    if rank == 0 and world == 1 and use_graph and not args.no_other_workloads:
        others = {}
        for name in OTHER_WORKLOADS:
            if name == args.workload:
                continue
            wo = WORKLOADS[name]
            nst, nwu = (3, 1) if wo["B"] >= 2048 else (5, 2)
            try:
                mso, so, bo_, po = time_pipeline(wo, device, rank, nst, nwu)
                others[name] = {"ms_per_step": mso, "value": wo["B"] / (mso * 1e-3), "unit": "sentences/s", "steps": nst, "warmup": nwu,
                                "per_gpu_batch": wo["B"], "seq_len": wo["L"], "critic_layers": wo["n_layer"], "d_model": wo["d_model"]}
                del so, bo_, po
            except Exception as e:                                   # noqa: BLE001 -- a side workload must not cost the headline line
                others[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w, args.workload)
    host = None
    if rank == 0 and not args.no_cpu_baseline:
        host = host_path(w, per_stage["pretrain"]["sentences_per_s"] if per_stage else None)

    if rank == 0:
        line = {
            "metric": "train sentences/sec on Yelp-shaped batches (pretrain+warmup+optimize steps per batch)",
            "value": value, "unit": "sentences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "f32": "f32", "fp8w": "bf16 (fp8 e4m3 encoder-layer weights)"}[args.precision], "data": "synthetic",
            "config": {"workload": args.workload, "per_gpu_batch": w["B"], "global_batch": w["B"] * world,
                       "seq_len": w["L"], "vocab": w["V"], "critic_layers": w["n_layer"], "d_model": w["d_model"],
                       "parallelism": f"dp{world}", "stages": "pretrain+warmup+optimize(G+D)", "weights": "random-init",
                       "launch": ("hipGraph replay" if reducer is None else "hipGraph segments + eager all-reduce") if use_graph else "eager",
                       "rccl": rccl, "encoder_split": bool(split_ok and gen_fn.split_enabled(w["B"]))},
            "roofline": roofline, "cpu_baseline": cpu, "host_path": host, "per_stage": per_stage, "f32_mode": f32_mode, "fp8w_mode": fp8w_mode, "workloads": others,
        }
        from consistent__style_transfer_amd.gen_fn import check_exchange_timeouts
        check_exchange_timeouts()                        # the split encoder kernel's bounded spins: a timeout would have falsified the run
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
