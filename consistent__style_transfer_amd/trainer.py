"""The build's own training loop, restating the pytorch_lightning 0.6/0.7 Trainer semantics the
reference relies on (call sites main_pretrain.py:132-158,
main_warmup.py:96-131, main_optimize.py:201-255) as an explicit contract:

  * a 5-batch sanity validation runs BEFORE training and its `validation_end` side effects
    (checkpoint writes, best-loss bookkeeping, freeze flags) fire once at start;
  * each epoch: every training batch goes through the stage's `train_step` (optimizer order,
    requires_grad toggling, clipping over all live gradients, D-every-4th-batch: stages.py);
    scalars are logged every 10 steps; then the whole validation set is evaluated in eval() /
    no_grad and `validation_end` decides checkpoints;
  * EarlyStopping(monitor='val_loss', mode='min', min_delta=0): stop once `patience` consecutive
    validations did not improve on the best value;
  * no optimizer state is checkpointed (checkpoint_callback=False everywhere); model weights are
    saved as plain state_dicts under the reference's file names.

Steps are replayed as hipGraphs on one GPU (graphs.GraphedStep); with more than one rank every
process builds the same GLOBAL batch, takes its rows (parallel.shard_batch) and the flat gradient
buffers are averaged over RCCL before clipping (parallel.GradReducer).
"""
import csv
import json
import os
import random
import time

import torch

from .graphs import GraphedStep
from .loader import GlobalBatchSampler, PrefetchBatches, iterate_batches
from .parallel import GradReducer, broadcast_tensors, check_replicas, init_distributed, shard_batch


class ScalarLogger:
    """log_dir/<name>/version_<k>/{meta_tags.csv, metrics.jsonl} -- the hparams table and the scalar
    stream the reference sends to TensorBoard / test-tube (Metrics/logging)."""

    def __init__(self, save_dir, name, hparams, version=None, enabled=True):
        self.enabled = enabled
        if not enabled:
            return
        root = os.path.join(save_dir, name)
        os.makedirs(root, exist_ok=True)
        if version is None or version < 0:
            existing = [int(d.split("_")[1]) for d in os.listdir(root) if d.startswith("version_") and d.split("_")[1].isdigit()]
            version = max(existing) + 1 if existing else 0
        self.dir = os.path.join(root, f"version_{version}")
        os.makedirs(self.dir, exist_ok=True)
        with open(os.path.join(self.dir, "meta_tags.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["key", "value"])
            for k, v in sorted(vars(hparams).items()):
                w.writerow([k, v])
        self.f = open(os.path.join(self.dir, "metrics.jsonl"), "a")

    def log(self, step, scalars):
        if self.enabled:
            self.f.write(json.dumps({"step": step, "time": time.time(), **scalars}) + "\n")
            self.f.flush()


class EarlyStopping:
    def __init__(self, patience):
        self.patience, self.best, self.wait = patience, float("inf"), 0

    def should_stop(self, value):
        if value < self.best:
            self.best, self.wait = value, 0
            return False
        self.wait += 1
        return self.wait >= self.patience


def _to_float(v):
    if isinstance(v, torch.Tensor):
        return float(v.detach().float().mean().item())
    return None if v is None else float(v)


class Trainer:
    def __init__(self, args, patience, log_name, sanity_batches=5):
        self.args = args
        self.rank, self.local, self.world = init_distributed()
        self.device = torch.device("cuda", self.local % torch.cuda.device_count() if self.world > 1
                                   else int(str(args.device).split(",")[0] or 0))
        torch.cuda.set_device(self.device)
        self.early = EarlyStopping(patience)
        self.logger = ScalarLogger(args.log_dir, log_name, args, getattr(args, "restore_version", None), enabled=self.rank == 0)
        self.reducer = GradReducer(self.world) if self.world > 1 else None
        self.use_graph = not getattr(args, "no_graph", False)      # world > 1: graph segments around the eager all-reduces
        self.sanity_batches = sanity_batches
        self.global_step = 0
        self.current_epoch = 0
        self.replica_check_every = int(getattr(args, "replica_check_every", 200) or 0)

    def sync_replicas(self, stage):
        """Called once after stage.setup_optim(): rank 0's parameters everywhere, then verified."""
        if self.world > 1:
            broadcast_tensors(stage.replicated_tensors())
            check_replicas(stage.replicated_tensors(), "after the initial broadcast")

    def put(self, batch):
        batch = shard_batch(batch, self.rank, self.world)
        return tuple(t.to(self.device, non_blocking=True) for t in batch)

    def coins(self, n):
        """One scheduled-sampling draw per decode step for the whole (global) batch, identical on
        every rank (rnn.py:91 draws random.random() < 1/2 per step)."""
        r = random.Random(self.args.seed * 1000003 + self.global_step)
        return torch.tensor([int(r.random() < 0.5) for _ in range(n)], dtype=torch.int32).to(self.device, non_blocking=True)

    def mean_over_ranks(self, value):
        if self.world == 1:
            return value
        on_dev = torch.distributed.get_backend() == "nccl"
        t = torch.tensor([value], dtype=torch.float64, device=self.device if on_dev else "cpu")
        torch.distributed.all_reduce(t)
        return float(t.item()) / self.world

    def fit(self, stage, train_ds, val_ds, collate, batch_size):
        """`stage` provides: train_batch(trainer, batch, batch_idx) -> dict of scalars,
        validation_step(trainer, batch) -> float or tuple, validation_end(trainer, outputs) -> val_loss."""
        args = self.args
        val_sampler = GlobalBatchSampler(len(val_ds), batch_size, shuffle=False, world=self.world)
        train_sampler = GlobalBatchSampler(len(train_ds), batch_size, shuffle=True, seed=args.seed, world=self.world)

        # validation never reads the label cache of the seeded TRAINING stream (loader.PretrainCollate.without_cache)
        val_collate = collate.without_cache() if hasattr(collate, "without_cache") else collate

        def validate(limit=None):
            stage.eval()
            outs = []
            with torch.no_grad():
                for bi, batch in iterate_batches(val_ds, val_sampler, val_collate, seed=args.seed + 99):
                    if limit is not None and bi >= limit:
                        break
                    outs.append(stage.validation_step(self, self.put(batch)))
            stage.train()
            return outs

        from . import gen_fn
        outs = validate(limit=self.sanity_batches)              # sanity check, side effects included (validation_end below)
        if not gen_fn.probe_split():
            # the sanity pass is also the split encoder kernel's residency probe (eager, before anything is captured): a workgroup that gave
            # up poisoned its rows with NaN, so the pass is repeated on the one-workgroup kernel before its side effects (checkpoints) fire
            outs = validate(limit=self.sanity_batches)
        gen_fn.check_exchange_timeouts()
        stage.validation_end(self, outs)
        t_last, n_last = time.time(), 0
        done = False
        workers = int(getattr(args, "prefetch_workers", 0) or 0)
        prefetch = PrefetchBatches(train_ds, train_sampler, collate, seed=args.seed, workers=workers) if workers > 0 else None
        for epoch in range(args.epochs):
            self.current_epoch = epoch
            train_sampler.set_epoch(epoch)
            for bi, batch in (prefetch if prefetch is not None else iterate_batches(train_ds, train_sampler, collate, seed=args.seed)):
                scalars = stage.train_batch(self, self.put(batch), bi)
                n_last += batch[0].shape[0]
                self.global_step += 1
                if self.global_step % 10 == 0:
                    vals = {k: _to_float(v) for k, v in scalars.items() if v is not None}
                    gen_fn.check_exchange_timeouts()             # the readback above synchronised: a timed-out (NaN) step stops here
                    now = time.time()
                    vals["sentences_per_sec"] = n_last / max(now - t_last, 1e-9)
                    t_last, n_last = now, 0
                    self.logger.log(self.global_step, vals)
                if self.world > 1 and self.replica_check_every and self.global_step % self.replica_check_every == 0:
                    check_replicas(stage.replicated_tensors(), f"step {self.global_step}")
                if args.max_steps is not None and self.global_step >= args.max_steps:
                    done = True
                    break
            if self.world > 1:
                check_replicas(stage.replicated_tensors(), f"end of epoch {epoch}")
            outs = validate(limit=args.val_batches)
            gen_fn.check_exchange_timeouts()                     # before validation_end saves a checkpoint
            val_loss = stage.validation_end(self, outs)
            self.logger.log(self.global_step, {"val_loss": val_loss, "epoch": epoch})
            if self.rank == 0:
                print(f"epoch {epoch}: val_loss {val_loss:.6f}", flush=True)
            if done or self.early.should_stop(val_loss):
                break
        if prefetch is not None:
            prefetch.close()
        torch.cuda.synchronize()


class StepCache:
    """hipGraph replay of a stage's train_step, re-captured whenever the static key changes (batch
    shapes; the pretrain freeze flags; the optimize stage's D-update variant).

    Real batches are padded to the per-batch maximum and `transfer_noise` moves tokens between sentences
    (loader.py:46-70), so the shape key keeps changing: 200 pretrain batches of the Yelp dev sample give ~30 distinct
    (x, nx_1, nx_2, nx_3) shapes.  The cache is therefore unbounded in the number of captures over time (pointer
    tables are allocated on demand, optim.FlatGroup.reserve_tables) but bounded in what it keeps: at most `capacity`
    graphs, least recently used evicted, all of them captured into ONE shared memory pool (they are never replayed
    concurrently and their outputs are read before the next replay), so resident memory is the largest graph's
    activations, not the sum.  Batches are never padded to a fixed length: PAD positions are part of every loss
    (main_pretrain.py:73)."""

    def __init__(self, enabled, seed_modules, reducer=None, capacity=None):
        """`reducer` (world > 1): fn must accept a `reducer=` keyword; see graphs.GraphedStep."""
        from collections import OrderedDict
        self.enabled, self.seed_modules, self.reducer = enabled, seed_modules, reducer
        self.graphs = OrderedDict()
        self.capacity = int(os.environ.get("CST_GRAPH_CACHE", "32")) if capacity is None else capacity
        self.pool = None
        self.captures = self.evictions = 0

    def run(self, key, fn, inputs):
        if not self.enabled:
            return fn(*inputs) if self.reducer is None else fn(*inputs, reducer=self.reducer)
        k = (key,) + tuple((tuple(t.shape), t.dtype) for t in inputs)
        g = self.graphs.get(k)
        if g is None:
            while len(self.graphs) >= max(1, self.capacity):
                _, old = self.graphs.popitem(last=False)
                # replays are asynchronous: the evicted graph's last replay (and collectives it deferred) may still be reading its
                # pinned pointer tables, which release() hands back to the pool for the next capture
                if self.reducer is not None and hasattr(self.reducer, "wait"):
                    self.reducer.wait()
                torch.cuda.current_stream().synchronize()
                old.release()
                self.evictions += 1
            if self.pool is None:
                self.pool = torch.cuda.graph_pool_handle()
            # construction runs the step once eagerly on these inputs (a real training step), then captures
            g = self.graphs[k] = GraphedStep(fn, list(inputs), self.seed_modules, warmup=1, reducer=self.reducer, pool=self.pool)
            self.captures += 1
            return g.first_out
        self.graphs.move_to_end(k)
        return g(*inputs)
