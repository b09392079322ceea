"""Data parallelism: one process per GPU, sentence minibatches sharded by rows, gradient
all-reduce (average) over RCCL/xGMI on the flat gradient buffers of optim.FlatGroup.

The reference has no distributed code; this is new.  Contract:
  * the GLOBAL batch is built on the host exactly as collate_* does (noise functions mix tokens
    across the whole batch), padded to the global maximum length, then rows are split evenly
    (shard_batch); scheduled-sampling coins and dropout seeds are identical on every rank, and every kernel indexes its
    dropout mask by the GLOBAL batch row (init_distributed registers the rank: cst_set_drop_shard), so the shards together draw
    exactly the masks the one-process global-batch run draws;
  * each FlatGroup's gradient buffer is all-reduced in buckets; the global grad-norm clip and Adam
    then run on identical, already averaged gradients on every rank.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("CST_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            # one process per GPU; ranks may only share a device in single-GPU rehearsals (gloo)
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if world > 1:
        from ._lib import call_plain
        call_plain("cst_set_drop_shard", rank)     # shard_batch gives rank r the rows [r*n, (r+1)*n): mask indices start at r * numel_local
    return rank, local, world


def shard_batch(batch, rank, world):
    """Rows [rank*B/world, (rank+1)*B/world) of every tensor of a globally collated batch."""
    if world == 1:
        return batch
    out = []
    for t in batch:
        B = t.shape[0]
        assert B % world == 0, f"global batch {B} is not divisible by world size {world}"
        n = B // world
        out.append(t[rank * n:(rank + 1) * n].contiguous())
    return tuple(out)


class GradReducer:
    """all-reduce(average) of FlatGroup gradients, bucketed so that several collectives are in
    flight (RCCL runs them on its own stream; they overlap with the gather / sum-of-squares
    kernels of later groups)."""

    def __init__(self, world, bucket_elems=16 * 1024 * 1024, force=False):
        self.world = world
        self.force = force            # run the collectives even in a one-rank group (exercises the RCCL path on a single GPU)
        self.bucket = bucket_elems
        self.avg = dist.is_initialized() and dist.get_backend() == "nccl"
        self._pending = []

    def __call__(self, groups, defer=False):
        """defer=True (RCCL only): start the collectives and return -- they run on RCCL's stream next to whatever the
        caller launches next (the next critic's forward/backward); the next non-deferred call, or wait(), makes the
        current stream wait for everything outstanding."""
        if self.world == 1 and not self.force:
            return
        for g in groups:
            g._ssq_ready = False                          # flat_g is about to change: the gather's sums of squares no longer describe it
        if self.avg:
            for g in groups:
                flat = g.flat_g
                for s in range(0, flat.numel(), self.bucket):
                    self._pending.append(dist.all_reduce(flat[s:s + self.bucket], op=dist.ReduceOp.AVG, async_op=True))
            if not defer:
                self.wait()
            return
        # gloo (CPU tests, single-GPU rehearsals): no AVG and no device tensors -> stage through the host
        for g in groups:
            flat = g.flat_g
            host = flat.detach().cpu() if flat.is_cuda else flat
            works = [dist.all_reduce(host[s:s + self.bucket], op=dist.ReduceOp.SUM, async_op=True)
                     for s in range(0, host.numel(), self.bucket)]
            for w in works:
                w.wait()
            host.div_(self.world)
            if flat.is_cuda:
                flat.copy_(host)

    def wait(self):
        for w in self._pending:
            w.wait()
        self._pending = []


def _on_device():
    return dist.is_initialized() and dist.get_backend() == "nccl"


def broadcast_tensors(tensors, src=0):
    """Make every rank start from rank `src`'s values (parameters: flat buffers of the trained groups and the frozen
    critics' tensors).  Needed because each process builds its modules itself; identical seeds make this a no-op,
    a forgotten seed or a checkpoint loaded on one rank only is repaired instead of silently training N models."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in tensors:
        if _on_device() or not t.is_cuda:
            dist.broadcast(t, src)
        else:                                    # gloo with device tensors (single-GPU rehearsals): through the host
            h = t.detach().cpu()
            dist.broadcast(h, src)
            t.copy_(h)


def replica_checksum(tensors):
    """Order-independent exact checksum of the BIT PATTERNS of fp32 tensors (int64 sum of the int32 views): equal
    bits <=> equal checksum up to collisions, and no floating-point reduction order enters.  Diagnostic only -- this is
    not on the compute path."""
    tot = 0
    for t in tensors:
        tot += int(t.detach().reshape(-1).view(torch.int32).sum(dtype=torch.int64).item())
    return tot & 0x7FFFFFFFFFFFFFFF


def check_replicas(tensors, what=""):
    """All ranks must hold bit-identical copies of `tensors` (the data-parallel contract: identical gradients enter
    identical clips and Adam steps).  all-reduce MIN and MAX of the checksum; raise on every rank if they differ."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return True
    c = replica_checksum(tensors)
    dev = tensors[0].device if _on_device() else "cpu"
    lo = torch.tensor([c], dtype=torch.int64, device=dev)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if int(lo.item()) != int(hi.item()):
        raise RuntimeError(f"data-parallel replicas diverged{' (' + what + ')' if what else ''}: parameter checksum differs "
                           f"across ranks (this rank {c:#x}, min {int(lo.item()):#x}, max {int(hi.item()):#x})")
    return True


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return value
    on_dev = dist.get_backend() == "nccl"
    t = torch.tensor([value], dtype=torch.float64, device=device if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
