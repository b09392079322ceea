"""Flat-buffer Adam + global-norm clipping on the device (no host synchronisation).

Restates what the reference gets from torch.optim.Adam (main_pretrain.py:61-64, main_warmup.py:41-43,
main_optimize.py:73-76) and Trainer(gradient_clip_val=...) -> torch.nn.utils.clip_grad_norm_
(main_pretrain.py:139, main_warmup.py:103, main_optimize.py:211).

All parameters of a group live in ONE contiguous fp32 buffer (each nn.Parameter is a view into
it) and so do their gradients, first and second moments.  One kernel launch each gathers the
per-parameter gradients autograd produced (cst_multi_accumulate), sums squares, scales by the clip
coefficient and applies Adam.  The flat gradient buffer is also what the data-parallel
all-reduce operates on (parallel.py).
"""
import os
import weakref

import torch

from ._lib import call

_FUSED_CLIP = os.environ.get("CST_FUSED_CLIP", "1") != "0"       # A/B switch: global-norm clip folded into the Adam pass

MT_CHUNK = 4096


class CaptureRecord:
    """What a hipGraph capture took from / did to the FlatGroups it touched: the gradient-pointer tables it baked in
    (returned to their group when the graph is dropped) and the groups whose Adam step it contains (their `version`
    has to move on every replay, see graphs.GraphedStep.__call__)."""

    def __init__(self):
        self.tables = []            # (group, (pinned host table, device table))
        self.stepped = []           # groups, in first-step order, without duplicates

    def release(self):
        for grp, tab in self.tables:
            grp._free_tables.append(tab)
        self.tables = []


class FlatSlice:
    """A contiguous range of a group's flat gradient buffer: the unit parallel.GradReducer all-reduces when the backward pass
    hands over gradients bucket by bucket."""

    def __init__(self, group, lo, hi, tag=""):
        self.group, self.lo, self.hi, self.tag = group, lo, hi, tag
        self.flat_g = group.flat_g[lo:hi]


class FlatGroup:
    _live = weakref.WeakSet()       # every group alive in this process (graphs.GraphedStep tops up their table pools)
    _record = None                  # the CaptureRecord of the capture in progress, if any

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        assert self.params, "empty parameter group"
        dev = self.params[0].device
        self.lr, self.betas, self.eps = lr, betas, eps
        sizes = [p.numel() for p in self.params]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 3) // 4 * 4                      # keep every tensor 16-byte aligned
        self.total = tot
        self.offsets, self.sizes = offs, sizes
        self.flat_p = torch.zeros(tot, device=dev, dtype=torch.float32)
        for p, o, n in zip(self.params, offs, sizes):
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view_as(p.data)
        self.flat_g = torch.zeros(tot, device=dev, dtype=torch.float32)
        self.m = torch.zeros(tot, device=dev, dtype=torch.float32)
        self.v = torch.zeros(tot, device=dev, dtype=torch.float32)
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self._ssq_ready = False                           # _gather_ssq holds the per-chunk sums of squares of flat_g as it is NOW
        self.has_grad = False                             # flat_g holds gradients not yet consumed by a step
        self.direct = False                               # ops may write weight gradients straight into grad_view() (bucketed backward only)
        self.version = 0                                  # bumped by every step(): invalidates cached bf16 weight copies
        for p in self.params:
            p._cst_group = self
        ct, cs = [], []
        for t, n in enumerate(sizes):
            for s in range(0, n, MT_CHUNK):
                ct.append(t)
                cs.append(s)
        self.nchunks = len(ct)
        self._gather_ssq = torch.zeros(self.nchunks, device=dev, dtype=torch.float32)     # written by every gather_grads (cst_multi_accumulate)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=dev)
        self.chunk_start = torch.tensor(cs, dtype=torch.int64, device=dev)
        self.dst_off = torch.tensor(offs, dtype=torch.int64, device=dev)
        self.sizes_dev = torch.tensor(sizes, dtype=torch.int64, device=dev)
        self._src_host = torch.zeros(len(sizes), dtype=torch.int64).pin_memory() if dev.type == "cuda" else None
        self.srcs = torch.zeros(len(sizes), dtype=torch.int64, device=dev)
        self._null_srcs = torch.zeros(len(sizes), dtype=torch.int64, device=dev)   # "no gradient for any tensor": sumsq_into's read-only pass
        self._keep = None
        # gradient-pointer tables for hipGraph captures: a captured graph re-reads its pinned table on every replay, so
        # every capture owns the tables it used.  They are allocated here and in eager calls only (never while a
        # stream is capturing) and there is no fixed number of them: reserve_tables() tops the free pool up.
        self._free_tables = []
        self._eager_gathers = 0                           # gather_grads calls since the last reserve_tables()
        self.reserve_tables()
        FlatGroup._live.add(self)

    def _new_table(self):
        dev = self.flat_p.device
        n = len(self.sizes)
        return (torch.zeros(n, dtype=torch.int64).pin_memory() if dev.type == "cuda" else torch.zeros(n, dtype=torch.int64),
                torch.zeros(n, dtype=torch.int64, device=dev))

    def reserve_tables(self, n=None):
        """Make sure the next capture finds a free pointer table for every gather_grads call it will record: twice the
        number of calls the eager pass in front of it made (at least 4).  Must not run while a stream is capturing."""
        if n is None:
            n = max(4, 2 * self._eager_gathers)
        self._eager_gathers = 0
        while len(self._free_tables) < n:
            self._free_tables.append(self._new_table())

    def gather_grads(self, accumulate):
        """Move p.grad of every parameter into the flat gradient buffer (+= when `accumulate`, the
        reference's behaviour while an optimizer has not called zero_grad), then drop p.grad."""
        grads = [p.grad for p in self.params]
        if not accumulate:
            if any(g is None for g in grads):
                call("cst_zero", self.flat_g, 4 * self.total)
        src_host, srcs = self._src_host, self.srcs
        if self.flat_p.is_cuda and torch.cuda.is_current_stream_capturing():
            # a captured graph re-reads the pinned table on every replay: each capture needs its own
            # (two graphs over the same group -- e.g. the optimize stage's D-update / no-update
            # variants -- must not see each other's gradient addresses)
            if not self._free_tables:
                raise RuntimeError("FlatGroup.gather_grads: no free gradient-pointer table while capturing -- the capture "
                                   "was not preceded by FlatGroup.reserve_tables() (graphs.GraphedStep does that)")
            tab = self._free_tables.pop()                 # pre-allocated: no host allocation while capturing
            src_host, srcs = tab
            if FlatGroup._record is not None:
                FlatGroup._record.tables.append((self, tab))
        else:
            self._eager_gathers += 1
        for i, g in enumerate(grads):
            if g is not None and not g.is_contiguous():
                grads[i] = g.contiguous()
            src_host[i] = grads[i].data_ptr() if grads[i] is not None else 0
        srcs.copy_(src_host, non_blocking=True)
        self._keep = grads                                 # keep sources alive until the kernel ran
        call("cst_multi_accumulate", srcs, self.dst_off, self.sizes_dev, self.chunk_tensor, self.chunk_start,
             self.nchunks, self.flat_g, int(accumulate), self._gather_ssq)
        # the gather has just seen every element of flat_g: its per-chunk sums of squares stand until something else writes flat_g
        # (an all-reduce, an in-place clip, a direct-slot write, zero_grad) -- parallel.GradReducer and the methods below reset the flag
        self._ssq_ready = not self.direct
        for p in self.params:
            p.grad = None
        self.has_grad = True

    def _index_of(self, p):
        idx = getattr(self, "_pindex", None)
        if idx is None:
            idx = self._pindex = {id(q): k for k, q in enumerate(self.params)}
        return idx[id(p)]

    def grad_view(self, p):
        """The slot of parameter p in the flat gradient buffer, shaped like p."""
        i = self._index_of(p)
        return self.flat_g[self.offsets[i]:self.offsets[i] + self.sizes[i]].view(p.shape)

    def put_grads(self, params, grads):
        """Bucketed backward (stages.bucketed_backward): the gradients of `params`, just returned by torch.autograd.grad, go
        into their flat slots.  A gradient that a kernel already wrote THERE (ops: weight gradients of the encoder layers
        take their slot as the GEMM's output while `direct` is on) costs nothing; the rest (biases, LayerNorm
        parameters, embeddings) move with one multi-tensor copy.  flat_g is zero when a step starts (zero_grad), every
        parameter is produced once per step on this path, so slots are written, not accumulated."""
        dst, src = [], []
        for p, g in zip(params, grads):
            if g is None:
                continue
            v = self.grad_view(p)
            if g.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(g.reshape(p.shape))
        if dst:
            torch._foreach_copy_(dst, src)
        self.has_grad = True
        self._ssq_ready = False

    def span(self, params):
        """(lo, hi) of the flat range covered by `params`, which must be adjacent in the group's order."""
        ks = sorted(self._index_of(p) for p in params)
        assert ks == list(range(ks[0], ks[-1] + 1)), "bucket parameters must be adjacent in the flat order"
        hi = self.offsets[ks[-1] + 1] if ks[-1] + 1 < len(self.offsets) else self.total
        return self.offsets[ks[0]], hi

    def sumsq_into(self, out):
        if not self._ssq_ready:
            # the same launch with no sources: it reads every chunk of flat_g and leaves the same per-chunk partials in the same
            # summation order, so the norm does not depend on which path produced them (bit for bit: the clip coefficient scales every
            # parameter, and Adam turns a last-bit difference of it into lr-sized differences)
            call("cst_multi_accumulate", self._null_srcs, self.dst_off, self.sizes_dev, self.chunk_tensor, self.chunk_start,
                 self.nchunks, self.flat_g, 1, self._gather_ssq)
            self._ssq_ready = True
        call("cst_sumsq_partials", self._gather_ssq, self.nchunks, out)          # added in index order by one block

    def clip(self, sumsq, max_norm):
        self._ssq_ready = False                           # scaled in place (by a coefficient that lives on the device)
        call("cst_clip_scale", self.flat_g, self.total, sumsq, float(max_norm))

    def step(self):
        self.version += 1
        rec = FlatGroup._record
        if rec is not None and not any(g is self for g in rec.stepped):
            rec.stepped.append(self)
        call("cst_add_i32", self.step_dev, 1)
        pend, self._pending_clip = getattr(self, "_pending_clip", None), None
        if pend is not None:
            # the clip that preceded this step was deferred to it (clip_groups(stepping=...)): one pass instead of two
            call("cst_adam_step_clipped", self.flat_p, self.flat_g, self.m, self.v, self.total, float(self.lr),
                 float(self.betas[0]), float(self.betas[1]), float(self.eps), self.step_dev, pend[0], float(pend[1]))
            return
        call("cst_adam_step", self.flat_p, self.flat_g, self.m, self.v, self.total, float(self.lr),
             float(self.betas[0]), float(self.betas[1]), float(self.eps), self.step_dev)

    def zero_grad(self):
        call("cst_zero", self.flat_g, 4 * self.total)      # a kernel, not a memset node (cst_common.h)
        self.has_grad = False
        self._ssq_ready = False

    def grad_of(self, p):
        i = next(k for k, q in enumerate(self.params) if q is p)
        return self.flat_g[self.offsets[i]:self.offsets[i] + self.sizes[i]].view_as(p)


def clip_groups(groups, max_norm, scratch, stepping=()):
    """clip_grad_norm_ over every group that currently holds gradients (one global norm).  Groups listed in `stepping` take their
    optimizer step right after this call and zero their gradients after it: their scaling is folded into that step
    (cst_adam_step_clipped reads g * coef, the value the in-place scaling would have stored) instead of a separate pass over the
    gradient buffer.  `scratch` must stay untouched until those steps have been issued."""
    live = [g for g in groups if g.has_grad]
    if not live or max_norm is None or max_norm <= 0:
        return
    call("cst_zero", scratch, 4 * scratch.numel())
    for g in live:
        g.sumsq_into(scratch)
    for g in live:
        if _FUSED_CLIP and any(g is s for s in stepping):
            g._pending_clip = (scratch, max_norm)
        else:
            g.clip(scratch, max_norm)
