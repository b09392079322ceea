"""HIP-graph capture of whole training steps.

A stage step is ~500-1500 small kernel launches (18 sequential decode steps x 3 decodes, 4-6
encoder layers forward and backward, optimiser); launched eagerly the host cannot feed the GPU
(each launch costs 5-10 us of host time for kernels that run 2-20 us).  Every entry point of
libcst_hip.so is capture-safe (no allocation, no synchronisation), so the whole step -- forward,
backward, gradient gather, clipping, Adam -- is recorded once into a hipGraph and replayed.

Anything that varies between steps is read from device memory: the batch (static input buffers),
the scheduled-sampling coins (int32 device vector consumed by cst_embed_gather), the dropout seed
(a device word added to every call site's seed and bumped by the graph itself) and Adam's step
counter.
"""
import torch

from . import ops
from ._lib import call
from .optim import CaptureRecord, FlatGroup


class GraphedStep:
    def __init__(self, fn, example_inputs, seed_modules=(), warmup=2, reducer=None, pool=None):
        """fn(*static_inputs) -> dict of tensors.  `example_inputs`: tensors fixing shapes/dtypes.
        `seed_modules`: modules whose dropout seed must advance on every replay.

        With a `reducer` (data parallelism) fn is called as fn(*static_inputs, reducer=hook): every
        call of the hook -- the points where the stage all-reduces its flat gradient buffers -- ends the
        graph being captured and starts the next one, and on replay the collective runs eagerly on the
        stream between the two segment graphs (forward/backward/gather | all-reduce | clip/Adam ...).

        `pool`: a torch.cuda.graph_pool_handle() shared with other GraphedSteps that are never replayed concurrently
        and whose outputs are consumed before the next replay (trainer.StepCache: one graph per batch shape) -- their
        activations then share one arena instead of one private pool per captured shape."""
        dev = example_inputs[0].device
        self.static_in = [t.clone() for t in example_inputs]
        self.seed_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        for m in seed_modules:
            for sub in m.modules():
                st = getattr(sub, "_seed_state", None)
                if st is not None:
                    st.seed_dev = self.seed_dev
        self.fn = fn
        self.reducer = reducer
        self._hook = reducer                              # eager passes: the real collective
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):       # at least one eager pass (allocator + lazy init); it is a real step
                self.first_out = self._body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        ops.capture_scope_reset()
        from . import gen_fn
        gen_fn.reserve_capture_workspaces()               # the split encoder kernel's exchange buffers: never created inside a capture
        self.graphs, self.points = [], []
        self.record = CaptureRecord()
        for grp in list(FlatGroup._live):                 # pointer tables for every gather the capture will record
            grp.reserve_tables()
        FlatGroup._record = self.record
        try:
            if reducer is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    self.static_out = self._body()
                self.graphs.append(g)
            else:
                self._capture_segments(s, pool)
        finally:
            FlatGroup._record = None
            ops.capture_scope_reset()
            self._hook = reducer
        torch.cuda.synchronize()

    def release(self):
        """Drop the captured graphs and hand their gradient-pointer tables back (trainer.StepCache eviction)."""
        self.graphs, self.points, self.static_out = [], [], None
        self.record.release()

    def _capture_segments(self, side, pool=None):
        if pool is None:
            pool = torch.cuda.graph_pool_handle()
        cur = [torch.cuda.CUDAGraph()]

        def hook(groups, defer=False):
            cur[0].capture_end()
            self.graphs.append(cur[0])
            self.points.append((list(groups), defer))     # nothing has run during capture: nothing to reduce yet
            cur[0] = torch.cuda.CUDAGraph()
            cur[0].capture_begin(pool=pool, capture_error_mode="thread_local")

        self._hook = hook
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            # thread_local: RCCL's watchdog / proxy threads keep querying events while this thread captures
            cur[0].capture_begin(pool=pool, capture_error_mode="thread_local")
            self.static_out = self._body()
            cur[0].capture_end()
            self.graphs.append(cur[0])
        torch.cuda.current_stream().wait_stream(side)

    def _body(self):
        call("cst_add_i32", self.seed_dev, 7919)          # new dropout masks on every replay
        if self.reducer is None:
            return self.fn(*self.static_in)
        return self.fn(*self.static_in, reducer=self._hook)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src, non_blocking=True)
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.points):
                groups, defer = self.points[i]
                self.reducer(groups, defer) if defer else self.reducer(groups)
        # the replay ran Adam on these groups through the C ABI: nothing torch can see changed, so the caches keyed on
        # (tensor version, group version) -- ops.weight_bf16's copies used by EAGER code such as validation -- must
        # be told.  Without this an eager forward after a run of replays reads bf16 weights from before the replays.
        for grp in self.record.stepped:
            grp.version += 1
        return self.static_out
