#!/usr/bin/env python3
"""Census of the torch (aten) ops one eager bench step issues next to the library's kernels: op, shapes and the innermost
Python frame inside this repo -- the glue that shows up as fill / copy / elementwise kernels in the rocprof summary.
Usage: tools/aten_census.py [stage]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

bench.ONLY = sys.argv[1] if len(sys.argv) > 1 else None
w = bench.WORKLOADS[os.environ.get("CST_CENSUS_WORKLOAD", bench.HEADLINE)]
dev = torch.device("cuda:0")
stages_ = bench.build_stages(w, dev)
batches = bench.make_batches(w, 0, dev)
for it in range(2):
    bench.run_step(stages_, batches, it, None)
torch.cuda.synchronize()
cnt = collections.Counter()
SKIP = ("view", "reshape", "t.default", "transpose", "slice", "select", "detach", "alias", "as_strided", "unsqueeze", "squeeze",
        "expand", "permute", "_unsafe_view", "split", "unbind", "sym_", "empty", "is_", "stride", "size", "numel", "dim", "storage_offset",
        "_local_scalar", "lift_fresh", "_to_copy.default_cpu")


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(s in name for s in SKIP):
            shp = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))[:2]
            site = "engine"
            for fr in reversed(traceback.extract_stack()[:-1]):
                fn = fr.filename
                if "/repo/" in fn and not fn.endswith("aten_census.py") and "torch/" not in fn:
                    site = f"{os.path.basename(fn)}:{fr.lineno}"
                    break
            cnt[(name, shp, site)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    bench.run_step(stages_, batches, 4, None)
torch.cuda.synchronize()
tot = sum(cnt.values())
print("aten ops in one step:", tot)
for (name, shp, site), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:90]:
    print(f"{n:4d}  {name:28s} {str(shp):48s} {site}")
