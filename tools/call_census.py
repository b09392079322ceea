#!/usr/bin/env python3
"""Census of one bench step's calls into libcst_hip.so: per entry point and argument shape, with the Python call
site (innermost frame outside ops.py/_lib.py).  Usage: tools/call_census.py [entry-point substring] [workload]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from consistent__style_transfer_amd import _lib

pat = sys.argv[1] if len(sys.argv) > 1 else ""
w = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else bench.HEADLINE]
dev = torch.device("cuda:0")
stages_ = bench.build_stages(w, dev)
batches = bench.make_batches(w, 0, dev)
for it in range(2):
    bench.run_step(stages_, batches, it, None)
torch.cuda.synchronize()
orig = _lib.call
cnt = collections.Counter()

def spy(name, *args):
    if pat in name:
        shp = tuple(a for a in args if isinstance(a, int) and not isinstance(a, bool))[:6]
        site = "?"
        for fr in reversed(traceback.extract_stack()[:-1]):
            fn = os.path.basename(fr.filename)
            if fn not in ("ops.py", "_lib.py", "call_census.py"):
                site = f"{fn}:{fr.lineno}"
                break
        inner = [f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in traceback.extract_stack()[:-1] if os.path.basename(fr.filename) == "ops.py"]
        cnt[(name, shp, site, inner[-1] if inner else "")] += 1
    return orig(name, *args)

for m in list(sys.modules.values()):
    if m is not None and getattr(m, "call", None) is orig:
        m.call = spy
bench.run_step(stages_, batches, 4, None)
torch.cuda.synchronize()
for (name, shp, site, inner), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:24s} {str(shp):40s} {site:28s} {inner}")
