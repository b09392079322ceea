#!/usr/bin/env python3
"""Which Python call sites issue device copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer) in one eager stage step?  Wraps
torch.Tensor.copy_ / clone / contiguous / to and torch.cuda memcpy entry points are not visible from Python, so this uses torch.profiler:
CPU ops whose children launch a Memcpy, grouped by op name, shapes and the innermost frame inside this repo.  Usage: tools/copy_census.py [stage]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
import bench

bench.ONLY = sys.argv[1] if len(sys.argv) > 1 else "pretrain"
w = bench.WORKLOADS[bench.HEADLINE]
dev = torch.device("cuda:0")
stages_ = bench.build_stages(w, dev)
batches = bench.make_batches(w, 0, dev)
for it in range(2):
    bench.run_step(stages_, batches, it, None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    bench.run_step(stages_, batches, 2, None)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if "memcpy" in ev.name.lower() or "copyBuffer" in ev.name:
        cnt[("device-side", ev.name[:60], "")] += 1
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::to"):
        site = ""
        for fr in (ev.stack or []):
            if "consistent__style_transfer_amd" in fr or "bench.py" in fr:
                site = fr.split("/")[-1][:70]
                break
        cnt[(ev.name, str(ev.input_shapes)[:60], site)] += 1
for (n, s, site), c in cnt.most_common(40):
    print(f"{c:5d}  {n:18s} {s:62s} {site}")
