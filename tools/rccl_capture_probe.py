#!/usr/bin/env python3
"""Can an RCCL all-reduce be captured INSIDE a hipGraph on this stack (one-rank group on one GPU)?  If it can, the data-parallel step would
not have to be cut into graph segments at every reduce point (graphs.GraphedStep._capture_segments).    python tools/rccl_capture_probe.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
x = torch.arange(1 << 20, device=dev, dtype=torch.float32)
y = torch.zeros_like(x)
dist.all_reduce(x.clone(), op=dist.ReduceOp.AVG)          # communicator set-up outside any capture
torch.cuda.synchronize()
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            y.copy_(x)
            y.mul_(2.0)
            works = [dist.all_reduce(y[i * (1 << 18):(i + 1) * (1 << 18)], op=dist.ReduceOp.AVG, async_op=True) for i in range(4)]
            for w in works:
                w.wait()
            y.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    for k in range(3):
        x.fill_(float(k))
        g.replay()
        torch.cuda.synchronize()
        assert torch.all(y == 2.0 * k + 1.0), (k, y[:4])
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    print(f"captured: replay of copy + scale + 4 async all-reduces + add = {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per replay; results correct")
except Exception as e:                                    # noqa: BLE001
    print("capture of an RCCL collective failed:", type(e).__name__, str(e)[:300])
dist.destroy_process_group()
