import sys; sys.path.insert(0, "/root/repo")
import torch
from consistent__style_transfer_amd import ops
for R, C in [(512, 2048), (2048, 512), (512, 512), (1536, 512), (10000, 768), (4608, 768)]:
    x = torch.randn(R, C, device="cuda")
    for _ in range(3): ops.cast_bf16(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): ops.cast_bf16(x)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    print(R, C, "us:", a.elapsed_time(b) * 1000 / 20)
