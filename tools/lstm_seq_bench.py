import os, sys
sys.path.insert(0, "/root/repo")
import torch
from consistent__style_transfer_amd import ops, gen_fn
from consistent__style_transfer_amd._lib import call
B, L, H = 256, 18, 256
whh = [torch.randn(4 * H, H, device="cuda") * 0.08 for _ in range(2)]
wb = [ops.cast_bf16(w, want_t=False)[0] for w in whh]
xp = [torch.randn(B, L * 4 * H, device="cuda") * 0.5 for _ in range(2)]
h0 = torch.randn(B, 2 * H, device="cuda") * 0.5
genc, cenc = torch.empty(2, L, B, 4 * H, device="cuda"), torch.zeros(2, L, B, H, device="cuda")
hprev, c_cat = torch.empty(2, B, L, H, device="cuda"), torch.empty(B, 2 * H, device="cuda")
mem, memb = torch.empty(B, L, 2 * H, device="cuda"), torch.zeros(B, L * 2 * H, device="cuda", dtype=torch.int16)
fo = [gen_fn._lstm_frag_order(w, H) for w in wb]
f = lambda: call("cst_lstm_seq_fwd", fo[0], fo[1], xp[0], xp[1], h0, 2 * H, genc[0], genc[1], cenc[0], cenc[1], hprev[0], hprev[1], None, None, c_cat, 2 * H, mem, memb, B, L, H)
for _ in range(3): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): f()
b.record(); torch.cuda.synchronize()
print("lstm_seq_fwd us:", a.elapsed_time(b) * 100)
from consistent__style_transfer_amd._lib import call_plain
nb = call_plain("cst_lstm_seq_xchg_bytes", B)
xchg = torch.zeros(nb, device="cuda", dtype=torch.uint8)
f2 = lambda: call("cst_lstm_seq_fwd_split", fo[0], fo[1], xp[0], xp[1], h0, 2 * H, genc[0], genc[1], cenc[0], cenc[1], hprev[0], hprev[1], None, None, c_cat, 2 * H, mem, memb, B, L, H, xchg, nb)
for _ in range(3): f2()
torch.cuda.synchronize()
a.record()
for _ in range(10): f2()
b.record(); torch.cuda.synchronize()
print("lstm_seq_fwd_split us (incl. the workspace zero fill):", a.elapsed_time(b) * 100, " timeout word:", int(xchg[-16:].view(torch.int32)[0].item()))

wt = [gen_fn._lstm_frag_order_t(ops.cast_bf16(w)[1], H) for w in whh]
dc_cat, dmem = torch.randn(B, 2 * H, device="cuda"), torch.randn(B, L * 2 * H, device="cuda")
dge, dh0 = torch.empty(2, B, L, 4 * H, device="cuda"), torch.empty(B, 2 * H, device="cuda")
dgb = torch.empty(2, B, L, 4 * H, device="cuda", dtype=torch.int16)
g = lambda: call("cst_lstm_seq_bwd", wt[0], wt[1], genc[0], genc[1], cenc[0], cenc[1], c_cat, 2 * H, dc_cat, 2 * H, dmem, dge[0], dge[1], dgb[0], dgb[1], dh0, 2 * H, B, L, H)
for _ in range(3): g()
torch.cuda.synchronize()
a.record()
for _ in range(10): g()
b.record(); torch.cuda.synchronize()
print("lstm_seq_bwd us:", a.elapsed_time(b) * 100)
