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
f = lambda: call("cst_lstm_seq_fwd", gen_fn._lstm_frag_order(wb[0], H), gen_fn._lstm_frag_order(wb[1], H), xp[0], xp[1], h0, 2 * H, genc[0], genc[1], cenc[0], cenc[1], hprev[0], hprev[1], None, None, c_cat, 2 * H, mem, memb, B, L, H)
for _ in range(3): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): f()
b.record(); torch.cuda.synchronize()
print("lstm_seq_fwd us:", a.elapsed_time(b) * 100)
