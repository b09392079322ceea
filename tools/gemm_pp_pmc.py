#!/usr/bin/env python3
"""A few launches of the ping-pong GEMM and of the tile kernels on the encoder-layer shapes for rocprofv3 --pmc (tools/profile_gemm_pp_pmc.sh),
and (`summary <dir>`) the per-kernel, per-shape averages of those passes with the DERIVED figures the counters are collected for:

  MFMA busy %  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)      (busy cycles summed over the chip's SIMDs)
  HBM-side bytes = 2 x FETCH_SIZE KiB (gfx950 counts a wide coalesced read at half its bytes: MI355X_MICROARCH.md, HBM) + WRITE_SIZE KiB
  L2 hit rate  = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
"""
import collections
import csv
import os
import sys

SHAPES = [(9216, 2048, 768), (9216, 768, 2048), (9216, 2304, 768), (4608, 2048, 768), (4096, 4096, 4096)]
PP = {(9216, 2048, 768): 504, (9216, 768, 2048): 503, (9216, 2304, 768): 703, (4608, 2048, 768): 504, (4096, 4096, 4096): 804}


def run():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from consistent__style_transfer_amd import ops
    for M, N, K in SHAPES:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        Ab, Bb = ops.cast_bf16(A, want_t=False)[0], ops.cast_bf16(B, want_t=False)[0]
        Cb = torch.empty(M, N, device="cuda", dtype=torch.int16)
        for _ in range(4):
            ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb, tile=999)
            ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb, tile=1000 + PP[(M, N, K)])
        torch.cuda.synchronize()
    # the grouped weight-gradient launch of one d = 768 encoder layer (four TT products, 336 tiles, whole-K workgroups)
    d, F = 768, 2048
    for T in GROUP_T:
        bf = lambda r, c: torch.randn(r, c, device="cuda").to(torch.bfloat16).view(torch.int16)
        dfb, wh, dhb, wy1, dob, watt, dqb, wx = bf(T, d), bf(T, F), bf(T, F), bf(T, d), bf(T, d), bf(T, d), bf(T, 3 * d), bf(T, d)
        outs = [torch.empty(d, F, device="cuda"), torch.empty(F, d, device="cuda"), torch.empty(d, d, device="cuda"), torch.empty(3 * d, d, device="cuda")]
        for _ in range(4):
            with ops.tt_group():
                ops.gemm_bf16_tt(dfb, wh, d, F, C=outs[0])
                ops.gemm_bf16_tt(dhb, wy1, F, d, C=outs[1])
                ops.gemm_bf16_tt(dob, watt, d, d, C=outs[2])
                ops.gemm_bf16_tt(dqb, wx, 3 * d, d, C=outs[3])
        torch.cuda.synchronize()


GROUP_T = (9216, 4608)


def summary(out):
    # a kernel's launches appear in the order of run(): per shape 4 x (tile kernel, ping-pong kernel); cast kernels are skipped by name
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in ("sq", "mem1", "mem2"):
        allrows = [r for r in csv.DictReader(open(f"{out}/{f}.csv")) if "cst_gemm_bf16" in r["Kernel_Name"]]
        rows = [r for r in allrows if "tt_group" not in r["Kernel_Name"]]
        grp = collections.OrderedDict()
        for r in allrows:
            if "tt_group" in r["Kernel_Name"]:
                grp.setdefault(r["Dispatch_Id"], []).append(r)
        for i, (_, rs) in enumerate(grp.items()):                       # 4 launches per token count, in the order of run()
            key = (("group", GROUP_T[min(i // 4, len(GROUP_T) - 1)]), "tt")
            for r in rs:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[key]["dur_us"].append((int(rs[0]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])) / 1e3)
        by_disp = collections.OrderedDict()
        for r in rows:
            by_disp.setdefault(r["Dispatch_Id"], []).append(r)
        for i, (_, rs) in enumerate(by_disp.items()):
            shape = SHAPES[i // 8]
            kern = "pp" if "pp_kernel" in rs[0]["Kernel_Name"] else "tile"
            for r in rs:
                agg[(shape, kern)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[(shape, kern)]["dur_us"].append((int(rs[0]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])) / 1e3)
    print("# rocprofv3 --pmc passes (SQ | FETCH_SIZE, GRBM_GUI_ACTIVE | WRITE_SIZE, TCC_HIT, TCC_MISS), per launch, bf16 output; pp = cst_gemm_bf16_pp_kernel")
    print("# (wave tile as forced in tools/gemm_pp_pmc.py), tile = cst_gemm_bf16_kernel (LDS-DMA tile kernels, the plan's own tile).  Durations are those under the counters.")
    for (shape, kern), d in agg.items():
        m = {k: sum(v) / len(v) for k, v in d.items()}
        if shape[0] == "group":                               # one layer's four dW: sum of M_p N_p = 7168 x 768 outputs over T tokens, fp32 out
            M, N, K = 7168, 768, shape[1]
            minb_override = 2.0 * K * (768 + 2048 + 2048 + 768 + 768 + 768 + 2304 + 768) + 4.0 * M * N
            label = f"dW group of a d=768 layer, {K} tokens"
        else:
            M, N, K = shape
            minb_override, label = None, None
        busy = 100.0 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(m.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1)
        hbm = 2 * m.get("FETCH_SIZE", 0) * 1024 + m.get("WRITE_SIZE", 0) * 1024
        minb = minb_override if minb_override is not None else 2.0 * (M * K + N * K) + 2.0 * M * N
        hit = m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1)
        print(f"{label or f'{M}x{N}x{K}'} {kern:4s} dur {m['dur_us']:6.1f} us | MFMA busy {busy:5.1f} % | fabric-side bytes {hbm / 1e6:6.1f} MB = {hbm / minb:4.2f} x the minimal {minb / 1e6:.1f} MB "
              f"(fetched {2 * m.get('FETCH_SIZE', 0) * 1024 / 1e6:.1f}, written {m.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f}) | L2 hit {100 * hit:4.1f} % | "
              f"LDS bank-conflict cycles / LDS active {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f} | "
              f"wave cycles {m.get('SQ_WAVE_CYCLES', 0):.3g}, waiting {m.get('SQ_WAIT_ANY', 0):.3g}, issue-stalled {m.get('SQ_WAIT_INST_ANY', 0):.3g}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "summary":
        summary(sys.argv[2])
    else:
        run()
