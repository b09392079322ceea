#!/usr/bin/env python3
"""The big-tile ping-pong GEMM (csrc/gemm_pp.hip) on the GPU: `check` = results against an fp32 product of the same bf16-rounded operands over
wave tiles, ragged shapes and every epilogue; `bench` = the encoder-layer shapes of the step on every wave tile of the menu, next to the
LDS-DMA tile kernels (tile code 999) and the vendor GEMM (torch.matmul, calibration only), random data, hipGraph of 20 launches each.

    python tools/gemm_pp_bench.py check
    python tools/gemm_pp_bench.py bench [M N K ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from consistent__style_transfer_amd import ops  # noqa: E402
from consistent__style_transfer_amd._lib import call_plain  # noqa: E402

MENU = [(8, 4), (7, 4), (6, 4), (5, 4), (4, 4), (8, 3), (7, 3), (6, 3), (5, 3), (8, 2), (7, 2), (6, 2)]
ENC = [(9216, 2048, 768), (9216, 768, 2048), (9216, 2304, 768), (9216, 768, 2304), (9216, 768, 768),
       (4608, 2048, 768), (4608, 768, 2048), (4608, 2304, 768), (4608, 768, 2304), (4608, 768, 768), (4096, 4096, 4096)]


def code(tm, tn):
    return 1000 + 100 * tm + tn


def bf(x):
    return x.view(torch.bfloat16).float()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1000 / n)
    return best


def check():
    torch.manual_seed(0)
    worst = 0.0
    cases = []
    # (M, N, K, wave tiles): whole tiles, ragged rows, ragged columns, one K-tile pair, many tiles per workgroup
    for (M, N, K) in [(2048, 1024, 256), (1000, 520, 192), (9216, 768, 128), (4608, 2304, 768), (3000, 1028, 320)]:
        for (tm, tn) in MENU:
            cases.append((M, N, K, tm, tn, "plain"))
    for epi in ("bias", "relu_drop", "leaky", "gate3", "gate4", "addend", "accum", "both", "alpha"):
        for (tm, tn) in [(8, 4), (7, 3), (5, 3), (6, 2)]:
            cases.append((1500, 776, 256, tm, tn, epi))
    for (M, N, K, tm, tn, epi) in cases:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        Ab, Bb = ops.cast_bf16(A, want_t=False)[0], ops.cast_bf16(B, want_t=False)[0]
        ref = bf(Ab)[:, :K] @ bf(Bb)[:, :K].t()
        Np = (N + 63) // 64 * 64
        kw, C, Cb = {}, torch.full((M, N), 7.0, device="cuda"), None
        if epi == "bias":
            bias = torch.randn(N, device="cuda")
            kw = dict(bias=bias)
            ref = ref + bias
        elif epi == "relu_drop":
            bias = torch.randn(N, device="cuda")
            kw = dict(bias=bias, act=1, drop=ops.Drop(0.1, 123, 5))
            C, Cb = None, torch.zeros(M, Np, device="cuda", dtype=torch.int16)
            # same masks, same arithmetic on the tile kernels: compare with THEM (bit-level agreement of the mask placement)
            Cb2 = torch.zeros(M, Np, device="cuda", dtype=torch.int16)
            ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb2, tile=128, **kw)
            ref = bf(Cb2)[:, :N]
        elif epi == "leaky":
            kw = dict(act=2)
            ref = torch.where(ref > 0, ref, 0.1 * ref)
        elif epi in ("gate3", "gate4"):
            aux = ops.cast_bf16(torch.randn(M, N, device="cuda"), want_t=False)[0]
            kw = dict(aux=aux, act=3 if epi == "gate3" else 4, gate_scale=1.25)
            gate = bf(aux)[:, :N] > 0
            ref = torch.where(gate, ref * 1.25, torch.zeros_like(ref)) if epi == "gate3" else torch.where(gate, ref, 0.1 * ref)
        elif epi == "addend":
            add = torch.randn(M, N, device="cuda")
            kw = dict(addend=add)
            ref = ref + add
        elif epi == "accum":
            kw = dict(accumulate=True)
            ref = ref + 7.0
        elif epi == "both":
            Cb = torch.zeros(M, Np, device="cuda", dtype=torch.int16)
        elif epi == "alpha":
            kw = dict(alpha=0.5)
            ref = 0.5 * ref
        assert call_plain("cst_gemm_bf16_pp_config", 9216, 2304, 768) > 0
        ops.gemm_bf16(Ab, Bb, M, N, C=C, Cb=Cb, tile=code(tm, tn), **kw)
        torch.cuda.synchronize()
        scale = float(ref.abs().max())
        if C is not None:
            err = float((C - ref).abs().max()) / scale
            assert err < 2e-5 or epi == "relu_drop", (M, N, K, tm, tn, epi, "fp32 out", err)
            worst = max(worst, err)
        if Cb is not None:
            got = bf(Cb)[:, :N]
            err = float((got - ref).abs().max()) / scale
            assert err < (1e-6 if epi == "relu_drop" else 6e-3), (M, N, K, tm, tn, epi, "bf16 out", err)
            if Np != N:
                assert int(Cb[:, N:].abs().max()) == 0, "columns beyond N were written"
        print(f"ok {M}x{N}x{K} wave tile {tm}x{tn} {epi}", flush=True)
    print(f"all {len(cases)} cases ok, worst fp32 deviation {worst:.2e} of the largest element")


OCC2 = [(4, 3), (6, 2), (5, 2), (4, 2), (3, 3)]          # wave tiles built for two workgroups per CU (128 registers, 80 KB of LDS)


def bench(shapes):
    """Per shape: vendor, tile kernels, the model's pick, then every variant: `TMxTN` persistent one workgroup per CU, `TMxTN/2` two persistent
    workgroups per CU, `TMxTN/2t` two per CU with one tile per workgroup (grid = tiles), `TMxTN/t` one per CU with one tile per workgroup."""
    for M, N, K in shapes:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        Ab16, Bb16 = A.bfloat16(), B.bfloat16()
        out16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t_blas = timed(lambda: torch.matmul(Ab16, Bb16.t(), out=out16))
        Ab, Bb = ops.cast_bf16(A, want_t=False)[0], ops.cast_bf16(B, want_t=False)[0]
        Cb = torch.empty(M, (N + 63) // 64 * 64, device="cuda", dtype=torch.int16)
        C = torch.empty(M, N, device="cuda")
        t_old = timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb, tile=999))
        t_old32 = timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, C=C, tile=999))
        auto = call_plain("cst_gemm_bf16_pp_config", M, N, K)
        t_auto = timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb)) if auto else float("nan")
        per = {}
        for tm, tn in MENU:
            tiles = -(-M // (32 * tm)) * -(-N // (64 * tn))
            if tiles < 96:
                continue
            per[f"{tm}x{tn}"] = timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb, tile=code(tm, tn)), n=10)
            if (tm, tn) in ((8, 4), (5, 4), (7, 3), (5, 3)):
                per[f"{tm}x{tn}/t"] = timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb, tile=20000 + code(tm, tn)), n=10)
        for tm, tn in OCC2:
            per[f"{tm}x{tn}/2"] = timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb, tile=10000 + code(tm, tn)), n=10)
            per[f"{tm}x{tn}/2t"] = timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb, tile=30000 + code(tm, tn)), n=10)
        best = min(per, key=per.get)
        t32 = timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, C=C, tile=(30000 if best.endswith("/2t") else 10000 if best.endswith("/2") else 20000 if best.endswith("/t") else 0)
                                          + code(*[int(v) for v in best.split("/")[0].split("x")])), n=10)
        fl = 2.0 * M * N * K
        tf = lambda t: fl / t / 1e6
        print(f"{M:6d}x{N:5d}x{K:5d} vendor {t_blas:6.1f} ({tf(t_blas):5.0f}) | tile kernels {t_old:6.1f} ({tf(t_old):5.0f}), fp32 out {t_old32:6.1f} | auto {t_auto:6.1f} cfg {auto} | "
              f"best {per[best]:6.1f} ({tf(per[best]):5.0f}) {best}, fp32 out {t32:6.1f} | " + " ".join(f"{k}:{t:.1f}" for k, t in per.items()), flush=True)


def gn(shapes):
    """XCD strip width (tile columns per strip, CST_GEMM_GN is read once per process: one subprocess per value)."""
    import subprocess
    for g in (2, 4, 8, 16):
        env = dict(os.environ, CST_GEMM_GN=str(g))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "gn1"] + [str(v) for s3 in shapes for v in s3], env=env, capture_output=True, text=True).stdout
        print(f"GN={g}: " + out.strip().replace("\n", " || "), flush=True)


def gn1(shapes):
    for M, N, K in shapes:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        Ab, Bb = ops.cast_bf16(A, want_t=False)[0], ops.cast_bf16(B, want_t=False)[0]
        Cb = torch.empty(M, (N + 63) // 64 * 64, device="cuda", dtype=torch.int16)
        row = [f"{lab} {timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb, tile=t), n=10):.1f}" for lab, t in
               (("5x4", code(5, 4)), ("7x3", code(7, 3)), ("4x4", code(4, 4)), ("5x3", code(5, 3)), ("6x2/2t", 30000 + code(6, 2)))]
        print(f"{M}x{N}x{K}: " + " ".join(row))


def epi(shapes):
    """What the step's epilogues cost on top of the plain product (same launch-graph timing): FFN1 forward (bias + ReLU + dropout -> bf16),
    the FFN hidden gradient (ReLU' / dropout' gate read from a bf16 aux -> bf16), the residual products (fp32 out, + addend)."""
    for M, N, K in shapes:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        Ab, Bb = ops.cast_bf16(A, want_t=False)[0], ops.cast_bf16(B, want_t=False)[0]
        Cb = torch.empty(M, (N + 63) // 64 * 64, device="cuda", dtype=torch.int16)
        C = torch.empty(M, N, device="cuda")
        bias, add = torch.randn(N, device="cuda"), torch.randn(M, N, device="cuda")
        aux = ops.cast_bf16(torch.randn(M, N, device="cuda"), want_t=False)[0]
        kinds = {"plain bf16": dict(Cb=Cb), "bias+relu+drop bf16": dict(Cb=Cb, bias=bias, act=1, drop=ops.Drop(0.1, 1, 2)),
                 "gate(aux) bf16": dict(Cb=Cb, aux=aux, act=3, gate_scale=1.1), "bias f32": dict(C=C, bias=bias), "addend f32": dict(C=C, addend=add)}
        for label, tile in (("tile kernels", 999), ("auto", 0), ("5x4", code(5, 4)), ("7x3", code(7, 3)), ("6x2/2t", 30000 + code(6, 2)), ("4x4", code(4, 4)), ("5x3", code(5, 3))):
            row = []
            for kn, kw in kinds.items():
                row.append(f"{kn} {timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, tile=tile, **kw), n=10):.1f}")
            print(f"{M}x{N}x{K} {label:12s}: " + " | ".join(row), flush=True)


def abl(shapes, cfgs=((8, 4), (7, 3))):
    """Timing ablations (bench build: CST_BENCH_VARIANTS=1 python -m consistent__style_transfer_amd.build): what the K loop's parts cost.
    1 no LDS-DMA, 2 no MFMA, 4 no fragment reads, 8 one barrier per phase, 16 no vmcnt waits, 32 no epilogue."""
    assert call_plain("cst_bench_variants") == 1, "needs the bench build"
    combos = [0, 32, 16, 1 | 16, 4, 2, 8, 2 | 4, 1 | 16 | 4, 1 | 16 | 2 | 4, 1 | 16 | 2 | 4 | 8, 1 | 16 | 2 | 4 | 8 | 32]
    for M, N, K in shapes:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        Ab, Bb = ops.cast_bf16(A, want_t=False)[0], ops.cast_bf16(B, want_t=False)[0]
        Cb = torch.empty(M, (N + 63) // 64 * 64, device="cuda", dtype=torch.int16)
        for tm, tn in cfgs:
            row = []
            for c in combos:
                os.environ["CST_PP_ABL"] = str(c)
                row.append((c, timed(lambda: ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb, tile=code(tm, tn)), n=10)))
            os.environ.pop("CST_PP_ABL")
            print(f"{M}x{N}x{K} cfg {tm}x{tn}: " + "  ".join(f"[{c}] {t:.1f}" for c, t in row), flush=True)


def stamps(shapes, cfgs=((8, 4), (5, 4), (7, 3), (5, 3))):
    """Bench build: where one workgroup's time goes (cycle stamps of waves 0 and 4 of workgroups 0 and 1 at the section boundaries)."""
    import ctypes
    from consistent__style_transfer_amd import build
    so = ctypes.CDLL(build.LIB)
    assert call_plain("cst_bench_variants") == 1, "needs the bench build"
    os.environ["CST_PP_STAMPS"] = "1"
    for M, N, K in shapes:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        Ab, Bb = ops.cast_bf16(A, want_t=False)[0], ops.cast_bf16(B, want_t=False)[0]
        Cb = torch.empty(M, (N + 63) // 64 * 64, device="cuda", dtype=torch.int16)
        for tm, tn in cfgs:
            for _ in range(5):
                ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb, tile=code(tm, tn))
            buf = (ctypes.c_ulonglong * 64)()
            rc = so.cst_gemm_bf16_pp_stamps(buf)
            assert rc == 0, rc
            for w in range(4):
                v = list(buf[16 * w:16 * w + 16])
                n = min(int(v[12]), 12)
                cyc = [v[i + 1] - v[i] for i in range(n - 1)]
                us = (v[14] - v[13]) / 100.0
                clk = (v[n - 1] - v[0]) / max(us, 1e-9) / 1000.0
                names = ["prologue"] + [x for r in range(6) for x in (f"K loop {r}", f"epilogue {r}")]
                print(f"{M}x{N}x{K} cfg {tm}x{tn} wg {w // 2} group {w % 2}: {us:6.1f} us in-kernel, ~{clk:.2f} GHz | " +
                      "  ".join(f"{names[i]} {c} ({c / clk / 1000.0:.1f} us)" for i, c in enumerate(cyc)), flush=True)
    os.environ.pop("CST_PP_STAMPS")


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        check()
    elif mode in ("gn", "gn1"):
        v = [int(x) for x in sys.argv[2:]]
        (gn if mode == "gn" else gn1)([tuple(v[i:i + 3]) for i in range(0, len(v), 3)] or [(9216, 2048, 768), (9216, 768, 2048), (9216, 2304, 768)])
    elif mode == "epi":
        v = [int(x) for x in sys.argv[2:]]
        epi([tuple(v[i:i + 3]) for i in range(0, len(v), 3)] or [(9216, 2048, 768), (9216, 768, 2048)])
    elif mode == "stamps":
        v = [int(x) for x in sys.argv[2:]]
        stamps([tuple(v[i:i + 3]) for i in range(0, len(v), 3)] or [(9216, 2048, 768), (9216, 768, 2048), (4096, 4096, 4096)])
    elif mode == "abl":
        v = [int(x) for x in sys.argv[2:]]
        abl([tuple(v[i:i + 3]) for i in range(0, len(v), 3)] or [(4096, 4096, 4096), (9216, 2048, 768)])
    else:
        v = [int(x) for x in sys.argv[2:]]
        bench([tuple(v[i:i + 3]) for i in range(0, len(v), 3)] or ENC)
