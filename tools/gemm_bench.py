#!/usr/bin/env python3
"""Micro-benchmark of cst_gemm on the shapes the training step uses (GPU box only).
Times each shape with HIP events over many back-to-back launches (launch overhead amortised)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from consistent__style_transfer_amd import ops

SHAPES = [  # (M, N, K, a_kmajor, b_kmajor, note)
    (9216, 2048, 512, 1, 1, "FFN1 fwd (Matcher)"), (9216, 512, 2048, 1, 1, "FFN2 fwd"), (9216, 1536, 512, 1, 1, "QKV fwd"),
    (9216, 512, 2048, 1, 0, "FFN1 dgrad"), (9216, 2048, 512, 1, 0, "FFN2 dgrad"),
    (2048, 512, 9216, 0, 0, "FFN1 wgrad"), (512, 2048, 9216, 0, 0, "FFN2 wgrad"),
    (4608, 10000, 512, 1, 1, "vocab fwd"), (4608, 512, 10000, 1, 0, "vocab dgrad"), (10000, 512, 4608, 0, 0, "vocab wgrad"),
    (256, 2048, 640, 1, 1, "dec gates"), (256, 1024, 256, 1, 1, "enc gates"), (256, 10000, 512, 1, 1, "fn_2 step"),
    (256, 512, 1024, 1, 1, "fn_1 step"), (256, 640, 2048, 1, 0, "dec dXH"), (256, 256, 1024, 1, 0, "enc dh"),
    (256, 512, 10000, 1, 0, "fn_2 dgrad step"), (4096, 1200, 1200, 1, 1, "highway"), (69632, 300, 16, 1, 1, "disc conv"),
]


TILES = [0, 64, 128, 129]        # auto, 64x64, 128x128 with 4 waves, 128x128 with 8 waves


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    ops.set_precision(prec)
    for M, N, K, akm, bkm, note in SHAPES:
        A = torch.randn((M, K) if akm else (K, M), device="cuda")
        B = torch.randn((N, K) if bkm else (K, N), device="cuda")
        C = torch.empty(M, N, device="cuda")
        res = []
        for tile in TILES:
            for _ in range(3):
                ops.gemm(A, akm, B, bkm, C, M, N, K, tile=tile)
            torch.cuda.synchronize()
            n = 20
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(n):
                    ops.gemm(A, akm, B, bkm, C, M, N, K, tile=tile)
            g.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            res.append(a.elapsed_time(b) * 1000 / n)
        fl = 2.0 * M * N * K
        by = 4.0 * (M * K + N * K + M * N)
        best = min(res)
        print(f"{note:18s} {M:6d}x{N:5d}x{K:5d} a{akm}b{bkm}  " + "  ".join(f"t{t}:{u:7.1f}us" for t, u in zip(TILES, res))
              + f"   best {fl / best / 1e6:7.1f} TF/s  {by / best / 1e6:5.2f} TB/s(min-traffic)")


ENC_SHAPES = [  # the encoder-layer products of the bench workloads (T = 4608 MLM / 9216 Matcher; d = 512 / 768; F = 2048)
    (4608, 2304, 768, 1, 1, "QKV fwd d768 MLM"), (9216, 2304, 768, 1, 1, "QKV fwd d768 Mat"), (4608, 2048, 768, 1, 1, "FFN1 fwd d768"),
    (9216, 2048, 768, 1, 1, "FFN1 fwd d768 Mat"), (4608, 768, 2048, 1, 1, "FFN2 fwd d768"), (9216, 768, 2048, 1, 1, "FFN2 fwd d768 Mat"),
    (4608, 768, 768, 1, 1, "out fwd d768"), (9216, 768, 768, 1, 1, "out fwd d768 Mat"), (4608, 768, 2304, 1, 1, "QKV dgrad d768"),
    (4608, 1536, 512, 1, 1, "QKV fwd d512"), (4608, 2048, 512, 1, 1, "FFN1 fwd d512"), (9216, 2048, 512, 1, 1, "FFN1 fwd d512 Mat"),
    (9216, 1536, 512, 1, 1, "QKV fwd d512 Mat"), (9216, 512, 2048, 1, 1, "FFN2 fwd d512 Mat"), (15360, 2048, 512, 1, 1, "FFN1 book MLM"),
    (30720, 2048, 512, 1, 1, "FFN1 book Mat"), (4608, 10000, 768, 1, 1, "vocab fwd d768"),
]


def main_bf16(shapes=None, out_bf16=False, splitk=0):
    print("---- bf16-operand NT GEMM (global_load_lds ring); tile codes: 0 auto, 64/128 2-stage ring, +1 3-stage, +2 4-stage, 256 = 8-wave 256x256"
          + ("; bf16-only C" if out_bf16 else "; fp32 C"))
    for M, N, K, akm, bkm, note in (shapes or SHAPES):
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        Ab, _ = ops.cast_bf16(A, want_t=False)
        Bb, _ = ops.cast_bf16(B, want_t=False)
        C = torch.empty(M, N, device="cuda") if not out_bf16 else None
        Cb = torch.empty(M, (N + 63) // 64 * 64, device="cuda", dtype=torch.int16) if out_bf16 else None
        res = []
        tiles = [int(x) for x in os.environ.get("CST_BENCH_TILES", "0,64,65,128,129,130,256").split(",")]
        for tile in tiles:
            for _ in range(3):
                ops.gemm_bf16(Ab, Bb, M, N, C=C, Cb=Cb, tile=tile, splitk=splitk)
            torch.cuda.synchronize()
            n = 20
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(n):
                    ops.gemm_bf16(Ab, Bb, M, N, C=C, Cb=Cb, tile=tile, splitk=splitk)
            g.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            res.append(a.elapsed_time(b) * 1000 / n)
        fl = 2.0 * M * N * K
        best = min(res)
        print(f"{note:18s} {M:6d}x{N:5d}x{K:5d}  " + "  ".join(f"t{t}:{u:7.1f}us" for t, u in zip(tiles, res)) + f"   best {fl / best / 1e6:7.1f} TF/s")


def main_pmc():
    """A few eager launches per (shape, tile) for rocprofv3 --pmc (tools/profile_gemm_pmc.sh): kernels are told apart by name +
    grid size in the counter CSV."""
    shapes = [(9216, 2048, 768), (4608, 2304, 768), (9216, 768, 2048)]
    for M, N, K in shapes:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        Ab, _ = ops.cast_bf16(A, want_t=False)
        Bb, _ = ops.cast_bf16(B, want_t=False)
        C = torch.empty(M, N, device="cuda")
        for tile in (64, 128, 256):
            for _ in range(6):
                ops.gemm_bf16(Ab, Bb, M, N, C=C, tile=tile)
            torch.cuda.synchronize()


def main_abl():
    """Where the 256-wide kernels' time goes: the same launch with parts switched off (CST_GB_ABL bits: 1 no DMA, 2 no MFMA, 4 no
    fragment reads, 8 no write-out; results are wrong, only the time means something)."""
    shapes = [(9216, 2048, 768), (4608, 2304, 768), (9216, 768, 2048), (16384, 4096, 1024)]
    masks = [0, 8, 1, 9, 2, 6, 14, 7, 15]
    for M, N, K in shapes:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        Ab, _ = ops.cast_bf16(A, want_t=False)
        Bb, _ = ops.cast_bf16(B, want_t=False)
        C = torch.empty(M, N, device="cuda")
        for tile in (64, 128, 256, 252):
            row = []
            for mask in masks:
                os.environ["CST_GB_ABL"] = str(mask)
                for _ in range(3):
                    ops.gemm_bf16(Ab, Bb, M, N, C=C, tile=tile)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(20):
                    ops.gemm_bf16(Ab, Bb, M, N, C=C, tile=tile)
                b.record()
                torch.cuda.synchronize()
                row.append(a.elapsed_time(b) * 1000 / 20)
            os.environ["CST_GB_ABL"] = "0"
            print(f"{M}x{N}x{K} t{tile}  " + "  ".join(f"abl{m}:{u:6.1f}" for m, u in zip(masks, row)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "abl":
        main_abl()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pmc":
        main_pmc()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bf16nt":
        main_bf16()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rec":
        # the decoder's per-step products (M = batch = 256): tile 64 = 2-stage ring, 66 = 4-stage ring; split-K chosen by the library
        os.environ.setdefault("CST_BENCH_TILES", "0,64,65,66")
        main_bf16([(256, 2048, 640, 1, 1, "dec gates fwd"), (256, 512, 1024, 1, 1, "fn_1 fwd"), (256, 10000, 512, 1, 1, "fn_2 fwd"),
                   (256, 512, 10048, 1, 1, "fn_2 dgrad"), (256, 1024, 512, 1, 1, "fn_1 dgrad"), (256, 640, 2048, 1, 1, "gates dgrad"),
                   (256, 10000, 128, 1, 1, "dp += dx E^T"), (4608, 512, 10048, 1, 1, "fn_2 dgrad all steps"), (4608, 1024, 512, 1, 1, "fn_1 dgrad all steps")])
        print("---- the same without split-K (one launch instead of product + reduce)")
        main_bf16([(256, 2048, 640, 1, 1, "dec gates fwd"), (256, 512, 1024, 1, 1, "fn_1 fwd"), (256, 1024, 512, 1, 1, "fn_1 dgrad"),
                   (256, 640, 2048, 1, 1, "gates dgrad"), (256, 512, 10048, 1, 1, "fn_2 dgrad")], splitk=1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "enc":
        main_bf16(ENC_SHAPES)
        main_bf16(ENC_SHAPES[:6], out_bf16=True)
        sys.exit(0)
    main()
