"""GPU parity of the individual HIP kernels (called through the C ABI) against plain fp32 torch
on the CPU.  Tolerances: the f32 MFMA mode and all VALU kernels are exact fp32 arithmetic in a
different summation order (rtol 2e-4); the bf16 MFMA mode rounds operands to 8 significant bits
(rtol 3e-2 of the output scale)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import rng as orng  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from consistent__style_transfer_amd import ops as o
    return o


def dev(t):
    return t.cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).float()


def close(a, b, rtol, atol, msg=""):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=rtol, atol=atol, err_msg=msg)


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(37, 53, 19), (256, 512, 128), (700, 390, 100), (1024, 768, 96), (130, 10000, 64),
                                   (64, 64, 10000)])
@pytest.mark.parametrize("akm,bkm", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_layouts(ops, prec, shape, akm, bkm):
    ops.set_precision(prec)
    M, N, K = shape
    A = rnd(M, K, seed=1)
    B = rnd(K, N, seed=2)
    ref = A @ B
    Ad = dev(A if akm else A.t().contiguous())
    Bd = dev(B.t().contiguous() if bkm else B)
    C = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm(Ad, akm, Bd, bkm, C, M, N, K)
    tol = 2e-4 if prec == "f32" else 2e-2
    close(C, ref, tol, tol * math.sqrt(K), f"{shape} {akm}{bkm}")
    ops.set_precision("bf16")


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("shape,akm,bkm", [((256, 512, 10000), 1, 0), ((300, 24, 65536), 0, 0), ((37, 53, 1900), 1, 1),
                                           ((256, 256, 1024), 1, 0), ((100, 1200, 4096), 0, 0)])
def test_gemm_splitk(ops, prec, shape, akm, bkm):
    """Long-K / few-tile shapes take the split-K path (auto and forced) with a gated, accumulating epilogue."""
    ops.set_precision(prec)
    M, N, K = shape
    A, B = rnd(M, K, seed=1), rnd(K, N, seed=2)
    aux, c0 = rnd(M, N, seed=3), rnd(M, N, seed=4)
    ref = c0 + torch.where(aux > 0, A @ B, 0.1 * (A @ B))
    Ad = dev(A if akm else A.t().contiguous())
    Bd = dev(B.t().contiguous() if bkm else B)
    tol = 2e-4 if prec == "f32" else 2e-2
    for sk in (0, 7):
        C = dev(c0.clone())
        ops.gemm(Ad, akm, Bd, bkm, C, M, N, K, aux=dev(aux), act=4, accumulate=True, splitk=sk)
        close(C, ref, tol, tol * math.sqrt(K), f"{shape} splitk={sk}")
    ops.set_precision("bf16")


def _bf16_round(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("R,C", [(5, 7), (64, 64), (100, 130), (4608, 512), (37, 2048), (2048, 512), (10000, 768), (101, 36), (3, 4), (130, 260)])
def test_cast_bf16(ops, R, C):
    x = rnd(R, C, seed=1)
    rm, tr = ops.cast_bf16(dev(x))
    ref = _bf16_round(x)
    got = rm.view(torch.bfloat16).float().cpu()
    assert got.shape == (R, (C + 63) // 64 * 64)
    assert torch.equal(got[:, :C], ref) and float(got[:, C:].abs().sum()) == 0.0
    gt = tr.view(torch.bfloat16).float().cpu()
    assert gt.shape == (C, (R + 63) // 64 * 64)
    assert torch.equal(gt[:, :R], ref.t()) and float(gt[:, R:].abs().sum()) == 0.0
    # dropout in the cast == oracle mask; bf16 -> bf16 transpose
    d = ops.Drop(0.25, 3, 1003)
    rm2, _ = ops.cast_bf16(dev(x), want_t=False, drop=d)
    mask = torch.from_numpy(orng.dropout_mask(3, 1003, (R, C), 0.25))
    assert torch.equal(rm2.view(torch.bfloat16).float().cpu()[:, :C], _bf16_round(x * mask))
    _, t2 = ops.cast_bf16(rm[:, :C], want_rm=False)
    assert torch.equal(t2.view(torch.bfloat16).float().cpu()[:, :R], ref.t())
    close(ops.colsum_bf16(rm, C), ref.sum(0), 1e-3, 1e-2)


@pytest.mark.parametrize("shape", [(37, 53, 19), (256, 512, 128), (700, 390, 100), (9216, 2048, 512), (4608, 512, 2048),
                                   (2048, 512, 9216), (256, 2048, 640), (256, 512, 1024), (130, 10000, 64), (64, 64, 10000)])
@pytest.mark.parametrize("tile", [0, 64, 128, 136])
def test_gemm_bf16(ops, shape, tile):
    M, N, K = shape
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    Ab, _ = ops.cast_bf16(dev(A), want_t=False)
    Bb, _ = ops.cast_bf16(dev(B), want_t=False)
    ref = _bf16_round(A) @ _bf16_round(B).t()
    C = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_bf16(Ab, Bb, M, N, C=C, tile=tile)
    close(C, ref, 2e-3, 2e-3 * math.sqrt(K), f"{shape}")
    # epilogue: bias + addend + gate by a bf16 aux + bf16 output
    bias, add, aux = rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, seed=5)
    auxb, _ = ops.cast_bf16(dev(aux), want_t=False)
    Cb = torch.zeros(M, (N + 63) // 64 * 64, device="cuda", dtype=torch.int16)
    C2 = torch.empty(M, N, device="cuda")
    ops.gemm_bf16(Ab, Bb, M, N, C=C2, Cb=Cb, bias=dev(bias), addend=dev(add), aux=auxb, act=3, gate_scale=1.25, tile=tile)
    ref2 = torch.where(_bf16_round(aux) > 0, (ref + bias + add) * 1.25, torch.zeros_like(ref))
    close(C2, ref2, 2e-3, 2e-3 * math.sqrt(K))
    close(Cb.view(torch.bfloat16).float()[:, :N], ref2, 1e-2, 1e-2 * math.sqrt(K))
    # relu + dropout epilogue, bf16-only output
    d = ops.Drop(0.3, 11, 1002)
    Cb3 = torch.zeros(M, (N + 63) // 64 * 64, device="cuda", dtype=torch.int16)
    ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb3, bias=dev(bias), act=1, drop=d, tile=tile)
    mask = torch.from_numpy(orng.dropout_mask(11, 1002, (M, N), 0.3))
    close(Cb3.view(torch.bfloat16).float()[:, :N], torch.relu(ref + bias) * mask, 1e-2, 1e-2 * math.sqrt(K))
    # accumulate + alpha into a strided C (the write-out's load-carrying instantiation reads the old C ahead of its stores)
    big = torch.full((M, N + 8), 2.0, device="cuda")
    ops.gemm_bf16(Ab, Bb, M, N, C=big[:, 4:4 + N], accumulate=True, alpha=0.5, tile=tile)
    close(big[:, 4:4 + N], 2.0 + 0.5 * ref, 2e-3, 2e-3 * math.sqrt(K))
    assert (big[:, :4] == 2.0).all() and (big[:, 4 + N:] == 2.0).all()


@pytest.mark.parametrize("shape", [(256, 256, 64), (256, 256, 128), (512, 256, 192), (256, 768, 320), (4608, 2304, 768), (9216, 2048, 512),
                                   (4608, 2048, 768), (1024, 512, 2304), (9216, 2304, 768)])
@pytest.mark.parametrize("tile", [256, 252, 248])
def test_gemm_bf16_256_tile(ops, shape, tile):
    """The 256-wide kernels (tile code 256: 256 x 256 / 8 waves / 4-stage ring; 252: 128 x 256 / 4 waves / 3-stage ring, two
    workgroups per CU; 248: persistent loader / consumer kernel, 256 x 128 tiles): two, four, odd and many 32-deep K-tiles,
    one and several tiles per workgroup; every epilogue of the small-tile kernel.  These kernels never dispatch on their own and are
    compiled only into bench builds (CST_BENCH_VARIANTS=1 python -m consistent__style_transfer_amd.build --force)."""
    from consistent__style_transfer_amd._lib import call_plain
    if not call_plain("cst_bench_variants"):
        pytest.skip("bench-only GEMM variants are not in the shipped library")
    test_gemm_bf16(ops, shape, tile)
    M, N, K = shape
    # identity check with an asymmetric B: C = I[:, :K] B^T must reproduce B^T exactly (catches a transposed C map, a symmetric
    # operand would not) -- cdna_hip_programming.md section 3
    A = torch.zeros(M, K)
    A[torch.arange(min(M, K)), torch.arange(min(M, K))] = 1.0
    B = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251) - 125.0          # exactly representable in bf16
    Ab, _ = ops.cast_bf16(dev(A), want_t=False)
    Bb, _ = ops.cast_bf16(dev(B), want_t=False)
    C = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_bf16(Ab, Bb, M, N, C=C, tile=tile)
    ref = torch.zeros(M, N)
    ref[:min(M, K)] = B.t()[:min(M, K)]
    assert torch.equal(C.cpu(), ref)


# ------------------------------------------------------------------------------------------ big-tile ping-pong GEMM (csrc/gemm_pp.hip)
def _pp_code(tm, tn, occ2=False, one_tile=False):
    return 1000 + 100 * tm + tn + (10000 if occ2 else 0) + (20000 if one_tile else 0)


def _bfview(x):
    return x.view(torch.bfloat16).float()


PP_BUILDS = [(8, 4, False, False), (7, 3, False, False), (5, 4, False, False), (5, 3, False, True), (6, 2, False, False), (4, 4, False, False),
             (4, 3, True, False), (6, 2, True, True), (5, 2, True, True), (4, 2, True, False), (3, 3, True, True)]


@pytest.mark.parametrize("shape", [(1280, 1024, 256), (1000, 520, 192), (3000, 1028, 320), (2304, 768, 768)])
def test_gemm_pp_every_build_equals_the_tile_kernels_bit_for_bit(ops, shape):
    """Round 4: every build of the ping-pong kernel (wave tile, one or two workgroups per CU, persistent or one tile per workgroup) against an
    fp32 product of the same bf16-rounded operands AND against the LDS-DMA tile kernels (tile code 999).  All of them sum a dot product in the
    same order -- K-tiles in sequence, two v_mfma_f32_16x16x32_bf16 per K-tile -- so the fp32 outputs must be IDENTICAL: which build the
    measured choice picks can never change a result.  Ragged rows and columns, several tiles per workgroup, 3 to 12 K-tiles."""
    M, N, K = shape
    A, B = dev(rnd(M, K, seed=1)), dev(rnd(N, K, seed=2))
    Ab, Bb = ops.cast_bf16(A, want_t=False)[0], ops.cast_bf16(B, want_t=False)[0]
    ref = _bfview(Ab)[:, :K] @ _bfview(Bb)[:, :K].t()
    base = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_bf16(Ab, Bb, M, N, C=base, tile=999)
    assert float((base - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    Np = (N + 63) // 64 * 64
    for tm, tn, occ2, one in PP_BUILDS:
        C = torch.full((M, N), float("nan"), device="cuda")
        Cb = torch.full((M, Np), 0x7FC0, device="cuda", dtype=torch.int16)
        ops.gemm_bf16(Ab, Bb, M, N, C=C, tile=_pp_code(tm, tn, occ2, one))
        ops.gemm_bf16(Ab, Bb, M, N, Cb=Cb, tile=_pp_code(tm, tn, occ2, one))
        assert torch.equal(C, base), (shape, tm, tn, occ2, one, float((C - base).abs().max()))
        assert torch.equal(_bfview(Cb)[:, :N], base.to(torch.bfloat16).float()), (shape, tm, tn, occ2, one, "bf16 output")
        if Np != N:
            assert bool((Cb[:, N:] == 0x7FC0).all()), "columns beyond N were written"
    auto = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_bf16(Ab, Bb, M, N, C=auto)                       # the entry point's own (measured) choice
    assert torch.equal(auto, base)


@pytest.mark.parametrize("build", [(8, 4, False, False), (7, 3, False, False), (5, 3, False, True), (4, 3, True, False), (6, 2, True, True)])
def test_gemm_pp_epilogues(ops, build):
    """Every epilogue of cst_gemm_bf16 on the ping-pong kernel: bias, ReLU + dropout (same counter-based masks as the tile kernels: compared
    with them), LeakyReLU, the two gates (act 3 / 4 with a bf16 aux), addend, C += (accumulate), alpha, fp32 + bf16 outputs together."""
    tm, tn, occ2, one = build
    code = _pp_code(tm, tn, occ2, one)
    M, N, K = 1500, 776, 256
    A, B = dev(rnd(M, K, seed=3)), dev(rnd(N, K, seed=4))
    Ab, Bb = ops.cast_bf16(A, want_t=False)[0], ops.cast_bf16(B, want_t=False)[0]
    raw = _bfview(Ab)[:, :K] @ _bfview(Bb)[:, :K].t()
    Np = (N + 63) // 64 * 64
    bias, add = dev(rnd(N, seed=5)), dev(rnd(M, N, seed=6))
    aux = ops.cast_bf16(dev(rnd(M, N, seed=7)), want_t=False)[0]
    gate = _bfview(aux)[:, :N] > 0
    cases = {
        "bias": (dict(bias=bias), raw + bias),
        "leaky": (dict(act=2), torch.where(raw > 0, raw, 0.1 * raw)),
        "gate3": (dict(aux=aux, act=3, gate_scale=1.25), torch.where(gate, raw * 1.25, torch.zeros_like(raw))),
        "gate4": (dict(aux=aux, act=4), torch.where(gate, raw, 0.1 * raw)),
        "addend": (dict(addend=add, bias=bias), raw + bias + add),
        "alpha": (dict(alpha=0.5), 0.5 * raw),
    }
    for name, (kw, ref) in cases.items():
        C = torch.full((M, N), float("nan"), device="cuda")
        Cb = torch.zeros(M, Np, device="cuda", dtype=torch.int16)
        ops.gemm_bf16(Ab, Bb, M, N, C=C, Cb=Cb, tile=code, **kw)
        sc = float(ref.abs().max())
        assert float((C - ref).abs().max()) < 2e-5 * sc, (build, name)
        assert torch.equal(_bfview(Cb)[:, :N], C.to(torch.bfloat16).float()), (build, name, "bf16 twin")
    C = torch.full((M, N), 7.0, device="cuda")
    ops.gemm_bf16(Ab, Bb, M, N, C=C, tile=code, accumulate=True)
    assert float((C - (raw + 7.0)).abs().max()) < 2e-5 * float(raw.abs().max()), (build, "accumulate")
    drop = ops.Drop(0.1, 123, 5)
    got, want = torch.zeros(M, Np, device="cuda", dtype=torch.int16), torch.zeros(M, Np, device="cuda", dtype=torch.int16)
    ops.gemm_bf16(Ab, Bb, M, N, Cb=got, tile=code, bias=bias, act=1, drop=drop)
    ops.gemm_bf16(Ab, Bb, M, N, Cb=want, tile=128, bias=bias, act=1, drop=drop)
    assert torch.equal(got, want), (build, "ReLU + dropout: mask placement or arithmetic differs from the tile kernels")
    assert 0.3 < float((_bfview(got)[:, :N] == 0).float().mean()) < 0.7


@pytest.mark.parametrize("shape", [(64, 128, 64), (130, 300, 200), (4608, 2304, 768), (4608, 768, 2048), (256, 512, 1024), (96, 2048, 640)])
def test_gemm_bf16_w8_fp8_weights(ops, shape):
    """W8A16 (BASELINE configs[4]): fp8 e4m3 (OCP) weights with per-output-channel scales.  The quantiser is checked against
    torch.float8_e4m3fn bit for bit (which also proves the OCP -- not FNUZ -- encoding), the GEMM against the product with the
    DEQUANTISED weights (so only bf16-product rounding is left), and against the unquantised product within the e4m3 grid error."""
    M, N, K = shape
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2)
    W[3 % N] *= 37.0                                               # rows of very different magnitude: per-row scales matter
    Ab, _ = ops.cast_bf16(dev(A), want_t=False)
    q, sc = ops.cast_fp8_rows(dev(W))
    assert q.shape == (N, (K + 63) // 64 * 64) and q.dtype == torch.uint8
    ref_sc = W.abs().amax(1) / 448.0
    close(sc, ref_sc, 1e-6, 0.0)
    ref_q = (W * (1.0 / sc.cpu())[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)      # the kernel multiplies by 1 / scale
    assert torch.equal(q[:, :K].cpu().view(torch.float8_e4m3fn).float(), ref_q.float())
    assert int(q[:, K:].cpu().sum()) == 0
    Wdq = ref_q.float() * ref_sc[:, None]
    ref = _bf16_round(A) @ Wdq.t()
    C = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_bf16_w8(Ab, q, sc, M, N, C=C)
    close(C, ref, 2e-3, 2e-3 * math.sqrt(K) * float(W.abs().max()) / 4, f"{shape}")
    full = A @ W.t()
    assert rel(C, full) < 6e-2                                      # e4m3: 3 mantissa bits -> ~3 % rms per weight, averaged down over K
    # transposed quantisation (the dgrad operand): W^T with scales along its own rows
    qt, sct = ops.cast_fp8_rows(dev(W), transposed=True)
    ref_sct = W.abs().amax(0) / 448.0
    close(sct, ref_sct, 1e-6, 0.0)
    ref_qt = (W.t() * (1.0 / sct.cpu())[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    assert torch.equal(qt[:, :N].cpu().view(torch.float8_e4m3fn).float(), ref_qt.float())
    # epilogue: bias + relu + bf16-only output, as the FFN1 product uses it
    bias = rnd(N, seed=3)
    Cb = torch.zeros(M, (N + 63) // 64 * 64, device="cuda", dtype=torch.int16)
    ops.gemm_bf16_w8(Ab, q, sc, M, N, Cb=Cb, bias=dev(bias), act=1)
    close(Cb.view(torch.bfloat16).float()[:, :N], torch.relu(ref + bias), 1e-2, 1e-2 * math.sqrt(K) * float(W.abs().max()) / 4)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("tile", [64, 128])
def test_gemm_epilogues(ops, tile):
    ops.set_precision("f32")
    M, N, K = 150, 90, 70
    A, W = rnd(M, K, seed=3), rnd(N, K, seed=4)
    bias, add, aux = rnd(N, seed=5), rnd(M, N, seed=6), rnd(M, N, seed=7)
    base = A @ W.t() + bias + add
    for act, ref in ((0, base), (1, torch.relu(base)), (2, F.leaky_relu(base, 0.1)),
                     (3, torch.where(aux > 0, base * 1.5, torch.zeros_like(base))),
                     (4, torch.where(aux > 0, base, 0.1 * base))):
        C = torch.empty(M, N, device="cuda")
        ops.gemm(dev(A), 1, dev(W), 1, C, M, N, K, bias=dev(bias), addend=dev(add), aux=dev(aux), act=act,
                 gate_scale=1.5, tile=tile)
        close(C, ref, 2e-4, 2e-4, f"act {act}")
    # accumulate + alpha + strided C
    big = torch.ones(M, N + 10, device="cuda")
    ops.gemm(dev(A), 1, dev(W), 1, big[:, 3:3 + N], M, N, K, accumulate=True, alpha=0.5, tile=tile)
    close(big[:, 3:3 + N], 1.0 + 0.5 * (A @ W.t()), 2e-4, 2e-4)
    assert float(big[:, :3].sum()) == 3 * M and float(big[:, 3 + N:].sum()) == 7 * M
    # dropout epilogue == oracle mask
    d = ops.Drop(0.3, 1234, 77)
    C = torch.empty(M, N, device="cuda")
    ops.gemm(dev(A), 1, dev(W), 1, C, M, N, K, drop=d, tile=tile)
    mask = torch.from_numpy(orng.dropout_mask(1234, 77, (M, N), 0.3))
    close(C, (A @ W.t()) * mask, 2e-4, 2e-4)
    ops.set_precision("bf16")


def test_linear_autograd(ops):
    ops.set_precision("f32")
    x, W, b = rnd(33, 20, seed=1), rnd(17, 20, seed=2), rnd(17, seed=3)
    for act in (0, 1, 2):
        xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
        y = xr @ Wr.t() + br
        y = torch.relu(y) if act == 1 else (F.leaky_relu(y, 0.1) if act == 2 else y)
        (y * rnd(33, 17, seed=9)).sum().backward()
        xg, Wg, bg = (dev(t).requires_grad_(True) for t in (x, W, b))
        yg = ops.linear(xg, Wg, bg, act=act)
        (yg * dev(rnd(33, 17, seed=9))).sum().backward()
        close(yg, y, 2e-4, 2e-5)
        close(xg.grad, xr.grad, 2e-4, 2e-5)
        close(Wg.grad, Wr.grad, 2e-4, 2e-5)
        close(bg.grad, br.grad, 2e-4, 2e-5)
    ops.set_precision("bf16")


# --------------------------------------------------------------------------------- row kernels
@pytest.mark.parametrize("R,V", [(7, 53), (64, 1000), (33, 10000), (5, 10003), (3, 40000)])
def test_token_ce(ops, R, V):
    x = rnd(R, V, seed=1, scale=3.0)
    t = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(5))
    xr = x.clone().requires_grad_(True)
    ref = F.cross_entropy(xr, t) * 0.7
    ref.backward()
    xg = dev(x).requires_grad_(True)
    loss = ops.token_ce(xg, dev(t), weight=0.7)
    loss.backward()
    close(loss.reshape(()), ref, 1e-5, 1e-6)
    close(xg.grad, xr.grad, 1e-4, 1e-8)
    # strided rows (a (B,T,V) slice) and unit_grad path
    big = dev(rnd(R, 3 * V, seed=2, scale=2.0))
    view = big[:, V:2 * V].detach().requires_grad_(True) if V % 4 == 0 else None
    if view is not None:
        l2 = ops.token_ce(view, dev(t), unit_grad=True)
        l2.backward()
        vr = big[:, V:2 * V].cpu().clone().requires_grad_(True)
        r2 = F.cross_entropy(vr, t)
        r2.backward()
        close(l2.reshape(()), r2, 1e-5, 1e-6)
        close(view.grad, vr.grad, 1e-4, 1e-8)


@pytest.mark.parametrize("R,V,tau", [(6, 53, 0.1), (40, 10000, 0.1), (9, 10000, 1.0), (4, 4099, 0.5)])
def test_softmax_tau(ops, R, V, tau):
    x = rnd(R, V, seed=3, scale=2.0)
    ref = torch.softmax(x / tau, -1)
    p = torch.empty(R, V, device="cuda")
    am = torch.empty(R, dtype=torch.int64, device="cuda")
    ops.softmax_tau(dev(x), 1.0 / tau, p, am)
    close(p, ref, 2e-4, 1e-9)
    assert torch.equal(am.cpu(), p.cpu().argmax(-1))
    assert torch.equal(am.cpu(), ref.argmax(-1))
    am2 = ops.argmax_rows(dev(x))
    assert torch.equal(am2.cpu(), x.argmax(-1))
    dp = rnd(R, V, seed=4)
    pr = ref.clone().requires_grad_(False)
    xr = x.clone().requires_grad_(True)
    (torch.softmax(xr / tau, -1) * dp).sum().backward()
    dx = torch.empty(R, V, device="cuda")
    ops.softmax_tau_bwd(p, dev(dp), 1.0 / tau, dx)
    close(dx, xr.grad, 5e-4, 1e-5)      # p*(dp - sum(dp p)) cancels; the sum order depends on the workgroup width


@pytest.mark.parametrize("R,V,with_gather", [(40, 10000, True), (40, 10000, False), (9, 1000, True), (5, 4100, False)])
def test_softmax_tau_bf16_twin(ops, R, V, with_gather):
    """cst_softmax_tau_gather_b: same p / argmax / fed-back embedding as the plain entries, plus bf16(p) with zero padding."""
    x = dev(rnd(R, V, seed=5, scale=2.0))
    E, Vp, T = 16, (V + 63) // 64 * 64, 3
    tab = dev(rnd(V, E, seed=6))
    p0 = torch.empty(R, V, device="cuda")
    am0 = torch.empty(R, dtype=torch.int64, device="cuda")
    e0 = torch.full((R, E), float("nan"), device="cuda")
    ops.softmax_tau(x, 2.0, p0, am0, gather=dict(table=tab, out=e0) if with_gather else None)
    p1 = torch.empty(R, V, device="cuda")
    am1 = torch.empty(R, dtype=torch.int64, device="cuda")
    e1 = torch.full((R, E), float("nan"), device="cuda")
    big = torch.full((R, T * Vp), 0x7fc0, dtype=torch.int16, device="cuda")       # rows strided like the decoder's buffer
    pb = big[:, Vp:2 * Vp]
    ops.softmax_tau(x, 2.0, p1, am1, gather=dict(table=tab, out=e1) if with_gather else None, p_b=pb)
    assert torch.equal(p0, p1) and torch.equal(am0, am1)
    if with_gather:
        assert torch.equal(e0, e1) and torch.equal(e1, tab[am1])
    assert torch.equal(pb[:, :V].contiguous().view(torch.bfloat16), p1.to(torch.bfloat16))
    assert (pb[:, V:] == 0).all()
    assert (big[:, :Vp] == 0x7fc0).all() and (big[:, 2 * Vp:] == 0x7fc0).all()


def test_shared_soft_embed_bf16_twin_matches_staged_product(ops):
    """ops.SharedSoftEmbedFn on the softmax kernel's bf16 twin == the fp32-staged product (both round operands to bf16)."""
    ops.set_precision("bf16")
    R, V = 192, 10000
    Vp = (V + 63) // 64 * 64
    x = dev(rnd(R, V, seed=7, scale=2.0))
    tabs = [dev(rnd(V, 128, seed=8)), dev(rnd(V, 256, seed=9)), dev(rnd(64, V, seed=10))]
    evs = (False, False, True)
    douts = [dev(rnd(R, w, seed=11 + i)) for i, w in enumerate((128, 256, 64))]
    res = []
    for twin in (False, True):
        p = torch.empty(R, V, device="cuda")
        pb = torch.empty(R, Vp, dtype=torch.int16, device="cuda") if twin else None
        ops.softmax_tau(x, 1.0, p, None, p_b=pb)
        p.requires_grad_(True)
        if twin:
            ops._side_put(p, pb)
        outs = ops.SharedSoftEmbedFn.apply(p, evs, *tabs)
        torch.autograd.backward(outs, douts)
        res.append(([o.detach().clone() for o in outs], p.grad.clone()))
    for a, b in zip(res[0][0], res[1][0]):
        close(a, b, 1e-3, 1e-4)
    close(res[0][1], res[1][1], 2e-2, 2e-2 * float(res[0][1].abs().max()))     # d p: dout is rounded to bf16 by a cast here, in staging there
    ref = torch.softmax(x, -1).cpu() @ tabs[1].cpu()
    close(res[1][0][1], ref, 3e-2, 3e-2 * float(ref.abs().max()))


@pytest.mark.parametrize("ev", [False, True])
def test_soft_embed_bf16_twin_matches_staged_product(ops, ev):
    """ops.SoftEmbedFn (discriminator.py:39 on the generated distribution, main_optimize.py:119-120) with the softmax kernel's bf16 twin:
    out = p @ table on the bf16 GEMM, d table = dout^T p through the transposed-read product, d p on the bf16 GEMM -- against the
    fp32-staged products of the same node (both round their operands to bf16) and an fp64 product."""
    ops.set_precision("bf16")
    R, V, E = 192, 10000, 128
    Vp = (V + 63) // 64 * 64
    x = dev(rnd(R, V, seed=7, scale=2.0))
    tab0 = dev(rnd(E, V, seed=10) if ev else rnd(V, E, seed=10))
    dout = dev(rnd(R, E, seed=11))
    res = []
    for twin in (False, True):
        p = torch.empty(R, V, device="cuda")
        pb = torch.empty(R, Vp, dtype=torch.int16, device="cuda") if twin else None
        ops.softmax_tau(x, 1.0, p, None, p_b=pb)
        p.requires_grad_(True)
        tab = tab0.clone().requires_grad_(True)
        if twin:
            ops._side_put(p, pb)
        names = []
        orig = ops.call
        ops.call = lambda nm, *a: (names.append(nm), orig(nm, *a))[1]
        try:
            out = ops.SoftEmbedFn.apply(p, tab, ev)
            out.backward(dout)
        finally:
            ops.call = orig
        assert ("cst_gemm" in names) == (not twin), names          # with the twin no fp32-staged product is left
        res.append((out.detach().clone(), p.grad.clone(), tab.grad.clone()))
    close(res[0][0], res[1][0], 1e-3, 1e-4)
    close(res[0][1], res[1][1], 2e-2, 2e-2 * float(res[0][1].abs().max()))
    close(res[0][2], res[1][2], 2e-2, 2e-2 * float(res[0][2].abs().max()))
    pr = torch.softmax(x.double(), -1).cpu()
    t64 = tab0.double().cpu()
    ref = pr @ (t64.T if ev else t64)
    close(res[1][0], ref.float(), 3e-2, 3e-2 * float(ref.abs().max()))
    dt = dout.double().cpu().T @ pr if ev else pr.T @ dout.double().cpu()
    close(res[1][2], dt.float(), 3e-2, 3e-2 * float(dt.abs().max()))


def test_argmax_ties_first_index(ops):
    x = torch.zeros(3, 1000)
    x[0, 17] = x[0, 500] = 2.0
    x[1, 999] = 1.0
    assert ops.argmax_rows(dev(x)).cpu().tolist() == [17, 999, 0]


@pytest.mark.parametrize("T,d", [(10, 32), (257, 512), (64, 768), (5, 1024), (9, 100), (33, 256)])
def test_add_layernorm(ops, T, d):
    from consistent__style_transfer_amd.ops import _ln_bwd, _ln_fwd
    x, res, g, b = rnd(T, d, seed=1), rnd(T, d, seed=2), 1 + 0.1 * rnd(d, seed=3), rnd(d, seed=4)
    drop = ops.Drop(0.1, 99, 1001)
    mask = torch.from_numpy(orng.dropout_mask(99, 1001, (T, d), 0.1))
    xr, rr, gr, br = (t.clone().requires_grad_(True) for t in (x, res, g, b))
    zr = rr + xr * mask
    yr = F.layer_norm(zr, (d,), gr, br, 1e-5)
    w = rnd(T, d, seed=5)
    (yr * w).sum().backward()
    z, y = torch.empty(T, d, device="cuda"), torch.empty(T, d, device="cuda")
    mean, rstd = torch.empty(T, device="cuda"), torch.empty(T, device="cuda")
    _ln_fwd(dev(x), dev(res), dev(g), dev(b), drop, z, y, mean, rstd)
    close(y, yr, 2e-4, 2e-5)
    close(z, zr, 1e-6, 1e-6)
    dz, dg, db = _ln_bwd(dev(w), z, mean, rstd, dev(g), True)
    close(dz, rr.grad, 5e-4, 5e-5)
    close(dg, gr.grad, 5e-4, 5e-4)
    close(db, br.grad, 5e-4, 5e-4)


def test_colsum_and_reduce(ops):
    from consistent__style_transfer_amd._lib import call
    for M, N in ((5, 7), (4608, 2048), (1000, 130), (1024, 1536), (300, 260), (9216, 512)):
        x = rnd(M, N, seed=1)
        close(ops.colsum(dev(x)), x.sum(0), 2e-4, 2e-3)
        acc = torch.ones(N, device="cuda")
        ops.colsum(dev(x), out=acc, accumulate=True)
        close(acc, 1 + x.sum(0), 2e-4, 2e-3)
    out = torch.zeros(1, device="cuda")
    call("cst_reduce_sum", dev(x.reshape(-1)), x.numel(), 0.5, out, 0)
    close(out, (0.5 * x.sum()).reshape(1), 1e-4, 1e-2)


# ----------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,S,H,hd,p", [(2, 6, 4, 8, 0.0), (3, 13, 2, 32, 0.1), (2, 36, 8, 64, 0.0), (2, 36, 8, 64, 0.1),
                                        (1, 60, 8, 64, 0.0), (2, 18, 8, 96, 0.1), (1, 64, 2, 64, 0.0),
                                        (2, 1, 2, 16, 0.0), (2, 17, 4, 16, 0.1), (2, 48, 2, 32, 0.1), (3, 33, 2, 64, 0.1),
                                        # 64 < S <= 128: the query-blocked kernels of attention_long.hip (book corpus: the Matcher
                                        # attends over cat(x1, x2), 60 positions before transfer_noise, ~80 after)
                                        (2, 65, 2, 64, 0.0), (2, 80, 8, 64, 0.1), (1, 96, 2, 64, 0.1), (2, 97, 2, 64, 0.0),
                                        (1, 128, 2, 64, 0.1), (2, 72, 4, 8, 0.1), (1, 128, 4, 8, 0.0), (2, 80, 2, 96, 0.1),
                                        (1, 128, 2, 96, 0.1)])
def test_mha(ops, B, S, H, hd, p):
    from consistent__style_transfer_amd._lib import call
    d = H * hd
    qkv = rnd(B, S, 3 * d, seed=1, scale=0.7)
    q, k, v = (t.reshape(B, S, H, hd).permute(0, 2, 1, 3) for t in qkv.split(d, -1))
    qkv_r = qkv.clone().requires_grad_(True)
    qr, kr, vr = (t.reshape(B, S, H, hd).permute(0, 2, 1, 3) for t in qkv_r.split(d, -1))
    att = torch.softmax(qr @ kr.transpose(-1, -2) / math.sqrt(hd), -1)
    if p > 0:
        att = att * torch.from_numpy(orng.dropout_mask(7, 1000, (B, H, S, S), p))
    ref = (att @ vr).permute(0, 2, 1, 3).reshape(B, S, d)
    w = rnd(B, S, d, seed=2)
    (ref * w).sum().backward()
    drop = ops.Drop(p, 7, 1000)
    out, lse = torch.empty(B * S, d, device="cuda"), torch.empty(B * H * S, device="cuda")
    call("cst_mha_fwd", dev(qkv), out, lse, B, S, H, hd, *drop.args())
    close(out.view(B, S, d), ref, 3e-4, 3e-5)
    dqkv = torch.empty(B * S, 3 * d, device="cuda")
    call("cst_mha_bwd", dev(qkv), dev(w), lse, dqkv, B, S, H, hd, *drop.args())
    close(dqkv.view(B, S, 3 * d), qkv_r.grad, 1e-3, 1e-4)
    # the bf16 twins written next to the fp32 results (operands of the out-projection / in-projection products)
    outb = torch.zeros(B * S, d, device="cuda", dtype=torch.int16)
    dqb = torch.zeros(B * S, 3 * d, device="cuda", dtype=torch.int16)
    out2, dq2 = torch.empty_like(out), torch.empty_like(dqkv)
    call("cst_mha_fwd_b", dev(qkv), out2, lse, B, S, H, hd, *drop.args(), outb, d)
    call("cst_mha_bwd_b", dev(qkv), dev(w), lse, dq2, B, S, H, hd, *drop.args(), dqb, 3 * d)
    assert torch.equal(out2, out) and torch.equal(dq2, dqkv)
    assert torch.equal(outb.view(torch.bfloat16), out.to(torch.bfloat16))
    assert torch.equal(dqb.view(torch.bfloat16), dqkv.to(torch.bfloat16))


@pytest.mark.parametrize("B,S,H,hd,p", [(2, 36, 8, 64, 0.1), (3, 18, 8, 96, 0.1), (1, 64, 2, 64, 0.0), (2, 36, 8, 96, 0.0), (2, 7, 3, 96, 0.1)])
def test_mha_bf16_io_equals_fp32_io_on_rounded_inputs(ops, B, S, H, hd, p):
    """cst_mha_fwd_h / cst_mha_bwd_h: bf16 qkv and d(output) in HBM, bf16 LDS images, every product on the bf16 matrix pipe (Q K^T / dO V^T:
    exact products of bf16 values; Pd V, dS K, dS^T Q, Pd^T dO: Pd / dS rounded to bf16 once): the fp32-I/O kernels run on the bf16-rounded
    values give the same results up to that one rounding, with or without the optional fp32 results."""
    from consistent__style_transfer_amd._lib import call
    d = H * hd
    qkv = _bf16_round(rnd(B * S, 3 * d, seed=1, scale=0.7))
    w = _bf16_round(rnd(B * S, d, seed=2))
    qb = ops.cast_bf16(dev(qkv), want_t=False)[0][:, :3 * d].contiguous()        # dense [B*S, 3d]: the _h kernels take no leading dimension
    wb = ops.cast_bf16(dev(w), want_t=False)[0][:, :d].contiguous()
    drop = ops.Drop(p, 7, 1000)
    out, lse = torch.empty(B * S, d, device="cuda"), torch.empty(B * H * S, device="cuda")
    outb = torch.zeros(B * S, d, device="cuda", dtype=torch.int16)
    call("cst_mha_fwd_b", dev(qkv), out, lse, B, S, H, hd, *drop.args(), outb, d)
    out2, lse2 = torch.empty_like(out), torch.empty_like(lse)
    outb2, outb3 = torch.zeros_like(outb), torch.zeros_like(outb)
    call("cst_mha_fwd_h", qb, out2, lse2, B, S, H, hd, *drop.args(), outb2, d)
    call("cst_mha_fwd_h", qb, None, lse2, B, S, H, hd, *drop.args(), outb3, d)          # bf16 result only
    # bf16 LDS images, Q K^T on the bf16 matrix pipe: exact products, another summation order (a few ulps: lse); round 3: Pd is rounded to
    # bf16 for Pd V on the bf16 pipe, so the output carries one bf16 rounding of a factor (relative L2 <= 3e-3)
    close(lse2, lse, 2e-6, 2e-6)
    relf = ((out2 - out).norm() / out.norm()).item()
    assert relf < 3e-3, relf
    close(out2, out, 2.0 ** -6, 2.0 ** -7 * out.abs().max().item())
    assert torch.equal(outb3, outb2)
    assert torch.equal(outb2.view(torch.bfloat16).float().cpu(), _bf(out2.cpu()))
    dq = torch.empty(B * S, 3 * d, device="cuda")
    dqb = torch.zeros(B * S, 3 * d, device="cuda", dtype=torch.int16)
    call("cst_mha_bwd_b", dev(qkv), dev(w), lse, dq, B, S, H, hd, *drop.args(), dqb, 3 * d)
    dq2, dqb2, dqb3 = torch.empty_like(dq), torch.zeros_like(dqb), torch.zeros_like(dqb)
    call("cst_mha_bwd_h", qb, wb, lse, dq2, B, S, H, hd, *drop.args(), dqb2, 3 * d)
    call("cst_mha_bwd_h", qb, wb, lse, None, B, S, H, hd, *drop.args(), dqb3, 3 * d)
    # backward: bf16 LDS images, Q K^T and dO V^T on the bf16 matrix pipe (exact products, another summation order); round 3: dS and Pd are
    # rounded to bf16 for the three output products (dS K, dS^T Q, Pd^T dO on the bf16 pipe), so the gradients carry one bf16 rounding of a
    # factor: relative L2 error <= 3e-3 per tensor, element-wise within 2^-7 of the row scale (measured 1.6e-3 / well inside)
    rel = ((dq2 - dq).norm() / dq.norm()).item()
    assert rel < 3e-3, rel
    close(dq2, dq, 2.0 ** -6, 2.0 ** -7 * dq.abs().max().item())
    assert torch.equal(dqb3, dqb2)
    assert torch.equal(dqb2.view(torch.bfloat16).float().cpu(), _bf(dq2.cpu()))         # the bf16 twin is the rounding of the fp32 result
    with pytest.raises(RuntimeError, match="head dims 64 / 96"):
        call("cst_mha_fwd_h", qb, None, lse2, B, S, H * hd // 32, 32, *drop.args(), outb3, d)


def test_zero_arena_hands_out_zeroed_disjoint_slices(ops):
    """ops.zero_arena: the first scope of a tag measures (every buffer gets its own fill), later scopes zero the measured prefix with one
    kernel and hand out disjoint slices of it -- also after the previous scope's users dirtied them; outside a scope nothing changes."""
    dev = torch.device("cuda")
    tag = ("test", 1)
    x = rnd(300, 70, seed=3)

    def body():
        a = ops.zeros(5, 7, device=dev)
        b = ops.zeros(1000, device=dev, dtype=torch.int64)
        c = ops.colsum(dev_(x))
        cb = ops.colsum_bf16(ops.cast_bf16(dev_(x), want_t=False)[0], 70)
        return a, b, c, cb

    dev_ = lambda t: t.to(dev)
    with ops.zero_arena(tag, dev):
        a0, b0, c0, cb0 = body()
    st = ops._ARENA[ops._devkey(dev)]
    assert st["hw"][tag] > 0 and not st["active"]
    base, end = st["buf"].data_ptr(), st["buf"].data_ptr() + st["buf"].numel() * 4
    assert not (base <= a0.data_ptr() < end)                                              # measuring pass: own buffers
    st["buf"][:st["hw"][tag]].fill_(0x7F7F7F7F)                                           # dirty the prefix
    with ops.zero_arena(tag, dev):
        a1, b1, c1, cb1 = body()
        ptrs = sorted((t.data_ptr(), t.numel() * t.element_size()) for t in (a1, b1, c1, cb1))
        assert all(base <= p < end for p, _ in ptrs)
        assert all(p0 + n0 <= p1 for (p0, n0), (p1, _) in zip(ptrs, ptrs[1:]))           # disjoint
        assert (a1 == 0).all() and (b1 == 0).all()
        close(c1, x.sum(0), 1e-5, 1e-4)
        close(cb1, _bf16_round(x).sum(0), 1e-5, 1e-3)
    assert torch.equal(c1.cpu(), c0.cpu()) or torch.allclose(c1, c0, rtol=1e-5, atol=1e-5)
    t = ops.zeros(4, 4, device=dev)
    assert not (base <= t.data_ptr() < end) and (t == 0).all()


def test_mha_rejects_unsupported_lengths(ops):
    from consistent__style_transfer_amd._lib import call
    B, S, H, hd = 1, 129, 2, 64
    qkv = torch.zeros(B * S, 3 * H * hd, device="cuda")
    with pytest.raises(RuntimeError, match="S=129 unsupported"):
        call("cst_mha_fwd", qkv, torch.empty(B * S, H * hd, device="cuda"), torch.empty(B * H * S, device="cuda"), B, S, H, hd, *ops.NO_DROP.args())
    with pytest.raises(RuntimeError, match="head dim 32 unsupported for S > 64"):
        call("cst_mha_fwd", torch.zeros(100, 3 * 64, device="cuda"), torch.empty(100, 64, device="cuda"), torch.empty(2 * 100, device="cuda"),
             1, 100, 2, 32, *ops.NO_DROP.args())


@pytest.mark.parametrize("B,L,D", [(3, 5, 32), (7, 18, 512), (2, 30, 512), (2, 64, 96)])
def test_dot_attn(ops, B, L, D):
    from consistent__style_transfer_amd._lib import call
    q, mem, w = rnd(B, D, seed=1), rnd(B, L, D, seed=2), rnd(B, D, seed=3)
    qr, mr = q.clone().requires_grad_(True), mem.clone().requires_grad_(True)
    a = torch.softmax(torch.einsum("bd,bld->bl", qr, mr) / math.sqrt(D), -1)
    ref = torch.einsum("bl,bld->bd", a, mr)
    (ref * w).sum().backward()
    qd = dev(torch.cat([q, torch.zeros(B, 11)], 1))              # strided query rows
    out, pr = torch.empty(B, D + 5, device="cuda"), torch.empty(B, L, device="cuda")
    dropped = torch.empty(B, 2 * D + 3, device="cuda")
    db = torch.zeros(B, 2 * D + 6, device="cuda", dtype=torch.int16)
    call("cst_dot_attn_fwd", qd, D + 11, dev(mem), out, D + 5, pr, B, L, D, dropped, 2 * D + 3, db, 2 * D + 6, 0.25, 9, 105, None)
    close(out[:, :D], ref, 3e-4, 3e-5)
    mask = torch.from_numpy(orng.dropout_mask(9, 105, (B, 2 * D), 0.25))
    close(dropped[:, :2 * D], torch.cat([q, ref.detach()], 1) * mask, 3e-4, 3e-5)
    close(db.view(torch.bfloat16).float()[:, :2 * D], torch.cat([q, ref.detach()], 1) * mask, 1e-2, 1e-2)
    close(pr, a, 3e-4, 1e-6)
    dq, dmem = torch.ones(B, D, device="cuda"), torch.zeros(B, L, D, device="cuda")
    call("cst_dot_attn_bwd", dev(w), D, qd, D + 11, dev(mem), pr, dq, D, 1, dmem, B, L, D)
    close(dq, 1 + qr.grad, 5e-4, 5e-5)
    close(dmem, mr.grad, 5e-4, 5e-5)


# --------------------------------------------------------------------------------- LSTM cell
def test_lstm_cell(ops):
    from consistent__style_transfer_amd.gen_fn import _cell_bwd, _cell_fwd
    B, H = 5, 24
    g, c = rnd(B, 4 * H, seed=1), rnd(B, H, seed=2)
    gr, cr = g.clone().requires_grad_(True), c.clone().requires_grad_(True)
    i, f, gg, o = gr.split(H, 1)
    c2 = torch.sigmoid(f) * cr + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    wh, wc = rnd(B, H, seed=3), rnd(B, H, seed=4)
    ((h2 * wh).sum() + (c2 * wc).sum()).backward()
    gd, cd = dev(g), dev(c)
    h_out, c_out, h2nd = torch.empty(B, H + 3, device="cuda"), torch.empty(B, H, device="cuda"), torch.empty(B, H, device="cuda")
    _cell_fwd(gd, cd, h_out[:, :H], c_out, h2nd, B, H)
    close(h_out[:, :H], h2, 2e-5, 2e-6)
    close(h2nd, h2, 2e-5, 2e-6)
    close(c_out, c2, 2e-5, 2e-6)
    dg, dcp = torch.empty(B, 4 * H, device="cuda"), torch.empty(B, H, device="cuda")
    half = dev(wh * 0.25)
    _cell_bwd(gd, cd, c_out, dev(wh * 0.75), half, dev(wc), dg, dcp, B, H)
    close(dg, gr.grad, 2e-4, 2e-5)
    close(dcp, cr.grad, 2e-4, 2e-5)


# ---------------------------------------------------------------------- embedding / conv / pool
def test_embed_gather_scatter(ops):
    V, E, R = 37, 12, 20
    tab = rnd(V, E, seed=1)
    ida = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(1))
    idb = torch.randint(0, V, (R, 3), generator=torch.Generator().manual_seed(2))
    d = ops.Drop(0.2, 5, 201)
    mask = torch.from_numpy(orng.dropout_mask(5, 201, (R, E), 0.2))
    for coin in (0, 1):
        out = torch.empty(R, E, device="cuda")
        cd = torch.tensor([coin], dtype=torch.int32, device="cuda")
        ops.embed_gather(dev(tab), out, ids_a=dev(ida), ids_b=dev(idb)[:, 1], ldb=3, coin=cd, drop=d)
        sel = ida if coin else idb[:, 1]
        close(out, tab[sel] * mask, 1e-6, 1e-7)
        dt = torch.zeros(V, E, device="cuda")
        g = rnd(R, E, seed=3)
        ops.embed_scatter_add(dt, dev(g), ids_a=dev(ida), ids_b=dev(idb)[:, 1], ldb=3, coin=cd, drop=d)
        ref = torch.zeros(V, E).index_add_(0, sel, g * mask)
        close(dt, ref, 1e-5, 1e-6)
    # transposed table (columns of an [E,V] Linear weight)
    W = rnd(E, V, seed=4)
    out = torch.empty(R, E, device="cuda")
    ops.embed_gather(dev(W), out, ids_a=dev(ida), transposed=True)
    close(out, W.t()[ida], 1e-6, 1e-7)


def test_tps_embed(ops):
    B, L1, L2, d, V = 3, 5, 4, 16, 29
    Et, Ep, Es = (rnd(V, d, seed=1).requires_grad_(True), rnd(100, d, seed=2).requires_grad_(True),
                  rnd(2, d, seed=3).requires_grad_(True))
    x1 = torch.randint(0, V, (B, L1), generator=torch.Generator().manual_seed(3))
    p2 = torch.softmax(rnd(B, L2, V, seed=4), -1).requires_grad_(True)
    ref = torch.cat([Et[x1] + Ep[:L1] + Es[0], p2 @ Et + Ep[:L2] + Es[1]], 1)
    w = rnd(B, L1 + L2, d, seed=5)
    (ref * w).sum().backward()
    ops.set_precision("f32")
    Etg, Epg, Esg = (dev(t.detach()).requires_grad_(True) for t in (Et, Ep, Es))
    p2g = dev(p2.detach()).requires_grad_(True)
    out = ops.TpsEmbedFn.apply(dev(x1), p2g, Etg, Epg, Esg)
    (out * dev(w)).sum().backward()
    ops.set_precision("bf16")
    close(out, ref, 2e-4, 2e-5)
    for a, b in ((Etg, Et), (Epg, Ep), (Esg, Es), (p2g, p2)):
        close(a.grad, b.grad, 3e-4, 3e-5)


@pytest.mark.parametrize("mode,B,L,E,R,nf", [(0, 3, 7, 16, 4, 5), (1, 3, 7, 16, 4, 6),
                                              (1, 2, 7, 128, 16, 300),        # reference constants, fused kernels
                                              (1, 40, 18, 128, 16, 300),      # several workgroups, > 1 sample per gather workgroup
                                              (1, 5, 5, 32, 4, 13),           # T = 1 for k = 5, ragged filter tile
                                              (1, 2, 6, 24, 2, 7)])           # k * E/R > 40 for k >= 4: im2col fallback
def test_conv_bank(ops, mode, B, L, E, R, nf):
    ops.set_precision("f32")
    e = rnd(B, L, E, seed=1).requires_grad_(True)
    if mode == 0:
        convs = [torch.nn.Conv2d(1, nf, (k, E), padding=(k - 1, 0)) for k in (3, 4, 5)]
        ys = [F.relu(c(e.unsqueeze(1))).squeeze(3) for c in convs]
        ref = torch.cat([F.max_pool1d(y, y.size(2)).squeeze(2) for y in ys], 1)
    else:
        es = E // R
        convs = [torch.nn.Conv2d(1, nf, (f, es), stride=(1, es)) for f in (2, 3, 4, 5)]
        cons = [F.relu(c(e.unsqueeze(1))) for c in convs]
        pools = [F.max_pool2d(c, (c.size(2), 1)).squeeze(2) for c in cons]
        ref = torch.cat(pools, 1).permute(0, 2, 1).contiguous().view(B * R, -1)
    w = rnd(*ref.shape, seed=2)
    (ref * w).sum().backward()
    eg = dev(e.detach()).requires_grad_(True)
    wb = []
    for c in convs:
        wb += [dev(c.weight.detach()).requires_grad_(True), dev(c.bias.detach()).requires_grad_(True)]
    out = ops.ConvBankFn.apply(eg, mode, R if mode else 1, *wb)
    (out * dev(w)).sum().backward()
    ops.set_precision("bf16")
    close(out, ref, 2e-4, 2e-5)
    close(eg.grad, e.grad, 3e-4, 3e-5)
    for i, c in enumerate(convs):
        close(wb[2 * i].grad, c.weight.grad, 3e-4, 3e-5)
        close(wb[2 * i + 1].grad, c.bias.grad, 3e-4, 3e-5)


@pytest.mark.parametrize("train_w", [True, False])
def test_conv_bank_bf16_path(ops, train_w):
    """TextCNN at the reference constants (classifier.py:18,30: E = 128, 128 filters, k = 3, 4, 5) in bf16 mode: the window rows exist in
    bf16 only (cst_im2col_b), conv / dgrad / weight gradient run on the bf16 GEMMs (cst_seqmax_bwd_b feeds them), no fp32-staged
    product.  Against torch on operands rounded to bf16 (what both this path and the staged kernel multiply)."""
    ops.set_precision("bf16")
    B, L, E, nf = 64, 14, 128, 128                       # B (L + k - 1) = 1024 / 1088 / 1152: multiples of 64
    bf = lambda t: t.to(torch.bfloat16).float()
    e = bf(rnd(B, L, E, seed=1)).requires_grad_(True)
    convs = [torch.nn.Conv2d(1, nf, (k, E), padding=(k - 1, 0)) for k in (3, 4, 5)]
    for c in convs:
        c.weight.data = bf(c.weight.data)
    ys = [F.relu(c(e.unsqueeze(1))).squeeze(3) for c in convs]
    ref = torch.cat([F.max_pool1d(y, y.size(2)).squeeze(2) for y in ys], 1)
    w = rnd(*ref.shape, seed=2)
    (ref * w).sum().backward()
    eg = dev(e.detach()).requires_grad_(True)
    wb = []
    for c in convs:
        wb += [dev(c.weight.detach()).requires_grad_(train_w), dev(c.bias.detach()).requires_grad_(train_w)]
    names = []
    orig = ops.call
    ops.call = lambda nm, *a: (names.append(nm), orig(nm, *a))[1]
    try:
        out = ops.ConvBankFn.apply(eg, 0, 1, *wb)
        (out * dev(w)).sum().backward()
    finally:
        ops.call = orig
    assert "cst_gemm" not in names and "cst_im2col" not in names and names.count("cst_im2col_b") == 3, names
    close(out, ref, 2e-3, 2e-3)
    # gradients: dy is rounded to bf16 on this path (fp32 in torch)
    close(eg.grad, e.grad, 2e-2, 2e-2 * float(e.grad.abs().max()))
    if train_w:
        for i, c in enumerate(convs):
            close(wb[2 * i].grad, c.weight.grad, 2e-2, 2e-2 * float(c.weight.grad.abs().max()))
            close(wb[2 * i + 1].grad, c.bias.grad, 2e-2, 2e-2 * float(c.bias.grad.abs().max()))


def test_small_ops(ops):
    from consistent__style_transfer_amd._lib import call
    # highway
    h, pr = rnd(9, 33, seed=1).requires_grad_(True), rnd(9, 33, seed=2).requires_grad_(True)
    ref = torch.sigmoid(h) * F.relu(h) + (1 - torch.sigmoid(h)) * pr
    w = rnd(9, 33, seed=3)
    (ref * w).sum().backward()
    hg, pg = dev(h.detach()).requires_grad_(True), dev(pr.detach()).requires_grad_(True)
    out = ops.HighwayFn.apply(hg, pg)
    (out * dev(w)).sum().backward()
    close(out, ref, 1e-5, 1e-6)
    close(hg.grad, h.grad, 1e-4, 1e-6)
    close(pg.grad, pr.grad, 1e-4, 1e-6)
    # losses
    x, t = rnd(50, seed=4).requires_grad_(True), rnd(50, seed=5)
    for ref_fn, mine in ((lambda: F.mse_loss(x, t) * 0.5, lambda xg: ops.mse_loss(xg, dev(t), weight=0.5)),
                         (lambda: F.mse_loss(x, torch.full_like(x, 0.25)), lambda xg: ops.mse_loss(xg, None, 0.25)),
                         (lambda: F.binary_cross_entropy_with_logits(x, torch.ones_like(x)), lambda xg: ops.bce_logits_loss(xg, 1.0)),
                         (lambda: F.binary_cross_entropy_with_logits(x, torch.zeros_like(x)) * 2, lambda xg: ops.bce_logits_loss(xg, 0.0, 2.0))):
        x.grad = None
        r = ref_fn()
        (r * 3.0).backward()
        xg = dev(x.detach()).requires_grad_(True)
        l = mine(xg)
        (l * 3.0).sum().backward()
        close(l.reshape(()), r, 1e-5, 1e-6)
        close(xg.grad, x.grad, 1e-4, 1e-7)
    # seq max
    xs = rnd(4, 9, 20, seed=6).requires_grad_(True)
    r = xs.max(1).values
    (r * rnd(4, 20, seed=7)).sum().backward()
    xg = dev(xs.detach()).requires_grad_(True)
    o = ops.SeqMaxFn.apply(xg)
    (o * dev(rnd(4, 20, seed=7))).sum().backward()
    close(o, r, 0, 0)
    close(xg.grad, xs.grad, 0, 0)
    # dropout == oracle mask
    xx = rnd(13, 40, seed=8)
    y = ops.dropout2d(dev(xx), ops.Drop(0.5, 42, 2000))
    close(y, xx * torch.from_numpy(orng.dropout_mask(42, 2000, (13, 40), 0.5)), 1e-6, 1e-7)


def test_clip_and_adam(ops):
    from consistent__style_transfer_amd._lib import call
    n = 5000
    p0, g0 = rnd(n, seed=1), rnd(n, seed=2) * 3
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3)
    p, m, v = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    for it in range(3):
        pr.grad = g0.clone() * (it + 1)
        torch.nn.utils.clip_grad_norm_([pr], 5.0)
        opt.step()
        g = dev(g0 * (it + 1))
        ss, part = torch.zeros(1, device="cuda"), torch.zeros(1024, device="cuda")
        call("cst_sumsq_accumulate", g, n, ss, part)
        call("cst_clip_scale", g, n, ss, 5.0)
        close(g, pr.grad, 1e-5, 1e-6)
        call("cst_add_i32", step, 1)
        call("cst_adam_step", p, g, m, v, n, 1e-3, 0.9, 0.999, 1e-8, step)
        close(p, pr, 1e-5, 1e-6)


def test_sumsq_is_bitwise_reproducible(ops):
    """The gradient norm feeds the clip coefficient of every parameter: it must not depend on block arrival order
    (data-parallel replicas compare their parameters bit for bit, parallel.check_replicas)."""
    from consistent__style_transfer_amd._lib import call
    g = torch.randn(9_294_464, device="cuda")                    # the generator's parameter count
    part = torch.zeros(1024, device="cuda")
    vals = set()
    for _ in range(20):
        ss = torch.zeros(1, device="cuda")
        call("cst_sumsq_accumulate", g, g.numel(), ss, part)
        vals.add(ss.item())
    assert len(vals) == 1, vals
    np.testing.assert_allclose(vals.pop(), float((g.double() ** 2).sum()), rtol=1e-5)


# ------------------------------------------------------------- fused recurrent-step entry points
@pytest.mark.parametrize("B,T,L,D,p", [(3, 4, 5, 32, 0.0), (5, 6, 7, 64, 0.25), (2, 18, 18, 512, 0.1)])
def test_dot_attn_bwd_steps_equals_per_step_calls(ops, B, T, L, D, p):
    """All decode steps in one launch == T launches of cst_dropout + cst_dot_attn_bwd (the path it replaces)."""
    from consistent__style_transfer_amd._lib import call
    W = 2 * D
    g = dev(rnd(B, T * W, seed=1))
    q = dev(rnd(B, T * W, seed=2))                                   # h_s sits in the first D columns of each step's slot
    mem = dev(rnd(B, L, D, seed=3))
    pr = torch.softmax(dev(rnd(T, B, L, seed=4)), -1).contiguous()
    ref_g, ref_dm = g.clone(), torch.zeros(B, L, D, device="cuda")
    for s in range(T):
        sl = ref_g[:, s * W:(s + 1) * W]
        if p > 0:
            ops.dropout2d(sl, ops.Drop(p, 11, 300 + s), out=sl)
        call("cst_dot_attn_bwd", sl[:, D:], T * W, q[:, s * W:s * W + D], T * W, mem, pr[s], sl[:, :D], T * W, 1, ref_dm, B, L, D)
    got_g, got_dm = g.clone(), torch.zeros(B, L, D, device="cuda")
    call("cst_dot_attn_bwd_steps", got_g, T * W, W, q, T * W, W, mem, pr, got_dm, B, T, L, D, *ops.Drop(p, 11, 300).args())
    close(got_g.view(B, T, W)[:, :, :D], ref_g.view(B, T, W)[:, :, :D], 2e-5, 2e-6)
    close(got_dm, ref_dm, 2e-5, 2e-5)


@pytest.mark.parametrize("pair", [False, True])
@pytest.mark.parametrize("B,H,K,splitk", [(5, 16, 64, 0), (256, 256, 256, 2), (256, 512, 640, 0)])
def test_gemm_bf16_lstm_equals_gemm_plus_cell(ops, B, H, K, splitk, pair):
    from consistent__style_transfer_amd._lib import call
    np_ = 2 if pair else 1
    probs, refs = [], []
    for i in range(np_):
        A, Wt = rnd(B, K, seed=10 + i, scale=0.3), rnd(4 * H, K, seed=20 + i, scale=0.3)
        Ab, Wb = ops.cast_bf16(dev(A), want_t=False)[0], ops.cast_bf16(dev(Wt), want_t=False)[0]
        bias, add, c_prev = dev(rnd(4 * H, seed=30 + i)), dev(rnd(B, 4 * H + 8, seed=40 + i)), dev(rnd(B, H, seed=50 + i))
        # reference: the two-kernel path
        g = torch.empty(B, 4 * H, device="cuda")
        ops.gemm_bf16(Ab, Wb, B, 4 * H, C=g, bias=bias, addend=add[:, :4 * H])
        h, c, h2 = (torch.empty(B, H, device="cuda") for _ in range(3))
        hb = torch.zeros(B, H, device="cuda", dtype=torch.int16)
        call("cst_lstm_cell_fwd", g, 4 * H, c_prev, H, h, H, c, H, h2, H, hb, H, None, 0, B, H)
        refs.append((g, h, c, h2, hb))
        probs.append(dict(Ab=Ab, Bb=Wb, gates=torch.empty(B, 4 * H, device="cuda"), c_prev=c_prev, h_out=torch.empty(B, H, device="cuda"),
                          c_out=torch.empty(B, H, device="cuda"), h_out2=torch.empty(B, H, device="cuda"), bias=bias, addend=add[:, :4 * H],
                          hb=torch.zeros(B, H, device="cuda", dtype=torch.int16)))
    from consistent__style_transfer_amd import gen_fn
    old = ops.LSTM_SPLITK
    ops.LSTM_SPLITK = splitk
    try:
        gen_fn._gemm_cell_fwd(probs, B, H)
    finally:
        ops.LSTM_SPLITK = old
    for pr, (g, h, c, h2, hb) in zip(probs, refs):
        close(pr["gates"], g, 1e-5, 1e-5)
        close(pr["h_out"], h, 1e-5, 1e-5)
        close(pr["c_out"], c, 1e-5, 1e-5)
        close(pr["h_out2"], h2, 1e-5, 1e-5)
        close(pr["hb"].view(torch.bfloat16).float(), hb.view(torch.bfloat16).float(), 1e-2, 1e-2)


@pytest.mark.parametrize("B,H,n_extra,pair", [(5, 16, 0, False), (256, 256, 0, True), (256, 512, 128, False)])
def test_gemm_bf16_lstm_bwd_equals_gemm_plus_cell_bwd(ops, B, H, n_extra, pair):
    from consistent__style_transfer_amd import gen_fn
    from consistent__style_transfer_amd._lib import call
    N, K = n_extra + H, 4 * H
    probs, refs = [], []
    for i in range(2 if pair else 1):
        dgn, Wt = rnd(B, K, seed=60 + i, scale=0.3), rnd(N, K, seed=70 + i, scale=0.3)
        Ab, Wb = ops.cast_bf16(dev(dgn), want_t=False)[0], ops.cast_bf16(dev(Wt), want_t=False)[0]
        gates = torch.sigmoid(dev(rnd(B, 4 * H, seed=80 + i)))
        c_prev, c_new, dh_x, dc_in = (dev(rnd(B, H, seed=90 + 4 * i + j)) for j in range(4))
        full = torch.empty(B, N, device="cuda")
        ops.gemm_bf16(Ab, Wb, B, N, C=full)
        dg, dcp = torch.empty(B, 4 * H, device="cuda"), torch.empty(B, H, device="cuda")
        dgb = torch.zeros(B, 4 * H, device="cuda", dtype=torch.int16)
        dh = full[:, n_extra:].contiguous()
        call("cst_lstm_cell_bwd", gates, 4 * H, c_prev, H, c_new, H, dh_x, H, dh, H, dc_in, H, dg, 4 * H, dcp, H, dgb, 4 * H, B, H)
        refs.append((full, dg, dcp, dgb))
        probs.append(dict(Ab=Ab, Bb=Wb, gates=gates, c_prev=c_prev, c_new=c_new, dh_extra=dh_x, dc_in=dc_in,
                          dgates=torch.empty(B, 4 * H, device="cuda"), dc_prev=torch.empty(B, H, device="cuda"),
                          dgb=torch.zeros(B, 4 * H, device="cuda", dtype=torch.int16), n_extra=n_extra,
                          extra_out=torch.empty(B, n_extra + 4, device="cuda")[:, :n_extra] if n_extra else None))
    gen_fn._gemm_cell_bwd(probs, B, H)
    for pr, (full, dg, dcp, dgb) in zip(probs, refs):
        close(pr["dgates"], dg, 2e-5, 2e-5)
        close(pr["dc_prev"], dcp, 2e-5, 2e-5)
        close(pr["dgb"].view(torch.bfloat16).float(), dgb.view(torch.bfloat16).float(), 1e-2, 1e-2)
        if n_extra:
            close(pr["extra_out"], full[:, :n_extra], 2e-5, 2e-5)


@pytest.mark.parametrize("K,M,N,splitk", [(64, 8, 8, 1), (128, 200, 136, 1), (4608, 512, 2048, 0), (9216, 1536, 512, 0),
                                          (4608, 512, 512, 4), (1024, 2048, 512, 1)])
def test_gemm_bf16_tt(ops, K, M, N, splitk):
    """C = A^T B from operands whose row index is the contraction index (hardware transposed LDS reads)."""
    A, Bm = rnd(K, M, seed=1, scale=0.5), rnd(K, N, seed=2, scale=0.5)
    Ab, Bb = ops.cast_bf16(dev(A), want_t=False)[0], ops.cast_bf16(dev(Bm), want_t=False)[0]
    Af = Ab.view(torch.bfloat16).float()[:, :M].cpu().double()
    Bf = Bb.view(torch.bfloat16).float()[:, :N].cpu().double()
    ref = (Af.T @ Bf).float()
    got = ops.gemm_bf16_tt(Ab, Bb, M, N, splitk=splitk)
    close(got, ref, 2e-4, 2e-4 * math.sqrt(K))
    acc = torch.ones(M, N, device="cuda")
    ops.gemm_bf16_tt(Ab, Bb, M, N, C=acc, accumulate=True, splitk=splitk)
    close(acc, 1 + ref, 2e-4, 2e-4 * math.sqrt(K))


def test_gemm_bf16_tt_group_equals_whole_k_products_bit_for_bit(ops):
    """cst_gemm_bf16_tt_group_*: the products recorded inside `with ops.tt_group()` are launched as one kernel whose workgroups each run
    the whole contraction of one 128 x 128 output tile -- the same workgroup body as a single product with splitk = 1, so every output
    must equal that launch bit for bit (ragged M / N, an accumulating output, different K per problem, more problems than one launch
    holds), and the fp32 reference to bf16-operand accuracy."""
    shapes = [(768, 2048, 1152), (2048, 768, 1152), (776, 768, 1152), (2304, 520, 1152), (256, 384, 576),
              (768, 768, 1152), (128, 2048, 1152), (1024, 136, 1152), (2048, 2048, 576), (896, 384, 576)]       # 10 > CST_TT_GROUP_MAX = 8
    prob = []
    for i, (M, N, K) in enumerate(shapes):
        Ab = ops.cast_bf16(dev(rnd(K, M, seed=10 + i, scale=0.5)), want_t=False)[0]
        Bb = ops.cast_bf16(dev(rnd(K, N, seed=40 + i, scale=0.5)), want_t=False)[0]
        prob.append((Ab, Bb, M, N, K))
    single = []
    for i, (Ab, Bb, M, N, K) in enumerate(prob):
        C = torch.full((M, N), 0.5, device="cuda") if i == 2 else torch.empty(M, N, device="cuda")
        single.append(ops.gemm_bf16_tt(Ab, Bb, M, N, C=C, accumulate=i == 2, splitk=1))
    outs = [torch.full((M, N), 0.5, device="cuda") if i == 2 else torch.full((M, N), float("nan"), device="cuda") for i, (_, _, M, N, _) in enumerate(prob)]
    with ops.tt_group():
        for i, (Ab, Bb, M, N, K) in enumerate(prob):
            ops.gemm_bf16_tt(Ab, Bb, M, N, C=outs[i], accumulate=i == 2)
    for i, (Ab, Bb, M, N, K) in enumerate(prob):
        assert torch.equal(outs[i], single[i]), f"problem {i} ({M}x{N}x{K}): grouped launch differs from the whole-K single launch"
        ref = (Ab.view(torch.bfloat16).float()[:, :M].cpu().double().T @ Bb.view(torch.bfloat16).float()[:, :N].cpu().double()).float()
        close(outs[i], ref + (0.5 if i == 2 else 0.0), 2e-4, 2e-4 * math.sqrt(K))
    # a group too small to fill the chip is launched product by product: the default (split-K) result, bit for bit
    few = [prob[0], prob[4]]                                  # 96 + 6 tiles
    dflt = [ops.gemm_bf16_tt(Ab, Bb, M, N) for Ab, Bb, M, N, K in few]
    got = [torch.empty(M, N, device="cuda") for _, _, M, N, _ in few]
    with ops.tt_group():
        for (Ab, Bb, M, N, K), C in zip(few, got):
            ops.gemm_bf16_tt(Ab, Bb, M, N, C=C)
    for a, b in zip(got, dflt):
        assert torch.equal(a, b)
    # the group is closed again after the block, and a second begin inside an open group is refused
    with ops.tt_group():
        with pytest.raises(RuntimeError, match="already open"):
            ops.call("cst_gemm_bf16_tt_group_begin", None, 0)
    with pytest.raises(RuntimeError, match="no group is open"):
        ops.call("cst_gemm_bf16_tt_group_end")


def test_gemm_bf16_tt_group_split_contraction_sums_partials_in_split_order(ops):
    """The weight gradients of one d = 512 encoder layer are 192 tiles: the grouped launch splits every contraction S ways (S workgroups a
    tile; the one that finishes last adds the partial tiles in split order and leaves the tile's counter at zero).  Whatever S the rule
    picks, every output must equal, bit for bit, the fp32 sum in split order of the whole-K products over the S row ranges -- on every
    one of several launches (the counters are reused), with an accumulating output among them."""
    from consistent__style_transfer_amd._lib import call_plain
    d, F, K = 512, 2048, 4608
    shapes = [(d, F), (F, d), (d, d), (3 * d, d)]
    prob = []
    for i, (M, N) in enumerate(shapes):
        prob.append((ops.cast_bf16(dev(rnd(K, M, seed=70 + i, scale=0.5)), want_t=False)[0], ops.cast_bf16(dev(rnd(K, N, seed=80 + i, scale=0.5)), want_t=False)[0], M, N))
    outs = None
    for rep in range(3):
        outs = [torch.full((M, N), 0.25, device="cuda") if i == 1 else torch.full((M, N), float("nan"), device="cuda") for i, (_, _, M, N) in enumerate(prob)]
        with ops.tt_group():
            for i, (Ab, Bb, M, N) in enumerate(prob):
                ops.gemm_bf16_tt(Ab, Bb, M, N, C=outs[i], accumulate=i == 1)
        S = call_plain("cst_gemm_bf16_tt_group_last_splits")
        assert S >= 2, f"expected a split contraction for 192 tiles (got {S}); CST_TT_GROUP_SPLITS / CST_TT_GROUP_MIN set?"
        kps = -(-(-(-K // S)) // 64) * 64
        for i, (Ab, Bb, M, N) in enumerate(prob):
            want = None
            for s_ in range(S):
                k0, k1 = s_ * kps, min(K, (s_ + 1) * kps)
                part = ops.gemm_bf16_tt(Ab[k0:k1], Bb[k0:k1], M, N, splitk=1)
                want = part if want is None else want + part
            if i == 1:
                want = want + 0.25
            assert torch.equal(outs[i], want), f"launch {rep}, problem {i}: the {S}-way split sum is not the in-order sum of its partial products"
    ws = ops._workspace(outs[0].device)
    assert int((ws[ops.WS_FLOATS:] != 0).sum()) == 0, "the tile counters were not left at zero"


def test_encoder_layer_bf16_tt_weight_grads_match_exact_mode(ops):
    """Token count % 64 == 0 switches the bf16 encoder layer to the transposed-read weight-gradient GEMM; its
    gradients must agree with the exact-fp32 layer to bf16 accuracy (relative Frobenius error)."""
    from consistent__style_transfer_amd.model._common import EncoderLayerParams
    B, S, d, H, F = 4, 16, 64, 4, 128
    torch.manual_seed(3)
    lay = EncoderLayerParams(d, F).cuda()
    x0 = dev(rnd(B * S, d, seed=5))
    w = dev(rnd(B * S, d, seed=6))
    res = {}
    for mode, fn in (("f32", ops.EncoderLayerFn), ("bf16", ops.EncoderLayerBf16Fn)):
        ops.set_precision(mode)
        for p in lay.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y = fn.apply(x, *lay.flat(), B, S, H, ops.NO_DROP, 0)
        (y * w).sum().backward()
        res[mode] = [y.detach(), x.grad] + [p.grad.clone() for p in lay.parameters()]
    ops.set_precision("bf16")
    for a, b in zip(res["bf16"], res["f32"]):
        err = (a - b).norm() / (b.norm() + 1e-12)
        assert err < 3e-2, float(err)


@pytest.mark.parametrize("B,H,K,L,p", [(5, 64, 128, 7, 0.0), (256, 512, 640, 18, 0.1)])
def test_gemm_bf16_lstm_attn_equals_cell_then_attention(ops, B, H, K, L, p):
    """Decoder step front half: one fused second launch == cst_gemm_bf16_lstm + cst_dot_attn_fwd."""
    from consistent__style_transfer_amd import gen_fn
    from consistent__style_transfer_amd._lib import call
    A, Wt = rnd(B, K, seed=1, scale=0.3), rnd(4 * H, K, seed=2, scale=0.3)
    Ab, Wb = ops.cast_bf16(dev(A), want_t=False)[0], ops.cast_bf16(dev(Wt), want_t=False)[0]
    bias, c_prev, mem = dev(rnd(4 * H, seed=3)), dev(rnd(B, H, seed=4)), dev(rnd(B, L, H, seed=5))
    drop = ops.Drop(p, 21, 407)

    def bufs():
        return dict(gates=torch.empty(B, 4 * H, device="cuda"), h=torch.empty(B, 2 * H + 4, device="cuda"), c=torch.empty(B, H, device="cuda"),
                    h2=torch.empty(B, H, device="cuda"), hb2=torch.zeros(B, H, device="cuda", dtype=torch.int16),
                    pr=torch.empty(B, L, device="cuda"), dr=torch.empty(B, 2 * H, device="cuda"),
                    db=torch.zeros(B, 2 * H, device="cuda", dtype=torch.int16))
    r, f = bufs(), bufs()
    gen_fn._gemm_cell_fwd([dict(Ab=Ab, Bb=Wb, gates=r["gates"], c_prev=c_prev, h_out=r["h"][:, :H], c_out=r["c"], h_out2=r["h2"], bias=bias,
                                hb2=r["hb2"])], B, H)
    call("cst_dot_attn_fwd", r["h"][:, :H], 2 * H + 4, mem, r["h"][:, H:2 * H], 2 * H + 4, r["pr"], B, L, H, r["dr"], 2 * H, r["db"], 2 * H, *drop.args())
    call("cst_gemm_bf16_lstm_attn", Ab, Ab.stride(0), Wb, Wb.stride(0), B, H, Ab.shape[1], bias, f["gates"], 4 * H, c_prev, H,
         f["h"][:, :H], 2 * H + 4, f["c"], H, f["h2"], H, f["hb2"], H, mem, L, f["h"][:, H:2 * H], 2 * H + 4, f["pr"],
         f["dr"], 2 * H, f["db"], 2 * H, *drop.args(), ops.LSTM_SPLITK, ops._workspace(Ab.device), ops.WS_FLOATS)
    for k in ("gates", "c", "h2", "pr", "dr"):
        close(f[k], r[k], 2e-5, 2e-5, k)
    close(f["h"][:, :2 * H], r["h"][:, :2 * H], 2e-5, 2e-5)
    for k in ("hb2", "db"):
        close(f[k].view(torch.bfloat16).float(), r[k].view(torch.bfloat16).float(), 1e-2, 1e-2, k)


def test_token_ce_bf16_twin(ops):
    """cst_token_ce_b: the bf16 gradient equals bf16(dlogits) and its K padding is zero."""
    from consistent__style_transfer_amd._lib import call
    R, V = 37, 10000
    Vp = (V + 63) // 64 * 64
    logits = dev(rnd(R, V, seed=1, scale=2.0))
    tgt = torch.randint(0, V, (R,), device="cuda")
    row, dl = torch.empty(R, device="cuda"), torch.empty(R, V, device="cuda")
    dlb = torch.full((R, Vp), 0x7fc0, device="cuda", dtype=torch.int16)          # NaN pattern: every element must be overwritten
    call("cst_token_ce_b", logits, V, tgt, R, V, row, dl, V, 1.0 / R, dlb, Vp)
    ref_row, ref_dl = torch.empty(R, device="cuda"), torch.empty(R, V, device="cuda")
    call("cst_token_ce", logits, V, tgt, R, V, ref_row, ref_dl, V, 1.0 / R)
    assert torch.equal(dl, ref_dl) and torch.equal(row, ref_row)
    twin = dlb.view(torch.bfloat16)
    assert torch.equal(twin[:, :V], dl.to(torch.bfloat16))
    assert (dlb[:, V:] == 0).all()


def test_generator_teacher_forced_grads_with_and_without_bf16_twin(ops):
    """The vocabulary dgrad / fn_2 weight gradient through the bf16 twin of dlogits agree with the fp32-operand path."""
    from consistent__style_transfer_amd import model, synthetic as syn
    from helpers import CONFIGS
    from test_gpu_modules import set_constants
    set_constants(model, CONFIGS["ref"])
    ops.set_precision("bf16")
    B, L, V = 64, 8, 1000
    torch.manual_seed(0)
    g = model.DenoiseLSTM(V, 2, L).cuda().eval()
    x, lab = (t.cuda() for t in syn.optimize_batch(B, L, V, 3))
    coins = torch.ones(L, dtype=torch.int32, device="cuda")
    grads = []
    for use_twin in (True, False):
        g.zero_grad()
        lg = g(x, lab, x, lab, coins=coins)
        loss = ops.token_ce(lg.view(-1, V), x.reshape(-1), unit_grad=True)
        if not use_twin:
            ops._SIDE_BF16.clear()
        loss.backward()
        grads.append({k: p.grad.clone() for k, p in g.named_parameters()})
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        err = (a - b).norm() / (b.norm() + 1e-12)
        assert err < 3e-2, (k, float(err))


def test_vocab_proj_bf16_matches_linear(ops):
    """The bf16 output head (with and without the producer-written twins) against the fp32-operand linear."""
    T, d, V = 128, 64, 1000
    x0, W0, b0, w = dev(rnd(T, d, seed=1)), dev(rnd(V, d, seed=2, scale=0.2)), dev(rnd(V, seed=3)), dev(rnd(T, V, seed=4))
    res = []
    for fn in (ops.linear, ops.vocab_proj):
        ops.set_precision("f32" if fn is ops.linear else "bf16")
        x, W, b = (t.clone().requires_grad_(True) for t in (x0, W0, b0))
        y = fn(x, W, b)
        (y * w).sum().backward()
        res.append([y.detach(), x.grad, W.grad, b.grad])
    ops.set_precision("bf16")
    for a, r in zip(res[1], res[0]):
        assert (a - r).norm() / r.norm() < 2e-2
    # and through the token-CE twin: same gradients as without it
    tgt = torch.randint(0, V, (T,), device="cuda")
    grads = []
    for use_twin in (True, False):
        x, W, b = (t.clone().requires_grad_(True) for t in (x0, W0, b0))
        loss = ops.token_ce(ops.vocab_proj(x, W, b), tgt, unit_grad=True)
        if not use_twin:
            ops._SIDE_BF16.clear()
        loss.backward()
        grads.append([x.grad, W.grad, b.grad])
    for a, r in zip(*grads):
        assert (a - r).norm() / (r.norm() + 1e-12) < 1e-2


def test_lstm_seq_fwd_equals_per_step_path(ops):
    """The one-launch BiLSTM encoder forward against the per-step gate GEMM + cell launches it replaces."""
    from consistent__style_transfer_amd import gen_fn
    from consistent__style_transfer_amd._lib import call
    B, L, H = 32, 5, 256
    whh = [dev(rnd(4 * H, H, seed=1 + d, scale=0.08)) for d in range(2)]
    wb = [ops.cast_bf16(w, want_t=False)[0] for w in whh]
    xp = [dev(rnd(B, L * 4 * H, seed=3 + d, scale=0.5)) for d in range(2)]
    h0 = dev(rnd(B, 2 * H, seed=5, scale=0.5))

    def bufs():
        return dict(genc=torch.empty(2, L, B, 4 * H, device="cuda"), cenc=torch.zeros(2, L, B, H, device="cuda"),
                    hprev=torch.empty(2, B, L, H, device="cuda"), c_cat=torch.empty(B, 2 * H, device="cuda"),
                    mem=torch.empty(B, L, 2 * H, device="cuda"), memb=torch.zeros(B, L * 2 * H, device="cuda", dtype=torch.int16))
    r, f = bufs(), bufs()
    # reference: the per-step fused GEMM + cell (itself tested against gemm + cell kernels)
    mem2, zeros_c = r["mem"].view(B, L * 2 * H), torch.zeros(B, H, device="cuda")
    for d in range(2):
        order = list(range(L)) if d == 0 else list(range(L - 1, -1, -1))
        hp2 = r["hprev"][d].view(B, L * H)
        for n, t in enumerate(order):
            if n == 0:
                h_in = h0[:, d * H:(d + 1) * H]
                hp2[:, t * H:(t + 1) * H].copy_(h_in)
                Ab, c_in = ops.cast_bf16(h_in, want_t=False)[0], zeros_c
            else:
                tp = order[n - 1]
                Ab, c_in = r["memb"][:, tp * 2 * H + d * H: tp * 2 * H + (d + 1) * H], r["cenc"][d, tp]
            last = n == L - 1
            gen_fn._gemm_cell_fwd([dict(Ab=Ab, Bb=wb[d], gates=r["genc"][d, t], c_prev=c_in, h_out=mem2[:, t * 2 * H + d * H: t * 2 * H + (d + 1) * H],
                                        c_out=r["c_cat"][:, d * H:(d + 1) * H] if last else r["cenc"][d, t],
                                        h_out2=None if last else hp2[:, order[n + 1] * H:(order[n + 1] + 1) * H],
                                        addend=xp[d][:, t * 4 * H:(t + 1) * 4 * H], hb=r["memb"][:, t * 2 * H + d * H: t * 2 * H + (d + 1) * H])], B, H)
    hpb = torch.empty(2, B, L, H, device="cuda", dtype=torch.int16)
    call("cst_lstm_seq_fwd", gen_fn._lstm_frag_order(wb[0], H), gen_fn._lstm_frag_order(wb[1], H), xp[0], xp[1], h0, 2 * H, f["genc"][0], f["genc"][1], f["cenc"][0], f["cenc"][1],
         f["hprev"][0], f["hprev"][1], hpb[0], hpb[1], f["c_cat"], 2 * H, f["mem"], f["memb"], B, L, H)
    torch.cuda.synchronize()
    # round 3: the split form (two workgroups per row group exchanging h_t every step) must give the one-workgroup kernel's results
    # bit for bit (same products, same summation order), with its timeout word untouched
    from consistent__style_transfer_amd._lib import call_plain
    s2 = bufs()
    hpb2 = torch.empty(2, B, L, H, device="cuda", dtype=torch.int16)
    nb = call_plain("cst_lstm_seq_xchg_bytes", B)
    xchg = torch.full((nb,), 0x5A, device="cuda", dtype=torch.uint8)                  # dirty workspace: the entry point zeroes it ...
    xchg[-16:] = 0                                                                  # ... but for the sticky timeout word at its end
    for _ in range(3):                                                              # repeated launches reuse the workspace
        call("cst_lstm_seq_fwd_split", gen_fn._lstm_frag_order(wb[0], H), gen_fn._lstm_frag_order(wb[1], H), xp[0], xp[1], h0, 2 * H, s2["genc"][0], s2["genc"][1],
             s2["cenc"][0], s2["cenc"][1], s2["hprev"][0], s2["hprev"][1], hpb2[0], hpb2[1], s2["c_cat"], 2 * H, s2["mem"], s2["memb"], B, L, H, xchg, nb)
    torch.cuda.synchronize()
    assert int(xchg[-16:].view(torch.int32)[0].item()) == 0, "a workgroup gave up waiting for its partner"
    for k in ("genc", "hprev", "c_cat", "mem", "memb"):
        assert torch.equal(s2[k], f[k]), k
    assert torch.equal(hpb2, hpb)
    assert torch.equal(hpb.view(torch.bfloat16), f["hprev"].to(torch.bfloat16))      # the optional bf16 twin of hprev
    for k in ("genc", "hprev", "c_cat", "mem"):
        close(f[k], r[k], 2e-3, 2e-3, k)                       # bf16 h feedback: rounding-level differences compound over the steps
    for d in range(2):                                        # cell states of every step but the last (that one lives in c_cat)
        order = list(range(L)) if d == 0 else list(range(L - 1, -1, -1))
        for t in order[:-1]:
            close(f["cenc"][d, t], r["cenc"][d, t], 2e-3, 2e-3)
    close(f["memb"].view(torch.bfloat16).float(), r["memb"].view(torch.bfloat16).float(), 2e-2, 2e-2)


def test_lstm_seq_split_timeout_poisons_reports_and_falls_back(ops):
    """what happens when the split encoder kernel's workgroup pairs are NOT resident together.  The test-only entry
    point launches the first half of every pair alone: every workgroup runs into its bounded spin at the first exchange (~0.2 s), and must
    (a) set the sticky timeout word, (b) leave NaN -- not numbers computed from stale hidden states -- in every later encoder state and in
    c_last, (c) make gen_fn.check_exchange_timeouts() raise and switch the process to the one-workgroup kernel, (d) leave a workspace
    that the next (healthy) launch can use after the word is cleared."""
    from consistent__style_transfer_amd import gen_fn
    from consistent__style_transfer_amd._lib import call, call_plain
    B, L, H = 16, 2, 256                                     # one row group, ONE exchange (the last step has none)
    whh = [dev(rnd(4 * H, H, seed=1 + d, scale=0.08)) for d in range(2)]
    wb = [ops.cast_bf16(w, want_t=False)[0] for w in whh]
    xp = [dev(rnd(B, L * 4 * H, seed=3 + d, scale=0.5)) for d in range(2)]
    h0 = dev(rnd(B, 2 * H, seed=5, scale=0.5))
    assert call_plain("cst_lstm_seq_split_workgroups", B) == 4
    assert call_plain("cst_lstm_seq_split_capacity") == torch.cuda.get_device_properties(0).multi_processor_count

    def run(entry, xchg):
        b = dict(genc=torch.zeros(2, L, B, 4 * H, device="cuda"), cenc=torch.zeros(2, L, B, H, device="cuda"),
                 hprev=torch.zeros(2, B, L, H, device="cuda"), c_cat=torch.zeros(B, 2 * H, device="cuda"),
                 mem=torch.zeros(B, L, 2 * H, device="cuda"), memb=torch.zeros(B, L * 2 * H, device="cuda", dtype=torch.int16))
        call(entry, gen_fn._lstm_frag_order(wb[0], H), gen_fn._lstm_frag_order(wb[1], H), xp[0], xp[1], h0, 2 * H, b["genc"][0], b["genc"][1],
             b["cenc"][0], b["cenc"][1], b["hprev"][0], b["hprev"][1], None, None, b["c_cat"], 2 * H, b["mem"], b["memb"], B, L, H, xchg, xchg.numel())
        torch.cuda.synchronize()
        return b

    saved = dict(gen_fn._XCHG), dict(gen_fn._SPLIT)
    try:
        gen_fn._XCHG.clear()
        xchg = gen_fn._xchg_workspace(torch.device("cuda", 0), B)         # registered: exchange_timed_out() reads its last word
        bad = run("cst_lstm_seq_fwd_split_lone_half", xchg)
        assert int(xchg[-16:].view(torch.int32)[0].item()) == 1
        # half p = 0 owns hidden units 0..127 of each direction; its second (= last) step ran on a NaN partner half
        for d, t_last in ((0, L - 1), (1, 0)):
            assert torch.isnan(bad["mem"][:, t_last, d * H:d * H + 128]).all(), "stale numbers instead of NaN in the encoder states"
            assert torch.isnan(bad["c_cat"][:, d * H:d * H + 128]).all(), "stale numbers instead of NaN in c_last"
        assert gen_fn.split_enabled(B)
        with pytest.raises(RuntimeError, match="timed out"):
            gen_fn.check_exchange_timeouts()
        assert not gen_fn.split_enabled(B), "a timeout must switch the process to cst_lstm_seq_fwd"
        # probe form: clears the word and reports instead of raising
        gen_fn._SPLIT.update(disabled=False, why=None)
        assert gen_fn.probe_split() is False and not gen_fn.split_enabled(B)
        assert int(xchg[-16:].view(torch.int32)[0].item()) == 0
        good = run("cst_lstm_seq_fwd_split", xchg)
        assert int(xchg[-16:].view(torch.int32)[0].item()) == 0
        assert torch.isfinite(good["mem"]).all() and torch.isfinite(good["c_cat"]).all()
    finally:
        gen_fn._XCHG.clear(); gen_fn._XCHG.update(saved[0])
        gen_fn._SPLIT.clear(); gen_fn._SPLIT.update(saved[1])


def test_lstm_seq_bwd_equals_per_step_path(ops):
    """The one-launch BiLSTM encoder backward against the per-step cell backward + dh GEMM launches it replaces."""
    from consistent__style_transfer_amd import gen_fn
    from consistent__style_transfer_amd._lib import call
    B, L, H = 32, 5, 256
    whh = [dev(rnd(4 * H, H, seed=1 + d, scale=0.08)) for d in range(2)]
    wt = [ops.cast_bf16(w)[1] for w in whh]                                   # W_hh^T [H, 4H] bf16
    genc = torch.sigmoid(dev(rnd(2, L, B, 4 * H, seed=3)))
    genc[:, :, :, 2 * H:3 * H] = torch.tanh(dev(rnd(2, L, B, H, seed=4)))    # the g gate lives in (-1, 1)
    cenc, c_cat = dev(rnd(2, L, B, H, seed=5)), dev(rnd(B, 2 * H, seed=6))
    dc_cat, dmem = dev(rnd(B, 2 * H, seed=7)), dev(rnd(B, L * 2 * H, seed=8))
    zeros_c = torch.zeros(B, H, device="cuda")
    # reference: the per-step path of gen_fn (cell backward, then fused dh GEMM + cell backward of the step before)
    r_dge, r_dh0 = torch.empty(2, B, L, 4 * H, device="cuda"), torch.empty(B, 2 * H, device="cuda")
    for d in range(2):
        order = list(range(L)) if d == 0 else list(range(L - 1, -1, -1))
        dg2d, dce = r_dge[d].view(B, L * 4 * H), torch.empty(B, H, device="cuda")
        dgb = torch.zeros(B, 4 * H, device="cuda", dtype=torch.int16)
        for n_ in range(L - 1, -1, -1):
            t = order[n_]
            lastf = n_ == L - 1
            c_new = c_cat[:, d * H:(d + 1) * H] if lastf else cenc[d, t]
            c_prev = zeros_c if n_ == 0 else cenc[d, order[n_ - 1]]
            dgt, dmt = dg2d[:, t * 4 * H:(t + 1) * 4 * H], dmem[:, t * 2 * H + d * H: t * 2 * H + (d + 1) * H]
            if lastf:
                gen_fn._cell_bwd(genc[d, t], c_prev, c_new, dmt, None, dc_cat[:, d * H:(d + 1) * H], dgt, dce, B, H, dgb=dgb)
            else:
                gen_fn._gemm_cell_bwd([dict(Ab=dgb, Bb=wt[d], gates=genc[d, t], c_prev=c_prev, c_new=c_new, dh_extra=dmt, dc_in=dce,
                                            dgates=dgt, dc_prev=dce, dgb=dgb)], B, H)
        ops.gemm_bf16(dgb, wt[d], B, H, C=r_dh0[:, d * H:(d + 1) * H])
    f_dge, f_dh0 = torch.empty(2, B, L, 4 * H, device="cuda"), torch.empty(B, 2 * H, device="cuda")
    f_dgb = torch.empty(2, B, L, 4 * H, device="cuda", dtype=torch.int16)
    call("cst_lstm_seq_bwd", gen_fn._lstm_frag_order_t(wt[0], H), gen_fn._lstm_frag_order_t(wt[1], H), genc[0], genc[1], cenc[0], cenc[1],
         c_cat, 2 * H, dc_cat, 2 * H, dmem, f_dge[0], f_dge[1], f_dgb[0], f_dgb[1], f_dh0, 2 * H, B, L, H)
    torch.cuda.synchronize()
    assert torch.equal(f_dgb.view(torch.bfloat16), f_dge.to(torch.bfloat16))         # the optional bf16 twin of dgates
    close(f_dge, r_dge, 5e-3, 5e-3)                  # bf16 dgates feedback: rounding-level differences compound over the steps
    close(f_dh0, r_dh0, 5e-3, 5e-3)


# ------------------------------------------------------------------------------------------ decoder step kernels (csrc/decode.hip)
def _bf(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("B,Hd,mode", [(16, 64, "teacher"), (40, 512, "feedback"), (256, 512, "teacher"), (256, 512, "first")])
def test_dec_gates(ops, B, Hd, mode):
    """rnn.py:74, :88-96: token choice (arg-max word of the previous step / teacher token by the coin), embedding + dropout, gate
    product over [x_t | h_{t-1}], LSTM cell -- against the same arithmetic in fp32 torch on bf16-rounded operands."""
    from consistent__style_transfer_amd._lib import call, call_plain
    E, V = 128, 300
    K = E + Hd
    W, bias = rnd(4 * Hd, K, seed=1, scale=0.05), rnd(4 * Hd, seed=2, scale=0.1)
    table = rnd(V, E, seed=3)
    hprev, cprev = rnd(B, Hd, seed=4, scale=0.5), rnd(B, Hd, seed=5, scale=0.5)
    x0 = rnd(B, E, seed=6)
    g = torch.Generator().manual_seed(7)
    ids_arg = torch.randint(0, V, (B,), generator=g)
    ids_t = torch.randint(0, V, (B, 5), generator=g)
    ids_arg[0] = V + 3                                               # out-of-range id -> zero embedding (as cst_embed_gather does)
    for coin in ((0, 1) if mode == "teacher" else (1,)):
        A = torch.zeros(B, K)
        A[:, :E] = x0
        A[:, E:] = hprev
        Ab, _ = ops.cast_bf16(dev(A), want_t=False)
        Wb, _ = ops.cast_bf16(dev(W), want_t=False)
        # the row's arg-max word sits in ONE of the NG group slots (a different one per row), the others hold smaller / empty words
        NG = call_plain("cst_argmax_groups")
        packed = torch.zeros(NG, B, dtype=torch.int64)
        packed[torch.arange(B) % NG, torch.arange(B)] = (torch.full((B,), 1 << 40, dtype=torch.int64)) | (0xFFFFFFFF - ids_arg)
        packed[(torch.arange(B) + 1) % NG, torch.arange(B)] = (torch.full((B,), 1 << 39, dtype=torch.int64)) | (0xFFFFFFFF - 1)
        packed = packed.cuda()
        d = ops.Drop(0.1, 5, 203)
        gates, c_out, h_out = (torch.full((B, n), float("nan"), device="cuda") for n in (4 * Hd, Hd, Hd))
        hb = torch.zeros(B, Hd, device="cuda", dtype=torch.int16)
        xb = torch.zeros(B, E, device="cuda", dtype=torch.int16)
        first = mode == "first"
        call("cst_dec_gates", Ab, Ab.stride(0), Wb, Wb.stride(0), None if first else packed,
             dev(ids_t)[:, 2] if mode == "teacher" else None, 5 if mode == "teacher" else 0,
             torch.tensor([coin], dtype=torch.int32, device="cuda") if mode == "teacher" else None,
             dev(table), E, V, *d.args(), None if first else xb, E, dev(bias), dev(cprev), Hd,
             gates, 4 * Hd, c_out, Hd, h_out, Hd, hb, Hd, B, E, Hd)
        if first:
            x = _bf(x0)
        else:
            tok = ids_arg.clone() if (mode == "feedback" or coin) else ids_t[:, 2].clone()
            okm = (tok >= 0) & (tok < V)
            emb = table[tok.clamp(0, V - 1)] * okm[:, None].float()
            mask = torch.from_numpy(orng.dropout_mask(5, 203, (B, E), 0.1))
            x = _bf(emb * mask)
            assert torch.equal(xb.view(torch.bfloat16).float().cpu(), x)
        pre = torch.cat([x, _bf(hprev)], 1) @ _bf(W).t() + bias
        i, f, gg, o = (pre[:, q * Hd:(q + 1) * Hd] for q in range(4))
        i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
        c = f * cprev + i * gg
        h = o * torch.tanh(c)
        close(gates, torch.cat([i, f, gg, o], 1), 2e-3, 2e-3, f"gates {mode} coin={coin}")
        close(c_out, c, 2e-3, 2e-3)
        close(h_out, h, 2e-3, 2e-3)
        assert torch.equal(hb.view(torch.bfloat16).float().cpu(), _bf(h_out.cpu()))


@pytest.mark.parametrize("shape", [(256, 512, 1024), (16, 64, 128), (40, 1024, 512), (256, 512, 640), (100, 96, 1280)])
@pytest.mark.parametrize("act", [0, 2])
def test_gemm_bf16_skinny(ops, shape, act):
    from consistent__style_transfer_amd._lib import call
    M, N, K = shape
    A, Bm, bias = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    Ab, _ = ops.cast_bf16(dev(A), want_t=False)
    Bb, _ = ops.cast_bf16(dev(Bm), want_t=False)
    C = torch.full((M, N), float("nan"), device="cuda")
    Cb = torch.zeros(M, N, device="cuda", dtype=torch.int16)
    d = ops.Drop(0.2, 4, 77) if act == 2 else ops.NO_DROP
    call("cst_gemm_bf16_skinny", Ab, Ab.stride(0), Bb, Bb.stride(0), C, N, Cb, N, M, N, K, dev(bias), act, *d.args())
    ref = _bf(A) @ _bf(Bm).t() + bias
    if act == 2:
        ref = torch.where(ref > 0, ref, 0.1 * ref) * torch.from_numpy(orng.dropout_mask(4, 77, (M, N), 0.2))
    close(C, ref, 2e-3, 2e-3 * math.sqrt(K), f"{shape}")
    assert torch.equal(Cb.view(torch.bfloat16).float().cpu(), _bf(C.cpu()))


@pytest.mark.parametrize("shape", [(256, 10000, 512), (16, 208, 512), (70, 1000, 64), (256, 384, 128)])
def test_gemm_bf16_argmax(ops, shape):
    """fn_2 of a decode step with the row arg-max folded into packed words by atomic max: the product equals cst_gemm_bf16, the ids
    equal torch.argmax of THAT product -- the first index among equal maxima (duplicated weight rows force exact ties)."""
    from consistent__style_transfer_amd._lib import call, call_plain
    M, N, K = shape
    A, Bm = rnd(M, K, seed=1), rnd(N, K, seed=2)
    Bm[N // 2 + 5] = Bm[7]                                  # columns 7 and N/2 + 5 tie exactly in every row
    Bm[N - 1] = Bm[3]
    A[:, :] = A.abs()                                       # and make those columns likely maxima of some rows
    Bm[7] = Bm[7].abs() * 3
    Bm[N // 2 + 5] = Bm[7]
    Ab, _ = ops.cast_bf16(dev(A), want_t=False)
    Bb, _ = ops.cast_bf16(dev(Bm), want_t=False)
    C = torch.full((M, N), float("nan"), device="cuda")
    NG = call_plain("cst_argmax_groups")
    packed = torch.zeros(NG, M, device="cuda", dtype=torch.int64)
    call("cst_gemm_bf16_argmax", Ab, Ab.stride(0), Bb, Bb.stride(0), C, N, M, N, K, packed)
    C2 = torch.empty(M, N, device="cuda")
    ops.gemm_bf16(Ab, Bb, M, N, C=C2, tile=64, splitk=1)
    assert torch.equal(C, C2)
    ids = torch.empty(M, device="cuda", dtype=torch.int64)
    call("cst_unpack_argmax", packed, ids, M, 1)
    ref = C.cpu().argmax(-1)                                 # torch returns the first maximal index
    assert torch.equal(ids.cpu(), ref)
    assert int((ref == 7).sum()) > 0                         # the tie really decides rows
    # a second call on the same words with the same data changes nothing (max is idempotent) ...
    before = packed.clone()
    call("cst_gemm_bf16_argmax", Ab, Ab.stride(0), Bb, Bb.stride(0), C, N, M, N, K, packed)
    assert torch.equal(packed, before)


@pytest.mark.parametrize("B,L", [(16, 8), (256, 18), (40, 39), (8, 64)])
def test_dec_attn(ops, B, L):
    """rnn.py:46-50, :76, :79 for the decode loop's shapes: attention output, weights and the dropped bf16 FFN input."""
    from consistent__style_transfer_amd._lib import call
    D = 512
    q, mem = rnd(B, D, seed=1), rnd(B, L, D, seed=2)
    out = torch.full((B, D), float("nan"), device="cuda")
    p = torch.full((B, L), float("nan"), device="cuda")
    db = torch.zeros(B, 2 * D, device="cuda", dtype=torch.int16)
    d = ops.Drop(0.1, 9, 104)
    call("cst_dec_attn", dev(q), D, dev(mem), out, D, p, B, L, D, db, 2 * D, *d.args())
    a = torch.softmax(torch.einsum("bd,bld->bl", q, mem) / math.sqrt(D), -1)
    ref = torch.einsum("bl,bld->bd", a, mem)
    close(p, a, 2e-4, 1e-6)
    close(out, ref, 2e-4, 1e-5)
    mask = torch.from_numpy(orng.dropout_mask(9, 104, (B, 2 * D), 0.1))
    want = _bf(torch.cat([q, out.cpu()], 1) * mask)
    assert torch.equal(db.view(torch.bfloat16).float().cpu(), want)


@pytest.mark.parametrize("shape", [(256, 10000), (16, 208), (70, 1000), (256, 33), (300, 4097), (64, 32)])
def test_dec_fn2(ops, shape):
    """cst_dec_fn2 (A-stationary fn_2, K = 512): product against the bf16-rounded reference, arg-max ids == torch.argmax of the
    kernel's own product with duplicated weight rows forcing exact ties, repeated launches bit-identical (no split-K, no float atomics)."""
    from consistent__style_transfer_amd._lib import call, call_plain
    M, N = shape
    K = 512
    A, Bm = rnd(M, K, seed=1).abs(), rnd(N, K, seed=2)
    if N > 40:
        Bm[7] = Bm[7].abs() * 3
        Bm[N // 2 + 5] = Bm[7]                              # columns 7 and N/2 + 5 tie exactly in every row
        Bm[N - 1] = Bm[3]
    Ab, _ = ops.cast_bf16(dev(A), want_t=False)
    Bb, _ = ops.cast_bf16(dev(Bm), want_t=False)
    NG = call_plain("cst_argmax_groups")
    ldc = N + 8
    Cbuf = torch.full((M, ldc), float("nan"), device="cuda")
    packed = torch.zeros(NG, M, device="cuda", dtype=torch.int64)
    call("cst_dec_fn2", Ab, Ab.stride(0), Bb, Bb.stride(0), Cbuf, ldc, M, N, K, packed)
    C = Cbuf[:, :N]
    assert torch.isnan(Cbuf[:, N:]).all()                    # nothing written past the row
    ref = _bf(A) @ _bf(Bm).t()
    close(C, ref, 2e-3, 2e-3 * math.sqrt(K), f"{shape}")
    ids = torch.empty(M, device="cuda", dtype=torch.int64)
    call("cst_unpack_argmax", packed, ids, M, 1)
    want = C.cpu().argmax(-1)
    assert torch.equal(ids.cpu(), want)
    if N > 40:
        assert int((want == 7).sum()) > 0
    C2 = torch.full((M, ldc), float("nan"), device="cuda")
    p2 = torch.zeros(NG, M, device="cuda", dtype=torch.int64)
    call("cst_dec_fn2", Ab, Ab.stride(0), Bb, Bb.stride(0), C2, ldc, M, N, K, p2)
    assert torch.equal(C2[:, :N], C) and torch.equal(p2.max(0).values, packed.max(0).values)


@pytest.mark.parametrize("B,L,T", [(256, 18, 3), (16, 40, 2), (5, 64, 2), (33, 7, 4)])
def test_dec_attn_cell_bwd(ops, B, L, T):
    """cst_dec_attn_cell_bwd + cst_dec_attn_dmem against a torch autograd restatement of rnn.py:46-50 (single-query attention) and the
    LSTM cell (rnn.py:75), T chained steps sharing dc and the memory gradient."""
    from consistent__style_transfer_amd._lib import call
    D, W_ = 512, 1024
    mem = rnd(B, L, D, seed=1)
    hs, g = rnd(B, T, D, seed=2), rnd(B, T, W_, seed=3)                       # h_s; d[h_s | a_s]
    gates_pre, c_prev = rnd(T, B, 4 * D, seed=4), rnd(T, B, D, seed=5)
    dh2, dc0 = rnd(T, B, D, seed=6), rnd(B, D, seed=7)
    # forward restatement with autograd
    memr = mem.clone().requires_grad_(True)
    dmem_ref = torch.zeros_like(mem)
    want_dg, want_ds, dc = [], [], dc0.clone()
    act, cnew, patt = [], [], []
    for s in range(T):
        gp = gates_pre[s]
        i, f, gg, o = torch.sigmoid(gp[:, :D]), torch.sigmoid(gp[:, D:2 * D]), torch.tanh(gp[:, 2 * D:3 * D]), torch.sigmoid(gp[:, 3 * D:])
        act.append(torch.cat([i, f, gg, o], 1))
        cnew.append(f * c_prev[s] + i * gg)
        patt.append(torch.softmax(torch.einsum("bd,bld->bl", hs[:, s], mem) / math.sqrt(D), -1))
    for s in range(T - 1, -1, -1):
        h = hs[:, s].clone().requires_grad_(True)
        a = torch.einsum("bl,bld->bd", torch.softmax(torch.einsum("bd,bld->bl", h, memr) / math.sqrt(D), -1), memr)
        gq, gm = torch.autograd.grad((a * g[:, s, D:]).sum(), [h, memr])
        dmem_ref += gm
        dh = g[:, s, :D] + gq + dh2[s]
        gp = gates_pre[s].clone().requires_grad_(True)
        cp = c_prev[s].clone().requires_grad_(True)
        i, f, gg, o = torch.sigmoid(gp[:, :D]), torch.sigmoid(gp[:, D:2 * D]), torch.tanh(gp[:, 2 * D:3 * D]), torch.sigmoid(gp[:, 3 * D:])
        c = f * cp + i * gg
        hh = o * torch.tanh(c)
        dgp, dcp = torch.autograd.grad((hh * dh).sum() + (c * dc).sum(), [gp, cp])
        want_dg.append(dgp)
        dc = dcp
    want_dg = want_dg[::-1]
    # device
    gd, memd, hsd = dev(g.reshape(B, T * W_)), dev(mem), dev(hs.reshape(B, T * D))
    pd, actd, cpd, cnd, dh2d = dev(torch.stack(patt)), dev(torch.stack(act)), dev(c_prev), dev(torch.stack(cnew)), dev(dh2)
    dcd = dev(dc0)
    ds_all = torch.full((T, B, L), float("nan"), device="cuda")
    dgd = torch.full((T, B, 4 * D), float("nan"), device="cuda")
    dgb = torch.zeros(T, B, 4 * D, device="cuda", dtype=torch.int16)
    for s in range(T - 1, -1, -1):
        call("cst_dec_attn_cell_bwd", gd[:, s * W_:], T * W_, memd, pd[s], ds_all[s], B, L, D, actd[s], 4 * D, cpd[s], D, cnd[s], D,
             dh2d[s], D, dcd, D, dgd[s], 4 * D, dcd, D, dgb[s], 4 * D)
    dmem = torch.zeros(B, L, D, device="cuda")
    call("cst_dec_attn_dmem", gd[:, D:], T * W_, W_, hsd, T * D, D, pd, ds_all, dmem, B, T, L, D)
    for s in range(T):
        close(dgd[s], want_dg[s], 2e-4, 2e-5, f"dgates step {s}")
    assert torch.equal(dgb.view(torch.bfloat16).float().cpu(), _bf(dgd.cpu()))
    close(dcd, dc, 2e-4, 2e-5, "dc")
    close(dmem, dmem_ref, 2e-4, 2e-5, "dmem")


@pytest.mark.parametrize("M,V,p", [(256, 10000, 0.1), (16, 208, 0.1), (37, 1000, 0.0), (5, 36, 0.25)])
def test_dec_dxe(ops, M, V, p):
    """cst_dec_dxe: C += dropout(g) E^T on bf16-rounded operands (rnn.py:84-85's straight-through gradient), the dropped rows written once,
    nothing written outside the (M, V) block of a wider buffer."""
    from consistent__style_transfer_amd._lib import call
    K, ldg, ldc = 128, 640, 3 * V
    g, E, C0 = rnd(M, ldg, seed=1), rnd(V, K, seed=2), rnd(M, ldc, seed=3)
    Eb, _ = ops.cast_bf16(dev(E), want_t=False)
    gd, C = dev(g), dev(C0)
    gx = torch.full((M, K), float("nan"), device="cuda")
    d = ops.Drop(p, 5, 321) if p > 0 else ops.NO_DROP
    call("cst_dec_dxe", gd, ldg, gx if p > 0 else None, K, Eb, Eb.stride(0), C[:, V:], ldc, M, V, K, *d.args())
    mask = torch.from_numpy(orng.dropout_mask(5, 321, (M, K), p)) if p > 0 else torch.ones(M, K)
    want_gx = g[:, :K] * mask
    if p > 0:
        assert torch.equal(gx.cpu(), want_gx)
    ref = C0[:, V:2 * V] + _bf(want_gx) @ _bf(E).t()
    close(C[:, V:2 * V], ref, 2e-3, 2e-3 * math.sqrt(K), f"{M}x{V}")
    assert torch.equal(C[:, :V].cpu(), C0[:, :V]) and torch.equal(C[:, 2 * V:].cpu(), C0[:, 2 * V:])


def test_gather_grads_leaves_the_sum_of_squares_for_the_clip(ops):
    """optim.FlatGroup.gather_grads: the multi-tensor accumulate also leaves per-chunk sums of squares of the flat gradient buffer as it is
    afterwards (overwrite and += modes, parameters without a gradient included), so clip_groups needs no pass of its own; anything that
    rewrites flat_g in between (in-place clip, zero_grad) falls back to the full pass."""
    from consistent__style_transfer_amd import optim
    torch.manual_seed(0)
    shapes = [(300, 17), (4096,), (5000, 3), (7,), (64, 64)]
    params = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
    grp = optim.FlatGroup(params, 1e-3)

    def norm2():
        out = torch.zeros(1, device="cuda")
        grp.sumsq_into(out)
        return out.item()

    g1 = [torch.randn(*s, device="cuda") for s in shapes]
    for p, g in zip(params, g1):
        p.grad = g.clone()
    params[3].grad = None                                       # a parameter without a gradient: its slot is zero after an overwrite gather
    grp.gather_grads(False)
    assert grp._ssq_ready
    want = sum((g.double() ** 2).sum().item() for i, g in enumerate(g1) if i != 3)
    got = norm2()
    assert abs(got - want) <= 1e-5 * want, (got, want)
    assert abs((grp.flat_g.double() ** 2).sum().item() - want) <= 1e-9 * want
    g2 = [torch.randn(*s, device="cuda") for s in shapes]
    for p, g in zip(params, g2):
        p.grad = g.clone()
    params[1].grad = None                                       # += mode: this slot keeps what it accumulated, and it counts
    grp.gather_grads(True)
    want2 = (grp.flat_g.double() ** 2).sum().item()
    got2 = norm2()
    assert grp._ssq_ready and abs(got2 - want2) <= 1e-5 * want2, (got2, want2)
    a, b = norm2(), norm2()
    assert a == b                                               # fixed summation order
    ss = torch.zeros(1, device="cuda")
    grp.sumsq_into(ss)
    grp.clip(ss, 1.0)                                           # in place: the partials no longer describe flat_g
    assert not grp._ssq_ready
    want3 = (grp.flat_g.double() ** 2).sum().item()
    assert abs(norm2() - want3) <= 1e-5 * want3 and abs(want3 - 1.0) < 1e-3
    grp.zero_grad()
    assert not grp._ssq_ready and norm2() == 0.0
