"""CPU: the C-ABI library loads and exports every symbol include/cst_hip.h declares; argument
validation returns status codes (no compute is launched without a GPU)."""
import ctypes
import os
import re

import pytest

from consistent__style_transfer_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    protos = _lib.parse_header()
    assert len(protos) >= 35
    L = _lib.lib()
    for name in protos:
        assert name in L.fn, name
    assert L.fn["cst_abi_version"]() == 1


def test_header_cites_reference_call_sites():
    text = open(_lib.HEADER_PATH).read()
    for ref in ("rnn.py", "mlm.py", "match.py", "classifier.py", "discriminator.py", "main_optimize.py"):
        assert ref in text


def test_argument_errors_are_status_codes_not_exceptions():
    L = _lib.lib()
    rc = L.fn["cst_gemm"](None, 0, 1, None, 0, 1, None, 0, 4, 4, 4, None, None, 0, None, 0, 0, 1.0, 0, 1.0, 0,
                          1, 0, 0, 0, 0, 0, 0, 0.0, 0, 0, None, 0, 0, None, 0, None)
    assert rc == 1
    assert "null operand" in L.last_error()
    rc = L.fn["cst_mha_fwd"](1, 1, 1, 2, 200, 8, 64, 0.0, 0, 0, None, None)
    assert rc == 1 and "unsupported" in L.last_error()
    assert L.fn["cst_layernorm_bwd_workspace_floats"](4608, 512) == 2 * 1024 * 512


def test_workspace_queries_state_the_split_k_need():
    """Every entry point that takes a workspace has a *_workspace_floats twin: shape-only, callable without a GPU.  They
    return what the library would use given an unlimited workspace -- splits x M x N floats (x batch / problems), 0 for one pass."""
    L = _lib.lib()
    q = L.fn
    # encoder-layer products fill the chip by themselves: no split
    assert q["cst_gemm_bf16_workspace_floats"](9216, 2048, 768, 0, 0) == 0
    assert q["cst_gemm_workspace_floats"](9216, 2048, 768, 1, 0, 0, 0) == 0
    # a decoder-step product (M = batch): few tiles, split along K -- a whole number of [M, N] slabs, at most K / 256 of them
    need = q["cst_gemm_bf16_workspace_floats"](256, 512, 1024, 0, 0)
    assert need > 0 and need % (256 * 512) == 0 and need // (256 * 512) <= 1024 // 256
    assert q["cst_gemm_bf16_workspace_floats"](256, 512, 1024, 0, 1) == 0                 # split-K forbidden by the caller
    assert q["cst_gemm_bf16_workspace_floats"](256, 512, 1024, 64, 4) == 4 * 256 * 512     # forced
    # weight gradient with a long contraction: dW[2304, 768] over 9216 tokens
    need = q["cst_gemm_bf16_tt_workspace_floats"](2304, 768, 9216, 0)
    assert need > 0 and need % (2304 * 768) == 0
    # recurrent products always write slabs: both encoder directions, 4H = 1024 gate columns, forced 2-way split
    assert q["cst_gemm_bf16_lstm_workspace_floats"](256, 1024, 256, 2, 2) == 2 * 2 * 256 * 1024
    need = q["cst_gemm_workspace_floats"](256, 512, 2048, 1, 0, 0, 0)
    assert need > 0 and need % (256 * 512) == 0
    assert q["cst_relconv_bwd_weight_workspace_floats"](256, 18, 3, 128, 300) > 0
    # the package's per-device workspace (ops.WS_FLOATS) covers every product of the benchmark workloads at its preferred split
    from consistent__style_transfer_amd import ops
    for M, N, K in [(256, 2048, 640), (256, 512, 1024), (256, 10000, 512), (256, 512, 10048), (256, 640, 2048), (4608, 512, 10048),
                    (512, 2048, 640), (512, 10000, 512)]:
        assert q["cst_gemm_bf16_workspace_floats"](M, N, K, 0, 0) <= ops.WS_FLOATS, (M, N, K)


def test_fused_entry_points_validate_their_limits():
    """The fused kernels state their shape limits as status 1 + message (the host wrappers fall back before that)."""
    L = _lib.lib()
    P = 16                                                    # a fake, 16-byte aligned, never dereferenced pointer
    # transposed-read GEMM: K must be a multiple of 64, M and N multiples of 8
    assert L.fn["cst_gemm_bf16_tt"](P, 8, P, 8, P, 8, 8, 8, 100, 0, 1, None, 0, None) == 1 and "multiple of 64" in L.last_error()
    assert L.fn["cst_gemm_bf16_tt"](P, 8, P, 8, P, 8, 12, 8, 64, 0, 1, None, 0, None) == 1
    # RelGAN_D convolution: window k * E/R <= 40
    assert L.fn["cst_relconv_fwd"](P, 2, 18, 128, 8, 5, P, P, 300, P, 300, P, None) == 1 and "limits" in L.last_error()
    assert L.fn["cst_relconv_fwd"](P, 2, 3, 128, 16, 5, P, P, 300, P, 300, P, None) == 1 and "L >= k" in L.last_error()
    assert L.fn["cst_relconv_bwd_weight_workspace_floats"](256, 16, 5, 128, 300) == 256 * 41 * 300
    # all-steps decoder attention backward: 2 D <= 1024, L <= 64
    assert L.fn["cst_dot_attn_bwd_steps"](P, 0, 0, P, 0, 0, P, P, P, 4, 3, 7, 768, 0.0, 0, 0, None, None) == 1
    assert L.fn["cst_dot_attn_bwd_steps"](P, 0, 0, P, 0, 0, P, P, P, 4, 3, 70, 64, 0.0, 0, 0, None, None) == 1
    # whole-sequence encoder kernels: H == 256, B % 16 == 0
    assert L.fn["cst_lstm_seq_fwd"](P, P, P, P, P, 512, P, P, P, P, P, P, None, None, P, 512, P, P, 32, 5, 128, None) == 1 and "H == 256" in L.last_error()
    assert L.fn["cst_lstm_seq_bwd"](P, P, P, P, P, P, P, 512, P, 512, P, P, P, None, None, P, 512, 24, 5, 256, None) == 1
    # softmax with a bf16 twin: the twin pointer is mandatory
    assert L.fn["cst_softmax_tau_gather_b"](P, 10000, 1.0, P, 10000, None, 10048, 10048, None, 4, 10000, None, 0, 0, None, 0, None, 0,
                                            None, 1, None, 0.0, 0, 0, None, None) == 1 and "null bf16" in L.last_error()
    # bf16 token-CE twin needs the vector path
    assert L.fn["cst_token_ce_b"](P, 10, P, 4, 10, P, P, 10, 1.0, P, 16, None) == 1


def test_no_cpu_fallback_in_product_path():
    """The product package must not import the oracle or route compute through torch on CPU."""
    root = os.path.dirname(os.path.abspath(_lib.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_host_library_exports_every_symbol_of_its_header_and_validates_arguments():
    """include/cst_host.h <-> libcst_host.so (the host side of the pretrain labels, csrc/host_wmd.cpp): every declared symbol is
    exported, bad arguments are status codes, and the header cites the reference call sites it replaces."""
    from consistent__style_transfer_amd import wmd
    hdr = os.path.join(os.path.dirname(_lib.HEADER_PATH), "cst_host.h")
    text = re.sub(r"/\*.*?\*/", " ", open(hdr).read(), flags=re.S)
    names = re.findall(r"\b(cst_host_\w+)\s*\(", text)
    assert set(names) >= {"cst_host_abi_version", "cst_host_emd", "cst_host_wmd_labels"}
    L = wmd.host_lib()
    for n in names:
        assert hasattr(L, n), n
    assert L.cst_host_abi_version() == 1
    assert L.cst_host_emd(0, 3, None, None, None, None) == 1
    assert L.cst_host_wmd_labels(None, None, None, None, 4, 0, 4, None, 0, None, 8, 1, None) == 1
    raw = open(hdr).read()
    assert "src/wmd.py:31-45" in raw and "src/loader.py:60" in raw


def test_decode_entry_points_validate_their_limits():
    """csrc/decode.hip: shape limits are status 1 + message, the LDS need is a shape-only query."""
    L = _lib.lib()
    P = 16
    assert L.fn["cst_dec_gates_lds_bytes"](128, 512) == 10 * 96 * 128
    assert L.fn["cst_argmax_groups"]() in (16, 32, 64)
    assert L.fn["cst_dec_gates"](P, 640, P, 640, None, None, 0, None, None, 0, 0, 0.0, 0, 0, None, None, 0, P, P, 512,
                                 P, 2048, P, 512, P, 512, None, 0, 256, 64, 512, None) == 1 and "E == 128" in L.last_error()
    assert L.fn["cst_gemm_bf16_skinny"](P, 2048, P, 2048, P, 512, None, 0, 256, 512, 2048, None, 0, 0.0, 0, 0, None, None) == 1 and "whole" in L.last_error()
    assert L.fn["cst_dec_fn2"](P, 640, P, 640, P, 10000, 256, 10000, 640, None, None) == 1 and "K = 512" in L.last_error()
    assert L.fn["cst_dec_attn"](P, 1024, P, P, 1024, P, 4, 18, 256, None, 0, 0.0, 0, 0, None, None) == 1 and "D == 512" in L.last_error()
    # the soft decode's backward step (round 3)
    assert L.fn["cst_dec_dxe"](P, 640, None, 0, P, 128, P, 10000, 256, 10000, 64, 0.0, 0, 0, None, None) == 1 and "K == 128" in L.last_error()
    assert L.fn["cst_dec_dxe"](P, 640, None, 0, P, 128, P, 10002, 256, 10002, 128, 0.0, 0, 0, None, None) == 1 and "multiples of 4" in L.last_error()
    assert L.fn["cst_dec_attn_cell_bwd"](P, 1024, P, P, P, 4, 18, 256, P, 1024, P, 256, P, 256, None, 0, None, 0, P, 1024, P, 256, None, 0, None) == 1 \
        and "D == 512" in L.last_error()
    assert L.fn["cst_dec_attn_dmem"](P, 1024, 1024, P, 1024, 1024, P, P, P, 4, 21, 80, 512, None) == 1 and "L <= 64" in L.last_error()
    assert L.fn["cst_sumsq_partials"](None, 4, P, None) == 1 and "bad arguments" in L.last_error()


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start the ranks itself (child processes of
    torch.distributed.run, before the parent touches the GPU), relay their output and leave with their status.  Here (no GPU) the
    children stop at bench.py's own "needs a GPU" assertion: what is checked is the launch, the relay and the status."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert "launching 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr, r.stderr[-2000:]
    assert "--nproc-per-node=2" in r.stderr
    assert r.returncode != 0                                  # the children's failure (no GPU here) is the parent's status
    assert "bench.py needs a GPU" in r.stderr or "WORLD_SIZE=2" in r.stderr or "ChildFailedError" in r.stderr, r.stderr[-2000:]
