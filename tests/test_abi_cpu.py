"""CPU: the C-ABI library loads and exports every symbol include/cst_hip.h declares; argument
validation returns status codes (no compute is launched without a GPU)."""
import ctypes
import os
import re

import pytest

from consistent__style_transfer_amd import _lib


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
    rc = L.fn["cst_mha_fwd"](1, 1, 1, 2, 100, 8, 64, 0.0, 0, 0, None, None)
    assert rc == 1 and "unsupported" in L.last_error()
    assert L.fn["cst_layernorm_bwd_workspace_floats"](4608, 512) == 2 * 1024 * 512


def test_no_cpu_fallback_in_product_path():
    """The product package must not import the oracle or route compute through torch on CPU."""
    root = os.path.dirname(os.path.abspath(_lib.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
