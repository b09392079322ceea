"""ctypes binding of libcst_hip.so -- the only route from the Python host code to compute.

There is NO fallback: if the shared library is missing or a symbol of include/cst_hip.h is not
exported, importing this module raises.  Prototypes are parsed from the header, so the header is
the single source of truth for the ABI.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcst_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "cst_hip.h")

_CT = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
    "uint32_t": ctypes.c_uint32, "double": ctypes.c_double,
}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every prototype in include/cst_hip.h."""
    with open(path) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int|long)\s+(cst_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = ctypes.c_char_p if "char" in ret else _CT[ret.strip()]
        argtypes = []
        for a in [s.strip() for s in args.split(",")]:
            if not a or a == "void":
                continue
            if "*" in a:
                argtypes.append(ctypes.c_void_p)
            else:
                argtypes.append(_CT[a.split()[-2] if len(a.split()) > 1 else a])
        protos[name] = (restype, argtypes)
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the product path.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.fn = {}
        for name, (restype, argtypes) in self.protos.items():
            try:
                f = getattr(self.cdll, name)
            except AttributeError as e:
                raise ImportError(f"libcst_hip.so does not export {name} declared in include/cst_hip.h") from e
            f.restype = restype
            f.argtypes = argtypes
            self.fn[name] = f

    def last_error(self):
        return self.fn["cst_last_error"]().decode()


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return x.data_ptr()
    return x


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional per-entry-point timing with HIP events recorded on the launch stream (used by
    bench.py's roofline leg; never active in the timed throughput region)."""

    def __init__(self, by_shape=False):
        self.by_shape = by_shape
        self.records = []           # (name, start_event, end_event, work) ; work = flops for cst_gemm

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b, work in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "work": 0.0})
            d["calls"] += 1
            d["ms"] += a.elapsed_time(b)
            d["work"] += work
        return out


TIMER = None


def set_timer(t):
    global TIMER
    TIMER = t


def _gemm_key(args):
    # (A, lda, akm, B, ldb, bkm, C, ldc, M, N, K, ...) -> flops and a layout tag
    M, N, K = args[8], args[9], args[10]
    batch = args[21]
    return 2.0 * M * N * K * batch


def call(name, *args):
    """Invoke an int-status entry point on torch's current stream; raise on a non-zero status."""
    L = lib()
    if TIMER is not None:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = L.fn[name](*[_ptr(x) for x in args], stream_ptr())
        b.record()
        key = name
        if name == "cst_gemm" and TIMER.by_shape:
            key = f"cst_gemm[{args[8]}x{args[9]}x{args[10]} a{args[2]}b{args[5]}]"
        TIMER.records.append((key, a, b, _gemm_key(args) if name == "cst_gemm" else 0.0))
    else:
        rc = L.fn[name](*[_ptr(a) for a in args], stream_ptr())
    if rc != 0:
        raise RuntimeError(f"{name} failed (status {rc}): {L.last_error()}")


def call_plain(name, *args):
    """Entry points without a stream / status (workspace queries)."""
    return lib().fn[name](*[_ptr(a) for a in args])


def gemm_profile(enable):
    lib().fn["cst_gemm_profile_enable"](1 if enable else 0)


def gemm_profile_shapes(max_records=200000):
    """-> list of (which, M, N, K, ms) for every GEMM launch recorded since gemm_profile(True)."""
    mnk = (ctypes.c_int * (3 * max_records))()
    ms = (ctypes.c_double * max_records)()
    which = (ctypes.c_int * max_records)()
    n = ctypes.c_long()
    rc = lib().fn["cst_gemm_profile_shapes"](max_records, mnk, ms, which, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(lib().last_error())
    return [(which[i], mnk[3 * i], mnk[3 * i + 1], mnk[3 * i + 2], ms[i]) for i in range(n.value)]


def gemm_profile_read():
    """-> {kernel: (total kernel ms, total FLOP, total minimal operand bytes, launches)} since enable."""
    out = {}
    for which, name in ((2, "cst_gemm_bf16_pp_kernel"), (0, "cst_gemm_kernel"), (1, "cst_gemm_bf16_kernel")):        # (reading 1 clears the records)
        ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        n = ctypes.c_long()
        rc = lib().fn["cst_gemm_profile_read"](which, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(n))
        if rc != 0:
            raise RuntimeError(lib().last_error())
        out[name] = (ms.value, fl.value, by.value, n.value)
    return out
