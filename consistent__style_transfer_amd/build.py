"""Build libcst_hip.so (the C-ABI HIP library) in-tree for gfx950 with hipcc.

The shared object is written next to the sources (consistent__style_transfer_amd/csrc/) so that
it travels to the GPU box with the repository snapshot; it is git-ignored.
"""
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libcst_hip.so")
SOURCES = ["common.hip", "gemm.hip", "gemm_bf16.hip", "gemm_pp.hip", "rowwise.hip", "attention.hip", "attention_long.hip", "pointwise.hip", "relconv.hip", "lstm_seq.hip", "decode.hip"]
ARCH = "gfx950"


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libcst_hip.so cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


HOST_LIB = os.path.join(CSRC, "libcst_host.so")
BENCH_SOURCES = ["gemm_rs.hip"]          # only with CST_BENCH_VARIANTS=1: measured experiments that the step never dispatches (tools/gemm_rs_bench.py)
HOST_SOURCES = ["host_wmd.cpp"]


def build_host_lib(force=False, verbose=True):
    """libcst_host.so: the host-side C++ of the path (exact transportation solver behind the pretrain stage's WMD labels) -- g++, no GPU."""
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found: libcst_host.so cannot be built")
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    if force or _stale(HOST_LIB, srcs):
        cmd = [cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", HOST_LIB] + srcs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return HOST_LIB


def build_lib(force=False, verbose=True):
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    objs = []
    bench = os.environ.get("CST_BENCH_VARIANTS") == "1"
    # the two flavours never share an object file (-DCST_BENCH_VARIANTS changes what gemm_bf16 / gemm_pp hold), and the library records
    # which flavour it was linked from: switching the flavour relinks even when every object is up to date
    suffix = ".bench.o" if bench else ".o"
    stamp = os.path.join(CSRC, ".lib_flavour")
    flavour = "bench" if bench else "ship"
    for s in SOURCES + (BENCH_SOURCES if bench else []):
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", suffix))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-c", src, "-o", obj]
            if bench:                                                # bench-only GEMM variants + timing ablations (gemm_bf16.hip, gemm_pp.hip)
                cmd.insert(-4, "-DCST_BENCH_VARIANTS")
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    have = open(stamp).read().strip() if os.path.exists(stamp) else None
    if force or have != flavour or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(stamp, "w") as f:
            f.write(flavour + "\n")
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    build_host_lib(force="--force" in sys.argv)
    print(LIB)
    print(HOST_LIB)
