"""Counter-based dropout RNG contract shared by the HIP kernels and the oracle.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference draws dropout masks from
torch's global generator (nn.Dropout at src/model/rnn.py:40, classifier.py:20,
discriminator.py:29 and inside nn.TransformerEncoderLayer); that stream cannot be
reproduced by any other implementation, so the build defines its own contract:

    keep(seed, stream, idx)  <=>  (mix32(seed, stream, idx) >> 8) >= floor(p * 2**24)

with ``idx`` the row-major linear element index of the tensor the mask applies to,
``stream`` a per-call-site id and ``seed`` a per-step seed.  Kept values are scaled by
1/(1-p) in fp32.  The device side is ``cst_mix32`` in csrc/cst_common.h; this file is the
same integer arithmetic in numpy, so masks agree bit for bit.
"""
import numpy as np


def mix32(seed, stream, idx):
    """lowbias32-style avalanche of (seed, stream, idx); all uint32, wraps mod 2**32."""
    idx = np.asarray(idx, dtype=np.uint64)
    m = np.uint64(0xFFFFFFFF)
    x = (idx ^ ((np.uint64(stream) * np.uint64(0x9E3779B1)) & m)) & m
    x = (x * np.uint64(0x85EBCA6B) + np.uint64(seed)) & m
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & m
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & m
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def keep_threshold(p):
    """floor(p * 2^24) in the kernel's arithmetic (cst_make_drop: float p, float product): 0.3f * 2^24 rounds to 5033165.0 in fp32
    where the float64 product floors to 5033164 -- one element in 16.7 M, found by a 21 M-element GEMM epilogue test."""
    return int(np.floor(np.float32(np.float32(p) * np.float32(16777216.0))))


def dropout_mask(seed, stream, shape, p):
    """float32 mask holding 0 or 1/(1-p); identical to the device kernels' mask."""
    n = int(np.prod(shape))
    if p <= 0.0:
        return np.ones(shape, dtype=np.float32)
    h = mix32(seed, stream, np.arange(n, dtype=np.uint64))
    keep = (h >> np.uint32(8)) >= np.uint32(keep_threshold(p))
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return (keep.astype(np.float32) * scale).reshape(shape)
