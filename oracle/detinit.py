"""Deterministic test weights and inputs owned by the build (TEST INFRASTRUCTURE).

Weights are a pure function of (state_dict key, shape, seed), so golden fixtures only need to
store *outputs*: the same weights are rebuilt on the GPU box and loaded into the oracle, the
HIP modules, and (in the build container) the imported reference modules.
"""
import zlib

import numpy as np
import torch


def det_tensor(key, shape, seed=0, scale=None):
    rs = np.random.RandomState((zlib.crc32(key.encode()) + 7919 * seed) % (2 ** 31 - 1))
    a = rs.standard_normal(size=tuple(shape)).astype(np.float32)
    if scale is None:
        if key.endswith("norm1.weight") or key.endswith("norm2.weight"):
            return torch.from_numpy(1.0 + 0.1 * a)
        if key.endswith("bias"):
            scale = 0.05
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            scale = 0.7 / np.sqrt(max(fan_in, 1))
        else:
            scale = 0.1
        if "embedding" in key and "embeddings" not in key:
            scale = 0.5
    return torch.from_numpy(a * np.float32(scale))


def det_state_dict(shapes, seed=0):
    """shapes: mapping key -> shape (e.g. {k: v.shape for k, v in module.state_dict().items()})."""
    return {k: det_tensor(k, tuple(s), seed) for k, s in shapes.items()}


def det_tokens(B, L, V, seed=0, full=False):
    """Right-padded id batch shaped like loader.align output: ids in [4,V), PAD=0."""
    rs = np.random.RandomState(1000 + seed)
    x = rs.randint(4, V, size=(B, L)).astype(np.int64)
    if not full:
        lens = rs.randint(max(1, L // 2), L + 1, size=(B,))
        lens[0] = L                                   # batch max length reaches L
        for b in range(B):
            x[b, lens[b]:] = 0
    return torch.from_numpy(x)
