"""Shared helpers for the parity tests (oracle <-> golden fixtures <-> HIP path)."""
import json
import os

import numpy as np
import torch

from oracle.configs import CONFIGS
from oracle.detinit import det_state_dict, det_tensor

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEEDS = {"G": 1, "cls": 2, "mat": 3, "dn": 4, "disc": 5}      # as tests/golden/make_golden.py::build


def load_golden(kind, name):
    return dict(np.load(os.path.join(GOLDEN, f"{kind}_{name}.npz")))


def sd_shapes(name):
    with open(os.path.join(GOLDEN, f"state_dict_shapes_{name}.json")) as f:
        return json.load(f)


def det_params(name, which, requires_grad=False, device="cpu"):
    shapes = sd_shapes(name)[which]
    sd = det_state_dict(shapes, SEEDS[which])
    out = {}
    for k, v in sd.items():
        v = v.to(device)
        out[k] = v.requires_grad_(True) if requires_grad else v
    return out


def soft_input(B, L, V, seed, device="cpu"):
    logits = det_tensor(f"soft{seed}", (B, L, V), seed, scale=2.0)
    return torch.softmax(logits, -1).detach().to(device).requires_grad_(True)


def lossw(key, t):
    return (t * det_tensor(key, t.shape, 9, scale=1.0).to(t.device)).sum()


def check_grads(G, prefix, named_grads, rtol, atol, input_grad=None):
    """Compare gradients with the fixture: whole arrays where stored, else norm + strided sample."""
    checked = 0
    for k, g in named_grads.items():
        g = g.detach().cpu().numpy()
        full, nrm, smp = f"{prefix}.grad.{k}", f"{prefix}.gradnorm.{k}", f"{prefix}.gradsample.{k}"
        if full in G:
            np.testing.assert_allclose(g, G[full], rtol=rtol, atol=atol, err_msg=full)
            checked += 1
        elif nrm in G:
            stride = (g.size + G[smp].size - 1) // G[smp].size
            # the stride used by make_golden is 97 (tiny) or 1009 (ref)
            for st in (97, 1009):
                if g.reshape(-1)[::st].size == G[smp].size:
                    stride = st
            np.testing.assert_allclose(g.reshape(-1)[::stride], G[smp], rtol=rtol, atol=atol, err_msg=smp)
            np.testing.assert_allclose(np.sqrt((g.astype(np.float64) ** 2).sum()), G[nrm][0], rtol=max(rtol, 1e-4),
                                       err_msg=nrm)
            checked += 1
    if input_grad is not None:
        check_grads(G, prefix, {"__input__": input_grad}, rtol, atol)
        checked += 1
    assert checked > 0, f"no gradient of {prefix} was checked"
    return checked


def rel_l2(a, b):
    """||a - b|| / ||b|| in float64 (b = the reference)."""
    a = np.asarray(a.detach().cpu() if hasattr(a, "detach") else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if hasattr(b, "detach") else b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def grad_rel_l2(G, prefix, named_grads, input_grad=None):
    """Relative L2 deviation of every gradient from the fixture (whole arrays where stored, else the strided sample
    scaled by the stored norm).  -> {key: deviation}"""
    out = {}
    items = dict(named_grads)
    if input_grad is not None:
        items["__input__"] = input_grad
    for k, g in items.items():
        g = g.detach().cpu().numpy()
        full, nrm, smp = f"{prefix}.grad.{k}", f"{prefix}.gradnorm.{k}", f"{prefix}.gradsample.{k}"
        if full in G:
            if np.linalg.norm(G[full]) > 1e-6:
                out[k] = rel_l2(g, G[full])
        elif nrm in G and G[nrm][0] > 1e-6:
            stride = next(st for st in (97, 1009) if g.reshape(-1)[::st].size == G[smp].size)
            # deviation of the sample, measured against the per-element RMS of the whole gradient
            rms = G[nrm][0] / np.sqrt(g.size)
            out[k] = float(np.sqrt(np.mean((g.reshape(-1)[::stride].astype(np.float64) - G[smp]) ** 2)) / rms)
    return out


_REPORT = os.path.join(os.path.dirname(GOLDEN.rstrip("/")), "..", "gpurun_out", "parity_report.jsonl")


def report(test, **metrics):
    """Append measured deviations to gpurun_out/parity_report.jsonl (scratch; the numbers quoted in DESIGN.md come from it)."""
    try:
        os.makedirs(os.path.dirname(_REPORT), exist_ok=True)
        with open(_REPORT, "a") as f:
            f.write(json.dumps({"test": test, **metrics}) + "\n")
    except OSError:
        pass


__all__ = ["CONFIGS", "load_golden", "sd_shapes", "det_params", "soft_input", "lossw", "check_grads", "rel_l2", "grad_rel_l2", "report"]
