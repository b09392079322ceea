#!/usr/bin/env python3
"""Which products' bf16 rounding moves the optimize loss curve ("Next round" item 3; reference loop: src/main_optimize.py:93-124).

The b16 fixture (B = 16, reference generator constants, critics of width 768 / head dim 96, 20 steps at lr 1e-4) is run with ONE module
at a time moved to the exact fp32 matrix pipe (`--precision f32` for that module only: forward AND backward), and with one module at a
time LEFT on the bf16 pipe while everything else is exact.  Per run: the largest deviation of every logged scalar from the reference
curve (tests/golden/curves_b16.npz), at step 0 and over the 20 steps, and what the same assignment costs per optimize step at the
headline workload (yelp_6l_d768_b256, hipGraph replay).

    python tools/precision_attrib.py [--no-cost]      -> table on stdout + gpurun_out/precision_attrib.json

A module's scope is entered and left inside the autograd graph as well (identity nodes around the module that switch ops.set_precision when
the backward pass reaches them): the precision decisions of this code base are taken at call time, in the forward and in the backward.
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from consistent__style_transfer_amd import ops  # noqa: E402

MODULES = {"generator": "generator", "textcnn": "classifier", "matcher": "matcher", "relgan_d": "disc"}
COLS = ["g_total", "G", "STI", "CP", "BK", "D"]


class _Mark(torch.autograd.Function):
    """Identity whose BACKWARD switches the precision: placed on a module's output (-> the module's precision) and on its inputs (-> the
    base precision), so the module's backward nodes, which sit between the two in the engine's order, run under the module's precision."""

    @staticmethod
    def forward(ctx, x, prec):
        ctx.prec = prec
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ops.set_precision(ctx.prec)
        return g, None


def scope(mod, prec, base):
    orig = mod.forward

    def fwd(*a, **k):
        a = tuple(_Mark.apply(t, base) if torch.is_tensor(t) and t.requires_grad else t for t in a)
        ops.set_precision(prec)
        try:
            y = orig(*a, **k)
        finally:
            ops.set_precision(base)
        return _Mark.apply(y, prec) if torch.is_tensor(y) and y.requires_grad else y

    mod.forward = fwd


def assign(stage, base, overrides):
    """EVERY module gets a scope (its own precision or the base one): the state is then set explicitly at every module boundary of both
    passes -- a scope whose inputs carry no gradient (the generator's token ids) is never left by the backward pass itself."""
    ops.set_precision(base)
    for name in MODULES:
        scope(getattr(stage, MODULES[name]), overrides.get(name, base), base)


def curve(base, overrides, ref_ids=None):
    from curve_inputs import curve_lr, opt_batch
    from helpers import CONFIGS, load_golden
    from test_gpu_stages import cu, make_opt
    name = "b16"
    c, G = CONFIGS[name], load_golden("curves", name)
    st = make_opt(name, lr=curve_lr(name, "optimize"))
    assign(st, base, overrides)
    rows, ids = [], []
    for it in range(G["optimize.curve"].shape[0]):
        ops.set_precision(base)                             # (a generator scope is left only when the backward pass ends)
        lg = st.train_step(cu(opt_batch(c, it)), it, coins=G["optimize.coins"][it])
        rows.append([lg["g_total"].item(), lg["G"].item(), lg["STI"].item(), lg["CP_logits"].mean().item(), lg["BK"].item(), lg["D"].item()])
        ids.append(lg["sample_ids"].cpu().numpy().copy())
    ops.set_precision("bf16")
    dev = np.abs(np.array(rows) - G["optimize.curve"])
    res = {"step0": [float(v) for v in dev[0]], "max": [float(v) for v in dev.max(0)], "worst": float(dev.max()), "ids": ids}
    if ref_ids is not None:
        # the sampled token ids of the generator step (argmax of sample_p, rnn.py:84) against the all-exact run's: the first step at which
        # any token differs, how many sentences differ at the end, and the curve's deviation over the steps BEFORE that first flip
        diff = [bool((a != b).any()) for a, b in zip(ids, ref_ids)]
        first = diff.index(True) if True in diff else len(diff)
        res["first_flip_step"] = first
        res["flipped_rows_last_step"] = int((ids[-1] != ref_ids[-1]).any(axis=1).sum())
        res["worst_before_flip"] = float(dev[:first].max()) if first > 0 else None
    return res


def cost(base, overrides, steps=12):
    """ms per optimize step (generator + discriminator step) at the headline workload under this assignment, hipGraph replay."""
    import bench
    from consistent__style_transfer_amd.graphs import GraphedStep
    w = bench.WORKLOADS[bench.HEADLINE]
    dev = torch.device("cuda", 0)
    ops.set_precision(base)
    _, _, opt = bench.build_stages(w, dev)
    assign(opt, base, overrides)
    bo = bench.make_batches(w, 0, dev, n=1)[0][2]
    c0 = bench.coins_tensor(0, w["L"], dev)

    def step(x, lab, coins, variant):
        ops.set_precision(base)
        return opt.train_step((x, lab), variant, coins=coins)

    g_d = GraphedStep(lambda x, lab, coins: step(x, lab, coins, 0), list(bo) + [c0], [opt])
    g_n = GraphedStep(lambda x, lab, coins: step(x, lab, coins, 1), list(bo) + [c0], [opt])
    for it in range(3):
        (g_d if it % 4 == 0 else g_n)(*bo, c0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        (g_d if it % 4 == 0 else g_n)(*bo, c0)
    torch.cuda.synchronize()
    ms = 1000.0 * (time.perf_counter() - t0) / steps
    del g_d, g_n, opt
    torch.cuda.empty_cache()
    ops.set_precision("bf16")
    return ms


def main():
    with_cost = "--no-cost" not in sys.argv
    runs = [("all exact (--precision f32)", "f32", {}), ("all bf16 (the benchmarked mode)", "bf16", {})]
    runs += [(f"{m} exact, rest bf16", "bf16", {m: "f32"}) for m in MODULES]
    runs += [("generator + matcher exact, rest bf16", "bf16", {"generator": "f32", "matcher": "f32"})]
    runs += [(f"{m} bf16, rest exact", "f32", {m: "bf16"}) for m in MODULES]
    out, ref_ids = [], None
    print(f"{'assignment':40s} {'worst':>9s} | " + " ".join(f"{c:>9s}" for c in COLS) + " | step 0: " + " ".join(f"{c:>9s}" for c in COLS) + " | ms/step (headline)")
    for label, base, ov in runs:
        r = curve(base, ov, ref_ids)
        if ref_ids is None:
            ref_ids = r["ids"]                              # first run = all exact: the token ids every other run is compared with
        del r["ids"]
        r["label"], r["base"], r["overrides"] = label, base, ov
        r["optimize_ms_per_step"] = cost(base, ov) if with_cost else None
        out.append(r)
        ms = f"{r['optimize_ms_per_step']:.2f}" if r["optimize_ms_per_step"] is not None else "-"
        print(f"{label:40s} {r['worst']:9.2e} | " + " ".join(f"{v:9.2e}" for v in r["max"]) + " |         " + " ".join(f"{v:9.2e}" for v in r["step0"]) + f" | {ms}"
              + (f" | first flipped token at step {r['first_flip_step']}, before it worst {r['worst_before_flip']}, {r['flipped_rows_last_step']}/16 sentences differ at step 19" if "first_flip_step" in r else ""), flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "precision_attrib.json"), "w") as f:
        json.dump({"fixture": "b16 optimize curve, 20 steps, lr 1e-4 (tests/golden/curves_b16.npz)", "columns": COLS, "runs": out}, f, indent=1)


if __name__ == "__main__":
    main()
