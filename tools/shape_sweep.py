#!/usr/bin/env python3
"""Robustness sweep of the CoR2 head over region counts and batch sizes: forward + backward in eval mode with K4 in
its rank-folded and in its R-GEMM form must agree (logits, loss, every parameter gradient), for shapes on both sides of
every kernel-variant boundary (16-region blocks, 4/8-sample groups, the 32-sample form switch, N > 112).
    python tools/shape_sweep.py"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vqa_playground_pytorch_amd import CoR2Model, ops  # noqa: E402
from vqa_playground_pytorch_amd.trainer import kld_sum_loss  # noqa: E402


def run(model, v, q, a, form):
    ops._K4_FORM = form
    model.zero_grad(set_to_none=True)
    logits = model({"v": v, "q_idxes": q})
    loss = kld_sum_loss(logits, a)
    loss.backward()
    return logits.detach().clone(), loss.item(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = CoR2Model(["PAD"], 300).to(dev).eval()
    worst = 0.0
    for N, B in itertools.product((1, 5, 16, 17, 36, 48, 49, 80, 81, 112, 113, 150), (1, 7, 31, 33, 130)):
        g = torch.Generator(device="cpu").manual_seed(N * 1000 + B)
        v = torch.randn(B, N, 2048, generator=g).to(dev)
        q = torch.randn(B, 2400, generator=g).to(dev)
        a = torch.softmax(torch.randn(B, 300, generator=g), 1).to(dev)
        l0, s0, g0 = run(model, v, q, a, "folded")
        l1, s1, g1 = run(model, v, q, a, "engine")
        # relative L2 difference per tensor: a pre-activation that is zero to rounding may take the other side of a relu
        # in the second run and move ONE term of a weight gradient (visible in a max-abs metric, not a kernel difference)
        err, who = ((l0 - l1).norm() / l1.norm().clamp_min(1e-12)).item(), "logits"
        for n in g0:
            scale = g1[n].norm().item()
            if n.endswith(".bias"):   # some bias gradients are mathematically zero (anything that shifts the logits of all
                # regions alike vanishes in the softmax over regions): measure them against their layer's weight gradient
                scale = max(scale, g1[n[:-4] + "weight"].norm().item())
            e = (g0[n] - g1[n]).norm().item() / max(scale, 1e-12)
            if e > err:
                err, who = e, n
        # (a single flipped relu term of a heavy-tailed gradient row can be 2e-3 of a region projection's weight gradient)
        tol = 1e-2 if who.startswith("compress_v") else 1e-3
        ok = err < tol and abs(s0 - s1) <= 1e-4 * abs(s1) and all(torch.isfinite(t).all() for t in g0.values())
        worst = max(worst, err)
        print("N=%3d B=%3d  loss %.4f / %.4f  worst rel diff %.2e (%s)  %s" % (N, B, s0, s1, err, who, "ok" if ok else "MISMATCH"),
              flush=True)
        if not ok:
            sys.exit(1)
    print("all shapes agree; worst relative difference %.2e" % worst)


if __name__ == "__main__":
    main()
