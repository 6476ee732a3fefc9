#!/usr/bin/env python3
"""Grouped-GEMM phases of the CoR2 step (B = 512) as raw problem tables -- no autograd, no epilogue fusion: one
vqa_grouped_gemm + one vqa_grouped_epilogue (plain slab sums) per phase -- against the library on the same products.

    [VQA_GROUPED_ENGINE=split|mfma] python tools/gg_bench.py [--rounds 5] [--batch 512] [--check] [--gemm-only]
    rocprofv3 --kernel-trace --output-format csv -d /tmp/gg -o gg -- python3 tools/gg_bench.py --rounds 2 ; python tools/by_grid.py /tmp/gg 1

GPU-side time: each batch of calls is queued behind a device-side sleep (the Python cost of a phase exceeds its kernels)."""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from vqa_playground_pytorch_amd import head  # noqa: E402

dev = torch.device("cuda:0")
NT, NN, TN = head.NT, head.NN, head.TN


def phases(B):
    return {
        "q_proj_fwd": [(NT, B, 310, 2400)] * 4,
        "q_proj_bwd": [(TN, 310, 2400, B)] * 4,
        "gates_h2_fwd": [(NT, B, 2048, 310)] * 2 + [(NT, B, 1020, 310)] * 3,
        "gates_h2_bwd": [(NN, B, 310, 2048)] * 2 + [(NN, B, 310, 1020)] * 3 + [(TN, 2048, 310, B)] * 2 + [(TN, 1020, 310, B)] * 3,
        "vector_fusion_fwd": [(NT, B, 1020, 1240), (NT, B, 1020, 310)],
        "vector_fusion_bwd": [(NN, B, 1240, 1020), (TN, 1020, 1240, B), (NN, B, 310, 1020), (TN, 1020, 310, B)],
        "classifier_fwd": [(NT, B, 2000, 510)],
        "classifier_bwd": [(NN, B, 510, 2000), (TN, 2000, 510, B)],
        "glimpse_fwd": [(NT, B, 155, 2048)] * 4,
        "glimpse_bwd": [(NN, B, 2048, 155)] * 4 + [(TN, 155, 2048, B)] * 4,
        # round 6: the backward with every weight-gradient product pulled out of its phase into ONE launch at the end of backward
        # (K = B for all of them: no contraction split, direct output), the phases keeping their data-gradient products
        "deferred_dw": [(TN, 2000, 510, B)] + [(TN, 1020, 620, B)] * 2 + [(TN, 155, 2048, B)] * 8 + [(TN, 2048, 310, B)] * 2
        + [(TN, 1020, 310, B)] * 3 + [(TN, 310, 2400, B)] * 4,
        "deferred_dw_noglimpse": [(TN, 2000, 510, B)] + [(TN, 1020, 620, B)] * 2 + [(TN, 2048, 310, B)] * 2
        + [(TN, 1020, 310, B)] * 3 + [(TN, 310, 2400, B)] * 4,
        "classifier_dx": [(NN, B, 510, 2000)],
        "vector_fusion_dx": [(NN, B, 620, 1020)] * 2,
        "gates_h2_dx": [(NN, B, 310, 2048)] * 2 + [(NN, B, 310, 1020)] * 3,
        "glimpse_dx": [(NN, B, 2048, 155)] * 4,
    }


def operands(form, M, N, K):
    if form == NT:
        return torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5
    if form == NN:
        return torch.randn(M, K, device=dev), torch.randn(K, N, device=dev) / K ** 0.5
    return torch.randn(K, M, device=dev), torch.randn(K, N, device=dev) / K ** 0.5


def library(form, a, b):
    if form == NT:
        return a @ b.t()
    if form == NN:
        return a @ b
    return a.t() @ b


def gpu_time(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(48_000_000)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--gemm-only", action="store_true", help="skip the epilogue launches: back-to-back GEMM launches alone")
    args = ap.parse_args()
    if args.gemm_only:
        head.Phase._epilogue = lambda self, L_, jobs, tag: None
    torch.manual_seed(0)
    total_g = total_l = 0.0
    print("== grouped GEMM phases vs library GEMMs, B = %d (us, median of %d rounds; FLOP rate of the grouped form)" % (args.batch, args.rounds))
    for name, probs in phases(args.batch).items():
        if args.only and name not in args.only.split(","):
            continue
        ops_ = [operands(*p) for p in probs]
        outs = [torch.empty(p[1], p[2], device=dev) for p in probs]
        flops = sum(2.0 * p[1] * p[2] * p[3] for p in probs)
        info = {}

        def grouped():
            ph = head.Phase(dev, name)
            for (form, M, N, K), (a, b), o in zip(probs, ops_, outs):
                t = ph.target(M, N)
                if a.shape[1] % 2:            # a 155-wide operand: the 4-byte-aligned forms (head.GlimpseProjections)
                    form = {NN: head.NN_A4, TN: head.TN_A4}[form]
                ph.gemm(t, form, a, a.shape[1], b, b.shape[1], K)
                ph.job(head.EPI_SUM, t, o, N)
            if not info:
                sized = ph._size()
                info["parts"] = sorted({(p["ksplit"], p["splits"]) for _, p in sized})
                if ph._engine() == "split":
                    info["items"] = sum(-(-t.M // 128) * -(-t.N // head.Phase.split_tile_cols(t.N)) * p["splits"] for t, p in sized)
                else:
                    info["items"] = sum(-(-t.M // head.Phase.TILE_M) * -(-t.N // 64) * p["splits"] for t, p in sized)
            ph.run()

        def lib():
            return [library(p[0], a, b) for p, (a, b) in zip(probs, ops_)]

        with torch.no_grad():
            if args.check:
                grouped()
                worst = 0.0
                for p, (a, b), o in zip(probs, ops_, outs):
                    ref = library(p[0], a.double(), b.double())
                    worst = max(worst, float((o.double() - ref).abs().max() / ref.abs().max()))
                assert worst < 2e-4, (name, worst)
            tg = statistics.median(gpu_time(grouped) for _ in range(args.rounds))
            tl = statistics.median(gpu_time(lib) for _ in range(args.rounds))
        total_g += tg
        total_l += tl
        print("  %-18s %2d products  grouped %6.1f  (%5.1f TF/s)   library %6.1f   items %4d, (ksplit, parts) %s"
              % (name, len(probs), tg, flops / tg / 1e6, tl, info["items"], info["parts"]))
    print("  %-18s              grouped %6.1f                 library %6.1f" % ("total", total_g, total_l))


if __name__ == "__main__":
    main()
