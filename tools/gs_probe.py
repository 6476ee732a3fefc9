#!/usr/bin/env python3
"""Per-step and per-item cost of csrc/grouped_gemm_split.hip: ONE problem of 256 tiles (one per CU), contraction in one part,
timed at several K -- time(K) = fixed + steps * per_step.

    python tools/gs_probe.py [--form nt|nn|tn] [--cols 160|128] [--engine split|mfma]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from vqa_playground_pytorch_amd import _lib, head, ops  # noqa: E402

dev = torch.device("cuda:0")


def gpu_time(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--form", default="nt")
    ap.add_argument("--cols", type=int, default=160)
    ap.add_argument("--engine", default="split")
    ap.add_argument("--tiles", type=int, default=256)
    args = ap.parse_args()
    L_ = _lib.lib()
    form = {"nt": head.NT, "nn": head.NN, "tn": head.TN}[args.form]
    rows_t = 128 if args.engine == "split" else 64
    cols_t = args.cols if args.engine == "split" else 64
    tm = 16 if args.tiles >= 256 else 4
    tn = args.tiles // tm
    M, N = rows_t * tm, cols_t * tn
    print("form %s, %d x %d tiles of %d x %d (M = %d, N = %d), engine %s" % (args.form, tm, tn, rows_t, cols_t, M, N, args.engine))
    res = []
    for K in (64, 320, 640, 1280, 2560):
        if form == head.NT:
            a, b, lda, ldb = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), K, K
        elif form == head.NN:
            a, b, lda, ldb = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), K, N
        else:
            a, b, lda, ldb = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), M, N
        slab = torch.empty(M, N, device=dev)
        arr = (head.GemmProblem * 1)()
        arr[0] = head.GemmProblem(a.data_ptr(), b.data_ptr(), slab.data_ptr(), None, M * N, lda, ldb, M, N, K, form, K, 0, 0, 0, 0, 0)
        fn = L_.vqa_grouped_gemm_split if args.engine == "split" else L_.vqa_grouped_gemm
        st = ops._stream()

        def run():
            _lib.check(fn(arr, 1, st), "probe")

        run()
        ref = (a.double() @ b.double().t()) if form == head.NT else (a.double() @ b.double()) if form == head.NN else (a.double().t() @ b.double())
        err = float((slab.double() - ref).abs().max() / ref.abs().max())
        us = gpu_time(run)
        res.append((K, us))
        print("  K = %5d  %7.1f us   %6.1f TF/s   err %.1e" % (K, us, 2.0 * M * N * K / us / 1e6, err))
    (k0, t0), (k1, t1) = res[-2], res[-1]
    per = (t1 - t0) / ((k1 - k0) / 32)
    print("  per 32-deep step: %.2f us;  fixed: %.1f us" % (per, t1 - per * k1 / 32))


if __name__ == "__main__":
    main()
