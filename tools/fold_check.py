#!/usr/bin/env python3
"""Rank-folded K4 against the tile-engine K4 on the same inputs: max error and per-launch time.
    python tools/fold_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vqa_playground_pytorch_amd import _lib, ops  # noqa: E402

if os.environ.get("FOLD_CHECK_LIB"):          # an experimental build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["FOLD_CHECK_LIB"])


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    L_ = _lib.lib()
    only_first = bool(os.environ.get("FOLD_CHECK_FIRST"))
    for (B, N, L, H, R) in [(512, 36, 310, 510, 2)] if only_first else [(512, 36, 310, 510, 2), (5, 36, 310, 510, 2), (7, 37, 64, 130, 3), (3, 1, 30, 66, 4),
                            (128, 100, 310, 510, 2), (9, 16, 34, 64, 1), (512, 36, 310, 510, 1)]:
        torch.manual_seed(B + N)
        x = torch.randn(B, N, L, device=dev)
        h2 = torch.randn(B, R, H, device=dev)
        w1 = [torch.randn(H, L, device=dev) / L ** 0.5 for _ in range(R)]
        b1 = [torch.randn(H, device=dev) for _ in range(R)]
        out0 = torch.empty(B, N, H, device=dev)
        out1 = torch.full((B, N, H), float("nan"), device=dev)
        wp, bp = ops._ptr_array(w1), ops._ptr_array(b1)
        st = torch.cuda.current_stream().cuda_stream

        def old():
            rc = L_.vqa_lowrank_bilinear_fusion_fwd(ops._p(x), L, wp, bp, ops._p(h2), ops._p(out0), None, B, N, L, H, R, st)
            _lib.check(rc, "k4")

        def new():
            rc = L_.vqa_lowrank_bilinear_fusion_folded_fwd(ops._p(x), L, wp, bp, ops._p(h2), ops._p(out1), B, N, L, H, R, st)
            _lib.check(rc, "k4")

        old()
        new()
        torch.cuda.synchronize()
        ref = sum((x.double() @ w1[r].double().t() + b1[r].double()) * h2[:, r].double().unsqueeze(1) for r in range(R))
        e_old = ((out0 - ref).abs().max() / ref.abs().max()).item()
        e_new = ((out1 - ref).abs().max() / ref.abs().max()).item()
        t_old, t_new = timeit(old), timeit(new)
        print("B=%d N=%d L=%d H=%d R=%d fwd: err engine %.2e folded %.2e | engine %.1f us folded %.1f us (x%.2f)"
              % (B, N, L, H, R, e_old, e_new, t_old, t_new, t_old / t_new), flush=True)

        # ---- backward
        g = torch.randn(B, N, H, device=dev)
        h1 = torch.empty(B * N, R, H, device=dev)
        rc = L_.vqa_lowrank_bilinear_fusion_fwd(ops._p(x), L, wp, bp, ops._p(h2), ops._p(out0), ops._p(h1), B, N, L, H, R, st)
        _lib.check(rc, "k4")
        outs = []
        for _ in range(2):
            outs.append(dict(dx=torch.full((B, N, L), float("nan"), device=dev), dh2=torch.full((B, R, H), float("nan"), device=dev),
                             dw=[torch.full((H, L), float("nan"), device=dev) for _ in range(R)],
                             db=[torch.full((H,), float("nan"), device=dev) for _ in range(R)]))
        ws0_b = L_.vqa_lowrank_bilinear_fusion_bwd_workspace_bytes(B, N, L, H, R)
        ws1_b = L_.vqa_lowrank_bilinear_fusion_folded_bwd_workspace_bytes(B, N, L, H, R)
        ws0 = torch.empty(ws0_b // 4 + 4, device=dev)
        ws1 = torch.empty(ws1_b // 4 + 4, device=dev)
        dwp = [ops._ptr_array(o["dw"]) for o in outs]
        dbp = [ops._ptr_array(o["db"]) for o in outs]

        def old_b():
            o = outs[0]
            rc = L_.vqa_lowrank_bilinear_fusion_bwd(ops._p(x), L, wp, ops._p(h2), ops._p(h1), ops._p(g), ops._p(o["dx"]), dwp[0],
                                                    dbp[0], ops._p(o["dh2"]), ops._p(ws0), ws0_b, B, N, L, H, R, st)
            _lib.check(rc, "k4 bwd")

        def new_b():
            o = outs[1]
            rc = L_.vqa_lowrank_bilinear_fusion_folded_bwd(ops._p(x), L, wp, bp, ops._p(h2), ops._p(g), ops._p(o["dx"]), dwp[1],
                                                           dbp[1], ops._p(o["dh2"]), ops._p(ws1), ws1_b, B, N, L, H, R, st)
            _lib.check(rc, "k4 folded bwd")

        old_b()
        new_b()
        torch.cuda.synchronize()
        xd, gd, hd = x.double(), g.double(), h2.double()
        ref_dx = sum((gd * hd[:, r].unsqueeze(1)) @ w1[r].double() for r in range(R))
        ref_dw = [((gd * hd[:, r].unsqueeze(1)).reshape(-1, H).t() @ xd.reshape(-1, L)) for r in range(R)]
        ref_db = [(gd * hd[:, r].unsqueeze(1)).sum((0, 1)) for r in range(R)]
        ref_dh2 = torch.stack([(gd * (xd @ w1[r].double().t() + b1[r].double())).sum(1) for r in range(R)], 1)

        def err(a, b_):
            return ((a - b_).abs().max() / b_.abs().max()).item()

        for name, o in zip(("engine", "folded"), outs):
            print("   bwd %s: dx %.2e dw %.2e db %.2e dh2 %.2e" % (
                name, err(o["dx"], ref_dx), max(err(o["dw"][r], ref_dw[r]) for r in range(R)),
                max(err(o["db"][r], ref_db[r]) for r in range(R)), err(o["dh2"], ref_dh2)), flush=True)
        t_old, t_new = timeit(old_b), timeit(new_b)
        print("   bwd: engine %.1f us folded %.1f us (x%.2f)" % (t_old, t_new, t_old / t_new), flush=True)


if __name__ == "__main__":
    main()
