#!/usr/bin/env python3
"""The question encoder's per-step product (3 x [B,2400] x [2400,2400]^T) on the split engine's batched NT kernel
(csrc/gru_gemm.hip) against torch.bmm (hipBLASLt), and the input projections' shape.  Knobs: VQA_GRU_GEMM_RB = 7 | 8 | 9,
VQA_GRU_GEMM_ORDER = col | row.    python tools/gru_step_bench.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vqa_playground_pytorch_amd import _lib, ops  # noqa: E402


def gpu_time(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(24_000_000)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (G, M, N, K, tag) in ((3, B, 2400, 2400, "recurrent step"), (3, 26 * B, 2400, 620, "input projections"),
                              (3, 26 * B, 620, 2400, "input projections, data gradient")):
        a = torch.randn(G, M, K, device=dev)
        w = torch.randn(G, N, K, device=dev) / K ** 0.5
        c = torch.empty(G, M, N, device=dev)
        img = ops.split_weights(w)
        flops = 2.0 * G * M * N * K
        t_lib = gpu_time(lambda: torch.bmm(a, w.transpose(1, 2)))
        line = "%-34s G=%d M=%5d N=%4d K=%4d  library %7.1f us (%5.1f TF/s)" % (tag, G, M, N, K, t_lib, flops / t_lib / 1e6)
        for rb in ("7", "8", "9"):
            for order in ("col", "row"):
                _lib.set_option("VQA_GRU_GEMM_RB", rb)
                _lib.set_option("VQA_GRU_GEMM_ORDER", order)
                t = gpu_time(lambda: ops.gemm_nt_split_batched(a, 0, M * K, K, img, c, None, w, False, G, M, N, K))
                line += "  rb%s/%s %6.1f (%5.1f)" % (rb, order, t, flops / t / 1e6)
        ref = torch.bmm(a.double(), w.double().transpose(1, 2))
        err = float((c.double() - ref).abs().max() / ref.abs().max())
        print(line + "   max err %.1e" % err, flush=True)


if __name__ == "__main__":
    main()
