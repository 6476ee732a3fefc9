#!/usr/bin/env python3
"""Would the split engine's batched NT kernel (csrc/gru_gemm.hip) run the forward products of the [B,.] layers faster than the
grouped launches do?  The phases' NT products as same-shaped batches, no contraction split (q_proj / vector_fusion / classifier
would need one: their time here / parts + a slab epilogue is the estimate).   python tools/head_nt_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vqa_playground_pytorch_amd import _lib, ops  # noqa: E402
from gru_step_bench import gpu_time  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B = 512
    cases = [("q_proj_fwd (4 x 2400 -> 310)", 4, B, 310, 2400), ("gates (2 x 310 -> 2048)", 2, B, 2048, 310), ("h2 (3 x 310 -> 1020)", 3, B, 1020, 310),
             ("vector_fusion (1240 -> 1020)", 1, B, 1020, 1240), ("classifier (510 -> 2000)", 1, B, 2000, 510),
             ("classifier dx (2000 -> 510)", 1, B, 510, 2000), ("gates dx (2 x 2048 -> 310)", 2, B, 310, 2048), ("glimpse (4 x 2048 -> 155)", 4, B, 156, 2048)]
    for tag, G, M, N, K in cases:
        lda = (K + 3) // 4 * 4                      # rows 16-byte aligned (a [B,310] activation would be stored 312 wide)
        a = torch.randn(G, M, lda, device=dev)
        w = torch.randn(G, N, K, device=dev) / K ** 0.5
        c = torch.empty(G, M, N, device=dev)
        img = ops.split_weights(w)
        flops = 2.0 * G * M * N * K
        line = "%-32s %5.2f GFLOP" % (tag, flops / 1e9)
        for rb in ("7", "8"):
            _lib.set_option("VQA_GRU_GEMM_RB", rb)
            t = gpu_time(lambda: ops.gemm_nt_split_batched(a, 0, M * lda, lda, img, c, None, w, False, G, M, N, K), n=50)
            wgs = -(-M // (16 * int(rb))) * -(-N // 160) * G
            line += "   rb%s: %6.1f us (%5.1f TF/s, %3d workgroups, %2d chunks)" % (rb, t, flops / t / 1e6, wgs, -(-K // 64) * 2)
        t = gpu_time(lambda: torch.bmm(a[:, :, :K], w.transpose(1, 2)), n=50)
        print(line + "   library %6.1f us" % t, flush=True)


if __name__ == "__main__":
    main()
