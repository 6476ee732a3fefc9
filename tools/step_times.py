#!/usr/bin/env python3
"""Per-step GPU time of the first replays after bench.py's warm-up: is the driver's 20-step region steady state?
    python tools/step_times.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vqa_playground_pytorch_amd import CoR2Model  # noqa: E402
from vqa_playground_pytorch_amd.trainer import DataParallelTrainer  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, ROT = 512, 4
    model = CoR2Model(["PAD"], 2000).to(dev).train()
    tr = DataParallelTrainer(model, lr=1e-4, clip=0.25, graph=True, adopt_inputs=True, input_slots=ROT)
    batches = [({"v": torch.randn(B, 36, 2048, device=dev), "q_idxes": torch.randn(B, 2400, device=dev)},
                torch.softmax(2.0 * torch.randn(B, 2000, device=dev), dim=1)) for _ in range(ROT)]
    for i in range(2 + 2 * ROT):
        tr.step(*batches[i % ROT])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        tr.step(*batches[i % ROT])
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print("per-step ms:", " ".join("%.3f" % x for x in ms))
    for lo in range(0, n, 10):
        print("  steps %2d-%2d: mean %.4f ms" % (lo, lo + 9, sum(ms[lo:lo + 10]) / len(ms[lo:lo + 10])))


if __name__ == "__main__":
    main()
