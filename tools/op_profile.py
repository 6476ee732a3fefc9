#!/usr/bin/env python3
"""Which framework ops launch the small kernels of a CoR2 training step?  torch.profiler over one eager step, grouped by
(op, input shapes), device time.   python tools/op_profile.py [--model oda]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from vqa_playground_pytorch_amd import CoR2Model, ODAModel  # noqa: E402
from vqa_playground_pytorch_amd.trainer import DataParallelTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="cor2")
    ap.add_argument("--batch", type=int, default=512)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cls, nans = (CoR2Model, 2000) if args.model == "cor2" else (ODAModel, 3000)
    torch.manual_seed(0)
    model = cls(["PAD"], nans).to(dev).train()
    tr = DataParallelTrainer(model, graph=False)
    B = args.batch
    v, q = torch.randn(B, 36, 2048, device=dev), torch.randn(B, 2400, device=dev)
    a = torch.softmax(torch.randn(B, nans, device=dev), 1)
    for _ in range(3):
        tr.step({"v": v, "q_idxes": q}, a)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        tr.step({"v": v, "q_idxes": q}, a)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dt = getattr(e, "self_device_time_total", None)
        if dt is None:
            dt = e.self_cuda_time_total
        if dt > 0:
            rows.append((dt, e.count, e.key, str(e.input_shapes)[:110]))
    rows.sort(reverse=True)
    for dt, n, key, shapes in rows:
        if dt / n < 12.0:
            print("%8.1f us  x%-3d %-34s %s" % (dt, n, key[:34], shapes))


if __name__ == "__main__":
    main()
