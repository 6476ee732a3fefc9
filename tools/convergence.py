#!/usr/bin/env python3
"""Does the replayed step train?  A teacher CoR2 head (fixed random weights, eval mode) labels synthetic batches with
its softmax answers; a student with other initial weights is trained on them with the reference's recipe (KLD-sum,
clip 0.25, Adam 1e-4, per-iteration exponential lr, dropout on) through DataParallelTrainer's hipGraph replay.
Prints the mean loss per sample over windows of steps and one JSON line; exits non-zero if the loss did not fall.
    python tools/convergence.py [--steps 1500] [--batch 512] [--model cor2|oda] [--no-graph] [--out file.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vqa_playground_pytorch_amd import CoR2Model, ODAModel  # noqa: E402
from vqa_playground_pytorch_amd.trainer import DataParallelTrainer  # noqa: E402


def teacher_answers(teacher, v, q, sharpness=3.0):
    """Soft targets: the teacher's logits standardised per sample and sharpened, so that the answer distribution is
    far from uniform (a freshly initialised head emits logits of ~1e-2) and depends on the regions and the question."""
    z = teacher({"v": v, "q_idxes": q})
    z = (z - z.mean(dim=1, keepdim=True)) / z.std(dim=1, keepdim=True)
    return torch.softmax(sharpness * z, dim=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--pool", type=int, default=16, help="distinct resident batches, cycled")
    ap.add_argument("--window", type=int, default=100)
    ap.add_argument("--model", default="cor2", choices=["cor2", "oda"])
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cls, nans = (CoR2Model, 2000) if args.model == "cor2" else (ODAModel, 3000)
    B = args.batch
    torch.manual_seed(1)
    teacher = cls(["PAD"], nans).to(dev).eval()
    torch.manual_seed(2)
    student = cls(["PAD"], nans).to(dev).train()
    g = torch.Generator(device="cpu").manual_seed(3)
    pool = []
    with torch.no_grad():
        for _ in range(args.pool):
            v = torch.randn(B, 36, 2048, generator=g).to(dev)
            q = torch.randn(B, 2400, generator=g).to(dev)
            a = teacher_answers(teacher, v, q)
            pool.append((v, q, a))
    del teacher
    tr = DataParallelTrainer(student, lr=args.lr, clip=0.25, graph=not args.no_graph)
    windows, acc, n = [], None, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(args.steps):
        v, q, a = pool[step % len(pool)]
        loss, _ = tr.step({"v": v, "q_idxes": q}, a)
        acc = loss.clone() if acc is None else acc + loss       # stays on the device: no host sync per step
        n += 1
        if n == args.window or step == args.steps - 1:
            windows.append(acc.item() / (n * B))
            acc, n = None, 0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # held-out check in eval mode: batches the student has never seen
    student.eval()
    torch.manual_seed(1)
    teacher = cls(["PAD"], nans).to(dev).eval()
    with torch.no_grad():
        v = torch.randn(B, 36, 2048, generator=g).to(dev)
        q = torch.randn(B, 2400, generator=g).to(dev)
        a = teacher_answers(teacher, v, q)
        from vqa_playground_pytorch_amd.trainer import kld_sum_loss
        held_out = kld_sum_loss(student({"v": v, "q_idxes": q}), a).item() / B
        uniform = (a * (torch.log(a.clamp_min(1e-30)) + torch.log(torch.tensor(float(nans))))).sum().item() / B
    rec = {"model": args.model, "batch": B, "steps": args.steps, "graph": tr._graph is not None, "lr": args.lr,
           "final_lr": tr.lr, "window": args.window, "loss_per_sample_by_window": [round(w, 5) for w in windows],
           "held_out_loss_per_sample_eval": round(held_out, 5), "uniform_answer_loss_per_sample": round(uniform, 5),
           "seconds": round(dt, 2), "samples_per_s": round(B * args.steps / dt, 1)}
    for i, w in enumerate(windows):
        print("steps %5d-%5d  mean KLD per sample %.4f" % (i * args.window, min((i + 1) * args.window, args.steps) - 1, w))
    print(json.dumps(rec))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)
    ok = windows[-1] < 0.8 * windows[0] and held_out < uniform
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
