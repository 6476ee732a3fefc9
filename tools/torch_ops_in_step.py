#!/usr/bin/env python3
"""Which torch (non-libvqa) ops does one CoR2 training step still issue?  One eager fwd+bwd+optimizer step at B = 512 under
torch.profiler with shapes; prints the aten ops that launch GPU kernels.
    python tools/torch_ops_in_step.py [oda|bf16]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_playground_pytorch_amd import CoR2Model, ODAModel  # noqa: E402
from vqa_playground_pytorch_amd.trainer import DataParallelTrainer  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
oda = len(sys.argv) > 1 and sys.argv[1] == "oda"
bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"       # BASELINE configs[4]: 100 regions, 128 per rank
kw = {"compute_dtype": torch.bfloat16} if bf16 else {}
model = (ODAModel if oda else CoR2Model)(["PAD"], 2000, **kw).to(dev).train()
tr = DataParallelTrainer(model, lr=1e-4, graph=False)
B = 128 if bf16 else 512
v, q = torch.randn(B, 100 if bf16 else 36, 2048, device=dev), torch.randn(B, 2400, device=dev)
a = torch.softmax(torch.randn(B, 2000, device=dev), 1)
for _ in range(3):
    tr.step({"v": v, "q_idxes": q}, a)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step({"v": v, "q_idxes": q}, a)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0)
    if t and (e.name.startswith("aten::") or "emcpy" in e.name or "emset" in e.name):
        stack = [s for s in (e.stack or []) if "vqa_playground_pytorch_amd" in s or "tools/" in s]
        rows.append((t, e.name, str(e.input_shapes)[:70], stack[0][-70:] if stack else ""))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print("aten ops with GPU time: %d, %.1f us" % (len(rows), tot))
for t, n, s, st in rows:
    print("%7.1f us  %-28s %-70s %s" % (t, n, s, st))
