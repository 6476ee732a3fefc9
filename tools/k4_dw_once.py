#!/usr/bin/env python3
"""K4's folded backward at the headline shape, a few launches (for rocprofv3 counter passes over the weight-gradient kernel):
    VQA_K4_DW_SPLIT=1|0 python tools/k4_dw_once.py [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_playground_pytorch_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, N, L, H, R = 512, 36, 310, 510, 2
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.relu(torch.randn(B, N, L, generator=g)).to(dev).requires_grad_()
h2 = torch.randn(B, R, H, generator=g).to(dev).requires_grad_()
ws = [(torch.randn(H, L, generator=g) / L ** 0.5).to(dev).requires_grad_() for _ in range(R)]
bs = [(0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_() for _ in range(R)]
go = torch.randn(B, N, H, generator=g).to(dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    out = ops.lowrank_bilinear_fusion(x, h2, ws, bs)
    out.backward(go)
torch.cuda.synchronize()
print("ok", float(ws[0].grad.abs().sum()))
