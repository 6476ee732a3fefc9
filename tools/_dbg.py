import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_playground_pytorch_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, N, L, G) in [(1, 1, 5, 1), (2, 3, 20, 2), (1, 2, 5, 1), (1, 1, 16, 1), (1,1,5,4)]:
    vl = torch.randn(B, N, L, device=dev, requires_grad=True); ql = torch.randn(B, L, device=dev, requires_grad=True)
    w = torch.randn(G, N * L, device=dev, requires_grad=True); bias = torch.zeros(G, device=dev, requires_grad=True)
    out = ops.object_difference_attention(vl, ql, w, bias, 0.0, 0)
    gl = torch.randn_like(out)
    out.backward(gl)
    # reference
    T = (vl * ql[:, None, :]).detach().double()
    X = (T[:, :, None, :] - T[:, None, :, :])          # [B,i,j,d]
    dw = torch.einsum("big,bijd->gjd", gl.double(), X).reshape(G, N * L)
    print((B, N, L, G), "d_w got", w.grad.flatten()[:6].tolist(), "want", dw.flatten()[:6].tolist())
