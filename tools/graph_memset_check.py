#!/usr/bin/env python3
"""Diagnostic: the K1 backward (zero-fill + atomically accumulating kernel) captured into a hipGraph and replayed four
times; every replay must reproduce the eager result.  With hipMemsetAsync as the zero-fill (a graph memset node) only
the first replay was right on ROCm 7.2 -- errors of 1e-3 at B <= 128 and 1e16 at B >= 512 from the second replay on --
which is why the library zero-fills with its own kernel (csrc/api.hip).  Run on the GPU box: python tools/graph_memset_check.py
"""
import sys, torch, ctypes
sys.path.insert(0, ".")
from vqa_playground_pytorch_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
def test(B, N=36, D=2048, G=4):
    torch.manual_seed(0)
    v = torch.randn(B, N, D, device=dev); q1 = torch.rand(B, D, device=dev); q2 = torch.rand(B, D, device=dev)
    al = torch.softmax(torch.randn(B, N, G, device=dev), 1); g = torch.randn(B, N, D, device=dev)
    d_alpha = torch.empty(B, N, device=dev); d_q1 = torch.empty(B, D, device=dev); d_q2 = torch.empty(B, D, device=dev)
    def call():
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = L.vqa_pairwise_relation_reduce_bwd(v.data_ptr(), q1.data_ptr(), q2.data_ptr(), al.data_ptr(), G, g.data_ptr(),
                                                d_alpha.data_ptr(), d_q1.data_ptr(), d_q2.data_ptr(), None, B, N, D, s)
        assert rc == 0
    call(); torch.cuda.synchronize(); ref = d_alpha.clone()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): call()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): call()
    out = []
    for i in range(4):
        gr.replay(); torch.cuda.synchronize()
        out.append(((d_alpha - ref).abs().max() / ref.abs().max()).item())
    print("B", B, "bytes", B * N * 4, "replay errs", out, flush=True)
for B in (6, 64, 128, 512, 2048):
    test(B)
