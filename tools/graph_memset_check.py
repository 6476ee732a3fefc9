#!/usr/bin/env python3
"""Diagnostic for the hipGraph memset-node hazard on ROCm 7.2 (run on the GPU box: python tools/graph_memset_check.py).

Part 1: raw hipMemsetAsync captured into a torch hipGraph, followed by a kernel that adds 1 to the buffer; the buffer
        must read 1 after every replay.
Part 2: the library's K1 backward (zero-fill + atomically accumulating kernel) captured and replayed; every replay must
        reproduce the eager result.  With hipMemsetAsync as the zero-fill only the first replay was right, which is why
        the library zero-fills with its own kernel (csrc/api.hip) and the loss is a fused kernel (csrc/loss.hip) instead
        of torch's multi-block sum (whose semaphores are zeroed by a memset).
"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def raw_memset(n_floats, pool_alloc):
    buf = torch.full((n_floats,), 5.0, device=dev) if not pool_alloc else None
    canary = torch.full((1024,), 7.0, device=dev)

    def body(b):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert hip.hipMemsetAsync(ctypes.c_void_p(b.data_ptr()), 0, b.numel() * 4, s) == 0
        b.add_(1.0)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body(buf if buf is not None else torch.empty(n_floats, device=dev))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        if buf is None:
            buf = torch.empty(n_floats, device=dev)   # lives in the graph's private pool
        body(buf)
    out = []
    for _ in range(4):
        g.replay()
        torch.cuda.synchronize()
        out.append((buf.min().item(), buf.max().item()))
    print("raw hipMemsetAsync, %8d floats, buffer in %s: (min,max) per replay %s  canary ok %s"
          % (n_floats, "graph pool" if pool_alloc else "normal pool", out, bool((canary == 7.0).all())))


def k1_backward(B, N=36, D=2048, G=4):
    from vqa_playground_pytorch_amd import _lib
    L = _lib.lib()
    torch.manual_seed(0)
    v = torch.randn(B, N, D, device=dev)
    q1, q2 = torch.rand(B, D, device=dev), torch.rand(B, D, device=dev)
    al = torch.softmax(torch.randn(B, N, G, device=dev), 1)
    g = torch.randn(B, N, D, device=dev)
    d_alpha = torch.empty(B, N, device=dev)
    d_q1, d_q2 = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)

    def call():
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = L.vqa_pairwise_relation_reduce_bwd(v.data_ptr(), q1.data_ptr(), q2.data_ptr(), al.data_ptr(), G, g.data_ptr(),
                                                None, d_alpha.data_ptr(), d_q1.data_ptr(), d_q2.data_ptr(), None, B, N, D, s)
        assert rc == 0

    call()
    torch.cuda.synchronize()
    ref = d_alpha.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        call()
    out = []
    for _ in range(4):
        gr.replay()
        torch.cuda.synchronize()
        out.append(((d_alpha - ref).abs().max() / ref.abs().max()).item())
    print("K1 backward B=%d: relative error of d_alpha per replay %s" % (B, ["%.1e" % e for e in out]))


if __name__ == "__main__":
    for n in (216, 4608, 18432, 1 << 20):
        raw_memset(n, False)
        raw_memset(n, True)
    for B in (6, 512):
        k1_backward(B)
