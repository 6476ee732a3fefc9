"""relation_dgrad on the split engine, ablated (VQA_SPLIT_DGRAD_TUNE: 1 no main loop, 2 no v loads, 4 no epilogue math, 5 = 1 + 4; the first of the two
timings of a line runs on a colder chip and reads ~10 % high):
where its time goes at B = 512.  python tools/dgrad_split_ablate.py"""
import ctypes, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def run(tune, shared, L=310):
    from vqa_playground_pytorch_amd import _lib
    L_ = _lib.lib()
    B, N, D = 512, 36, 2048
    dev = torch.device("cuda:0")
    gz = torch.randn(B * N, L, device=dev) / 64
    w = torch.randn(L, D, device=dev) / 45
    v = torch.randn(B * N, D, device=dev)
    dt = torch.empty(B, D, device=dev); dc = torch.empty(B, D, device=dev)
    wsb = L_.vqa_relation_projection_dgrad_split_workspace_bytes(D, L)
    ws = torch.empty(wsb // 4 + 64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    def call(p):
        rc = L_.vqa_relation_projection_dgrad_split(gz.data_ptr(), w.data_ptr(), v.data_ptr(), dt.data_ptr(), dc.data_ptr(), ws.data_ptr(), wsb,
                                                    p, 77, None, B, N, D, L, s)
        assert rc == 0, rc
    out = []
    for p in (0.0, 0.5):
        for _ in range(10): call(p)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): call(p)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 30 * 1e3)
    print("L %d tune %s shared %s: %.1f us (no mask)  %.1f us (p = 0.5)   [incl. the 5 us W^T pack launch]" % (L, tune, shared, out[0], out[1]))

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 310)
    else:
        for tune, shared in (("0", "1"), ("0", "0"), ("2", "1"), ("4", "1"), ("1", "1"), ("5", "1")):
            env = dict(os.environ, VQA_SPLIT_DGRAD_TUNE=tune, VQA_SPLIT_DGRAD_SHARED=shared)
            subprocess.run([sys.executable, os.path.abspath(__file__), tune, shared], env=env, check=True)
        # rows of 320 floats (64-byte aligned) instead of 310 (8-byte aligned): the same ten chunks
        for L in (320, 310, 320):
            subprocess.run([sys.executable, os.path.abspath(__file__), "0", "1", str(L)], env=dict(os.environ, VQA_SPLIT_DGRAD_TUNE="0"), check=True)
