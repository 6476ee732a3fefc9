#!/usr/bin/env python3
"""Per-kernel micro-benchmark of libvqa_mi355x.so at the BASELINE shapes (B=512, N=36, D=2048, L=310, H=510, G=4, R=2).
Interleaved rounds in ONE process (guide rule 24); prints median / min per variant and the roofline fraction.

    python tools/kbench.py [--only k4fwd,k4bwd,k1,k2,k3,k3a,k5 | bf16 | head] [--tiles 128x128,64x64] [--rounds 20]
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vqa_playground_pytorch_amd import _lib, ops  # noqa: E402

B, N, D, L, H, G, R = 512, 36, 2048, 310, 510, 4, 2
dev = torch.device("cuda:0")


def timeit(fns, rounds, inner=5):
    """fns: {label: callable}; returns {label: [ms per call for each round]}"""
    out = {k: [] for k in fns}
    for k, f in fns.items():
        f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(inner):
                f()
            b.record()
            torch.cuda.synchronize()
            out[k].append(a.elapsed_time(b) / inner)
    return out


def report(title, res, work, unit):
    print("== " + title)
    for k, ms in res.items():
        med, mn = statistics.median(ms), min(ms)
        if unit == "TF":
            print("  %-28s median %8.1f us  min %8.1f us   %6.1f TF/s (%.1f%% of 157.3)" % (k, med * 1e3, mn * 1e3, work / med / 1e9, 100 * work / med / 1e9 / 157.3))
        else:
            print("  %-28s median %8.1f us  min %8.1f us   %6.0f GB/s (%.1f%% of 8000)" % (k, med * 1e3, mn * 1e3, work / med / 1e6, 100 * work / med / 1e6 / 8000))


def bf16_section(rounds):
    """bf16 engine at the configs[4] shapes (B=128, N=100: M=12800) against torch's bf16 matmul (hipBLASLt)."""
    M = 12800
    shapes_nt = [("compress fwd  [M,2048]x[320,2048]^T", M, 320, 2048), ("compress dx   [M,320]x[2048,320]^T", M, 2048, 320),
                 ("K4 dx         [M,1024]x[320,1024]^T", M, 320, 1024)]
    for title, m, n, k in shapes_nt:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        b = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
        fns = {"torch (hipBLASLt)": lambda: torch.nn.functional.linear(a, b)}
        for t in ("128x128", "128x64", "64x128", "64x64"):
            fns["engine NT " + t] = (lambda t=t: (_lib.set_option("VQA_BF16_TILE", t), ops.gemm_bf16_nt(a, b))[1])
        res = timeit(fns, rounds)
        _lib.set_option("VQA_BF16_TILE", None)
        print("== bf16 NT " + title)
        for key, ms in res.items():
            med = statistics.median(ms)
            print("  %-22s median %7.1f us  %7.1f TF/s (%.1f%% of 2500)" % (key, med * 1e3, 2.0 * m * n * k / med / 1e9, 2.0 * m * n * k / med / 1e9 / 25))
    shapes_tn = [("compress dW   [M,320]^T x [M,2048]", M, 320, 2048), ("K4 dW         [M,1024]^T x [M,320]", M, 1024, 320)]
    for title, kd, n1, n2 in shapes_tn:
        a = torch.randn(kd, n1, device=dev).to(torch.bfloat16)
        b = torch.randn(kd, n2, device=dev).to(torch.bfloat16)
        fns = {"torch (hipBLASLt)": lambda: a.t() @ b}
        for t in ("128x128", "128x64", "64x128", "64x64"):
            fns["engine TN " + t] = (lambda t=t: (_lib.set_option("VQA_BF16_TILE", t), ops.gemm_bf16_tn(a, b))[1])
        res = timeit(fns, rounds)
        _lib.set_option("VQA_BF16_TILE", None)
        print("== bf16 TN " + title)
        for key, ms in res.items():
            med = statistics.median(ms)
            print("  %-22s median %7.1f us  %7.1f TF/s (%.1f%% of 2500)" % (key, med * 1e3, 2.0 * kd * n1 * n2 / med / 1e9, 2.0 * kd * n1 * n2 / med / 1e9 / 25))


def head_section(rounds):
    """Grouped phases of the [B,.] layers (head.py: one grouped GEMM launch + one grouped epilogue launch) against the same
    layers on torch / library GEMMs, B = 512, forward only, GPU-side time (each batch of calls is queued behind a device-side
    sleep: the Python cost of a phase is larger than its kernels)."""
    from vqa_playground_pytorch_amd import head
    Bh = 512
    q = torch.randn(Bh, 2400, device=dev)
    ws = [torch.randn(310, 2400, device=dev) / 49 for _ in range(4)]
    bs = [torch.zeros(310, device=dev) for _ in range(4)]
    x = torch.randn(Bh, 510, device=dev)
    wc, bc = torch.randn(2000, 510, device=dev) / 22, torch.zeros(2000, device=dev)
    wcat, bcat = torch.cat(ws), torch.cat(bs)

    def gpu_time(f, n=30):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(48_000_000)
        a.record()
        for _ in range(n):
            f()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3

    cases = {
        "4 x [512,2400]->310 + bias + relu   grouped     ": lambda: head.QuestionProjections.apply(q, 0.0, 0, (), 0.0, 0, (), *ws, *bs),
        "   (as ONE [512,2400]->1240 GEMM)    torch       ": lambda: torch.relu(torch.nn.functional.linear(q, wcat, bcat)),
        "[512,510]->2000 + bias               grouped     ": lambda: head.Classifier.apply(x, wc, bc, 0.0, 0),
        "                                     torch linear": lambda: torch.nn.functional.linear(x, wc, bc),
    }
    print("== grouped head phases vs library, B = %d (us per phase, median of %d rounds)" % (Bh, rounds))
    with torch.no_grad():
        for name, f in cases.items():
            print("  %s %7.1f" % (name, statistics.median(gpu_time(f) for _ in range(rounds))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--tiles", default="128x128x1,128x128x2,64x128x2,64x128x3,128x64x2,128x64x3,64x64x1,64x64x2,64x64x3")
    ap.add_argument("--rounds", type=int, default=15)
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    if only == {"bf16"}:
        return bf16_section(args.rounds)
    if only == {"head"}:
        return head_section(min(args.rounds, 5))
    want = lambda k: not only or k in only  # noqa: E731
    tiles = args.tiles.split(",")
    torch.manual_seed(0)
    v = torch.randn(B, N, D, device=dev)
    g_v = torch.randn(B, N, D, device=dev)
    q1, q2 = torch.rand(B, D, device=dev), torch.rand(B, D, device=dev)
    logits = torch.randn(B, N, G, device=dev)
    alpha = torch.softmax(logits, 1)
    x = torch.randn(B, N, L, device=dev)
    h2 = torch.randn(B, R, H, device=dev)
    ws = [torch.randn(H, L, device=dev) / L ** 0.5 for _ in range(R)]
    bs = [torch.randn(H, device=dev) * 0.1 for _ in range(R)]
    gout = torch.randn(B, N, H, device=dev)
    gp = torch.randn(B, G, D, device=dev)

    def set_tile(t):      # (the library reads its knobs from the environment once: switch them through vqa_set_option)
        _lib.set_option("VQA_GEMM_TILE", t or None)

    if want("k4fwd"):
        fns = {}
        for t in tiles + [""]:
            def f(t=t):
                set_tile(t)
                ops.LowRankBilinearFusion.apply(x, h2, False, *ws, *bs)
            fns["fwd(no h1) tile=%s" % (t or "auto")] = f
        xr = x.clone().requires_grad_()
        for t in tiles:
            def f(t=t):
                set_tile(t)
                ops.LowRankBilinearFusion.apply(xr, h2, False, *ws, *bs)
            fns["fwd(+h1)  tile=%s" % t] = f
        report("K4 forward  (M=18432, K=310, N=2x510)", timeit(fns, args.rounds), B * (2 * R * N * L * H + 2 * R * N * H), "TF")
        set_tile("")
    if want("k4bwd"):
        fns = {}
        xr = x.clone().requires_grad_()
        wr = [w.clone().requires_grad_() for w in ws]
        br = [b.clone().requires_grad_() for b in bs]
        for t in tiles + [""]:
            set_tile(t)
            out = ops.LowRankBilinearFusion.apply(xr, h2, False, *wr, *br)

            def f(t=t, out=out):
                set_tile(t)
                torch.autograd.grad(out, [xr] + wr + br, gout, retain_graph=True)
            fns["bwd tile=%s" % (t or "auto")] = f
        report("K4 backward (dx + dW + dh2 + db)", timeit(fns, args.rounds), B * 4 * R * N * L * H, "TF")
        set_tile("")
    if want("k1"):
        fns = {"fwd factored": lambda: ops.pairwise_relation_reduce(v, q1, q2, alpha, 0, 1),
               "fwd pairwise": lambda: ops.pairwise_relation_reduce(v, q1, q2, alpha, 0, 0)}
        report("K1 forward", timeit(fns, args.rounds), B * (2 * N * D + 2 * D + N) * 4, "GB")
        q1r, q2r, ar = q1.clone().requires_grad_(), q2.clone().requires_grad_(), alpha.clone().requires_grad_()
        out = ops.pairwise_relation_reduce(v, q1r, q2r, ar, 0, 1)
        fns = {"bwd (no dv)": lambda: torch.autograd.grad(out, [q1r, q2r, ar], g_v, retain_graph=True)}
        report("K1 backward", timeit(fns, args.rounds), B * (2 * N * D + 4 * D + 2 * N) * 4, "GB")
    if want("k3"):
        fns = {"fwd": lambda: ops.softmax_attention_pool(logits, v)}
        report("K3 forward", timeit(fns, args.rounds), B * (N * D + 2 * N * G + G * D) * 4, "GB")
        lr, vr = logits.clone().requires_grad_(), v.clone().requires_grad_()
        a1, p1 = ops.softmax_attention_pool(lr, v)
        a2, p2 = ops.softmax_attention_pool(lr, vr)
        fns = {"bwd (no dv)": lambda: torch.autograd.grad(p1, [lr], gp, retain_graph=True),
               "bwd (+dv)": lambda: torch.autograd.grad(p2, [lr, vr], gp, retain_graph=True)}
        res = timeit(fns, args.rounds)
        report("K3 backward (no dv)", {"bwd (no dv)": res["bwd (no dv)"]}, B * (N * D + G * D + 3 * N * G) * 4, "GB")
        report("K3 backward (+dv)", {"bwd (+dv)": res["bwd (+dv)"]}, B * (2 * N * D + G * D + 3 * N * G) * 4, "GB")
    if want("k3a"):
        fz = torch.randn(B, N, H, device=dev)
        wa, ba = torch.randn(G, H, device=dev) / H ** 0.5, torch.zeros(G, device=dev)
        pa = float(os.environ.get("K3A_P", "0.5"))
        fns = {"fwd (p=%g)" % pa: lambda: ops.attention_logits(fz, wa, ba, pa, 7)}
        report("K3a logits forward", timeit(fns, args.rounds), B * N * (H + G) * 4, "GB")
        fr, wr_ = fz.clone().requires_grad_(), wa.clone().requires_grad_()
        out = ops.attention_logits(fr, wr_, ba, pa, 7)
        gl = torch.randn(B, N, G, device=dev)
        fns = {"bwd (dx + dw)": lambda: torch.autograd.grad(out, [fr, wr_], gl, retain_graph=True)}
        report("K3a logits backward", timeit(fns, args.rounds), B * N * (2 * H + G) * 4, "GB")
    if want("k1k5"):
        wp = torch.randn(L, D, device=dev) / D ** 0.5
        bp = torch.zeros(L, device=dev)
        tt, cc = torch.randn(B, D, device=dev).requires_grad_(), torch.rand(B, D, device=dev).requires_grad_()
        wr2 = wp.clone().requires_grad_()
        yy = ops.relation_projection(v, tt, cc, wr2, bp, 0.5, 11)
        gyy = torch.randn(B, N, L, device=dev)
        fns = {"bwd (dW + fused d_t/d_c2)": lambda: torch.autograd.grad(yy, [tt, cc, wr2], gyy, retain_graph=True)}
        report("K1->K5 backward", timeit(fns, args.rounds), 2 * 2 * B * N * D * L, "TF")
    if want("k5"):
        w = torch.randn(L, D, device=dev) / D ** 0.5
        bias = torch.randn(L, device=dev) * 0.1
        fns = {}
        for t in tiles + [""]:
            def f(t=t):
                set_tile(t)
                ops.LinearAct.apply(v, w, bias, 1, 0.0, 0)
            fns["fwd p=0 tile=%s" % (t or "auto")] = f
        def fd():
            set_tile("")
            ops.LinearAct.apply(v, w, bias, 1, 0.5, 77)
        fns["fwd p=0.5 tile=auto"] = fd
        fns["torch: dropout+F.linear+relu"] = lambda: torch.relu(torch.nn.functional.linear(torch.nn.functional.dropout(v, 0.5, True), w, bias))
        report("K5 forward (compress_v: M=18432, K=2048, N=310)", timeit(fns, args.rounds), 2 * B * N * D * L, "TF")
        set_tile("")
        vr = v.clone().requires_grad_()
        wr, br = w.clone().requires_grad_(), bias.clone().requires_grad_()
        gy = torch.randn(B, N, L, device=dev)
        y1 = ops.LinearAct.apply(v, wr, br, 1, 0.5, 77)
        y2 = ops.LinearAct.apply(vr, wr, br, 1, 0.5, 77)
        yt = torch.relu(torch.nn.functional.linear(torch.nn.functional.dropout(vr, 0.5, True), wr, br))
        fns = {"bwd dW+db (no dx)": lambda: torch.autograd.grad(y1, [wr, br], gy, retain_graph=True),
               "bwd dW+db+dx": lambda: torch.autograd.grad(y2, [vr, wr, br], gy, retain_graph=True),
               "torch bwd dW+db+dx": lambda: torch.autograd.grad(yt, [vr, wr, br], gy, retain_graph=True)}
        res = timeit(fns, args.rounds)
        report("K5 backward, dW only", {"bwd dW+db (no dx)": res["bwd dW+db (no dx)"]}, 2 * B * N * D * L, "TF")
        report("K5 backward, dW + dx", {k: res[k] for k in ("bwd dW+db+dx", "torch bwd dW+db+dx")}, 4 * B * N * D * L, "TF")
    if want("k2"):
        vl = torch.rand(B, N, L, device=dev)
        ql = torch.rand(B, L, device=dev)
        wk = torch.randn(G, N * L, device=dev) / (N * L) ** 0.5
        bk = torch.randn(G, device=dev) * 0.1
        fns = {"fwd p=0": lambda: ops.object_difference_attention(vl, ql, wk, bk, 0.0, 0),
               "fwd p=0.5": lambda: ops.object_difference_attention(vl, ql, wk, bk, 0.5, 7)}
        # VALU-bound: N*N*L mask elements per sample, 2 flop per (element, glimpse) + the difference
        report("K2 forward (ODA difference attention; 'TF' = 2*G*N*N*L*B flop)", timeit(fns, args.rounds), 2 * G * N * N * L * B, "TF")
        vr, qr, wr, br = (t.clone().requires_grad_() for t in (vl, ql, wk, bk))
        gl = torch.randn(B, N, G, device=dev)
        o0 = ops.object_difference_attention(vr, qr, wr, br, 0.0, 0)
        o5 = ops.object_difference_attention(vr, qr, wr, br, 0.5, 7)
        fns = {"bwd p=0": lambda: torch.autograd.grad(o0, [vr, qr, wr, br], gl, retain_graph=True),
               "bwd p=0.5": lambda: torch.autograd.grad(o5, [vr, qr, wr, br], gl, retain_graph=True)}
        report("K2 backward (data + weight passes)", timeit(fns, args.rounds), 2 * 2 * G * N * N * L * B, "TF")
        def torch_ref():
            vq = ((vl[:, :, None, :] - vl[:, None, :, :]) * ql[:, None, None, :]).reshape(B, N, N * L)
            return torch.nn.functional.linear(torch.nn.functional.dropout(vq, 0.5, True), wk, bk)
        report("reference: torch broadcast difference + dropout + linear (materialises [B,36,11160])", timeit({"torch fwd p=0.5": torch_ref}, args.rounds), 2 * G * N * N * L * B, "TF")
    if want("copy"):
        y = torch.empty_like(v)
        fns = {"torch copy 151MB": lambda: y.copy_(v)}
        report("reference: device copy (read+write)", timeit(fns, args.rounds), 2 * v.numel() * 4, "GB")
    if want("gemm"):
        w = torch.randn(L, D, device=dev)
        bias = torch.randn(L, device=dev)
        fns = {"F.linear 18432x2048x310": lambda: torch.nn.functional.linear(v, w, bias)}
        report("reference: hipBLASLt compress_v", timeit(fns, args.rounds), 2 * B * N * D * L, "TF")
        w2 = torch.randn(2 * H, L, device=dev)
        fns = {"F.linear 18432x310x1020": lambda: torch.nn.functional.linear(x, w2)}
        report("reference: hipBLASLt K4-shaped GEMM", timeit(fns, args.rounds), 2 * B * N * L * 2 * H, "TF")


if __name__ == "__main__":
    main()
