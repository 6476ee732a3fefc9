import torch
dev = torch.device("cuda:0")
cases = [((512, 1020), (1020, 310), (1020, 512), (512, 310)), ((512, 1020), (1020, 1240), (1020, 512), (512, 1240)),
         ((512, 2000), (2000, 510), (2000, 512), (512, 510))]
side = torch.cuda.Stream()
for sa, sb, sc, sd in cases:
    a, b, c, d = (torch.randn(*s, device=dev) for s in (sa, sb, sc, sd))
    o1 = torch.empty(sa[0], sb[1], device=dev); o2 = torch.empty(sc[0], sd[1], device=dev)
    def seq():
        torch.mm(a, b, out=o1); torch.mm(c, d, out=o2)
    def par():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            torch.mm(c, d, out=o2)
        torch.mm(a, b, out=o1)
        cur.wait_stream(side)
    for name, fn in (("seq", seq), ("par", par)):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): fn()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): fn()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        print(sa, sb, name, "%.1f us per pair" % (e0.elapsed_time(e1) * 1e3 / 200))
