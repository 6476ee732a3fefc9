"""GPU parity of the grouped head (K6: csrc/grouped_gemm.hip + vqa_playground_pytorch_amd/head.py) -- the [B,.]-sized
layers of CoR2 / ODA as phases of one grouped GEMM launch + one grouped epilogue launch -- against float64 matmuls and
against the same layers written with plain torch ops (MyLinear / putils.Linear / MutanFusion semantics: config/CoR2.py:
94-122, putils/__init__.py:16-33,232-238).  The whole-model comparisons with the reference's goldens and the float64
restatement (tests/test_gpu_models.py) run through these phases as well."""
import numpy as np
import pytest
import torch

from oracle import seeded

pytestmark = pytest.mark.gpu
RTOL = 2e-4      # fp32 MFMA products against float64, on the output's scale


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def g(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    return t.requires_grad_() if grad else t


def close(name, got, want, rtol=RTOL):
    got = got.detach().double().cpu().numpy()
    want = np.asarray(want.detach().double().cpu().numpy() if isinstance(want, torch.Tensor) else want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert np.isfinite(got).all(), name
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-20)
    assert err <= rtol, "%s: rel err %.3e" % (name, err)


@pytest.fixture(scope="module", params=["split", "mfma"])
def head(request):
    """The module with the phases' GEMM launch pinned to one engine: csrc/grouped_gemm_split.hip (the default with the split
    products) or csrc/grouped_gemm.hip (fp32 MFMA)."""
    from vqa_playground_pytorch_amd import head as h
    before = h.Phase.ENGINE
    h.Phase.ENGINE = request.param
    yield h
    h.Phase.ENGINE = before


@pytest.mark.parametrize("M,N,K", [(512, 310, 2400), (3, 155, 2048), (64, 64, 16), (130, 2000, 510), (1, 310, 310), (77, 620, 1240)])
def test_grouped_gemm_all_forms_against_float64(head, M, N, K):
    """One launch holding an NT, an NN and a TN product (and the 4-byte-aligned-A variants), each split over its contraction:
    sum of the slabs == the float64 product; the TN problems' column sums == the bias gradient."""
    x = seeded.seeded_array((M, K), 501)
    w = seeded.seeded_array((N, K), 502) / np.sqrt(K)
    dy = seeded.seeded_array((M, N), 503)
    xt, wt, dyt = g(x), g(w), g(dy)
    ph = head.Phase(dev(), "test")
    t_nt = ph.target(M, N)
    ph.gemm(t_nt, head.NT, xt, K, wt, K, K)
    y = torch.empty(M, N, device=dev())
    ph.job(head.EPI_SUM, t_nt, y, N)
    outs = {}
    if N % 2 == 0:
        t_nn = ph.target(M, K)
        ph.gemm(t_nn, head.NN, dyt, N, wt, K, N)
        outs["dx"] = torch.empty(M, K, device=dev())
        ph.job(head.EPI_SUM, t_nn, outs["dx"], K)
        t_tn = ph.target(N, K)
        ph.gemm(t_tn, head.TN, dyt, N, xt, K, M, colsum=True)
        outs["dw"], outs["db"] = torch.empty(N, K, device=dev()), torch.empty(N, device=dev())
        ph.job(head.EPI_SUM, t_tn, outs["dw"], K)
        ph.job(head.EPI_SUM, t_tn, outs["db"], N, colsum=True)
    t_nn4 = ph.target(M, K)
    ph.gemm(t_nn4, head.NN_A4, dyt, N, wt, K, N)
    outs["dx4"] = torch.empty(M, K, device=dev())
    ph.job(head.EPI_SUM, t_nn4, outs["dx4"], K)
    t_tn4 = ph.target(N, K)
    ph.gemm(t_tn4, head.TN_A4, dyt, N, xt, K, M, colsum=True)
    outs["dw4"], outs["db4"] = torch.empty(N, K, device=dev()), torch.empty(N, device=dev())
    ph.job(head.EPI_SUM, t_tn4, outs["dw4"], K)
    ph.job(head.EPI_SUM, t_tn4, outs["db4"], N, colsum=True)
    ph.run()
    x64, w64, dy64 = x.astype(np.float64), w.astype(np.float64), dy.astype(np.float64)
    close("y", y, x64 @ w64.T)
    for k, want in (("dx", dy64 @ w64), ("dw", dy64.T @ x64), ("db", dy64.sum(0)), ("dx4", dy64 @ w64), ("dw4", dy64.T @ x64),
                    ("db4", dy64.sum(0))):
        if k in outs:
            close(k, outs[k], want)


def test_grouped_gemm_shared_target_and_offsets(head):
    """Two problems adding into ONE output (a K-concatenated product whose operands live in different tensors) and operands
    addressed at element offsets inside larger tensors (a glimpse block of [B,G,D], a column block of W)."""
    B, Gn, D, A = 37, 4, 256, 155
    pooled = seeded.seeded_array((B, Gn, D), 511)
    w = seeded.seeded_array((A, D), 512) / np.sqrt(D)
    x1, x2 = seeded.seeded_array((B, 100), 513), seeded.seeded_array((B, 60), 514)
    wf = seeded.seeded_array((90, 160), 515)
    pt, wt, x1t, x2t, wft = g(pooled), g(w), g(x1), g(x2), g(wf)
    ph = head.Phase(dev(), "test")
    t1 = ph.target(B, A)
    ph.gemm(t1, head.NT, pt, Gn * D, wt, D, D, a_off=2 * D)                 # glimpse 2 of the pooled tensor
    y1 = torch.zeros(B, 4 * A, device=dev())
    ph.job(head.EPI_SUM, t1, y1, 4 * A, out_off=A)                         # written at column block 1 of a wider output
    t2 = ph.target(B, 90)
    ph.gemm(t2, head.NT, x1t, 100, wft, 160, 100)                           # [x1 | x2] @ wf^T without the concatenation
    ph.gemm(t2, head.NT, x2t, 60, wft, 160, 60, b_off=100)
    y2 = torch.empty(B, 90, device=dev())
    ph.job(head.EPI_SUM, t2, y2, 90)
    ph.run()
    close("glimpse block", y1[:, A:2 * A], pooled[:, 2].astype(np.float64) @ w.astype(np.float64).T)
    assert float(y1[:, :A].abs().max()) == 0.0 and float(y1[:, 2 * A:].abs().max()) == 0.0
    close("concatenated", y2, np.concatenate([x1, x2], 1).astype(np.float64) @ wf.astype(np.float64).T)


def _mask(ops, rows, cols, p, seed):
    return ops.linear_dropout_mask(rows, cols, p, seed, dev())


@pytest.mark.parametrize("B", [5, 130])
@pytest.mark.parametrize("train", [False, True])
def test_head_phases_match_plain_layers(head, B, train):
    """The six phases chained as CoR2 chains them, against the same layers written with torch ops and fed the SAME dropout
    masks (exported through the head's mask spy): outputs and every gradient, 1e-3 of each tensor's scale."""
    from vqa_playground_pytorch_amd import ops
    Q, A, D, H, R, Gn, GA, C = 96, 62, 128, 102, 2, 4, 31, 50
    p = 0.5 if train else 0.0
    rng = np.random.RandomState(7)
    P = {}

    def par(name, *shape, scale=None):
        a = rng.standard_normal(shape).astype(np.float32) * (scale if scale is not None else 1.0 / np.sqrt(shape[-1]))
        P[name] = g(a, True)
        return P[name]

    wq = [par("wq%d" % i, A, Q) for i in range(4)]
    bq = [par("bq%d" % i, A, scale=0.1) for i in range(4)]
    we = [par("we%d" % i, D, A) for i in range(2)]
    be = [par("be%d" % i, D, scale=0.1) for i in range(2)]
    w2 = [[par("w2_%d_%d" % (f, r), H, A) for r in range(R)] for f in range(3)]
    b2 = [[par("b2_%d_%d" % (f, r), H, scale=0.1) for r in range(R)] for f in range(3)]
    wg = [[par("wg%d_%d" % (a_, i), GA, D) for i in range(Gn)] for a_ in range(2)]
    bg = [[par("bg%d_%d" % (a_, i), GA, scale=0.1) for i in range(Gn)] for a_ in range(2)]
    w1 = [par("w1f_%d" % r, H, 2 * Gn * GA) for r in range(R)]
    b1 = [par("b1f_%d" % r, H, scale=0.1) for r in range(R)]
    wc, bc = par("wc", C, H), par("bc", C, scale=0.1)
    q = g(rng.standard_normal((B, Q)))
    pooled = [g(rng.standard_normal((B, Gn, D)), True) for _ in range(2)]
    g_q1, g_q2 = g(rng.standard_normal((B, D))), g(rng.standard_normal((B, D)))
    g_h2 = [g(rng.standard_normal((B, R, H))) for _ in range(2)]
    g_logits = g(rng.standard_normal((B, C)))

    masks = {}
    head._mask_spy = lambda site, rows, cols, p_, seed: masks.__setitem__(site, _mask(ops, rows, cols, p_, seed))
    try:
        lows = head.QuestionProjections.apply(q, p, 11, (2, 3), p, 12, (), *wq, *bq)
        s = 1.0 / (1.0 - p) if p else 1.0
        flat = we[0], be[0], we[1], be[1]
        for f in range(3):
            flat += tuple(w2[f]) + tuple(b2[f])
        q1, q2, h2a, h2b, h2f = head.GatesAndRankFactors.apply(4, (2, 3), ((0, R), (0, R), (1, R)), (1.0, 1.0, s, s), *lows, *flat)
        v_att = [head.GlimpseProjections.apply(pooled[a_], *wg[a_], *bg[a_]) for a_ in range(2)]
        x = head.VectorFusion.apply(h2f, p, 13, 2, v_att[0], v_att[1], *w1, *b1)
        logits = head.Classifier.apply(x, wc, bc, p, 13)
    finally:
        head._mask_spy = None
    loss = (logits * g_logits).sum() + (q1 * g_q1).sum() + (q2 * g_q2).sum() + (h2a * g_h2[0]).sum() + (h2b * g_h2[1]).sum()
    loss.backward()
    got = {k: v.grad.clone() for k, v in P.items()}
    got_pooled = [t.grad.clone() for t in pooled]
    for v in list(P.values()) + pooled:
        v.grad = None

    # the same computation with plain torch ops (float64) and the exported masks
    d = lambda t: t.detach().double().requires_grad_(t.requires_grad)  # noqa: E731
    P64 = {k: d(v) for k, v in P.items()}
    q64 = q.double()
    m_in = masks["question_in"].double().view(4, B, Q) if p else torch.ones(4, B, Q, device=dev(), dtype=torch.float64)
    m_g = masks["question_out"].double().view(2, B, A) if p else torch.ones(2, B, A, device=dev(), dtype=torch.float64)
    m_x = masks["fusion_out"].double() if p else torch.ones(B, H, device=dev(), dtype=torch.float64)
    lows = [torch.relu((q64 * m_in[i]) @ P64["wq%d" % i].t() + P64["bq%d" % i]) for i in range(4)]
    gates = [torch.sigmoid((lows[2 + i] * m_g[i]) @ P64["we%d" % i].t() + P64["be%d" % i]) for i in range(2)]
    h2s = [torch.stack([lows[0 if f < 2 else 1] @ P64["w2_%d_%d" % (f, r)].t() + P64["b2_%d_%d" % (f, r)] for r in range(R)], 1)
           for f in range(3)]
    pooled64 = [d(t) for t in pooled]
    v64 = [torch.cat([torch.relu(pooled64[a_][:, i] @ P64["wg%d_%d" % (a_, i)].t() + P64["bg%d_%d" % (a_, i)]) for i in range(Gn)], 1)
           for a_ in range(2)]
    vf = torch.cat(v64, 1)
    x64 = sum((vf @ P64["w1f_%d" % r].t() + P64["b1f_%d" % r]) * h2s[2][:, r] for r in range(R))
    logits64 = (x64 * m_x) @ P64["wc"].t() + P64["bc"]
    loss64 = (logits64 * g_logits.double()).sum() + (gates[0] * g_q1.double()).sum() + (gates[1] * g_q2.double()).sum() + \
        (h2s[0] * g_h2[0].double()).sum() + (h2s[1] * g_h2[1].double()).sum()
    loss64.backward()
    close("logits", logits, logits64, 1e-3)
    close("q1", q1, gates[0], 1e-3)
    close("h2 (fusion 2)", h2b, h2s[1], 1e-3)
    for k in P:
        close("d " + k, got[k], P64[k].grad, 1e-3)
    for a_ in range(2):
        close("d pooled%d" % a_, got_pooled[a_], pooled64[a_].grad, 1e-3)
    if p:
        for site, m in masks.items():
            assert set(torch.unique(m).tolist()) == {0.0, 2.0}, site


def _classes(t):
    """0 finite, 1 +Inf, 2 -Inf, 3 NaN"""
    return torch.where(torch.isnan(t), 3, torch.where(torch.isposinf(t), 1, torch.where(torch.isneginf(t), 2, 0)))


@pytest.mark.parametrize("kind", ["flt_max", "nonfinite", "spread"])
def test_grouped_split_engine_edge_values(kind):
    """csrc/grouped_gemm_split.hip at the edges of fp32, all three forms in one launch, against csrc/grouped_gemm.hip (fp32
    MFMA) and float64: operands next to FLT_MAX (their leading bf16 plane rounds to Inf), +-Inf / NaN, exponents spread over
    2^-40 .. 2^40.  Same class (finite / +Inf / -Inf / NaN) at every output on both engines -- the split kernel recomputes a
    non-finite accumulator as an fp32 dot product of the original operands -- and, where finite, an error on the scale of
    sum |a| |b| within 2 x the fp32 MFMA kernel's."""
    from vqa_playground_pytorch_amd import head as h
    M, K, N = 256, 384, 310
    gen = torch.Generator(device="cpu").manual_seed(2718)
    x = torch.randn(M, K, generator=gen)
    w = torch.randn(N, K, generator=gen) / K ** 0.5
    gy = torch.randn(M, N, generator=gen) / 16
    fmax = torch.finfo(torch.float32).max
    if kind == "flt_max":
        for i, (m, k) in enumerate([(0, 0), (5, 17), (127, 383), (128, 31), (255, 200)]):
            x[m, k] = fmax * (1.0 if i % 2 == 0 else -1.0) * (1.0 - 2.0 ** -(9 + i))
            w[:, k] *= 2.0 ** -30
            gy[m, :] *= 2.0 ** -60
        w[3, 40] = -fmax * (1.0 - 2.0 ** -12)
        x[:, 40] *= 2.0 ** -40
        gy[7, 7] = fmax * (1.0 - 2.0 ** -10)
        x[7, :] *= 2.0 ** -40
        w[7, :] *= 2.0 ** -40
    elif kind == "nonfinite":
        inf, nan = float("inf"), float("nan")
        x[1, 3], x[150, 200], x[151, 200], x[255, 383] = inf, -inf, nan, inf
        x[70, 10], x[70, 11] = inf, -inf
        w[5, 9], w[300, 100], w[17, 383] = inf, nan, -inf
        w[40, 3] = 0.0
        gy[9, 9], gy[100, 309], gy[101, 0] = inf, nan, -inf
    else:
        x = x * torch.exp2(torch.randint(-40, 41, x.shape, generator=gen).float())
        w = w * torch.exp2(torch.randint(-40, 41, w.shape, generator=gen).float())
        gy = gy * torch.exp2(torch.randint(-40, 41, gy.shape, generator=gen).float())
    xt, wt, gt = x.to(dev()), w.to(dev()), gy.to(dev())

    def run(engine):
        before = h.Phase.ENGINE
        h.Phase.ENGINE = engine
        try:
            ph = h.Phase(dev(), "edge")
            outs = {"y": torch.empty(M, N, device=dev()), "dx": torch.empty(M, K, device=dev()),
                    "dw": torch.empty(N, K, device=dev()), "db": torch.empty(N, device=dev())}
            t1 = ph.target(M, N)
            ph.gemm(t1, h.NT, xt, K, wt, K, K)
            ph.job(h.EPI_SUM, t1, outs["y"], N)
            t2 = ph.target(M, K)
            ph.gemm(t2, h.NN, gt, N, wt, K, N)
            ph.job(h.EPI_SUM, t2, outs["dx"], K)
            t3 = ph.target(N, K)
            ph.gemm(t3, h.TN, gt, N, xt, K, M, colsum=True)
            ph.job(h.EPI_SUM, t3, outs["dw"], K)
            ph.job(h.EPI_SUM, t3, outs["db"], N, colsum=True)
            ph.run()
            torch.cuda.synchronize()
            return outs
        finally:
            h.Phase.ENGINE = before

    got_s, got_m = run("split"), run("mfma")
    x64, w64, g64 = xt.double(), wt.double(), gt.double()
    refs = {"y": (x64 @ w64.t(), x64.abs() @ w64.abs().t()), "dx": (g64 @ w64, g64.abs() @ w64.abs()),
            "dw": (g64.t() @ x64, g64.abs().t() @ x64.abs()), "db": (g64.sum(0), g64.abs().sum(0))}
    touched = 0
    for name, (ref, scale) in refs.items():
        cs, cm = _classes(got_s[name]), _classes(got_m[name])
        assert torch.equal(cs, cm), "%s: %d outputs differ in class between the engines" % (name, int((cs != cm).sum()))
        touched += int((cm != 0).sum())
        fin = (cm == 0) & torch.isfinite(ref) & torch.isfinite(scale)
        if kind != "nonfinite":
            assert bool((cm == 0).all()), "%s: the fp32 MFMA kernel itself is not finite here -- the case is mis-built" % name
        unit = scale.clamp_min(1e-300)
        es = ((got_s[name].double() - ref).abs() / unit)[fin]
        em = ((got_m[name].double() - ref).abs() / unit)[fin]
        rs, rm = es.pow(2).mean().sqrt().item(), em.pow(2).mean().sqrt().item()
        assert es.max().item() <= 4e-6, (name, es.max().item())
        assert rs <= 2.0 * rm + 1e-9, (name, rs, rm)
    if kind == "nonfinite":
        assert touched > 0
