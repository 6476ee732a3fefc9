"""The split engine (csrc/gemm_f32_split.hpp, csrc/linear_split.hip; the default since round 5, VQA_F32_PRODUCTS=mfma selects the
fp32 MFMA engine): K5's forward and weight
gradient with every fp32 product formed on the bf16 matrix pipe from exact three-way bf16 splits of its operands.  Its claim is
"an fp32 GEMM, not a reduced-precision one": at BASELINE configs[1]'s size both engines are measured against a float64 matmul
and the split engine's error has to be of the fp32 MFMA engine's size -- measured on MI355X, rms error over the result's rms:
y 5.0e-7 vs 5.7e-7, d_w 5.4e-7 vs 4.4e-7 (4.8e-7 vs 3.1e-7 with the relu gate and dropout), d_b 1.9e-7 vs 2.3e-7; the largest
single errors are equal (y 9e-6 vs 8e-6 of the rms).  The two engines round the same contraction in different places (six
accumulations per 32-deep chunk against eight), which is all that separates them: a bf16 GEMM of the same operands is 4e-3
off.  The dropout masks and relu gates are the same functions of (seed, element) on both.  The whole-model comparisons against the
float64 oracle run in tests/test_gpu_models.py (variant split_products), at the fp32 engine's bars."""
import numpy as np
import pytest
import torch

from oracle import kernels_np as K
from oracle import seeded

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from vqa_playground_pytorch_amd import _lib, ops as o
    _lib.lib()
    return o


def err(got, ref):
    """(max, rms) error relative to the rms of the reference"""
    d = (got.double() - ref).abs()
    rms = ref.pow(2).mean().sqrt().item()
    return d.max().item() / rms, d.pow(2).mean().sqrt().item() / rms


def run(ops, monkeypatch, engine, x, w, b, gy, p, seed, act="relu"):
    monkeypatch.setenv("VQA_F32_PRODUCTS", engine)
    wt, bt = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = ops.linear_act(x, wt, bt, act, p, seed)
    y.backward(gy)
    return y.detach(), wt.grad.detach(), bt.grad.detach()


@pytest.mark.parametrize("p", [0.0, 0.5])
@pytest.mark.parametrize("wide", [False, True])
def test_split_engine_is_an_fp32_gemm(ops, monkeypatch, p, wide):
    """compress_v at B = 512: y [18432,310], d_w [310,2048], d_b on both engines against float64.  `wide` spreads the operands'
    magnitudes over 2^-12 .. 1 (the split's low planes then carry bits of very different weight)."""
    M, Kd, N, seed = 512 * 36, 2048, 310, 20260
    gen = torch.Generator(device="cpu").manual_seed(7 + wide)
    def draw(*shape, scale=1.0):
        t = torch.randn(*shape, generator=gen) * scale
        if wide:
            t = t * torch.exp2(-torch.randint(0, 13, shape, generator=gen).float())
        return t.to(dev())
    x, w, b, gy = draw(M, Kd), draw(N, Kd, scale=Kd ** -0.5), draw(N, scale=0.1), draw(M, N, scale=1.0 / 64)
    assert ops.split_products.__doc__ and ops._lib.lib().vqa_linear_split_supported(M, Kd, N, Kd, p) == 1
    y_m, dw_m, db_m = run(ops, monkeypatch, "mfma", x, w, b, gy, p, seed)
    y_s, dw_s, db_s = run(ops, monkeypatch, "split", x, w, b, gy, p, seed)
    mask = ops.linear_dropout_mask(M, Kd, p, seed, dev()).double() if p > 0 else None
    xm = x.double() * mask if p > 0 else x.double()
    pre = xm @ w.double().t() + b.double()
    ref_y = torch.relu(pre)
    # the relu gate of the weight gradient is each engine's own y > 0 (a pre-activation within rounding of zero may fall either
    # way): compare each engine's gradients against the float64 product gated by ITS gate
    for name, (y, dw, db) in (("mfma", (y_m, dw_m, db_m)), ("split", (y_s, dw_s, db_s))):
        gz = gy.double() * (y > 0)
        ref_dw, ref_db = gz.t() @ xm, gz.sum(0)
        e = {"y": err(y, ref_y), "d_w": err(dw, ref_dw), "d_b": err(db, ref_db)}
        print("[%s p=%.1f wide=%d] " % (name, p, wide) + "  ".join("%s max %.2e rms %.2e" % (k, *v) for k, v in e.items()))
        if name == "mfma":
            e_mfma = e
        for k, (mx, rm) in e.items():
            assert mx <= 2e-5 and rm <= 2e-6, (name, k, mx, rm)          # fp32 GEMM accuracy at K = 2048 / M = 18432
            if name == "split":
                assert rm <= 2.0 * e_mfma[k][1] + 1e-8, (k, rm, e_mfma[k][1])     # of the fp32 MFMA engine's size
                assert mx <= 2.0 * e_mfma[k][0] + 1e-7, (k, mx, e_mfma[k][0])
    # same masks, same gates: the engines agree far below the tolerance of any parity test
    assert err(y_s, y_m.double())[0] <= 2e-5          # (two fp32 results, each ~5e-6 of the rms from the float64 value at worst)
    assert ((y_s > 0) != (y_m > 0)).sum().item() <= 64    # of 5.7 M relu gates


@pytest.mark.parametrize("M,Kd,N,act,p", [(1152, 128, 16, "relu", 0.5), (2000, 256, 310, None, 0.0), (4609, 2048, 310, "relu", 0.5),
                                          (18432, 2048, 310, "relu", 0.0), (1300, 192, 38, "relu", 0.0)])
def test_split_engine_shapes_against_numpy(ops, monkeypatch, M, Kd, N, act, p):
    """ragged row counts, one-chunk-pair K, N off the 16-column blocks, K not a multiple of 128 (forward only on the split
    engine then; the weight gradient stays on the fp32 engine) -- against the numpy restatement with the written mask"""
    seed = 991
    x = seeded.seeded_array((M, Kd), 601)
    w = seeded.seeded_array((N, Kd), 602, scale=1.0 / np.sqrt(Kd))
    b = seeded.seeded_array((N,), 603, scale=0.1)
    gy = seeded.seeded_array((M, N), 604)
    monkeypatch.setenv("VQA_F32_PRODUCTS", "split")
    assert ops.split_products(M, Kd, N, Kd, p)
    mask = ops.linear_dropout_mask(M, Kd, p, seed, dev()).cpu().numpy() if p > 0 else None
    g = lambda a, rg=False: torch.from_numpy(a).to(dev()).requires_grad_(rg)
    xt, wt, bt = g(x), g(w, True), g(b, True)
    y = ops.linear_act(xt, wt, bt, act, p, seed)
    y_np = K.linear_act_fwd(x, w, b, act, mask)
    assert np.abs(y.detach().cpu().numpy() - y_np).max() <= 2e-5 * np.abs(y_np).max()
    y.backward(g(gy))
    _, dw, db = K.linear_act_bwd(x, w, y.detach().cpu().numpy(), gy, act, mask)
    assert np.abs(wt.grad.cpu().numpy() - dw).max() <= 2e-5 * np.abs(dw).max()
    assert np.abs(bt.grad.cpu().numpy() - db).max() <= 2e-5 * np.abs(db).max()


def test_split_engine_refuses_what_it_cannot_run(ops):
    L = ops._lib.lib()
    assert L.vqa_linear_split_supported(18432, 2048, 310, 2048, 0.5) == 1
    assert L.vqa_linear_split_supported(18432, 2048, 310, 2048, 0.3) == 0      # one-bit masks only
    assert L.vqa_linear_split_supported(18432, 2000, 310, 2000, 0.0) == 0      # K % 64
    assert L.vqa_linear_split_supported(144, 2048, 310, 2048, 0.0) == 0        # short matrices stay on the LDS tile engine
    x = torch.zeros(18432, 2000, device=dev())
    y = torch.empty(18432, 310, device=dev())
    ws = torch.empty(1 << 20, device=dev())
    rc = L.vqa_linear_act_fwd_split(x.data_ptr(), 2000, x.data_ptr(), None, y.data_ptr(), ws.data_ptr(), ws.numel() * 4, 18432, 2000,
                                    310, 1, 0.0, 0, None, None)
    assert rc != 0


@pytest.mark.parametrize("p", [0.0, 0.5])
def test_split_relation_projection_gradients_at_size(ops, monkeypatch, p):
    """The fused relation + projection node (K1 -> K5) at B = 512 on both engines against float64: d_t, d_c2 (the data gradient
    reduced in the GEMM tile, csrc/relation_dgrad{,_split}.hip), d_w, d_b.  Each engine's gradients are compared with the float64
    closed form gated by ITS OWN forward's relu gates; the split engine's error has to be of the fp32 MFMA engine's size."""
    B, N, D, L, seed = 512, 36, 2048, 310, 4711
    gen = torch.Generator(device="cpu").manual_seed(11)
    v = torch.randn(B, N, D, generator=gen).to(dev())
    t = torch.randn(B, D, generator=gen).to(dev())
    c2 = torch.sigmoid(torch.randn(B, D, generator=gen)).to(dev())
    w = (torch.randn(L, D, generator=gen) / D ** 0.5).to(dev())
    b = (0.1 * torch.randn(L, generator=gen)).to(dev())
    gy = (torch.randn(B, N, L, generator=gen) / 64).to(dev())
    mask = ops.linear_dropout_mask(B * N, D, p, seed, dev()).view(B, N, D).double() if p > 0 else None
    x64 = t.double()[:, None, :] + c2.double()[:, None, :] * v.double()
    if mask is not None:
        x64 = x64 * mask
    errs = {}
    for engine in ("mfma", "split"):
        monkeypatch.setenv("VQA_F32_PRODUCTS", engine)
        tt, ct, wt, bt = (z.clone().requires_grad_(True) for z in (t, c2, w, b))
        y = ops.relation_projection(v, tt, ct, wt, bt, p, seed)
        y.backward(gy)
        gz = gy.double() * (y.detach() > 0)
        dx = gz.reshape(B * N, L) @ w.double()
        dx = dx.view(B, N, D) * (mask if mask is not None else 1.0)
        ref = {"d_t": dx.sum(1), "d_c2": (dx * v.double()).sum(1), "d_w": gz.reshape(B * N, L).t() @ x64.reshape(B * N, D),
               "d_b": gz.sum((0, 1))}
        got = {"d_t": tt.grad, "d_c2": ct.grad, "d_w": wt.grad, "d_b": bt.grad}
        errs[engine] = {k: err(got[k], ref[k]) for k in ref}
        print("[%s p=%.1f] " % (engine, p) + "  ".join("%s max %.2e rms %.2e" % (k, *e) for k, e in errs[engine].items()))
        for k, (mx, rm) in errs[engine].items():
            assert mx <= 2e-5 and rm <= 2e-6, (engine, k, mx, rm)
    for k in errs["split"]:
        assert errs["split"][k][1] <= 2.0 * errs["mfma"][k][1] + 1e-8, (k, errs["split"][k], errs["mfma"][k])
        assert errs["split"][k][0] <= 2.0 * errs["mfma"][k][0] + 1e-7, (k, errs["split"][k], errs["mfma"][k])


# ---- the engine's domain: all of fp32 ---------------------------------------------------------------
# A three-way bf16 split is exact for finite values whose leading plane is finite; +-Inf, NaN and values within half a bf16
# ulp of FLT_MAX are outside of it (plane 0 = Inf, plane 1 = Inf - Inf = NaN).  The kernels detect what such an operand does
# to their accumulators (non-finite) and recompute those outputs as fp32 dot products of the original operands
# (gemm_f32_split.hpp, any_nonfinite).  Checked here against the fp32 MFMA engine and float64, through the C ABI.

EDGE_M, EDGE_K, EDGE_N = 4608, 256, 310      # (M >= 4096: both engines run their tall-matrix kernels, forward and weight gradient)


def _classes(t):
    """0 finite, 1 +Inf, 2 -Inf, 3 NaN"""
    return torch.where(torch.isnan(t), 3, torch.where(torch.isposinf(t), 1, torch.where(torch.isneginf(t), 2, 0)))


def _edge_operands(kind, gen):
    M, Kd, N = EDGE_M, EDGE_K, EDGE_N
    x = torch.randn(M, Kd, generator=gen)
    w = torch.randn(N, Kd, generator=gen) / Kd ** 0.5
    gy = torch.randn(M, N, generator=gen) / 64
    fmax = torch.finfo(torch.float32).max
    if kind == "flt_max":
        # values whose bf16 rounding is Inf (> 3.3961e38) next to partners small enough that every sum stays finite in fp32
        for i, (m, k) in enumerate([(0, 0), (5, 17), (143, 255), (144, 31), (1000, 100), (4607, 128), (4607, 129)]):
            x[m, k] = fmax * (1.0 if i % 2 == 0 else -1.0) * (1.0 - 2.0 ** -(9 + i))
            w[:, k] *= 2.0 ** -30
            gy[m, :] *= 2.0 ** -60                 # (the weight gradient multiplies x[m, k] with gy[m, :]; 4608 rows are added up)
        for (n, k) in [(3, 40), (309, 200)]:
            w[n, k] = -fmax * (1.0 - 2.0 ** -12)
            x[:, k] *= 2.0 ** -40
        for (m, n) in [(7, 7), (2000, 300)]:
            gy[m, n] = fmax * (1.0 - 2.0 ** -10)
            x[m, :] *= 2.0 ** -40
    elif kind == "nonfinite":
        inf, nan = float("inf"), float("nan")
        x[1, 3], x[150, 200], x[151, 200], x[2000, 0], x[4607, 255] = inf, -inf, nan, inf, nan
        x[700, 10], x[700, 11] = inf, -inf          # Inf - Inf in one row
        w[5, 9], w[300, 100], w[17, 255] = inf, nan, -inf
        w[40, 3] = 0.0                               # x[1, 3] = Inf meets an exact zero: NaN on any engine
        gy[9, 9], gy[1500, 309], gy[1501, 0], gy[4607, 150] = inf, nan, -inf, inf
    elif kind == "tiny":
        x = torch.exp2(-126.0 + 26.0 * torch.rand(M, Kd, generator=gen)) * torch.sign(torch.randn(M, Kd, generator=gen))
        w = torch.randn(N, Kd, generator=gen)
        gy = torch.randn(M, N, generator=gen)
    elif kind == "spread":
        x = x * torch.exp2(torch.randint(-40, 41, x.shape, generator=gen).float())
        w = w * torch.exp2(torch.randint(-40, 41, w.shape, generator=gen).float())
        gy = gy * torch.exp2(torch.randint(-40, 41, gy.shape, generator=gen).float())
    else:
        raise ValueError(kind)
    return x.to(dev()), w.to(dev()), gy.to(dev())


def _compare_with_engine(name, got_s, got_m, ref64, scale64, finite_everywhere, err_factor=2.0):
    """got_s / got_m: the split / fp32 MFMA engine's result; ref64: float64; scale64: sum_k |a_k| |b_k| (what fp32 rounding errors
    of a dot product are proportional to).  Non-finite outputs: the same class (+Inf / -Inf / NaN) at the same places on both
    engines.  Finite outputs: the split engine's error in units of scale is within err_factor of the fp32 MFMA engine's."""
    cs, cm = _classes(got_s), _classes(got_m)
    assert torch.equal(cs, cm), "%s: %d outputs differ in class (finite / +Inf / -Inf / NaN) between the engines; first at %s: split %s, mfma %s" % (
        name, int((cs != cm).sum()), (cs != cm).nonzero()[0].tolist(), got_s[tuple((cs != cm).nonzero()[0])].item(),
        got_m[tuple((cs != cm).nonzero()[0])].item())
    fin = (cm == 0) & torch.isfinite(ref64) & torch.isfinite(scale64)
    if finite_everywhere:
        assert bool((cm == 0).all()), "%s: the fp32 MFMA engine itself is not finite here -- the case is mis-built" % name
    unit = scale64.clamp_min(1e-300)
    es = ((got_s.double() - ref64).abs() / unit)[fin]
    em = ((got_m.double() - ref64).abs() / unit)[fin]
    rs, rm = es.pow(2).mean().sqrt().item(), em.pow(2).mean().sqrt().item()
    print("[%s] finite outputs %d of %d; error / sum|a||b|: split max %.2e rms %.2e, mfma max %.2e rms %.2e"
          % (name, int(fin.sum()), fin.numel(), es.max().item(), rs, em.max().item(), rm))
    assert es.max().item() <= 4e-6, (name, es.max().item())          # an fp32 dot product of this length: ~1e-7 sqrt(K) at worst
    assert rs <= err_factor * rm + 1e-9, (name, rs, rm)
    return int((cm != 0).sum())


@pytest.mark.parametrize("p", [0.0, 0.5])
@pytest.mark.parametrize("kind", ["flt_max", "nonfinite", "tiny", "spread"])
def test_split_engine_edge_values(ops, monkeypatch, kind, p):
    """vqa_linear_act_fwd_split / vqa_linear_act_dw_split on operands at the edges of fp32 (act = none, so that every class of
    value reaches the output): next to FLT_MAX, +-Inf / NaN, 2^-126 .. 2^-100, exponents spread over 2^-40 .. 2^40."""
    M, Kd, N, seed = EDGE_M, EDGE_K, EDGE_N, 77
    gen = torch.Generator(device="cpu").manual_seed(1234)
    x, w, gy = _edge_operands(kind, gen)
    b = torch.zeros(N, device=dev())
    assert ops._lib.lib().vqa_linear_split_supported(M, Kd, N, Kd, p) == 1
    y_m, dw_m, _ = run(ops, monkeypatch, "mfma", x, w, b, gy, p, seed, act=None)
    y_s, dw_s, _ = run(ops, monkeypatch, "split", x, w, b, gy, p, seed, act=None)
    mask = ops.linear_dropout_mask(M, Kd, p, seed, dev()).double() if p > 0 else torch.ones(M, Kd, device=dev(), dtype=torch.float64)
    # a dropped element is an exact zero on both engines (a select, not a multiply): Inf * 0 never happens through the mask
    xm = torch.where(mask > 0, x.double() * mask, torch.zeros_like(mask))
    ref_y, sc_y = xm @ w.double().t(), xm.abs() @ w.double().abs().t()
    ref_dw, sc_dw = gy.double().t() @ xm, gy.double().abs().t() @ xm.abs()
    finite = kind != "nonfinite"
    n_y = _compare_with_engine("%s p=%.1f y" % (kind, p), y_s, y_m, ref_y, sc_y, finite)
    n_dw = _compare_with_engine("%s p=%.1f d_w" % (kind, p), dw_s, dw_m, ref_dw, sc_dw, finite)
    if kind == "nonfinite":
        assert n_y > 0 and n_dw > 0                  # the case does reach the outputs
        ok = torch.isfinite(ref_y)
        assert torch.equal(_classes(y_s)[~ok], _classes(ref_y.float())[~ok])     # and float64 agrees on which class
    if kind == "flt_max":
        assert float(y_s.abs().max()) > 1e7 and float(dw_s.abs().max()) > 1e-30   # the large operands did take part


@pytest.mark.parametrize("kind", ["flt_max", "nonfinite"])
def test_split_engine_edge_values_with_relu_gate(ops, monkeypatch, kind):
    """the same with the relu epilogue and its gate in the weight gradient (the shape of compress_v): engine against engine"""
    M, Kd, N, seed = EDGE_M, EDGE_K, EDGE_N, 78
    gen = torch.Generator(device="cpu").manual_seed(4321)
    x, w, gy = _edge_operands(kind, gen)
    b = (0.1 * torch.randn(N, generator=gen)).to(dev())
    y_m, dw_m, db_m = run(ops, monkeypatch, "mfma", x, w, b, gy, 0.5, seed)
    y_s, dw_s, db_s = run(ops, monkeypatch, "split", x, w, b, gy, 0.5, seed)
    for name, a, c in (("y", y_s, y_m), ("d_w", dw_s, dw_m), ("d_b", db_s, db_m)):
        assert torch.equal(_classes(a), _classes(c)), name
        fin = torch.isfinite(c)
        if name == "d_w":       # gates of pre-activations within rounding of zero may differ between the engines: rows apart
            same_gate = ((y_s > 0) == (y_m > 0)).all(0)
            fin = fin & same_gate[:, None]
        d = (a.double() - c.double()).abs()[fin]
        scale = c.double().abs()[fin].clamp_min(1e-30)
        big = c.double().abs()[fin].max().item()
        assert (d <= 1e-4 * scale + 1e-5 * big).all(), (name, (d / scale).max().item())


@pytest.mark.parametrize("kind", ["flt_max", "nonfinite"])
def test_split_relation_dgrad_edge_values(ops, monkeypatch, kind):
    """vqa_relation_projection_dgrad_split with gz values the split cannot represent -- at the END of a row's neighbour too: the
    contraction is padded from L = 310 to 320, so a row's loads pick up the next row's first ten values against zero planes
    (0 * Inf = NaN without the repair path) -- against the fp32 MFMA kernel."""
    B, N, D, L, seed = 64, 36, 256, 310, 5
    gen = torch.Generator(device="cpu").manual_seed(99)
    v = torch.randn(B, N, D, generator=gen).to(dev())
    t = torch.randn(B, D, generator=gen).to(dev())
    c2 = torch.sigmoid(torch.randn(B, D, generator=gen)).to(dev())
    w = (torch.randn(L, D, generator=gen) / D ** 0.5)
    b = (1.0 + 0.1 * torch.randn(L, generator=gen)).to(dev())      # (most units open: the upstream gradient gets through the gate)
    gy = (torch.randn(B, N, L, generator=gen) / 64)
    fmax = torch.finfo(torch.float32).max
    special = [(0, 1, 0), (3, 35, 5), (10, 0, 309), (63, 35, 9), (40, 17, 3)]    # (sample, region, unit); units < 10 sit in the padding
    for i, (bb, n, l) in enumerate(special):                                       # of the row before
        if kind == "flt_max":
            gy[bb, n, l] = fmax * (1.0 - 2.0 ** -(9 + i)) * (-1.0) ** i
            w[l, :] *= 2.0 ** -40
        else:
            gy[bb, n, l] = [float("inf"), float("-inf"), float("nan")][i % 3]
    w, gy = w.to(dev()), gy.to(dev())
    out = {}
    for engine in ("mfma", "split"):
        monkeypatch.setenv("VQA_F32_PRODUCTS", engine)
        tt, ct, wt, bt = (z.clone().requires_grad_(True) for z in (t, c2, w, b))
        y = ops.relation_projection(v, tt, ct, wt, bt, 0.5, seed)
        y.backward(gy)
        out[engine] = (y.detach(), tt.grad, ct.grad, wt.grad)
    same_gates = ((out["split"][0] > 0) == (out["mfma"][0] > 0)).all()
    for name, a, c in zip(("y", "d_t", "d_c2", "d_w"), out["split"], out["mfma"]):
        assert torch.equal(_classes(a), _classes(c)), "%s: classes differ at %d places" % (name, int((_classes(a) != _classes(c)).sum()))
        fin = torch.isfinite(c)
        d = (a.double() - c.double()).abs()[fin]
        big = c.double().abs()[fin].max().item()
        if bool(same_gates):
            assert d.max().item() <= 2e-5 * big, (name, d.max().item(), big)
    if kind == "nonfinite":
        assert int((~torch.isfinite(out["mfma"][1])).sum()) > 0
        # only the samples that own a special value are touched: the padding leaks nothing into the neighbours
        touched = sorted({bb for bb, _, _ in special})
        bad_rows = (~torch.isfinite(out["split"][1])).any(1).nonzero().flatten().tolist()
        assert set(bad_rows) <= set(touched), (bad_rows, touched)
