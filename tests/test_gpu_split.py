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


# ---- K1 -> K5 in one kernel (round 6: VERDICT r05 missing #2; SURVEY 8f row 1, second half) -------------------------------------
def _relation_case(B, N, D, L, gen, scale_t=1.0):
    v = torch.randn(B, N, D, generator=gen).to(dev())
    t = (scale_t * torch.randn(B, D, generator=gen)).to(dev())
    c2 = torch.sigmoid(torch.randn(B, D, generator=gen)).to(dev())
    w = (torch.randn(L, D, generator=gen) / D ** 0.5).to(dev())
    b = (0.1 * torch.randn(L, generator=gen)).to(dev())
    gy = (torch.randn(B, N, L, generator=gen) / 64).to(dev())
    return v, t, c2, w, b, gy


def _relation_run(ops, monkeypatch, fused, v, t, c2, w, b, gy, p, seed, pregated=False):
    monkeypatch.setenv("VQA_F32_PRODUCTS", "split")
    monkeypatch.setenv("VQA_RELATION_FUSED", "1" if fused else "0")
    tt, ct, wt, bt = (z.clone().requires_grad_(True) for z in (t, c2, w, b))
    seen = []
    inner = ops._launch
    monkeypatch.setattr(ops, "_launch", lambda name, *a, **k: (seen.append(name), inner(name, *a, **k))[1])
    y = ops.relation_projection(v, tt, ct, wt, bt, p, seed, pregated)
    g = gy * (y.detach() > 0) if pregated else gy
    y.backward(g)
    monkeypatch.setattr(ops, "_launch", inner)
    return y.detach(), {"d_t": tt.grad, "d_c2": ct.grad, "d_w": wt.grad, "d_b": bt.grad}, seen


@pytest.mark.parametrize("B,N,D,L,p,pregated", [(64, 36, 2048, 310, 0.5, False), (64, 36, 2048, 310, 0.0, True), (33, 36, 256, 310, 0.5, True),
                                                 (41, 37, 384, 64, 0.5, False), (13, 100, 256, 310, 0.5, False), (145, 8, 128, 16, 0.0, False),
                                                 (2, 1000, 128, 34, 0.5, True)])
def test_relation_linear_fused_is_bit_identical_to_the_materialised_path(ops, monkeypatch, B, N, D, L, p, pregated):
    """y = relu(drop(t + c2 v) W^T + b) with the relation step applied to the GEMM's A fragments in registers
    (vqa_relation_linear_fwd_split; the weight gradient recomputes t + c2 v from v the same way, vqa_relation_linear_dw_split)
    against the path that writes v2 = drop(t + c2 v) to HBM first (vqa_relation_apply_fwd + vqa_linear_act_{fwd,dw}_split).  Both
    form the same fp32 value per element (one fma), apply the same mask bit and run the same split / MFMA / reduction order; the
    materialised path scales the kept elements by 2 before the GEMM, the fused one scales the result (powers of two: exact).  So
    the two agree BIT FOR BIT -- y, d_w, d_b, and d_t / d_c2 (same data-gradient kernel on the same gated gradient).  Shapes: the
    training tile-aligned case (36 regions: a 144-row tile is 4 whole samples), samples straddling tiles (37, 100, 1000 rows per
    sample, 8), row counts that are not a multiple of the tile, both relu-gate conventions."""
    L_ = ops._lib.lib()
    assert L_.vqa_relation_linear_split_supported(B, N, D, L, p) == 1
    gen = torch.Generator(device="cpu").manual_seed(B * 1000 + N)
    v, t, c2, w, b, gy = _relation_case(B, N, D, L, gen)
    M, seed, P = B * N, 99, (lambda z: z.data_ptr() if z is not None else None)
    ws = torch.empty(max(L_.vqa_linear_act_fwd_split_workspace_bytes(D, L), L_.vqa_linear_act_dw_split_workspace_bytes(M, D, L)) // 4 + 64,
                     device=dev())
    check = ops._lib.check
    # forward, through the C ABI: fused  vs  apply + linear
    y1, y0, x0 = torch.empty(B, N, L, device=dev()), torch.empty(B, N, L, device=dev()), torch.empty_like(v)
    check(L_.vqa_relation_linear_fwd_split(P(v), P(t), P(c2), P(w), P(b), P(y1), P(ws), ws.numel() * 4, B, N, D, L, 1, p, seed, None, None), "fused fwd")
    check(L_.vqa_relation_apply_fwd(P(v), P(t), P(c2), P(x0), p, seed, None, B, N, D, None), "apply")
    check(L_.vqa_linear_act_fwd_split(P(x0), D, P(w), P(b), P(y0), P(ws), ws.numel() * 4, M, D, L, 1, 0.0, 0, None, None), "linear fwd")
    assert torch.equal(y1, y0), (y1 - y0).abs().max().item()
    # weight gradient: fused  vs  the materialised input
    act = 0 if pregated else 1
    g = gy * (y0 > 0) if pregated else gy
    out = {}
    for fused in (True, False):
        dw, db = torch.empty(L, D, device=dev()), torch.empty(L, device=dev())
        gz = torch.empty_like(gy) if act else None
        if fused:
            check(L_.vqa_relation_linear_dw_split(P(v), P(t), P(c2), P(y0) if act else None, P(g), P(dw), P(db), P(gz), P(ws), ws.numel() * 4,
                                                  B, N, D, L, act, p, seed, None, None), "fused dw")
        else:
            check(L_.vqa_linear_act_dw_split(P(x0), D, P(y0) if act else None, P(g), P(dw), P(db), P(gz), P(ws), ws.numel() * 4, M, D, L, act,
                                             0.0, 0, None, None), "linear dw")
        torch.cuda.synchronize()
        out[fused] = (dw, db, gz)
    for a, c, name in zip(out[True], out[False], ("d_w", "d_b", "gz")):
        if a is not None:
            assert torch.equal(a, c), (name, (a - c).abs().max().item())
    if N == 36:   # ... and as the autograd node the model calls (its data-gradient kernel takes 36-region samples)
        y1a, g1, seen1 = _relation_run(ops, monkeypatch, True, v, t, c2, w, b, gy, p, seed, pregated)
        y0a, g0, seen0 = _relation_run(ops, monkeypatch, False, v, t, c2, w, b, gy, p, seed, pregated)
        assert "relation_linear_fwd_split" in seen1 and "relation_linear_dw_split" in seen1 and "relation_apply_fwd" not in seen1, seen1
        assert "relation_apply_fwd" in seen0 and "linear_act_fwd_split" in seen0 and "relation_linear_fwd_split" not in seen0, seen0
        assert torch.equal(y1a, y0a) and torch.equal(y1a, y1)
        for k in g1:
            assert torch.equal(g1[k], g0[k]), (k, (g1[k] - g0[k]).abs().max().item())
    # and against float64 (the materialised path's own accuracy claim, restated for the fused one)
    mask = ops.linear_dropout_mask(M, D, p, seed, dev()).view(B, N, D).double() if p > 0 else 1.0
    x64 = (t.double()[:, None, :] + c2.double()[:, None, :] * v.double()) * mask
    ref = torch.relu(x64.reshape(M, D) @ w.double().t() + b.double()).view(B, N, L)
    mx, rm = err(y1, ref)
    assert mx <= 2e-5 and rm <= 2e-6, (mx, rm)
    gz64 = (g.double() * (1.0 if pregated else (y0 > 0))).reshape(M, L)
    mx, rm = err(out[True][0], gz64.t() @ x64.reshape(M, D))
    assert mx <= 2e-5 and rm <= 2e-6, (mx, rm)


def test_relation_linear_fused_refuses_what_it_cannot_run(ops):
    L_ = ops._lib.lib()
    assert L_.vqa_relation_linear_split_supported(512, 36, 2048, 310, 0.5) == 1
    assert L_.vqa_relation_linear_split_supported(128, 100, 2048, 310, 0.5) == 1
    assert L_.vqa_relation_linear_split_supported(512, 36, 2048, 310, 0.3) == 0        # p = 0.5 masks only
    assert L_.vqa_relation_linear_split_supported(512, 4, 2048, 310, 0.5) == 0         # < 8 rows per sample
    assert L_.vqa_relation_linear_split_supported(512, 36, 2048 + 64, 310, 0.5) == 0   # D % 128
    assert L_.vqa_relation_linear_split_supported(512, 9, 2048, 310, 0.5) == 0         # 17 samples' (t, c2) rows do not fit LDS
    assert L_.vqa_relation_linear_split_supported(16, 36, 2048, 310, 0.5) == 0         # M < 1152: not a tall projection
    v = torch.zeros(512, 4, 2048, device=dev())
    rc = L_.vqa_relation_linear_fwd_split(v.data_ptr(), v.data_ptr(), v.data_ptr(), v.data_ptr(), None, v.data_ptr(), v.data_ptr(), 1 << 30,
                                          512, 4, 2048, 310, 1, 0.5, 0, None, None)
    assert rc != 0 and b"outside the fused form" in L_.vqa_last_error()


@pytest.mark.parametrize("kind", ["flt_max", "nonfinite", "nonfinite_masked"])
def test_relation_linear_fused_edge_values(ops, monkeypatch, kind):
    """The repair path with the relation operands: t / c2 / v next to FLT_MAX or non-finite.  The fused kernels and the materialised
    path return the same class (finite / +Inf / -Inf / NaN) everywhere and the same bits wherever the result is finite and the
    split was exact; where the repair path ran (a non-finite accumulator recomputed as an fp32 dot product of t + c2 v) they agree
    to fp32 dot-product accuracy.
    (A non-finite element that the mask DROPS is 0 in every in-register mask of this library -- the bits are cleared, as in K5's own
    forward -- and 0 x Inf = NaN in a path that multiplies a stored tensor by 0 / 2; so the non-finite t / c2 cases, which reach
    dropped and kept rows alike, are compared without dropout, and under dropout only kept elements of v are made non-finite.)"""
    B, N, D, L, p = 40, 36, 256, 310, (0.0 if kind == "nonfinite" else 0.5)
    gen = torch.Generator(device="cpu").manual_seed(5)
    v, t, c2, w, b, gy = _relation_case(B, N, D, L, gen)
    fmax = torch.finfo(torch.float32).max
    if kind == "flt_max":
        t[3, 7] = fmax * (1.0 - 2.0 ** -10)
        w[:, 7] *= 2.0 ** -30
        gy[3] *= 2.0 ** -60
        v[9, 5, 100] = -fmax * (1.0 - 2.0 ** -11)
        c2[9, 100] = 0.5
        w[:, 100] *= 2.0 ** -30
        gy[9] *= 2.0 ** -60
    elif kind == "nonfinite":
        t[2, 0], t[17, 255] = float("inf"), float("nan")
        c2[5, 9] = float("-inf")
        v[30, 35, 128] = float("inf")
        v[31, 0, 3] = float("nan")
    else:
        keep = ops.linear_dropout_mask(B * N, D, p, 7, dev()).view(B, N, D) > 0
        for (bb, nn, kk), val in (((30, 35, 128), float("inf")), ((31, 0, 3), float("nan")), ((0, 0, 0), float("-inf")), ((39, 35, 255), float("nan"))):
            while not bool(keep[bb, nn, kk]):
                kk = (kk + 1) % D
            v[bb, nn, kk] = val
    y1, g1, _ = _relation_run(ops, monkeypatch, True, v, t, c2, w, b, gy, p, 7)
    if kind == "flt_max":
        # (the materialised path is no yardstick here: it stores the kept elements times 2, which overflows next to FLT_MAX; the
        #  fused kernels mask unscaled and scale the result) -- float64 is
        mask = ops.linear_dropout_mask(B * N, D, p, 7, dev()).view(B, N, D).double()
        x64 = (t.double()[:, None, :] + c2.double()[:, None, :] * v.double()) * mask
        ref = torch.relu(x64.reshape(B * N, D) @ w.double().t() + b.double()).view(B, N, L)
        scale = (x64.abs().reshape(B * N, D) @ w.double().abs().t()).view(B, N, L) + b.double().abs()
        assert bool(torch.isfinite(y1).all()) and bool(torch.isfinite(g1["d_w"]).all()) and bool(torch.isfinite(g1["d_b"]).all())
        assert ((y1.double() - ref).abs() / scale).max().item() <= 4e-6
        gz = (gy.double() * (y1 > 0)).reshape(B * N, L)
        ref_dw, sc_dw = gz.t() @ x64.reshape(B * N, D), gz.abs().t() @ x64.abs().reshape(B * N, D)
        assert ((g1["d_w"].double() - ref_dw).abs() / sc_dw.clamp_min(1e-300)).max().item() <= 4e-6
        return
    y0, g0, _ = _relation_run(ops, monkeypatch, False, v, t, c2, w, b, gy, p, 7)
    assert torch.equal(_classes(y1), _classes(y0))
    fin = torch.isfinite(y0)
    assert (y1[fin] - y0[fin]).abs().max().item() <= 1e-4 * y0[fin].abs().max().item()
    for k in ("d_w", "d_b"):
        assert torch.equal(_classes(g1[k]), _classes(g0[k])), k
        fin = torch.isfinite(g0[k])
        assert (g1[k][fin] - g0[k][fin]).abs().max().item() <= 1e-4 * g0[k][fin].abs().max().item(), k


@pytest.mark.parametrize("M,N1,N2,pad", [(13312, 2400, 620, 0), (1152, 310, 128, 0), (4001, 482, 260, 4), (2304, 2400, 2400, 0), (1200, 16, 64, 8)])
def test_gemm_tn_split_against_float64(ops, M, N1, N2, pad):
    """vqa_gemm_tn_split (the question encoder's weight gradients: g^T x over all T*B rows; csrc/gru_gemm.hip) against float64:
    wide gradients packed in column groups (2400 = 5 groups of 30 blocks), row counts that are not a multiple of the 32-row
    chunk, padded row strides, the smallest shapes."""
    gen = torch.Generator(device="cpu").manual_seed(M + N1)
    g = (torch.randn(M, N1 + pad, generator=gen) / 32).to(dev())
    x = torch.randn(M, N2 + pad, generator=gen).to(dev())
    d_w = torch.empty(N1, N2, device=dev())
    assert ops.gemm_tn_split(g, 0, N1 + pad, x, 0, N2 + pad, d_w, M, N1, N2)
    ref = g[:, :N1].double().t() @ x[:, :N2].double()
    mx, rm = err(d_w, ref)
    assert mx <= 2e-5 and rm <= 2e-6, (mx, rm)
    # non-finite operands: the repair path
    g[5, 3], x[M - 1, N2 - 1] = float("inf"), float("nan")
    assert ops.gemm_tn_split(g, 0, N1 + pad, x, 0, N2 + pad, d_w, M, N1, N2)
    ref = g[:, :N1].t() @ x[:, :N2]
    assert torch.equal(_classes(d_w), _classes(ref))
    # outside the engine: the wrapper says so instead of launching
    assert not ops.gemm_tn_split(g, 0, N1 + pad, x, 0, N2 + pad, d_w, 1000, N1, N2)
