"""The split engine (csrc/gemm_f32_split.hpp, csrc/linear_split.hip; opt-in by VQA_F32_PRODUCTS=split): K5's forward and weight
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
