"""GPU parity tests: every HIP kernel (through the C ABI, via ops.py) against the float64 numpy oracle
(oracle/kernels_np.py) on the same seeded inputs.  Tolerance: 1e-3 relative fp32 is the north-star bar;
these kernels are expected to sit near 1e-5, so the tests use RTOL below and print the achieved error.
Run on the GPU box with:  python -m pytest tests -m gpu
"""
import os

import numpy as np
import pytest
import torch

from oracle import kernels_np as K
from oracle import seeded

pytestmark = pytest.mark.gpu

RTOL = 2e-4  # well inside the 1e-3 north-star tolerance


def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def g(a, requires_grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    return t.requires_grad_() if requires_grad else t


def close(name, got, want, rtol=RTOL):
    got = got.detach().cpu().numpy().astype(np.float64) if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert np.isfinite(got).all(), name + ": non-finite output"
    scale = max(np.abs(want).max(), 1e-20)
    err = np.abs(got - want).max() / scale
    assert err <= rtol, "%s: rel err %.3e > %.1e" % (name, err, rtol)
    return err


@pytest.fixture(scope="module")
def ops():
    from vqa_playground_pytorch_amd import ops as o
    return o


# ----------------------------------------------------------------------------------------------- K1
@pytest.mark.parametrize("B,N,D,G,glimpse", [(3, 5, 12, 1, 0), (2, 36, 2048, 4, 0), (2, 7, 260, 3, 2),
                                             (1, 1, 4, 1, 0), (2, 100, 512, 4, 0), (2, 37, 1024, 2, 1)])
@pytest.mark.parametrize("mode", [0, 1])
def test_pairwise_relation_fwd_bwd(ops, B, N, D, G, glimpse, mode):
    v = seeded.seeded_array((B, N, D), 101)
    q1 = 1 / (1 + np.exp(-seeded.seeded_array((B, D), 102)))
    q2 = 1 / (1 + np.exp(-seeded.seeded_array((B, D), 103)))
    al = np.abs(seeded.seeded_array((B, N, G), 104)) + 0.05          # deliberately not normalised
    gout = seeded.seeded_array((B, N, D), 105)
    vt, q1t, q2t, alt = g(v, True), g(q1, True), g(q2, True), g(al, True)
    out = ops.pairwise_relation_reduce(vt, q1t, q2t, alt, glimpse=glimpse, mode=mode)
    close("v2", out, K.pairwise_relation_reduce_fwd(v, q1, q2, al[:, :, glimpse]))
    out.backward(g(gout))
    da, dq1, dq2, dv = K.pairwise_relation_reduce_bwd(v, q1, q2, al[:, :, glimpse], gout)
    da_full = np.zeros_like(al, dtype=np.float64)
    da_full[:, :, glimpse] = da
    close("d_alpha", alt.grad, da_full)
    close("d_q1", q1t.grad, dq1)
    close("d_q2", q2t.grad, dq2)
    close("d_v", vt.grad, dv)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pairwise_relation_dual_outputs(ops, dtype):
    """dual=True: two aliases of v2, one per consumer; their gradients are added inside the backward kernel."""
    B, N, D, G = 3, 36, 512, 4
    gen = torch.Generator(device="cpu").manual_seed(9)
    v = torch.randn(B, N, D, generator=gen).to(dev()).to(dtype)
    q1, q2 = torch.rand(B, D, generator=gen).to(dev()), torch.rand(B, D, generator=gen).to(dev())
    al = torch.softmax(torch.randn(B, N, G, generator=gen), 1).to(dev())
    ga, gb = (torch.randn(B, N, D, generator=gen).to(dev()).to(dtype) for _ in range(2))
    outs = {}
    for dual in (False, True):
        leaves = [t.clone().requires_grad_() for t in (v, q1, q2, al)]
        if dual:
            a, b = ops.pairwise_relation_reduce(*leaves, glimpse=1, mode=1, dual=True)
            assert a.data_ptr() == b.data_ptr()
            ((a.float() * ga.float()).sum() + (b.float() * gb.float()).sum()).backward()
        else:
            a = ops.pairwise_relation_reduce(*leaves, glimpse=1, mode=1)
            (a.float() * (ga.float() + gb.float())).sum().backward()
        outs[dual] = [t.grad.float() for t in leaves]
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5     # bf16: the single-tensor path rounds ga+gb to bf16 first
    for x, y in zip(outs[False], outs[True]):
        assert ((x - y).abs().max() / x.abs().max()).item() <= tol
    # one consumer only: the other gradient is None
    leaves = [t.clone().requires_grad_() for t in (v, q1, q2, al)]
    a, b = ops.pairwise_relation_reduce(*leaves, glimpse=1, mode=1, dual=True)
    (b.float() * (ga.float() + gb.float())).sum().backward()
    assert ((leaves[1].grad - outs[False][1]).abs().max() / outs[False][1].abs().max()).item() <= tol


def test_pairwise_relation_modes_agree_full_size(ops):
    """BASELINE size (B=512, 36x2048): pairwise and factored evaluation agree, and the kernel is linear in alpha."""
    B, N, D = 512, 36, 2048
    gen = torch.Generator(device="cpu").manual_seed(1)
    v = torch.randn(B, N, D, generator=gen).to(dev())
    q1 = torch.rand(B, D, generator=gen).to(dev())
    q2 = torch.rand(B, D, generator=gen).to(dev())
    al = torch.softmax(torch.randn(B, N, 4, generator=gen), dim=1).to(dev())
    a = ops.pairwise_relation_reduce(v, q1, q2, al, 0, 0)
    b = ops.pairwise_relation_reduce(v, q1, q2, al, 0, 1)
    scale = a.abs().max().item()
    assert (a - b).abs().max().item() <= 1e-5 * scale
    c = ops.pairwise_relation_reduce(v, q1, q2, 2.0 * al, 0, 1)
    assert (c - 2.0 * b).abs().max().item() <= 1e-5 * scale
    # closed form with normalised alpha: q1 * pooled + q2 * v
    pooled = torch.einsum("bn,bnd->bd", al[:, :, 0], v)
    ref = q1[:, None, :] * pooled[:, None, :] + q2[:, None, :] * v
    assert (b - ref).abs().max().item() <= 2e-5 * scale


def test_pairwise_relation_rejects_bad_shapes(ops):
    from vqa_playground_pytorch_amd import _lib
    v = torch.zeros(2, 5, 10, device=dev())  # D % 4 != 0
    with pytest.raises(_lib.VqaLibraryError):
        ops.pairwise_relation_reduce(v, torch.zeros(2, 10, device=dev()), torch.zeros(2, 10, device=dev()),
                                     torch.zeros(2, 5, 1, device=dev()))
    with pytest.raises(ValueError):
        ops.pairwise_relation_reduce(torch.zeros(2, 5, 8, device=dev()), torch.zeros(2, 8, device=dev()),
                                     torch.zeros(2, 8, device=dev()), torch.zeros(2, 4, 1, device=dev()))


# ----------------------------------------------------------------------------------------------- K3
@pytest.mark.parametrize("B,N,D,G", [(3, 5, 12, 2), (4, 36, 2048, 4), (2, 100, 2048, 4), (2, 1, 8, 1),
                                     (3, 70, 1028, 8), (2, 36, 4096, 3)])
@pytest.mark.parametrize("with_ext,need_dv", [(True, True), (False, False)])
def test_softmax_attention_pool(ops, B, N, D, G, with_ext, need_dv):
    logits = 2.0 * seeded.seeded_array((B, N, G), 201)
    v = seeded.seeded_array((B, N, D), 202)
    gp = seeded.seeded_array((B, G, D), 203)
    ga = seeded.seeded_array((B, N, G), 204)
    lt, vt = g(logits, True), g(v, need_dv)
    alpha, pooled = ops.softmax_attention_pool(lt, vt)
    a_np, p_np = K.softmax_attention_pool_fwd(logits, v)
    close("alpha", alpha, a_np)
    close("pooled", pooled, p_np)
    loss = (pooled * g(gp)).sum()
    if with_ext:
        loss = loss + (alpha * g(ga)).sum()
    loss.backward()
    dl, dv = K.softmax_attention_pool_bwd(a_np, v, gp, ga if with_ext else None)
    close("d_logits", lt.grad, dl)
    if need_dv:
        close("d_v", vt.grad, dv)
    else:
        assert vt.grad is None


@pytest.mark.parametrize("p_drop", [0.0, 0.5, 0.25])
@pytest.mark.parametrize("B,N,D,G,want_first", [(3, 36, 2048, 4, True), (2, 7, 72, 3, False), (5, 100, 256, 4, True)])
def test_softmax_attention_pool_with_glimpse_dropout(ops, B, N, D, G, want_first, p_drop):
    """K3 with MyATT's glimpse-projection dropout in its store and the undropped glimpse 0 as a second output: equal to
    the plain kernel followed by the exported mask; backward against the oracle fed the masked gradient."""
    logits = 2.0 * seeded.seeded_array((B, N, G), 221)
    v = seeded.seeded_array((B, N, D), 222)
    gp = seeded.seeded_array((B, G, D), 223)
    gf = seeded.seeded_array((B, D), 224)
    seed = 1234567
    lt = g(logits, True)
    res = ops.softmax_attention_pool_drop(lt, g(v), p_drop, seed, want_first)
    alpha, pooled = res[0], res[1]
    mask = ops.linear_dropout_mask(B * G, D, p_drop, seed, lt.device).view(B, G, D) if p_drop else torch.ones(B, G, D, device=lt.device)
    a_np, p_np = K.softmax_attention_pool_fwd(logits, v)
    m_np = mask.cpu().numpy()
    close("alpha", alpha, a_np)
    close("pooled", pooled, p_np * m_np)
    if p_drop:
        frac = float((m_np == 0).mean())
        assert abs(frac - p_drop) < 0.02 and set(np.unique(m_np)) <= {0.0, np.float32(1.0 / (1.0 - p_drop))}
    loss = (pooled * g(gp)).sum()
    if want_first:
        close("first", res[2], p_np[:, 0])
        loss = loss + (res[2] * g(gf)).sum()
    loss.backward()
    gp_total = gp * m_np
    if want_first:
        gp_total[:, 0] += gf
    dl, _ = K.softmax_attention_pool_bwd(a_np, v, gp_total, None)
    close("d_logits", lt.grad, dl)


@pytest.mark.parametrize("p_drop", [0.0, 0.5])
@pytest.mark.parametrize("B,N,D,G,with_ext,need_dv", [(3, 36, 2048, 4, True, False), (2, 7, 1028, 2, False, True),
                                                      (4, 100, 2048, 4, True, True), (2, 36, 1540, 8, False, False)])
def test_softmax_attention_pool_backward_single_launch(ops, lib_option, B, N, D, G, with_ext, need_dv, p_drop):
    """The one-launch backward (a workgroup owns a sample: dot products, d_v and the softmax backward together; taken from
    B = 512 on, here forced for small batches) against the oracle, with the gradient on alpha, d_v, glimpse dropout and
    the gradient of the undropped glimpse 0."""
    lib_option("VQA_K3_FUSED_MIN_B", 1)
    logits = 2.0 * seeded.seeded_array((B, N, G), 241)
    v = seeded.seeded_array((B, N, D), 242)
    gp = seeded.seeded_array((B, G, D), 243)
    ga = seeded.seeded_array((B, N, G), 244)
    gf = seeded.seeded_array((B, D), 245)
    seed = 99
    lt, vt = g(logits, True), g(v, need_dv)
    alpha, pooled, first = ops.softmax_attention_pool_drop(lt, vt, p_drop, seed, True)
    mask = ops.linear_dropout_mask(B * G, D, p_drop, seed, lt.device).view(B, G, D).cpu().numpy() if p_drop else np.ones((B, G, D))
    loss = (pooled * g(gp)).sum() + (first * g(gf)).sum()
    if with_ext:
        loss = loss + (alpha * g(ga)).sum()
    loss.backward()
    a_np, _ = K.softmax_attention_pool_fwd(logits, v)
    gp_total = gp * mask
    gp_total[:, 0] += gf
    dl, dv = K.softmax_attention_pool_bwd(a_np, v, gp_total, ga if with_ext else None)
    close("d_logits", lt.grad, dl)
    if need_dv:
        close("d_v", vt.grad, dv)


def test_softmax_attention_pool_large_logits(ops):
    """softmax must be max-shifted: logits around +/-80 would overflow a naive exp."""
    B, N, D, G = 2, 36, 64, 4
    logits = 80.0 * seeded.seeded_array((B, N, G), 211)
    v = seeded.seeded_array((B, N, D), 212)
    alpha, pooled = ops.softmax_attention_pool(g(logits), g(v))
    a_np, p_np = K.softmax_attention_pool_fwd(logits, v)
    close("alpha", alpha, a_np)
    close("pooled", pooled, p_np)
    assert abs(alpha.sum(dim=1).cpu().numpy() - 1.0).max() < 1e-5


@pytest.mark.parametrize("p_drop", [0.5, 0.25])
@pytest.mark.parametrize("shape,groups", [((64, 2400), 4), ((33, 510), 0), ((7, 3, 310), 0), ((6, 31), 3), ((16, 2048), 2)])
def test_dropout_groups(ops, shape, groups, p_drop, monkeypatch):
    """ops.dropout: G independent hash-mask draws over one input (or one, groups = 0); forward equals x times the exported
    mask, backward is the masked sum over the groups; the drop rate is p."""
    x = seeded.seeded_array(shape, 231)
    seed = 424242
    monkeypatch.setattr(ops, "next_dropout_seed", lambda: seed)
    xt = g(x, True)
    y = ops.dropout(xt, p_drop, groups=groups)
    G = max(groups, 1)
    K = shape[-1]
    M = int(np.prod(shape[:-1]))
    # (the mask is a function of the flat element index: export it as one even-length row when K is odd)
    mask = (ops.linear_dropout_mask(G * M, K, p_drop, seed, xt.device) if K % 2 == 0 else
            ops.linear_dropout_mask(1, G * M * K, p_drop, seed, xt.device)).cpu().numpy().reshape((G,) + tuple(shape))
    assert tuple(y.shape) == ((groups,) + tuple(shape) if groups else tuple(shape))
    close("out", y.reshape((G,) + tuple(shape)), x[None] * mask)
    assert abs(float((mask == 0).mean()) - p_drop) < (0.03 if mask.size > 20000 else 0.12)
    gy = seeded.seeded_array(tuple(y.shape), 232)
    y.backward(g(gy))
    close("d_x", xt.grad, (gy.reshape((G,) + tuple(shape)) * mask).sum(0))


# ----------------------------------------------------------------------------------------------- K4
def _fusion_case(ops, B, N, L, H, R, seed, need_dx=True, two_d=False):
    x = seeded.seeded_array((B, L) if two_d else (B, N, L), seed)
    w1 = seeded.seeded_array((R, H, L), seed + 1, scale=1.0 / np.sqrt(L))
    b1 = seeded.seeded_array((R, H), seed + 2, scale=0.1)
    h2 = seeded.seeded_array((B, R, H), seed + 3)
    go = seeded.seeded_array((B, H) if two_d else (B, N, H), seed + 4)
    xt, h2t = g(x, need_dx), g(h2, True)
    ws = [g(w1[r], True) for r in range(R)]
    bs = [g(b1[r], True) for r in range(R)]
    out = ops.lowrank_bilinear_fusion(xt, h2t, ws, bs)
    x3 = x.reshape(B, N, L)
    out_np, _ = K.lowrank_bilinear_fusion_fwd(x3, w1, b1, h2)
    close("out", out.reshape(B, N, H), out_np)
    out.backward(g(go))
    dx, dw1, db1, dh2 = K.lowrank_bilinear_fusion_bwd(x3, w1, b1, h2, go.reshape(B, N, H))
    if need_dx:
        close("d_x", xt.grad.reshape(B, N, L), dx)
    close("d_h2", h2t.grad, dh2)
    for r in range(R):
        close("d_w1[%d]" % r, ws[r].grad, dw1[r])
        close("d_b1[%d]" % r, bs[r].grad, db1[r])


@pytest.mark.parametrize("form", ["folded", "engine"])
@pytest.mark.parametrize("B,N,L,H,R", [(3, 5, 8, 16, 2), (4, 36, 310, 510, 2), (2, 100, 310, 510, 2),
                                       (5, 7, 34, 66, 3), (1, 1, 2, 2, 1), (9, 36, 310, 510, 2)])
def test_lowrank_bilinear_fusion(ops, monkeypatch, form, B, N, L, H, R):
    """Both forms of K4: the rank-folded one (one contraction per sample against sum_r h2_r (.) W1_r, nothing saved for
    backward) and the R-GEMM tile-engine one."""
    monkeypatch.setattr(ops, "_K4_FORM", form)
    _fusion_case(ops, B, N, L, H, R, 300)


@pytest.mark.parametrize("B,N,L,H,R", [
    (3, 16, 32, 48, 2),      # L a multiple of the chunk: the bias rides in an extra chunk; one region block
    (13, 17, 30, 80, 4),     # two region blocks, rank 4, batch tail of the 8- and 4-sample groups
    (6, 32, 18, 64, 3), (5, 33, 50, 130, 1), (4, 48, 310, 510, 4),
    (3, 49, 70, 96, 2), (2, 80, 40, 62, 2), (3, 81, 36, 160, 1), (2, 112, 64, 100, 2),   # 5- and 7-block variants
    (17, 36, 310, 330, 2),   # output features: 5 tiles of 64 forward, 80-wide tiles in the data gradient
])
def test_lowrank_bilinear_fusion_folded_shapes(ops, monkeypatch, B, N, L, H, R):
    from vqa_playground_pytorch_amd import _lib
    assert _lib.lib().vqa_lowrank_bilinear_fusion_folded_supported(B, N, L, H, R) == 1
    monkeypatch.setattr(ops, "_K4_FORM", "folded")
    _fusion_case(ops, B, N, L, H, R, 360)
    _fusion_case(ops, B, N, L, H, R, 370, need_dx=False)


def test_lowrank_bilinear_fusion_folded_limits(ops, monkeypatch):
    """Shapes outside the folded form are served by the tile-engine kernels (same results, checked above)."""
    monkeypatch.setattr(ops, "_K4_FORM", "folded")
    from vqa_playground_pytorch_amd import _lib
    sup = _lib.lib().vqa_lowrank_bilinear_fusion_folded_supported
    assert sup(4, 113, 310, 510, 2) == 0 and sup(4, 36, 310, 510, 5) == 0 and sup(4, 100, 310, 510, 3) == 0
    assert sup(4, 36, 311, 510, 2) == 0 and sup(4, 100, 310, 510, 2) == 1
    _fusion_case(ops, 2, 113, 20, 34, 2, 380)      # N too large: falls through to the engine form
    _fusion_case(ops, 3, 36, 20, 34, 5, 381)       # R too large
    monkeypatch.setattr(ops, "_K4_FORM", "auto")   # the default: folded from 32 samples on, engine below
    _fusion_case(ops, 33, 36, 40, 66, 2, 382)
    _fusion_case(ops, 31, 36, 40, 66, 2, 383)


@pytest.mark.parametrize("B,L,H,R", [(7, 1240, 510, 2), (5, 620, 510, 5), (64, 1240, 510, 2)])
def test_lowrank_bilinear_fusion_2d(ops, B, L, H, R):
    _fusion_case(ops, B, 1, L, H, R, 320, two_d=True)


@pytest.mark.parametrize("tile", ["128x128", "64x128", "128x64", "64x64"])
def test_lowrank_bilinear_fusion_every_tile_shape(ops, tile, monkeypatch, lib_option):
    lib_option("VQA_GEMM_TILE", tile)
    monkeypatch.setattr(ops, "_K4_FORM", "engine")
    _fusion_case(ops, 6, 36, 310, 510, 2, 340)
    _fusion_case(ops, 3, 50, 70, 130, 2, 350, need_dx=False)


def test_lowrank_bilinear_fusion_golden(ops, golden_dir):
    """Against the reference's own MutanFusion output (tests/golden/blocks.npz, mutan3d)."""
    blocks = np.load(os.path.join(golden_dir, "blocks.npz"))
    from vqa_playground_pytorch_amd.layers import MutanFusion
    mf = seeded.load_state(MutanFusion(8, 6, 16, 2), 21).to(dev())
    y1, y2 = g(seeded.seeded_array((3, 5, 8), 22), True), g(seeded.seeded_array((3, 6), 23), True)
    o = mf(y1, y2)
    (o * g(seeded.seeded_array((3, 5, 16), 24))).sum().backward()
    close("out", o, blocks["mutan3d.out"])
    close("dx1", y1.grad, blocks["mutan3d.dx1"])
    close("dx2", y2.grad, blocks["mutan3d.dx2"])
    for name, p in mf.named_parameters():
        close(name, p.grad, blocks["mutan3d.g." + name + ".full"])
    mf2 = seeded.load_state(MutanFusion(8, 6, 16, 3), 25).to(dev())
    z1, z2 = g(seeded.seeded_array((3, 8), 26), True), g(seeded.seeded_array((3, 6), 27), True)
    o2 = mf2(z1, z2)
    (o2 * g(seeded.seeded_array((3, 16), 28))).sum().backward()
    close("out2d", o2, blocks["mutan2d.out"])
    close("dx1_2d", z1.grad, blocks["mutan2d.dx1"])
    close("dx2_2d", z2.grad, blocks["mutan2d.dx2"])
    for name, p in mf2.named_parameters():
        close(name, p.grad, blocks["mutan2d.g." + name + ".full"])


@pytest.mark.parametrize("B", [40, 5])
def test_lowrank_bilinear_fusion_gated_dx(ops, B):
    """gate_dx: the gradient of x comes back multiplied by (x > 0) -- the relu gate of the layer that produced x, applied in
    the store of the folded data-gradient kernel (B >= 32) or as a torch op behind the R-GEMM form -- everything else unchanged."""
    N, L, H, R = 36, 70, 130, 2
    gen = torch.Generator(device="cpu").manual_seed(12)
    x = torch.relu(torch.randn(B, N, L, generator=gen)).to(dev())          # about half the entries are exactly 0
    h2 = torch.randn(B, R, H, generator=gen).to(dev())
    ws = [(torch.randn(H, L, generator=gen) / L ** 0.5).to(dev()).requires_grad_() for _ in range(R)]
    bs = [(torch.randn(H, generator=gen) * 0.1).to(dev()).requires_grad_() for _ in range(R)]
    go = torch.randn(B, N, H, generator=gen).to(dev())
    grads = []
    for gate in (False, True):
        xr, hr = x.clone().requires_grad_(), h2.clone().requires_grad_()
        out = ops.lowrank_bilinear_fusion(xr, hr, ws, bs, gate_dx=gate)
        grads.append(torch.autograd.grad(out, [xr, hr] + ws + bs, go))
    plain, gated = grads
    assert torch.equal(gated[0], plain[0] * (x > 0)), "d_x must be exactly the ungated gradient times (x > 0)"
    assert (gated[0] != plain[0]).any()
    for a, b in zip(gated[1:], plain[1:]):
        assert torch.equal(a, b)


def test_block_goldens_conv_linear(ops, golden_dir):
    """HIP-backed MyConv1d (softmax over regions, relu) and MyLinear (sigmoid) against the reference's own outputs
    (tests/golden/blocks.npz conv_softmax / conv_relu / linear_sigmoid; config/CoR2.py:56-122), eval mode."""
    blocks = np.load(os.path.join(golden_dir, "blocks.npz"))
    from vqa_playground_pytorch_amd.layers import MyConv1d, MyLinear
    f = g(seeded.seeded_array((3, 5, 16), 32))
    cv = seeded.load_state(MyConv1d(16, 2, 1, 1, p=0.5, af="softmax", dim=1), 31).eval().to(dev())
    close("conv_softmax", cv(f), blocks["conv_softmax.out"])
    cr = seeded.load_state(MyConv1d(16, 7, 1, 1, p=0.5, af="relu"), 33).eval().to(dev())
    close("conv_relu", cr(f), blocks["conv_relu.out"])
    ml = seeded.load_state(MyLinear(16, 7, p=0.5, af="sigmoid"), 34).eval().to(dev())
    close("linear_sigmoid", ml(f), blocks["linear_sigmoid.out"])


def test_block_goldens_attention(ops, golden_dir):
    """HIP-backed MyATT (K3a logits + K3 softmax/pooling + the batched glimpse projections) against the reference's MyATT:
    outputs, input gradients and every parameter gradient (blocks.npz att.*; config/CoR2.py:125-157)."""
    blocks = np.load(os.path.join(golden_dir, "blocks.npz"))
    from vqa_playground_pytorch_amd.layers import MyATT
    att = seeded.load_state(MyATT(16, 2, 12, 8, af="relu"), 41).eval().to(dev())
    inp, fu = g(seeded.seeded_array((3, 5, 12), 42), True), g(seeded.seeded_array((3, 5, 16), 43), True)
    xv, latt = att(inp, fu)
    assert isinstance(latt, (tuple, list)) and len(latt) == 2 and latt[0].shape == (3, 5, 1)
    alpha = torch.cat(list(latt), dim=2)
    ((xv * g(seeded.seeded_array((3, 8), 44))).sum() + (alpha * g(seeded.seeded_array((3, 5, 2), 45))).sum()).backward()
    close("att.x_v", xv, blocks["att.x_v"])
    close("att.alpha", alpha, blocks["att.alpha"])
    close("att.dinputs", inp.grad, blocks["att.dinputs"])
    close("att.dfuse", fu.grad, blocks["att.dfuse"])
    for name, p in att.named_parameters():
        want = blocks["att.g." + name + ".full"]
        got = p.grad.detach().cpu().numpy().astype(np.float64)
        # (the conv bias sits in front of a softmax over regions: its gradient is mathematically zero -- the sum of O(1)
        #  terms that cancel, i.e. ~1e-7 of rounding noise on both sides -- and is compared absolutely)
        atol = 1e-6 if name == "conv_att.conv.bias" else 1e-7
        assert np.abs(got - want).max() <= RTOL * np.abs(want).max() + atol, name


@pytest.mark.parametrize("mode", [0, 1])
def test_block_goldens_relation(ops, golden_dir, mode):
    """K1 (pairwise and factored form) against the reference's decare_cat + alpha-weighted reduce at small dims
    (blocks.npz decare.*; config/CoR2.py:191-199, :216); the gates come from HIP-backed MyLinear stacks."""
    blocks = np.load(os.path.join(golden_dir, "blocks.npz"))
    from vqa_playground_pytorch_amd.layers import MyLinear
    mods = [seeded.load_state(MyLinear(6, 4, p=0.5, af="relu"), 51), seeded.load_state(MyLinear(4, 12, p=0.5, af="sigmoid"), 52),
            seeded.load_state(MyLinear(6, 4, p=0.5, af="relu"), 53), seeded.load_state(MyLinear(4, 12, p=0.5, af="sigmoid"), 54)]
    c1, e1, c2, e2 = (m.eval().to(dev()) for m in mods)
    vv, qq = g(seeded.seeded_array((3, 5, 12), 55)), g(seeded.seeded_array((3, 6), 56))
    al = torch.softmax(g(seeded.seeded_array((3, 5, 1), 57)), dim=1)
    q1, q2 = e1(c1(qq)), e2(c2(qq))
    close("decare.q1", q1, blocks["decare.q1"])
    close("decare.q2", q2, blocks["decare.q2"])
    v2 = ops.pairwise_relation_reduce(vv, q1, q2, al.contiguous(), glimpse=0, mode=mode)
    close("decare.v2", v2, blocks["decare.v2"])
    # the materialised pairwise tensor itself, row by row through the kernel: a one-hot alpha picks out[b, i, :, :]
    cat = blocks["decare.cat"]                                        # [3, 5, 5, 12]: [b, i, j, :] = v_i q1 + v_j q2
    for i in range(5):
        onehot = torch.zeros(3, 5, 1, device=dev())
        onehot[:, i] = 1.0
        close("decare.cat[:, %d]" % i, ops.pairwise_relation_reduce(vv, q1, q2, onehot, glimpse=0, mode=mode), cat[:, i])


@pytest.mark.parametrize("form", ["folded", "engine"])
def test_lowrank_bilinear_fusion_full_size_properties(ops, monkeypatch, form):
    """B=512 (M=18432): rank-sum linearity in h2 and agreement with a torch fp32 matmul restatement on GPU."""
    monkeypatch.setattr(ops, "_K4_FORM", form)
    B, N, L, H, R = 512, 36, 310, 510, 2
    gen = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(B, N, L, generator=gen).to(dev())
    h2 = torch.randn(B, R, H, generator=gen).to(dev())
    ws = [(torch.randn(H, L, generator=gen) / L ** 0.5).to(dev()) for _ in range(R)]
    bs = [(0.1 * torch.randn(H, generator=gen)).to(dev()) for _ in range(R)]
    out = ops.lowrank_bilinear_fusion(x, h2, ws, bs)
    ref = sum((x.double() @ ws[r].double().t() + bs[r].double()) * h2[:, r, None, :].double() for r in range(R))
    scale = ref.abs().max().item()
    assert (out.double() - ref).abs().max().item() <= 2e-5 * scale
    out2 = ops.lowrank_bilinear_fusion(x, 3.0 * h2, ws, bs)
    assert (out2 - 3.0 * out).abs().max().item() <= 1e-5 * scale * 3


@pytest.mark.parametrize("form", ["folded", "engine", "folded+dw_split"])
@pytest.mark.parametrize("B,N", [(1501, 36), (203, 100), (4099, 36), (512, 36), (333, 36)])
def test_lowrank_bilinear_fusion_large_batch_against_torch_fp64(ops, monkeypatch, lib_option, form, B, N):
    """Folded K4, forward and every gradient, at batches that are no multiple of its 8- / 4-sample groups and give the
    weight-gradient kernel uneven sample slabs -- 4099 samples also exceed what one slab can keep of the question-side
    factors in LDS (regression: the slab count must grow with the batch); reference = torch autograd in fp64 on the GPU.
    folded+dw_split: the weight gradient on the split engine (csrc/bilinear_dw_split.hip, VQA_K4_DW_SPLIT=1; it serves N = 36 and
    B <= 512, every other shape stays on the fp32 kernels) -- same bars."""
    if form == "folded+dw_split":
        lib_option("VQA_K4_DW_SPLIT", "1")
        form = "folded"
    monkeypatch.setattr(ops, "_K4_FORM", form)
    L, H, R = 310, 510, 2
    gen = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(B, N, L, generator=gen).to(dev()).requires_grad_()
    h2 = torch.randn(B, R, H, generator=gen).to(dev()).requires_grad_()
    ws = [(torch.randn(H, L, generator=gen) / L ** 0.5).to(dev()).requires_grad_() for _ in range(R)]
    bs = [(0.1 * torch.randn(H, generator=gen)).to(dev()).requires_grad_() for _ in range(R)]
    go = torch.randn(B, N, H, generator=gen).to(dev())
    leaves = [x, h2, *ws, *bs]
    out = ops.lowrank_bilinear_fusion(x, h2, ws, bs)
    got = torch.autograd.grad(out, leaves, go)
    ld = [t.detach().double().requires_grad_() for t in leaves]
    ref_out = sum((ld[0] @ ld[2 + r].t() + ld[2 + R + r]) * ld[1][:, r, None, :] for r in range(R))
    want = torch.autograd.grad(ref_out, ld, go.double())
    assert (out.double() - ref_out).abs().max().item() <= 2e-5 * ref_out.abs().max().item()
    for name, a, b in zip(["d_x", "d_h2"] + ["d_w1[%d]" % r for r in range(R)] + ["d_b1[%d]" % r for r in range(R)], got, want):
        err = (a.double() - b).abs().max().item() / b.abs().max().item()
        assert err <= 2e-5, "%s: %.2e" % (name, err)


def test_relation_gates_match_autograd(ops):
    """ops.RelationGates: (t, c2) = (q1 * pooled, q2) handed out twice; both consumers' gradients are combined by ONE kernel.
    Against plain autograd on the same expression, with two, one and no second consumer."""
    B, D = 6, 72
    q1, q2, po = (seeded.seeded_array((B, D), 431 + i).astype(np.float32) for i in range(3))
    w = [seeded.seeded_array((B, D), 441 + i).astype(np.float32) for i in range(4)]
    for use in ((1, 1, 1, 1), (1, 1, 0, 0), (0, 1, 1, 0)):
        a, b, c = g(q1, True), g(q2, True), g(po, True)
        outs = ops.relation_gates(a, b, c)
        sum(u * (o * g(wi)).sum() for u, o, wi in zip(use, outs, w)).backward()
        a2, b2, c2 = g(q1, True), g(q2, True), g(po, True)
        t = a2 * c2
        sum(u * (o * g(wi)).sum() for u, o, wi in zip(use, (t, b2, t, b2), w)).backward()
        for name, x, y in (("d_q1", a, a2), ("d_q2", b, b2), ("d_pooled", c, c2)):
            close(name, x.grad, y.grad.cpu().numpy().astype(np.float64))
    # pooled = glimpse 0 of a [B,G,D] tensor read IN PLACE (rows G * D floats apart): no contiguous copy, same numbers
    full = g(seeded.seeded_array((B, 4, D), 451).astype(np.float32), True)
    a, b = g(q1, True), g(q2, True)
    outs = ops.relation_gates(a, b, full[:, 0])
    sum((o * g(wi)).sum() for o, wi in zip(outs, w)).backward()
    full2, a2, b2 = full.detach().clone().requires_grad_(), g(q1, True), g(q2, True)
    t = a2 * full2[:, 0]
    sum((o * g(wi)).sum() for o, wi in zip((t, b2, t, b2), w)).backward()
    close("t (strided pooled)", outs[0], t.detach().cpu().numpy().astype(np.float64))
    for name, x, y in (("d_q1", a, a2), ("d_q2", b, b2), ("d_pooled", full, full2)):
        close(name + " (strided pooled)", x.grad, y.grad.cpu().numpy().astype(np.float64))


# ----------------------------------------------------------------------------------------------- K2
@pytest.mark.parametrize("B,N,L,G", [(2, 4, 6, 3), (3, 36, 310, 4), (2, 13, 70, 2), (1, 1, 5, 1), (2, 37, 100, 8)])
def test_object_difference_no_dropout(ops, B, N, L, G):
    vl = np.abs(seeded.seeded_array((B, N, L), 401))
    ql = np.abs(seeded.seeded_array((B, L), 402))
    w = seeded.seeded_array((G, N * L), 403, scale=1.0 / np.sqrt(N * L))
    bias = seeded.seeded_array((G,), 404, scale=0.1)
    gl = seeded.seeded_array((B, N, G), 405)
    vt, qt, wt, bt = g(vl, True), g(ql, True), g(w, True), g(bias, True)
    logits = ops.object_difference_attention(vt, qt, wt, bt, 0.0, 0)
    close("logits", logits, K.object_difference_logits_fwd(vl, ql, w, bias))
    logits.backward(g(gl))
    dvl, dql, dw, db = K.object_difference_logits_bwd(vl, ql, w, gl)
    close("d_vl", vt.grad, dvl)
    close("d_ql", qt.grad, dql)
    close("d_w", wt.grad, dw)
    close("d_bias", bt.grad, db)


@pytest.mark.parametrize("B,N,L,G,p", [(3, 36, 310, 4, 0.5), (2, 13, 70, 2, 0.25), (130, 5, 64, 4, 0.5),
                                       (180, 13, 200, 2, 0.5), (140, 36, 310, 4, 0.5)])   # the last two: the chunk-split data gradient
def test_object_difference_with_dropout(ops, B, N, L, G, p):
    """The fused kernels regenerate the mask from (seed, index); export it and hand it to the oracle."""
    seed = 1234567
    vl = np.abs(seeded.seeded_array((B, N, L), 411))
    ql = np.abs(seeded.seeded_array((B, L), 412))
    w = seeded.seeded_array((G, N * L), 413, scale=1.0 / np.sqrt(N * L))
    bias = seeded.seeded_array((G,), 414, scale=0.1)
    gl = seeded.seeded_array((B, N, G), 415)
    mask = ops.object_difference_dropout_mask(B, N, L, p, seed, dev()).cpu().numpy()
    keep = (mask > 0).mean()
    assert set(np.unique(mask)).issubset({0.0, np.float32(1.0 / (1.0 - p))})
    assert abs(keep - (1.0 - p)) < 0.01, keep
    vt, qt, wt, bt = g(vl, True), g(ql, True), g(w, True), g(bias, True)
    logits = ops.object_difference_attention(vt, qt, wt, bt, p, seed)
    close("logits", logits, K.object_difference_logits_fwd(vl, ql, w, bias, mask))
    logits.backward(g(gl))
    dvl, dql, dw, db = K.object_difference_logits_bwd(vl, ql, w, gl, mask)
    close("d_vl", vt.grad, dvl)
    close("d_ql", qt.grad, dql)
    close("d_w", wt.grad, dw)
    close("d_bias", bt.grad, db)
    other = ops.object_difference_dropout_mask(B, N, L, p, seed + 1, dev()).cpu().numpy()
    assert 0.4 < (other == mask).mean() < 0.6 + abs(0.5 - p), "a different seed must give a different mask"


@pytest.mark.parametrize("B,N,L,G,p", [(3, 36, 310, 4, 0.5), (2, 13, 70, 2, 0.25), (2, 36, 310, 4, 0.0), (180, 13, 200, 2, 0.5)])
def test_object_difference_gated_region_gradient(ops, B, N, L, G, p):
    """gate_dvl: vl is a relu output (zeros where the layer in front was inactive) and d_vl comes back multiplied by
    (vl > 0) -- the gradient with respect to that layer's pre-activation; every other output is unchanged."""
    seed = 424242
    vl = np.maximum(seeded.seeded_array((B, N, L), 421), 0.0)            # about half the entries exactly 0
    ql = np.abs(seeded.seeded_array((B, L), 422))
    w = seeded.seeded_array((G, N * L), 423, scale=1.0 / np.sqrt(N * L))
    bias = seeded.seeded_array((G,), 424, scale=0.1)
    gl = seeded.seeded_array((B, N, G), 425)
    mask = ops.object_difference_dropout_mask(B, N, L, p, seed, dev()).cpu().numpy() if p else None
    vt, qt, wt, bt = g(vl, True), g(ql, True), g(w, True), g(bias, True)
    logits = ops.object_difference_attention(vt, qt, wt, bt, p, seed, gate_dvl=True)
    close("logits", logits, K.object_difference_logits_fwd(vl, ql, w, bias, mask))
    logits.backward(g(gl))
    dvl, dql, dw, db = K.object_difference_logits_bwd(vl, ql, w, gl, mask)
    assert 0.3 < (vl > 0).mean() < 0.7
    close("d_vl (gated)", vt.grad, dvl * (vl > 0))
    assert float(vt.grad.cpu()[torch.from_numpy(vl <= 0)].abs().max()) == 0.0
    close("d_ql", qt.grad, dql)
    close("d_w", wt.grad, dw)
    close("d_bias", bt.grad, db)


@pytest.mark.parametrize("B,N,L,G,p", [(3, 36, 310, 4, 0.5), (5, 13, 70, 2, 0.5), (140, 36, 310, 4, 0.5), (512, 36, 310, 4, 0.5),
                                       (600, 36, 310, 4, 0.5), (257, 33, 100, 3, 0.5), (7, 36, 310, 4, 0.0)])
def test_object_difference_staged_weight_gradient_is_the_unstaged_one(ops, lib_option, B, N, L, G, p):
    """The weight gradient that keeps the operands of its region loop in LDS (round 6, the default) against the kernel that
    loads them inside the loop (VQA_K2_WSTAGE=0): same operands, same order of accumulation -- equal bit for bit, groups of
    one, two and three samples, regions and features that do not fill the tiles."""
    seed = 77
    vt0 = g(np.maximum(seeded.seeded_array((B, N, L), 431), 0.0))
    qt0 = g(np.abs(seeded.seeded_array((B, L), 432)))
    wt0 = g(seeded.seeded_array((G, N * L), 433, scale=1.0 / np.sqrt(N * L)))
    bt0 = g(seeded.seeded_array((G,), 434, scale=0.1))
    gl = g(seeded.seeded_array((B, N, G), 435))
    grads = []
    for staged in ("1", "0"):
        lib_option("VQA_K2_WSTAGE", staged)
        vt, qt, wt, bt = (t.clone().requires_grad_() for t in (vt0, qt0, wt0, bt0))
        ops.object_difference_attention(vt, qt, wt, bt, p, seed).backward(gl)
        grads.append((wt.grad.clone(), vt.grad.clone(), bt.grad.clone()))
    assert float(grads[0][0].abs().max()) > 0.0
    for a, b in zip(*grads):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,N,L,G,p,gate", [(3, 36, 310, 4, 0.5, False), (5, 13, 70, 2, 0.5, True), (140, 36, 310, 4, 0.5, True),
                                            (600, 36, 310, 4, 0.5, False), (257, 33, 100, 3, 0.5, True), (7, 36, 310, 4, 0.0, True)])
def test_object_difference_fused_backward_is_the_two_kernel_one(ops, lib_option, B, N, L, G, p, gate):
    """VQA_K2_FUSED=1 (one pass over the mask for the data and the weight gradient; measured slower, not the default)
    against the default pair of kernels: the weight gradient bit for bit (same operands, same order), the data gradients
    -- summed in another fixed order -- within 4 ulp-sized steps of their tensor's scale."""
    seed = 78
    vt0 = g(np.maximum(seeded.seeded_array((B, N, L), 441), 0.0))
    qt0 = g(np.abs(seeded.seeded_array((B, L), 442)))
    wt0 = g(seeded.seeded_array((G, N * L), 443, scale=1.0 / np.sqrt(N * L)))
    bt0 = g(seeded.seeded_array((G,), 444, scale=0.1))
    gl = g(seeded.seeded_array((B, N, G), 445))
    grads = []
    for fused in ("1", "0"):
        lib_option("VQA_K2_FUSED", fused)
        vt, qt, wt, bt = (t.clone().requires_grad_() for t in (vt0, qt0, wt0, bt0))
        ops.object_difference_attention(vt, qt, wt, bt, p, seed, gate_dvl=gate).backward(gl)
        grads.append((wt.grad.clone(), bt.grad.clone(), vt.grad.clone(), qt.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    for name, a, b in (("d_vl", grads[0][2], grads[1][2]), ("d_ql", grads[0][3], grads[1][3])):
        scale = float(b.abs().max())
        assert scale > 0.0
        assert float((a - b).abs().max()) <= 5e-7 * scale * np.sqrt(N), name
    if gate:
        assert float(grads[0][2][vt0 <= 0].abs().max()) == 0.0


def test_object_difference_mask_is_unbiased(ops):
    """Statistics of the counter-hash mask at the ODA shape: per-region, per-feature and per-pair keep rates."""
    B, N, L, p = 8, 36, 310, 0.5
    m = (ops.object_difference_dropout_mask(B, N, L, p, 99, dev()) > 0).float().view(B, N, N, L)
    assert abs(m.mean().item() - 0.5) < 2e-3
    for dims in [(0, 2, 3), (0, 1, 3), (0, 1, 2), (1, 2, 3)]:
        r = m.mean(dim=dims)
        assert (r - 0.5).abs().max().item() < 0.02, dims
    # neighbouring regions share a hash word (one byte each): they must still be uncorrelated
    a, b2 = m[:, 0::4].flatten(), m[:, 1::4].flatten()
    corr = ((a - a.mean()) * (b2 - b2.mean())).mean() / (a.std() * b2.std())
    assert abs(corr.item()) < 5e-3


# ----------------------------------------------------------------------------------------------- K5
@pytest.mark.parametrize("M,Kd,N,act", [(15, 8, 6, "relu"), (144, 2048, 310, "relu"), (180, 70, 130, None),
                                        (1, 2, 2, "relu"), (700, 510, 4, None), (333, 310, 2048, "relu")])
@pytest.mark.parametrize("p", [0.0, 0.5])
def test_linear_act(ops, M, Kd, N, act, p):
    seed = 4242
    x = seeded.seeded_array((M, Kd), 501)
    w = seeded.seeded_array((N, Kd), 502, scale=1.0 / np.sqrt(Kd))
    b = seeded.seeded_array((N,), 503, scale=0.1)
    gy = seeded.seeded_array((M, N), 504)
    mask = None
    if p > 0:
        mask = ops.linear_dropout_mask(M, Kd, p, seed, dev()).cpu().numpy()
        assert set(np.unique(mask)).issubset({0.0, np.float32(2.0)})
        if M * Kd > 5000:
            assert abs((mask > 0).mean() - 0.5) < 0.02
    xt, wt, bt = g(x, True), g(w, True), g(b, True)
    y = ops.linear_act(xt, wt, bt, act, p, seed)
    y_np = K.linear_act_fwd(x, w, b, act, mask)
    close("y", y, y_np)
    y.backward(g(gy))
    # use the GPU's own y for the relu mask: elements within rounding of 0 could flip between fp32 and fp64
    dx, dw, db = K.linear_act_bwd(x, w, y.detach().cpu().numpy(), gy, act, mask)
    close("d_x", xt.grad, dx)
    close("d_w", wt.grad, dw)
    close("d_b", bt.grad, db)


def test_linear_act_3d_no_bias_no_dx(ops):
    x = seeded.seeded_array((3, 36, 64), 511)
    w = seeded.seeded_array((10, 64), 512, scale=0.125)
    xt, wt = g(x), g(w, True)
    y = ops.linear_act(xt, wt, None, "relu")
    assert y.shape == (3, 36, 10)
    close("y", y.reshape(108, 10), K.linear_act_fwd(x.reshape(108, 64), w, None, "relu"))
    y.sum().backward()
    _, dw, _ = K.linear_act_bwd(x.reshape(108, 64), w, y.detach().cpu().numpy().reshape(108, 10), np.ones((108, 10)), "relu")
    close("d_w", wt.grad, dw)


@pytest.mark.parametrize("engine", ["split", "mfma"])
def test_linear_act_full_size_vs_fp64(ops, engine, monkeypatch):
    """compress_v at B=512: [18432,2048] x [2048,310] against an fp64 matmul on the GPU, on both fp32 engines (split = the default:
    csrc/gemm_f32_split.hpp; mfma = v_mfma_f32_16x16x4_f32, csrc/gemm_f32_rt.hpp)."""
    monkeypatch.setenv("VQA_F32_PRODUCTS", engine)
    gen = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(512 * 36, 2048, generator=gen).to(dev())
    w = (torch.randn(310, 2048, generator=gen) / 2048 ** 0.5).to(dev())
    b = (0.1 * torch.randn(310, generator=gen)).to(dev())
    y = ops.linear_act(x, w, b, "relu")
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    assert (y.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


# ----------------------------------------------------------------------------------------------- column sums
@pytest.mark.parametrize("M,N", [(18432, 310), (512, 2048), (18432, 4), (7, 5), (1, 1), (3000, 155), (513, 2000)])
def test_column_sum(ops, M, N):
    x = seeded.seeded_array((M, N), 311)
    out = ops.column_sum(g(x))
    close("colsum", out, x.astype(np.float64).sum(0), 2e-5)
    assert torch.equal(out, ops.column_sum(g(x))), "fixed-order reduction must be bitwise reproducible"
    xb = g(x).to(torch.bfloat16)
    close("colsum_bf16", ops.column_sum(xb), xb.float().cpu().numpy().astype(np.float64).sum(0), 2e-5)


def test_linear_fn_matches_autograd(ops):
    """ops.linear = F.linear with the bias gradient from column_sum (replay-safe)."""
    x = g(seeded.seeded_array((6, 36, 310), 321), True)
    w = g(seeded.seeded_array((40, 310), 322), True)
    b = g(seeded.seeded_array((40,), 323), True)
    gy = g(seeded.seeded_array((6, 36, 40), 324))
    ops.linear(x, w, b).backward(gy)
    got = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    torch.nn.functional.linear(x, w, b).backward(gy)
    for name, a, t in zip(("d_x", "d_w", "d_b"), got, (x, w, b)):
        close(name, a, t.grad.cpu().numpy(), 1e-5)


@pytest.mark.parametrize("rows,need_dx", [(6 * 36, True), (6 * 36, False), (128 * 36, True), (128 * 36, False)])
def test_linear_fn_relu_epilogue(ops, rows, need_dx):
    """ops.linear(..., act="relu"): relu in the GEMM epilogue; backward applies the mask either as a tensor op (small M,
    or when the data gradient needs the masked gradient) or inside the tile engine's weight-gradient kernel (tall M, no
    data gradient: compress_v) -- all against torch autograd in fp64."""
    x = g(seeded.seeded_array((rows, 70), 331), need_dx)
    w = g(seeded.seeded_array((34, 70), 332), True)
    b = g(seeded.seeded_array((34,), 333), True)
    gy = g(seeded.seeded_array((rows, 34), 334))
    y = ops.linear(x, w, b, act="relu")
    y.backward(gy)
    xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
    zd = torch.nn.functional.linear(xd, wd, bd)
    # the reference takes the relu's side from the fp32 result: a pre-activation that is zero to rounding would
    # otherwise flip one term of the weight gradient between the precisions
    keep = (y.detach() > 0).double()
    yd = zd * keep
    yd.backward(gy.double())
    close("y", y, torch.relu(zd).detach().cpu().numpy(), 1e-5)
    assert (y >= 0).all()
    if need_dx:
        close("d_x", x.grad, xd.grad.cpu().numpy(), 2e-5)
    close("d_w", w.grad, wd.grad.cpu().numpy(), 2e-5)
    close("d_b", b.grad, bd.grad.cpu().numpy(), 2e-5)


@pytest.mark.parametrize("B,R,H", [(512, 2, 510), (512, 5, 510), (3, 1, 2), (7, 3, 66)])
def test_rank_product(ops, B, R, H):
    h1, h2 = g(seeded.seeded_array((B, R, H), 371), True), g(seeded.seeded_array((B, R, H), 372), True)
    go = g(seeded.seeded_array((B, H), 373))
    out = ops.rank_product(h1, h2)
    out.backward(go)
    a, b_ = h1.detach().double().requires_grad_(), h2.detach().double().requires_grad_()
    ref = (a * b_).sum(1)
    ref.backward(go.double())
    close("out", out, ref.detach().cpu().numpy(), 2e-6)
    close("d_h1", h1.grad, a.grad.cpu().numpy(), 2e-6)
    close("d_h2", h2.grad, b_.grad.cpu().numpy(), 2e-6)


def test_with_first_group(ops):
    pooled = g(seeded.seeded_array((6, 4, 8), 361), True)
    full, first = ops.with_first_group(pooled)
    assert torch.equal(full, pooled) and torch.equal(first, pooled[:, 0])
    ((full * full).sum() + 3.0 * first.sum()).backward()
    want = 2.0 * pooled.detach()
    want[:, 0] += 3.0
    assert torch.allclose(pooled.grad, want)
    pooled.grad = None
    full, first = ops.with_first_group(pooled)
    full.sum().backward()                       # the slice unused
    assert torch.equal(pooled.grad, torch.ones_like(pooled))


def test_split_groups(ops):
    """ops.split_groups: chunk views forward, one concatenation backward (an unused chunk counts as zeros)."""
    t = g(seeded.seeded_array((4, 5, 6), 341), True)
    a, b_, c = ops.split_groups(t, (1, 1, 2))
    assert a.shape == (5, 6) and c.shape == (2, 5, 6) and a.data_ptr() == t.data_ptr()
    (a.sum() * 2.0 + (c * c).sum()).backward()
    want = torch.zeros_like(t)
    want[0] = 2.0
    want[2:4] = 2.0 * t.detach()[2:4]
    assert torch.equal(t.grad, want)
    with pytest.raises(ValueError):
        ops.split_groups(t, (1, 2))


# ----------------------------------------------------------------------------------------------- loss
@pytest.mark.parametrize("B,C", [(4, 2000), (512, 2000), (3, 3000), (1, 7), (5, 4096)])
def test_kld_sum_loss(ops, B, C):
    """train.py:536-544 against the float64 closed form; some exact zeros in the target (0 log 0 = 0)."""
    z = 3.0 * seeded.seeded_array((B, C), 301)
    t = np.exp(2.0 * seeded.seeded_array((B, C), 302))
    t[:, ::5] = 0.0
    t = (t / t.sum(axis=1, keepdims=True)).astype(np.float32)
    zt = g(z, True)
    loss = ops.kld_sum_loss(zt, g(t))
    z64, t64 = z.astype(np.float64), t.astype(np.float64)
    lse = np.log(np.exp(z64 - z64.max(1, keepdims=True)).sum(1, keepdims=True)) + z64.max(1, keepdims=True)
    logp = z64 - lse
    with np.errstate(divide="ignore", invalid="ignore"):
        want = np.where(t64 > 0, t64 * (np.log(t64) - logp), 0.0).sum()
    assert abs(loss.item() - want) <= 1e-5 * abs(want), (loss.item(), want)
    (2.5 * loss).backward()
    close("d_logits", zt.grad, 2.5 * (np.exp(logp) * t64.sum(1, keepdims=True) - t64))
    again = ops.kld_sum_loss(g(z), g(t))
    assert again.item() == loss.item(), "fixed-order reduction must be bitwise reproducible"


# ----------------------------------------------------------------------------------------------- graph replays
@pytest.mark.parametrize("B", [6, 512])
def test_backward_kernels_repeat_under_graph_replay(ops, B):
    """K1 / K3 backward zero an accumulator and add into it with float atomics.  Captured into a hipGraph, every replay
    must reproduce the eager result (a hipMemsetAsync memset node did so on the first replay only; the library now
    zero-fills with its own kernel)."""
    N, D, G = 36, 2048, 4
    gen = torch.Generator(device="cpu").manual_seed(5)
    v = torch.randn(B, N, D, generator=gen).to(dev())
    q1, q2 = torch.rand(B, D, generator=gen).to(dev()), torch.rand(B, D, generator=gen).to(dev())
    logits = torch.randn(B, N, G, generator=gen).to(dev())
    gout = torch.randn(B, N, D, generator=gen).to(dev())
    dpool = torch.randn(B, G, D, generator=gen).to(dev())

    def step():
        al = logits.clone().requires_grad_()
        alpha, pooled = ops.softmax_attention_pool(al, v)
        v2 = ops.pairwise_relation_reduce(v, q1, q2, alpha, 0, 1)
        ((v2 * gout).sum() + (pooled * dpool).sum()).backward()
        return al.grad

    want = step().clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        got = step()
    for i in range(4):
        graph.replay()
        torch.cuda.synchronize()
        err = ((got - want).abs().max() / want.abs().max()).item()
        assert err <= 1e-5, "replay %d: d_logits differs from the eager result by %.3e" % (i, err)


# ----------------------------------------------------------------------------------------------- K3a attention logits
@pytest.mark.parametrize("B,N,Kd,G", [(3, 36, 510, 4), (2, 100, 510, 4), (5, 7, 64, 1), (2, 3, 512, 8), (1, 1, 2, 2)])
@pytest.mark.parametrize("p", [0.0, 0.5])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_attention_logits(ops, B, N, Kd, G, p, dtype):
    """logits = bias + (x * keep) w^T and its backward against the oracle's linear closed form, with the kernel's own
    dropout mask (exported through vqa_linear_dropout_mask) handed to the oracle."""
    bf = dtype == "bf16"
    ld = (Kd + 63) // 64 * 64 if bf else Kd
    x = seeded.seeded_array((B, N, Kd), 331)
    if bf:
        x = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    w = seeded.seeded_array((G, Kd), 332) / np.sqrt(Kd)
    b = seeded.seeded_array((G,), 333)
    gl = seeded.seeded_array((B, N, G), 334)
    xp = np.zeros((B, N, ld), np.float32)
    xp[..., :Kd] = x
    xt = g(xp)
    if bf:
        xt = xt.to(torch.bfloat16)
    xt.requires_grad_()
    wt, bt = g(w, True), g(b, True)
    seed = 777
    out = ops.attention_logits(xt, wt, bt, p, seed)
    assert out.dtype == torch.float32 and out.shape == (B, N, G)
    M = B * N
    mask = ops.linear_dropout_mask(M, Kd, p, seed, dev()).cpu().numpy() if p else None
    x2 = x.reshape(M, Kd)
    close("logits", out.reshape(M, G), K.linear_act_fwd(x2, w, b, None, mask))
    out.backward(g(gl))
    dx, dw, db = K.linear_act_bwd(x2, w, None, gl.reshape(M, G), None, mask)
    if bf:
        got = xt.grad.float().cpu().numpy().reshape(M, ld).astype(np.float64)
        assert np.abs(got[:, :Kd] - dx).max() <= 2.0 ** -8 * np.abs(dx).max() + 1e-6
        assert ld == Kd or np.abs(got[:, Kd:]).max() == 0.0
    else:
        close("d_x", xt.grad.reshape(M, Kd), dx)
    close("d_w", wt.grad, dw)
    close("d_b", bt.grad, db)


# ----------------------------------------------------------------------------------------------- batched [B,.] layers
@pytest.mark.parametrize("group_first", [False, True])
@pytest.mark.parametrize("act", [None, "relu", "sigmoid"])
def test_batched_linear(ops, group_first, act):
    """G same-shaped layers as one batched GEMM with the fused bias/activation epilogue and activation-gradient/bias-
    gradient prologue, against per-layer torch autograd."""
    B, G, Kd, A = 37, 3, 50, 21
    x = g(seeded.seeded_array((B, G, Kd), 341), True)
    w = g(seeded.seeded_array((G, A, Kd), 342) / np.sqrt(Kd), True)
    b = g(seeded.seeded_array((G, A), 343), True)
    gy = g(seeded.seeded_array((G, B, A) if group_first else (B, G, A), 344))
    out = ops.batched_linear(x, w, b, group_first, act)
    out.backward(gy)
    got = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    f = {None: lambda z: z, "relu": torch.relu, "sigmoid": torch.sigmoid}[act]
    ref = torch.stack([f(torch.nn.functional.linear(x[:, k], w[k], b[k])) for k in range(G)], 0 if group_first else 1)
    close("out", out, ref.detach().cpu().numpy(), 1e-5)
    ref.backward(gy)
    for name, a_, t in zip(("d_x", "d_w", "d_b"), got, (x, w, b)):
        close(name, a_, t.grad.cpu().numpy(), 1e-5)


# ----------------------------------------------------------------------------------------------- K1 closed form
@pytest.mark.parametrize("B,N,D", [(3, 36, 2048), (2, 100, 512), (5, 7, 260), (1, 1, 4), (300, 36, 1024)])
@pytest.mark.parametrize("p", [0.0, 0.5])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_relation_apply(ops, B, N, D, p, dtype):
    """out = keep * (t + c2 * v) with t = q1 * sum_i alpha_i v_i, c2 = (sum_i alpha_i) q2 equals the oracle's pairwise
    relation reduce (every (i,j) term) times the kernel's own dropout mask; backward against the closed form."""
    bf = dtype == "bf16"
    v = seeded.seeded_array((B, N, D), 401)
    if bf:
        v = torch.from_numpy(v).to(torch.bfloat16).float().numpy()
    q1 = 1 / (1 + np.exp(-seeded.seeded_array((B, D), 402)))
    q2 = 1 / (1 + np.exp(-seeded.seeded_array((B, D), 403)))
    al = np.abs(seeded.seeded_array((B, N), 404)) + 0.05
    gout = seeded.seeded_array((B, N, D), 405)
    if bf:
        gout = torch.from_numpy(gout).to(torch.bfloat16).float().numpy()
    t = (q1 * np.einsum("bn,bnd->bd", al, v)).astype(np.float32)
    c2 = (al.sum(1, keepdims=True) * q2).astype(np.float32)
    seed = 4242
    mask = ops.linear_dropout_mask(B * N, D, p, seed, dev()).cpu().numpy().reshape(B, N, D).astype(np.float64) if p else 1.0
    vt = g(v).to(torch.bfloat16) if bf else g(v)
    vt.requires_grad_()
    tt, ct = g(t, True), g(c2, True)
    out = ops.relation_apply(vt, tt, ct, p, seed)
    want = K.pairwise_relation_reduce_fwd(v, q1, q2, al) * mask if not bf else \
        (t[:, None, :].astype(np.float64) + c2[:, None, :].astype(np.float64) * v) * mask
    tol = 2.0 ** -7 if bf else RTOL
    close("out", out.float(), want, tol)
    out.backward(g(gout).to(out.dtype))
    gm = gout.astype(np.float64) * mask
    close("d_t", tt.grad, gm.sum(1))
    close("d_c2", ct.grad, (gm * v).sum(1))
    close("d_v", vt.grad.float(), c2[:, None, :].astype(np.float64) * gm, tol)


@pytest.mark.parametrize("form", ["two workgroups per CU", "one workgroup per CU", "split engine"])
@pytest.mark.parametrize("p", [0.0, 0.5, 0.25])
@pytest.mark.parametrize("B,D,L", [(4, 2048, 310), (5, 256, 310), (33, 128, 34), (1, 64, 32), (8, 320, 48)])
def test_relation_projection_fused_backward(ops, B, D, L, p, form, lib_option, monkeypatch):
    """relation step + second region projection as one node (K1 -> K5): the output equals relation_apply followed by
    linear_act, and d_t / d_c2 / dW / db equal the gradients of that composition (fp64 closed form of the data gradient:
    the masked grad_x summed over the 36 regions) -- without the [B,36,D] data gradient ever being written.  Both forms of
    the data-gradient kernel (csrc/relation_dgrad.hip: the default, and VQA_RELDG_TUNE=0), and the one on the split engine
    (csrc/relation_dgrad_split.hip, VQA_F32_PRODUCTS=split, the default; the projection itself then runs there too where its shape allows)."""
    if form == "one workgroup per CU":
        lib_option("VQA_RELDG_TUNE", 0)
    monkeypatch.setenv("VQA_F32_PRODUCTS", "split" if form == "split engine" else "mfma")
    N = 36
    v = seeded.seeded_array((B, N, D), 411)
    t = seeded.seeded_array((B, D), 412)
    c2 = 1 / (1 + np.exp(-seeded.seeded_array((B, D), 413)))
    w = seeded.seeded_array((L, D), 414, scale=1.0 / np.sqrt(D))
    bias = seeded.seeded_array((L,), 415, scale=0.1)
    gy = seeded.seeded_array((B, N, L), 416)
    seed = 777
    vt = g(v)
    assert ops.relation_projection_supported(vt, g(w))
    tt, ct, wt, bt = g(t, True), g(c2, True), g(w, True), g(bias, True)
    y = ops.relation_projection(vt, tt, ct, wt, bt, p, seed)
    # the same function from the two separate ops
    t2, c2_, w2, b2 = g(t, True), g(c2, True), g(w, True), g(bias, True)
    x2 = ops.relation_apply(vt, t2, c2_, p, seed)
    y2 = ops.linear_act(x2, w2, b2, "relu", 0.0, 0)
    assert torch.equal(y, y2)
    y.backward(g(gy))
    y2.backward(g(gy))
    # fp64 closed form
    mask = ops.linear_dropout_mask(B * N, D, p, seed, dev()).cpu().numpy().reshape(B, N, D).astype(np.float64) if p else 1.0
    x = (t[:, None, :].astype(np.float64) + c2[:, None, :] * v.astype(np.float64)) * mask
    pre = x @ w.astype(np.float64).T + bias
    gz = gy * (pre > 0)
    dx = (gz @ w.astype(np.float64)) * mask
    close("d_t", tt.grad, dx.sum(1), 2e-4)
    close("d_c2", ct.grad, (dx * v).sum(1), 2e-4)
    close("d_t vs two-op path", tt.grad, t2.grad.cpu().numpy(), 2e-4)
    close("d_c2 vs two-op path", ct.grad, c2_.grad.cpu().numpy(), 2e-4)
    close("d_w", wt.grad, np.einsum("bnl,bnd->ld", gz, x), 2e-4)
    close("d_b", bt.grad, gz.sum((0, 1)), 2e-4)


@pytest.mark.parametrize("p", [0.0, 0.5, 0.25])
def test_relation_projection_pairwise_forward(ops, p):
    """The fused relation + projection node with its forward on the PAIRWISE kernel (every (i, j) term, dropout in the
    store): the same output and gradients as the closed-form forward it stands in for (t = q1 sum_i alpha_i v_i, c2 = q2
    for a softmax alpha), and the dropped relation tensor equals the oracle's pairwise reduce times the exported mask."""
    B, N, D, L, G = 260, 36, 2048, 310, 4         # B * D >= 2^19: the in-register pairwise kernel
    v = seeded.seeded_array((B, N, D), 421)
    q1 = 1 / (1 + np.exp(-seeded.seeded_array((B, D), 422)))
    q2 = 1 / (1 + np.exp(-seeded.seeded_array((B, D), 423)))
    logits = seeded.seeded_array((B, N, G), 424)
    w = seeded.seeded_array((L, D), 425, scale=1.0 / np.sqrt(D))
    bias = seeded.seeded_array((L,), 426, scale=0.1)
    gy = seeded.seeded_array((B, N, L), 427)
    seed = 31337
    vt = g(v)
    alpha = torch.softmax(g(logits), dim=1)
    s0 = torch.einsum("bn,bnd->bd", alpha[:, :, 0], vt)
    assert ops.pairwise_projection_supported(vt) and ops.relation_projection_supported(vt, g(w))
    outs = []
    for pairwise in (False, True):
        q1t, q2t, wt, bt = g(q1, True), g(q2, True), g(w, True), g(bias, True)
        s_ = s0.clone().requires_grad_()
        y = ops.relation_projection(vt, q1t * s_, q2t, wt, bt, p, seed, False, (q1t, q2t, alpha, 0) if pairwise else None)
        y.backward(g(gy))
        outs.append((y.detach(), q1t.grad, q2t.grad, s_.grad, wt.grad, bt.grad))
    # y agrees to rounding.  The gradients agree up to the relu gates that the two forwards' last-bit differences flip
    # (pre-activations within 1e-7 of zero; the same effect as any reordering of the forward GEMM): direction and size.
    close("y (pairwise forward vs closed form)", outs[1][0], outs[0][0].cpu().numpy(), 5e-5)
    for name, a, b in zip(("d_q1", "d_q2", "d_s", "d_w", "d_b"), outs[0][1:], outs[1][1:]):
        a64, b64 = a.double().flatten(), b.double().flatten()
        cos = float((a64 @ b64) / (a64.norm() * b64.norm()))
        assert cos > 0.99999, (name, cos)
        close(name + " (pairwise forward vs closed form)", b, a.cpu().numpy(), 6e-2)   # (one flipped gate moves an element by ~1-2 %)
    # the relation tensor itself: pairwise kernel with dropout in the store == oracle pairwise reduce x exported mask
    x = torch.empty_like(vt)
    from vqa_playground_pytorch_amd import _lib
    import ctypes
    a_ptr = ctypes.c_void_p(alpha.data_ptr())
    sv, sp = ops._seed_args(seed)
    q1d, q2d = g(q1), g(q2)                        # (kept alive across the asynchronous launch)
    _lib.check(_lib.lib().vqa_pairwise_relation_reduce_drop_fwd(ops._p(vt), ops._p(q1d), ops._p(q2d), a_ptr, G, ops._p(x), p, sv,
                                                               sp, B, N, D, ops._stream()), "pairwise_drop_fwd")
    torch.cuda.synchronize()
    mask = ops.linear_dropout_mask(B * N, D, p, seed, dev()).cpu().numpy().reshape(B, N, D) if p else 1.0
    want = K.pairwise_relation_reduce_fwd(v[:8], q1[:8], q2[:8], alpha[:8, :, 0].cpu().numpy().astype(np.float64))
    close("relation tensor", x[:8], want * (mask[:8] if p else 1.0))
