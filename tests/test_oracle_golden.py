"""Pin the oracle (oracle/) against the golden vectors produced by the imported reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import kernels_np as K
from oracle import reference_faithful as RF
from oracle import seeded

RTOL = 1e-5  # oracle-vs-reference bar; the HIP path's bar is 1e-3


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, rtol=RTOL, atol=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max()
    assert err <= atol + rtol * scale, "max abs err %.3e vs scale %.3e" % (err, scale)


@pytest.fixture(scope="module")
def blocks(golden_dir):
    return np.load(os.path.join(golden_dir, "blocks.npz"))


def check_grads(module, gold, prefix):
    for name, p in module.named_parameters():
        g = p.grad.numpy()
        close(np.sqrt((g.astype(np.float64) ** 2).sum()), gold[prefix + name + ".norm"], rtol=1e-4)
        close(g.reshape(g.shape[0], -1)[:8, :8], gold[prefix + name + ".corner"], rtol=1e-4, atol=1e-6)
        if prefix + name + ".full" in gold:
            close(g, gold[prefix + name + ".full"], rtol=1e-4, atol=1e-6)


def test_per_sample_ops(blocks):
    x1, x2 = t(seeded.seeded_array((3, 5, 8), 11)), t(seeded.seeded_array((3, 8), 12))
    close(RF.per_sample_mul(x1, x2).numpy(), blocks["bmul.out"])
    x4 = t(seeded.seeded_array((3, 5, 5, 8), 13))
    close(RF.per_sample_mul(x4, x2).numpy(), blocks["bmul4.out"])
    a, b = t(seeded.seeded_array((3, 2, 5), 14)), t(seeded.seeded_array((3, 5, 12), 15))
    close(RF.per_sample_matmul(a, b).numpy(), blocks["bmatmul.out"])


def test_lowrank_bilinear_3d(blocks):
    mf = seeded.load_state(RF.LowRankBilinear(8, 6, 16, 2), 21)
    y1 = t(seeded.seeded_array((3, 5, 8), 22)).requires_grad_()
    y2 = t(seeded.seeded_array((3, 6), 23)).requires_grad_()
    go = t(seeded.seeded_array((3, 5, 16), 24))
    o = mf(y1, y2)
    (o * go).sum().backward()
    close(o.detach().numpy(), blocks["mutan3d.out"])
    close(y1.grad.numpy(), blocks["mutan3d.dx1"])
    close(y2.grad.numpy(), blocks["mutan3d.dx2"])
    check_grads(mf, blocks, "mutan3d.g.")
    # numpy closed form (forward + hand-derived backward) against the same goldens
    sd = {k: v.detach().numpy() for k, v in mf.state_dict().items()}
    w1 = np.stack([sd["list_linear1.%d.linear.weight" % r] for r in range(2)])
    b1 = np.stack([sd["list_linear1.%d.linear.bias" % r] for r in range(2)])
    w2 = np.stack([sd["list_linear2.%d.linear.weight" % r] for r in range(2)])
    b2 = np.stack([sd["list_linear2.%d.linear.bias" % r] for r in range(2)])
    x2 = y2.detach().numpy().astype(np.float64)
    h2 = np.einsum("bl,rhl->brh", x2, w2) + b2[None]
    out, _ = K.lowrank_bilinear_fusion_fwd(y1.detach().numpy(), w1, b1, h2)
    close(out, blocks["mutan3d.out"])
    dx, dw1, db1, dh2 = K.lowrank_bilinear_fusion_bwd(y1.detach().numpy(), w1, b1, h2, go.numpy())
    close(dx, blocks["mutan3d.dx1"])
    for r in range(2):
        close(dw1[r], blocks["mutan3d.g.list_linear1.%d.linear.weight.full" % r], rtol=1e-4)
        close(db1[r], blocks["mutan3d.g.list_linear1.%d.linear.bias.full" % r], rtol=1e-4)
        close(dh2[:, r].sum(0), blocks["mutan3d.g.list_linear2.%d.linear.bias.full" % r], rtol=1e-4)
    close(np.einsum("brh,rhl->bl", dh2, w2), blocks["mutan3d.dx2"])


def test_lowrank_bilinear_2d(blocks):
    mf = seeded.load_state(RF.LowRankBilinear(8, 6, 16, 3), 25)
    z1 = t(seeded.seeded_array((3, 8), 26)).requires_grad_()
    z2 = t(seeded.seeded_array((3, 6), 27)).requires_grad_()
    go = t(seeded.seeded_array((3, 16), 28))
    o = mf(z1, z2)
    (o * go).sum().backward()
    close(o.detach().numpy(), blocks["mutan2d.out"])
    close(z1.grad.numpy(), blocks["mutan2d.dx1"])
    close(z2.grad.numpy(), blocks["mutan2d.dx2"])
    check_grads(mf, blocks, "mutan2d.g.")


def test_drop_layers_eval(blocks):
    f = t(seeded.seeded_array((3, 5, 16), 32))
    cv = seeded.load_state(RF.DropConv1x1(16, 2, p=0.5, af="softmax", dim=1), 31).eval()
    close(cv(f).detach().numpy(), blocks["conv_softmax.out"])
    cr = seeded.load_state(RF.DropConv1x1(16, 7, p=0.5, af="relu"), 33).eval()
    close(cr(f).detach().numpy(), blocks["conv_relu.out"])
    ml = seeded.load_state(RF.DropLinear(16, 7, p=0.5, af="sigmoid"), 34).eval()
    close(ml(f).detach().numpy(), blocks["linear_sigmoid.out"])


def test_shape_errors():
    with pytest.raises(ValueError):
        RF.DropLinear(16, 7)(torch.zeros(2, 15))
    with pytest.raises(ValueError):
        RF.DropConv1x1(16, 7)(torch.zeros(2, 16))
    with pytest.raises(ValueError):
        RF.CheckedLinear(16, 7)(torch.zeros(2, 3, 15))
    with pytest.raises(AssertionError):
        RF.GlimpseAttention(16, 3, 12, 8)


def test_glimpse_attention(blocks):
    att = seeded.load_state(RF.GlimpseAttention(16, 2, 12, 8, af="relu"), 41).eval()
    inp = t(seeded.seeded_array((3, 5, 12), 42)).requires_grad_()
    fu = t(seeded.seeded_array((3, 5, 16), 43)).requires_grad_()
    xv, latt = att(inp, fu)
    alpha = torch.cat(latt, dim=2)
    ((xv * t(seeded.seeded_array((3, 8), 44))).sum() + (alpha * t(seeded.seeded_array((3, 5, 2), 45))).sum()).backward()
    close(xv.detach().numpy(), blocks["att.x_v"])
    close(alpha.detach().numpy(), blocks["att.alpha"])
    close(inp.grad.numpy(), blocks["att.dinputs"])
    close(fu.grad.numpy(), blocks["att.dfuse"])
    check_grads(att, blocks, "att.g.")
    # K3 closed form: logits -> alpha, pooled; backward to dlogits, dv
    sd = {k: v.detach().numpy().astype(np.float64) for k, v in att.state_dict().items()}
    wa, ba = sd["conv_att.conv.weight"][:, :, 0], sd["conv_att.conv.bias"]
    logits = fu.detach().numpy().astype(np.float64) @ wa.T + ba
    a_np, pooled = K.softmax_attention_pool_fwd(logits, inp.detach().numpy())
    close(a_np, blocks["att.alpha"])
    # chain by hand through the per-glimpse relu-linears to get dpooled, then K3 backward
    gxv = seeded.seeded_array((3, 8), 44).astype(np.float64)
    dpooled = np.zeros_like(pooled)
    for g in range(2):
        w, b = sd["list_linear_v_fusion.%d.linear.weight" % g], sd["list_linear_v_fusion.%d.linear.bias" % g]
        pre = pooled[:, g] @ w.T + b
        dpooled[:, g] = ((pre > 0) * gxv[:, 4 * g:4 * g + 4]) @ w
    dlogits, dv = K.softmax_attention_pool_bwd(a_np, inp.detach().numpy(), dpooled,
                                               seeded.seeded_array((3, 5, 2), 45))
    close(dv, blocks["att.dinputs"])
    close(dlogits @ wa, blocks["att.dfuse"])


def test_pairwise_relation(blocks):
    class H(torch.nn.Module):
        pass

    h = RF.CoR2Oracle.__new__(RF.CoR2Oracle)
    torch.nn.Module.__init__(h)
    h.compress_q_1 = seeded.load_state(RF.DropLinear(6, 4, p=0.5, af="relu"), 51).eval()
    h.expand_q_1 = seeded.load_state(RF.DropLinear(4, 12, p=0.5, af="sigmoid"), 52).eval()
    h.compress_q_2 = seeded.load_state(RF.DropLinear(6, 4, p=0.5, af="relu"), 53).eval()
    h.expand_q_2 = seeded.load_state(RF.DropLinear(4, 12, p=0.5, af="sigmoid"), 54).eval()
    vv, qq = t(seeded.seeded_array((3, 5, 12), 55)), t(seeded.seeded_array((3, 6), 56))
    al = torch.softmax(t(seeded.seeded_array((3, 5, 1), 57)), dim=1)
    cat = h.pairwise_tensor(vv, qq)
    close(cat.detach().numpy(), blocks["decare.cat"])
    v2 = (al.view(3, 5, 1, 1) * cat).sum(1)
    close(v2.detach().numpy(), blocks["decare.v2"])
    v2_np = K.pairwise_relation_reduce_fwd(vv.numpy(), blocks["decare.q1"], blocks["decare.q2"], al[:, :, 0].numpy())
    close(v2_np, blocks["decare.v2"])


def test_pairwise_relation_backward_vs_autograd():
    B, N, D = 2, 5, 12
    v = t(seeded.seeded_array((B, N, D), 61)).double().requires_grad_()
    q1 = torch.sigmoid(t(seeded.seeded_array((B, D), 62))).double().requires_grad_()
    q2 = torch.sigmoid(t(seeded.seeded_array((B, D), 63))).double().requires_grad_()
    al = (0.3 + torch.softmax(t(seeded.seeded_array((B, N), 64)), dim=1)).double().requires_grad_()  # sum != 1 on purpose
    g = t(seeded.seeded_array((B, N, D), 65)).double()
    cat = v.view(B, N, 1, D) * q1.view(B, 1, 1, D) + v.view(B, 1, N, D) * q2.view(B, 1, 1, D)
    v2 = (al.view(B, N, 1, 1) * cat).sum(1)
    (v2 * g).sum().backward()
    close(K.pairwise_relation_reduce_fwd(v.detach().numpy(), q1.detach().numpy(), q2.detach().numpy(), al.detach().numpy()),
          v2.detach().numpy(), rtol=1e-12, atol=1e-12)
    da, dq1, dq2, dv = K.pairwise_relation_reduce_bwd(v.detach().numpy(), q1.detach().numpy(), q2.detach().numpy(),
                                                     al.detach().numpy(), g.numpy())
    close(da, al.grad.numpy(), rtol=1e-12, atol=1e-12)
    close(dq1, q1.grad.numpy(), rtol=1e-12, atol=1e-12)
    close(dq2, q2.grad.numpy(), rtol=1e-12, atol=1e-12)
    close(dv, v.grad.numpy(), rtol=1e-12, atol=1e-12)


def test_object_difference_vs_autograd_with_mask():
    B, N, L, G = 2, 4, 6, 3
    vl = t(seeded.seeded_array((B, N, L), 71)).double().requires_grad_()
    ql = t(seeded.seeded_array((B, L), 72)).double().requires_grad_()
    w = t(seeded.seeded_array((G, N * L), 73)).double().requires_grad_()
    bias = t(seeded.seeded_array((G,), 74)).double().requires_grad_()
    mask = t((np.random.RandomState(75).rand(B, N, N * L) > 0.5).astype(np.float64) * 2.0)
    gl = t(seeded.seeded_array((B, N, G), 76)).double()
    vq = RF.ODAOracle.difference_tensor(vl, ql) * mask
    logits = vq @ w.t() + bias
    (logits * gl).sum().backward()
    close(K.object_difference_logits_fwd(vl.detach().numpy(), ql.detach().numpy(), w.detach().numpy(),
                                         bias.detach().numpy(), mask.numpy()), logits.detach().numpy(), 1e-12, 1e-12)
    dvl, dql, dw, db = K.object_difference_logits_bwd(vl.detach().numpy(), ql.detach().numpy(), w.detach().numpy(),
                                                      gl.numpy(), mask.numpy())
    close(dvl, vl.grad.numpy(), 1e-12, 1e-12)
    close(dql, ql.grad.numpy(), 1e-12, 1e-12)
    close(dw, w.grad.numpy(), 1e-12, 1e-12)
    close(db, bias.grad.numpy(), 1e-12, 1e-12)


def _full(model_cls, fname, nans, golden_dir):
    gold = np.load(os.path.join(golden_dir, fname))
    model = seeded.load_state(model_cls(nans), 0).eval()
    v, q, a = seeded.seeded_inputs(4, answers=nans, seed=1)
    qt = t(q).requires_grad_()
    logits = model({"v": t(v), "q": qt})
    loss = RF.kld_sum_loss(logits, t(a))
    loss.backward()
    close(logits.detach().numpy(), gold["logits"])
    close(loss.item(), gold["loss"])
    close(qt.grad.numpy(), gold["dq"], rtol=1e-4)
    check_grads(model, gold, "g.")
    return model, gold


def test_cor2_full_b4(golden_dir):
    model, gold = _full(RF.CoR2Oracle, "cor2_b4.npz", 2000, golden_dir)
    for k in ["v2_feature", "fusion_vq1", "fusion_vq2", "compress_v", "att1.x_v", "att2.x_v"]:
        close(model.taps[k].detach().numpy(), gold[k])
    close(torch.cat(model.alpha_dict["alpha1"], 2).detach().numpy(), gold["alpha_dict.alpha1"])
    close(torch.cat(model.alpha_dict["alpha2"], 2).detach().numpy(), gold["alpha_dict.alpha2"])
    close(model.alpha_dict["feature"].detach().numpy(), gold["alpha_dict.feature"])
    assert model.alpha_dict["feature"].shape == (4, 2, 2048)
    assert sum(p.numel() for p in model.parameters()) == 11940244


def test_oda_full_b4(golden_dir):
    model, gold = _full(RF.ODAOracle, "oda_b4.npz", 3000, golden_dir)
    close(model.alpha_dict["alphas"].detach().numpy(), gold["alpha_dict.alphas"])
    close(model.taps["att.x_v"].detach().numpy(), gold["att.x_v"])
    assert sum(p.numel() for p in model.parameters()) == 7348434


@pytest.mark.parametrize("cls,fname,nans", [(RF.CoR2Oracle, "cor2_b4.npz", 2000), (RF.ODAOracle, "oda_b4.npz", 3000)])
def test_train3_trajectory(cls, fname, nans, golden_dir):
    gold = np.load(os.path.join(golden_dir, fname))
    model = seeded.load_state(cls(nans), 0).eval()
    batches = []
    for step in range(3):
        v, q, a = seeded.seeded_inputs(4, answers=nans, seed=1 + 100 + step)
        batches.append((t(v), t(q), t(a)))
    losses, norms, lr = RF.train_steps(model, batches)
    close(np.array(losses), gold["train3.loss"], rtol=1e-5)
    close(np.array(norms), gold["train3.gnorm"], rtol=1e-4)
    close(lr, gold["train3.lr"], rtol=1e-12)
    for name, p in model.named_parameters():
        w = p.detach().numpy().astype(np.float64)
        close(np.sqrt((w ** 2).sum()), gold["train3.w." + name + ".norm"], rtol=1e-6)
        close(w.sum(), gold["train3.w." + name + ".sum"], rtol=1e-5, atol=1e-4)


def test_b1_works():
    """The reference raises IndexError at B=1 (squeeze drops the batch axis); the restatement must not."""
    model = seeded.load_state(RF.CoR2Oracle(50, feat=32, qdim=24, low=10, hidden=12, glimpses=2, att_dim=8), 3).eval()
    v, q, _ = seeded.seeded_inputs(1, regions=5, feat=32, qdim=24, answers=50, seed=4)
    assert model({"v": t(v), "q": t(q)}).shape == (1, 50)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference only exists in the authoring container")
def test_golden_recipe_regenerates_committed_fixtures(golden_dir, tmp_path):
    """tests/golden/make_golden.py -- the committed recipe -- still runs next to the repo's own ``config`` package (it
    loads the reference's config/CoR2.py / config/ODA.py BY PATH) and reproduces the committed building-block and
    encoder fixtures bit for bit.  Runs in a child process: the recipe mocks third-party modules in sys.modules."""
    import subprocess
    import sys
    out = str(tmp_path)
    script = os.path.join(golden_dir, "make_golden.py")
    r = subprocess.run([sys.executable, script, "--out", out, "--only", "blocks,encoder"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for name in ("blocks.npz", "encoder.npz"):
        new, old = np.load(os.path.join(out, name)), np.load(os.path.join(golden_dir, name))
        assert sorted(new.files) == sorted(old.files)
        for k in new.files:
            assert np.array_equal(new[k], old[k]), (name, k)


def test_mixed_precision_oracle_is_anchored_to_the_pinned_one():
    """oracle/mixed_precision.py (the checker of the bf16 configuration) with its roundings switched off computes the same
    function as the reference-faithful restatement above -- logits, attention maps and every parameter gradient, float64,
    1e-9 -- and with them on stays within bf16 distance of it.  This is what ties the bf16 GPU tests to the reference."""
    import torch
    from oracle import mixed_precision as MP
    v, q, a = (torch.from_numpy(x).double() for x in seeded.seeded_inputs(3, answers=120, seed=9))
    ref = seeded.load_state(RF.CoR2Oracle(120), 0).eval().double()
    exact = seeded.load_state(MP.CoR2MixedOracle(120, rounding=False), 0).eval().double()
    rounded = seeded.load_state(MP.CoR2MixedOracle(120, rounding=True), 0).eval().double()
    out = {}
    for name, m in (("ref", ref), ("exact", exact), ("rounded", rounded)):
        lo = m({"v": v, "q": q})
        RF.kld_sum_loss(lo, a).backward()
        out[name] = (lo.detach(), {n: p.grad.clone() for n, p in m.named_parameters()},
                     torch.cat(m.alpha_dict["alpha2"], 2).detach())

    def rel(x, y):
        return float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))

    assert rel(out["exact"][0], out["ref"][0]) < 1e-9 and rel(out["exact"][2], out["ref"][2]) < 1e-9
    for n, g in out["ref"][1].items():
        assert float((out["exact"][1][n] - g).abs().max()) <= 1e-9 * float(g.abs().max()) + 1e-12, n
    assert 1e-6 < rel(out["rounded"][0], out["ref"][0]) < 3e-2        # really rounds, and only by bf16-sized amounts
