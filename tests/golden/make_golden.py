#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the authoring container (needs /root/reference).  It imports the
reference's ``config/CoR2.py`` / ``config/ODA.py`` / ``putils`` with the missing
third-party packages mocked, swaps the question encoder for an
identity so the 2400-d question vector is an input, overwrites every parameter
from ``oracle.seeded`` and records OUTPUTS only (inputs/params are regenerated
from seeds by the tests).  Nothing of the reference's source text is stored.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
    python tests/golden/make_golden.py --out /tmp/g --only blocks,encoder   # elsewhere / a subset
"""
import contextlib
import importlib
import io
import os
import sys
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import seeded  # noqa: E402

REF = "/root/reference"


def _load_by_path(name, path):
    """Import one of the reference's files under a private module name.  The repo has its own ``config`` package (the
    drop-in modules), so ``import config.CoR2`` would resolve to the REPO's model; the reference's files are therefore
    loaded by path (their own ``from putils import *`` still resolves through sys.path to /root/reference/putils)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    for m in ["deepdish", "h5py", "nltk", "nltk.corpus", "nltk.parse", "nltk.parse.stanford", "yagmail",
              "munch", "configobj", "passlib", "passlib.hash", "tables", "torchvision",
              "torchvision.transforms", "torchvision.models", "lda", "redis", "spacy", "cv2"]:
        try:
            importlib.import_module(m)
        except Exception:
            sys.modules[m] = mock.MagicMock(name=m)
    with contextlib.redirect_stdout(io.StringIO()):
        cor = _load_by_path("_reference_config_CoR2", os.path.join(REF, "config", "CoR2.py"))
        oda = _load_by_path("_reference_config_ODA", os.path.join(REF, "config", "ODA.py"))
        putils = importlib.import_module("putils")
    assert os.path.realpath(putils.__file__).startswith(REF), putils.__file__
    assert os.path.realpath(cor.__file__).startswith(REF) and os.path.realpath(oda.__file__).startswith(REF)

    class Identity(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            return x

    cor.SkipThoughts = oda.SkipThoughts = Identity
    return cor, oda, putils


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def grad_digest(model, out, prefix="g."):
    """Per-parameter gradient record: Frobenius norm, an 8x8 corner, and the whole
    tensor when it is small (<= 8192 elements)."""
    for name, p in model.named_parameters():
        g = p.grad.detach().numpy()
        out[prefix + name + ".norm"] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        g2 = g.reshape(g.shape[0], -1)
        out[prefix + name + ".corner"] = g2[:8, :8].copy()
        if g.size <= 8192:
            out[prefix + name + ".full"] = g.copy()


def kld_sum(logits, target):
    # train.py:536-544 -- KLDivLoss(size_average=False)(log_softmax(x), a)
    return torch.nn.KLDivLoss(reduction="sum")(torch.nn.functional.log_softmax(logits, dim=1), target)


def blocks(cor, oda, putils, out_dir=HERE):
    out = {}
    x1 = t(seeded.seeded_array((3, 5, 8), 11))
    x2 = t(seeded.seeded_array((3, 8), 12))
    out["bmul.out"] = putils.bmul(x1, x2).numpy()
    x4 = t(seeded.seeded_array((3, 5, 5, 8), 13))
    out["bmul4.out"] = putils.bmul(x4, x2).numpy()
    a = t(seeded.seeded_array((3, 2, 5), 14))
    b = t(seeded.seeded_array((3, 5, 12), 15))
    out["bmatmul.out"] = putils.bmatmul(a, b).numpy()

    # MutanFusion(8, 6, 16, R=2): 3-D x1 against 2-D x2, forward + all grads
    mf = seeded.load_state(putils.MutanFusion(8, 6, 16, 2), 21)
    y1 = t(seeded.seeded_array((3, 5, 8), 22)).requires_grad_()
    y2 = t(seeded.seeded_array((3, 6), 23)).requires_grad_()
    go = t(seeded.seeded_array((3, 5, 16), 24))
    o = mf(y1, y2)
    (o * go).sum().backward()
    out["mutan3d.out"] = o.detach().numpy()
    out["mutan3d.dx1"] = y1.grad.numpy()
    out["mutan3d.dx2"] = y2.grad.numpy()
    grad_digest(mf, out, "mutan3d.g.")

    # MutanFusion(8, 6, 16, R=3): 2-D against 2-D (the fusion_final shape class)
    mf2 = seeded.load_state(putils.MutanFusion(8, 6, 16, 3), 25)
    z1 = t(seeded.seeded_array((3, 8), 26)).requires_grad_()
    z2 = t(seeded.seeded_array((3, 6), 27)).requires_grad_()
    go2 = t(seeded.seeded_array((3, 16), 28))
    o2 = mf2(z1, z2)
    (o2 * go2).sum().backward()
    out["mutan2d.out"] = o2.detach().numpy()
    out["mutan2d.dx1"] = z1.grad.numpy()
    out["mutan2d.dx2"] = z2.grad.numpy()
    grad_digest(mf2, out, "mutan2d.g.")

    # MyConv1d softmax over regions (dim=1), eval mode
    cv = seeded.load_state(cor.MyConv1d(16, 2, 1, 1, p=0.5, af="softmax", dim=1), 31).eval()
    f = t(seeded.seeded_array((3, 5, 16), 32))
    out["conv_softmax.out"] = cv(f).detach().numpy()
    cr = seeded.load_state(cor.MyConv1d(16, 7, 1, 1, p=0.5, af="relu"), 33).eval()
    out["conv_relu.out"] = cr(f).detach().numpy()
    ml = seeded.load_state(cor.MyLinear(16, 7, p=0.5, af="sigmoid"), 34).eval()
    out["linear_sigmoid.out"] = ml(f).detach().numpy()

    # MyATT(fuse 16, G=2, inputs 12, att 8, relu), eval mode, forward + grads
    att = seeded.load_state(cor.MyATT(16, 2, 12, 8, af="relu"), 41).eval()
    inp = t(seeded.seeded_array((3, 5, 12), 42)).requires_grad_()
    fu = t(seeded.seeded_array((3, 5, 16), 43)).requires_grad_()
    xv, latt = att(inp, fu)
    alpha = torch.cat(latt, dim=2)
    gxv = t(seeded.seeded_array((3, 8), 44))
    galpha = t(seeded.seeded_array((3, 5, 2), 45))
    ((xv * gxv).sum() + (alpha * galpha).sum()).backward()
    out["att.x_v"] = xv.detach().numpy()
    out["att.alpha"] = alpha.detach().numpy()
    out["att.dinputs"] = inp.grad.numpy()
    out["att.dfuse"] = fu.grad.numpy()
    grad_digest(att, out, "att.g.")

    # decare_cat + alpha-weighted reduce (CoR2.py:191-199, :216) at small dims
    class Holder:
        pass

    h = Holder()
    h.compress_q_1 = seeded.load_state(cor.MyLinear(6, 4, p=0.5, af="relu"), 51).eval()
    h.expand_q_1 = seeded.load_state(cor.MyLinear(4, 12, p=0.5, af="sigmoid"), 52).eval()
    h.compress_q_2 = seeded.load_state(cor.MyLinear(6, 4, p=0.5, af="relu"), 53).eval()
    h.expand_q_2 = seeded.load_state(cor.MyLinear(4, 12, p=0.5, af="sigmoid"), 54).eval()
    vv = t(seeded.seeded_array((3, 5, 12), 55))
    qq = t(seeded.seeded_array((3, 6), 56))
    al = torch.softmax(t(seeded.seeded_array((3, 5, 1), 57)), dim=1)
    cat = cor.Model.decare_cat(h, vv, vv, qq)
    out["decare.cat"] = cat.detach().numpy()
    out["decare.q1"] = h.expand_q_1(h.compress_q_1(qq)).detach().numpy()
    out["decare.q2"] = h.expand_q_2(h.compress_q_2(qq)).detach().numpy()
    out["decare.v2"] = (al.contiguous().view(3, 5, 1, 1) * cat).sum(1).detach().numpy()
    np.savez_compressed(os.path.join(out_dir, "blocks.npz"), **out)
    print("blocks.npz:", len(out), "arrays")


def full_model(mod, name, nans, seed_w, seed_in, feature_key, out_dir=HERE):
    model = seeded.load_state(mod.Model(["PAD", "UNK"], nans), seed_w).eval()  # eval: dropout off, grads on
    v, q, a = seeded.seeded_inputs(4, answers=nans, seed=seed_in)
    vt, qt, at = t(v), t(q).requires_grad_(), t(a)
    caps = {}

    def hook(key):
        def fn(_m, _i, o):
            caps[key] = o
        return fn

    hooks = []
    for key in ["compress_v", "compress_q", "fusion_vq1", "fusion_vq2", "compress_v2", "fusion_final",
                "att1", "att2", "att", "linear_q"]:
        if hasattr(model, key):
            hooks.append(getattr(model, key).register_forward_hook(hook(key)))
    if hasattr(model, "compress_v2"):  # input of compress_v2 = the K1 output v2 [B,36,2048] (CoR2.py:216-218)
        hooks.append(model.compress_v2.register_forward_pre_hook(
            lambda _m, i: caps.__setitem__("v2_feature", i[0])))
    logits = model({"v": vt, "q_idxes": qt})
    loss = kld_sum(logits, at)
    loss.backward()
    out = {"logits": logits.detach().numpy(), "loss": np.float64(loss.item()), "dq": qt.grad.numpy()}
    for key, val in caps.items():
        if isinstance(val, tuple):  # MyATT -> (x_v, tuple of G [B,N,1])
            out[key + ".x_v"] = val[0].detach().numpy()
            out[key + ".alpha"] = torch.cat(val[1], dim=2).detach().numpy()
        else:
            out[key] = val.detach().numpy()
    ad = model.alpha_dict
    for k, val in ad.items():
        if isinstance(val, (tuple, list)):
            out["alpha_dict." + k] = torch.cat(list(val), dim=2).detach().numpy()
        else:
            out["alpha_dict." + k] = val.detach().numpy()
    grad_digest(model, out)
    for h in hooks:
        h.remove()

    # 3 optimisation steps with the train.py:41-107 / :286-299 step semantics, dropout off:
    # scheduler.step() BEFORE optimizer.step(), KLD-sum loss, clip_grad_norm_(0.25), Adam(lr).
    model = seeded.load_state(mod.Model(["PAD", "UNK"], nans), seed_w).eval()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=mod.lr)
    sch = torch.optim.lr_scheduler.ExponentialLR(opt, 0.5 ** (1 / 50000))
    losses, gnorms = [], []
    import warnings
    for step in range(3):
        v, q, a = seeded.seeded_inputs(4, answers=nans, seed=seed_in + 100 + step)
        lg = model({"v": t(v), "q_idxes": t(q)})
        ls = kld_sum(lg, t(a))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sch.step()
        opt.zero_grad()
        ls.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25)
        opt.step()
        losses.append(ls.item())
        gnorms.append(float(gn))
    out["train3.loss"] = np.array(losses, np.float64)
    out["train3.gnorm"] = np.array(gnorms, np.float64)
    out["train3.lr"] = np.float64(opt.param_groups[0]["lr"])
    for pname, p in model.named_parameters():
        w = p.detach().numpy().astype(np.float64)
        out["train3.w." + pname + ".sum"] = np.float64(w.sum())
        out["train3.w." + pname + ".norm"] = np.float64(np.sqrt((w ** 2).sum()))
    np.savez_compressed(os.path.join(out_dir, name + "_b4.npz"), **out)
    size = os.path.getsize(os.path.join(out_dir, name + "_b4.npz"))
    print(name + "_b4.npz:", len(out), "arrays,", size // 1024, "KiB, loss", out["loss"])


def encoder(putils, out_dir=HERE):
    """BayesianGRU(620, 2400, af='relu') + padded embedding, the pieces SkipThoughts assembles (its constructor
    downloads weight files, so the pieces are built directly): eval-mode outputs + one gradient digest."""
    out = {}
    vocab, T, B = 9, 6, 4
    emb = torch.nn.Embedding(vocab, 620, padding_idx=0)
    gru = putils.BayesianGRU(input_size=620, hidden_size=2400, dropout=0.25, return_last=True, af="relu")
    holder = torch.nn.Module()
    holder.embedding, holder.gru = emb, gru
    seeded.load_state(holder, 91)
    holder.eval()
    idx = torch.tensor([[3, 1, 4, 1, 5, 2], [2, 7, 0, 0, 0, 0], [8, 0, 0, 0, 0, 0], [1, 2, 3, 4, 0, 0]])
    x = emb(idx)
    lengths = (idx.size(1) - idx.eq(0).sum(1)).long()
    y = gru(x, lengths)
    (y * t(seeded.seeded_array((B, 2400), 92))).sum().backward()
    out["q"] = y.detach().numpy()
    out["all_hiddens_norm"] = np.float64(gru.all_hiddens.double().norm().item())
    out["g.embedding"] = emb.weight.grad.numpy()
    out["g.weight_hn.norm"] = np.float64(gru.gru_cell.weight_hn.weight.grad.double().norm().item())
    out["g.weight_ir.bias"] = gru.gru_cell.weight_ir.bias.grad.numpy()
    np.savez_compressed(os.path.join(out_dir, "encoder.npz"), **out)
    print("encoder.npz:", len(out), "arrays")


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE, help="directory the .npz files are written to (default: this directory)")
    ap.add_argument("--only", default="", help="comma list out of blocks,encoder,cor2,oda (default: all)")
    args = ap.parse_args(argv)
    only = set(filter(None, args.only.split(",")))
    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cor, oda, putils = import_reference()
    if not only or "blocks" in only:
        blocks(cor, oda, putils, args.out)
    if not only or "encoder" in only:
        encoder(putils, args.out)
    if not only or "cor2" in only:
        full_model(cor, "cor2", 2000, seed_w=0, seed_in=1, feature_key="feature", out_dir=args.out)
    if not only or "oda" in only:
        full_model(oda, "oda", 3000, seed_w=0, seed_in=1, feature_key=None, out_dir=args.out)


if __name__ == "__main__":
    main()
