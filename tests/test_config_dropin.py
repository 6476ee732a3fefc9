"""The repo-root ``config.CoR2`` / ``config.ODA`` modules expose the reference's module surface
(train.py does importlib.import_module(args.cf) and reads these attributes)."""
import importlib

import pytest

ATTRS = ["data_dir", "process_dir", "log_dir", "analyze_dir", "method_name", "version", "samplingans",
         "loss_metric", "vgenome", "version1_multiple_choices", "arch", "size", "nans", "splitnum", "mwc", "mql",
         "target_list", "epochs", "resume", "print_freq", "lr", "load_mem",
         "batch_size", "clip_grad", "test_dev_range", "test_range", "debug", "Model", "MyConv1d", "MyLinear", "MyATT"]


@pytest.mark.parametrize("name,nans,epochs,bs", [("config.CoR2", 2000, 70, 100), ("config.ODA", 3000, 100, 256)])
def test_config_module_surface(name, nans, epochs, bs):
    cf = importlib.import_module(name)
    for a in ATTRS:
        assert hasattr(cf, a), a
    if name == "config.CoR2":  # ODA leaves these two to train.py's defaults (train.py:323-447)
        assert cf.restart_epoch is None and cf.keeping_epoch == 40 and cf.load_mem is None
    else:
        assert cf.load_mem == "DB" and list(cf.test_dev_range) == [100]
    assert cf.nans == nans and cf.epochs == epochs and cf.batch_size == bs
    assert cf.lr == 1e-4 and cf.loss_metric == "KLD" and cf.arch == "rcnn" and cf.splitnum == 2
    assert cf.method_name.endswith("_VAL") and cf.log_dir.endswith("_VAL") and cf.analyze_dir.endswith("_VAL")
    assert not hasattr(cf, "sgd")  # train.py:402-406 requires this
    m = cf.Model(["PAD", "UNK"], cf.nans)
    assert hasattr(m, "seq2vec") and hasattr(m, "alpha_dict")


def _reference_shaped_sample(B, nans, vocab, dev, seed=0):
    """What datasets.py:906-970 yields per batch: 'v' float [B,36,2048], 'q_idxes' int64 [B,26] left-aligned and zero
    padded (:671-672), 'q_id' int64 [B], 'a' the soft target [B,nans] (:963-969)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    lengths = torch.randint(3, 27, (B,), generator=g)
    q = torch.randint(2, vocab, (B, 26), generator=g) * (torch.arange(26)[None, :] < lengths[:, None])
    a = torch.zeros(B, nans)
    for b in range(B):
        idx = torch.randperm(nans, generator=g)[:3]
        a[b, idx] = torch.tensor([0.6, 0.3, 0.1])
    return {"v": torch.randn(B, 36, 2048, generator=g).to(dev), "q_idxes": q.to(dev), "q_id": torch.arange(B).to(dev),
            "a": a.to(dev)}


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["config.CoR2", "config.ODA"])
def test_config_model_runs_the_reference_call_sequence(name):
    """train.py:515-517 builds ``cf.Model(q_vocab_wordlist, len(a_vocab))`` -- two arguments -- wraps it in nn.DataParallel
    and feeds it the loader's dict with int64 token ids (train.py:63); the loss is KLD-sum on log_softmax (:536-544), then
    clip_grad_norm_(0.25) and Adam (:81-86).  The drop-in module must run exactly that, with the question encoder in
    place, and agree with its own two halves (encoder, then head on the encoder's vector)."""
    import torch
    import torch.nn as nn
    cf = importlib.import_module(name)
    assert cf.question_encoder == "skipthoughts"
    dev = torch.device("cuda:0")
    vocab = ["PAD", "UNK"] + ["w%d" % i for i in range(60)]
    nans = 120
    torch.manual_seed(5)
    model = cf.Model(vocab, nans)
    assert any(k.startswith("seq2vec.embedding") for k in model.state_dict())
    assert any(k.startswith("seq2vec.gru.gru_cell.weight_hn") for k in model.state_dict())
    model = nn.DataParallel(model, device_ids=[0]).cuda()
    sample = _reference_shaped_sample(6, nans, len(vocab), dev)

    model.eval()
    with torch.no_grad():
        out = model(sample)
        assert out.shape == (6, nans) and torch.isfinite(out).all()
        qvec = model.module.seq2vec(sample["q_idxes"])
        assert qvec.shape == (6, 2400)
        two_halves = model.module({"v": sample["v"], "q_idxes": qvec})           # float [B,2400]: taken as the question vector
    assert torch.allclose(out, two_halves, rtol=1e-5, atol=1e-6)

    model.train()
    optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=cf.lr)
    before = model.module.linear_classif.linear.weight.detach().clone()
    output = model(sample)
    loss = nn.KLDivLoss(reduction="sum")(torch.nn.functional.log_softmax(output, dim=1), sample["a"])
    optimizer.zero_grad()
    loss.backward()
    total = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25)
    optimizer.step()
    assert torch.isfinite(loss) and torch.isfinite(total)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    assert not torch.equal(before, model.module.linear_classif.linear.weight)


@pytest.mark.gpu
def test_config_vector_mode_has_no_encoder(monkeypatch):
    cf = importlib.import_module("config.CoR2")
    monkeypatch.setattr(cf, "question_encoder", "vector")
    m = cf.Model(["PAD", "UNK"], 50)
    assert not any(k.startswith("seq2vec") for k in m.state_dict())
