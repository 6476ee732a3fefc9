"""Checkpoint files and optimizer re-creation of the reference's epoch loop (train.py:250-299, :718-731), host logic on
CPU: file names, key names, resume semantics (Adam state continues, the exponential schedule starts over)."""
import os
import types

import pytest
import torch
import torch.nn as nn

from vqa_playground_pytorch_amd.trainer import DataParallelTrainer


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(12, 16)
        self.frozen = nn.Linear(3, 3)
        for p in self.frozen.parameters():
            p.requires_grad = False
        self.b = nn.Linear(16, 9)

    def forward(self, sample):
        return self.b(torch.tanh(self.a(sample["x"])))


def batches(n, seed=3):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(8, 12, generator=g), torch.softmax(torch.randn(8, 9, generator=g), 1)) for _ in range(n)]


def make(seed):
    torch.manual_seed(seed)
    model = Tiny()
    return model, DataParallelTrainer(model, lr=1e-2, clip=0.25, gamma=0.9)


class Logger:
    def __init__(self):
        self.saved = None

    def to_json(self, path):
        self.saved = path
        with open(path, "w") as f:
            f.write("{}")


def test_checkpoint_files_and_resume(tmp_path):
    data = batches(4)
    model, tr = make(0)
    for x, a in data[:2]:
        tr.step({"x": x}, a)
    logger = Logger()
    path = tr.save_checkpoint({"epoch": 7, "exp_logger": logger}, str(tmp_path))
    assert path == os.path.join(str(tmp_path), "epoch_7")
    assert sorted(os.listdir(path)) == ["ckpt_info.pth.tar", "ckpt_model.pth.tar", "ckpt_optim.pth.tar", "logger.json"]
    assert logger.saved == os.path.join(path, "logger.json")
    saved_model = torch.load(os.path.join(path, "ckpt_model.pth.tar"))
    assert list(saved_model) == list(model.state_dict())
    saved_optim = torch.load(os.path.join(path, "ckpt_optim.pth.tar"))
    # only the trainable parameters, in model.parameters() order (train.py:288-292)
    assert saved_optim["param_groups"][0]["params"] == [0, 1, 2, 3]
    assert sorted(saved_optim["state"]) == [0, 1, 2, 3]
    assert int(saved_optim["state"][0]["step"]) == 2

    # resume into a differently initialised replica; the schedule restarts from lr0 (the reference builds a new
    # ExponentialLR before load_checkpoint and does not save it), Adam's moments and step count continue
    model2, tr2 = make(99)
    got_logger = tr2.load_checkpoint(path)
    assert isinstance(got_logger, Logger)
    tr.iteration = 0
    for x, a in data[2:]:
        l1, n1 = tr.step({"x": x}, a)
        l2, n2 = tr2.step({"x": x}, a)
        assert l1.item() == pytest.approx(l2.item(), rel=1e-6)
        assert tr.lr == pytest.approx(tr2.lr, rel=1e-12)
    for p, q in zip(model.parameters(), model2.parameters()):
        torch.testing.assert_close(p, q, rtol=1e-6, atol=1e-7)


def test_begin_epoch_recreates_optimizer_like_the_reference_loop():
    import config.CoR2 as cf       # restart_epoch None, keeping_epoch 40
    data = batches(3)
    model, tr = make(0)
    tr.base_lr = cf.lr
    for x, a in data:
        tr.step({"x": x}, a)
    assert tr.iteration == 3 and tr.lr < cf.lr
    assert tr.begin_epoch(5, cf) is True            # epoch < keeping_epoch: learning_scheduler(cf) again
    assert tr.iteration == 0 and tr.lr == cf.lr
    assert len(tr.optimizer.state_dict()["state"]) == 0
    tr.step({"x": data[0][0]}, data[0][1])
    assert tr.begin_epoch(cf.keeping_epoch, cf) is False
    assert tr.iteration == 1
    other = types.SimpleNamespace(lr=3e-4, restart_epoch=50)
    assert tr.begin_epoch(49, other) is False
    assert tr.begin_epoch(50, other) is True
    assert tr.lr == 3e-4


def test_fresh_step_after_reset_equals_new_trainer():
    data = batches(3)
    model, tr = make(0)
    for x, a in data[:2]:
        tr.step({"x": x}, a)
    tr.reset_optimizer()
    torch.manual_seed(1)
    model2 = Tiny()
    model2.load_state_dict(model.state_dict())
    tr2 = DataParallelTrainer(model2, lr=1e-2, clip=0.25, gamma=0.9)
    x, a = data[2]
    tr.step({"x": x}, a)
    tr2.step({"x": x}, a)
    for p, q in zip(model.parameters(), model2.parameters()):
        torch.testing.assert_close(p, q, rtol=0, atol=0)
