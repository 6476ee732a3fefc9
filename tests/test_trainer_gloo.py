"""world_size-2 `gloo` test of the data-parallel step (flat-gradient SUM all-reduce, clip, Adam, per-iteration
exponential lr): two ranks on half batches must reproduce one process on the whole batch.  Also pins the
step semantics against the reference-generated 3-step trajectory through the oracle model."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from vqa_playground_pytorch_amd.trainer import DataParallelTrainer, FlatGradients, kld_sum_loss


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(12, 16)
        self.b = nn.Linear(16, 9)

    def forward(self, sample):
        return self.b(torch.tanh(self.a(sample["x"])))


def make_data(steps=4, batch=8):
    g = torch.Generator().manual_seed(5)
    return [(torch.randn(batch, 12, generator=g), torch.softmax(torch.randn(batch, 9, generator=g), 1)) for _ in range(steps)]


def run_single():
    torch.manual_seed(0)
    model = Tiny()
    tr = DataParallelTrainer(model, lr=1e-2, clip=0.25)
    out = []
    for x, a in make_data():
        loss, norm = tr.step({"x": x}, a)
        out.append((loss.item(), norm.item()))
    return out, [p.detach().clone() for p in model.parameters()], tr.lr


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # different init per rank: the trainer must broadcast rank 0's weights
    model = Tiny()
    if rank == 0:
        torch.manual_seed(0)
        model = Tiny()
    tr = DataParallelTrainer(model, lr=1e-2, clip=0.25)
    losses = []
    for x, a in make_data():
        loss, norm = tr.step({"x": tr.shard(x)}, tr.shard(a))
        t = loss.clone()
        dist.all_reduce(t)
        losses.append((t.item(), norm.item()))
    q.put((rank, losses, [p.detach().numpy().copy() for p in model.parameters()], tr.lr))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_losses, ref_params, ref_lr = run_single()
    for rank, losses, params, lr in results:
        assert lr == pytest.approx(ref_lr, rel=1e-12)
        for (l, n), (rl, rn) in zip(losses, ref_losses):
            assert l == pytest.approx(rl, rel=1e-5)     # summed shard losses == global-batch loss (SUM reduction)
            assert n == pytest.approx(rn, rel=1e-5)     # clip sees the global-batch gradient norm on every rank
        for p, rp in zip(params, ref_params):
            np.testing.assert_allclose(p, rp.numpy(), rtol=1e-4, atol=1e-6)
    for p0, p1 in zip(results[0][2], results[1][2]):
        np.testing.assert_allclose(p0, p1, rtol=0, atol=1e-7)   # replicas stay in lock-step


def test_flat_gradients_views_and_clip():
    torch.manual_seed(1)
    model = Tiny()
    fg = FlatGradients(model.parameters())
    x, a = make_data(1)[0]
    kld_sum_loss(model({"x": x}), a).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.equal(ref, fg.buffer)                      # grads ARE the buffer (views)
    manual = torch.sqrt(sum((p.grad ** 2).sum() for p in model.parameters()))
    norm = fg.clip_(0.25)
    assert norm.item() == pytest.approx(manual.item(), rel=1e-6)
    assert torch.linalg.vector_norm(fg.buffer).item() == pytest.approx(min(0.25, manual.item()), rel=1e-4)
    fg.zero()
    assert all((p.grad == 0).all() for p in model.parameters())


def test_step_semantics_match_reference_trajectory(golden_dir):
    """Through the oracle CoR2 model on CPU: the trainer's loss/clip/Adam/lr order reproduces the 3-step
    trajectory recorded from the reference model with a hand-written train.py:41-107 loop."""
    from oracle import reference_faithful as RF
    from oracle import seeded
    gold = np.load(os.path.join(golden_dir, "cor2_b4.npz"))
    model = seeded.load_state(RF.CoR2Oracle(2000), 0).eval()
    tr = DataParallelTrainer(model, lr=1e-4, clip=0.25)
    losses, norms = [], []
    for step in range(3):
        v, q, a = (torch.from_numpy(x) for x in seeded.seeded_inputs(4, answers=2000, seed=101 + step))
        loss, norm = tr.step({"v": v, "q": q}, a)
        losses.append(loss.item())
        norms.append(norm.item())
    np.testing.assert_allclose(losses, gold["train3.loss"], rtol=1e-5)
    np.testing.assert_allclose(norms, gold["train3.gnorm"], rtol=1e-4)
    assert tr.lr == pytest.approx(float(gold["train3.lr"]), rel=1e-12)
    for name, p in model.named_parameters():
        w = p.detach().numpy().astype(np.float64)
        assert np.sqrt((w ** 2).sum()) == pytest.approx(float(gold["train3.w." + name + ".norm"]), rel=2e-5)
