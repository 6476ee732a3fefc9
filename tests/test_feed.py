"""Sample wire format + prefetcher."""
import numpy as np
import pytest
import torch

from vqa_playground_pytorch_amd import feed


def items(n, C=20, T=6, seed=0):
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        ids = rs.choice(C, size=3, replace=False)
        p = rs.dirichlet(np.ones(3))
        out.append({"v": rs.standard_normal((4, 8)).astype(np.float32), "q_idxes": [1 + i, 2, 0, 0, 0, 0][:T], "q_id": 100 + i,
                    "a_10_idx": [(int(c), float(x)) for c, x in zip(ids, p)]})
    return out


def test_soft_target_matches_reference_construction():
    # datasets.py:963-969: zero vector of len(a_vocab), then a[c_id] = c_prob for every (c_id, c_prob)
    a = feed.soft_target([(3, 0.5), (7, 0.3), (0, 0.2)], 10)
    want = np.zeros(10, np.float32)
    want[3], want[7], want[0] = 0.5, 0.3, 0.2
    assert np.array_equal(a.numpy(), want) and abs(a.sum().item() - 1.0) < 1e-6


def test_collate_and_shard():
    its = items(6)
    b = feed.collate(its, 20)
    assert b["v"].shape == (6, 4, 8) and b["q_idxes"].dtype == torch.long and b["a"].shape == (6, 20)
    assert b["q_id"].tolist() == [100, 101, 102, 103, 104, 105]
    assert torch.allclose(b["a"].sum(1), torch.ones(6))
    s1 = feed.shard(b, 1, 3)
    assert torch.equal(s1["v"], b["v"][2:4]) and torch.equal(s1["a"], b["a"][2:4])
    assert feed.shard(b, 0, 1) is b


def test_prefetcher_cpu_passthrough_keeps_order():
    batches = [feed.collate(items(2, seed=s), 20) for s in range(5)]
    got = list(feed.DevicePrefetcher(batches, "cpu"))
    assert len(got) == 5 and all(torch.equal(g["v"], b["v"]) for g, b in zip(got, batches))


@pytest.mark.gpu
def test_prefetcher_gpu_delivers_identical_batches_with_slot_reuse():
    dev = torch.device("cuda:0")
    batches = [feed.collate(items(8, seed=s), 20) for s in range(7)]       # 7 batches through 2 slots
    pf = feed.DevicePrefetcher(batches, dev, depth=2)
    for i, g in enumerate(pf):
        assert g["v"].device.type == "cuda"
        x = g["v"] * 2.0                                                    # consumer work on the current stream
        assert torch.equal(g["v"].cpu(), batches[i]["v"]) and torch.equal(g["a"].cpu(), batches[i]["a"])
        assert torch.equal(g["q_idxes"].cpu(), batches[i]["q_idxes"]) and torch.equal(x.cpu(), batches[i]["v"] * 2.0)
    assert i == 6 and len(pf.slots) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("last", [3, 1])
def test_prefetcher_gpu_short_final_batch(last):
    """The reference's DataLoader has no drop_last (datasets.py:975): the final batch of an epoch is shorter -- also by all
    but ONE sample, which a plain copy_ into the full-size slot would silently broadcast."""
    dev = torch.device("cuda:0")
    batches = [feed.collate(items(8, seed=s), 20) for s in range(4)] + [feed.collate(items(last, seed=9), 20)]
    n = 0
    for g, b in zip(feed.DevicePrefetcher(batches, dev, depth=2), batches):     # (a yielded batch lives until the next one)
        for k in b:
            assert g[k].shape == b[k].shape and torch.equal(g[k].cpu(), b[k]), k
        n += 1
    assert n == 5
