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


def _store(tmp_path, n_img=11, N=4, D=8, seed=3):
    rs = np.random.RandomState(seed)
    feats = rs.standard_normal((n_img, N, D)).astype(np.float32)
    np.save(tmp_path / "size,rcnn_arch,224.npy", feats)
    names = ["COCO_train2014_%012d.jpg" % (7 * i + 1) for i in range(n_img)]
    (tmp_path / "size,rcnn_arch,224.txt").write_text("".join(n + "\n" for n in names), encoding="utf-8")
    return feats, names


def test_feature_store_reads_the_on_disk_layout(tmp_path):
    """SURVEY 8f row 4 / VERDICT r05 missing #4: the store is the reference's 'att' array [n_img,N,D] fp32 + the one-name-per-row
    text file (datasets.py:367,405-412,570-571; utils.py:452-454), memory-mapped; a sample's regions are
    feature[name_to_idx[img_filename]] (datasets.py:912-913)."""
    feats, names = _store(tmp_path)
    for workers in (1, 3):
        st = feed.FeatureStore(tmp_path / "size,rcnn_arch,224.npy", tmp_path / "size,rcnn_arch,224.txt", workers=workers)
        assert st.populate() is st
        assert len(st) == 11 and st.sample_shape == (4, 8) and st.index(names[5]) == 5
        out = torch.empty(9, 4, 8)
        got = st.gather([10, 0, 5, 5, 3, 9, 1], out)
        assert got.shape == (7, 4, 8) and np.array_equal(got.numpy(), feats[[10, 0, 5, 5, 3, 9, 1]])
        half = st.gather([2, 8], torch.empty(2, 4, 8, dtype=torch.bfloat16))
        assert torch.equal(half, torch.from_numpy(feats[[2, 8]]).to(torch.bfloat16))
        with pytest.raises(KeyError):
            st.index("no_such_image.jpg")
        with pytest.raises(IndexError):
            st.gather([11], out)
        with pytest.raises(ValueError):
            st.gather([0], torch.empty(1, 4, 9))
    # the same bytes inside another container (a contiguous HDF5 dataset is such a block): opened in place at its offset
    raw = tmp_path / "container.bin"
    raw.write_bytes(b"\x89HDF" + b"\0" * 92 + feats.tobytes())
    st = feed.FeatureStore(raw, names, shape=feats.shape, offset=96)
    assert np.array_equal(st.gather([4], torch.empty(1, 4, 8)).numpy(), feats[[4]])
    with pytest.raises(ValueError):
        feed.FeatureStore(tmp_path / "size,rcnn_arch,224.npy", names[:-1])


def test_store_batches_equal_the_reference_items_collated(tmp_path):
    """store_batches == collate() of the items the reference's Inner.__getitem__ would build (datasets.py:906-969), batch by
    batch, shuffled or not, the short last batch included (no drop_last, datasets.py:975)."""
    feats, names = _store(tmp_path)
    st = feed.FeatureStore(tmp_path / "size,rcnn_arch,224.npy", names, workers=2)
    rs = np.random.RandomState(1)
    qa = []
    for i in range(13):
        ids = rs.choice(20, size=3, replace=False)
        p = rs.dirichlet(np.ones(3))
        qa.append({"img_filename": names[int(rs.randint(11))], "q_idxes": [1 + i, 2, 3, 0, 0, 0], "q_id": 500 + i,
                   "a_10_idx": [(int(c), float(x)) for c, x in zip(ids, p)]})
    for shuffle in (False, True):
        order = np.arange(13)
        if shuffle:
            np.random.RandomState(9).shuffle(order)
        got = [{k: t.clone() for k, t in b.items()} for b in feed.store_batches(st, qa, 5, 20, shuffle=shuffle, seed=9, pin=False)]
        assert [b["v"].shape[0] for b in got] == [5, 5, 3]
        for k, b in enumerate(got):
            its = [dict(qa[i], v=feats[st.index(qa[i]["img_filename"])]) for i in order[5 * k:5 * k + 5]]
            want = feed.collate(its, 20)
            for key in ("v", "q_idxes", "q_id", "a"):
                assert torch.equal(b[key], want[key]), (shuffle, k, key)
    # test split: no answers; bf16 transport; the ring reuses its staging tensors
    test_items = [{"v_idx": i % 11, "q_idxes": [1, 0, 0], "q_id": i} for i in range(8)]
    seen = []
    for b in feed.store_batches(st, test_items, 4, 20, pin=False, region_dtype=torch.bfloat16, ring=2):
        assert set(b) == {"v", "q_idxes", "q_id"} and b["v"].dtype == torch.bfloat16
        seen.append(b["v"].data_ptr())
    assert len(seen) == 2 and seen[0] != seen[1]
    # through the prefetcher (CPU pass-through here; tests/test_gpu_models.py feeds a trainer from it on the GPU)
    out = list(feed.DevicePrefetcher(feed.store_batches(st, qa, 5, 20, pin=False, ring=4), "cpu", depth=2))
    assert len(out) == 3 and torch.equal(out[2]["q_id"], torch.tensor([510, 511, 512]))


def test_store_batches_with_a_background_producer(tmp_path):
    """prefetch > 0: the batches are assembled by a background thread; same batches, same order, and a ring that is too small for
    the look-ahead is refused (a slot would be refilled under the consumer)."""
    feats, names = _store(tmp_path)
    st = feed.FeatureStore(tmp_path / "size,rcnn_arch,224.npy", names, workers=2)
    qa = [{"v_idx": (3 * i) % 11, "q_idxes": [i, 1, 0], "q_id": i, "a_10_idx": [((i * 7) % 20, 0.75), ((i * 7 + 1) % 20, 0.25)]} for i in range(23)]
    plain = [{k: t.clone() for k, t in b.items()} for b in feed.store_batches(st, qa, 4, 20, pin=False)]
    ahead = [{k: t.clone() for k, t in b.items()} for b in feed.store_batches(st, qa, 4, 20, pin=False, ring=5, prefetch=2)]
    assert len(plain) == len(ahead) == 6
    for b0, b1 in zip(plain, ahead):
        assert all(torch.equal(b0[k], b1[k]) for k in b0)
    assert torch.allclose(plain[0]["a"].sum(1), torch.ones(4))
    with pytest.raises(ValueError):
        list(feed.store_batches(st, qa, 4, 20, pin=False, ring=3, prefetch=2))
    # a consumer that stops early does not leave the producer blocked
    it = feed.store_batches(st, qa, 4, 20, pin=False, ring=5, prefetch=2)
    next(it)
    it.close()
    # float question vectors (the identity-encoder slot)
    qv = [{"v_idx": 0, "q_idxes": np.full(5, 0.5, np.float32), "q_id": 0}]
    b = next(feed.store_batches(st, qv, 1, 20, pin=False, q_dtype=torch.float32))
    assert b["q_idxes"].dtype == torch.float32 and float(b["q_idxes"].sum()) == 2.5


def test_feature_store_bf16_rounding_is_round_to_nearest_even(tmp_path):
    """gather(..., out=bf16) rounds in numpy inside its worker threads (not on torch's intra-op pool): the same bits as torch's
    float32 -> bfloat16 conversion on ties, halfway cases, subnormals, +-Inf, the largest finite values (which round to Inf) and NaN."""
    special = np.array([0.0, -0.0, 1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -7, 1.0 + 3 * 2.0 ** -8, -(1.0 + 2.0 ** -8), 3.3961775e38, 3.4e38, -3.4028235e38,
                        np.inf, -np.inf, 1e-40, -1e-45, 2.0 ** -126, 65504.0, 0.1, 1 / 3], np.float32)
    rs = np.random.RandomState(4)
    feats = rs.standard_normal((6, 4, 8)).astype(np.float32) * np.exp2(rs.randint(-30, 30, (6, 4, 8))).astype(np.float32)
    feats.reshape(-1)[:special.size] = special
    feats[5, 3, 7] = np.nan
    np.save(tmp_path / "f.npy", feats)
    for workers in (1, 3):
        st = feed.FeatureStore(tmp_path / "f.npy", workers=workers)
        got = st.gather([0, 5, 2, 5, 1, 3, 4], torch.empty(7, 4, 8, dtype=torch.bfloat16))
        want = torch.from_numpy(feats[[0, 5, 2, 5, 1, 3, 4]]).to(torch.bfloat16)
        nan = torch.isnan(want)
        assert torch.equal(torch.isnan(got), nan)
        assert torch.equal(got.view(torch.int16)[~nan], want.view(torch.int16)[~nan])


def test_store_batches_over_epochs_keeps_its_staging(tmp_path):
    """epochs = k: k passes, each with its own shuffle (seed + epoch), through ONE ring of staging tensors."""
    feats, names = _store(tmp_path)
    st = feed.FeatureStore(tmp_path / "size,rcnn_arch,224.npy", names, workers=1)
    qa = [{"v_idx": i % 11, "q_idxes": [i, 0], "q_id": i} for i in range(10)]
    table = feed.qa_table(st, qa, 20)
    seen, ptrs = [], set()
    for b in feed.store_batches(st, table, 4, 20, shuffle=True, seed=3, pin=False, ring=2, epochs=3):
        seen.append(b["q_id"].tolist())
        ptrs.add(b["v"].data_ptr())
    assert len(seen) == 9 and [len(x) for x in seen] == [4, 4, 2] * 3 and len(ptrs) == 2
    epochs = [sorted(sum(seen[3 * e:3 * e + 3], [])) for e in range(3)]
    assert epochs == [list(range(10))] * 3 and seen[0:3] != seen[3:6]
    it = feed.store_batches(st, table, 4, 20, pin=False, epochs=None)
    assert [next(it)["q_id"].tolist() for _ in range(4)][3] == [0, 1, 2, 3]       # wraps around for ever
    it.close()
