"""Host -> device feed for the hot path: the sample wire format of the reference's loader (datasets.py:893-977) and an
asynchronous, double-buffered H2D prefetcher.

The reference yields dict batches ``{'v' [B,36,2048] fp32, 'q_idxes' [B,26] int64 (0 = PAD, left aligned), 'q_id',
'a' [B,num_ans] fp32 soft targets}`` from a ``DataLoader(pin_memory=True)`` (datasets.py:975-977) and moves them to the
GPU synchronously inside the step (train.py:54-57).  At 100 k samples/s the 151 MB of region features per 512-sample
batch is the next bottleneck (PCIe Gen5 x16: ~2.5-3 ms per batch), so here the copy of batch t+1 runs on its own HIP
stream from pinned staging buffers while batch t trains.

`FeatureStore` + `store_batches` are the part in front of that: the reference keeps the region features of ALL images in one
HDF5 file (`size,rcnn_arch,224.hy`, dataset 'att' [n_img,36,2048] float32, datasets.py:367,405-412) next to a text file with one
image file name per row (`.txt`, utils.py:452-454; row i names feature i, datasets.py:570-571) and indexes it per sample
(`outer.data['img']['feature'][visual_index]`, datasets.py:912-913).  h5py is not part of this image, so the store here is the
SAME array as a raw `.npy` file opened with numpy.memmap (tools/hy_to_npy.py converts wherever h5py exists; a contiguous HDF5
dataset is this byte range already, `FeatureStore(..., offset=)` opens it in place when the offset is known): a batch is
gathered row by row straight into a pinned staging buffer, in a few threads (one memcpy stream tops out near 10 GB/s, a
512-sample batch is 151 MB), and handed to `DevicePrefetcher`.
"""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def soft_target(a_10_idx, num_ans, out=None):
    """datasets.py:963-969: dense soft-answer vector; ``a_10_idx`` = [(answer id, probability), ...]."""
    a = out if out is not None else torch.zeros(num_ans, dtype=torch.float32)
    a.zero_()
    for c_id, c_prob in a_10_idx:
        a[c_id] = c_prob
    return a


def collate(items, num_ans, pin=False):
    """List of reference-style items {'v' [36,2048], 'q_idxes' [T], 'q_id', 'a_10_idx' or 'a'} -> one batch dict of
    contiguous host tensors (optionally pinned), the layout Model.forward and the trainer expect."""
    B = len(items)
    v0 = torch.as_tensor(items[0]["v"])
    mk = (lambda *s, dtype: torch.empty(*s, dtype=dtype).pin_memory()) if pin else (lambda *s, dtype: torch.empty(*s, dtype=dtype))
    batch = {"v": mk(B, *v0.shape, dtype=torch.float32),
             "q_idxes": mk(B, len(items[0]["q_idxes"]), dtype=torch.long),
             "q_id": torch.tensor([int(it.get("q_id", i)) for i, it in enumerate(items)], dtype=torch.long)}
    has_a = "a" in items[0] or "a_10_idx" in items[0]
    if has_a:
        batch["a"] = mk(B, num_ans, dtype=torch.float32)
    for i, it in enumerate(items):
        batch["v"][i].copy_(torch.as_tensor(it["v"], dtype=torch.float32))
        batch["q_idxes"][i].copy_(torch.as_tensor(it["q_idxes"], dtype=torch.long))
        if has_a:
            if "a" in it:
                batch["a"][i].copy_(torch.as_tensor(it["a"], dtype=torch.float32))
            else:
                soft_target(it["a_10_idx"], num_ans, out=batch["a"][i])
    return batch


def shard(batch, rank, world):
    """Rank r's contiguous slice [r*B/P, (r+1)*B/P) of every tensor of a global batch (SURVEY 8e partitioning)."""
    if world == 1:
        return batch
    out = {}
    for k, t in batch.items():
        per = t.size(0) // world
        out[k] = t[rank * per:(rank + 1) * per]
    return out


class DevicePrefetcher:
    """Iterates device-resident batches; the H2D copies of the next batch overlap the consumer's work on the current
    one.  ``depth`` staging slots are allocated once in pinned host memory and once on the device (no per-batch
    allocation), copies run on a dedicated stream, and each yielded batch is handed to the consumer's stream with an
    event wait -- the consumer never blocks the host."""

    def __init__(self, batches, device, depth=2):
        self.it = iter(batches)
        self.device = torch.device(device)
        self.depth = depth
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.slots = []          # [{'host': {...}, 'dev': {...}, 'ready': Event, 'free': Event}]
        self.queue = []
        self.n = 0

    def _slot(self, batch):
        if len(self.slots) < self.depth:
            s = {"host": {},   # pinned staging, allocated lazily and only for sources that are not page-locked
                 "dev": {k: torch.empty_like(t, device=self.device) for k, t in batch.items()} if self.cuda else None,
                 "ready": torch.cuda.Event() if self.cuda else None,
                 "free": torch.cuda.Event() if self.cuda else None}
            self.slots.append(s)
            return s
        s = self.slots[self.n % self.depth]
        if self.cuda:
            s["free"].synchronize()      # the consumer finished with this slot's previous batch
        return s

    def _issue(self):
        try:
            batch = next(self.it)
        except StopIteration:
            return False
        if not self.cuda:
            self.queue.append((None, batch))
            return True
        s = self._slot(batch)
        self.n += 1
        # The slots are sized by the first batch.  The reference's DataLoader has no drop_last (datasets.py:975), so the
        # last batch of an epoch is shorter: it is staged into / yielded as the leading rows of the slot buffers (a plain
        # copy_ would raise on the size mismatch -- or, for a remainder of ONE sample, silently broadcast it).
        src, dev = {}, {}
        for k, t in batch.items():
            full = s["dev"][k]
            if t.shape[1:] != full.shape[1:] or t.dtype != full.dtype or t.size(0) > full.size(0):
                full = s["dev"][k] = torch.empty_like(t, device=self.device)      # a larger / differently shaped batch
                s["host"].pop(k, None)
            b = t.size(0)
            if t.is_pinned():
                src[k] = t                              # already page-locked (DataLoader(pin_memory=True)): DMA from it
            else:
                if k not in s["host"] or s["host"][k].shape[1:] != t.shape[1:] or s["host"][k].size(0) < b:
                    s["host"][k] = torch.empty_like(t).pin_memory()
                stage = s["host"][k].narrow(0, 0, b)
                stage.copy_(t)                          # pageable -> pinned staging slot (a host memcpy: ~10 GB/s)
                src[k] = stage
            dev[k] = full.narrow(0, 0, b) if b != full.size(0) else full
        with torch.cuda.stream(self.stream):
            for k in batch:
                if dev[k].shape != src[k].shape:
                    raise RuntimeError("prefetcher: staging shape %s does not match the batch %s"
                                       % (tuple(dev[k].shape), tuple(src[k].shape)))
                dev[k].copy_(src[k], non_blocking=True)
            s["ready"].record(self.stream)
        self.queue.append((s, dev))
        return True

    def __iter__(self):
        for _ in range(self.depth):
            if not self._issue():
                break
        while self.queue:
            s, dev = self.queue.pop(0)
            if s is not None:
                torch.cuda.current_stream(self.device).wait_event(s["ready"])
            yield dev
            if s is not None:
                s["free"].record(torch.cuda.current_stream(self.device))
            self._issue()


class FeatureStore:
    """The reference's region-feature store (datasets.py:367,405-412: dataset 'att' [n_img,N,D] float32 of the `.hy` file; row i
    belongs to the image named on line i of the `.txt` beside it, datasets.py:570-571) as a read-only memory map.

    path: a `.npy` file holding the array (numpy.load(mmap_mode="r")), or any file that holds it as one contiguous float32 block
    at byte `offset` with `shape` given (a contiguous HDF5 dataset, a raw dump).  names: the `.txt` path or a list of image file
    names; `name_to_idx` is the reference's dict of the same name.  gather(indices, out) copies rows into `out` (a pinned host
    tensor [B,N,D] fp32, or bf16 -- then the rows are rounded on the way, half the PCIe bytes for the bf16 path) with `workers`
    threads; nothing is read until a row is asked for (the file stays on disk / in the page cache, like the reference's
    h5py handle with load_mem=None, datasets.py:565-566)."""

    def __init__(self, path, names=None, shape=None, offset=0, workers=4):
        if shape is None:
            self.features = np.load(path, mmap_mode="r")
        else:
            self.features = np.memmap(path, dtype=np.float32, mode="r", offset=int(offset), shape=tuple(int(x) for x in shape))
        if self.features.ndim != 3 or self.features.dtype != np.float32:
            raise ValueError("feature store %s: expected a float32 [n_img, regions, dim] array, got %s %s"
                             % (path, self.features.dtype, self.features.shape))
        if isinstance(names, (str, os.PathLike)):
            with open(names, encoding="utf-8") as fh:                          # utils.file2data(..., 'txt'): utils.py:452-454
                names = fh.read().split("\n")[:-1]
        self.idx_to_name = list(names) if names is not None else None
        self.name_to_idx = {n: i for i, n in enumerate(self.idx_to_name)} if names is not None else None
        if self.idx_to_name is not None and len(self.idx_to_name) != self.features.shape[0]:
            raise ValueError("feature store %s: %d names for %d feature rows" % (path, len(self.idx_to_name), self.features.shape[0]))
        self.workers = max(1, int(workers))
        self._pool = ThreadPoolExecutor(self.workers) if self.workers > 1 else None
        self._lock = threading.Lock()

    def __len__(self):
        return self.features.shape[0]

    @property
    def sample_shape(self):
        return tuple(self.features.shape[1:])

    def index(self, img_filename):
        """datasets.py:912: `outer.data['img']['name_to_idx'][item_vqa['img_filename']]` (KeyError for an unknown image)."""
        return self.name_to_idx[img_filename]

    def gather(self, indices, out):
        idx = np.asarray(indices, dtype=np.int64)
        if idx.ndim != 1 or out.shape[0] < idx.size or tuple(out.shape[1:]) != self.sample_shape:
            raise ValueError("gather: out %s does not take %d rows of %s" % (tuple(out.shape), idx.size, self.sample_shape))
        if idx.size and (idx.min() < 0 or idx.max() >= len(self)):
            raise IndexError("gather: feature index out of range [0, %d)" % len(self))
        if out.dtype == torch.float32:
            dst = out.numpy()

            def copy(lo, hi):
                for r in range(lo, hi):
                    dst[r] = self.features[idx[r]]
        elif out.dtype == torch.bfloat16:
            def copy(lo, hi):
                for r in range(lo, hi):
                    out[r].copy_(torch.from_numpy(np.array(self.features[idx[r]])))     # (a writable copy) round-to-nearest-even
        else:
            raise ValueError("gather: out must be float32 or bfloat16")
        n = idx.size
        if self._pool is None or n < 2 * self.workers:
            copy(0, n)
        else:
            step = -(-n // self.workers)
            list(self._pool.map(lambda lo: copy(lo, min(lo + step, n)), range(0, n, step)))
        return out[:n]


def store_batches(store, qa_items, batch_size, num_ans, shuffle=False, seed=0, pin=True, region_dtype=torch.float32, ring=3):
    """The reference's loader (datasets.py:893-977: `Inner.__getitem__` + DataLoader(batch_size, shuffle, pin_memory=True), no
    drop_last) over a FeatureStore: yields batch dicts {'v' [B,N,D], 'q_idxes' [B,T] int64, 'q_id' [B], 'a' [B,num_ans]} of
    host tensors.  qa_items: the reference's per-question records -- 'img_filename' (or 'v_idx'), 'q_idxes', 'q_id',
    'a_10_idx' [(answer id, probability), ...] (or a dense 'a'; neither for test splits).  The staging tensors come from a
    ring of `ring` pinned sets reused in turn (a yielded batch stays valid until `ring - 1` further batches have been drawn;
    DevicePrefetcher(depth <= ring - 1) copies it out before that).  region_dtype = torch.bfloat16 stores the regions rounded
    to bf16 (the bf16 path's transport format)."""
    n = len(qa_items)
    order = np.arange(n)
    if shuffle:
        np.random.RandomState(seed).shuffle(order)
    T = len(qa_items[0]["q_idxes"])
    has_a = "a" in qa_items[0] or "a_10_idx" in qa_items[0]
    mk = (lambda *s, dtype: torch.empty(*s, dtype=dtype).pin_memory()) if pin else (lambda *s, dtype: torch.empty(*s, dtype=dtype))
    slots = []
    for lo in range(0, n, batch_size):
        ids = order[lo:lo + batch_size]
        b = len(ids)
        k = (lo // batch_size) % max(1, ring)
        if len(slots) <= k:
            slots.append({"v": mk(batch_size, *store.sample_shape, dtype=region_dtype), "q_idxes": mk(batch_size, T, dtype=torch.long),
                          "a": mk(batch_size, num_ans, dtype=torch.float32) if has_a else None})
        s = slots[k]
        its = [qa_items[i] for i in ids]
        rows = [it["v_idx"] if "v_idx" in it else store.index(it["img_filename"]) for it in its]
        batch = {"v": store.gather(rows, s["v"]), "q_idxes": s["q_idxes"][:b],
                 "q_id": torch.tensor([int(it.get("q_id", i)) for i, it in zip(ids, its)], dtype=torch.long)}
        for r, it in enumerate(its):
            batch["q_idxes"][r].copy_(torch.as_tensor(it["q_idxes"], dtype=torch.long))
        if has_a:
            batch["a"] = s["a"][:b]
            for r, it in enumerate(its):
                if "a" in it:
                    batch["a"][r].copy_(torch.as_tensor(it["a"], dtype=torch.float32))
                else:
                    soft_target(it["a_10_idx"], num_ans, out=batch["a"][r])
        yield batch
