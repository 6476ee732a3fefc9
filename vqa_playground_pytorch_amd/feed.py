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

    def populate(self):
        """Map every page of the store into this process now (MADV_POPULATE_READ where the kernel has it, else one read per 4 KiB
        page).  A cold mapping pays a minor page fault per page on first touch -- 72 per sample, under the process's mmap lock, so
        a 16-thread gather of 512 cold rows takes 60-90 ms instead of 2 (measured, tools/feed_host_profile.py); a store that fits
        RAM is worth populating once, a larger one lives with first-touch faults like the reference's h5py reads do."""
        mm = getattr(self.features, "_mmap", None)
        try:
            if mm is None:
                raise OSError
            mm.madvise(22)                                     # MADV_POPULATE_READ (Linux >= 5.14)
        except (OSError, ValueError, AttributeError):
            flat = self.features.reshape(-1)
            step = 1 << 20                                     # 4 MiB of floats per slice: one element per page is read
            total = 0.0
            for lo in range(0, flat.size, step * 256):
                total += float(flat[lo:lo + step * 256:1024].sum())
        return self

    @property
    def sample_shape(self):
        return tuple(self.features.shape[1:])

    def index(self, img_filename):
        """datasets.py:912: `outer.data['img']['name_to_idx'][item_vqa['img_filename']]` (KeyError for an unknown image)."""
        return self.name_to_idx[img_filename]

    def gather(self, indices, out):
        """out[r] = feature[indices[r]] (rows of N x D floats), fp32 or rounded to bf16 (nearest even), on `workers` threads."""
        idx = np.asarray(indices, dtype=np.int64)
        if idx.ndim != 1 or out.shape[0] < idx.size or tuple(out.shape[1:]) != self.sample_shape:
            raise ValueError("gather: out %s does not take %d rows of %s" % (tuple(out.shape), idx.size, self.sample_shape))
        if idx.size and (idx.min() < 0 or idx.max() >= len(self)):
            raise IndexError("gather: feature index out of range [0, %d)" % len(self))
        n = idx.size
        if out.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("gather: out must be float32 or bfloat16")
        if n == 0:
            return out[:0]
        # ONE native call for the batch (csrc/feed_host.hip: `workers` threads, memcpy or round-to-nearest-even to bf16): a ctypes call
        # drops the GIL once, where a numpy.take per worker thread took and returned it per call -- next to a training loop that holds
        # the GIL for milliseconds at a time that was 9-19 ms per batch instead of 2-3 (tools/feed_bench.py --store)
        if self.features.flags["C_CONTIGUOUS"] and out.is_contiguous():
            from . import _lib
            try:
                L_ = _lib.lib()
            except _lib.VqaLibraryError:
                L_ = None
            if L_ is not None:
                idx = np.ascontiguousarray(idx)
                _lib.check(L_.vqa_host_gather_rows(self.features.ctypes.data, len(self), int(np.prod(self.sample_shape)), idx.ctypes.data, n,
                                                   out.data_ptr(), int(out.dtype == torch.bfloat16), self.workers), "host_gather_rows")
                return out[:n]
        # (no library: numpy, one take per worker thread; bf16 through torch's conversion)
        dst = out.numpy() if out.dtype == torch.float32 else np.empty((n,) + self.sample_shape, dtype=np.float32)

        def copy(lo, hi):
            if hi > lo:
                np.take(self.features, idx[lo:hi], axis=0, out=dst[lo:hi], mode="clip")   # (range-checked above; mode="raise" buffers `out`)
        if self._pool is None or n < 2 * self.workers:
            copy(0, n)
        else:
            step = -(-n // self.workers)
            list(self._pool.map(lambda lo: copy(lo, min(lo + step, n)), range(0, n, step)))
        if out.dtype == torch.bfloat16:
            out[:n].copy_(torch.from_numpy(dst[:n]))
        return out[:n]


class _QaTable:
    """The per-question records turned into arrays ONCE (the reference keeps dicts and pays Python per sample per epoch in its
    DataLoader workers): feature rows, the question matrix, ids, and the soft answers as CSR (datasets.py:963-969: a[c_id] = c_prob,
    the last pair of a duplicated id wins)."""

    def __init__(self, store, qa_items, num_ans, q_dtype):
        n = len(qa_items)
        self.rows = np.fromiter((it["v_idx"] if "v_idx" in it else store.index(it["img_filename"]) for it in qa_items), np.int64, n)
        self.q_ids = np.fromiter((int(it.get("q_id", i)) for i, it in enumerate(qa_items)), np.int64, n)
        q_np = {torch.long: np.int64, torch.float32: np.float32, torch.float64: np.float64}.get(q_dtype, np.float32)
        self.q = np.asarray([np.asarray(it["q_idxes"]) for it in qa_items]).astype(q_np, copy=False)
        self.kind = "dense" if "a" in qa_items[0] else ("soft" if "a_10_idx" in qa_items[0] else None)
        if self.kind == "dense":
            self.a = np.asarray([np.asarray(it["a"], dtype=np.float32) for it in qa_items])
        elif self.kind == "soft":
            counts = np.fromiter((len(it["a_10_idx"]) for it in qa_items), np.int64, n)
            self.ptr = np.concatenate([[0], np.cumsum(counts)])
            flat = [pair for it in qa_items for pair in it["a_10_idx"]]
            self.cols = np.fromiter((int(c) for c, _ in flat), np.int64, len(flat))
            self.vals = np.fromiter((float(p) for _, p in flat), np.float32, len(flat))
            if len(flat) and (self.cols.min() < 0 or self.cols.max() >= num_ans):
                raise IndexError("store_batches: answer id outside [0, %d)" % num_ans)

    def fill_answers(self, a, ids):
        dst = a.numpy()
        if self.kind == "dense":
            dst[...] = self.a[ids]
            return
        dst.fill(0.0)
        lo, hi = self.ptr[ids], self.ptr[ids + 1]
        counts = hi - lo
        if counts.sum():
            rows = np.repeat(np.arange(len(ids)), counts)
            src = np.concatenate([np.arange(x, y) for x, y in zip(lo, hi)]) if len(ids) else np.zeros(0, np.int64)
            dst[rows, self.cols[src]] = self.vals[src]        # (in order: the last pair of a duplicated id wins, as in the reference's loop)


def qa_table(store, qa_items, num_ans, q_dtype=torch.long):
    """The array form of the per-question records that store_batches works on; build it once, pass it instead of the list."""
    return _QaTable(store, qa_items, num_ans, q_dtype)


def store_batches(store, qa_items, batch_size, num_ans, shuffle=False, seed=0, pin=True, region_dtype=torch.float32, ring=3,
                  q_dtype=torch.long, prefetch=0, epochs=1, producers=1):
    """The reference's loader (datasets.py:893-977: `Inner.__getitem__` + DataLoader(batch_size, shuffle, pin_memory=True), no
    drop_last) over a FeatureStore: yields batch dicts {'v' [B,N,D], 'q_idxes' [B,T], 'q_id' [B], 'a' [B,num_ans]} of host tensors.
    qa_items: the reference's per-question records -- 'img_filename' (or 'v_idx'), 'q_idxes', 'q_id', 'a_10_idx' [(answer id,
    probability), ...] (or a dense 'a'; neither for test splits).  The staging tensors come from a ring of `ring` pinned sets reused
    in turn (a yielded batch stays valid until `ring - 1 - prefetch` further batches have been drawn; DevicePrefetcher(depth) needs
    ring >= depth + prefetch + 1).  region_dtype = torch.bfloat16 stores the regions rounded to bf16 (the transport format: half the
    PCIe bytes; the fp32 path widens them exactly on the device).  q_dtype: torch.long for token ids (the reference), a float type when
    'q_idxes' holds precomputed question vectors (the models' identity-encoder slot).  The records are turned into arrays once
    per call (`qa_table(...)` does it ahead of time: pass its result as qa_items to reuse it over epochs).  prefetch > 0: batches are assembled by a
    background thread, up to `prefetch` ahead of the consumer (the DataLoader's worker processes, as one thread: numpy and torch
    release the GIL in the copies that dominate); producers > 1: that many threads, each assembling whole batches, delivered in
    order.  epochs: passes over the records (None: for ever), each with its own shuffle
    (seed + epoch) -- ONE generator for a whole run keeps its pinned staging ring (page-locking 160 MB per slot takes tens of
    milliseconds: a generator per epoch would pay that again every time)."""
    n = len(qa_items.rows) if isinstance(qa_items, _QaTable) else len(qa_items)

    def order_of(epoch):
        o = np.arange(n)
        if shuffle:
            np.random.RandomState(seed + epoch).shuffle(o)
        return o
    table = qa_items if isinstance(qa_items, _QaTable) else _QaTable(store, qa_items, num_ans, q_dtype)
    T = table.q.shape[1]
    has_a = table.kind is not None
    mk = (lambda *s, dtype: torch.empty(*s, dtype=dtype).pin_memory()) if pin else (lambda *s, dtype: torch.empty(*s, dtype=dtype))
    slots = []

    slot_lock = threading.Lock()

    def assemble(k, job):
        order, lo = job
        ids = order[lo:lo + batch_size]
        b = len(ids)
        with slot_lock:
            while len(slots) <= k:
                slots.append({"v": mk(batch_size, *store.sample_shape, dtype=region_dtype), "q_idxes": mk(batch_size, T, dtype=q_dtype),
                              "a": mk(batch_size, num_ans, dtype=torch.float32) if has_a else None})
        s = slots[k]
        batch = {"v": store.gather(table.rows[ids], s["v"]), "q_idxes": s["q_idxes"][:b], "q_id": torch.from_numpy(table.q_ids[ids])}
        batch["q_idxes"].numpy()[...] = table.q[ids]
        if has_a:
            batch["a"] = s["a"][:b]
            table.fill_answers(batch["a"], ids)
        return batch

    def jobs():
        epoch = 0
        while epochs is None or epoch < epochs:
            order = order_of(epoch)
            for lo in range(0, n, batch_size):
                yield order, lo
            epoch += 1
    ring = max(1, ring)
    if prefetch <= 0:
        for j, job in enumerate(jobs()):
            yield assemble(j % ring, job)
        return
    if ring < prefetch + 2:
        raise ValueError("store_batches: ring=%d is too small for prefetch=%d (a slot would be refilled while still in use)" % (ring, prefetch))
    from collections import deque
    ex = ThreadPoolExecutor(max(1, int(producers)))
    window, it = deque(), enumerate(jobs())

    def submit_next():
        try:
            j, job = next(it)
        except StopIteration:
            return False
        window.append(ex.submit(assemble, j % ring, job))
        return True
    try:
        for _ in range(prefetch):
            if not submit_next():
                break
        while window:
            fut = window.popleft()
            batch = fut.result()
            submit_next()
            yield batch
    finally:
        for fut in window:
            fut.cancel()
        ex.shutdown(wait=True)
