"""Host -> device feed for the hot path: the sample wire format of the reference's loader (datasets.py:893-977) and an
asynchronous, double-buffered H2D prefetcher.

The reference yields dict batches ``{'v' [B,36,2048] fp32, 'q_idxes' [B,26] int64 (0 = PAD, left aligned), 'q_id',
'a' [B,num_ans] fp32 soft targets}`` from a ``DataLoader(pin_memory=True)`` (datasets.py:975-977) and moves them to the
GPU synchronously inside the step (train.py:54-57).  At 100 k samples/s the 151 MB of region features per 512-sample
batch is the next bottleneck (PCIe Gen5 x16: ~2.5-3 ms per batch), so here the copy of batch t+1 runs on its own HIP
stream from pinned staging buffers while batch t trains.
"""
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
