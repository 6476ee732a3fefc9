#!/usr/bin/env python3
"""PCIe-inclusive rate of the CoR2 training step: batches start in (pinned) host memory and are streamed through
feed.DevicePrefetcher while the hipGraph-replayed step runs.  NOT the headline number (bench.py times inputs resident
in HBM); it sizes the feed path of SURVEY 8f row 4.   python tools/feed_bench.py [--steps 30]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vqa_playground_pytorch_amd import CoR2Model, feed  # noqa: E402
from vqa_playground_pytorch_amd.trainer import DataParallelTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--depth", type=int, default=3, help="device slots of the prefetcher = input slots of the trainer (a step graph per "
                    "slot reads it in place); 2: the host waits for step t before it may refill the slot step t + 2 reads")
    ap.add_argument("--copy", action="store_true", help="the trainer copies every batch into its own graph input buffers (round 5's form)")
    ap.add_argument("--store", type=int, default=0, metavar="N_IMG",
                    help="feed from a feed.FeatureStore of N_IMG synthetic images (a .npy under /tmp, memory-mapped) through "
                         "feed.store_batches: the per-sample row gather into pinned staging is part of what is timed")
    ap.add_argument("--workers", type=int, default=8, help="gather threads of the FeatureStore")
    ap.add_argument("--producers", type=int, default=2, help="threads that assemble whole batches ahead of the consumer")
    ap.add_argument("--switch-interval", type=float, default=0.0, help="sys.setswitchinterval (seconds; 0 = leave Python's 5 ms): how "
                    "long the trainer's thread may keep the GIL while the producer thread waits for it")
    ap.add_argument("--transport", choices=["f32", "bf16"], default="f32",
                    help="bf16: the regions cross PCIe as bf16 (half the bytes) and are widened on the device (ops.widen_bf16); the step is the fp32 one")
    args = ap.parse_args()
    if args.switch_interval > 0:
        sys.setswitchinterval(args.switch_interval)
    dev = torch.device("cuda:0")
    B = args.batch
    torch.manual_seed(0)
    vdt = torch.bfloat16 if args.transport == "bf16" else torch.float32
    host = [{"v": torch.randn(B, 36, 2048).to(vdt).pin_memory(), "q_idxes": torch.randn(B, 2400).pin_memory(),
             "a": torch.softmax(torch.randn(B, 2000), 1).pin_memory()} for _ in range(3)]
    store = qa = None
    if args.store:
        import numpy as np
        path = "/tmp/feed_bench_store_%d.npy" % args.store
        if not os.path.exists(path):
            mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(args.store, 36, 2048))
            rs = np.random.RandomState(0)
            for lo in range(0, args.store, 256):
                mm[lo:lo + 256] = rs.standard_normal((min(256, args.store - lo), 36, 2048)).astype(np.float32)
            mm.flush()
            del mm
        store = feed.FeatureStore(path, workers=args.workers).populate()     # (steady state: no first-touch page faults in the timed region)
        rs = np.random.RandomState(1)
        qa = [{"v_idx": int(rs.randint(args.store)), "q_idxes": rs.standard_normal(2400).astype(np.float32), "q_id": i,
               "a_10_idx": [(int(c), 0.1) for c in rs.choice(2000, 10, replace=False)]} for i in range(B * 8)]
        qa = feed.qa_table(store, qa, 2000, q_dtype=torch.float32)          # the records as arrays, once
    model = CoR2Model(["PAD"], 2000).to(dev).train()
    tr = DataParallelTrainer(model, graph=True) if args.copy else \
        DataParallelTrainer(model, graph=True, adopt_inputs=True, input_slots=args.depth)

    stats = {}

    def stream(n):
        if store is not None:        # the reference loader's batches from the memory-mapped store (question vectors stand in for ids)
            loader = feed.store_batches(store, qa, B, 2000, shuffle=True, seed=0, pin=True, region_dtype=vdt, ring=args.depth + 2 * args.producers + 2,
                                        q_dtype=torch.float32, prefetch=2 * args.producers, epochs=None, producers=args.producers)
            waited = 0.0
            for _ in range(n):
                t_w = time.perf_counter()
                b = next(loader)
                waited += time.perf_counter() - t_w
                yield {"v": b["v"], "q_idxes": b["q_idxes"], "a": b["a"]}
            loader.close()
            stats["producer_wait_ms_per_batch"] = 1e3 * waited / max(n, 1)
            return
        for i in range(n):
            yield host[i % len(host)]

    pre = feed.DevicePrefetcher(stream(4 + 2 * args.depth), dev, depth=args.depth)
    slots = []
    for b in pre:                                         # warm-up + graph capture (one graph per device slot, each replayed once)
        tr.step({"v": b["v"], "q_idxes": b["q_idxes"]}, b["a"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # (the timed stream goes through the SAME prefetcher object's device slots: a fresh one would allocate new slots, i.e. new graphs)
    pre.it = iter(stream(args.steps))
    for b in pre:
        tr.step({"v": b["v"], "q_idxes": b["q_idxes"]}, b["a"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if stats:
        print("  the consumer waited %.2f ms per batch for the producer thread" % stats["producer_wait_ms_per_batch"])
    # raw H2D rate of one batch for reference
    d = torch.empty_like(host[0]["v"], device=dev)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        d.copy_(host[0]["v"], non_blocking=True)
    torch.cuda.synchronize()
    h2d = 10 * host[0]["v"].numel() * host[0]["v"].element_size() / (time.perf_counter() - t1) / 1e9
    print("host-fed (%s transport, %s%s): %.1f samples/s (%.3f ms/step, graph=%s); pinned H2D of v alone: %.1f GB/s"
          % (args.transport, "copied into the graph's buffers" if args.copy else "%d slots read in place" % args.depth,
             ", FeatureStore of %d images, %d gather threads" % (args.store, args.workers) if args.store else "", B * args.steps / dt, 1e3 * dt / args.steps, tr._graph is not None, h2d))


if __name__ == "__main__":
    main()
