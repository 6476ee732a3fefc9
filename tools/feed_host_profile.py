#!/usr/bin/env python3
"""Host side of the feed alone (no GPU work): how long does feed.store_batches take to assemble one 512-sample batch from a
memory-mapped FeatureStore into pinned staging?   python tools/feed_host_profile.py [workers] [n_img]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vqa_playground_pytorch_amd import feed  # noqa: E402


def main():
    if os.environ.get("TORCH_THREADS"):
        torch.set_num_threads(int(os.environ["TORCH_THREADS"]))
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    path = "/tmp/feed_profile_store_%d.npy" % n_img
    if not os.path.exists(path):
        mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(n_img, 36, 2048))
        mm[:] = 1.0
        mm.flush()
        del mm
    st = feed.FeatureStore(path, workers=workers)
    if os.environ.get("POPULATE", "1") == "1":
        t = time.perf_counter()
        st.populate()
        print("populate: %.0f ms" % (1e3 * (time.perf_counter() - t)))
    rs = np.random.RandomState(0)
    pin = torch.cuda.is_available()
    B = 512
    qa = [{"v_idx": int(rs.randint(n_img)), "q_idxes": rs.standard_normal(2400).astype(np.float32), "q_id": k,
           "a_10_idx": [(int(c), 0.1) for c in rs.choice(2000, 10, replace=False)]} for k in range(B * 6)]
    table = feed.qa_table(st, qa, 2000, q_dtype=torch.float32)
    print("torch threads %d, cpu count %d, pinned staging %s, %d gather threads" % (torch.get_num_threads(), os.cpu_count(), pin, workers))
    for dt in (torch.float32, torch.bfloat16):
        out = torch.empty(B, 36, 2048, dtype=dt)
        if pin:
            out = out.pin_memory()
        idx = rs.randint(0, n_img, B)
        ts = []
        for _ in range(12):
            t = time.perf_counter()
            st.gather(idx, out)
            ts.append(1e3 * (time.perf_counter() - t))
        print("gather %-8s ms per call: %s" % (str(dt).split(".")[1], " ".join("%.1f" % x for x in ts)))
        ts = []
        t = time.perf_counter()
        for b in feed.store_batches(st, table, B, 2000, pin=pin, region_dtype=dt, q_dtype=torch.float32, ring=3):
            ts.append(1e3 * (time.perf_counter() - t))
            t = time.perf_counter()
        print("store_batches %-8s ms per batch: %s" % (str(dt).split(".")[1], " ".join("%.1f" % x for x in ts)))


if __name__ == "__main__":
    main()
