#!/usr/bin/env python3
"""Two data-parallel ranks on ONE GPU (gloo with device tensors): exercises the multi-rank step -- per-rank shards, the
flat-gradient SUM all-reduce between the replayed hipGraphs, the two-half backward with the first all-reduce under the
second half -- on a single-GPU box.  Rank 0 prints ONE JSON line with the global losses / gradient norms of 7 steps.

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/dp2_one_gpu.py
    BACKEND=nccl python -m torch.distributed.run --nproc-per-node 2 ... tools/dp2_one_gpu.py    # rank r on cuda:r, RCCL across them
    python tools/dp2_one_gpu.py            # the same steps in one process on the whole batch (the expected record)
    NCCL1=1 python tools/dp2_one_gpu.py    # one process, a ONE-rank "nccl" (= RCCL) group, the all-reduce forced
  env: G=1|0 graph replay or kernel by kernel;  VQA_DP_OVERLAP=1: backward in two halves (forced at world size 1);  MODEL=cor2|oda;
       SLOTS=k: the batch lives in k resident copies visited in turn, the trainer captures one forward + backward graph per copy
       (input_slots, bench.py's rotating batches) -- same numbers as SLOTS=1
(tests/test_gpu_dp2.py runs all combinations and compares them with the single-process record; tests/test_gpu_nccl1.py
runs the single-process steps with the all-reduce issued through a one-rank RCCL communicator and demands bit-identical
results: RCCL init, the collective between the replayed graphs, the async work handle beside the second backward graph.)"""
import faulthandler
import json
import os
import sys

if os.environ.get("VQA_DUMP_AFTER"):       # debugging aid: every thread's stack after N seconds, then exit (a hang shows where)
    faulthandler.dump_traceback_later(int(os.environ["VQA_DUMP_AFTER"]), exit=True)

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_playground_pytorch_amd import CoR2Model, ODAModel  # noqa: E402
from vqa_playground_pytorch_amd.trainer import DataParallelTrainer  # noqa: E402

multi = "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
# BACKEND=gloo (default): every rank on cuda:0, the collectives on gloo -- runs on a one-GPU box;
# BACKEND=nccl: rank r on cuda:r, the collectives on RCCL across the devices (needs >= world GPUs: the xGMI path itself)
backend = os.environ.get("BACKEND", "gloo")
local = int(os.environ.get("LOCAL_RANK", "0")) if (multi and backend == "nccl") else 0
dev = torch.device("cuda", local)
torch.cuda.set_device(local)
if multi:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
rank = dist.get_rank() if multi else 0
nccl1 = not multi and os.environ.get("NCCL1") == "1"
if nccl1:
    import socket
    with socket.socket() as _s:
        _s.bind(("127.0.0.1", 0))
        _port = _s.getsockname()[1]
    os.environ["VQA_FORCE_ALLREDUCE"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _port, rank=0, world_size=1, device_id=dev)
torch.manual_seed(100 + rank)              # different initial weights per rank: the trainer broadcasts rank 0's
cls = {"cor2": CoR2Model, "oda": ODAModel}[os.environ.get("MODEL", "cor2")]
if rank == 0:
    torch.manual_seed(100)
model = cls(["PAD"], 300).to(dev).eval()    # eval mode: no dropout, so N ranks on shards == 1 process on the batch
split = os.environ.get("VQA_DP_OVERLAP", "0") == "1"
slots = int(os.environ.get("SLOTS", "1"))
tr = DataParallelTrainer(model, lr=1e-4, graph=os.environ.get("G", "1") == "1",
                         overlap=("force" if not multi else True) if split else False, adopt_inputs=slots > 1, input_slots=slots)
torch.manual_seed(5)
v, q = torch.randn(8, 36, 2048, device=dev), torch.randn(8, 2400, device=dev)
a = torch.softmax(torch.randn(8, 300, device=dev), 1)
ring = [({"v": tr.shard(v).clone(), "q_idxes": tr.shard(q).clone()}, tr.shard(a).clone()) for _ in range(slots)]
losses, norms = [], []
for i in range(7):
    loss, norm = tr.step(*ring[i % slots])
    t = loss.clone()
    if multi:
        dist.all_reduce(t)
    losses.append(t.item())
    norms.append(norm.item())
w = torch.cat([p.detach().reshape(-1)[:100] for p in model.parameters()])
if multi:
    w0 = w.clone()
    dist.broadcast(w0, 0)
    assert torch.equal(w, w0), "replicas diverged"
if rank == 0:
    print(json.dumps({"world": dist.get_world_size() if multi else 1, "graph": tr._graph is not None, "overlap": bool(tr.overlap),
                      "reduce": bool(tr.reduce), "backend": dist.get_backend() if dist.is_initialized() else None,
                      "graphs": sorted(k for k in (tr._graph or {}) if k in ("front", "front_a", "front_b", "tail")),
                      "slots": len(getattr(tr, "_slots", [])),
                      "losses": losses, "norms": norms, "weight_digest": float(w.double().sum().item())}), flush=True)
if multi or nccl1:
    dist.barrier()
    dist.destroy_process_group()
