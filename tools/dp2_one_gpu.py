# two ranks on ONE GPU with gloo (CUDA tensors): exercises graph replay + the all-reduce(s) between the graphs.
#   G=1|0 graph replay or eager;  VQA_DP_OVERLAP=1: backward in two halves, first all-reduce under the second half
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from vqa_playground_pytorch_amd import CoR2Model
from vqa_playground_pytorch_amd.trainer import DataParallelTrainer
dist.init_process_group("gloo")
rank = dist.get_rank()
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
torch.manual_seed(rank)
model = CoR2Model(["PAD"], 300).to(dev).eval()
tr = DataParallelTrainer(model, lr=1e-4, graph=os.environ.get("G", "1") == "1")
torch.manual_seed(5)
v = torch.randn(8, 36, 2048, device=dev); q = torch.randn(8, 2400, device=dev); a = torch.softmax(torch.randn(8, 300, device=dev), 1)
out = []
for i in range(7):
    loss, norm = tr.step({"v": tr.shard(v), "q_idxes": tr.shard(q)}, tr.shard(a))
    t = loss.clone(); dist.all_reduce(t)
    out.append((round(t.item(), 4), round(norm.item(), 3)))
if rank == 0: print(("graph" if tr._graph is not None else "eager") + (" overlap" if tr.overlap else ""), out, flush=True)
w = torch.cat([p.detach().reshape(-1)[:100] for p in model.parameters()])
w2 = w.clone(); dist.broadcast(w2, 0)
assert torch.equal(w, w2), "replicas diverged"
dist.barrier(); dist.destroy_process_group()
