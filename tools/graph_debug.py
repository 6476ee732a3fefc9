import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import torch
from vqa_playground_pytorch_amd import CoR2Model, ops
from vqa_playground_pytorch_amd.trainer import DataParallelTrainer, kld_sum_loss
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(os.environ.get("B", "8"))
model = CoR2Model(["PAD"], 300).to(dev)
if os.environ.get("EVAL"): model.eval()
tr = DataParallelTrainer(model, graph=False)
v = torch.randn(B, 36, 2048, device=dev); q = torch.randn(B, 2400, device=dev); a = torch.softmax(torch.randn(B, 300, device=dev), 1)
s = {"v": v, "q_idxes": q}
for i in range(3):
    tr._set_step_scalars(); tr._front(s, a); tr._tail()
torch.cuda.synchronize(); print("eager ok", flush=True)
stage = os.environ.get("STAGE", "all")
g1 = torch.cuda.CUDAGraph()
pool = torch.cuda.graph_pool_handle()
if stage in ("fwd",):
    with torch.no_grad():
        with torch.cuda.graph(g1, pool=pool):
            out = model(s)
    print("captured fwd", flush=True); g1.replay(); torch.cuda.synchronize(); print("replayed fwd", flush=True); sys.exit(0)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    tr._front(s, a)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
with torch.cuda.graph(g1, pool=pool):
    loss = tr._front(s, a)
print("captured front", flush=True)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, pool=pool):
    tr._tail()
print("captured tail", flush=True)
for i in range(3):
    tr._set_step_scalars(); g1.replay(); g2.replay()
torch.cuda.synchronize(); print("replayed", loss.item(), flush=True)
