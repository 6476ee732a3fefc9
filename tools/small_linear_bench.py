import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from vqa_playground_pytorch_amd import ops
dev = torch.device("cuda:0")
def timeit(f, n=30, inner=10):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(n):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(2_000_000); a.record()
        for _ in range(inner): f()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b)/inner*1e3)
    return statistics.median(ts)
for (M,K,N) in [(512,2400,310),(512,310,2048),(512,2048,156),(512,510,2000),(512,310,510)]:
    x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)/K**0.5; b=torch.randn(N,device=dev)
    gy=torch.randn(M,N,device=dev)
    xr=x.clone().requires_grad_(); wr=w.clone().requires_grad_(); br=b.clone().requires_grad_()
    t_f=timeit(lambda: ops.LinearAct.apply(x,w,b,1,0.5,3))
    y=ops.LinearAct.apply(xr,wr,br,1,0.5,3)
    t_b=timeit(lambda: torch.autograd.grad(y,[xr,wr,br],gy,retain_graph=True))
    lin=lambda: torch.relu(torch.nn.functional.linear(torch.nn.functional.dropout(x,0.5,True),w,b))
    t_tf=timeit(lin)
    yt=torch.relu(torch.nn.functional.linear(torch.nn.functional.dropout(xr,0.5,True),wr,br))
    t_tb=timeit(lambda: torch.autograd.grad(yt,[xr,wr,br],gy,retain_graph=True))
    print("M=%d K=%d N=%d  mine fwd %6.1f us bwd %6.1f us | torch fwd %6.1f us bwd %6.1f us"%(M,K,N,t_f,t_b,t_tf,t_tb))
