import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "edge-enhancement_amd"))
import torch
from eeadv import ops
dev="cuda:0"
x=torch.rand(100,3,64,64,device=dev); g=torch.randn_like(x); x0=x.clone()
xh=torch.rand_like(x); wts=ops.EdgeWeights(1.0)
_,gate,_=ops.frontend_fwd(x,xh,wts,0.0,76/255,1.0)
def bracket(fn,n=200):
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a,b in ev:
        a.record(); fn(); b.record()
        torch.cuda.synchronize()   # isolate: like an eager step where the stream is not backed up
    return sum(a.elapsed_time(b) for a,b in ev)/n*1e3
def bracket_busy(fn,n=200):
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a,b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a,b in ev)/n*1e3
for name,fn in [("empty",lambda:None),("pgd_step",lambda:ops.pgd_step_(x,g,x0,2/255,16/255)),("frontend_bwd",lambda:ops.frontend_bwd(g,gate,x,wts,0.0,76/255,1.0)),("frontend_fwd",lambda:ops.frontend_fwd(x,xh,wts,0.0,76/255,1.0))]:
    print(name, "isolated %.2f us   back-to-back %.2f us"%(bracket(fn), bracket_busy(fn)))
