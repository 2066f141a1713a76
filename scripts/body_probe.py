#!/usr/bin/env python3
"""Un-profiled cost of ONE pass of the classifier inside the attack loop (resnet18 @ 64x64, B = 100, train-mode BatchNorm, gradient with
respect to the image only), graph-replayed: the number the per-kernel probes have to add up to.  Run under different EEADV_STOCK_GLUE /
EEADV_* settings to A/B a kernel IN PLACE (cold caches, real neighbours), e.g. EEADV_STOCK_GLUE=s2small."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from eeadv import functional as EF, models as M  # noqa: E402

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda:0"
torch.manual_seed(0)
model = M.make_resnet(18, "tiny").to(dev).train()
x = torch.rand(B, 3, 64, 64, device=dev, requires_grad=True)
y = torch.randint(0, 200, (B,), device=dev)


def one():
    with EF.input_grad_only():
        loss = F.cross_entropy(model(x), y)
        (g,) = torch.autograd.grad(loss, [x])
    return g


for _ in range(3):
    one()
torch.cuda.synchronize()
iters = 10
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    one()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            one()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * a.elapsed_time(b) / (5 * iters))
print("EEADV_STOCK_GLUE=%-12s one forward + input gradient: %.1f us" % (os.environ.get("EEADV_STOCK_GLUE", ""), best), flush=True)
