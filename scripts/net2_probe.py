#!/usr/bin/env python3
"""Un-profiled cost of Net_2's convolutional half (B = 50): the two fused launches each way (ee_net2.hip) against the stock ATen / MIOpen
sequence; graph-replayed back-to-back launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from eeadv import ops  # noqa: E402

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = "cuda:0"


def timeit(fn, iters=30, reps=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / (iters * reps)


x = torch.rand(B, 1, 28, 28, device=dev)
w1, b1 = torch.randn(32, 1, 5, 5, device=dev) * 0.2, torch.randn(32, device=dev) * 0.1
w2, b2 = torch.randn(64, 32, 5, 5, device=dev) * 0.05, torch.randn(64, device=dev) * 0.1
draw = (torch.rand(B, 64, device=dev) < 0.5).float()
drop = draw * 2
a2, saved, _ = ops.net2_conv_fwd(x, w1, b1, w2, b2, draw, 0.5)
da2 = torch.randn_like(a2)


def stock_fwd():
    h = F.relu(F.max_pool2d(F.conv2d(x, w1, b1), 2))
    return F.relu(F.max_pool2d(F.conv2d(h, w2, b2) * drop.view(B, 64, 1, 1), 2))


xr = x.clone().requires_grad_(True)


def stock_both():
    h = F.relu(F.max_pool2d(F.conv2d(xr, w1, b1), 2))
    h = F.relu(F.max_pool2d(F.conv2d(h, w2, b2) * drop.view(B, 64, 1, 1), 2))
    return torch.autograd.grad(h, [xr], da2)


print("%-44s %8.1f us" % ("fused forward (2 launches)", timeit(lambda: ops.net2_conv_fwd(x, w1, b1, w2, b2, draw, 0.5))))
print("%-44s %8.1f us" % ("fused backward (2 launches)", timeit(lambda: ops.net2_conv_bwd(da2, a2, saved, w1, w2, draw, 0.5))))
print("%-44s %8.1f us" % ("stock forward", timeit(stock_fwd)))
print("%-44s %8.1f us" % ("stock forward + backward", timeit(stock_both)))
