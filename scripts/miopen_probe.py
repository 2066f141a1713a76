#!/usr/bin/env python3
"""Un-profiled cost of the convolutions of ResNet-18 @ 64x64, B = 100 that still run on MIOpen (find on): forward and backward-data,
graph-replayed back-to-back launches (layout transposes and zero fills of the solver included - what the step really pays)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
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


LAYERS = (("stem 3->64 7x7 s2 @64", 3, 64, 64, 7, 2, 3), ("l2.0 64->128 3x3 s2 @16", 64, 128, 16, 3, 2, 1), ("l2 128 3x3 s1 @8", 128, 128, 8, 3, 1, 1),
          ("l3.0 128->256 3x3 s2 @8", 128, 256, 8, 3, 2, 1), ("l3 256 3x3 s1 @4", 256, 256, 4, 3, 1, 1), ("l4.0 256->512 3x3 s2 @4", 256, 512, 4, 3, 2, 1),
          ("l4 512 3x3 s1 @2", 512, 512, 2, 3, 1, 1))
print("%-28s %10s %10s %10s" % ("layer", "fwd us", "bwd-data", "wrw"))
for name, ci, co, hw, k, s, p in LAYERS:
    x = torch.randn(B, ci, hw, hw, device=dev, requires_grad=True)
    w = torch.randn(co, ci, k, k, device=dev, requires_grad=True)
    y = F.conv2d(x, w, None, s, p)
    dy = torch.randn_like(y)
    xd, wd = x.detach(), w.detach()
    t = [timeit(lambda: F.conv2d(xd, wd, None, s, p)),
         timeit(lambda: torch.ops.aten.convolution_backward(dy, xd, wd, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, False, False])),
         timeit(lambda: torch.ops.aten.convolution_backward(dy, xd, wd, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [False, True, False]))]
    print("%-28s %10.1f %10.1f %10.1f" % (name, t[0], t[1], t[2]), flush=True)
from eeadv import ops  # noqa: E402
x = torch.randn(B, 3, 64, 64, device=dev)
w = torch.randn(64, 3, 7, 7, device=dev)
print("%-28s %10.1f   (MIOpen %.1f)" % ("stem fwd on ee_conv.hip", timeit(lambda: ops.stem7x7s2_fwd(x, w)), timeit(lambda: F.conv2d(x, w, None, 2, 3))))
