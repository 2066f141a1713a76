#!/usr/bin/env python3
"""Per-layer cost of the ResNet-18 @ 64x64 convolutions at the bench batch (MIOpen find on): forward and backward-data
microseconds with torch events, 50 back-to-back launches each."""
import sys
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda:0"
layers = [("stem 7x7s2", 3, 64, 64, 7, 2, 3), ("l1 3x3", 64, 64, 16, 3, 1, 1), ("l2.0 3x3s2", 64, 128, 16, 3, 2, 1), ("l2 3x3", 128, 128, 8, 3, 1, 1),
          ("l2 ds1x1s2", 64, 128, 16, 1, 2, 0), ("l3.0 3x3s2", 128, 256, 8, 3, 2, 1), ("l3 3x3", 256, 256, 4, 3, 1, 1), ("l3 ds1x1s2", 128, 256, 8, 1, 2, 0),
          ("l4.0 3x3s2", 256, 512, 4, 3, 2, 1), ("l4 3x3", 512, 512, 2, 3, 1, 1), ("l4 ds1x1s2", 256, 512, 4, 1, 2, 0)]
counts = {"stem 7x7s2": 1, "l1 3x3": 4, "l2.0 3x3s2": 1, "l2 3x3": 3, "l2 ds1x1s2": 1, "l3.0 3x3s2": 1, "l3 3x3": 3, "l3 ds1x1s2": 1, "l4.0 3x3s2": 1,
          "l4 3x3": 3, "l4 ds1x1s2": 1}


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


tot_f = tot_b = 0.0
print("%-14s %9s %9s %8s   GFLOP" % ("layer", "fwd us", "bwdD us", "x count"))
for name, ci, co, hw, k, s, p in layers:
    x = torch.randn(B, ci, hw, hw, device=dev, requires_grad=True)
    w = torch.randn(co, ci, k, k, device=dev)
    y = F.conv2d(x, w, None, s, p)
    dy = torch.randn_like(y)
    tf = timeit(lambda: F.conv2d(x, w, None, s, p))
    tb = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, False, False]))
    gf = 2.0 * B * co * ci * k * k * y.shape[2] * y.shape[3] / 1e9
    print("%-14s %9.1f %9.1f %8d   %.2f" % (name, tf, tb, counts[name], gf))
    tot_f += tf * counts[name]
    tot_b += tb * counts[name]
print("per iteration: fwd %.0f us, bwd-data %.0f us" % (tot_f, tot_b))
