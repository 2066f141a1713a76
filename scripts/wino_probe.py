#!/usr/bin/env python3
"""Un-profiled cost of the 3x3 convolution on 128-channel 8x8 maps (ResNet-18 layer2, B = 100): MIOpen (Winograd on the vector ALUs), the
direct MFMA kernel of ee_conv.hip, Winograd F(2x2,3x3) on the matrix cores (ee_wino.hip); graph-replayed back-to-back launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from eeadv import functional as EF, ops  # noqa: E402

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda:0"


def timeit(fn, iters=30, reps=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("PROBE_EAGER"):  # plain launches, for counter collection
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        return 0.0
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


SHAPES = ((128, 8), (64, 16), (256, 4)) if os.environ.get("PROBE_EAGER") else ((128, 8), (64, 16), (256, 4), (64, 8), (256, 8))  # counters: the three shapes of the bench
for c, hw in SHAPES:
    x = torch.randn(B, c, hw, hw, device=dev)
    w = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
    u = EF._rearranged(w, "wino_f").contiguous()
    if os.environ.get("PROBE_EAGER"):
        timeit(lambda: ops.wino3x3(x, u))
        continue
    print("%3d ch %2dx%-2d:  MIOpen %6.1f us   Winograd MFMA %6.1f us" % (
        c, hw, hw, timeit(lambda: F.conv2d(x, w, None, 1, 1)), timeit(lambda: ops.wino3x3(x, u))), flush=True)
