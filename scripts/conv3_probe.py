#!/usr/bin/env python3
"""Un-profiled cost of the 3x3 stride-1 convolutions of ResNet-18 @ 64x64, B = 100: MIOpen (find on) against ee_conv3x3s1 (MFMA implicit
GEMM), forward and backward-data, graph-replayed back-to-back launches (no host gaps, no profiler)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from eeadv import ops  # noqa: E402

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


print("%-12s %10s %10s %10s %10s   TFLOP/s(best fwd)" % ("layer", "miopen fwd", "ours fwd", "miopen bwd", "ours bwd"))
LAYERS = (("l1 64ch 16x16", 64, 16), ("l2 128ch 8x8", 128, 8), ("l3 256ch 4x4", 256, 4))
if os.environ.get("PROBE_LAYER"):
    LAYERS = tuple(l for l in LAYERS if l[0].startswith(os.environ["PROBE_LAYER"]))
for name, c, hw in LAYERS:
    x = torch.randn(B, c, hw, hw, device=dev, requires_grad=True)
    w = torch.randn(c, c, 3, 3, device=dev)
    dy = torch.randn(B, c, hw, hw, device=dev)
    xd = x.detach()
    t = [timeit(lambda: F.conv2d(xd, w, None, 1, 1)), timeit(lambda: ops.conv3x3s1_fwd(xd, w)),
         timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])),
         timeit(lambda: ops.conv3x3s1_bwd_data(dy, w))]
    gf = 2.0 * B * c * c * 9 * hw * hw / 1e9
    print("%-12s %10.1f %10.1f %10.1f %10.1f   %.1f" % (name, t[0], t[1], t[2], t[3], gf / max(min(t[0], t[1]), 1e-9)))
