#!/usr/bin/env python3
"""Weight gradient of the stride-1 3x3 layers of ResNet-18 at 64x64 inputs: ee_wrw.hip (Winograd F(3x3,2x2) + fixed-order reduce) against
ATen / MIOpen (find on; its layout transposes and zero fills included), graph-replayed back-to-back launches.
    python3 scripts/wrw_probe.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402

from eeadv import ops  # noqa: E402

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda:0"


def timeit(fn, iters=10, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("PROBE_EAGER") == "1":
        for _ in range(iters):
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


print("B = %d; us per weight gradient: ee_wrw.hip | ATen/MIOpen | algorithmic GFLOP | TFLOP/s (ee_wrw)" % B)
tot = [0.0, 0.0]
for C, H, mult in ((64, 16, 4), (128, 8, 3), (256, 4, 3), (512, 2, 3)):
    x = torch.randn(B, C, H, H, device=dev)
    dy = torch.randn(B, C, H, H, device=dev)
    w = torch.zeros(C, C, 3, 3, device=dev)
    t_own = timeit(lambda: ops.wrw3x3(x, dy))
    t_ref = 0.0 if os.environ.get("PROBE_OWN_ONLY") == "1" else timeit(
        lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    gf = 2.0 * 9 * C * C * B * H * H / 1e9
    tot[0] += mult * t_own
    tot[1] += mult * t_ref
    print("%3d ch %2dx%-2d x%d  %8.1f %8.1f   %.2f  %6.1f" % (C, H, H, mult, t_own, t_ref, gf, gf / t_own * 1e3 if t_own else 0), flush=True)
print("per update (13 layers): ee_wrw %.0f us, MIOpen %.0f us" % tuple(tot))
