#!/usr/bin/env python3
"""Weight gradient of ResNet-50's 1x1 / stride 1 convolutions at 224x224, batch 32: ee_wrw.hip's NCHW product (+ its fixed-order sum) against
ATen / MIOpen (find on; its layout transposes and zero fills included), graph-replayed back-to-back launches.
    python3 scripts/wrw1x1_probe.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402

from eeadv import ops  # noqa: E402

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda:0"


def timeit(fn, iters=10, reps=5):
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


print("B = %d; us per weight gradient: ee_wrw.hip | ATen/MIOpen | GFLOP | TFLOP/s (ee_wrw)" % B)
tot = [0.0, 0.0]
own_only = os.environ.get("PROBE_OWN_ONLY") == "1"
for ci, co, hw, mult in ((64, 64, 56, 1), (64, 256, 56, 4), (256, 64, 56, 2), (256, 128, 56, 1), (128, 512, 28, 4), (512, 128, 28, 3), (512, 256, 28, 1),
                         (256, 1024, 14, 6), (1024, 256, 14, 5), (1024, 512, 14, 1)):
    x = torch.randn(B, ci, hw, hw, device=dev)
    dy = torch.randn(B, co, hw, hw, device=dev)
    w = torch.zeros(co, ci, 1, 1, device=dev)
    t_own = timeit(lambda: ops.wrw1x1(x, dy))
    t_ref = 0.0 if own_only else timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
    gf = 2.0 * ci * co * B * hw * hw / 1e9
    tot[0] += mult * t_own
    tot[1] += mult * t_ref
    print("%4d -> %4d @%2d x%d  %8.1f %8.1f   %.2f  %6.1f" % (ci, co, hw, mult, t_own, t_ref, gf, gf / t_own * 1e3), flush=True)
print("per repeat (28 layers): ee_wrw %.0f us, MIOpen %.0f us" % tuple(tot))
