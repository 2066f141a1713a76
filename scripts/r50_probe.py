#!/usr/bin/env python3
"""BASELINE config 5 (ResNet-50 @ 224 x 224, per-rank batch 32): un-profiled cost of every distinct convolution of the network on MIOpen
(find on) - forward, backward-data, weight gradient, NCHW and channels_last - in graph-replayed back-to-back launches (the solver's layout
transposes and zero fills included), next to the three ways of running the stride-2 1x1 shortcut convolutions.  Multiplicity = how
often the shape occurs in the network, so that `total` is what a repeat of free-AT pays for its convolutions.

    python3 scripts/r50_probe.py [batch]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda:0"


def timeit(fn, iters=10, reps=3):
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


# (name, cin, cout, input hw, k, stride, pad, multiplicity) - models.Bottleneck: 1x1, 3x3 (carries the stride), 1x1 x4; [3, 4, 6, 3] blocks
LAYERS = [("stem 7x7 s2 3->64 @224", 3, 64, 224, 7, 2, 3, 1)]
inp, hw = 64, 56
for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), 1):
    out_hw = hw // stride
    LAYERS.append(("l%d.0 1x1 %d->%d @%d" % (li, inp, planes, hw), inp, planes, hw, 1, 1, 0, 1))
    LAYERS.append(("l%d.0 3x3 s%d %d @%d" % (li, stride, planes, hw), planes, planes, hw, 3, stride, 1, 1))
    LAYERS.append(("l%d.0 ds 1x1 s%d %d->%d @%d" % (li, stride, inp, 4 * planes, hw), inp, 4 * planes, hw, 1, stride, 0, 1))
    LAYERS.append(("l%d 1x1 %d->%d @%d" % (li, planes, 4 * planes, out_hw), planes, 4 * planes, out_hw, 1, 1, 0, blocks))
    LAYERS.append(("l%d 1x1 %d->%d @%d" % (li, 4 * planes, planes, out_hw), 4 * planes, planes, out_hw, 1, 1, 0, blocks - 1))
    LAYERS.append(("l%d 3x3 s1 %d @%d" % (li, planes, out_hw), planes, planes, out_hw, 3, 1, 1, blocks - 1))
    inp, hw = 4 * planes, out_hw

only = os.environ.get("R50_ONLY")
tot = {"nchw": [0.0, 0.0, 0.0], "cl": [0.0, 0.0, 0.0]}
flops_tot = 0.0
print("B = %d; us per launch: NCHW fwd / bwd-data / wrw | channels_last fwd / bwd-data / wrw | GFLOP (one direction) | TFLOP/s NCHW fwd bwd wrw" % B)
for name, ci, co, h, k, s, p, mult in LAYERS:
    if only and only not in name:
        continue
    res = {}
    for fmt in ("nchw", "cl"):
        mf = torch.channels_last if fmt == "cl" else torch.contiguous_format
        x = torch.randn(B, ci, h, h, device=dev).contiguous(memory_format=mf)
        w = torch.randn(co, ci, k, k, device=dev).contiguous(memory_format=mf)
        y = F.conv2d(x, w, None, s, p)
        dy = torch.randn_like(y)
        need_dx = ci > 3
        t = [timeit(lambda: F.conv2d(x, w, None, s, p)),
             timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, False, False])) if need_dx else 0.0,
             timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [False, True, False]))]
        res[fmt] = t
        for i in range(3):
            tot[fmt][i] += mult * t[i]
    oh = (h + 2 * p - k) // s + 1
    gf = 2.0 * ci * co * k * k * oh * oh * B / 1e9
    flops_tot += mult * gf
    a, c = res["nchw"], res["cl"]
    tf = lambda us: gf / us * 1e3 if us else 0.0
    print("%-30s x%d  %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f | %7.2f | %5.1f %5.1f %5.1f" % (name, mult, a[0], a[1], a[2], c[0], c[1], c[2], gf, tf(a[0]), tf(a[1]), tf(a[2])), flush=True)
print("total per pass (us): NCHW fwd %.0f bwd-data %.0f wrw %.0f = %.0f | channels_last fwd %.0f bwd-data %.0f wrw %.0f = %.0f | %.1f GFLOP per direction" % (
    tot["nchw"][0], tot["nchw"][1], tot["nchw"][2], sum(tot["nchw"]), tot["cl"][0], tot["cl"][1], tot["cl"][2], sum(tot["cl"]), flops_tot))

# ---- the stride-2 1x1 shortcut convolutions: ee_conv.hip's kernel (built for 16 / 8 / 4-wide maps) against MIOpen and against
# "subsample, then a stride-1 1x1 convolution" (a strided copy + a plain GEMM)
from eeadv import ops  # noqa: E402
print("stride-2 1x1 shortcut: ee_conv.hip fwd / bwd | MIOpen fwd / bwd | subsample + 1x1 fwd")
for ci, co, h in ((256, 512, 56), (512, 1024, 28), (1024, 2048, 14)):
    x = torch.randn(B, ci, h, h, device=dev)
    w = torch.randn(co, ci, 1, 1, device=dev)
    dy = torch.randn(B, co, h // 2, h // 2, device=dev)
    t = [timeit(lambda: ops.conv1x1s2_fwd(x, w)), timeit(lambda: ops.conv1x1s2_bwd(dy, w, h, h)),
         timeit(lambda: F.conv2d(x, w, None, 2, 0)),
         timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])),
         timeit(lambda: F.conv2d(x[:, :, ::2, ::2].contiguous(), w))]
    print("%4d->%4d @%2d   %8.1f %8.1f | %8.1f %8.1f | %8.1f" % (ci, co, h, *t), flush=True)
