#!/usr/bin/env python3
"""train-mode BatchNorm with the kernel boundary as the exchange, priced on layer1's conv1 -> bn1 -> relu -> conv2 forward
(resnet.py:44-59; 64 channels, 16x16 maps, batch 100), graph-replayed back to back:

  A (the product path)   ee_wino3x3_f32 | ee_bn_act_fwd_f32(training, relu) | ee_wino3x3_f32                      3 launches
  B (kernel boundary)    ee_wino3x3_stats_f32 (epilogue: per-image plane moments) | ee_wino3x3_bn_train_pre_f32   2 launches
                         (prologue: merge of the 100 partials per channel, normalisation + ReLU while staging)

and the difference of the two results (the statistics are summed in another order: rounding level)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
sys.path.insert(0, ROOT)
import ctypes  # noqa: E402

import torch  # noqa: E402

from eeadv import _native as N, functional as Fn, ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
C, H = 64, 16
torch.manual_seed(0)
x = torch.relu(torch.randn(B, C, H, H, device=dev))
w1 = torch.randn(C, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
w2 = torch.randn(C, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
u1, u2 = Fn.wino_sets(w1)[0], Fn.wino_sets(w2)[0]
gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.2
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
y1 = torch.empty(B, C, H, H, device=dev)
y2 = torch.empty(B, C, H, H, device=dev)
stats = torch.empty(C, B, 2, device=dev)


def path_a():
    c1 = ops.wino3x3(x, u1)
    a1, _, _ = ops.bn_act_fwd(c1, None, gamma, beta, rm, rv, 0.1, 1e-5, True, True)
    return ops.wino3x3(a1, u2)


def path_b():
    N.check(N.lib.ee_wino3x3_stats_f32(x.data_ptr(), u1.data_ptr(), y1.data_ptr(), stats.data_ptr(), B, C, C, H, st()), "stats")
    N.check(N.lib.ee_wino3x3_bn_train_pre_f32(y1.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, u2.data_ptr(), y2.data_ptr(), B, C, C, H, st()), "pre")
    return y2


def path_convs_only():
    return ops.wino3x3(ops.wino3x3(x, u1), u2)


def timeit(fn, iters=40, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * a.elapsed_time(b) / iters)
    return best


ra, rb = path_a(), path_b().clone()
torch.cuda.synchronize()
print("B = %d, %d channels, %dx%d:  max |A - B| = %.3e of max |A| = %.3f" % (B, C, H, H, float((ra - rb).abs().max()), float(ra.abs().max())))
ta, tb, tc = timeit(path_a), timeit(path_b), timeit(path_convs_only)
print("A  conv | bn(train)+relu | conv   : %6.2f us  (3 launches)" % ta)
print("B  conv+moments | merge+bn+relu+conv: %6.2f us  (2 launches)" % tb)
print("   the two convolutions alone       : %6.2f us" % tc)
print("   -> BatchNorm costs %.2f us as a launch, %.2f us across the kernel boundary" % (ta - tc, tb - tc))
