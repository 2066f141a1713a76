#!/usr/bin/env python3
"""ee_dense.hip against the Tensile product it replaces (layer4 of ResNet-18 at 64x64 inputs: 100 x 2048 x 2048), graph-replayed back to back,
forward and backward-data, plain and with the eval-mode BatchNorm folded in (against product + ee_bn_act_* launches)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from eeadv import functional as Fn, ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
C = 512
torch.manual_seed(0)
x = torch.randn(B, C, 2, 2, device=dev)
w = torch.randn(C, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
w2, w2t = Fn._dense_weight(w, "s1"), Fn._dense_weight(w, "s1t")
g, b, rm, rv = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
out = torch.relu(torch.randn(B, C, 2, 2, device=dev))
dy = torch.randn(B, C, 2, 2, device=dev)
zero = torch.zeros_like(dy)


def timeit(fn, iters=40, reps=5):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters):
            fn()
    gr.replay()
    best = 1e9
    for _ in range(reps):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        gr.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * a.elapsed_time(e) / iters)
    return best


rows = [
    ("torch.mm (Tensile)                          ", lambda: torch.mm(x.reshape(B, -1), w2)),
    ("ee_dense2x2_f32                             ", lambda: ops.dense2x2(x, w2)),
    ("torch.mm + ee_bn_act_fwd(eval, relu)        ", lambda: ops.bn_act_fwd(torch.mm(x.reshape(B, -1), w2).view(B, C, 2, 2), None, g, b, rm, rv, 0.1, 1e-5, False, True)),
    ("ee_dense2x2_bn_eval_fwd_f32                 ", lambda: ops.dense2x2_bn_eval_fwd(x, w2, (rm, rv, g, b, 1e-5), None, True)),
    ("ee_bn_act_bwd(eval) + torch.mm (transposed) ", lambda: torch.mm(ops.bn_act_bwd(dy, out, zero, g, None, None, rm, rv, 1e-5, False, True, True, True, False)[0].reshape(B, -1), w2.t())),
    ("ee_dense2x2_bn_eval_bwd_f32                 ", lambda: ops.dense2x2_bn_eval_bwd(dy, None, out, w2t, (rv, g, 1e-5), True)),
]
print("B = %d, %d -> %d channels on a 2x2 map (%d x %d x %d product, %.2f GFLOP)" % (B, C, C, B, 4 * C, 4 * C, 2e-9 * B * 16 * C * C))
for name, fn in rows:
    us = timeit(fn)
    print("%s %7.2f us  %6.1f TFLOP/s" % (name, us, 2e-6 * B * 16 * C * C / us))
