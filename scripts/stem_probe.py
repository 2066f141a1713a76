#!/usr/bin/env python3
"""Un-profiled cost of the stem's pieces at the bench shape (B = 100, 64 x 64 images): convolution forward / backward-data, fused
BatchNorm + ReLU + MaxPool forward / backward against the separate kernels; graph-replayed back-to-back launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402

from eeadv import ops  # noqa: E402

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


img = torch.rand(B, 3, 64, 64, device=dev)
w = torch.randn(64, 3, 7, 7, device=dev) / 12
x = ops.stem7x7s2_fwd(img, w)
gamma, beta = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
y, code, sm, si = ops.bn_relu_pool_fwd(x, gamma, beta, rm, rv, 0.1, 1e-5, True)
dyp = torch.randn_like(y)
yfull, sm2, si2 = ops.bn_act_fwd(x, None, gamma, beta, rm, rv, 0.1, 1e-5, True, True)
dfull = torch.randn_like(yfull)
rows = [
    ("stem conv fwd (MFMA)", lambda: ops.stem7x7s2_fwd(img, w)),
    ("stem conv bwd-data (MFMA)", lambda: ops.stem7x7s2_bwd_data(dfull, w, 64, 64)),
    ("bn+relu+pool fwd, fused (stats + apply/pool)", lambda: ops.bn_relu_pool_fwd(x, gamma, beta, rm, rv, 0.1, 1e-5, True)),
    ("  eval mode (apply/pool only)", lambda: ops.bn_relu_pool_fwd(x, gamma, beta, rm, rv, 0.1, 1e-5, False)),
    ("bn+relu fwd, separate", lambda: ops.bn_act_fwd(x, None, gamma, beta, rm, rv, 0.1, 1e-5, True, True)),
    ("maxpool fwd, separate", lambda: ops.maxpool3s2_fwd(yfull)),
    ("bn+relu+pool bwd, fused (partial + apply)", lambda: ops.bn_relu_pool_bwd(dyp, code, x, gamma, beta, sm, si, None, None, 1e-5, True, True, False)),
    ("  partial only (no dx)", lambda: ops.bn_relu_pool_bwd(dyp, code, x, gamma, beta, sm, si, None, None, 1e-5, True, False, True)),
    ("maxpool bwd, separate", lambda: ops.maxpool3s2_bwd(dyp, code, 32, 32)),
    ("bn+relu bwd, separate", lambda: ops.bn_act_bwd(dfull, yfull, x, gamma, sm2, si2, None, None, 1e-5, True, True, True, False, False)),
]
for name, fn in rows:
    print("%-50s %8.1f us" % (name, timeit(fn)), flush=True)
print("%-50s %8.1f us" % ("stem conv fwd + BatchNorm moments in the epilogue", timeit(lambda: ops.stem7x7s2_fwd(img, w, True))), flush=True)
_, st = ops.stem7x7s2_fwd(img, w, True)
print("%-50s %8.1f us" % ("bn+relu+pool fwd, fused, moments from the conv", timeit(lambda: ops.bn_relu_pool_fwd(x, gamma, beta, rm, rv, 0.1, 1e-5, True, st))), flush=True)
