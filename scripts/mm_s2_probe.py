#!/usr/bin/env python3
"""Would layer4.0's stride-2 convolution (256 -> 512, 4x4 -> 2x2) pay as a dense product like its stride-1 neighbours?  Un-profiled,
graph-replayed: [B,16 Cin] x [16 Cin, 4 Cout] (forward) and [B, 4 Cout] x [4 Cout, 16 Cin] (backward-data) on the BLAS."""
import sys

import torch

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


for name, ci, co, hw in (("l4.0 256->512 4x4->2x2", 256, 512, 4), ("l3.0 128->256 8x8->4x4", 128, 256, 8)):
    k, n = ci * hw * hw, co * (hw // 2) ** 2
    x = torch.randn(B, k, device=dev)
    w2 = torch.randn(k, n, device=dev)
    dy = torch.randn(B, n, device=dev)
    print("%-26s W2 %5.1f MB   fwd %6.1f us   bwd-data %6.1f us" % (name, k * n * 4 / 1e6, timeit(lambda: torch.mm(x, w2)), timeit(lambda: torch.mm(dy, w2.t()))), flush=True)
