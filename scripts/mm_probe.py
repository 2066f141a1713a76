#!/usr/bin/env python3
"""Is a 2x2-map 3x3 convolution cheaper as one dense product?  torch.mm timings at the layer-4 shape."""
import torch
dev = "cuda:0"
torch.backends.cudnn.benchmark = True


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n // 20):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / (n // 20 * 20)


for B in (100,):
    x = torch.randn(B, 2048, device=dev)
    w = torch.randn(2048, 2048, device=dev)
    print("mm   [%d,2048]x[2048,2048]      %.1f us" % (B, timeit(lambda: torch.mm(x, w))))
    print("mm   [%d,2048]x[2048,2048]^T    %.1f us" % (B, timeit(lambda: torch.mm(x, w.t()))))
    xc = torch.randn(B, 512, 2, 2, device=dev, requires_grad=True)
    wc = torch.randn(512, 512, 3, 3, device=dev)
    dy = torch.randn(B, 512, 2, 2, device=dev)
    print("conv 512->512 3x3 @2x2 fwd        %.1f us" % timeit(lambda: torch.nn.functional.conv2d(xc, wc, None, 1, 1)))
    print("conv bwd-data                     %.1f us" % timeit(lambda: torch.ops.aten.convolution_backward(dy, xc, wc, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])))
    x3 = torch.randn(B, 256, 4, 4, device=dev, requires_grad=True)
    w3 = torch.randn(256, 256, 3, 3, device=dev)
    print("conv 256->256 3x3 @4x4 fwd        %.1f us" % timeit(lambda: torch.nn.functional.conv2d(x3, w3, None, 1, 1)))
    xm = torch.randn(B, 4096, device=dev)
    wm = torch.randn(4096, 4096, device=dev)
    print("mm   [%d,4096]x[4096,4096]      %.1f us" % (B, timeit(lambda: torch.mm(xm, wm))))
