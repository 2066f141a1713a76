#!/usr/bin/env python3
"""Round 4 pilot: layer4's dense product ([100 x 2048] . [2048 x 2048], ee_dense.hip's shape) on the BF16 matrix cores with three-piece operands
(ee_bf16x3.hip) beside the f32 matrix-core kernels (ours, Tensile): time (graph-replayed back to back) and error against float64."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402

from eeadv import ops  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, iters=30, reps=4):
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


for M, K, Nn in ((100, 2048, 2048), (100, 1024, 2048), (512, 2048, 2048)):
    x = torch.randn(M, K, device=dev)
    x[x.abs() < 0.3] = 0.0  # post-ReLU-like sparsity does not matter to the arithmetic; keep some exact zeros
    w = torch.randn(K, Nn, device=dev) / K ** 0.5
    wt = w.t().contiguous()
    ref = x.double() @ w.double()
    scale = float(ref.abs().max())
    x3, w3 = ops.split_bf16x3(x), ops.split_bf16x3(wt)
    assert float((x3[0].double() + x3[1].double() + x3[2].double() - x.double()).abs().max()) == 0.0, "the split is not exact"
    c3 = ops.gemm_bf16x3_nt(x3, w3)
    cf = x @ w
    cd = ops.dense2x2(x.view(M, K // 4, 2, 2), w).view(M, Nn) if K % 512 == 0 else None
    print(f"[{M} x {K}] . [{K} x {Nn}]   max |err| / max |ref| against float64:")
    print(f"    bf16x3 (6 terms)      {float((c3.double() - ref).abs().max()) / scale:.3e}   mean {float((c3.double() - ref).abs().mean()) / scale:.3e}")
    print(f"    f32 MFMA (Tensile)    {float((cf.double() - ref).abs().max()) / scale:.3e}   mean {float((cf.double() - ref).abs().mean()) / scale:.3e}")
    if cd is not None:
        print(f"    f32 MFMA (ee_dense)   {float((cd.double() - ref).abs().max()) / scale:.3e}   mean {float((cd.double() - ref).abs().mean()) / scale:.3e}")
    print(f"    time: bf16x3 product {timeit(lambda: ops.gemm_bf16x3_nt(x3, w3)):6.2f} us | split of A {timeit(lambda: ops.split_bf16x3(x)):5.2f} us | "
          f"Tensile f32 {timeit(lambda: x @ w):6.2f} us" + (f" | ee_dense f32 {timeit(lambda: ops.dense2x2(x.view(M, K // 4, 2, 2), w)):6.2f} us" if cd is not None else ""), flush=True)
