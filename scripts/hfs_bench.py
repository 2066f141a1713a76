#!/usr/bin/env python3
"""HighFreqSuppress: the band kernel on the matrix cores (ee_hfs_mfma_f32) against the LDS kernel (ee_hfs_f32, planes <= 64 x 64) and
the dense rocBLAS form (three launches) it replaces for larger planes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402

from eeadv import hfs as HF, ops  # noqa: E402


def timeit(fn, iters=20, reps=3):
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


for (B, C, n, r) in [(32, 3, 224, 16), (256, 3, 224, 16), (100, 3, 64, 8), (1600, 3, 64, 8), (50, 1, 28, 4)]:
    x = torch.rand(B, C, n, n, device="cuda:0")
    op = HF.HFSOperator(n, n, r, "cuda:0")
    nbytes = 8 * x.numel()
    rows = [("mfma band kernel", lambda: ops.hfs_mfma(x, *op.mfma))]
    if op.kernel is not None:
        rows.append(("LDS kernel (VALU)", lambda: ops.hfs(x, *op.kernel)))
    rows.append(("dense rocBLAS x3", lambda: op._apply(x, op.Bcat, op.Ar, op.Ai)))
    for name, fn in rows:
        us = timeit(fn)
        print("%-20s %4dx%dx%dx%d  %9.2f us  %8.1f GB/s (%.2f of 8 TB/s)" % (name, B, C, n, n, us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000))
