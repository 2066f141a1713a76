#!/usr/bin/env python3
"""Micro-benchmark of the hand-written kernels in isolation (back-to-back launches, torch events on the
launch stream): per-kernel microseconds and algorithmic GB/s at several batch sizes."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from eeadv import ops  # noqa: E402


def timeit(fn, iters=200, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="100x3x64x64,400x3x64x64,1600x3x64x64,50x1x28x28,32x3x224x224")
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    dev = "cuda:0"
    wts = ops.EdgeWeights(1.0)
    print("%-18s %-16s %9s %9s" % ("kernel", "shape", "us", "GB/s"))
    for shp in a.shapes.split(","):
        B, C, H, W = map(int, shp.split("x"))
        x = torch.rand(B, C, H, W, device=dev)
        xh = torch.rand(B, C, H, W, device=dev)
        g = torch.randn(B, C, H, W, device=dev)
        x0 = x.clone()
        px = B * H * W
        _, gate, _ = ops.frontend_fwd(x, xh, wts, 0.0, 76 / 255, 1.0)
        _, _, _, sgx, sgy = ops.frontend_fwd_save(x, xh, wts, 0.0, 76 / 255, 1.0)
        u = torch.randn(B, 1, H, W, device=dev)
        gl, ge = ops.frontend_bwd(g, gate, x, wts, 0.0, 76 / 255, 1.0)
        y = torch.randint(0, 200, (B,), device=dev)
        z = torch.randn(B, 200, device=dev)
        from eeadv import hfs as HF
        import utils.core as core
        hfs_rows = []
        if H <= 64 and W <= 64 and H == W:
            r = 8 if H == 64 else 4
            op = HF.HFSOperator(H, W, r, dev)
            sq = core.Add_Square(C, H, 16 / 255, n_queries=1)
            d = sq.prepare(x)
            hfs_rows = [
                ("hfs", lambda: op.forward(x), 8 * C * px),
                ("hfs+square", lambda: op.forward_square(x, 16 / 255, d), 8 * C * px),
                ("hfs*dsquare", lambda: op.backward_square(g, x, 16 / 255, d), 12 * C * px),
                ("add_square_fwd", lambda: ops.add_square_fwd(x, 16 / 255, d["stripe"], d["sq_sign"], d["sq_pos"], d["sq_size"]), 8 * C * px),
                ("hfs dense(rocBLAS)", lambda: op._apply(x, op.Bcat, op.Ar, op.Ai), 8 * C * px),
            ]
        rows = hfs_rows + [
            ("pgd_step", lambda: ops.pgd_step_(x, g, x0, 2 / 255, 16 / 255), 16 * C * px),
            ("pgd_step_bcast", lambda: ops.pgd_step_bcast_(x, gl, ge, x0, 2 / 255, 16 / 255), (16 * C + 4) * px),
            ("frontend_fwd", lambda: ops.frontend_fwd(x, xh, wts, 0.0, 76 / 255, 1.0), 12 * C * px),
            ("frontend_bwd", lambda: ops.frontend_bwd(g, gate, x, wts, 0.0, 76 / 255, 1.0), 16 * C * px),
            ("frontend_fwd_save", lambda: ops.frontend_fwd_save(x, xh, wts, 0.0, 76 / 255, 1.0), 12 * C * px),
            ("frontend_bwd_saved", lambda: ops.frontend_bwd_saved(g, gate, sgx, sgy, wts, 0.0, 76 / 255, 1.0), 16 * C * px),
            ("edge125_fwd", lambda: ops.edge125_fwd(x, wts, 0.0, 76 / 255), (C + 1) * 4 * px),
            ("edge125_bwd", lambda: ops.edge125_bwd(x, u, wts, 0.0, 76 / 255), (C + 2) * 4 * px),
            ("ce_grad", lambda: ops.ce(z, y, "sum", 0.0, False, True), 2 * B * 200 * 4),
        ]
        for name, fn, nbytes in rows:
            us = timeit(fn, a.iters)
            print("%-18s %-16s %9.2f %9.1f" % (name, shp, us, nbytes / us / 1e3))


if __name__ == "__main__":
    main()
