#!/usr/bin/env python3
"""Times ee_chain_fwd_f32 / ee_chain_bwd_f32 alone (graph-replayed back-to-back launches) next to the separate kernels they
replace."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from eeadv import hfs as HF, ops  # noqa: E402


def timeit(fn, iters=50, reps=4):
    fn()
    torch.cuda.synchronize()
    if os.environ.get("CHAIN_BENCH_EAGER"):  # plain launches (for rocprofv3 --pmc passes): time is host-bound, counters are not
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return 1e3 * a.elapsed_time(b) / iters
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


def main():
    dev = "cuda:0"
    shapes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["100x3x64x64", "1600x3x64x64", "50x1x28x28"]
    wts = ops.EdgeWeights(1.0)
    for shp in shapes:
        B, C, H, W = map(int, shp.split("x"))
        r = 8 if H == 64 else 4
        op = HF.HFSOperator(H, W, r, dev)
        x = torch.rand(B, C, H, W, device=dev)
        x0 = x.clone()
        g = torch.randn(B, C, H, W, device=dev)
        eps, s = 16 / 255, max(int(round((0.8 * H * H) ** 0.5)), 1)
        state = torch.tensor([7, 0, 0, 0], dtype=torch.int64, device=dev)
        sizes = torch.tensor([s], dtype=torch.int32, device=dev)
        px = B * H * W
        x_in, gate, gx, gy, _ = ops.chain_fwd(x, op.chain, wts, 0.0, 76 / 255, 1.0, True, eps, s, state)
        rows = [
            ("chain_fwd", lambda: ops.chain_fwd(x, op.chain, wts, 0.0, 76 / 255, 1.0, True, eps, s, state), (9 * C + 8) * px),
            ("chain_fwd nosq", lambda: ops.chain_fwd(x, op.chain, wts, 0.0, 76 / 255, 1.0), (9 * C + 8) * px),
            ("chain_bwd", lambda: ops.chain_bwd_(x, g, gate, gx, gy, x0, op.chain, wts, 0.0, 76 / 255, 1.0, 2 / 255, 16 / 255), (17 * C + 8) * px),
        ]
        if True:
            d = dict(zip(("stripe", "sq_pos", "sq_sign"), ops.square_draw(B, C, H, sizes, state)))
            d["sq_size"] = sizes
            gate0 = (gate & 1).contiguous()
            x_lp = op.forward_square(x, eps, d)
            g_hfs, g_edge = ops.frontend_bwd_saved(g, gate0, gx, gy, wts, 0.0, 76 / 255, 1.0)
            g_lp = op.backward_square(g_hfs, x, eps, d)
            rows += [
                ("square_draw", lambda: ops.square_draw(B, C, H, sizes, state), 0),
                ("hfs<1>", lambda: op.forward_square(x, eps, d), 8 * C * px),
                ("frontend_fwd_save", lambda: ops.frontend_fwd_save(x, x_lp, wts, 0.0, 76 / 255, 1.0), 12 * C * px),
                ("frontend_bwd_saved", lambda: ops.frontend_bwd_saved(g, gate0, gx, gy, wts, 0.0, 76 / 255, 1.0), 16 * C * px),
                ("hfs<2>", lambda: op.backward_square(g_hfs, x, eps, d), 12 * C * px),
                ("pgd_step_bcast", lambda: ops.pgd_step_bcast_(x, g_lp, g_edge, x0, 2 / 255, 16 / 255), (16 * C + 4) * px),
            ]
        for name, fn, nbytes in rows:
            us = timeit(fn)
            print("%-20s %-14s %8.2f us %8.1f GB/s" % (name, shp, us, nbytes / us / 1e3))


if __name__ == "__main__":
    main()
