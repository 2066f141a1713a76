#!/usr/bin/env python3
"""The EE front end of one PGD iteration at ImageNet resolution (resnet_EE_square.py:169-184, configs_imagenet/ee_at_bpda3_square.yml: 224 x 224,
per-rank batch 32 of 256 / 8): ee_chain takes maps up to 64 x 64, so here the front end is its separate kernels - square_draw, the band
low-pass with Add_Square fused on load, the edge filter + combine (forward); the edge adjoint, the low-pass with d Add_Square on store, the
update with the broadcast edge gradient (backward).  Run under `rocprofv3 --kernel-trace --stats`; prints each launch's ALGORITHMIC bytes
 so that the stats' average durations turn into GB/s, and its own event timing of the six-launch sequence."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from eeadv import models as M, ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = M.make_resnet_ee(18, "imagenet", True, cize=224, r=16, w=1.0, with_gf=False, low=38.0, high=76.0, alpha=0, sigma=1.0,
                     type_canny="CannyFilter_step125_1", epsilon=4 / 255, n_queries=1).to(dev).eval()
x = torch.rand(B, 3, 224, 224, device=dev)
x0 = x.clone()
g_in = torch.randn(B, 3, 224, 224, device=dev)
assert m.manual_ok(x) and not m.chain_ok(x)


def iteration():
    x_in, ctx = m.front_manual(x)
    g_lp, g_edge = m.front_manual_backward(g_in, ctx)
    ops.pgd_step_bcast_(x, g_lp, g_edge, x0, 1 / 255, 4 / 255)


for _ in range(5):
    iteration()
torch.cuda.synchronize()
n = 50
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    iteration()
b.record()
torch.cuda.synchronize()
px, C = B * 224 * 224, 3
print("front end of one PGD iteration at %d x 3 x 224 x 224: %.1f us per iteration (eager launches, torch events)" % (B, 1e3 * a.elapsed_time(b) / n))
print("algorithmic bytes per launch (SURVEY 8(d)):")
for name, nbytes in (("square_draw_kernel", 4 * B * C * 224), ("hfs_band_kernel<1 (Add_Square on load)", 8 * C * px), ("edge_fwd_kernel (+ combine, gate, saved responses)", (12 * C + C + 8) * px),
                     ("edge_bwd_saved_kernel", (9 * C + 12) * px), ("hfs_band_kernel<2 (d Add_Square on store)", 12 * C * px), ("pgd_step_bcast_kernel", (16 * C + 4) * px)):
    print("  %-52s %8.2f MB" % (name, nbytes / 1e6))
