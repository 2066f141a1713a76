"""Round 4: Net_2's head backward as one launch (ops.net2_head_bwd) beside the two launches it replaces (ops.fc_ce_grad + the product with fc1's
weight), at the bench batch (50) and at 512.  Run under rocprofv3 --kernel-trace --stats for the per-kernel durations; prints event timings."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "edge-enhancement_amd"))
from eeadv import ops  # noqa: E402

dev = torch.device("cuda:0")
for B in (50, 512):
    g = torch.Generator(device="cpu").manual_seed(B)
    z1 = torch.randn(B, 1024, generator=g).to(dev)
    w1 = (torch.randn(1024, 1024, generator=g) / 32).to(dev)
    w2 = (torch.randn(10, 1024, generator=g) / 32).to(dev)
    b2 = torch.randn(10, generator=g).to(dev)
    y = torch.randint(0, 10, (B,), generator=g).to(dev)

    def one():
        return ops.net2_head_bwd(z1, w2, b2, y, w1, "sum")

    def two():
        return ops.fc_ce_grad(z1, w2, b2, y, "sum") @ w1

    for name, fn in (("one launch", one), ("two launches", two)):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"B {B:4d} {name:13s} {e0.elapsed_time(e1) / 300 * 1e3:8.2f} us per call (back-to-back, launch included)")
    print("max |one - two| =", float((one() - two()).abs().max()))
