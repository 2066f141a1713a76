#!/usr/bin/env python3
"""Diagnostic (GPU box): is the STOCK ATen / MIOpen ResNet backward reproducible, and how far from float64 is it?  Repeats the same
forward / backward of the all-stock network and prints the gradient error against a float64 run, with the deepest tensor (in
backward order) whose error exceeds 1e-4 of its norm.  Run under MIOPEN_DEBUG_CONV_* = 0 switches to find the solver family."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402

from eeadv import models  # noqa: E402

depth, B, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = "cuda:0"
models._STOCK = frozenset(("bn", "pool", "head", "conv", "stem", "dense", "conv3", "conv3s2"))
torch.manual_seed(0)
x = torch.rand(B, 3, 64, 64, device=dev)
dl = torch.randn(B, 200, device=dev)


def run(dt):
    torch.manual_seed(21)
    net = models.make_resnet(depth, "tiny").to(dev).to(dt).train()
    xi = x.to(dt).requires_grad_(True)
    grads = torch.autograd.grad(net(xi), [xi] + list(net.parameters()), dl.to(dt))
    return [g.double() for g in grads], ["x"] + [n for n, _ in net.named_parameters()]


r64, names = run(torch.float64)
print("env:", {k: v for k, v in os.environ.items() if k.startswith("MIOPEN")})
for rep in range(reps):
    g, _ = run(torch.float32)
    errs = [float((a - b).norm()) / max(float(b.norm()), 1e-30) for a, b in zip(g, r64)]
    tot = sum(float((a - b).norm()) ** 2 for a, b in zip(g, r64)) ** 0.5
    bad = [n for n, e in zip(names, errs) if e > 1e-4]
    print("rep %d: all-gradient error %.3e; tensors off by > 1e-4: %d of %d; deepest: %s" % (rep, tot, len(bad), len(names), bad[-1] if bad else "-"))
