#!/usr/bin/env python3
"""Which fused piece moves the input gradient of ResNet-18 (train mode) away from the all-stock result?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from eeadv import models  # noqa: E402

DEV = "cuda:0"
ALL = ("bn", "pool", "head", "conv", "stem", "dense", "conv3", "conv3s2")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
DEPTH = int(sys.argv[2]) if len(sys.argv) > 2 else 18
x = torch.rand(B, 3, 64, 64, device=DEV)
dl = torch.randn(B, 200, device=DEV)


def run(stock):
    models._STOCK = frozenset(stock)
    torch.manual_seed(21)
    net = models.make_resnet(DEPTH, "tiny").to(DEV).train()
    xi = x.clone().requires_grad_(True)
    logits = net(xi)
    (g,) = torch.autograd.grad(logits, [xi], dl)
    return logits.detach(), g


ref_l, ref_g = run(ALL)
l2, g2 = run(ALL)
rel = lambda a, b: float((a - b).norm() / b.norm())
print("stock vs stock (MIOpen noise): logits %.3e grad %.3e (scale %.3f) relL2 %.3e" % (float((l2 - ref_l).abs().max()), float((g2 - ref_g).abs().max()), float(ref_g.abs().max()), rel(g2, ref_g)))
for piece in ALL:
    l, g = run([p for p in ALL if p != piece])
    print("%-8s fused: logits diff %.3e   grad diff %.3e relL2 %.3e" % (piece, float((l - ref_l).abs().max()), float((g - ref_g).abs().max()), rel(g, ref_g)))
l, g = run([])
print("all fused: logits diff %.3e   grad diff %.3e relL2 %.3e" % (float((l - ref_l).abs().max()), float((g - ref_g).abs().max()), rel(g, ref_g)))
