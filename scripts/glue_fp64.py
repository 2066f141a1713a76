#!/usr/bin/env python3
"""Diagnostic (GPU box): the hand-written CNN glue vs the stock ATen / MIOpen modules, both against a float64 run of the same
network: per-tensor L2 error of every gradient, train mode.  python scripts/glue_fp64.py [depth] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402

from eeadv import models  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = "cuda:0"
ALL = frozenset(("bn", "pool", "head", "conv", "stem", "dense", "conv3", "conv3s2"))
torch.manual_seed(0)
x = torch.rand(B, 3, 64, 64, device=dev)
dl = torch.randn(B, 200, device=dev)
res = {}
for mode, stock, dt in (("fused", frozenset(), torch.float32), ("stock", ALL, torch.float32), ("stock2", ALL, torch.float32), ("f64", ALL, torch.float64)):
    models._STOCK = stock
    torch.manual_seed(21)
    net = models.make_resnet(depth, "tiny").to(dev).to(dt).train()
    xi = x.to(dt).requires_grad_(True)
    logits = net(xi)
    grads = torch.autograd.grad(logits, [xi] + list(net.parameters()), dl.to(dt))
    res[mode] = [logits.detach().double()] + [g.double() for g in grads]
    names = ["logits", "x"] + [n for n, _ in net.named_parameters()]
r64 = res["f64"]
print("%-34s %10s %10s %10s %10s" % ("tensor", "|g64|", "fused", "stock", "stock2"))
tot = {k: 0.0 for k in ("fused", "stock", "stock2")}
den = 0.0
rat = []
for i, n in enumerate(names):
    n64 = float(r64[i].norm())
    e = {k: float((res[k][i] - r64[i]).norm()) for k in tot}
    if i > 0:
        for k in tot:
            tot[k] += e[k] ** 2
        den += n64 ** 2
        rat.append((e["fused"] / max(e["stock"], 1e-30), n))
    if i < 6 or e["fused"] > 3 * e["stock"]:
        print("%-34s %10.3e %10.3e %10.3e %10.3e" % (n, n64, e["fused"], e["stock"], e["stock2"]))
if len(sys.argv) > 3:  # bisect: one hand-written piece at a time, everything else stock
    for piece in sorted(ALL):
        models._STOCK = ALL - {piece}
        torch.manual_seed(21)
        net = models.make_resnet(depth, "tiny").to(dev).train()
        xi = x.clone().requires_grad_(True)
        logits = net(xi)
        grads = torch.autograd.grad(logits, [xi] + list(net.parameters()), dl)
        e = sum(float((g.double() - r).norm()) ** 2 for g, r in zip(grads, r64[1:])) ** 0.5
        worst = max(((float((g.double() - r).norm()) / max(float((s_ - r).norm()), 1e-30), n) for g, r, s_, n in zip(grads, r64[1:], res["stock"][1:], names[1:])))
        print("only %-8s hand-written: all-gradient error %.3e   worst tensor ratio vs stock %.1f (%s)" % (piece, e, worst[0], worst[1]))
print("all gradients: |g64| %.3e  fused %.3e  stock %.3e  stock2 %.3e" % (den ** 0.5, tot["fused"] ** 0.5, tot["stock"] ** 0.5, tot["stock2"] ** 0.5))
rat.sort()
print("fused/stock error ratio per tensor: min %.2f  median %.2f  90%% %.2f  max %.2f (%s)" % (rat[0][0], rat[len(rat) // 2][0], rat[int(0.9 * len(rat))][0], rat[-1][0], rat[-1][1]))
