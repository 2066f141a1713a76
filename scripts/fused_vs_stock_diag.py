#!/usr/bin/env python3
"""Where the gradient error of the fused ResNet body sits, tensor by tensor, against a float64 run of the stock modules
(the measurement behind tests/test_gpu_path.py::test_fused_classifier_body_equals_stock_modules).  Five runs per side; prints each
run's total L2 error and the tensors carrying most of it.
    python3 scripts/fused_vs_stock_diag.py [depth] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402

from eeadv import models  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 18
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
DEV = torch.device("cuda", 0)
x = torch.rand(B, 3, 64, 64, device=DEV)
dl = torch.randn(B, 200, device=DEV)
ALL_STOCK = frozenset(("bn", "pool", "head", "conv", "stem", "dense", "conv3", "conv3s2"))


def run(stock, dt=torch.float32):
    models._STOCK = stock
    torch.manual_seed(21)
    net = models.make_resnet(depth, "tiny").to(DEV).to(dt).train()
    xi = x.to(dt).requires_grad_(True)
    logits = net(xi)
    names = ["input"] + [n for n, _ in net.named_parameters()]
    grads = torch.autograd.grad(logits, [xi] + list(net.parameters()), dl.to(dt))
    return names, logits.detach(), grads


names, l64, g64 = run(ALL_STOCK, torch.float64)
den = sum(float(g.norm()) ** 2 for g in g64) ** 0.5
print("resnet%d batch %d: |g64| = %.4g" % (depth, B, den))
for label, stock in (("fused", frozenset()), ("stock", ALL_STOCK), ("fused", frozenset())):
    for k in range(5):
        _, l, g = run(stock)
        errs = [float((a.double() - b).norm()) for a, b in zip(g, g64)]
        tot = sum(e * e for e in errs) ** 0.5
        top = sorted(range(len(errs)), key=lambda i: -errs[i])[:4]
        print("%s run %d: logits %.3e  total %.4g (%.2e of |g|)  top: %s" % (
            label, k, float((l.double() - l64).norm()), tot, tot / den,
            "  ".join("%s %.3g/%.3g" % (names[i], errs[i], float(g64[i].norm())) for i in top)), flush=True)

# the same comparison with the ReLU masks / pool argmax of the float64 run held to the fp32 run's (tests/test_gpu_path.py::replayed_body_gradients)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import pytest  # noqa: E402
import test_gpu_path as T  # noqa: E402

for seed, side in [(s, k) for s in range(4) for k in ("fused", "stock")]:
    with pytest.MonkeyPatch.context() as mp:
        names, l32, g32, l64, g64r = T.replayed_body_gradients(mp, depth, B, seed, frozenset(("bnpool",)) if side == "fused" else ALL_STOCK)
    errs = [float((a.double() - b).norm()) for a, b in zip(g32, g64r)]
    norms = [float(b.norm()) for b in g64r]
    den = sum(n * n for n in norms) ** 0.5
    tot = sum(e * e for e in errs) ** 0.5
    worst = sorted(range(len(errs)), key=lambda i: -errs[i] / (norms[i] + 1e-30))[:3]
    print("replay %s seed %d: logits %.3e  total %.3e of |g|  worst per tensor: %s" % (
        side, seed, float((l32.double() - l64).abs().max()), tot / den, "  ".join("%s %.2e" % (names[i], errs[i] / norms[i]) for i in worst)), flush=True)
models._STOCK = frozenset()
