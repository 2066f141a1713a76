"""Diagnostic (GPU box): how far apart are the free-AT noise gradients of the GPU path and the CPU oracle on resnet50 / 224 / B=4
with identical weights, and at which gradient magnitudes do their signs differ?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
sys.path.insert(0, ROOT)
from eeadv import ops  # noqa: E402
from eeadv.models import make_resnet  # noqa: E402
from oracle import ref_path as R  # noqa: E402

torch.manual_seed(3)
B, K = 4, 1000
cpu = R.resnet50(num_classes=K, imagenet_pool=True).train()
gpu = make_resnet(50, "imagenet").to("cuda").train()
gpu.load_state_dict(cpu.state_dict())
x = torch.rand(B, 3, 224, 224)
y = torch.randint(0, K, (B,))
for mode in ("train", "eval"):
    cpu.train(mode == "train"), gpu.train(mode == "train")
    xc = x.clone().requires_grad_(True)
    lc = torch.nn.functional.cross_entropy(cpu(xc), y)
    (gc,) = torch.autograd.grad(lc, xc)
    xg = x.cuda().requires_grad_(True)
    lg = torch.nn.functional.cross_entropy(gpu(xg), y.cuda())
    (gg,) = torch.autograd.grad(lg, xg)
    xd = x.double().requires_grad_(True)
    cpu64 = R.resnet50(num_classes=K, imagenet_pool=True).double().train(mode == "train")
    cpu64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in cpu.state_dict().items()})
    l64 = torch.nn.functional.cross_entropy(cpu64(xd), y)
    (g64,) = torch.autograd.grad(l64, xd)
    gc, gg, g64 = gc.numpy(), gg.cpu().numpy(), g64.numpy()
    s = np.abs(g64).max()
    print(mode, "loss cpu %.6f gpu %.6f f64 %.6f" % (lc.item(), lg.item(), l64.item()))
    print("  max|g| %.3e   err(cpu32,f64)/max %.2e   err(gpu32,f64)/max %.2e   err(gpu,cpu)/max %.2e" % (
        s, np.abs(gc - g64).max() / s, np.abs(gg - g64).max() / s, np.abs(gg - gc).max() / s))
    flip = np.sign(gg) != np.sign(gc)
    print("  sign flips gpu vs cpu: %d of %d (%.3f %%); largest |g_cpu| among them / max = %.2e" % (
        flip.sum(), flip.size, 100 * flip.mean(), np.abs(gc[flip]).max() / s if flip.any() else 0))
    for q in (1e-4, 1e-3, 2e-3, 5e-3, 1e-2, 2e-2, 5e-2):
        print("    share of |g_cpu| <= %.0e * max: %.3f" % (q, (np.abs(gc) <= q * s).mean()))
