#!/usr/bin/env python3
"""Back-to-back launches of the MFMA 3x3 convolution (layer-1 shape) for PMC / trace runs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from eeadv import ops  # noqa: E402

B, C, H = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (100, 64, 16)))
x = torch.randn(B, C, H, H, device="cuda:0")
w = torch.randn(C, C, 3, 3, device="cuda:0")
for _ in range(30):
    y = ops.conv3x3s1_fwd(x, w)
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
