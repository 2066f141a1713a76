#!/usr/bin/env python3
"""What does a HIP-graph boundary cost on the device?  PGD-40 on the bench model as 8 / 4 / 2 / 1 graph replays (5 / 10 / 20 / 40 iterations
per captured graph), un-profiled, device time from events around the whole attack."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

os.environ["EEADV_GRAPH"] = "1"
from eeadv import engine  # noqa: E402
from utils import attacks as A  # noqa: E402

torch.backends.cudnn.benchmark = True
cfg = bench.WORKLOADS["tiny_ee_at"]
dev = torch.device("cuda", 0)
model = bench.build_model(cfg).to(dev).train()
dargs = bench.driver_args(cfg)
x = torch.rand(cfg["batch"], *cfg["shape"], device=dev)
y = torch.randint(0, cfg["classes"], (cfg["batch"],), device=dev)
for per in (5, 10, 20, 40):
    engine.MAX_ITERS_PER_GRAPH = per
    engine.clear_graphs()
    for _ in range(3):
        A.PGD(model, dargs, x, y, 40, cfg["alpha"])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        A.PGD(model, dargs, x, y, 40, cfg["alpha"])
    b.record()
    torch.cuda.synchronize()
    print("%2d iterations per graph (%d replays per attack): %.3f ms per PGD-40 attack" % (per, 40 // per, a.elapsed_time(b) / 5), flush=True)
