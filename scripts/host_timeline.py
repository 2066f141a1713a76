#!/usr/bin/env python3
"""Where the HOST spends a training step of the default bench workload (graph mode): wall time of every graph replay call and of the
whole train_batch call, against the step's device time - is the host ahead of the GPU (launch-ahead) or locked to it?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

os.environ["EEADV_GRAPH"] = "1"
from eeadv import engine, trainer  # noqa: E402

torch.backends.cudnn.benchmark = True
cfg = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "tiny_ee_at"]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model = bench.build_model(cfg).to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=cfg["lr"], momentum=cfg["momentum"], weight_decay=cfg["wd"])
dargs = bench.driver_args(cfg)
crit = trainer.make_criterion(dargs)
B = cfg["batch"]
batches = [(torch.rand(B, *cfg["shape"], device=dev), torch.randint(0, cfg["classes"], (B,), device=dev)) for _ in range(4)]
calls = []
orig = torch.cuda.CUDAGraph.replay


def timed(self):
    t = time.perf_counter()
    orig(self)
    calls.append(time.perf_counter() - t)


torch.cuda.CUDAGraph.replay = timed
for i in range(6):
    trainer.train_batch(model, crit, opt, dargs, *batches[i % 4], dev)
torch.cuda.synchronize()
for rep in range(2):
    calls.clear()
    host = []
    t0 = time.perf_counter()
    for i in range(10):
        t = time.perf_counter()
        trainer.train_batch(model, crit, opt, dargs, *batches[i % 4], dev)
        host.append(time.perf_counter() - t)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("10 steps: host returned after %.1f ms, device done after %.1f ms" % (1e3 * t_host, 1e3 * t_all))
    print("  train_batch host ms:", " ".join("%.2f" % (1e3 * h) for h in host))
    print("  graph replay calls per step: %d, host ms each:" % (len(calls) // 10), " ".join("%.2f" % (1e3 * c) for c in calls[:len(calls) // 10 * 2]))
