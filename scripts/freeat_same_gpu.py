#!/usr/bin/env python3
"""BASELINE config 5's multi-rank path on a one-GPU box: two ranks time-sharing the GPU over gloo run the free-AT step of bench.py's
`imagenet_free_at` workload as it runs at N > 1 - SyncBatchNorm on the fused kernels (eeadv.syncbn: one all_gather / all_reduce per layer),
the flat gradient buffer with its segment pieces (eeadv.ddp.FlatGradSync), trainer.FreeAtStep (eager under SyncBatchNorm) - and check that
the ranks end with identical parameters and running statistics, and that one step moved them.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 scripts/freeat_same_gpu.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from eeadv import ddp, syncbn, trainer  # noqa: E402
from eeadv.models import make_resnet  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    os.environ["EEADV_GRAPH"] = "1"  # FreeAtStep must decline to capture on its own (a gradient exchange and SyncBatchNorm are in the repeat)
    torch.manual_seed(1 + rank)
    depth, B, size = int(os.environ.get("DEPTH", "50")), int(os.environ.get("BATCH", "8")), int(os.environ.get("SIZE", "224"))
    model = ddp.convert_sync_batchnorm(make_resnet(depth, "imagenet").to(dev)).train()
    assert sum(isinstance(m, syncbn.SyncBatchNorm2d) for m in model.modules()) in (20, 53)
    sync = ddp.FlatGradSync(model)
    assert sync.segmented and len(sync.pieces) == 3
    opt = trainer.make_sgd(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    noise = torch.zeros(B * world, 3, size, size, device=dev)
    step = trainer.FreeAtStep(model, trainer.Criterion(), opt, noise, 4 / 255, 4 / 255, 4, sync=sync)
    before = torch.cat([p.detach().flatten() for p in model.parameters()]).clone()
    g = torch.Generator().manual_seed(50 + rank)
    t0 = time.perf_counter()
    for it in range(2):
        x = torch.rand(B, 3, size, size, generator=g).to(dev)
        y = torch.randint(0, 1000, (B,), generator=g).to(dev)
        loss, out = step(x, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert step.graph is None and torch.isfinite(loss)
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    stats = torch.cat([b.detach().flatten().float() for n, b in model.named_buffers() if "running" in n])
    ref, sref = flat.clone(), stats.clone()
    dist.broadcast(ref, 0)
    dist.broadcast(sref, 0)
    print("rank %d: resnet%d %dx3x%dx%d, 2 steps x 4 repeats in %.1f s: loss %.4f, max |param - rank0| %.3e, max |running stat - rank0| %.3e, moved %.3e, |delta| max %.4f"
          % (rank, depth, B, size, size, dt, float(loss), float((flat - ref).abs().max()), float((stats - sref).abs().max()),
             float((flat - before).abs().max()), float(noise[:B].abs().max())), flush=True)
    assert float((flat - ref).abs().max()) == 0.0 and float((stats - sref).abs().max()) == 0.0 and float((flat - before).abs().max()) > 0
    assert float(noise[:B].abs().max()) <= 4 / 255 + 1e-7 and float(noise[B:].abs().max()) == 0.0
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
