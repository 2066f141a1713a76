#!/usr/bin/env python3
"""BASELINE config 5's multi-rank repeat (SyncBatchNorm on the fused kernels + the flat gradient exchange in segment pieces) as ONE
captured HIP graph WITH its collectives inside - rehearsed on a one-GPU box: one rank over RCCL with every collective forced on
(EEADV_FORCE_COLLECTIVES=1: 2 x 53 SyncBatchNorm exchanges + 3 gradient pieces per repeat of ResNet-50), eager against graphed from the same
initial state: parameters / running statistics / noise must end identical, and the host time per repeat must drop below the device time.
    python scripts/freeat_graph_collectives.py [depth batch size imagenet|tiny]      (tiny: the 64 x 64 ResNet-18, whose step is bit-reproducible: exact check)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
sys.path.insert(0, ROOT)
os.environ["EEADV_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from eeadv import ddp, engine, syncbn, trainer  # noqa: E402
from eeadv.models import make_resnet  # noqa: E402


def run(graphed, depth, B, size, batches=3, repeats=4, flavour="imagenet"):
    os.environ["EEADV_GRAPH"] = "1" if graphed else "0"
    os.environ["EEADV_GRAPH_COLLECTIVES"] = "1" if graphed else "0"
    engine.clear_graphs()
    trainer.clear_update_graphs()
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    model = ddp.convert_sync_batchnorm(make_resnet(depth, flavour).to(dev)).train()
    classes = 1000 if flavour == "imagenet" else 200
    n_sync = sum(isinstance(m, syncbn.SyncBatchNorm2d) for m in model.modules())
    sync = ddp.FlatGradSync(model)
    assert sync.active and len(sync.pieces) == 3
    opt = trainer.make_sgd(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    noise = torch.zeros(B, 3, size, size, device=dev)
    step = trainer.FreeAtStep(model, trainer.Criterion(), opt, noise, 4 / 255, 4 / 255, repeats, sync=sync)
    g = torch.Generator().manual_seed(50)
    host = dev_ms = 0.0
    for it in range(batches):
        x = torch.rand(B, 3, size, size, generator=g).to(dev)
        y = torch.randint(0, classes, (B,), generator=g).to(dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        loss, out = step(x, y)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        if it == batches - 1:  # steady state: the graph exists (graphed) / everything is warm (eager)
            host, dev_ms = 1e3 * (t1 - t0) / repeats, e0.elapsed_time(e1) / repeats
    assert (step.graph is not None) == graphed
    flat = torch.cat([p.detach().flatten() for p in model.parameters()]).clone()
    stats = torch.cat([b.detach().flatten().float() for n, b in model.named_buffers() if "running" in n]).clone()
    return dict(loss=float(loss), flat=flat, stats=stats, noise=noise.clone(), host_ms=host, dev_ms=dev_ms, n_sync=n_sync)


def main():
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 224
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    flavour = sys.argv[4] if len(sys.argv) > 4 else "imagenet"
    eager = run(False, depth, B, size, flavour=flavour)
    eager2 = run(False, depth, B, size, flavour=flavour)  # MIOpen's weight-gradient solvers at ImageNet sizes accumulate with atomics: two EAGER runs differ too
    graph = run(True, depth, B, size, flavour=flavour)
    diff = lambda a, b, k: float((a[k] - b[k]).abs().max())
    dp, ds, dn = diff(eager, graph, "flat"), diff(eager, graph, "stats"), diff(eager, graph, "noise")
    bp, bs, bn_ = diff(eager, eager2, "flat"), diff(eager, eager2, "stats"), diff(eager, eager2, "noise")
    print("resnet%d, %d x 3 x %d x %d, one rank over RCCL, %d SyncBatchNorm layers (2 collectives each per repeat) + 3 gradient pieces" % (depth, B, size, size, eager["n_sync"]))
    print("eager   : host %.2f ms per repeat, device %.2f ms per repeat, loss %.5f" % (eager["host_ms"], eager["dev_ms"], eager["loss"]))
    print("graphed : host %.2f ms per repeat, device %.2f ms per repeat, loss %.5f" % (graph["host_ms"], graph["dev_ms"], graph["loss"]))
    print("after 3 batches x 4 repeats, eager vs graphed: max |parameter difference| %.3e, |running statistic| %.3e, |noise| %.3e" % (dp, ds, dn))
    print("                             eager vs eager  : max |parameter difference| %.3e, |running statistic| %.3e, |noise| %.3e" % (bp, bs, bn_))
    assert dp <= 3 * bp + 1e-6 and ds <= 3 * bs + 1e-6 and dn <= 3 * bn_ + 1e-6, "the captured repeat is further from the eager one than two eager runs are from each other"
    if flavour == "tiny":  # every kernel of the 64 x 64 ResNet-18 step is hand-written and reduces in a fixed order: bit for bit
        assert bp == 0.0 and dp == 0.0 and ds == 0.0 and dn == 0.0, "the captured repeat does not reproduce the eager one bit for bit"
    assert graph["host_ms"] < graph["dev_ms"], "the host is still behind the device"
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
