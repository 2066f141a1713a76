#!/usr/bin/env python3
"""Two data-parallel ranks sharing ONE GPU over gloo: an integration check of the multi-process path on a 1-GPU box (RCCL needs
one device per rank, so the collective backend differs from production; everything else is the N > 1 path of bench.py: the
captured attack graph, the captured forward + backward writing one flat gradient buffer, its all-reduce (eeadv.ddp.FlatGradSync),
the captured SGD step).
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 scripts/ddp_same_gpu.py
RCCL itself on a one-GPU box: ONE rank with the collectives forced on (communicator set-up, async all-reduce of the flat buffer
between the two captured graphs, RCCL's watchdog thread next to a stream capture):
    DDP_BACKEND=nccl EEADV_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 \
        --master-port 29512 scripts/ddp_same_gpu.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from eeadv import ddp, engine, trainer  # noqa: E402
from eeadv.models import make_resnet_ee  # noqa: E402


class Args(dict):
    __getattr__ = dict.get


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["EEADV_GRAPH"] = "1"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    backend = os.environ.get("DDP_BACKEND", "gloo")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1 + rank)
    model = make_resnet_ee(18, "tiny", square=True, cize=64, r=16, w=0.5, low=60.0, high=120.0, alpha=0.0, sigma=1,
                           type_canny="CannyFilter_step125_1", epsilon=0.05, n_queries=1).to(dev).train()
    sync = ddp.FlatGradSync(model)  # broadcasts rank 0's weights (the ranks were seeded differently)
    net = model
    opt = trainer.make_sgd(net.parameters(), lr=0.05, momentum=0.9, weight_decay=2e-4)  # what bench.py and the drivers build (torch's fused SGD)
    args = Args(method_name="EE_BPDA3_AT_square", random=True, epsilon=16 / 255, num_steps_1=4, step_size_1=2 / 255, num_classes=200)
    crit = trainer.make_criterion(args)
    trainer.PHASE_EVENTS = trainer.PhaseEvents()
    for step in range(6):  # two eager updates, the capture, three replays
        x = torch.rand(16, 3, 64, 64, device=dev)
        y = torch.randint(0, 200, (16,), device=dev)
        trainer.PHASE_EVENTS.start()
        loss, out = trainer.train_batch(net, crit, opt, args, x, y, dev, sync=sync)
    torch.cuda.synchronize()
    assert torch.isfinite(loss), loss
    # same initial weights (DDP broadcast) + averaged gradients -> identical parameters on every rank
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    ref = flat.clone()
    dist.broadcast(ref, 0)
    diff = float((flat - ref).abs().max())
    stats = torch.cat([b.detach().flatten().float() for n, b in model.named_buffers() if "running" in n])
    print("rank %d: loss %.4f  max |param - rank0 param| = %.3e  graphs %d  bn stats finite %s" % (
        rank, float(loss), diff, len(engine._GRAPHS), bool(torch.isfinite(stats).all())), flush=True)
    assert diff == 0.0
    assert len(engine._GRAPHS) == 1  # the attack replayed a captured graph
    upd = list(trainer._UPDATES.values())[0][3]
    assert len(upd) == 1 and list(upd.values())[0].graph2 is not None  # ... and the update replayed its two graphs around the all-reduce
    print("rank %d phases (ms): %s" % (rank, trainer.PHASE_EVENTS.summary()), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
