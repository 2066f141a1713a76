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

from eeadv import ddp, engine, runtime, trainer  # noqa: E402
from eeadv.models import make_resnet_ee  # noqa: E402


class Args(dict):
    __getattr__ = dict.get


def run_config(rank, world, dev, method, segmented, steps=6, batch=16, timed_from=0):
    """`steps` training steps (two eager updates, the capture, replays) of one configuration from a fixed seed; returns the flat parameters"""
    trainer._SEGMENTED = segmented
    trainer.clear_update_graphs()
    engine.clear_graphs()
    torch.manual_seed(1 + rank)
    runtime.reseed()  # the device-resident Philox state of the in-graph draws (Add_Square) restarts from the generator too
    runtime.draw_state(dev)
    model = make_resnet_ee(18, "tiny", square=True, cize=64, r=16, w=0.5, low=60.0, high=120.0, alpha=0.0, sigma=1,
                           type_canny="CannyFilter_step125_1", epsilon=0.05, n_queries=1).to(dev).train()
    sync = ddp.FlatGradSync(model)  # broadcasts rank 0's weights and buffers (the ranks were seeded differently)
    opt = trainer.make_sgd(model.parameters(), lr=0.05, momentum=0.9, weight_decay=2e-4)  # what bench.py and the drivers build (torch's fused SGD)
    args = Args(method_name=method, random=True, epsilon=16 / 255, num_steps_1=4, step_size_1=2 / 255, num_classes=200, beta=6.0)
    crit = trainer.make_criterion(args)
    trainer.PHASE_EVENTS = None
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    for step in range(steps):
        x = torch.rand(batch, 3, 64, 64, generator=g).to(dev)
        y = torch.randint(0, 200, (batch,), generator=g).to(dev)
        if step == timed_from:  # phase times of the replayed steps only
            torch.cuda.synchronize()
            trainer.PHASE_EVENTS = trainer.PhaseEvents()
        if trainer.PHASE_EVENTS is not None:
            trainer.PHASE_EVENTS.start()
        loss, out = trainer.train_batch(model, crit, opt, args, x, y, dev, sync=sync)
    torch.cuda.synchronize()
    assert torch.isfinite(loss), loss
    # same initial weights (broadcast) + averaged gradients -> identical parameters on every rank
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    ref = flat.clone()
    dist.broadcast(ref, 0)
    diff = float((flat - ref).abs().max())
    stats = torch.cat([b.detach().flatten().float() for n, b in model.named_buffers() if "running" in n])
    upd = list(list(trainer._UPDATES.values())[0][3].values())[0]
    shape = ("%d backward graphs, pieces %s MB" % (1 + len(upd.seg_graphs), [round(p.numel() * 4 / 1e6, 1) for p in sync.pieces])
             if hasattr(upd, "seg_graphs") else "graphs g1 g2 g3: %s" % [g is not None for g in (upd.g1, upd.g2, upd.g3)])
    print("rank %d %s segmented=%s: loss %.4f  max |param - rank0 param| = %.3e  attack graphs %d  %s  bn stats finite %s\n   phases (ms): %s" % (
        rank, method, segmented, float(loss), diff, len(engine._GRAPHS), shape, bool(torch.isfinite(stats).all()), trainer.PHASE_EVENTS.summary()), flush=True)
    assert diff == 0.0
    assert len(engine._GRAPHS) >= 1  # the attack replayed a captured graph
    if hasattr(upd, "seg_graphs"):
        assert upd.graph2 is not None and len(upd.seg_graphs) == (2 if segmented else 0)
    else:
        assert upd.g3 is not None  # TRADES: forward | attack | loss + backward | all-reduce | SGD
    trainer.PHASE_EVENTS = None
    return flat


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
    one = run_config(rank, world, dev, "EE_BPDA3_AT_square", segmented=False)
    seg = run_config(rank, world, dev, "EE_BPDA3_AT_square", segmented=True)
    # the backward cut at the layer boundaries is the same arithmetic, and since every weight gradient has a fixed summation order (ee_wrw.hip) the two forms agree EXACTLY
    rel = float((one - seg).abs().max() / one.abs().max())
    print("rank %d: segmented vs one-piece update, max relative parameter difference after 6 steps: %.3e" % (rank, rel), flush=True)
    # (with MIOpen's weight gradients - EEADV_STOCK_WRW=1 - six free-running steps at lr 0.05 amplify its atomics' rounding noise to 1e-1; tests/test_gpu_ddp.py compares step by step)
    run_config(rank, world, dev, "TRADES", segmented=True, steps=5)
    if os.environ.get("DDP_TIMING", "0") == "1":  # what cutting the backward into three graphs + three collectives costs per step (batch 100, replays only)
        for segmented in (False, True, False, True):
            run_config(rank, world, dev, "EE_BPDA3_AT_square", segmented=segmented, steps=16, batch=100, timed_from=6)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
