"""CPU, world_size 2 over gloo: the N > 1 path (batch sharding + gradient all-reduce, no collective inside
the attack loop).  Kernels cannot run here, so the attack goes through the opt-in torch plumbing path; what
is under test is the partitioning, seeding and gradient-averaging logic of eeadv.ddp / eeadv.trainer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tiny_models import Args, TinyNet, TinySegNet


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg():
    return Args(method_name="AT", random=False, epsilon=0.1, num_steps_1=3, step_size_1=0.02, num_classes=10, attack_method="PGD")


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "edge-enhancement_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    from eeadv import ddp, runtime, trainer
    runtime.allow_cpu_plumbing(True)
    ddp.setup(device="cpu")
    assert ddp.world() == 2 and ddp.rank() == rank and ddp.rank_seed(5) == 5 + rank and ddp.per_rank_batch(8) == 4
    torch.manual_seed(0)  # same weights on every rank
    os.environ["EEADV_GRAD_SYNC"] = "ddp"  # the stock multi-rank path, as the drivers / bench.py select it (ddp.make_grad_sync)
    model, sync = ddp.make_grad_sync(TinyNet(2, 8, 10, 7), "cpu")
    assert sync is None and type(model).__name__ == "DistributedDataParallel" and ddp.grad_sync_mode() == "ddp"
    os.environ["EEADV_GRAD_SYNC"] = "flat"
    assert ddp.grad_sync_mode() == "flat"
    os.environ.pop("EEADV_GRAD_SYNC")
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(123)
    X, Y = torch.rand(8, 2, 8, 8, generator=g), torch.randint(0, 10, (8,), generator=g)
    idx = ddp.shard_indices(8)
    x, y = X[idx], Y[idx]
    args = _cfg()
    crit = trainer.make_criterion(args)
    data_adv, _, _ = trainer.attack_for_training(model, crit, args, x, y, "cpu")
    loss, out = trainer.train_batch(model, crit, opt, args, x, y, "cpu")
    m1, m2 = ddp.gather_mean(float(loss), float(rank))
    tmax = ddp.max_over_ranks(1.0 + rank)
    torch.save({"idx": idx, "adv": data_adv, "params": [p.detach().clone() for p in model.parameters()], "loss": float(loss),
                "mean_loss": m1, "mean_rank": m2, "tmax": tmax}, os.path.join(out_dir, "r%d.pt" % rank))
    ddp.teardown()


def test_two_rank_gloo_training_step(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(str(tmp_path / "r0.pt"), weights_only=False)
    r1 = torch.load(str(tmp_path / "r1.pt"), weights_only=False)
    assert r0["idx"] == [0, 2, 4, 6] and r1["idx"] == [1, 3, 5, 7]
    # parameters are identical on both ranks after the all-reduced step
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    assert abs(r0["mean_loss"] - 0.5 * (r0["loss"] + r1["loss"])) < 1e-12 and r0["mean_rank"] == 0.5 and r0["tmax"] == 2.0
    # single-process emulation: same shards, gradients averaged by hand
    from eeadv import runtime, trainer
    import torch.nn.functional as F
    runtime.allow_cpu_plumbing(True)
    try:
        torch.manual_seed(0)
        net = TinyNet(2, 8, 10, 7)
        opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        g = torch.Generator().manual_seed(123)
        X, Y = torch.rand(8, 2, 8, 8, generator=g), torch.randint(0, 10, (8,), generator=g)
        args = _cfg()
        crit = trainer.make_criterion(args)
        grads = []
        for r, res in ((0, r0), (1, r1)):
            x, y = X[res["idx"]], Y[res["idx"]]
            adv, _, _ = trainer.attack_for_training(net, crit, args, x, y, "cpu")
            assert torch.equal(adv, res["adv"])  # no collective inside the attack: a rank's shard is attacked independently
            net.zero_grad()
            F.cross_entropy(net(adv), y).backward()
            grads.append([p.grad.clone() for p in net.parameters()])
        for p, g0, g1 in zip(net.parameters(), *grads):
            p.grad = (g0 + g1) / 2
        opt.step()
        for p, q in zip(net.parameters(), r0["params"]):
            np.testing.assert_allclose(p.detach().numpy(), q.numpy(), rtol=1e-6, atol=1e-7)
    finally:
        runtime.allow_cpu_plumbing(False)


def test_shard_indices_and_padding():
    from eeadv import ddp
    assert ddp.shard_indices(10, 0, 4) == [0, 4, 8] and ddp.shard_indices(10, 3, 4) == [3, 7, 1]
    assert sorted(sum((ddp.shard_indices(12, r, 4) for r in range(4)), [])) == list(range(12))
    with pytest.raises(ValueError):
        os.environ["WORLD_SIZE"] = "3"
        try:
            ddp.per_rank_batch(100)
        finally:
            os.environ.pop("WORLD_SIZE")


def _flat_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "edge-enhancement_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    from eeadv import ddp, runtime, trainer
    from utils import attacks as A
    runtime.allow_cpu_plumbing(True)
    ddp.setup(device="cpu")
    torch.manual_seed(rank)  # DIFFERENT initial weights: FlatGradSync must broadcast rank 0's
    model = TinyNet(2, 8, 10, 7 + rank)
    sync = ddp.FlatGradSync(model, chunks=3)
    n_par = sum(p.numel() for p in model.parameters())
    assert all(p.grad is v for p, v in zip(sync.params, sync.views)) and n_par <= sync.flat.numel() < n_par + 64
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(123)
    X, Y = torch.rand(8, 2, 8, 8, generator=g), torch.randint(0, 10, (8,), generator=g)
    idx = ddp.shard_indices(8)
    x, y = X[idx], Y[idx]
    losses = []
    for step in range(3):
        args = Args(method_name="AT", random=False, epsilon=0.1, num_steps_1=2, step_size_1=0.02, num_classes=10, attack_method="PGD", beta=6.0)
        crit = trainer.make_criterion(args)
        loss, out = trainer.train_batch(model, crit, opt, args, x, y, "cpu", sync=sync)
        losses.append(float(loss))
        assert all(p.grad is v for p, v in zip(sync.params, sync.views))
        if step == 0:
            opt.zero_grad(set_to_none=True)  # what the .loss() methods of ALP / TRADES do (attacks.py:265-266): the views must come back
            assert model.w1.grad is None
    saved = {"params": [p.detach().clone() for p in model.parameters()], "losses": losses, "flat": sync.flat.clone()}
    # a TRADES step (random start: not emulated below, but the ranks must still agree afterwards)
    args = Args(method_name="TRADES", random=False, epsilon=0.1, num_steps_1=2, step_size_1=0.02, num_classes=10, attack_method="PGD", beta=6.0)
    trainer.train_batch(model, trainer.make_criterion(args), opt, args, x, y, "cpu", sync=sync)
    saved["params_after_trades"] = [p.detach().clone() for p in model.parameters()]
    torch.save(saved, os.path.join(out_dir, "f%d.pt" % rank))
    ddp.teardown()


def test_flat_gradient_sync_two_ranks(tmp_path):
    """eeadv.ddp.FlatGradSync (the N > 1 update without DistributedDataParallel): rank 0's weights everywhere, gradients as
    views of one flat buffer that survive zero_grad(set_to_none), averaged over the ranks, identical parameters afterwards -
    and the same trajectory as one process that averages the two shards' gradients by hand."""
    port = _free_port()
    mp.spawn(_flat_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(str(tmp_path / "f0.pt"), weights_only=False)
    r1 = torch.load(str(tmp_path / "f1.pt"), weights_only=False)
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    assert torch.equal(r0["flat"], r1["flat"])  # the averaged gradient of the last step, on both ranks
    for a, b in zip(r0["params_after_trades"], r1["params_after_trades"]):
        assert torch.equal(a, b)
    from eeadv import runtime, trainer
    import torch.nn.functional as F
    runtime.allow_cpu_plumbing(True)
    try:
        net = TinyNet(2, 8, 10, 7)  # rank 0's initial weights
        opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        g = torch.Generator().manual_seed(123)
        X, Y = torch.rand(8, 2, 8, 8, generator=g), torch.randint(0, 10, (8,), generator=g)
        shards = ([0, 2, 4, 6], [1, 3, 5, 7])
        for step in range(3):
            args = Args(method_name="AT", random=False, epsilon=0.1, num_steps_1=2, step_size_1=0.02, num_classes=10, attack_method="PGD", beta=6.0)
            crit = trainer.make_criterion(args)
            grads = []
            for idx in shards:
                x, y = X[idx], Y[idx]
                adv, _, _ = trainer.attack_for_training(net, crit, args, x, y, "cpu")
                net.zero_grad()
                F.cross_entropy(net(adv), y).backward()
                grads.append([p.grad.clone() for p in net.parameters()])
            for p, g0, g1 in zip(net.parameters(), *grads):
                p.grad = (g0 + g1) / 2
            opt.step()
        for p, q in zip(net.parameters(), r0["params"]):
            np.testing.assert_allclose(p.detach().numpy(), q.numpy(), rtol=1e-5, atol=1e-6)
    finally:
        runtime.allow_cpu_plumbing(False)


def _seg_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "edge-enhancement_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    from eeadv import ddp, runtime, trainer
    runtime.allow_cpu_plumbing(True)
    ddp.setup(device="cpu")
    model = TinySegNet(2, 8, 10, 11 + rank)  # different weights per rank: rank 0's are broadcast
    model.register_buffer("stat", torch.full((3,), float(rank + 1)))  # buffers too, once, at construction
    sync = ddp.FlatGradSync(model)
    assert sync.segmented and len(sync.pieces) == 3 and float(model.stat[0]) == 1.0
    # the buffer is laid out in backward order, one 256-byte-aligned piece per segment
    assert [p.numel() for p in sync.pieces] == [(n + 63) // 64 * 64 for n in (model.w3.numel() + 5, model.w2.numel(), model.w1.numel())]
    assert model.w3.grad.data_ptr() == sync.flat.data_ptr() and model.w1.grad.data_ptr() == sync.pieces[2].data_ptr()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-2)
    g = torch.Generator().manual_seed(321)
    X, Y = torch.rand(8, 2, 8, 8, generator=g), torch.randint(0, 10, (8,), generator=g)
    idx = ddp.shard_indices(8)
    x, y = X[idx], Y[idx]
    crit = trainer.Criterion()
    issued = []
    start = sync.start
    sync.start = lambda piece=None: (issued.append(piece), start(piece))[1]
    upd = trainer._GraphedUpdate(model, crit, opt, x, y, sync)
    assert upd.segmented
    unused0 = model.unused.detach().clone()
    for step in range(3):  # the eager form of the segmented update (the captured form replays exactly these calls)
        upd.x.copy_(x)
        upd.y.copy_(y)
        loss, out = upd._body()
        assert model.unused.grad is None  # never reached by a backward: no view, SGD skips it (no weight decay on it)
    assert issued == [0, 1, 2] * 3  # one all-reduce per segment, each issued right after that segment's backward
    assert torch.equal(model.unused.detach(), unused0)
    torch.save({"params": [p.detach().clone() for p in model.parameters()], "loss": float(loss)}, os.path.join(out_dir, "s%d.pt" % rank))
    ddp.teardown()


def test_segmented_gradient_sync_two_ranks(tmp_path):
    """Round 3: the data-parallel update cut at the model's segment boundaries (trainer._GraphedUpdate with a segmented
    ddp.FlatGradSync): forward detached at the cuts, backward segment by segment, the all-reduce of a segment's piece issued
    before the next segment's backward.  Identical parameters on both ranks, equal to ONE process running an ordinary
    forward / backward per shard and averaging the gradients by hand; an unused parameter keeps grad None."""
    port = _free_port()
    mp.spawn(_seg_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(str(tmp_path / "s0.pt"), weights_only=False)
    r1 = torch.load(str(tmp_path / "s1.pt"), weights_only=False)
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    import torch.nn.functional as F
    net = TinySegNet(2, 8, 10, 11)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-2)
    g = torch.Generator().manual_seed(321)
    X, Y = torch.rand(8, 2, 8, 8, generator=g), torch.randint(0, 10, (8,), generator=g)
    for step in range(3):
        grads = []
        for idx in ([0, 2, 4, 6], [1, 3, 5, 7]):
            net.zero_grad()
            F.cross_entropy(net(X[idx]), Y[idx]).backward()
            grads.append([None if p.grad is None else p.grad.clone() for p in net.parameters()])
        for p, g0, g1 in zip(net.parameters(), *grads):
            p.grad = None if g0 is None else (g0 + g1) / 2
        opt.step()
    for p, q in zip(net.parameters(), r0["params"]):
        np.testing.assert_allclose(p.detach().numpy(), q.numpy(), rtol=1e-5, atol=1e-6)
