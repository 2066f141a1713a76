"""f5: SyncBatchNorm (ImageNet/experiments_imagenet.py:125, free_imagenet/AT_free_imagenet_ddp.py:149) as eeadv.syncbn - the fused
BatchNorm kernels around ONE collective per layer and direction.  World size 2: over gloo on the host (the torch restatement of the
kernels' formulas; what is under test is the exchange, the rank-ordered merge and the gradient bookkeeping) and, on the GPU box, two
ranks time-sharing cuda:0 through the HIP kernels.  Reference for both: one process, BatchNorm over the concatenated batch, float64."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(C=8, H=4):
    g = torch.Generator().manual_seed(77)
    X = torch.randn(8, C, H, H, generator=g) * 2 + 0.5
    R = torch.randn(8, C, H, H, generator=g)
    U = torch.randn(8, C, H, H, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    return X, R, U, gamma, beta


def _reference(relu, res):
    """float64, one process, the whole batch"""
    X, R, U, gamma, beta = (t.double() for t in _data())
    x = X.clone().requires_grad_(True)
    r = R.clone().requires_grad_(True)
    g, b = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(8, dtype=torch.float64), torch.ones(8, dtype=torch.float64)
    y = torch.nn.functional.batch_norm(x, rm, rv, g, b, True, 0.1, 1e-5)
    if res:
        y = y + r
    if relu:
        y = torch.relu(y)
    (y * U).sum().backward()
    return y.detach(), x.grad, (r.grad if res else None), g.grad, b.grad, rm, rv


def _worker(rank, world, port, out_dir, device):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "edge-enhancement_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    import torch.distributed as dist
    from eeadv import models, runtime, syncbn
    if device == "cpu":
        runtime.allow_cpu_plumbing(True)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, R, U, gamma, beta = _data()
    idx = list(range(rank, 8, world))  # a strided shard, as DistributedSampler deals them
    out = {}
    for relu in (False, True):
        for res in (False, True):
            bn = models.BatchNorm2d(8)
            with torch.no_grad():
                bn.weight.copy_(gamma)
                bn.bias.copy_(beta)
            holder = torch.nn.Sequential(bn)
            sbn = syncbn.convert_sync_batchnorm(holder)[0].to(device).train()
            assert isinstance(sbn, syncbn.SyncBatchNorm2d) and list(sbn.state_dict().keys()) == list(bn.state_dict().keys())
            x = X[idx].to(device).requires_grad_(True)
            r = R[idx].to(device).requires_grad_(True) if res else None
            y = models.bn_act(sbn, x, r, relu=relu)
            (y * U[idx].to(device)).sum().backward()
            out[(relu, res)] = dict(idx=idx, y=y.detach().cpu(), dx=x.grad.cpu(), dr=None if r is None else r.grad.cpu(), dg=sbn.weight.grad.cpu(),
                                    db=sbn.bias.grad.cpu(), rm=sbn.running_mean.cpu(), rv=sbn.running_var.cpu(), nbt=int(sbn.num_batches_tracked))
    # eval mode: running statistics, no collective (a rank-local batch_norm)
    sbn.eval()
    with torch.no_grad():
        ye = models.bn_act(sbn, X[idx].to(device), None, relu=True).cpu()
    want = torch.relu(torch.nn.functional.batch_norm(X[idx], sbn.running_mean.cpu(), sbn.running_var.cpu(), gamma, beta, False, 0.0, 1e-5))
    assert torch.allclose(ye, want, atol=1e-5)
    torch.save(out, os.path.join(out_dir, "sbn%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _check(tmp_path):
    parts = [torch.load(str(tmp_path / ("sbn%d.pt" % r)), weights_only=False) for r in range(2)]
    for relu in (False, True):
        for res in (False, True):
            y64, dx64, dr64, dg64, db64, rm64, rv64 = _reference(relu, res)
            a, b = parts[0][(relu, res)], parts[1][(relu, res)]
            what = "relu=%s residual=%s" % (relu, res)
            for p in (a, b):
                np.testing.assert_allclose(p["y"].numpy(), y64[p["idx"]].numpy(), atol=2e-6, rtol=1e-6, err_msg=what)
                np.testing.assert_allclose(p["dx"].numpy(), dx64[p["idx"]].numpy(), atol=5e-6, rtol=1e-5, err_msg=what)
                if res:
                    np.testing.assert_allclose(p["dr"].numpy(), dr64[p["idx"]].numpy(), atol=1e-6, rtol=1e-6, err_msg=what)
                # running statistics of the GLOBAL batch (unbiased variance over all 8 * 16 elements), identical on both ranks
                np.testing.assert_allclose(p["rm"].numpy(), rm64.numpy(), atol=1e-6, err_msg=what)
                np.testing.assert_allclose(p["rv"].numpy(), rv64.numpy(), atol=1e-6, rtol=1e-6, err_msg=what)
                assert p["nbt"] == 1
            assert torch.equal(a["rm"], b["rm"]) and torch.equal(a["rv"], b["rv"])  # the rank-ordered merge: the same bits on every rank
            # parameter gradients stay local (the training step's gradient exchange sums / averages them): the two ranks' add up to the full batch's
            np.testing.assert_allclose((a["dg"] + b["dg"]).numpy(), dg64.numpy(), atol=2e-5, rtol=1e-5, err_msg=what)
            np.testing.assert_allclose((a["db"] + b["db"]).numpy(), db64.numpy(), atol=2e-5, rtol=1e-5, err_msg=what)


def test_syncbn_two_ranks_gloo_host(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "cpu"), nprocs=2, join=True)
    _check(tmp_path)


@pytest.mark.gpu
def test_syncbn_two_ranks_sharing_the_gpu(tmp_path):
    """the same through the HIP kernels (ee_syncbn_*_f32): two processes on cuda:0, the [C, 3] / [C, 2] exchanges staged over gloo"""
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "cuda:0"), nprocs=2, join=True)
    _check(tmp_path)


@pytest.mark.gpu
def test_syncbn_one_rank_equals_the_fused_batchnorm():
    """no process group: the exchange is the identity - output, input gradient, parameter gradients and running statistics equal
    ee_bn_act_fwd / bwd on the same batch (to rounding: the split kernels partition the sums differently) on a ResNet-50-sized map"""
    from eeadv import models, syncbn
    dev = "cuda:0"
    torch.manual_seed(5)
    x = (torch.randn(8, 64, 56, 56, device=dev) * 1.5 + 0.3)
    u = torch.randn_like(x)
    res = torch.randn_like(x)
    outs = []
    for kind in ("plain", "sync"):
        bn = models.BatchNorm2d(64).to(dev).train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_()
            torch.manual_seed(6)
            bn.weight.copy_(torch.rand(64, device=dev) + 0.5)
            bn.bias.copy_(torch.randn(64, device=dev))
        if kind == "sync":
            bn = syncbn.convert_sync_batchnorm(torch.nn.Sequential(bn))[0]
        xi, ri = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        y = models.bn_act(bn, xi, ri, relu=True)
        (y * u).sum().backward()
        outs.append((y.detach(), xi.grad, ri.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()))
    for a, b, tol in zip(outs[0], outs[1], (2e-6, 2e-5, 0, 2e-4, 2e-4, 1e-6, 1e-6)):
        assert float((a - b).abs().max()) <= tol * max(1.0, float(a.abs().max())), (float((a - b).abs().max()), tol)
