"""TRAIN-mode BatchNorm with its batch statistics exchanged across the kernel boundary (round 4; ee_fuse.hpp: TrainBn,
ee_wino3x3_stats_f32 / ee_wino3x3_bn_train_pre_f32 / ee_conv3x3s2_pair_stats_fwd_f32).

The training drivers attack in train mode (experiments_tinyimagenet.py:234-282: `model.train()`, then PGD): resnet.py:44-49's
conv1 -> bn1 -> relu -> conv2 runs with BATCH statistics 10 times per step.  The bar is the unfused kernel sequence (convolution,
ee_bn_act_fwd_f32(training = 1), convolution), which the rest of the suite pins: the convolutions' raw outputs BIT-equal, the statistics
and everything behind them within rounding (they are summed in another order: 1e-6 relative), run-to-run bit-reproducible."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from eeadv import ops as _ops
    return _ops


def _close(a, b, rel, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = float(b.abs().max()) + 1e-30
    err = float((a - b).abs().max())
    assert err <= rel * scale, "%s: max |d| = %.3e of max |ref| = %.3e (allowed %.1e relative)" % (what, err, scale, rel)


@pytest.mark.parametrize("B,C,H", [(100, 64, 16), (5, 64, 16), (7, 128, 8), (2, 128, 8), (9, 256, 4), (4, 256, 4), (3, 32, 16)])
def test_conv_bn_relu_conv_across_the_kernel_boundary(ops, B, C, H):
    from eeadv import functional as Fn
    gen = torch.Generator().manual_seed(B + C + H)
    x = torch.relu(torch.randn(B, C, H, H, generator=gen)).to(DEV)
    w1 = (torch.randn(C, C, 3, 3, generator=gen) * (2.0 / (9 * C)) ** 0.5).to(DEV)
    w2 = (torch.randn(C, C, 3, 3, generator=gen) * (2.0 / (9 * C)) ** 0.5).to(DEV)
    gamma, beta = (torch.rand(C, generator=gen) + 0.5).to(DEV), (torch.randn(C, generator=gen) * 0.2).to(DEV)
    u1, u2 = Fn.wino_sets(w1)[0], Fn.wino_sets(w2)[0]
    rm_a, rv_a = torch.full((C,), 0.1, device=DEV), torch.full((C,), 0.9, device=DEV)
    rm_b, rv_b = rm_a.clone(), rv_a.clone()
    # unfused
    c1 = ops.wino3x3(x, u1)
    a1, sm, si = ops.bn_act_fwd(c1, None, gamma, beta, rm_a, rv_a, 0.1, 1e-5, True, True)
    want = ops.wino3x3(a1, u2)
    # across the boundary
    c1b, stats = ops.wino3x3_stats(x, u1)
    assert torch.equal(c1b, c1)
    pm = c1.double().mean((2, 3)).t()  # [C, B]
    pm2 = ((c1.double() - c1.double().mean((2, 3), keepdim=True)) ** 2).sum((2, 3)).t()
    _close(stats[:, :, 0], pm, 2e-6, "per-image means")
    _close(stats[:, :, 1], pm2, 1e-5, "per-image M2")
    got, smb, sib = ops.wino3x3_bn_train_pre(c1b, stats, H * H, gamma, beta, 1e-5, 0.1, rm_b, rv_b, u2)
    _close(smb, sm, 2e-6, "save_mean")
    _close(sib, si, 2e-6, "save_invstd")
    _close(rm_b, rm_a, 2e-6, "running_mean")
    _close(rv_b, rv_a, 2e-6, "running_var")
    _close(got, want, 2e-5, "conv2(relu(bn1(conv1 x)))")
    # bit-reproducible
    rm_c, rv_c = torch.full((C,), 0.1, device=DEV), torch.full((C,), 0.9, device=DEV)
    c1c, stats_c = ops.wino3x3_stats(x, u1)
    again, smc, _ = ops.wino3x3_bn_train_pre(c1c, stats_c, H * H, gamma, beta, 1e-5, 0.1, rm_c, rv_c, u2)
    assert torch.equal(again, got) and torch.equal(smc, smb) and torch.equal(rm_c, rm_b) and torch.equal(stats_c, stats)
    # no running statistics: allowed
    ops.wino3x3_bn_train_pre(c1b, stats, H * H, None, None, 1e-5, 0.1, None, None, u2)


@pytest.mark.parametrize("B,C", [(100, 64), (5, 64), (3, 32), (9, 128)])
def test_batchnorm_backward_across_the_kernel_boundary(ops, B, C):
    """conv2^T -> BatchNorm / ReLU backward (train mode, mask from x) -> conv1^T on 16x16 maps: the sums epilogue + merge / apply prologue
    against ee_wino3x3_f32 | ee_bn_act_bwd2_f32(training = 1) | ee_wino3x3_f32 - the convolutions' raw outputs bit-equal, the rest within
    rounding (the batch sums are taken in another order), bit-reproducible."""
    from eeadv import functional as Fn
    H = 16
    gen = torch.Generator().manual_seed(B + C)
    c1 = torch.randn(B, C, H, H, generator=gen).to(DEV)   # the BatchNorm's input (conv1's raw output)
    dc2 = torch.randn(B, C, H, H, generator=gen).to(DEV)  # the gradient of conv2's output
    w1 = (torch.randn(C, C, 3, 3, generator=gen) * (2.0 / (9 * C)) ** 0.5).to(DEV)
    w2 = (torch.randn(C, C, 3, 3, generator=gen) * (2.0 / (9 * C)) ** 0.5).to(DEV)
    gamma, beta = (torch.rand(C, generator=gen) + 0.5).to(DEV), (torch.randn(C, generator=gen) * 0.2).to(DEV)
    sm = c1.mean((0, 2, 3)).contiguous()
    si = (1.0 / torch.sqrt(c1.var((0, 2, 3), unbiased=False) + 1e-5)).contiguous()
    u1b, u2b = Fn.wino_sets(w1)[1], Fn.wino_sets(w2)[1]
    d_a1 = ops.wino3x3(dc2, u2b)
    d_c1 = ops.bn_act_bwd(d_a1, None, c1, gamma, sm, si, None, None, 1e-5, True, True, True, False, False, None, beta)[0]
    want = ops.wino3x3(d_c1, u1b)
    d_a1b, sums = ops.wino3x3_bwd_sums(dc2, u2b, c1, sm, si, gamma, beta)
    assert torch.equal(d_a1b, d_a1)
    shape = (1, C, 1, 1)
    dz = torch.where(((c1 - sm.view(shape)) * (si * gamma).view(shape) + beta.view(shape)) > 0, d_a1, torch.zeros_like(d_a1)).double()
    xhat = ((c1 - sm.view(shape)) * si.view(shape)).double()
    _close(sums[:, :, 0], dz.sum((2, 3)).t(), 1e-5, "per-image sums of dz")
    _close(sums[:, :, 1], (dz * xhat).sum((2, 3)).t(), 1e-5, "per-image sums of dz * xhat")
    got = ops.wino3x3_bn_train_bwd_pre(d_a1b, c1, sums, H * H, sm, si, gamma, beta, u1b)
    _close(got, want, 2e-5, "conv1^T(bn_relu_backward(conv2^T dc2))")
    again_d, again_s = ops.wino3x3_bwd_sums(dc2, u2b, c1, sm, si, gamma, beta)
    assert torch.equal(again_s, sums) and torch.equal(ops.wino3x3_bn_train_bwd_pre(again_d, c1, again_s, H * H, sm, si, gamma, beta, u1b), got)


@pytest.mark.parametrize("B,Cin,Cout,H", [(100, 64, 128, 16), (5, 64, 128, 16), (7, 128, 256, 8), (2, 128, 256, 8)])
@pytest.mark.parametrize("mt", ["222111", "111222"])
def test_pair_statistics_feed_the_consumer(ops, B, Cin, Cout, H, mt, monkeypatch):
    from eeadv import functional as Fn
    monkeypatch.setenv("EEADV_S2_MT", mt)
    gen = torch.Generator().manual_seed(B + Cin + H)
    x = torch.relu(torch.randn(B, Cin, H, H, generator=gen)).to(DEV)
    w3 = (torch.randn(Cout, Cin, 3, 3, generator=gen) * (2.0 / (9 * Cin)) ** 0.5).to(DEV)
    wd = (torch.randn(Cout, Cin, 1, 1, generator=gen) * (2.0 / Cin) ** 0.5).to(DEV)
    w2 = (torch.randn(Cout, Cout, 3, 3, generator=gen) * (2.0 / (9 * Cout)) ** 0.5).to(DEV)
    gamma, beta = (torch.rand(Cout, generator=gen) + 0.5).to(DEV), (torch.randn(Cout, generator=gen) * 0.2).to(DEV)
    w10 = Fn._dense_weight(w3, "s2p_f", wd)
    y3, y1 = ops.conv3x3s2_pair_fwd(x, w10, Cout)
    y3b, y1b, stats, cnt = ops.conv3x3s2_pair_stats_fwd(x, w10, Cout)
    assert torch.equal(y3b, y3) and torch.equal(y1b, y1)
    assert stats.shape[1] * cnt == B * (H // 2) ** 2
    rm_a, rv_a = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    rm_b, rv_b = rm_a.clone(), rv_a.clone()
    a1, sm, si = ops.bn_act_fwd(y3, None, gamma, beta, rm_a, rv_a, 0.1, 1e-5, True, True)
    u2 = Fn.wino_sets(w2)[0]
    want = ops.wino3x3(a1, u2)
    got, smb, sib = ops.wino3x3_bn_train_pre(y3b, stats, cnt, gamma, beta, 1e-5, 0.1, rm_b, rv_b, u2)
    _close(smb, sm, 2e-6, "save_mean")
    _close(sib, si, 2e-6, "save_invstd")
    _close(rv_b, rv_a, 2e-6, "running_var")
    _close(got, want, 2e-5, "conv2(relu(bn1(pair conv x)))")


def _resnet(seed=5):
    from eeadv import models as M
    torch.manual_seed(seed)
    return M.make_resnet(18, "tiny").to(DEV).train()


def _grad_and_state(model, x, y):
    from eeadv import engine
    spec = engine.LossSpec(engine.CE_SUM, y)
    g = engine.input_gradient(model, x.clone().requires_grad_(True), spec)
    stats = torch.cat([b.detach().flatten().float() for n, b in model.named_buffers() if "running" in n])
    return g, stats


def test_train_mode_attack_gradient_with_and_without_the_boundary_exchange(monkeypatch):
    """the train-mode classifier inside the attack loop: input gradient and running statistics with bn1's statistics crossing the kernel
    boundary against the BatchNorm-launch path (EEADV_STOCK_GLUE=trainfuse).  The two differ by the summation order of the statistics; a ReLU
    whose pre-activation sits within rounding of zero may flip between them (one flip moves the gradient by ~2e-3 of its norm)."""
    import copy
    from eeadv import models as M
    monkeypatch.setattr(M, "_TRAINFUSE_MAPS", frozenset([16, 8, 4]))  # every block the kernels take, not only the default (16x16 maps)
    m_f = _resnet()
    m_u = copy.deepcopy(m_f)
    x = torch.rand(16, 3, 64, 64, device=DEV)
    y = torch.randint(0, 200, (16,), device=DEV)
    g_f, s_f = _grad_and_state(m_f, x, y)
    fused = [n for n, mod in m_f.named_modules() if isinstance(mod, torch.nn.Conv2d) and "(train)" in mod.__dict__.get("_ee_route", "")]
    assert len(fused) == 6, fused  # conv2 of the six blocks of layers 1-3
    again, s_again = _grad_and_state(copy.deepcopy(m_u), x, y)
    monkeypatch.setattr(M, "_STOCK", frozenset(["trainfuse"]))
    g_u, s_u = _grad_and_state(m_u, x, y)
    assert not any("(train)" in mod.__dict__.get("_ee_route", "") for mod in m_u.modules() if isinstance(mod, torch.nn.Conv2d))
    _close(s_f, s_u, 5e-6, "running statistics after one train-mode forward")
    rel = float((g_f - g_u).norm() / g_u.norm())
    assert rel < 1e-2, rel
    assert torch.equal(again, g_f) and torch.equal(s_again, s_f)  # the same bits from a fresh copy of the model


def test_update_step_keeps_the_batchnorm_launches():
    """a train-mode forward whose backward wants PARAMETER gradients (the update of every training step) does not take the boundary path"""
    m = _resnet()
    x = torch.rand(8, 3, 64, 64, device=DEV)
    m(x).sum().backward()
    assert not any("(train)" in mod.__dict__.get("_ee_route", "") for mod in m.modules() if isinstance(mod, torch.nn.Conv2d))
    assert m.layer1[0].bn1.weight.grad is not None and m.layer1[0].conv2.weight.grad is not None
