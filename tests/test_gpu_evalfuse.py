"""Eval-mode BatchNorm folded into the convolution kernels (round 4; ee_fuse.hpp, ee_wino3x3_bn_eval_*, ee_conv3x3s2_pair_bn_eval_*).

What the reference does here: every validate() pass runs `model.eval()` and then PGD-10/50/100 (experiments_tinyimagenet.py:337,354-358),
and the inner loops of ALP / TRADES switch to eval mode too (utils/attacks.py:249, :405); BatchNorm then uses its running statistics.
The bar for the fused kernels is the UNFUSED kernel sequence they replace (convolution, then ee_bn_act_* with training = 0), which the
rest of the suite pins to the oracle / the reference: equal BIT FOR BIT, NaN patterns included (zeros compare equal whatever their sign:
the unfused backward forms `w * ((g - 0) - xhat * 0)`, whose zero can carry either sign)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from eeadv import ops as _ops
    return _ops


def _same(a, b, what):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape, what
    nan_a, nan_b = torch.isnan(a), torch.isnan(b)
    assert torch.equal(nan_a, nan_b), "%s: NaN pattern differs" % what
    ok = (a == b) | nan_a
    assert bool(ok.all()), "%s: %d of %d elements differ, max |d| = %g" % (what, int((~ok).sum()), ok.numel(), float((a - b)[~ok].abs().max()))


def _bn(C, gen):
    g = (torch.rand(C, generator=gen) + 0.5).to(DEV)
    b = (torch.randn(C, generator=gen) * 0.3).to(DEV)
    rm = (torch.randn(C, generator=gen) * 0.2).to(DEV)
    rv = (torch.rand(C, generator=gen) + 0.3).to(DEV)
    return g, b, rm, rv, 1e-5


def _wino_sets(w):
    from eeadv import functional as Fn
    return Fn.wino_sets(w)


@pytest.mark.parametrize("B,C,H", [(5, 64, 16), (3, 128, 8), (2, 128, 8), (7, 256, 4), (4, 256, 4), (1, 32, 16), (9, 64, 8)])
@pytest.mark.parametrize("with_res", [False, True])
def test_wino_bn_eval_forward_equals_the_two_kernels(ops, B, C, H, with_res):
    gen = torch.Generator().manual_seed(B * 1000 + C + H)
    x = torch.randn(B, C, H, H, generator=gen).to(DEV)
    w = (torch.randn(C, C, 3, 3, generator=gen) * (2.0 / (9 * C)) ** 0.5).to(DEV)
    res = torch.randn(B, C, H, H, generator=gen).to(DEV) if with_res else None
    g, b, rm, rv, eps = _bn(C, gen)
    u = _wino_sets(w)[0]
    want, _, _ = ops.bn_act_fwd(ops.wino3x3(x, u), res, g, b, rm, rv, 0.1, eps, False, True)
    got = ops.wino3x3_bn_eval_fwd(x, u, (rm, rv, g, b, eps), res, True)
    _same(got, want, "fused forward")
    assert float((got == 0).float().mean()) > 0.2  # the ReLU bites
    # without the ReLU, and without gamma / beta
    want2, _, _ = ops.bn_act_fwd(ops.wino3x3(x, u), res, None, None, rm, rv, 0.1, eps, False, False)
    _same(ops.wino3x3_bn_eval_fwd(x, u, (rm, rv, None, None, eps), res, False), want2, "fused forward, no affine parameters, no ReLU")


@pytest.mark.parametrize("B,C,H", [(5, 64, 16), (3, 128, 8), (2, 128, 8), (7, 256, 4), (4, 256, 4), (1, 32, 16)])
@pytest.mark.parametrize("pieces,want_dres,with_add", [(1, False, False), (2, True, False), (1, True, True), (2, False, True), (2, True, True)])
def test_wino_bn_eval_backward_equals_the_two_kernels(ops, B, C, H, pieces, want_dres, with_add):
    gen = torch.Generator().manual_seed(B * 1000 + C + H + pieces)
    y = torch.relu(torch.randn(B, C, H, H, generator=gen)).to(DEV)  # the forward's output: its sign pattern is the ReLU mask
    dy = torch.randn(B, C, H, H, generator=gen).to(DEV)
    dy2 = torch.randn(B, C, H, H, generator=gen).to(DEV) if pieces == 2 else None
    dx_add = torch.randn(B, C, H, H, generator=gen).to(DEV) if with_add else None
    w = (torch.randn(C, C, 3, 3, generator=gen) * (2.0 / (9 * C)) ** 0.5).to(DEV)
    g, b, rm, rv, eps = _bn(C, gen)
    ub = _wino_sets(w)[1]
    # the unfused sequence: BatchNorm + ReLU backward in eval mode (x only enters through xhat * 0), then the backward-data convolution
    xdummy = torch.randn(B, C, H, H, generator=gen).to(DEV)
    dz_scaled, dz, _, _ = ops.bn_act_bwd(dy, y, xdummy, g, None, None, rm, rv, eps, False, True, True, True, False, dy2)
    want = ops.wino3x3(dz_scaled, ub)
    if with_add:
        want = want + dx_add
    got, dres = ops.wino3x3_bn_eval_bwd(dy, dy2, y, ub, (rv, g, eps), want_dres, dx_add)
    _same(got, want, "fused backward-data")
    if want_dres:
        _same(dres, dz, "residual gradient")
    else:
        assert dres is None


@pytest.mark.parametrize("B,Cin,Cout,H", [(5, 64, 128, 16), (3, 128, 256, 8), (9, 256, 512, 4), (2, 32, 64, 8)])
@pytest.mark.parametrize("mt", ["222111", "111222"])
def test_s2_pair_bn_eval_equals_the_unfused_kernels(ops, B, Cin, Cout, H, mt, monkeypatch):
    from eeadv import functional as Fn
    monkeypatch.setenv("EEADV_S2_MT", mt)
    gen = torch.Generator().manual_seed(B + Cin + H)
    x = torch.randn(B, Cin, H, H, generator=gen).to(DEV)
    w3 = (torch.randn(Cout, Cin, 3, 3, generator=gen) * (2.0 / (9 * Cin)) ** 0.5).to(DEV)
    w1 = (torch.randn(Cout, Cin, 1, 1, generator=gen) * (2.0 / Cin) ** 0.5).to(DEV)
    g3, b3, rm3, rv3, eps = _bn(Cout, gen)
    g1, b1, rm1, rv1, _ = _bn(Cout, gen)
    w10f, w10b = Fn._dense_weight(w3, "s2p_f", w1), Fn._dense_weight(w3, "s2p_b", w1)
    y3, y1 = ops.conv3x3s2_pair_fwd(x, w10f, Cout)
    want3, _, _ = ops.bn_act_fwd(y3, None, g3, b3, rm3, rv3, 0.1, eps, False, True)
    want1, _, _ = ops.bn_act_fwd(y1, None, g1, b1, rm1, rv1, 0.1, eps, False, False)
    got3, got1 = ops.conv3x3s2_pair_bn_eval_fwd(x, w10f, Cout, (rm3, rv3, g3, b3, eps), (rm1, rv1, g1, b1, eps))
    _same(got3, want3, "relu(bn1(conv3x3s2))")
    _same(got1, want1, "bn_ds(conv1x1s2)")
    # backward-data
    OH = H // 2
    dy3 = torch.randn(B, Cout, OH, OH, generator=gen).to(DEV)
    dy1 = torch.randn(B, Cout, OH, OH, generator=gen).to(DEV)
    d3, _, _, _ = ops.bn_act_bwd(dy3, want3, y3, g3, None, None, rm3, rv3, eps, False, True, True, False, False)
    d1, _, _, _ = ops.bn_act_bwd(dy1, None, y1, g1, None, None, rm1, rv1, eps, False, False, True, False, False)
    want = ops.conv3x3s2_pair_bwd_data(d3, d1, w10b, Cin)
    got = ops.conv3x3s2_pair_bn_eval_bwd(dy3, want3, dy1, w10b, Cin, (rv3, g3, eps), (rv1, g1, eps))
    _same(got, want, "pair backward-data")


@pytest.mark.parametrize("B,Cin,Cout", [(100, 512, 512), (5, 512, 512), (33, 256, 512), (1, 128, 64), (64, 512, 256)])
def test_dense2x2_product_and_its_bn_eval_forms(ops, B, Cin, Cout):
    """ee_dense.hip: the 3x3 convolution on a 2x2 map as one hand-written product, against the Tensile product of the same rearranged weights
    (summation order differs: tolerance) and - the fused forms - against the plain hand-written product followed by ee_bn_act_*: bit for bit"""
    from eeadv import functional as Fn
    gen = torch.Generator().manual_seed(B + Cin + Cout)
    x = torch.randn(B, Cin, 2, 2, generator=gen).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) * (2.0 / (9 * Cin)) ** 0.5).to(DEV)
    w2, w2t = Fn._dense_weight(w, "s1"), Fn._dense_weight(w, "s1t")
    assert torch.equal(w2t, w2.t().contiguous())
    y = ops.dense2x2(x, w2)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    res = torch.randn(B, Cout, 2, 2, generator=gen).to(DEV)
    g, b, rm, rv, eps = _bn(Cout, gen)
    for r in (None, res):
        want, _, _ = ops.bn_act_fwd(y, r, g, b, rm, rv, 0.1, eps, False, True)
        _same(ops.dense2x2_bn_eval_fwd(x, w2, (rm, rv, g, b, eps), r, True), want, "fused forward")
    # backward-data: dz = (y > 0)(dy + dy2), dx = (w dz) W2^T + dx_add
    out = torch.relu(torch.randn(B, Cout, 2, 2, generator=gen)).to(DEV)
    dy, dy2 = torch.randn(B, Cout, 2, 2, generator=gen).to(DEV), torch.randn(B, Cout, 2, 2, generator=gen).to(DEV)
    dx_add = torch.randn(B, Cin, 2, 2, generator=gen).to(DEV)
    if Cout % 128 == 0:  # the backward product reduces over 4 Cout
        for pieces, add in ((1, None), (2, dx_add), (1, dx_add)):
            d2 = dy2 if pieces == 2 else None
            dz_scaled, dz, _, _ = ops.bn_act_bwd(dy, out, torch.zeros_like(dy), g, None, None, rm, rv, eps, False, True, True, True, False, d2)
            want = ops.dense2x2(dz_scaled, w2t)
            if add is not None:
                want = want + add
            got, dres = ops.dense2x2_bn_eval_bwd(dy, d2, out, w2t, (rv, g, eps), True, add)
            _same(got, want, "fused backward-data")
            _same(dres, dz, "residual gradient")


def _resnet(eval_mode=True, seed=3):
    from eeadv import models as M
    torch.manual_seed(seed)
    m = M.make_resnet(18, "tiny").to(DEV)
    # running statistics and affine parameters away from their initial (0, 1, 1, 0)
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.copy_((torch.randn(mod.num_features, generator=gen) * 0.1).to(DEV))
                mod.running_var.copy_((torch.rand(mod.num_features, generator=gen) + 0.5).to(DEV))
                mod.weight.copy_((torch.rand(mod.num_features, generator=gen) + 0.5).to(DEV))
                mod.bias.copy_((torch.randn(mod.num_features, generator=gen) * 0.1).to(DEV))
    return m.eval() if eval_mode else m.train()


def _input_gradient(model, x, y):
    from eeadv import engine
    spec = engine.LossSpec(engine.CE_SUM, y)
    xx = x.clone().requires_grad_(True)
    return engine.input_gradient(model, xx, spec)


def test_eval_mode_input_gradient_is_bit_identical_with_and_without_the_fusion(monkeypatch):
    """the whole eval-mode classifier: logits (autograd off) and d CE / d x (attack loop) through the fused blocks against the per-layer
    path (EEADV_STOCK_GLUE=evalfuse is read at import: the switch is flipped on the module)"""
    from eeadv import models as M
    m = _resnet()
    x = torch.rand(6, 3, 64, 64, device=DEV)
    y = torch.randint(0, 200, (6,), device=DEV)
    with torch.no_grad():
        logits_f = m(x)
    g_f = _input_gradient(m, x, y)
    routes = M.fallback_report(m)
    fused = [n for n, mod in m.named_modules() if isinstance(mod, torch.nn.Conv2d) and mod.__dict__.get("_ee_route", "").endswith("+bn")]
    assert len(fused) == 19, (fused, routes)  # every convolution behind the stem: 16 3x3 + 3 shortcut convolutions
    assert routes["count"] == 0, routes  # nothing of an eval-mode pass is left on MIOpen / Tensile
    monkeypatch.setattr(M, "_STOCK", frozenset(["evalfuse"]))
    with torch.no_grad():
        logits_u = m(x)
    g_u = _input_gradient(m, x, y)
    assert not any(mod.__dict__.get("_ee_route", "").endswith("+bn") for mod in m.modules() if isinstance(mod, torch.nn.Conv2d))
    # layers 1-3: the same Winograd / stride-2 kernels with and without the fold - bit for bit (checked per kernel above and, for the whole
    # model, below with the layer-4 fold switched off (EEADV_STOCK_GLUE=densefuse)); layer 4: ee_dense.hip against the Tensile product of the unfused path, another
    # summation order - rounding level on the logits, and a ReLU within rounding of zero may flip in the gradient
    assert float((logits_f - logits_u).abs().max()) <= 1e-5 * float(logits_u.abs().max())
    assert float((g_f - g_u).norm() / g_u.norm()) < 1e-2
    assert float(g_f.abs().max()) > 0
    monkeypatch.setattr(M, "_STOCK", frozenset(["densefuse"]))  # layer 4 on the per-layer path in both runs
    with torch.no_grad():
        logits_f3 = m(x)
    g_f3 = _input_gradient(m, x, y)
    assert sum(mod.__dict__.get("_ee_route", "").endswith("+bn") for mod in m.modules() if isinstance(mod, torch.nn.Conv2d)) == 14
    monkeypatch.setattr(M, "_STOCK", frozenset(["densefuse", "evalfuse"]))
    with torch.no_grad():
        logits_u3 = m(x)
    g_u3 = _input_gradient(m, x, y)
    _same(logits_f3, logits_u3, "eval-mode logits, layers 1-3 folded")
    _same(g_f3, g_u3, "eval-mode input gradient, layers 1-3 folded")


def test_train_mode_and_parameter_gradients_do_not_take_the_fused_blocks():
    """train mode keeps its batch statistics; an eval-mode forward with autograd on OUTSIDE the attack loop (somebody wants parameter
    gradients) keeps the per-layer Functions, whose backward produces them"""
    m = _resnet(eval_mode=False)
    x = torch.rand(4, 3, 64, 64, device=DEV)
    m(x)
    assert not any(mod.__dict__.get("_ee_route", "").endswith("+bn") for mod in m.modules() if isinstance(mod, torch.nn.Conv2d))
    m.eval()
    out = m(x)
    assert not any(mod.__dict__.get("_ee_route", "").endswith("+bn") for mod in m.modules() if isinstance(mod, torch.nn.Conv2d))
    out.sum().backward()
    assert m.layer1[0].conv1.weight.grad is not None and float(m.layer1[0].conv1.weight.grad.abs().max()) > 0
    with torch.no_grad():
        m(x)
    assert any(mod.__dict__.get("_ee_route", "").endswith("+bn") for mod in m.modules() if isinstance(mod, torch.nn.Conv2d))


@pytest.mark.parametrize("graph", [False, True])
def test_eval_mode_attack_fused_equals_unfused(monkeypatch, graph):
    """PGD-5 in eval mode (validate(), experiments_tinyimagenet.py:337,354-358), eager and from the captured graph: the adversarial batch
    with the fused blocks equals the per-layer path's bit for bit"""
    import utils.attacks as A
    from eeadv import engine, models as M
    monkeypatch.setenv("EEADV_GRAPH", "1" if graph else "0")
    monkeypatch.setattr(M, "_STOCK", frozenset(["densefuse"]))  # layer 4 stays on the per-layer path in both runs (its folded kernel sums in another order)
    engine.clear_graphs()
    m = _resnet()

    class Args:
        random, epsilon = True, 16 / 255
    x = torch.rand(8, 3, 64, 64, device=DEV)
    y = torch.randint(0, 200, (8,), device=DEV)
    noise = torch.zeros_like(x).uniform_(-16 / 255, 16 / 255)
    adv_f = A.PGD(m, Args, x, y, 5, 2 / 255, noise=noise)
    engine.clear_graphs()
    monkeypatch.setattr(M, "_STOCK", frozenset(["densefuse", "evalfuse"]))
    adv_u = A.PGD(m, Args, x, y, 5, 2 / 255, noise=noise)
    engine.clear_graphs()
    _same(adv_f, adv_u, "adversarial batch")
    assert float((adv_f - x).abs().max()) > 1 / 255
