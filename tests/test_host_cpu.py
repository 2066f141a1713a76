"""CPU: host logic of the boundary modules (no GPU, no kernels): opt-in torch plumbing path against the
oracle, refusal of CPU tensors by default, HFS operator matrices, Add_Square, helpers, model zoo."""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import ref_path as R
from tiny_models import Args, TinyNet


@pytest.fixture()
def plumbing():
    from eeadv import runtime
    runtime.allow_cpu_plumbing(True)
    yield
    runtime.allow_cpu_plumbing(False)


def test_cpu_tensors_are_refused_without_opt_in():
    import utils.attacks as A
    import utils.core as C
    net = TinyNet(2, 8, 10, 1)
    x, y = torch.rand(2, 2, 8, 8), torch.tensor([1, 2])
    with pytest.raises(RuntimeError, match="ROCm device only"):
        A.PGD(net, Args(random=False, epsilon=0.1), x, y, 1, 0.01)
    with pytest.raises(RuntimeError, match="ROCm device only"):
        C.CannyFilter_step125_1()(torch.rand(1, 3, 8, 8), high_threshold=0.3)
    with pytest.raises(RuntimeError, match="ROCm device only"):
        C.HighFreqSuppress(8, 8, 2)(torch.rand(1, 1, 8, 8))


def test_plumbing_attacks_equal_oracle(plumbing, golden):
    import utils.attacks as A
    G = golden("pgd_steps")
    x0, y = torch.from_numpy(G["x0"]), torch.from_numpy(G["y"])
    eps, alpha = [float(v) for v in G["eps_alpha"]]
    xa = A.PGD(TinyNet(2, 8, 10, 31), Args(random=True, epsilon=eps), x0, y, 6, alpha, noise=torch.from_numpy(G["pgd_noise"]))
    assert np.array_equal(xa.numpy(), G["pgd_final"])
    xc, tl = A.targeted_PGD(TinyNet(2, 8, 10, 33), Args(random=True, epsilon=eps), x0, y, 6, alpha, 10, "cpu",
                            noise=torch.from_numpy(G["tpgd_noise"]), label_offset=torch.from_numpy(G["tpgd_offset"]))
    assert np.array_equal(xc.numpy(), G["tpgd_final"]) and np.array_equal(tl.numpy(), G["tpgd_target"])
    assert np.array_equal(A.FGSM(TinyNet(2, 8, 10, 34), x0, y, targeted=True, step_size=0.007).numpy(), G["fgsm_t_final"])
    G2 = golden("avmix_cw")
    x0, y = torch.from_numpy(G2["x0"]), torch.from_numpy(G2["y"])
    eps, alpha = [float(v) for v in G2["eps_alpha"]]
    av = A.AVmixup(Args(random=True, epsilon=eps), 2.0, 1.0, 0.1, alpha, 5, num_classes=10, device="cpu")
    xm, ym = av.perturb(TinyNet(2, 8, 10, 51), x0, torch.eye(10)[y], noise=torch.from_numpy(G2["av_noise"]), beta=G2["av_beta"])
    assert np.array_equal(xm.numpy(), G2["av_x"]) and np.array_equal(ym.numpy(), G2["av_y"])
    net = TinyNet(2, 8, 10, 52)
    adv, p = A.CWLinfAttack(x0, net(x0).argmax(1), net, eps, None, eps, max_iters=4, target=torch.from_numpy(G2["cw_target"]),
                            n_class=10, cur_device="cpu", noise=torch.from_numpy(G2["cw_noise"]))
    assert np.array_equal(adv.numpy(), G2["cw_adv"])
    # TRADES / ALP loops (cannot run on CPU in the reference: device='cuda' is hard-coded there)
    torch.manual_seed(0)
    net = TinyNet(2, 8, 10, 3)
    nz = torch.randn(x0.shape)
    tr, rtr = A.Trades(alpha, eps, 3, 6.0), R.Trades(alpha, eps, 3, 6.0)
    logits = net(x0)
    assert torch.equal(tr.PGD_Linf(net, x0, logits, noise=nz), rtr.PGD_Linf(net, x0, logits, noise=nz))
    assert torch.equal(A.ALP(alpha, eps, 3).PGD_Linf(net, x0, y, noise=nz), R.ALP(alpha, eps, 3).PGD_Linf(net, x0, y, noise=nz))
    assert torch.equal(tr.PGD_L2(net, x0, logits, noise=nz), rtr.PGD_L2(net, x0, logits, noise=nz))


def test_plumbing_edge_modules_equal_reference_golden(plumbing, golden):
    import utils.core as C
    G = golden("edge125")
    for name in ["rand_tiny", "rect_mnist", "ramp_thr"]:
        x = torch.from_numpy(G[name + "__x"].copy()).requires_grad_(True)
        alpha, high = [float(v) for v in G[name + "__alpha_high"]]
        e = C.CannyFilter_step125_1(alpha=alpha)(x, low_threshold=high / 2, high_threshold=high, hysteresis=True)
        assert np.array_equal(e.detach().numpy().astype(np.uint8), G[name + "__edge"])
    Gc = golden("canny_full_unpinned")
    for cls in ("CannyFilter", "CannyFilter_BPDA"):
        x = torch.from_numpy(Gc["rand_rgb__x"].copy())
        alpha, low, high = [float(v) for v in Gc["rand_rgb__alpha_low_high"]]
        e = getattr(C, cls)(alpha=alpha)(x, low_threshold=low, high_threshold=high, hysteresis=True)
        assert np.array_equal(e.numpy(), Gc["rand_rgb__" + cls + "__edge"])
    assert sorted(k for k in C.CannyFilter().state_dict()) == ["weight_directional", "weight_gaussian", "weight_hysteresis",
                                                               "weight_sobel_x", "weight_sobel_y"]
    assert np.array_equal(np.stack(C.get_thin_kernels()), np.stack(R.get_thin_kernels()))
    Gk = golden("kernels")
    np.testing.assert_array_equal(C.get_gaussian_kernel(3, 0, 1), Gk["gauss_k3_mu0_s1"])
    np.testing.assert_array_equal(C.get_sobel_kernel(3), Gk["sobel_k3"])
    np.testing.assert_array_equal(C.HighFreqSuppress(64, 64, 8).temp.numpy()[0, 0, :, :, 0].astype(np.uint8), Gk["hfs_mask_64_64_8"])


@pytest.mark.parametrize("H,W,r", [(64, 64, 8), (28, 28, 4), (7, 9, 2), (9, 7, 2), (32, 32, 4)])
def test_hfs_operator_equals_fft_restatement(H, W, r):
    from eeadv.hfs import HFSOperator, hfs_matrices, keep_set
    assert keep_set(64, 8) == list(range(0, 8)) + list(range(56, 64))
    torch.manual_seed(1)
    x = torch.rand(2, 3, H, W)
    ref = R.HighFreqSuppress(H, W, r)
    xr = x.clone().requires_grad_(True)
    y_ref = ref(xr)
    op = HFSOperator(H, W, r, "cpu")
    np.testing.assert_allclose(op.forward(x).numpy(), y_ref.detach().numpy(), atol=2e-6)
    u = torch.randn_like(x)
    (y_ref * u).sum().backward()
    np.testing.assert_allclose(op.adjoint(u).numpy(), xr.grad.numpy(), atol=2e-6)
    Ar, Ai, B1, B2 = hfs_matrices(H, W, r)
    # float64 operator against a float64 FFT: the factorisation itself is exact to rounding
    xd = x[0, 0].double().numpy()
    z = np.fft.fft2(xd) * R.O.hfs_mask(H, W, r)
    want = np.fft.irfft2(z[:, : W // 2 + 1], s=(H, W))
    np.testing.assert_allclose(Ar @ xd @ B1 + Ai @ xd @ B2, want, atol=1e-12)


def test_add_square_plumbing_equals_restatement(plumbing):
    import utils.core as C
    torch.manual_seed(2)
    for (B, Cn, n, nq) in [(2, 3, 16, 3), (2, 1, 28, 1)]:
        ref, mod = R.Add_Square(Cn, n, 0.1, n_queries=nq), C.Add_Square(Cn, n, 0.1, n_queries=nq)
        assert mod.square_sizes("cpu")[0] == ref.sizes()
        d = ref.draw(B)
        x = torch.rand(B, Cn, n, n)
        assert torch.equal(mod(x, {"stripe": d["stripe"], "sq_pos": d["sq_pos"], "sq_sign": d["sq_sign"].reshape(nq, Cn)}), ref(x, d))
    m = C.Add_Square(3, 64, 16 / 255, n_queries=1)
    assert m.square_sizes("cpu")[0] == [57] and C.Add_Square(1, 28, 0.3, n_queries=1).square_sizes("cpu")[0] == [25]
    assert [m.p_selection(i) for i in (0, 11, 51, 201, 501, 1001, 2001, 4001, 6001, 8001)] == [0.8 / d for d in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512)]


def test_helpers(tmp_path):
    from utils import helper as Hh
    cfg = tmp_path / "c.yml"
    cfg.write_text("method_name: 'TRADES'\nstep_size_1: 0.0078\nbatch_size: 100\nstep_size_1: 0.0039\nrandom: true\n")
    ns = argparse.Namespace(config=str(cfg), evaluate=False, attack_method="PGD", no_cuda=True)
    c = Hh.parse_config_file(ns)
    assert c.method_name == "TRADES" and c.step_size_1 == 0.0039 and c.batch_size == 100 and c.no_cuda is True  # last duplicate key wins
    assert c["random"] is True
    with pytest.raises(AttributeError):
        c.type_canny
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
    lrs = []
    for ep in (0, 25, 26, 37, 38, 49):
        Hh.adjust_learning_rate_1(opt, ep, 0.1, 50)
        lrs.append(round(opt.param_groups[0]["lr"], 6))
    assert lrs == [0.1, 0.1, 0.01, 0.01, 0.001, 0.001]
    Hh.adjust_learning_rate(opt, 61, 0.1)
    assert abs(opt.param_groups[0]["lr"] - 0.001) < 1e-12
    Hh.adjust_learning_rate_free(opt, 8, 0.1, 4)
    assert abs(opt.param_groups[0]["lr"] - 0.01) < 1e-12
    m = Hh.AverageMeter()
    m.update(2.0, 2)
    m.update(4.0, 2)
    assert m.avg == 3.0 and m.val == 4.0 and m.count == 4
    z = torch.tensor([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1], [0.2, 0.3, 0.5]])
    p1, p2 = Hh.accuracy(z, torch.tensor([1, 2, 2]), topk=(1, 2))
    assert abs(p1.item() - 200 / 3) < 1e-4 and abs(p2.item() - 200 / 3) < 1e-4
    Hh.set_seed(3)
    a = torch.rand(2)
    Hh.set_seed(3)
    assert torch.equal(a, torch.rand(2))
    f, b = tmp_path / "a.pth", tmp_path / "b.pth"
    Hh.save_checkpoint({"epoch": 1}, True, str(f), str(b))
    assert torch.load(str(b), weights_only=True)["epoch"] == 1


def test_model_zoo_matches_reference_initialisation(golden):
    """Same seed -> same weights as the reference's plain models (checksums stored by make_golden.py)."""
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "edge-enhancement_amd")
    for sub in ("MNIST", "Tiny_ImageNet", "ImageNet"):
        sys.path.insert(0, os.path.join(pkg, sub))
    from models_mnist import Net_2, Net2_EE_square
    from models_tinyimagenet import resnet18, resnet18_EE_square
    import models_imagenet
    G = golden("e2e")
    ck = lambda m: sum(float(p.double().abs().sum()) for p in m.state_dict().values())
    torch.manual_seed(7)
    assert ck(Net_2()) == float(G["net2_checksum"])
    torch.manual_seed(8)
    assert ck(resnet18()) == float(G["rn18_checksum"])
    ee = resnet18_EE_square(cize=64, r=8, w=1.0, with_gf=False, low=38.0, high=76.0, alpha=0, sigma=1.0,
                            type_canny="CannyFilter_step125_1", epsilon=16 / 255, n_queries=1)
    keys = set(ee.state_dict())
    assert "weight_gaussian" in keys and "conv1.weight" in keys and "layer4.1.bn2.running_var" in keys and "fc.bias" in keys
    assert not any(k.startswith("canny.") for k in keys)  # use_cuda=True step125_1: plain tensors in the reference
    assert set(resnet18().state_dict()) == keys - {"weight_gaussian"}
    assert sum(p.numel() for p in resnet18().parameters()) == 11279112  # SURVEY: 11.279 M
    r50 = models_imagenet.resnet50()
    assert sum(p.numel() for p in r50.parameters()) == 25557032 and isinstance(r50.avgpool, torch.nn.AvgPool2d)
    n2 = Net2_EE_square(r=4, w=1.0, low=25.0, high=51.0, alpha=0.3, sigma=1.0, type_canny="CannyFilter", epsilon=0.3, n_queries=1)
    assert any(k.startswith("canny.weight_") for k in n2.state_dict())  # CannyFilter registers its weights
    with pytest.raises(NotImplementedError):
        Net2_EE_square(type_canny="nope")


def test_two_branch_backward_equals_one_backward():
    """trainer.two_branch_backward (the TRADES / ALP update without autograd's per-parameter accumulation, utils/attacks.py:264-272, :421-429):
    a loss over TWO forward passes through the same parameters, backpropagated pass by pass and joined by one multi-tensor add, gives the
    gradients of loss.backward() - including parameters only ONE of the passes reaches, and the fall-backs (one branch missing / detached)."""
    import torch.nn.functional as F
    from eeadv.trainer import two_branch_backward
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 4)).double()
    extra = torch.nn.Parameter(torch.randn(4, dtype=torch.float64))  # reached by the natural pass only
    params = list(net.parameters()) + [extra]
    x, xa, y = torch.randn(5, 6, dtype=torch.float64), torch.randn(5, 6, dtype=torch.float64), torch.randint(0, 4, (5,))

    def losses():
        nat, adv = net(x) + extra, net(xa)
        return F.cross_entropy(nat, y) + 6.0 * F.kl_div(F.log_softmax(adv, 1), F.softmax(nat, 1), reduction="batchmean"), nat, adv

    for p in params:
        p.grad = None
    losses()[0].backward()
    want = [p.grad.clone() for p in params]
    for p in params:
        p.grad = torch.full_like(p, 7.0)  # stale gradients must not leak into the result
    loss, nat, adv = losses()
    two_branch_backward(loss, nat, adv, params)
    for p, w in zip(params, want):
        assert torch.allclose(p.grad, w, rtol=1e-12, atol=1e-14)
    # fall-backs: a missing or detached branch is an ordinary backward (which ACCUMULATES, as loss.backward() does)
    for p in params:
        p.grad = None
    loss, nat, adv = losses()
    two_branch_backward(loss, nat, None, params)
    for p, w in zip(params, want):
        assert torch.allclose(p.grad, w, rtol=1e-12, atol=1e-14)
    for p in params:
        p.grad = None
    nat2 = net(x) + extra
    two_branch_backward(F.cross_entropy(nat2, y), nat2, net(xa).detach(), params)
    assert all(p.grad is not None for p in params)


def test_attack_forward_context_and_route_report(monkeypatch):
    """Host logic of round 4: functional.attack_forward() marks the forward passes of the attack loop (the fused eval / train blocks are chosen
    there only), input_only_forward() also holds under no_grad; models.fallback_report lists the convolutions whose last forward went to a
    vendor library; trainer.graph_collectives_enabled needs the switch AND an RCCL process group."""
    from eeadv import functional as EF, models as M, trainer
    assert not EF.attack_forward_active() and not EF.input_only_forward()
    with torch.no_grad():
        assert EF.input_only_forward() and not EF.attack_forward_active()
    with EF.attack_forward():
        assert EF.attack_forward_active() and EF.input_only_forward()
        with EF.attack_forward():  # nests
            assert EF.attack_forward_active()
        assert EF.attack_forward_active()
    assert not EF.attack_forward_active()
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 4, 1), torch.nn.Conv2d(4, 2, 1))
    assert M.fallback_report(net) == {"count": 0, "of": 0, "layers": []}  # nothing ran yet
    M._route(net[0], "miopen")
    M._route(net[1], "ee_wino")
    rep = M.fallback_report(net)
    assert rep["of"] == 2 and rep["count"] == 1 and rep["layers"] == ["0:miopen"]
    monkeypatch.setenv("EEADV_GRAPH_COLLECTIVES", "1")
    assert trainer.graph_collectives_enabled() is False  # no process group here
    monkeypatch.setenv("EEADV_GRAPH_COLLECTIVES", "0")
    assert trainer.graph_collectives_enabled() is False


def test_miopen_db_match_report(tmp_path):
    """runtime.shipped_miopen_db_matched: None without a private copy, False when the device / MIOpen build cannot be named (no ROCm device
    here) - bench.py prints it as config.miopen_db_matched, so a silent fall-back to MIOpen's search shows on the line"""
    from eeadv import runtime
    assert runtime.shipped_miopen_db_matched(None) is None
    assert runtime.shipped_miopen_db_matched(str(tmp_path / "missing")) is None
    (tmp_path / "gfx000ff.HIP.9_9_9_deadbeef.ufdb.txt").write_text("")  # a find-db of ANOTHER build / device appeared: MIOpen did not take ours
    assert runtime.shipped_miopen_db_matched(str(tmp_path)) is False
