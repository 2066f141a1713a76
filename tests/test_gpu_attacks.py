"""GPU parity of the attack loops that round 1 left without a pinned comparison:
utils.attacks on cuda:0 (HIP kernels through the C ABI) against the reference's own recorded trajectories
(tests/golden/{targeted,linf_loops,add_square}.npz) and against the CPU oracle, eager and from a captured HIP graph.

What is exact and what is not is spelled out in tests/replay.py: the update, start and loss kernels are bit-exact; the
classifier runs on MIOpen, so a free-running attack is compared through predictions + the fraction of identical pixels,
and the step-by-step replay bounds every possible divergence by the gradient magnitude at the step where it happened.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ee_oracle as O
from oracle import ref_path as R
from replay import replay_trajectory
from tiny_models import Args, TinyBNNet, TinyNet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5  # gradient agreement GPU (MIOpen) vs reference (oneDNN), relative to the step's largest entry


@pytest.fixture(scope="module")
def A():
    import utils.attacks as attacks
    return attacks


@pytest.fixture(params=["eager", "graph"])
def mode(request, monkeypatch):
    from eeadv import engine
    monkeypatch.setenv("EEADV_GRAPH", "1" if request.param == "graph" else "0")
    engine.clear_graphs()
    yield request.param
    engine.clear_graphs()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def same_fraction(a, b):
    return float((np.asarray(a) == np.asarray(b)).mean())


def check_free_run(got, want, net_cpu, what, frac=0.99):
    """A free-running GPU attack against the reference's final iterate: same predictions, logits within the north-star
    1e-4 wherever the inputs coincide is implied by the replay; here the end points are compared."""
    got = got.detach().cpu().numpy()
    assert got.dtype == np.float32 and got.shape == want.shape
    assert same_fraction(got, want) > frac, (what, same_fraction(got, want))
    with torch.no_grad():
        la, lb = net_cpu(torch.from_numpy(got)), net_cpu(torch.from_numpy(want))
    assert torch.equal(la.argmax(1), lb.argmax(1)), what


# ---------------------------------------------------------------------------------------------------------
# a2: targeted_PGD_trick, tar_alp_imagenet, AVmixup.tar_perturb
# ---------------------------------------------------------------------------------------------------------
def test_targeted_trick_vs_reference(A, golden, mode):
    from eeadv import engine
    G = golden("targeted")
    x0, y = dev(G["x0"]), dev(G["y"])
    eps, alpha = [float(v) for v in G["eps_alpha"]]
    for tag, b in (("trick_noise", True), ("trick_clean", False)):
        net = TinyNet(2, 8, 10, 71).to(DEV)
        args = Args(random=True, epsilon=eps, prob_start_from_clean=0.5)
        xa, tl = A.targeted_PGD_trick(net, args, x0, y, 5, alpha, 10, DEV, noise=dev(G[tag + "_init"]),
                                      label_offset=dev(G[tag + "_offset"]), start_from_noise=b)
        assert np.array_equal(tl.cpu().numpy(), G[tag + "_target"]) and not xa.requires_grad
        check_free_run(xa, G[tag + "_final"], TinyNet(2, 8, 10, 71), tag)
        if mode == "eager":
            st = replay_trajectory(net, G[tag + "_xs"], G[tag + "_gs"], G["x0"], engine.LossSpec(engine.CE_SUM, dev(G[tag + "_target"])),
                                   alpha, eps, -1, final=G[tag + "_final"], tol=TOL)
            assert len(st) == 5
    net = TinyNet(2, 8, 10, 71).to(DEV)
    xa, tl = A.targeted_PGD_trick(net, Args(random=False, epsilon=eps, prob_start_from_clean=0.5), x0, y, 5, alpha, 10, DEV,
                                  label_offset=dev(G["trick_norand_offset"]))
    assert np.array_equal(tl.cpu().numpy(), G["trick_norand_target"])
    check_free_run(xa, G["trick_norand_final"], TinyNet(2, 8, 10, 71), "trick_norand")
    # the Bernoulli itself: prob_start_from_clean 1.0 never starts from noise, -1.0 always does (attacks.py:70)
    x_clean, _ = A.targeted_PGD_trick(net, Args(random=True, epsilon=eps, prob_start_from_clean=1.0), x0, y, 0, alpha, 10, DEV)
    assert torch.equal(x_clean, x0.clamp(0, 1))
    x_noise, _ = A.targeted_PGD_trick(net, Args(random=True, epsilon=eps, prob_start_from_clean=-1.0), x0, y, 0, alpha, 10, DEV)
    assert float((x_noise - x0).abs().max()) > 0 and float((x_noise - x0).abs().max()) <= eps + 1e-6


def test_tar_alp_imagenet_vs_reference(A, golden, mode):
    from eeadv import engine
    G = golden("targeted")
    x0, y2 = dev(G["x0"]), dev(G["talp_y"])
    eps, alpha = [float(v) for v in G["eps_alpha"]]
    net = TinyNet(2, 8, 1000, 72).to(DEV)
    xa, tl = A.tar_alp_imagenet(net, Args(epsilon=eps), x0, y2, 5, alpha, DEV, noise=dev(G["talp_randn"]), label_offset=dev(G["talp_offset"]))
    assert np.array_equal(tl.cpu().numpy(), G["talp_target"])
    check_free_run(xa, G["talp_final"], TinyNet(2, 8, 1000, 72), "tar_alp_imagenet")
    if mode == "eager":
        # start: x + 0.001 * randn, NOT clamped (attacks.py:344) - bit-exact from the init kernel
        start = A._randn_start(x0, dev(G["talp_randn"]))
        assert np.array_equal(start.cpu().numpy(), G["talp_xs"][0])
        replay_trajectory(net, G["talp_xs"], G["talp_gs"], G["x0"], engine.LossSpec(engine.CE_SUM, dev(G["talp_target"])), alpha, eps, -1,
                          final=G["talp_final"], tol=TOL)
    # labels drawn on the device: fmod(y + randint(1, 1000), 1000) never returns y itself
    _, tl2 = A.tar_alp_imagenet(net, Args(epsilon=eps), x0, y2, 0, alpha, DEV)
    assert tl2.dtype == torch.int64 and bool(((tl2 != y2) & (tl2 >= 0) & (tl2 < 1000)).all())


def test_avmixup_tar_perturb_vs_reference(A, golden, mode):
    from eeadv import engine
    G = golden("targeted")
    x0, y = dev(G["x0"]), dev(G["y"])
    eps, alpha = [float(v) for v in G["eps_alpha"]]
    net = TinyNet(2, 8, 10, 73).to(DEV)
    av = A.AVmixup(Args(random=True, epsilon=eps), 2.0, 1.0, 0.1, alpha, 5, num_classes=10, device=DEV)
    onehot = torch.eye(10, device=DEV)[y]
    xm, ym = av.tar_perturb(net, x0, onehot, noise=dev(G["tav_noise"]), beta=G["tav_beta"], label_offset=dev(G["tav_offset"]))
    assert xm.dtype == torch.float32 and ym.dtype == torch.float64
    np.testing.assert_array_equal(ym.cpu().numpy(), G["tav_y"])  # float64 label mix: exact
    assert same_fraction(xm.cpu().numpy(), G["tav_x"]) > 0.99
    np.testing.assert_allclose(xm.cpu().numpy(), G["tav_x"], atol=4 * alpha + 1e-6)
    if mode == "eager":
        labels = torch.fmod(onehot + dev(G["tav_offset"]), 10).to(torch.float64).contiguous()  # the [B,K] float "labels" of :492
        replay_trajectory(net, G["tav_xs"], G["tav_gs"], G["x0"], engine.LossSpec(engine.SOFTCE, labels), alpha, eps, -1, tol=TOL)
        x_last = dev(O.pgd_step(G["tav_xs"][-1], G["tav_gs"][-1], G["x0"], alpha, eps, direction=-1))
        xm2, _ = av._vertex_mix(x0, x_last, onehot, G["tav_beta"])
        assert np.array_equal(xm2.cpu().numpy(), G["tav_x"])  # vertex + mix kernel from the reference's last iterate: bit-exact


# ---------------------------------------------------------------------------------------------------------
# a4 / a5: the *_Linf loops (CE mean / KL batchmean, unclamped randn start) and Trades.PGD_L2
# ---------------------------------------------------------------------------------------------------------
def test_trades_pgd_linf_vs_reference(A, golden, mode):
    from eeadv import engine
    G = golden("linf_loops")
    x0 = dev(G["x0"])
    eps, alpha = [float(v) for v in G["eps_alpha"]]
    net = TinyNet(2, 8, 10, 91).to(DEV).train()
    logits = net(x0)  # attached, as the drivers pass it (experiments_tinyimagenet.py:258-259)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), G["trades_logits"], atol=1e-5)
    tr = A.Trades(alpha, eps, 5, 6.0)
    xa = tr.PGD_Linf(net, x0, logits, noise=dev(G["trades_randn"]))
    assert not net.training and not xa.requires_grad  # eval() side effect (attacks.py:405)
    check_free_run(xa, G["trades_final"], TinyNet(2, 8, 10, 91), "Trades.PGD_Linf")
    xa2 = tr.PGD_Linf(net, x0, logits, noise=dev(G["trades_randn"]))  # second call: cached graph in graph mode
    assert torch.equal(xa, xa2)
    if mode == "eager":
        start = A._randn_start(x0, dev(G["trades_randn"]))
        assert np.array_equal(start.cpu().numpy(), G["trades_xs"][0]) and float(start.min()) < 0
        spec = engine.LossSpec(engine.KL, dev(G["trades_logits"]))
        st = replay_trajectory(net, G["trades_xs"], G["trades_gs"], G["x0"], spec, alpha, eps, 1, final=G["trades_final"], tol=TOL, atol=1e-7)
        assert all(s["undecided"] <= 0.02 * s["n"] for s in st), st  # the bound still decides >= 98 % of the elements
    # and against the oracle run here with fresh noise (not only the recorded case)
    torch.manual_seed(5)
    nz = torch.randn(3, 2, 8, 8)
    cpu = TinyNet(2, 8, 10, 91)
    want = R.Trades(alpha, eps, 5, 6.0).PGD_Linf(cpu, torch.from_numpy(G["x0"]), cpu(torch.from_numpy(G["x0"])), noise=nz)
    got = tr.PGD_Linf(net, x0, logits, noise=nz.to(DEV))
    check_free_run(got, want.numpy(), TinyNet(2, 8, 10, 91), "Trades.PGD_Linf fresh noise")


def test_alp_and_targeted_alp_pgd_linf_vs_reference(A, golden, mode):
    from eeadv import engine
    G = golden("linf_loops")
    x0, y = dev(G["x0"]), dev(G["y"])
    eps, alpha = [float(v) for v in G["eps_alpha"]]
    net = TinyNet(2, 8, 10, 92).to(DEV).train()
    xa = A.ALP(alpha, eps, 5, 1.0).PGD_Linf(net, x0, y, noise=dev(G["alp_randn"]))
    assert not net.training
    check_free_run(xa, G["alp_final"], TinyNet(2, 8, 10, 92), "ALP.PGD_Linf")
    net3 = TinyNet(2, 8, 10, 93).to(DEV).train()
    tal = A.targeted_ALP(alpha, eps, 5, 1.0, n_class=10)
    xt = tal.tarPGD_Linf(net3, x0, y, DEV, noise=dev(G["talpc_randn"]), label_offset=dev(G["talpc_offset"]))
    assert not net3.training
    check_free_run(xt, G["talpc_final"], TinyNet(2, 8, 10, 93), "targeted_ALP.tarPGD_Linf")
    xu = tal.PGD_Linf(net3, x0, y, noise=dev(G["talpu_randn"]))
    check_free_run(xu, G["talpu_final"], TinyNet(2, 8, 10, 93), "targeted_ALP.PGD_Linf")
    if mode == "eager":
        replay_trajectory(net, G["alp_xs"], G["alp_gs"], G["x0"], engine.LossSpec(engine.CE_MEAN, y), alpha, eps, 1,
                          final=G["alp_final"], tol=TOL)
        tl = torch.fmod(y + dev(G["talpc_offset"]), 10)
        replay_trajectory(net3, G["talpc_xs"], G["talpc_gs"], G["x0"], engine.LossSpec(engine.CE_MEAN, tl), alpha, eps, -1,
                          final=G["talpc_final"], tol=TOL)


def test_trades_pgd_l2_vs_reference(A, golden):
    """attacks.py:381-401 on the HIP path (ee_l2_step_f32): per-sample RMS of the gradient and of delta are fp32 sums in a
    different order than ATen's, so this one is a tolerance test: 1e-6 absolute on iterates in [0, 1]."""
    G = golden("linf_loops")
    x0 = dev(G["x0"])
    step, eps = [float(v) for v in G["tradesl2_step_eps"]]
    net = TinyNet(2, 8, 10, 91).to(DEV).train()
    logits = net(x0)
    # one update from the reference's recorded iterate and gradient: kernel vs the reference's next iterate
    from eeadv import ops
    xs, gs = G["tradesl2_xs"], G["tradesl2_gs"]
    for k in range(len(gs)):
        x = dev(xs[k]).clone()
        ops.l2_step_(x, dev(gs[k]), x0, step, eps, 0.0, 1.0)
        want = xs[k + 1] if k + 1 < len(xs) else G["tradesl2_final"]
        np.testing.assert_allclose(x.cpu().numpy(), want, atol=5e-7, err_msg="step %d" % k)
    # the whole loop: the step is alpha * g / rms(g), so the RELATIVE error of the first KL gradient (a difference of nearly
    # equal softmaxes at x + 0.001 * randn: ~5e-4 of its largest entry, see tests/replay.py) moves x by ~alpha * 3 * 5e-4
    xa = A.Trades(step, eps, 5, 6.0).PGD_L2(net, x0, logits, noise=dev(G["tradesl2_randn"]))
    assert not net.training and not xa.requires_grad
    np.testing.assert_allclose(xa.cpu().numpy(), G["tradesl2_final"], atol=2e-4)
    d = (xa - x0).flatten(1)
    assert float((d ** 2).mean(1).sqrt().max()) <= eps * (1 + 1e-5)  # inside the RMS ball (attacks.py:395-398)


# ---------------------------------------------------------------------------------------------------------
# a14: Add_Square against the reference's own forward (its draws recorded)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["tiny", "mnist", "nq12", "nq60_resc"])
def test_add_square_vs_reference(golden, tag):
    import utils.core as C
    from eeadv import hfs as HF, ops
    G = golden("add_square")
    B, Cn, n, nq, resc = [int(v) for v in G[tag + "__cfg"]]
    eps = float(G[tag + "__eps"])
    mod = C.Add_Square(Cn, n, eps, n_queries=nq, rescale_schedule=bool(resc))
    assert mod.square_sizes(DEV)[0] == G[tag + "__sq_size"].tolist()
    draws = {"stripe": dev(G[tag + "__stripe"]), "sq_pos": dev(G[tag + "__sq_pos"]), "sq_sign": dev(G[tag + "__sq_sign"])}
    x = dev(G[tag + "__x"]).requires_grad_(True)
    y = mod(x, draws)
    (y * dev(G[tag + "__u"])).sum().backward()
    assert np.array_equal(y.detach().cpu().numpy(), G[tag + "__y"])       # ee_add_square_fwd_f32: bit for bit
    if nq == 1:
        assert np.array_equal(x.grad.cpu().numpy(), G[tag + "__gx"])      # ee_add_square_bwd_f32: bit for bit
    else:
        # n_queries > 1 (no reference config): where (x - eps) + 2 eps == x + eps exactly, torch.min / torch.max hand HALF
        # the gradient to each argument and autograd re-adds the halves and quarters in arrival order - u/2 + u/4 rounds.
        # The kernel multiplies u by the exact derivative (0 or 1): equal to 1 ulp (156 of 1426 non-zero entries differ)
        np.testing.assert_allclose(x.grad.cpu().numpy(), G[tag + "__gx"], rtol=1.2e-7, atol=0)
    # fused into the low-pass kernel's load (hfs sq_mode 1) == low-pass of the reference's Add_Square output
    r = {64: 8, 28: 4, 16: 2}[n]
    op = HF.HFSOperator(n, n, r, DEV)
    assert op.kernel is not None
    d = mod.prepare(x.detach(), draws)
    fused = op.forward_square(x.detach(), eps, d)
    assert torch.equal(fused, op.forward(dev(G[tag + "__y"])))
    # and its backward (sq_mode 2): the derivative mask is the reference's own input gradient of sum(y * 1)
    ones = torch.ones_like(x)
    mask = ops.add_square_bwd(ones, x.detach(), eps, d["stripe"], d["sq_sign"], d["sq_pos"], d["sq_size"])
    xr = torch.from_numpy(G[tag + "__x"].copy()).requires_grad_(True)
    R.Add_Square(Cn, n, eps, n_queries=nq, rescale_schedule=bool(resc))(xr, {
        "stripe": torch.from_numpy(G[tag + "__stripe"]), "sq_pos": torch.from_numpy(G[tag + "__sq_pos"]),
        "sq_sign": torch.from_numpy(G[tag + "__sq_sign"]).reshape(nq, Cn, 1, 1)}).sum().backward()
    assert np.array_equal(mask.cpu().numpy(), xr.grad.numpy())
    u = dev(G[tag + "__u"])
    assert torch.equal(op.backward_square(u, x.detach(), eps, d), op.adjoint(u) * mask)


# ---------------------------------------------------------------------------------------------------------
# a15: the free-AT repeat (BASELINE config 5's inner loop) against the oracle, state re-synchronised before every repeat
# ---------------------------------------------------------------------------------------------------------
def _free_at_pair(arch, K):
    if arch == "tinynet":
        return TinyNet(2, 8, K, 17), TinyNet(2, 8, K, 17).double(), TinyNet(2, 8, K, 17).to(DEV)
    from eeadv.models import make_resnet, make_resnet_ee
    if arch == "resnet18_EE":  # AT_hfs_canny_free_imagenet_ddp.py's model family at its defaults (r 16, w 1, low 38, high 76, sigma 1)
        def ref():
            front = R.EEFront(224, 3, 16, 1.0, 38.0, 76.0, 0.0, 1.0, "CannyFilter_step125_1", False, False)
            return R.EEModel(front, R.resnet18(num_classes=K, imagenet_pool=True)).train()
        gpu = make_resnet_ee(18, "imagenet", False, cize=224, r=16, w=1.0, with_gf=False, low=38.0, high=76.0, alpha=0.0, sigma=1.0,
                             type_canny="CannyFilter_step125_1", num_classes=K)
        return ref(), ref().double(), gpu.to(DEV).train()
    return (R.resnet50(num_classes=K, imagenet_pool=True).train(), R.resnet50(num_classes=K, imagenet_pool=True).double().train(),
            make_resnet(50, "imagenet").to(DEV).train())


def _body_state(arch, state):
    """the oracle's EE model keeps the CNN under `net.`; the product model holds it at top level (resnet_EE.py)"""
    return {k[4:]: v for k, v in state.items() if k.startswith("net.")} if arch.endswith("_EE") else state


@pytest.mark.parametrize("arch,B,size,K,batches", [("tinynet", 6, 8, 10, 2), ("resnet50", 4, 224, 1000, 1), ("resnet18_EE", 4, 224, 1000, 1)])
def test_free_at_repeat_vs_oracle(arch, B, size, K, batches):
    """eeadv.trainer.free_at_repeat (ee_add_clamp_f32 -> model fwd/bwd -> ee_freeat_update_masked_f32 -> SGD) against
    oracle.ref_path.free_at_repeat (AT_free_imagenet_ddp.py:287-309): 4 repeats per batch, the persistent noise carried over.
    Before every repeat the GPU side is loaded with the oracle's weights, momentum buffers and noise, so each repeat is
    compared on its own:

      * loss and logits within 1e-4 of the fp32 oracle (north star; 1e-5 relative for logits beyond +-10), same predictions;
      * delta: the kernel applied to the GPU's own input gradient is bit-exact (oracle C update on that gradient);
      * the input gradient itself is judged against an fp64 run of the same repeat, because fp32 gradients of a deep ReLU
        network are not reproducible across implementations AT ALL: on resnet50 / 224 / batch 4 the fp32 oracle (oneDNN) is
        15 % of the largest entry away from fp64, the GPU path 3 % (scripts/freeat_diag.py; rounding flips ReLU / max-pool
        switches).  Required: err(GPU, fp64) <= 2 x err(fp32 oracle, fp64) and no more sign disagreements with fp64 than
        twice the oracle's;
      * where GPU and oracle agree on the gradient sign, delta is identical; rows beyond the batch are untouched.

    `resnet18_EE` = the model family of AT_hfs_canny_free_imagenet_ddp.py at that script's defaults (224 x 224, HighFreqSuppress
    r 16 on the band kernel, CannyFilter_step125_1, w 1, low 38, high 76): the same repeat through the edge-enhancing front end,
    with the floors written next to the assertions."""
    from eeadv import trainer
    torch.manual_seed(3)
    C = 2 if arch == "tinynet" else 3
    cpu, cpu64, gpu = _free_at_pair(arch, K)
    lr = 0.1 if arch == "tinynet" else 0.01  # batch 4 instead of 256: the reference's 0.1 blows the resnet50 logits up to +-55 in one step
    opt_c = torch.optim.SGD(cpu.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
    opt_64 = torch.optim.SGD(cpu64.parameters(), lr=0.0)
    # the product's edge filter keeps its fixed kernels as requires_grad=False Parameters like the reference (core.py:526-547:
    # they reach model.parameters() and the checkpoints but never change); the oracle holds them as buffers
    trained = lambda m: [p for p in m.parameters() if p.requires_grad]
    opt_g = torch.optim.SGD(trained(gpu), lr=lr, momentum=0.9, weight_decay=1e-4)
    a = e = 4.0 / 255
    noise_c = torch.zeros(B + 2, C, size, size)
    noise_g = noise_c.to(DEV)
    ce = torch.nn.functional.cross_entropy
    crit = trainer.Criterion()
    for batch in range(batches):
        x = torch.rand(B, C, size, size)
        x[0, :, :2, :4] = torch.tensor([0.0, 1.0, 0.004, 0.996])  # x + delta leaves [0, 1] here: the clamp masks the gradient
        y = torch.randint(0, K, (B,))
        xg, yg = x.to(DEV), y.to(DEV)
        for rep in range(4):
            what = "%s batch %d repeat %d" % (arch, batch, rep)
            state = cpu.state_dict()
            missing = gpu.load_state_dict(_body_state(arch, state), strict=not arch.endswith("_EE"))
            assert not [k for k in missing.missing_keys if "conv" in k or "bn" in k or "fc" in k], missing
            cpu64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in state.items()})
            opt_g.load_state_dict(opt_c.state_dict())
            noise_g.copy_(noise_c)
            before = noise_c.clone()
            _, out_64, g_64 = R.free_at_repeat(cpu64, ce, opt_64, x.double(), y, before.double(), a, e, want_grad=True)
            loss_c, out_c, g_c = R.free_at_repeat(cpu, ce, opt_c, x, y, noise_c, a, e, want_grad=True)
            loss_g, out_g, g_in1 = trainer.free_at_repeat(gpu, crit, opt_g, xg, yg, noise_g, a, e, return_input_grad=True)
            # 1e-4 absolute up to |logit| = 10, 1e-5 relative above - and in the L2 sense never further from fp64 than 3 x the
            # fp32 oracle is
            np.testing.assert_allclose(out_g.cpu().numpy(), out_c.numpy(), atol=1e-4, rtol=1e-5, err_msg=what)
            assert float((out_g.cpu().double() - out_64).norm()) <= 3 * float((out_c.double() - out_64).norm()) + 1e-5 * float(out_64.norm()), what
            assert abs(float(loss_g) - float(loss_c)) < 1e-4, what
            assert torch.equal(out_g.argmax(1).cpu(), out_c.argmax(1)), what
            s = x + before[:B]
            g_g = (g_in1.cpu() * ((s >= 0) & (s <= 1))).numpy()  # the clamp mask of :290, as the kernel applies it
            got = noise_g[:B].cpu().numpy()
            assert np.array_equal(got, O.freeat_update(before[:B].numpy(), g_g, a, e)), what  # update kernel: bit-exact
            assert torch.equal(noise_g[B:].cpu(), noise_c[B:]) and float(np.abs(got).max()) <= np.float32(e)
            g_c, g_64 = g_c.numpy(), g_64.numpy()
            scale = np.abs(g_64).max()
            err_g, err_c = np.abs(g_g - g_64).max() / scale, np.abs(g_c - g_64).max() / scale
            flip_g, flip_c = (np.sign(g_g) != np.sign(g_64)).mean(), (np.sign(g_c) != np.sign(g_64)).mean()
            # budget: twice the fp32 oracle's own distance from fp64 - or, where the oracle happens to land closer than fp32
            # convolution stacks usually do (resnet18_EE repeat 0: oracle 0.7 %, GPU 2 %, while on resnet50 the ORACLE sits at
            # 3 - 6 %: ReLU masks of pre-activations within fp32 rounding of zero flip between any two implementations), 5 % of the
            # largest entry and 0.2 % of the signs
            assert err_g <= max(2 * err_c, 5e-2) + 1e-6, (what, err_g, err_c)
            assert flip_g <= 2 * flip_c + 2e-3, (what, flip_g, flip_c)
            agree = np.sign(g_g) == np.sign(g_c)
            assert np.array_equal(got[agree], noise_c[:B].numpy()[agree]), what
            assert agree.mean() > 0.98, (what, agree.mean())
            # parameter gradients of the same backward (they persist until the next zero_grad), same budget against fp64
            num_g = num_c = den = 0.0
            assert len(trained(cpu)) == len(trained(gpu))
            for (n_, pc), pg, p64 in zip(cpu.named_parameters(), trained(gpu), cpu64.parameters()):
                r = p64.grad
                eg, ec = float((pg.grad.cpu().double() - r).norm()), float((pc.grad.double() - r).norm())
                assert eg <= 4 * ec + (1e-2 if arch.endswith("_EE") else 1e-6) * float(r.norm()) + 1e-12, (what, n_, eg, ec)
                num_g, num_c, den = num_g + eg ** 2, num_c + ec ** 2, den + float(r.norm()) ** 2
            assert num_g ** 0.5 <= 2 * num_c ** 0.5 + (5e-3 if arch.endswith("_EE") else 1e-6) * den ** 0.5, (what, num_g, num_c, den)
    last = "w2" if arch == "tinynet" else "fc.weight"
    assert not torch.equal(gpu.state_dict()[last].cpu(), _body_state(arch, state)[last])


@pytest.mark.parametrize("tag,net_cls,seed", [("plain", TinyNet, 3), ("bn", TinyBNNet, 4)])
def test_free_at_repeat_replays_the_reference_train(golden, tag, net_cls, seed):
    """a15 PINNED (round 3): eeadv.trainer.free_at_repeat on cuda:0 against the run of the reference's OWN train()
    (AT_free_imagenet_ddp.py:263-309, tests/golden/freeat.npz: 3 batches x 4 repeats, the last batch short, SGD with momentum and
    weight decay; `bn`: a train-mode BatchNorm).  The weights run freely on the GPU over all 12 repeats (SGD is smooth: they stay
    within 1e-5 of the recorded ones); the persistent buffer is re-loaded with the recorded one before every repeat, so that a
    gradient within rounding of zero - whose sign MIOpen and oneDNN may round differently, moving that pixel by 2 alpha - is
    counted where it happens instead of compounding.  Per repeat:

      * in1 = clamp(x + delta, 0, 1) (ee_add_clamp_f32): bit-exact;
      * logits within the north-star 1e-4, same predictions; dL/din1 within 1e-5 of its largest entry;
      * delta after the repeat: identical to the recorded buffer wherever the two masked gradients agree in sign (> 99.5 % of
        the entries), and ee_freeat_update_masked_f32 applied to the REFERENCE's gradient reproduces the recorded rows exactly;
      * rows beyond the batch: the reference clamps the whole buffer (:307), the kernel only the live rows - the script's buffer
        starts as zeros (:261) and is never loaded from anywhere, so |delta| <= clip_eps holds for every row at all times and the
        buffer-wide clamp never changes a value; the GPU side therefore starts from the fixture's buffer clamped once (its planted
        0.5 row is a state the reference cannot reach; the CPU test pins what the reference does with it);
      * every parameter after optimizer.step() within 1e-5 (absolute; they are O(0.1)), BatchNorm running mean within 1e-6."""
    from eeadv import ops, trainer
    G = golden("freeat")
    a, e = [float(v) for v in G[tag + "__step_eps"]]
    lr, mom, wd = [float(v) for v in G[tag + "__sgd"]]
    net = net_cls(2, 8, 10, seed).to(DEV).train()
    opt = torch.optim.SGD(net.parameters(), lr=lr, momentum=mom, weight_decay=wd)
    crit = trainer.Criterion()
    prev = np.clip(G[tag + "__noise0"], -np.float32(e), np.float32(e))
    i = 0
    for b in range(3):
        x, y = dev(G[tag + "__x%d" % b]), dev(G[tag + "__y%d" % b])
        n = x.shape[0]
        for rep in range(4):
            what = (tag, b, rep)
            noise = dev(prev)
            _, out, g_in1 = trainer.free_at_repeat(net, crit, opt, x, y, noise, a, e, return_input_grad=True)
            want = G[tag + "__deltas"][i]
            s = x.cpu().numpy() + prev[:n]
            inside = (s >= 0) & (s <= 1)
            assert np.array_equal(ops.add_clamp(x, dev(prev[:n]), 0.0, 1.0).cpu().numpy(), G[tag + "__in1_%d" % i]), what
            np.testing.assert_allclose(out.cpu().numpy(), G[tag + "__logits_%d" % i], atol=1e-4, rtol=0, err_msg=str(what))
            assert np.array_equal(out.argmax(1).cpu().numpy(), G[tag + "__logits_%d" % i].argmax(1)), what
            g_ref = G[tag + "__gin1_%d" % i]
            g_gpu = g_in1.cpu().numpy()
            assert np.abs(g_gpu - g_ref).max() <= 1e-5 * np.abs(g_ref).max(), what
            agree = np.sign(g_gpu * inside) == np.sign(g_ref * inside)
            got = noise.cpu().numpy()
            assert agree.mean() > 0.995, (what, agree.mean())
            assert np.array_equal(got[:n][agree], want[:n][agree]), what
            assert np.array_equal(got[n:], prev[n:]) and np.array_equal(np.clip(prev[n:], -np.float32(e), np.float32(e)), want[n:]), what
            # the update kernel on the reference's own gradient: the recorded rows, bit for bit
            exact = dev(prev)
            ops.freeat_update_masked_(exact, dev(g_ref), x, a, e)
            assert np.array_equal(exact.cpu().numpy(), want), what
            w = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy()
            assert np.abs(w - G[tag + "__weights"][i]).max() <= 1e-5, (what, np.abs(w - G[tag + "__weights"][i]).max())
            if tag == "bn":
                assert np.abs(net.bn.running_mean.cpu().numpy() - G[tag + "__bn_running_mean"][i]).max() <= 1e-6, what
            prev = want.copy()
            i += 1
    assert i == 12
