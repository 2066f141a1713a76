"""CPU: the oracle (oracle/ee_oracle.c + oracle/ref_path.py) against the fixtures that
tests/golden/make_golden.py generated from the reference itself.  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import ee_oracle as O
from oracle import ref_path as R

from tiny_models import Args, TinyBNNet, TinyNet


def test_fixed_weights_match_reference(golden):
    G = golden("kernels")
    np.testing.assert_array_equal(O.gaussian_kernel(3, 0, 1), G["gauss_k3_mu0_s1"])
    np.testing.assert_array_equal(O.gaussian_kernel(3, 0, 2.0), G["gauss_k3_mu0_s2"])
    np.testing.assert_array_equal(O.gaussian_kernel(5, 0, 1), G["gauss_k5_mu0_s1"])
    np.testing.assert_array_equal(O.gaussian_kernel(3, 0, 1, normalize=False), G["gauss_k3_unnorm"])
    np.testing.assert_array_equal(O.sobel_kernel(3), G["sobel_k3"])
    np.testing.assert_array_equal(O.sobel_kernel(5), G["sobel_k5"])
    g9, sx9, sy9 = O.edge_weights(1.0)
    assert abs(g9[0] - 0.07511361) < 1e-8 and abs(g9[4] - 0.20417996) < 1e-8
    assert sx9.tolist() == [-.5, 0, .5, -1, 0, 1, -.5, 0, .5]
    for (w, h, r) in [(64, 64, 8), (28, 28, 4), (224, 224, 16), (7, 9, 2), (32, 32, 4)]:
        np.testing.assert_array_equal(O.hfs_mask(w, h, r).astype(np.uint8), G["hfs_mask_%d_%d_%d" % (w, h, r)])
    assert O.hfs_mask(64, 64, 8).sum() == 256 and O.hfs_mask(28, 28, 4).sum() == 64


EDGE_CASES = ["rand_tiny", "rand_mnist", "rect_mnist", "rect_rgb", "ramp_thr", "ragged", "two_ch", "one_px", "thin", "big_mag"]


@pytest.mark.parametrize("name", EDGE_CASES)
def test_edge125_oracle_vs_reference(golden, name):
    """Edge bits and the NaN set of the input gradient are EXACT; finite gradients agree to 1e-6."""
    G = golden("edge125")
    x, (alpha, high), u = G[name + "__x"], G[name + "__alpha_high"], G[name + "__u"]
    e = O.edge125_fwd(x, alpha, high)
    assert np.array_equal(e.astype(np.uint8), G[name + "__edge"])
    g = O.edge125_bwd(x, u, alpha, high)
    ref = G[name + "__gx"]
    for c in range(ref.shape[1]):  # identical map in every input channel (SURVEY 8a')
        assert np.array_equal(np.isnan(ref[:, c:c + 1]), np.isnan(g))
        fin = ~np.isnan(g)
        np.testing.assert_allclose(g[fin], ref[:, c:c + 1][fin], rtol=0, atol=1e-6)
    # the torch restatement agrees too (bits may differ from the C oracle only through torch's conv order)
    xt = torch.from_numpy(x.copy()).requires_grad_(True)
    et = R.CannyFilter_step125_1(alpha=float(alpha))(xt, high_threshold=float(high))
    (et * torch.from_numpy(u)).sum().backward()
    assert np.array_equal(et.detach().numpy().astype(np.uint8), G[name + "__edge"])
    assert np.array_equal(np.isnan(xt.grad.numpy()), np.isnan(ref))


def test_survey_known_answers():
    torch.manual_seed(0)
    x = torch.rand(2, 3, 64, 64).numpy()
    assert int(O.edge125_fwd(x, 0.0, 76 / 255).sum()) == 568
    torch.manual_seed(0)
    x = torch.rand(2, 1, 28, 28).numpy()
    assert int(O.edge125_fwd(x, 0.3, 51 / 255).sum()) == 736


def test_nan_footprint_is_5x5_dilation_of_zero_magnitude(golden):
    G = golden("edge125")
    x, (alpha, high), u = G["rect_mnist__x"], G["rect_mnist__alpha_high"], G["rect_mnist__u"]
    _, mag, _, _ = O.edge125_fwd(x, alpha, high, want_internals=True)
    g = O.edge125_bwd(x, u, alpha, high)
    dil = torch.nn.functional.max_pool2d(torch.from_numpy((mag == 0).astype(np.float32)), 5, 1, 2).numpy() > 0
    assert np.array_equal(np.isnan(g), dil)


def test_pgd_trajectories(golden):
    G = golden("pgd_steps")
    x0 = G["x0"]
    eps, alpha = [float(v) for v in G["eps_alpha"]]
    assert np.array_equal(O.pgd_init(x0, G["pgd_noise"]), G["pgd_xs"][0])
    for tag, e, a, d in (("pgd", eps, alpha, 1), ("pgdb", 0.3, 0.01, 1), ("tpgd", eps, alpha, -1)):
        xs, gs, fin = G[tag + "_xs"], G[tag + "_gs"], G[tag + "_final"]
        for k in range(len(gs)):
            want = xs[k + 1] if k + 1 < len(xs) else fin
            assert np.array_equal(O.pgd_step(xs[k], gs[k], x0, a, e, direction=d), want), (tag, k)
    assert np.array_equal(O.fgsm_step(x0, G["fgsm_u_g"], 0.007), G["fgsm_u_final"])
    assert np.array_equal(O.fgsm_step(x0, G["fgsm_t_g"], 0.007, direction=-1), G["fgsm_t_final"])
    got = O.pgd_step(x0, G["special_g"], x0, alpha, eps)
    assert np.array_equal(got, G["special_final"])
    assert np.array_equal(got.reshape(-1)[:3], x0.reshape(-1)[:3])  # sign(NaN) = sign(+-0) = 0: no update


def test_ref_path_attacks_reproduce_reference(golden):
    G = golden("pgd_steps")
    x0, y = torch.from_numpy(G["x0"]), torch.from_numpy(G["y"])
    eps, alpha = [float(v) for v in G["eps_alpha"]]
    xa = R.PGD(TinyNet(2, 8, 10, 31), Args(random=True, epsilon=eps), x0, y, 6, alpha, noise=torch.from_numpy(G["pgd_noise"]))
    assert np.array_equal(xa.numpy(), G["pgd_final"])
    xb = R.PGD(TinyNet(2, 8, 10, 32), Args(random=False, epsilon=0.3), x0, y, 8, 0.01)
    assert np.array_equal(xb.numpy(), G["pgdb_final"])
    xc, tl = R.targeted_PGD(TinyNet(2, 8, 10, 33), Args(random=True, epsilon=eps), x0, y, 6, alpha, 10, "cpu",
                            noise=torch.from_numpy(G["tpgd_noise"]), label_offset=torch.from_numpy(G["tpgd_offset"]))
    assert np.array_equal(tl.numpy(), G["tpgd_target"]) and np.array_equal(xc.numpy(), G["tpgd_final"])
    for targeted, tag in ((False, "fgsm_u"), (True, "fgsm_t")):
        xf = R.FGSM(TinyNet(2, 8, 10, 34), x0, y, targeted=targeted, step_size=0.007)
        assert np.array_equal(xf.numpy(), G[tag + "_final"])


def test_losses_vs_reference(golden):
    G = golden("losses")
    for tag in "sti":
        la, lb, y = G[tag + "_la"], G[tag + "_lb"], G[tag + "_y"]
        B, K = la.shape
        v, d = O.ce(la, y, mean=False)
        np.testing.assert_allclose(v, G[tag + "_ce_sum"], rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(d, G[tag + "_ce_sum_g"], atol=1e-6)
        v, d = O.ce(la, y, mean=True)
        np.testing.assert_allclose(v, G[tag + "_ce_mean"], atol=1e-4)
        np.testing.assert_allclose(d, G[tag + "_ce_mean_g"], atol=1e-6)
        v, dq, dp = O.kl_batchmean(lb, la)
        np.testing.assert_allclose(v, G[tag + "_kl"], atol=1e-4)
        np.testing.assert_allclose(dq, G[tag + "_kl_gq"], atol=1e-6)
        np.testing.assert_allclose(dp, G[tag + "_kl_gp"], atol=1e-6)
        v, da = O.mse(la, lb)
        np.testing.assert_allclose(v, G[tag + "_mse"], rtol=1e-6, atol=1e-4)
        # composite losses (attacks.py:264-272, :421-429)
        ce_a, ce_b = O.ce(la, y, mean=True)[0], O.ce(lb, y, mean=True)[0]
        np.testing.assert_allclose(0.5 * ce_a + 0.5 * ce_b + 0.5 * v, G[tag + "_alp"], atol=1e-4)
        np.testing.assert_allclose(ce_a + 6.0 * O.kl_batchmean(lb, la)[0], G[tag + "_trades"], atol=1e-4, rtol=1e-6)
        soft = O.label_smoothing(np.eye(K, dtype=np.float32)[y], 0.1, K)
        np.testing.assert_array_equal(soft, G[tag + "_smooth_l2"])
        np.testing.assert_array_equal(O.label_smoothing(np.eye(K, dtype=np.float32)[y], 1.0, K), G[tag + "_smooth_l1"])
        v, dz = O.softce(la, soft.astype(np.float64), 1.0 / B)
        np.testing.assert_allclose(v, G[tag + "_softce_f64"], atol=1e-4)
        np.testing.assert_allclose(dz, G[tag + "_softce_g"], atol=1e-6)
        idx, _ = O.topk(la, y, 1)
        assert np.array_equal(idx[:, 0], G[tag + "_pred"])
        lt, yt = torch.from_numpy(la), torch.from_numpy(y)
        np.testing.assert_allclose(R.LabelSmoothLoss(0.1)(lt, yt).item(), G[tag + "_lsmooth"], atol=1e-6)
        np.testing.assert_allclose(R.l2_norm(lt.view(B, 1, 1, K)).numpy(), G[tag + "_l2norm"], rtol=1e-6)


def test_topk_matches_torch():
    rng = np.random.RandomState(0)
    z = rng.randn(64, 200).astype(np.float32)
    y = rng.randint(0, 200, 64)
    idx, correct = O.topk(z, y, 5)
    assert np.array_equal(idx, torch.from_numpy(z).topk(5, 1)[1].numpy())
    acc = R.accuracy(torch.from_numpy(z), torch.from_numpy(y), topk=(1, 5))
    assert correct[0] * 100.0 / 64 == acc[0].item() and correct[4] * 100.0 / 64 == acc[1].item()


def test_avmixup_and_cw(golden):
    G = golden("avmix_cw")
    x0, y = torch.from_numpy(G["x0"]), torch.from_numpy(G["y"])
    eps, alpha = [float(v) for v in G["eps_alpha"]]
    av = R.AVmixup(Args(random=True, epsilon=eps), 2.0, 1.0, 0.1, alpha, 5, num_classes=10, device="cpu")
    xm, ym = av.perturb(TinyNet(2, 8, 10, 51), x0, torch.eye(10)[y], noise=torch.from_numpy(G["av_noise"]), beta=G["av_beta"])
    assert ym.dtype == torch.float64
    assert np.array_equal(xm.numpy(), G["av_x"]) and np.array_equal(ym.numpy(), G["av_y"])
    # kernel-level oracle reproduces the mix from the last iterate
    x_last = O.pgd_step(G["av_xs"][-1], G["av_gs"][-1], G["x0"], alpha, eps)
    assert np.array_equal(O.avmix(x_last, G["x0"], G["av_beta"], 2.0), G["av_x"])
    net = TinyNet(2, 8, 10, 52)
    adv, p = R.CWLinfAttack(x0, net(x0).argmax(1), net, eps, None, eps, max_iters=4, target=torch.from_numpy(G["cw_target"]),
                            n_class=10, noise=torch.from_numpy(G["cw_noise"]))
    assert np.array_equal(adv.detach().numpy(), G["cw_adv"]) and np.array_equal(p.detach().numpy(), G["cw_p"])


def _checksum(model):
    return sum(float(p.double().abs().sum()) for p in model.state_dict().values())


def test_end_to_end_reference_models(golden):
    """Net_2 PGD-40 and resnet18 PGD-3 (eval mode): same seeds -> same weights -> bit-identical x_adv."""
    G = golden("e2e")
    torch.manual_seed(7)
    net = R.Net_2().eval()
    assert _checksum(net) == float(G["net2_checksum"])
    x, y = torch.from_numpy(G["net2_x"]), torch.from_numpy(G["net2_y"])
    xa = R.PGD(net, Args(random=True, epsilon=0.3), x, y, 40, 0.01, noise=torch.from_numpy(G["net2_noise"]))
    assert np.array_equal(xa.numpy(), G["net2_xadv"])
    with torch.no_grad():
        np.testing.assert_array_equal(net(xa).numpy(), G["net2_logits_adv"])
        assert np.array_equal(net(x).argmax(1).numpy(), G["net2_logits_clean"].argmax(1))
    torch.manual_seed(8)
    rn = R.resnet18().eval()
    assert _checksum(rn) == float(G["rn18_checksum"])
    x, y = torch.from_numpy(G["rn18_x"]), torch.from_numpy(G["rn18_y"])
    eps, alpha = 0.062745098039216, 0.007843137254902
    xa = R.PGD(rn, Args(random=True, epsilon=eps), x, y, 3, alpha, noise=torch.from_numpy(G["rn18_noise"]))
    assert np.array_equal(xa.numpy(), G["rn18_xadv"])
    with torch.no_grad():
        np.testing.assert_allclose(rn(xa).numpy(), G["rn18_logits_adv"], atol=1e-4)


@pytest.mark.parametrize("name", ["rand_rgb", "rand_mnist", "rect_rgb"])
@pytest.mark.parametrize("cls", ["CannyFilter", "CannyFilter_BPDA"])
def test_full_canny_restatement_UNPINNED(golden, name, cls):
    """PARITY UNPINNED: both sides use the DERIVED thin-kernel table (cv2 is absent); this only shows
    that the restatement reproduces the reference's forward/backward code given that table."""
    G = golden("canny_full_unpinned")
    x = torch.from_numpy(G[name + "__x"].copy()).requires_grad_(True)
    alpha, low, high = [float(v) for v in G[name + "__alpha_low_high"]]
    e = getattr(R, cls)(alpha=alpha)(x, low_threshold=low, high_threshold=high, hysteresis=True)
    (e * torch.from_numpy(G[name + "__u"])).sum().backward()
    assert np.array_equal(e.detach().numpy(), G[name + "__" + cls + "__edge"])
    ref = G[name + "__" + cls + "__gx"]
    assert np.array_equal(np.isnan(x.grad.numpy()), np.isnan(ref))
    fin = ~np.isnan(ref)
    np.testing.assert_allclose(x.grad.numpy()[fin], ref[fin], atol=1e-6)


@pytest.mark.parametrize("name", ["rand_rgb", "rand_mnist", "rect_rgb"])
@pytest.mark.parametrize("cls", ["CannyFilter", "CannyFilter_BPDA"])
def test_full_canny_c_oracle_UNPINNED(golden, name, cls):
    """The C restatement (the GPU tests' checker) against the same fixtures: edge maps exactly, input gradients with the
    NaN pattern exact and the finite values to 1e-6 (torch's CPU convolutions order their sums differently from N = 1 up)."""
    G = golden("canny_full_unpinned")
    x, u = G[name + "__x"], G[name + "__u"]
    alpha, low, high = [float(v) for v in G[name + "__alpha_low_high"]]
    if cls == "CannyFilter":
        e, g = O.canny_fwd(x, alpha, low, high), O.canny_bwd(x, u, alpha, low, high)
    else:
        e, g = O.canny_bpda_fwd(x, low, high), O.canny_bpda_bwd(x, u, low, high)
    assert np.array_equal(e, G[name + "__" + cls + "__edge"])
    ref = G[name + "__" + cls + "__gx"]
    g = np.broadcast_to(g, ref.shape)
    assert np.array_equal(np.isnan(g), np.isnan(ref))
    fin = ~np.isnan(ref)
    np.testing.assert_allclose(g[fin], ref[fin], atol=1e-6)


def test_hfs_restatement_properties_UNPINNED():
    """PARITY UNPINNED (torch.rfft no longer exists): linear, idempotent on its pass band, keeps DC."""
    torch.manual_seed(0)
    for (n, r) in [(64, 8), (28, 4)]:
        hfs = R.HighFreqSuppress(n, n, r)
        x = torch.rand(2, 3, n, n)
        y = hfs(x)
        assert y.dtype == torch.float32 and y.shape == x.shape
        np.testing.assert_allclose(y.mean((-1, -2)).numpy(), x.mean((-1, -2)).numpy(), atol=1e-6)
        np.testing.assert_allclose(hfs(2 * x + 1).numpy(), (2 * y + 1).numpy(), atol=1e-5)
        lowpass = torch.cos(2 * np.pi * 3 * torch.arange(n) / n)[None, :] * torch.ones(n, 1)
        np.testing.assert_allclose(hfs(lowpass[None, None]).numpy()[0, 0], lowpass.numpy(), atol=1e-5)
        hi = torch.cos(2 * np.pi * (r + 2) * torch.arange(n) / n)[None, :] * torch.ones(n, 1)
        assert hfs(hi[None, None]).abs().max() < 1e-5


def test_targeted_family_reproduces_reference(golden):
    """targeted_PGD_trick (both Bernoulli outcomes and args.random False), tar_alp_imagenet, AVmixup.tar_perturb:
    the restatements with the reference's recorded draws give the reference's outputs BIT FOR BIT, and the kernel-level
    oracle replays every recorded step (attacks.py:59-86, 337-357, 481-518)."""
    G = golden("targeted")
    x0, y = torch.from_numpy(G["x0"]), torch.from_numpy(G["y"])
    eps, alpha = [float(v) for v in G["eps_alpha"]]
    for tag, b in (("trick_noise", True), ("trick_clean", False)):
        assert bool(G[tag + "_u"] > 0.5) == b
        xa, tl = R.targeted_PGD_trick(TinyNet(2, 8, 10, 71), Args(random=True, epsilon=eps, prob_start_from_clean=0.5), x0, y, 5, alpha,
                                      10, "cpu", noise=torch.from_numpy(G[tag + "_init"]), label_offset=torch.from_numpy(G[tag + "_offset"]),
                                      start_from_noise=b)
        assert np.array_equal(tl.numpy(), G[tag + "_target"]) and np.array_equal(xa.numpy(), G[tag + "_final"])
        start = O.pgd_init(G["x0"], G[tag + "_init"] if b else np.zeros_like(G["x0"]))
        assert np.array_equal(start, G[tag + "_xs"][0])
        xs, gs = G[tag + "_xs"], G[tag + "_gs"]
        for k in range(len(gs)):
            want = xs[k + 1] if k + 1 < len(xs) else G[tag + "_final"]
            assert np.array_equal(O.pgd_step(xs[k], gs[k], G["x0"], alpha, eps, direction=-1), want), (tag, k)
    xa, tl = R.targeted_PGD_trick(TinyNet(2, 8, 10, 71), Args(random=False, epsilon=eps, prob_start_from_clean=0.5), x0, y, 5, alpha, 10,
                                  "cpu", label_offset=torch.from_numpy(G["trick_norand_offset"]))
    assert np.array_equal(tl.numpy(), G["trick_norand_target"]) and np.array_equal(xa.numpy(), G["trick_norand_final"])

    y2 = torch.from_numpy(G["talp_y"])
    xa, tl = R.tar_alp_imagenet(TinyNet(2, 8, 1000, 72), Args(epsilon=eps), x0, y2, 5, alpha, "cpu",
                                noise=torch.from_numpy(G["talp_randn"]), label_offset=torch.from_numpy(G["talp_offset"]))
    assert np.array_equal(tl.numpy(), G["talp_target"]) and np.array_equal(xa.numpy(), G["talp_final"])
    assert int(tl.max()) < 1000 and bool((tl != y2).all())
    # the start is x + 0.001 * randn WITHOUT a clamp: the first iterate leaves [0, 1] where x0 sits on the border
    start = O.pgd_init(G["x0"], (np.float32(0.001) * G["talp_randn"]).astype(np.float32), -np.inf, np.inf)
    assert np.array_equal(start, G["talp_xs"][0]) and (start.min() < 0 or start.max() > 1)

    av = R.AVmixup(Args(random=True, epsilon=eps), 2.0, 1.0, 0.1, alpha, 5, num_classes=10, device="cpu")
    xm, ym = av.tar_perturb(TinyNet(2, 8, 10, 73), x0, torch.eye(10)[y], noise=torch.from_numpy(G["tav_noise"]), beta=G["tav_beta"],
                            label_offset=torch.from_numpy(G["tav_offset"]))
    assert ym.dtype == torch.float64 and xm.dtype == torch.float32
    assert np.array_equal(xm.numpy(), G["tav_x"]) and np.array_equal(ym.numpy(), G["tav_y"])
    x_last = O.pgd_step(G["tav_xs"][-1], G["tav_gs"][-1], G["x0"], alpha, eps, direction=-1)
    assert np.array_equal(O.avmix(x_last, G["x0"], G["tav_beta"], 2.0), G["tav_x"])


def test_linf_loops_reproduce_reference(golden):
    """Trades.PGD_Linf / PGD_L2, ALP.PGD_Linf, targeted_ALP.{tarPGD_Linf, PGD_Linf} against the reference's own methods
    (their single `torch.randn(.., device='cuda')` drew on the host when the fixture was made): bit for bit."""
    G = golden("linf_loops")
    x0, y = torch.from_numpy(G["x0"]), torch.from_numpy(G["y"])
    eps, alpha = [float(v) for v in G["eps_alpha"]]
    net = TinyNet(2, 8, 10, 91)
    logits = net(x0)
    np.testing.assert_array_equal(logits.detach().numpy(), G["trades_logits"])
    m = TinyNet(2, 8, 10, 91).train()
    xa = R.Trades(alpha, eps, 5, 6.0).PGD_Linf(m, x0, logits, noise=torch.from_numpy(G["trades_randn"]))
    assert not m.training and np.array_equal(xa.numpy(), G["trades_final"])
    start = O.pgd_init(G["x0"], (np.float32(0.001) * G["trades_randn"]).astype(np.float32), -np.inf, np.inf)
    assert np.array_equal(start, G["trades_xs"][0]) and start.min() < 0  # unclamped start
    step_l2, eps_l2 = [float(v) for v in G["tradesl2_step_eps"]]
    xa = R.Trades(step_l2, eps_l2, 5, 6.0).PGD_L2(TinyNet(2, 8, 10, 91), x0, logits, noise=torch.from_numpy(G["tradesl2_randn"]))
    assert np.array_equal(xa.numpy(), G["tradesl2_final"])
    xa = R.ALP(alpha, eps, 5, 1.0).PGD_Linf(TinyNet(2, 8, 10, 92), x0, y, noise=torch.from_numpy(G["alp_randn"]))
    assert np.array_equal(xa.numpy(), G["alp_final"])
    tal = R.targeted_ALP(alpha, eps, 5, 1.0, n_class=10)
    xa = tal.tarPGD_Linf(TinyNet(2, 8, 10, 93), x0, y, "cpu", noise=torch.from_numpy(G["talpc_randn"]),
                         label_offset=torch.from_numpy(G["talpc_offset"]))
    assert np.array_equal(xa.numpy(), G["talpc_final"])
    xa = tal.PGD_Linf(TinyNet(2, 8, 10, 93), x0, y, noise=torch.from_numpy(G["talpu_randn"]))
    assert np.array_equal(xa.numpy(), G["talpu_final"])
    for tag, d in (("trades", 1), ("alp", 1), ("talpc", -1)):  # every recorded step through the kernel-level oracle
        xs, gs = G[tag + "_xs"], G[tag + "_gs"]
        for k in range(len(gs)):
            want = xs[k + 1] if k + 1 < len(xs) else G[tag + "_final"]
            assert np.array_equal(O.pgd_step(xs[k], gs[k], G["x0"], alpha, eps, direction=d), want), (tag, k)


ADD_SQUARE_CASES = ["tiny", "mnist", "nq12", "nq60_resc"]


@pytest.mark.parametrize("tag", ADD_SQUARE_CASES)
def test_add_square_reproduces_reference(golden, tag):
    """Add_Square (core.py:589-655): the reference's forward run on the host with its draws recorded; the restatement with
    those draws reproduces output and input gradient bit for bit, and derives the same square sizes (p_selection)."""
    G = golden("add_square")
    B, C, n, nq, resc = [int(v) for v in G[tag + "__cfg"]]
    eps = float(G[tag + "__eps"])
    mod = R.Add_Square(C, n, eps, n_queries=nq, rescale_schedule=bool(resc))
    assert mod.sizes() == G[tag + "__sq_size"].tolist()
    d = {"stripe": torch.from_numpy(G[tag + "__stripe"]), "sq_pos": torch.from_numpy(G[tag + "__sq_pos"]),
         "sq_sign": torch.from_numpy(G[tag + "__sq_sign"]).reshape(nq, C, 1, 1)}
    x = torch.from_numpy(G[tag + "__x"].copy()).requires_grad_(True)
    yv = mod(x, d)
    (yv * torch.from_numpy(G[tag + "__u"])).sum().backward()
    assert np.array_equal(yv.detach().numpy(), G[tag + "__y"])
    assert np.array_equal(x.grad.numpy(), G[tag + "__gx"])
    assert set(np.unique(G[tag + "__stripe"]).tolist()) <= {-1.0, 0.0, 1.0}
    assert (G[tag + "__sq_pos"] >= 0).all() and (G[tag + "__sq_pos"] + G[tag + "__sq_size"] <= n).all()


FREE_AT_CASES = [("plain", TinyNet, 3), ("bn", TinyBNNet, 4)]


@pytest.mark.parametrize("tag,net_cls,seed", FREE_AT_CASES)
def test_free_at_repeat_reproduces_the_reference_train(golden, tag, net_cls, seed):
    """a15, PINNED (round 3): tests/golden/freeat.npz was written by the reference's OWN train() (ImageNet/free_imagenet/
    AT_free_imagenet_ddp.py:263-309, the FunctionDef compiled from the parsed script - make_golden.py section 11) over three
    batches x 4 repeats (the last batch is short), SGD with momentum and weight decay, a train-mode BatchNorm in the second
    case.  The restatement R.free_at_repeat, driven the same way, reproduces BIT FOR BIT after every repeat: in1, the logits,
    dL/din1, the whole persistent buffer (incl. the row no batch reaches, which only the buffer-wide clamp_ of :307 touches),
    every parameter after optimizer.step() and the BatchNorm running mean.  One thread, as the fixture was made: oneDNN's
    weight-gradient reduction splits over threads (1 ulp on 14 of 1424 parameters with 8 threads)."""
    G = golden("freeat")
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        _free_at_replay(G, tag, net_cls, seed)
    finally:
        torch.set_num_threads(threads)


def _free_at_replay(G, tag, net_cls, seed):
    a, e = [float(v) for v in G[tag + "__step_eps"]]
    lr, mom, wd = [float(v) for v in G[tag + "__sgd"]]
    net = net_cls(2, 8, 10, seed).train()
    opt = torch.optim.SGD(net.parameters(), lr=lr, momentum=mom, weight_decay=wd)
    noise = torch.from_numpy(G[tag + "__noise0"].copy())
    i = 0
    for b in range(3):
        x, y = torch.from_numpy(G[tag + "__x%d" % b]), torch.from_numpy(G[tag + "__y%d" % b])
        for rep in range(4):
            before = noise.clone()
            _, out, g = R.free_at_repeat(net, F_ce, opt, x, y, noise, a, e, want_grad=True)
            what = (tag, b, rep)
            n = x.shape[0]
            assert np.array_equal((x + before[:n]).clamp(0, 1).numpy(), G[tag + "__in1_%d" % i]), what
            assert np.array_equal(out.numpy(), G[tag + "__logits_%d" % i]), what
            # the recorded gradient is dL/din1 BEFORE the mask of the in-place clamp; the restatement returns noise_batch.grad
            # (after it): equal where x + delta stayed inside [0, 1], zero elsewhere
            s = x + before[:n]
            inside = ((s >= 0) & (s <= 1)).numpy()
            assert np.array_equal(g.numpy(), np.where(inside, G[tag + "__gin1_%d" % i], np.float32(0))), what
            assert not inside.all() or b == 2 or i == 0  # from the second repeat on x + delta leaves [0, 1] at the two planted pixels
            assert np.array_equal(noise.numpy(), G[tag + "__deltas"][i]), what
            assert np.array_equal(torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy(), G[tag + "__weights"][i]), what
            if tag == "bn":
                assert np.array_equal(net.bn.running_mean.numpy(), G[tag + "__bn_running_mean"][i]), what
            # the kernel-level oracle on the recorded gradient gives the recorded rows
            assert np.array_equal(O.freeat_update(before[:n].numpy(), g.numpy(), a, e), G[tag + "__deltas"][i][:n]), what
            i += 1
    assert i == 12 and np.abs(G[tag + "__deltas"][-1][3]).max() == np.float32(e)  # the unreachable row: 0.5 -> clip_eps


def F_ce(z, y):
    return torch.nn.functional.cross_entropy(z, y)
