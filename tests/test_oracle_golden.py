"""CPU: the oracle (oracle/ee_oracle.c + oracle/ref_path.py) against the fixtures that
tests/golden/make_golden.py generated from the reference itself.  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import ee_oracle as O
from oracle import ref_path as R

from tiny_models import Args, TinyNet


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
