"""GPU parity: every HIP kernel, called through the C ABI, against the CPU oracle on the same inputs.

Bar (task spec / BASELINE north_star): bit-exact for binary / integer results and for the element-wise
fp32 updates and stencils (the oracle fixes the operation order); 1e-6 relative (well inside the 1e-4 the
north star allows) for kernels that call exp/log, whose libm differs between host and device.
"""
import numpy as np
import pytest
import torch

from oracle import ee_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bitexact(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if got.dtype == np.float32:
        nan_g, nan_w = np.isnan(got), np.isnan(want)
        assert np.array_equal(nan_g, nan_w), "%s: NaN pattern differs (%d vs %d)" % (what, nan_g.sum(), nan_w.sum())
        same = (bits(got) == bits(want)) | nan_g
        assert same.all(), "%s: %d of %d elements differ, max |d| = %g" % (
            what, (~same).sum(), same.size, np.nanmax(np.abs(got - want)))
    else:
        assert np.array_equal(got, want), what


@pytest.fixture(scope="module")
def ops():
    from eeadv import ops as _ops
    return _ops


EDGE_CASES = ["rand_tiny", "rand_mnist", "rect_mnist", "rect_rgb", "ramp_thr", "ragged", "two_ch", "one_px", "thin", "big_mag"]


@pytest.mark.parametrize("name", EDGE_CASES)
def test_edge125_fwd_bwd_golden_inputs(ops, golden, name):
    G = golden("edge125")
    x, (alpha, high), u = G[name + "__x"], G[name + "__alpha_high"], G[name + "__u"]
    wts = ops.EdgeWeights(1.0)
    e, mag = ops.edge125_fwd(dev(x), wts, float(alpha), float(high), want_mag=True)
    oe, omag, _, _ = O.edge125_fwd(x, alpha, high, want_internals=True)
    assert_bitexact(e.cpu().numpy(), oe, "edge")
    assert_bitexact(mag.cpu().numpy(), omag, "mag")
    # ... and directly against the reference's own output
    assert np.array_equal(e.cpu().numpy().astype(np.uint8), G[name + "__edge"])
    g = ops.edge125_bwd(dev(x), dev(u), wts, float(alpha), float(high))
    assert_bitexact(g.cpu().numpy(), O.edge125_bwd(x, u, alpha, high), "edge bwd")
    ref = G[name + "__gx"][:, :1]
    fin = ~np.isnan(ref)
    assert np.array_equal(np.isnan(g.cpu().numpy()), ~fin)
    np.testing.assert_allclose(g.cpu().numpy()[fin], ref[fin], rtol=0, atol=1e-6)


@pytest.mark.parametrize("shape", [(3, 3, 64, 64), (2, 1, 28, 28), (2, 3, 70, 50), (1, 2, 33, 130), (2, 4, 16, 16), (1, 3, 224, 224)])
def test_frontend_fwd_bwd_random(ops, shape):
    rng = np.random.RandomState(sum(shape))
    x = rng.rand(*shape).astype(np.float32)
    # low-pass branch output with negative / >1 values and exact 0 / tiny entries so the clamp gate flips
    xh = (rng.rand(*shape).astype(np.float32) * 1.4 - 0.3)
    xh.flat[::17] = 0.0
    xh.flat[5::23] = np.float32(1e-9)
    xh.flat[7::29] = np.float32(-1e-9)
    alpha, high, w = 0.0, 76.0 / 255, 1.0
    wts = ops.EdgeWeights(1.0)
    x_in, gate, edge = ops.frontend_fwd(dev(x), dev(xh), wts, alpha, high, w, want_edge=True)
    ox, ogate, oedge = O.frontend_fwd(x, xh, alpha, high, w)
    assert_bitexact(edge.cpu().numpy(), oedge, "edge")
    assert_bitexact(x_in.cpu().numpy(), ox, "x_in")
    assert np.array_equal(gate.cpu().numpy(), ogate)
    g_in = rng.randn(*shape).astype(np.float32)
    g_hfs, g_edge = ops.frontend_bwd(dev(g_in), gate, dev(x), wts, alpha, high, w)
    oh, oe = O.frontend_bwd(g_in, ogate, x, alpha, high, w)
    assert_bitexact(g_hfs.cpu().numpy(), oh, "g_hfs")
    assert_bitexact(g_edge.cpu().numpy(), oe, "g_edge")


def test_frontend_piecewise_constant_nan_grads(ops, golden):
    """mag == 0 regions give NaN input gradients, which the PGD step must turn into 'no update'."""
    x = golden("edge125")["rect_rgb__x"]
    rng = np.random.RandomState(5)
    xh = (x * np.float32(0.9)).astype(np.float32)
    wts = ops.EdgeWeights(1.0)
    x_in, gate, _ = ops.frontend_fwd(dev(x), dev(xh), wts, 0.0, 76.0 / 255, 1.0)
    g_in = rng.randn(*x.shape).astype(np.float32)
    g_hfs, g_edge = ops.frontend_bwd(dev(g_in), gate, dev(x), wts, 0.0, 76.0 / 255, 1.0)
    oh, oe = O.frontend_bwd(g_in, gate.cpu().numpy(), x, 0.0, 76.0 / 255, 1.0)
    assert np.isnan(oe).sum() > 100
    assert_bitexact(g_edge.cpu().numpy(), oe, "g_edge")
    x0 = x.copy()
    xd = dev(x.copy())
    ops.pgd_step_bcast_(xd, g_hfs, g_edge, dev(x0), 2 / 255, 16 / 255)
    g_full = oh + oe  # broadcast over channels
    want = O.pgd_step(x, g_full, x0, 2 / 255, 16 / 255)
    assert_bitexact(xd.cpu().numpy(), want, "pgd_step_bcast")
    assert np.array_equal(want[np.isnan(g_full)], x[np.isnan(g_full)])


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 1023, 4096 + 7, 100 * 3 * 64 * 64])
def test_pgd_elementwise(ops, n):
    rng = np.random.RandomState(n % 1000)
    x0 = rng.rand(n).astype(np.float32)
    noise = ((rng.rand(n) - 0.5) * 0.2).astype(np.float32)
    g = rng.randn(n).astype(np.float32)
    if n > 8:
        g[:6] = [np.nan, 0.0, -0.0, np.inf, -np.inf, 1e-40]
        x0[6:8] = [0.0, 1.0]
    eps, alpha = 16 / 255, 2 / 255
    x = ops.pgd_init(dev(x0), dev(noise))
    assert_bitexact(x.cpu().numpy(), O.pgd_init(x0, noise), "pgd_init")
    for direction in (1, -1):
        xc = x.clone()
        ops.pgd_step_(xc, dev(g), dev(x0), alpha, eps, direction=direction)
        assert_bitexact(xc.cpu().numpy(), O.pgd_step(x.cpu().numpy(), g, x0, alpha, eps, direction=direction), "pgd_step")
        f = ops.fgsm_step(x, dev(g), 0.007, direction=direction)
        assert_bitexact(f.cpu().numpy(), O.fgsm_step(x.cpu().numpy(), g, 0.007, direction=direction), "fgsm")
    ac = ops.add_clamp(dev(x0), dev(noise))
    assert_bitexact(ac.cpu().numpy(), O.add_clamp(x0, noise), "add_clamp")
    d = dev(noise.copy())
    ops.freeat_update_(d, dev(g), 4 / 255, 4 / 255)
    assert_bitexact(d.cpu().numpy(), O.freeat_update(noise, g, 4 / 255, 4 / 255), "freeat")


def test_pgd_step_unaligned_views(ops):
    rng = np.random.RandomState(3)
    n = 4099
    big = dev(rng.rand(n + 3).astype(np.float32))
    x0 = dev(rng.rand(n + 3).astype(np.float32))
    g = dev(rng.randn(n + 3).astype(np.float32))
    xv, x0v, gv = big[1:n + 1], x0[1:n + 1], g[1:n + 1]
    want = O.pgd_step(xv.cpu().numpy(), gv.cpu().numpy(), x0v.cpu().numpy(), 0.01, 0.3)
    guard = big[n + 1:].clone()
    ops.pgd_step_(xv, gv, x0v, 0.01, 0.3)
    assert_bitexact(xv.cpu().numpy(), want, "unaligned")
    assert torch.equal(big[n + 1:], guard)


def test_golden_pgd_trajectories(ops, golden):
    """Replay the reference's own PGD runs step by step: x_k, g_k -> x_{k+1} must match bit for bit."""
    G = golden("pgd_steps")
    x0 = G["x0"]
    eps, alpha = [float(v) for v in G["eps_alpha"]]
    x1 = ops.pgd_init(dev(x0), dev(G["pgd_noise"]))
    assert_bitexact(x1.cpu().numpy(), G["pgd_xs"][0], "random start")
    for tag, e, a, d in (("pgd", eps, alpha, 1), ("pgdb", 0.3, 0.01, 1), ("tpgd", eps, alpha, -1)):
        xs, gs, fin = G[tag + "_xs"], G[tag + "_gs"], G[tag + "_final"]
        for k in range(len(gs)):
            x = dev(xs[k].copy())
            ops.pgd_step_(x, dev(gs[k]), dev(x0), a, e, direction=d)
            want = xs[k + 1] if k + 1 < len(xs) else fin
            assert_bitexact(x.cpu().numpy(), want, "%s step %d" % (tag, k))
    for tag, d in (("fgsm_u", 1), ("fgsm_t", -1)):
        f = ops.fgsm_step(dev(x0), dev(G[tag + "_g"]), 0.007, direction=d)
        assert_bitexact(f.cpu().numpy(), G[tag + "_final"], tag)
    x = dev(x0.copy())
    ops.pgd_step_(x, dev(G["special_g"]), dev(x0), alpha, eps)
    assert_bitexact(x.cpu().numpy(), G["special_final"], "special values")


@pytest.mark.parametrize("B,K", [(4, 10), (100, 200), (37, 1000), (3, 1), (2, 70000 // 2)])
def test_losses(ops, B, K):
    rng = np.random.RandomState(B + K)
    za = (rng.randn(B, K) * 3).astype(np.float32)
    zb = (rng.randn(B, K) * 3).astype(np.float32)
    y = rng.randint(0, K, B).astype(np.int64)
    for mean in (False, True):
        loss, d = ops.ce(dev(za), dev(y), "mean" if mean else "sum")
        ol, od = O.ce(za, y, mean=mean)
        np.testing.assert_allclose(loss.item(), ol, rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(d.cpu().numpy(), od, rtol=1e-5, atol=1e-7)
    loss, dq, dp = ops.kl_batchmean(dev(zb), dev(za), want_dp=True)
    ol, odq, odp = O.kl_batchmean(zb, za)
    np.testing.assert_allclose(loss.item(), ol, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dq.cpu().numpy(), odq, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dp.cpu().numpy(), odp, rtol=1e-4, atol=2e-7)
    loss, da = ops.mse(dev(za), dev(zb))
    ol, oda = O.mse(za, zb)
    np.testing.assert_allclose(loss.item(), ol, rtol=2e-6)
    np.testing.assert_allclose(da.cpu().numpy(), oda, rtol=1e-6, atol=1e-9)
    if K > 1:
        t = rng.rand(B, K)
        t /= t.sum(1, keepdims=True)
        loss, dz = ops.softce(dev(za), dev(t), 1.0 / B)
        ol, odz = O.softce(za, t, 1.0 / B)
        np.testing.assert_allclose(loss.item(), ol, rtol=2e-6)
        np.testing.assert_allclose(dz.cpu().numpy(), odz, rtol=1e-5, atol=1e-7)
    k = min(5, K)
    idx, correct = ops.topk(dev(za), dev(y), k)
    oi, oc = O.topk(za, y, k)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(correct.cpu().numpy(), oc)


def test_topk_ties_and_golden_losses(ops, golden):
    z = np.zeros((3, 7), np.float32)
    z[1, [2, 5]] = 1.0
    idx, _ = ops.topk(dev(z), None, 3)
    assert idx.cpu().numpy().tolist() == [[0, 1, 2], [2, 5, 0], [0, 1, 2]]
    G = golden("losses")
    for tag in "sti":
        la, lb, y = G[tag + "_la"], G[tag + "_lb"], G[tag + "_y"]
        B = la.shape[0]
        loss, d = ops.ce(dev(la), dev(y), "sum")
        np.testing.assert_allclose(loss.item(), G[tag + "_ce_sum"], rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(d.cpu().numpy(), G[tag + "_ce_sum_g"], atol=1e-6)
        loss, d = ops.ce(dev(la), dev(y), "mean")
        np.testing.assert_allclose(loss.item(), G[tag + "_ce_mean"], atol=1e-4)
        np.testing.assert_allclose(d.cpu().numpy(), G[tag + "_ce_mean_g"], atol=1e-6)
        loss, dq, dp = ops.kl_batchmean(dev(lb), dev(la), want_dp=True)
        np.testing.assert_allclose(loss.item(), G[tag + "_kl"], atol=1e-4)
        np.testing.assert_allclose(dq.cpu().numpy(), G[tag + "_kl_gq"], atol=1e-6)
        np.testing.assert_allclose(dp.cpu().numpy(), G[tag + "_kl_gp"], atol=1e-6)
        loss, _ = ops.mse(dev(la), dev(lb))
        np.testing.assert_allclose(loss.item(), G[tag + "_mse"], atol=1e-4, rtol=1e-6)
        loss, d = ops.ce(dev(la), dev(y), "mean", smoothing=0.1)
        np.testing.assert_allclose(loss.item(), G[tag + "_lsmooth"], atol=1e-4)
        np.testing.assert_allclose(d.cpu().numpy(), G[tag + "_lsmooth_g"], atol=1e-6)
        idx, _ = ops.topk(dev(la), None, 1)
        assert np.array_equal(idx.cpu().numpy()[:, 0], G[tag + "_pred"])
        K = la.shape[1]
        soft = dev(G[tag + "_smooth_l2"].astype(np.float64))
        loss, dz = ops.softce(dev(la), soft, 1.0 / B)
        np.testing.assert_allclose(loss.item(), G[tag + "_softce_f64"], atol=1e-4)
        np.testing.assert_allclose(dz.cpu().numpy(), G[tag + "_softce_g"], atol=1e-6)
        w1 = dev(np.ones(B))
        lab = ops.avmix_labels(dev(y), w1, K, 1.0, 0.1)
        np.testing.assert_array_equal(lab.cpu().numpy(), G[tag + "_smooth_l1"].astype(np.float64))
        lab0 = ops.avmix_labels(dev(y), dev(np.zeros(B)), K, 1.0, 0.1)
        np.testing.assert_array_equal(lab0.cpu().numpy(), G[tag + "_smooth_l2"].astype(np.float64))


def test_avmix_golden(ops, golden):
    G = golden("avmix_cw")
    x0, y = G["x0"], G["y"]
    xs = G["av_xs"]
    eps, alpha = [float(v) for v in G["eps_alpha"]]
    # last PGD iterate: replay the final step, then vertex + mix
    x = dev(xs[-1].copy())
    ops.pgd_step_(x, dev(G["av_gs"][-1]), dev(x0), alpha, eps)
    w = G["av_beta"].reshape(-1)
    out = ops.avmix(x, dev(x0), dev(w), 2.0)
    assert_bitexact(out.cpu().numpy(), G["av_x"], "avmix x")
    assert_bitexact(out.cpu().numpy(), O.avmix(x.cpu().numpy(), x0, w, 2.0), "avmix vs oracle")
    lab = ops.avmix_labels(dev(y), dev(w), 10, 1.0, 0.1)
    np.testing.assert_array_equal(lab.cpu().numpy(), G["av_y"])


def test_rng_init_statistics(ops):
    n = 1 << 20
    x0 = torch.full((n,), 0.5, device=DEV)
    eps = 16 / 255
    a = ops.pgd_init_rng(x0, eps, 0, 1234, 0)
    b = ops.pgd_init_rng(x0, eps, 0, 1234, 0)
    c = ops.pgd_init_rng(x0, eps, 0, 1234, n // 4)
    assert torch.equal(a, b) and not torch.equal(a, c)
    d = (a - 0.5).cpu().numpy()
    assert d.min() >= -eps and d.max() < eps
    assert abs(d.mean()) < 3e-4 and abs(d.std() - 2 * eps / np.sqrt(12)) < 3e-4
    z = (ops.pgd_init_rng(x0, 0.001, 1, 99, 0, lo=-np.inf, hi=np.inf) - 0.5).cpu().numpy() / 0.001
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3


def test_cpu_tensor_is_refused(ops):
    from eeadv._native import EEError
    with pytest.raises(EEError):
        ops.pgd_step_(torch.zeros(4), torch.zeros(4), torch.zeros(4), 0.1, 0.1)


def test_bad_shapes_are_refused_before_launch(ops):
    from eeadv._native import EEError
    x = torch.zeros(1, 5, 8, 8, device=DEV)
    with pytest.raises(EEError):
        ops.edge125_fwd(x, ops.EdgeWeights(1.0), 0.0, 0.3)  # C > 4: EE_ERR_UNSUPPORTED
    with pytest.raises(ValueError):
        ops.pgd_step_(torch.zeros(4, device=DEV), torch.zeros(5, device=DEV), torch.zeros(4, device=DEV), 0.1, 0.1)


def test_freeat_masked_update_equals_reference_lines(ops):
    """AT_free_imagenet_ddp.py:287-307 on torch CPU ops vs the fused kernel (bit-exact)."""
    torch.manual_seed(2)
    B, full = 6, 10
    x = torch.rand(B, 3, 8, 8)
    x[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 0.001, 0.999])
    noise = (torch.rand(full, 3, 8, 8) - 0.5) * (8 / 255)
    w = torch.randn(3 * 8 * 8, 5)
    y = torch.randint(0, 5, (B,))
    step, eps = 4 / 255, 4 / 255
    # reference lines
    ref_noise = noise.clone()
    nb = ref_noise[0:B].clone().requires_grad_(True)
    in1 = x + nb
    in1.clamp_(0, 1.0)
    torch.nn.functional.cross_entropy(in1.flatten(1) @ w, y).backward()
    ref_noise[0:B] += step * torch.sign(nb.grad)
    ref_noise.clamp_(-eps, eps)
    # kernel path: gradient w.r.t. the clamped input, mask applied in the kernel
    xd, nd = dev(x.numpy()), dev(noise.numpy())
    in1d = ops.add_clamp(xd, nd[0:B].contiguous()).requires_grad_(True)
    assert np.array_equal(in1d.detach().cpu().numpy(), in1.detach().numpy())
    g_in1 = torch.autograd.grad(torch.nn.functional.cross_entropy(in1d.flatten(1) @ w.to(DEV), y.to(DEV)), in1d)[0]
    ops.freeat_update_masked_(nd, g_in1.contiguous(), xd, step, eps)
    got = nd.cpu().numpy()
    assert np.array_equal(got[B:], noise.numpy()[B:])  # rows beyond the batch are untouched
    same = (got[:B] == ref_noise.numpy()[:B]).mean()
    assert same > 0.995  # GEMM rounding may flip the sign of a ~0 gradient; everything else is bit-identical
    assert np.abs(got).max() <= eps + 1e-9


# ---- full CannyFilter (PARITY UNPINNED: derived thin-kernel table on both sides) ---------------------------------------
@pytest.mark.parametrize("shape,alpha,low,high", [((3, 3, 64, 64), 0.0, 38 / 255, 76 / 255), ((2, 1, 28, 28), 0.3, 25 / 255, 51 / 255),
                                                  ((2, 3, 70, 50), 0.0, 38 / 255, 76 / 255), ((2, 2, 17, 130), 0.05, 0.1, 0.2),
                                                  ((2, 4, 9, 9), 0.0, 0.05, 0.1), ((1, 3, 224, 224), 0.0, 38 / 255, 76 / 255)])
def test_canny_full_fwd_bwd_vs_oracle_UNPINNED(ops, shape, alpha, low, high):
    rng = np.random.RandomState(sum(shape))
    x = rng.rand(*shape).astype(np.float32)
    x[0, :, 2:8, 3:9] = 0.5  # flat patch: NaN orientation, zero magnitude
    wts = ops.EdgeWeights(1.0)
    e = ops.canny_fwd(dev(x), wts, alpha, low, high)
    oe = O.canny_fwd(x, alpha, low, high)
    # orientation uses atanf (device libm != host libm): a pixel exactly on a 22.5-degree boundary could differ; none may here
    assert_bitexact(e.cpu().numpy(), oe, "canny edge")
    u = rng.randn(shape[0], 1, shape[2], shape[3]).astype(np.float32)
    g = ops.canny_bwd(dev(x), dev(u), wts, alpha, low, high)
    assert_bitexact(g.cpu().numpy(), O.canny_bwd(x, u, alpha, low, high), "canny bwd")
    # fused front end
    xh = (rng.rand(*shape).astype(np.float32) * 1.4 - 0.3)
    x_in, gate, edge = ops.canny_frontend_fwd(dev(x), dev(xh), wts, alpha, low, high, 1.0, want_edge=True)
    assert_bitexact(edge.cpu().numpy(), oe, "fused edge")
    s = xh + np.float32(1.0) * oe
    assert_bitexact(x_in.cpu().numpy(), np.clip(s, 0, 1).astype(np.float32), "fused x_in")
    assert np.array_equal(gate.cpu().numpy(), ((s >= 0) & (s <= 1)).astype(np.uint8))
    g_in = rng.randn(*shape).astype(np.float32)
    g_hfs, g_edge = ops.canny_frontend_bwd(dev(g_in), gate, dev(x), wts, alpha, low, high, 1.0)
    gh = np.where(gate.cpu().numpy() > 0, g_in, np.float32(0))
    uu = gh[:, 0:1].copy()
    for c in range(1, shape[1]):
        uu = uu + gh[:, c:c + 1]
    assert_bitexact(g_hfs.cpu().numpy(), gh, "fused g_hfs")
    assert_bitexact(g_edge.cpu().numpy(), O.canny_bwd(x, uu * np.float32(1.0), alpha, low, high), "fused g_edge")


def test_canny_full_vs_reference_golden_UNPINNED(ops, golden):
    """Against the reference's own CannyFilter forward / backward run with the derived table (make_golden.py)."""
    G = golden("canny_full_unpinned")
    wts = ops.EdgeWeights(1.0)
    for name in ["rand_rgb", "rand_mnist", "rect_rgb"]:
        x, u = G[name + "__x"], G[name + "__u"]
        alpha, low, high = [float(v) for v in G[name + "__alpha_low_high"]]
        e = ops.canny_fwd(dev(x), wts, alpha, low, high)
        assert np.array_equal(e.cpu().numpy(), G[name + "__CannyFilter__edge"])
        g = ops.canny_bwd(dev(x), dev(u), wts, alpha, low, high).cpu().numpy()
        ref = G[name + "__CannyFilter__gx"][:, :1]
        assert np.array_equal(np.isnan(g), np.isnan(ref))
        fin = ~np.isnan(ref)
        np.testing.assert_allclose(g[fin], ref[fin], atol=1e-6)


@pytest.mark.parametrize("shape,low,high", [((2, 3, 64, 64), 0.05, 0.2), ((3, 1, 28, 28), 0.1, 0.3), ((2, 3, 17, 23), 0.02, 0.1),
                                            ((2, 2, 5, 4), 0.05, 0.2), ((1, 3, 80, 72), 0.2, 0.45)])
def test_canny_bpda_fwd_bwd_vs_oracle_UNPINNED(ops, shape, low, high):
    """CannyFilter_BPDA (core.py:386-505) on the HIP kernels vs the C restatement: bit-exact, NaN gradients included."""
    rng = np.random.default_rng(sum(shape))
    x = rng.random(shape, dtype=np.float32)
    x[0, :, : shape[2] // 2, : shape[3] // 2] = 0.25  # a flat patch: zero magnitude -> 0 * inf = NaN in the gradient
    u = rng.standard_normal((shape[0], 1) + shape[2:]).astype(np.float32)
    wts = ops.EdgeWeights(1.0)
    e, thin, t2 = ops.canny_bpda_fwd(dev(x), wts, low, high)
    assert_bitexact(e.cpu().numpy(), O.canny_bpda_fwd(x, low, high), "bpda fwd")
    g = ops.canny_bpda_bwd(dev(x), dev(u), thin, t2, wts, low, high)
    assert_bitexact(g.cpu().numpy(), O.canny_bpda_bwd(x, u, low, high), "bpda bwd")


def test_canny_bpda_module_vs_reference_golden_UNPINNED(golden):
    """utils.core.CannyFilter_BPDA (HIP path) against the fixtures generated by the reference's class (derived thin table)."""
    import utils.core as Cm
    G = golden("canny_full_unpinned")
    for name in ("rand_rgb", "rand_mnist", "rect_rgb"):
        x = dev(G[name + "__x"]).requires_grad_(True)
        _, low, high = [float(v) for v in G[name + "__alpha_low_high"]]
        filt = Cm.CannyFilter_BPDA(sigma=1, use_cuda=True)
        e = filt(x, low_threshold=low, high_threshold=high, hysteresis=True)
        (e * dev(G[name + "__u"])).sum().backward()
        assert np.array_equal(e.detach().cpu().numpy(), G[name + "__CannyFilter_BPDA__edge"]), name
        ref, got = G[name + "__CannyFilter_BPDA__gx"], x.grad.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)), name
        fin = ~np.isnan(ref)
        np.testing.assert_allclose(got[fin], ref[fin], atol=1e-6)


# ---- fused BatchNorm2d (+ residual) (+ ReLU): CNN-body glue, checked against torch's own fp32 ops (tolerance 1e-5) ----
@pytest.mark.parametrize("shape", [(100, 64, 32, 32), (40, 3, 36, 36), (100, 128, 8, 8), (7, 5, 3, 3), (2, 512, 2, 2), (3, 4, 1, 1), (100, 512, 1, 1)])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False), (False, True)])
@pytest.mark.parametrize("training", [True, False])
def test_bn_act_matches_torch(ops, shape, relu, res, training):
    import torch.nn.functional as F
    from eeadv.functional import BnActFn
    g = torch.Generator(device="cpu").manual_seed(sum(shape) + 2 * relu + res)
    B, C = shape[:2]
    x = (torch.randn(shape, generator=g) * 2 + 0.5).to(DEV).requires_grad_(True)
    r = torch.randn(shape, generator=g).to(DEV).requires_grad_(True) if res else None
    w = (torch.rand(C, generator=g) + 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(C, generator=g).to(DEV).requires_grad_(True)
    rm0, rv0 = torch.randn(C, generator=g).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    dy = torch.randn(shape, generator=g).to(DEV)

    rm_a, rv_a = rm0.clone(), rv0.clone()
    ref = F.batch_norm(x, rm_a, rv_a, w, b, training, 0.1, 1e-5)
    if res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    ins = [x, w, b] + ([r] if res else [])
    g_ref = torch.autograd.grad(ref, ins, dy)

    rm_b, rv_b = rm0.clone(), rv0.clone()
    got = BnActFn.apply(x, r, w, b, rm_b, rv_b, 0.1, 1e-5, training, relu)
    g_got = torch.autograd.grad(got, ins, dy)

    n = B * shape[2] * shape[3]
    if training and n == 1:
        return  # torch refuses one value per channel in training mode; nothing to compare
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(rm_b, rm_a, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv_b, rv_a, rtol=1e-5, atol=1e-6)
    scale = max(1.0, float(n) ** 0.5)
    for a, e, nm in zip(g_got, g_ref, ["dx", "dgamma", "dbeta", "dres"]):
        # ReLU ties (y == 0 exactly) cannot occur with these random inputs; sums over n elements get an n^(1/2)-scaled tolerance
        tol = 2e-5 * (scale if nm in ("dgamma", "dbeta") else 1.0)
        torch.testing.assert_close(a, e, rtol=2e-5, atol=tol, msg=lambda m: nm + ": " + m)


@pytest.mark.parametrize("shape", [(100, 64, 16, 16), (100, 128, 8, 8), (100, 256, 4, 4), (100, 64, 32, 32), (7, 5, 3, 3), (3, 4, 1, 1)])
@pytest.mark.parametrize("relu,res", [(True, True), (True, False), (False, True)])
def test_bn_act_forked_output_adds_the_two_gradients_on_load(ops, shape, relu, res):
    """BnActFn(fork=True) hands its output out twice (next block's convolution + identity branch, resnet.py:44-59); the two gradients
    reach the backward kernel separately and are added on load (every kernel variant: cached <256,2> / <256,7> / <1024,7>, split, generic):
    bit-identical to the unforked function fed with their sum (the same fp32 add autograd's own accumulation performs)."""
    from eeadv.functional import BnActFn
    g = torch.Generator(device="cpu").manual_seed(sum(shape) + relu + 2 * res)
    C = shape[1]
    x = (torch.randn(shape, generator=g) * 2 + 0.5).to(DEV).requires_grad_(True)
    r = torch.randn(shape, generator=g).to(DEV).requires_grad_(True) if res else None
    w = (torch.rand(C, generator=g) + 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(C, generator=g).to(DEV).requires_grad_(True)
    d1, d2 = torch.randn(shape, generator=g).to(DEV), torch.randn(shape, generator=g).to(DEV)
    ins = [x, w, b] + ([r] if res else [])
    stats = lambda: (torch.zeros(C, device=DEV), torch.ones(C, device=DEV))
    one = BnActFn.apply(x, r, w, b, *stats(), 0.1, 1e-5, True, relu)
    ref = torch.autograd.grad(one, ins, d1 + d2)
    ya, yb = BnActFn.apply(x, r, w, b, *stats(), 0.1, 1e-5, True, relu, True)
    assert ya.data_ptr() == yb.data_ptr() and torch.equal(ya, one)
    got = torch.autograd.grad([ya, yb], ins, [d1, d2])
    for a, e in zip(got, ref):
        assert torch.equal(a, e)
    # one consumer only: the other piece never arrives
    ya, yb = BnActFn.apply(x, r, w, b, *stats(), 0.1, 1e-5, True, relu, True)
    for a, e in zip(torch.autograd.grad([yb], ins, [d2]), torch.autograd.grad(BnActFn.apply(x, r, w, b, *stats(), 0.1, 1e-5, True, relu), ins, d2)):
        assert torch.equal(a, e)


@pytest.mark.parametrize("shape", [(100, 64, 16, 16), (100, 128, 8, 8), (100, 256, 4, 4), (100, 64, 32, 32), (7, 5, 3, 3), (2, 3, 5, 7)])
@pytest.mark.parametrize("training", [True, False])
def test_bn_act_backward_relu_mask_recomputed_from_x(ops, shape, training):
    """Without a residual branch the backward's ReLU mask (y > 0) is recomputed from x, gamma, beta with the forward's expression
    instead of reading y (ee_bn_act_bwd2_f32 with y = NULL): every kernel variant returns the same bits as with y, also where
    the pre-activation is exactly 0, tiny, or NaN."""
    g = torch.Generator(device="cpu").manual_seed(sum(shape) + training)
    C = shape[1]
    x = torch.randn(shape, generator=g) * 2 + 0.5
    w = (torch.rand(C, generator=g) + 0.5)
    w[0] = -w[0]
    b = torch.randn(C, generator=g)
    if C > 2:
        b[2] = 0.0
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    if not training and x.numel() > 60:
        x[0, 0].view(-1)[:3] = torch.tensor([float("nan"), float("inf"), float("-inf")])[:x[0, 0].numel()][:3] if x[0, 0].numel() >= 3 else x[0, 0].view(-1)[:3]
        x[0, C - 1].view(-1)[0] = rm[C - 1]  # x - mean == 0 exactly: pre-activation = beta
    x, w, b, rm, rv = (t.to(DEV) for t in (x, w, b, rm, rv))
    y, sm, si = ops.bn_act_fwd(x, None, w, b, rm.clone(), rv.clone(), 0.1, 1e-5, training, True)
    dy, dy2 = torch.randn(shape, generator=g).to(DEV), torch.randn(shape, generator=g).to(DEV)
    for second in (None, dy2):
        ref = ops.bn_act_bwd(dy, y, x, w, sm, si, rm, rv, 1e-5, training, True, True, False, True, second)
        got = ops.bn_act_bwd(dy, None, x, w, sm, si, rm, rv, 1e-5, training, True, True, False, True, second, b)
        for a, e in zip(got, ref):
            if e is not None:
                assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(e, nan=-7.0))


@pytest.mark.parametrize("shape", [(100, 128, 8, 8), (100, 256, 4, 4), (100, 512, 2, 2), (3, 6, 2, 2), (9, 4, 4, 8)])
@pytest.mark.parametrize("training", [True, False])
def test_bn_dual_equals_the_two_batchnorm_calls(ops, shape, training):
    """relu(bn2(conv2 out) + bn_ds(conv1x1 out)) - the end of a block with a down-sampling shortcut (resnet.py:54-59, :137-142) - in one
    launch each way: output, running statistics and every gradient bit-identical to BnActFn(xb, relu=False) -> BnActFn(xa, residual,
    relu=True), one gradient piece or two (forked output)."""
    from eeadv.functional import BnActFn, BnDualFn
    g = torch.Generator(device="cpu").manual_seed(sum(shape) + training)
    C = shape[1]
    assert ops.bn_dual_supported(torch.empty(shape, device=DEV))
    mk = lambda: (torch.randn(shape, generator=g) * 2 + 0.5).to(DEV).requires_grad_(True)
    xa, xb = mk(), mk()
    pr = lambda: ((torch.rand(C, generator=g) + 0.5).to(DEV).requires_grad_(True), torch.randn(C, generator=g).to(DEV).requires_grad_(True))
    (ga, ba), (gb, bb) = pr(), pr()
    st0 = [torch.randn(C, generator=g).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)]
    ins = [xa, xb, ga, ba, gb, bb]
    d1, d2 = torch.randn(shape, generator=g).to(DEV), torch.randn(shape, generator=g).to(DEV)
    s_two = [t.clone() for t in st0]
    res = BnActFn.apply(xb, None, gb, bb, s_two[2], s_two[3], 0.1, 1e-5, training, False)
    two = BnActFn.apply(xa, res, ga, ba, s_two[0], s_two[1], 0.2, 2e-5, training, True)
    s_one = [t.clone() for t in st0]
    one = BnDualFn.apply(xa, xb, ga, ba, s_one[0], s_one[1], 0.2, 2e-5, gb, bb, s_one[2], s_one[3], 0.1, 1e-5, training)
    assert torch.equal(one, two)
    for a, e in zip(s_one, s_two):
        assert torch.equal(a, e)
    for a, e in zip(torch.autograd.grad(one, ins, d1, retain_graph=True), torch.autograd.grad(two, ins, d1, retain_graph=True)):
        assert torch.equal(a, e)
    ya, yb = BnDualFn.apply(xa, xb, ga, ba, st0[0].clone(), st0[1].clone(), 0.2, 2e-5, gb, bb, st0[2].clone(), st0[3].clone(), 0.1, 1e-5, training, True)
    for a, e in zip(torch.autograd.grad([ya, yb], ins, [d1, d2]), torch.autograd.grad(two, ins, d1 + d2)):
        assert torch.equal(a, e)
    assert not ops.bn_dual_supported(torch.empty(100, 64, 32, 32, device=DEV))  # the stem's size: split kernels, two calls


def test_bn_act_only_input_grad_and_reproducible(ops):
    """The attack loop differentiates w.r.t. the input only (attacks.py:24): no parameter gradients are produced, and two
    runs give the same bits (fixed-order reductions)."""
    from eeadv.functional import BnActFn
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(16, 8, 4, 4, generator=g).to(DEV).requires_grad_(True)
    w, b = torch.ones(8, device=DEV, requires_grad=True), torch.zeros(8, device=DEV, requires_grad=True)
    outs = []
    for _ in range(2):
        rm, rv = torch.zeros(8, device=DEV), torch.ones(8, device=DEV)
        y = BnActFn.apply(x, None, w, b, rm, rv, 0.1, 1e-5, True, True)
        (gx,) = torch.autograd.grad(y.square().sum(), x)
        outs.append((y.detach().clone(), gx.clone(), rm, rv))
    for a, e in zip(outs[0], outs[1]):
        assert torch.equal(a, e)
    assert w.grad is None and b.grad is None


@pytest.mark.parametrize("shape", [(100, 64, 32, 32), (2, 3, 7, 9), (3, 2, 1, 1), (2, 2, 2, 5), (1, 1, 112, 112)])
def test_maxpool3s2_bit_identical_to_aten(ops, shape):
    """Stem MaxPool2d(3,2,1) (resnet.py:117): values, and the gradient routing (first maximum wins), equal ATen's bit for bit,
    ties and NaNs included."""
    import torch.nn.functional as F
    from eeadv.functional import MaxPool3s2Fn
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    x = torch.randn(shape, generator=g)
    x = torch.where(torch.rand(shape, generator=g) < 0.3, torch.zeros(()), x)  # many exact ties (post-ReLU zeros)
    if x.numel() > 50:
        x.view(-1)[7] = float("nan")
    x = x.to(DEV).requires_grad_(True)
    ref = F.max_pool2d(x, 3, 2, 1)
    got = MaxPool3s2Fn.apply(x)
    dy = torch.randn(ref.shape, generator=g).to(DEV)
    (g_ref,) = torch.autograd.grad(ref, x, dy)
    (g_got,) = torch.autograd.grad(got, x, dy)
    assert_bitexact(got.detach().cpu().numpy(), ref.detach().cpu().numpy(), "maxpool fwd")
    assert_bitexact(g_got.cpu().numpy(), g_ref.cpu().numpy(), "maxpool bwd")


@pytest.mark.parametrize("shape", [(100, 64, 32, 32), (70, 3, 6, 8), (3, 5, 7, 12), (2, 64, 112, 112), (130, 2, 1, 4)])
@pytest.mark.parametrize("training", [True, False])
def test_bn_relu_pool_fused_equals_the_two_kernels(ops, shape, training):
    """The ResNet stem's maxpool(relu(bn(x))) (resnet.py:113-117) in one pass each way (ee_bn.hip, bn_pool_*): pooled values, running
    statistics and the gradient routing are bit-identical to BnActFn -> MaxPool3s2Fn (the same statistics, the same ReLU bits, ATen's
    first-maximum rule), the gradient sums - taken over another partition - agree to rounding; and against torch's own ops."""
    import torch.nn.functional as F
    from eeadv.functional import BnActFn, BnReluPoolFn, MaxPool3s2Fn
    g = torch.Generator(device="cpu").manual_seed(sum(shape) + training)
    B, C = shape[:2]
    x = (torch.randn(shape, generator=g) * 2 + 0.5).to(DEV).requires_grad_(True)
    w = (torch.rand(C, generator=g) + 0.5).to(DEV)
    if C > 1:
        w[1] = -w[1]  # a negative scale: the pool's argmax is NOT the argmax of x there
    w.requires_grad_(True)
    b = torch.randn(C, generator=g).to(DEV).requires_grad_(True)
    rm0, rv0 = torch.randn(C, generator=g).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    assert ops.bn_relu_pool_supported(x)
    rm_a, rv_a = rm0.clone(), rv0.clone()
    two = MaxPool3s2Fn.apply(BnActFn.apply(x, None, w, b, rm_a, rv_a, 0.1, 1e-5, training, True))
    rm_b, rv_b = rm0.clone(), rv0.clone()
    one = BnReluPoolFn.apply(x, w, b, rm_b, rv_b, 0.1, 1e-5, training)
    from eeadv import _native as N
    if not training or N.lib.ee_bn_workspace_floats(B, C, shape[2] * shape[3]) > 0:
        # the unfused BatchNorm takes its statistics the same way (eval mode; or the split kernels, as for the stem at the reference batch)
        assert torch.equal(torch.nan_to_num(one, nan=-7.0), torch.nan_to_num(two, nan=-7.0)) and torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
    else:  # it keeps the channel in registers there: same two-pass variance, another summation order
        torch.testing.assert_close(one, two, rtol=1e-6, atol=2e-6)
        torch.testing.assert_close(rm_b, rm_a, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(rv_b, rv_a, rtol=1e-6, atol=1e-7)
    dy = torch.randn(two.shape, generator=g).to(DEV)
    g2 = torch.autograd.grad(two, [x, w, b], dy)
    g1 = torch.autograd.grad(one, [x, w, b], dy)
    n = B * shape[2] * shape[3]
    scale = max(1.0, float(n) ** 0.5)
    for a, e, nm in zip(g1, g2, ["dx", "dgamma", "dbeta"]):
        torch.testing.assert_close(a, e, rtol=2e-5, atol=2e-5 * (scale if nm != "dx" else 1.0), equal_nan=True, msg=lambda m: nm + ": " + m)
    assert torch.equal(g1[0] == 0, g2[0] == 0) or training  # eval mode: dx = a * dz, zero exactly where the ReLU / the pool cut it
    rm_c, rv_c = rm0.clone(), rv0.clone()
    ref = F.max_pool2d(F.relu(F.batch_norm(x, rm_c, rv_c, w, b, training, 0.1, 1e-5)), 3, 2, 1)
    torch.testing.assert_close(one, ref, rtol=1e-5, atol=2e-5, equal_nan=True)
    for a, e, nm in zip(g1, torch.autograd.grad(ref, [x, w, b], dy), ["dx", "dgamma", "dbeta"]):
        torch.testing.assert_close(a, e, rtol=2e-5, atol=2e-5 * (scale if nm != "dx" else 1.0), equal_nan=True, msg=lambda m: nm + " vs torch: " + m)
    with EF_input_grad_only():
        (gx,) = torch.autograd.grad(BnReluPoolFn.apply(x, w, b, rm0.clone(), rv0.clone(), 0.1, 1e-5, training), [x], dy)
    assert torch.equal(torch.nan_to_num(gx, nan=-7.0), torch.nan_to_num(g1[0], nan=-7.0))
    # forked output (layer1.0's convolution + its identity branch): the two gradient pieces are added on load
    ya, yb = BnReluPoolFn.apply(x, w, b, rm0.clone(), rv0.clone(), 0.1, 1e-5, training, True)
    d2 = torch.randn(two.shape, generator=g).to(DEV)
    for a, e in zip(torch.autograd.grad([ya, yb], [x, w, b], [dy, d2]),
                    torch.autograd.grad(BnReluPoolFn.apply(x, w, b, rm0.clone(), rv0.clone(), 0.1, 1e-5, training), [x, w, b], dy + d2)):
        assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(e, nan=-7.0))
    assert not ops.bn_relu_pool_supported(torch.empty(2, 3, 224, 224, device=DEV)) and not ops.bn_relu_pool_supported(torch.empty(2, 3, 8, 6, device=DEV))
    if not training and x.numel() > 50:
        # eval mode (the statistics do not see x): a NaN wins its pooling windows and gets no gradient through the ReLU, +-inf pass -
        # exactly as in the two separate kernels
        xn = x.detach().clone()
        xn.view(-1)[11], xn.view(-1)[-3], xn.view(-1)[5] = float("nan"), float("inf"), float("-inf")
        xn.requires_grad_(True)
        a = BnReluPoolFn.apply(xn, w, b, rm0.clone(), rv0.clone(), 0.1, 1e-5, False)
        e = MaxPool3s2Fn.apply(BnActFn.apply(xn, None, w, b, rm0.clone(), rv0.clone(), 0.1, 1e-5, False, True))
        assert bool(torch.isnan(a).any()) and torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(e, nan=-7.0))
        (ga,), (ge,) = torch.autograd.grad(a, [xn], dy), torch.autograd.grad(e, [xn], dy)
        assert torch.equal(torch.nan_to_num(ga, nan=-7.0), torch.nan_to_num(ge, nan=-7.0))


def EF_input_grad_only():
    from eeadv.functional import input_grad_only
    return input_grad_only()


@pytest.mark.parametrize("B,C,HW,K", [(100, 512, (2, 2), 200), (3, 2048, (7, 7), 1000), (5, 7, (1, 1), 3), (2, 64, (3, 5), 10)])
def test_pool_linear_head_matches_torch(ops, B, C, HW, K):
    import torch.nn.functional as F
    from eeadv.functional import PoolLinearFn
    g = torch.Generator(device="cpu").manual_seed(B + C + K)
    feat = torch.randn(B, C, *HW, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(K, C, generator=g) / C ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(K, generator=g).to(DEV).requires_grad_(True)
    dl = torch.randn(B, K, generator=g).to(DEV)
    ref = F.linear(F.adaptive_avg_pool2d(feat, 1).view(B, -1), w, b)
    got = PoolLinearFn.apply(feat, w, b)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
    for a, e in zip(torch.autograd.grad(got, [feat, w, b], dl), torch.autograd.grad(ref, [feat, w, b], dl, retain_graph=True)):
        torch.testing.assert_close(a, e, rtol=1e-5, atol=1e-5)
    (gx,) = torch.autograd.grad(PoolLinearFn.apply(feat, w, None), feat, dl)  # no bias; input gradient only
    torch.testing.assert_close(gx, torch.autograd.grad(ref, feat, dl)[0], rtol=1e-5, atol=1e-5)


def test_square_draw_one_launch_statistics_and_graph_replay(ops):
    """Add_Square's draws (core.py:637, :645, :648) from the device-side Philox state: right supports and frequencies,
    reproducible from the state, and a replayed HIP graph draws NEW numbers (the kernel advances the state itself)."""
    h, C, B = 64, 3, 100
    sizes = torch.tensor([41, 29, 1, 64, 7], dtype=torch.int32, device=DEV)
    state = torch.tensor([1234, 0, 0, 0], dtype=torch.int64, device=DEV)
    stripe, pos, sign = ops.square_draw(B, C, h, sizes, state)
    assert stripe.shape == (B, C, 1, h) and pos.shape == (5,) and sign.shape == (5, C) and pos.dtype == torch.int64
    assert set(np.unique(stripe.cpu().numpy())) <= {-1.0, 0.0, 1.0}
    assert abs(float(stripe.mean())) < 0.03 and float((stripe == 0).float().mean()) < 1e-3
    assert bool(((pos >= 0) & (pos <= (h - sizes).clamp(min=0))).all()) and int(pos[3]) == 0
    n_used = (B * C * h + 5 * (1 + C) + 3) // 4
    assert state.tolist() == [1234, n_used, 0, 0]
    state2 = torch.tensor([1234, 0, 0, 0], dtype=torch.int64, device=DEV)
    again = ops.square_draw(B, C, h, sizes, state2)
    assert torch.equal(again[0], stripe) and torch.equal(again[1], pos) and torch.equal(again[2], sign)
    nxt = ops.square_draw(B, C, h, sizes, state2)
    assert not torch.equal(nxt[0], stripe)
    # positions are uniform over [0, h - s]: many draws of one size
    many = torch.full((4000,), 24, dtype=torch.int32, device=DEV)
    _, p2, s2 = ops.square_draw(1, C, h, many, state2)
    cnt = torch.bincount(p2, minlength=41).float()
    assert int(p2.max()) <= 40 and float(cnt[:40].min()) > 50 and abs(float(s2.mean())) < 0.05
    # graph replay
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.square_draw(B, C, h, sizes, state2)
    g.replay()
    a = out[0].clone()
    g.replay()
    assert not torch.equal(out[0], a)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(100, 64, 128, 16, 16), (100, 128, 256, 8, 8), (100, 256, 512, 4, 4), (3, 6, 34, 10, 6),
                                            (2, 130, 200, 4, 4), (1, 2, 2, 2, 2), (5, 64, 256, 56, 56)])
def test_conv1x1s2_matches_aten(ops, B, Cin, Cout, H, W):
    """The shortcut convolution (resnet.py:137-142) on the f32 matrix cores vs ATen's conv2d: forward, input gradient and
    (through ATen) weight gradient."""
    import torch.nn.functional as F
    from eeadv.functional import Conv1x1S2Fn
    g = torch.Generator(device="cpu").manual_seed(B + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).to(DEV).requires_grad_(True)
    ref = F.conv2d(x, w, None, 2, 0)
    got = Conv1x1S2Fn.apply(x, w)
    dy = torch.randn(ref.shape, generator=g).to(DEV)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
    (gx, gw), (ex, ew) = torch.autograd.grad(got, [x, w], dy), torch.autograd.grad(ref, [x, w], dy)
    torch.testing.assert_close(gx, ex, rtol=1e-5, atol=2e-5)
    # both weight gradients come from MIOpen (sums over B*OH*OW terms, possibly through different solvers)
    torch.testing.assert_close(gw, ew, rtol=1e-4, atol=1e-4 * (B * H * W / 4) ** 0.5)


@pytest.mark.parametrize("B,K,H,W", [(100, 64, 64, 64), (3, 64, 224, 224), (2, 5, 10, 6), (1, 64, 2, 2), (2, 17, 70, 66)])
def test_stem_conv_bwd_data_matches_aten(ops, B, K, H, W):
    """d loss / d image through the stem convolution (resnet.py:112) vs ATen's convolution_backward."""
    import torch.nn.functional as F
    from eeadv.functional import StemConvFn
    g = torch.Generator(device="cpu").manual_seed(B + K + H)
    x = torch.randn(B, 3, H, W, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(K, 3, 7, 7, generator=g) / 12.0).to(DEV).requires_grad_(True)
    ref = F.conv2d(x, w, None, 2, 3)
    got = StemConvFn.apply(x, w)
    dy = torch.randn(ref.shape, generator=g).to(DEV)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
    (gx, gw), (ex, ew) = torch.autograd.grad(got, [x, w], dy), torch.autograd.grad(ref, [x, w], dy)
    torch.testing.assert_close(gx, ex, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gw, ew, rtol=1e-4, atol=1e-4 * (B * H * W / 4) ** 0.5)


@pytest.mark.parametrize("B,K,H,W", [(100, 64, 64, 64), (3, 128, 10, 64), (2, 64, 6, 128), (1, 64, 2, 64), (5, 192, 64, 192)])
def test_stem_conv_fwd_mfma_matches_aten(ops, B, K, H, W):
    """The stem convolution forward (resnet.py:112) on the f32 matrix cores vs ATen: image borders on all four sides, output rows
    that do not fill the 4-row tile (OH = 5, 3, 1), several column tiles and channel blocks; then through StemConvFn in a model-like
    call (no input gradient: the update pass)."""
    import torch.nn.functional as F
    from eeadv.functional import StemConvFn
    g = torch.Generator(device="cpu").manual_seed(B + K + H + W)
    x = torch.randn(B, 3, H, W, generator=g).to(DEV)
    w = (torch.randn(K, 3, 7, 7, generator=g) / 12.0).to(DEV)
    assert ops.stem7x7s2_fwd_supported(x, w)
    ref = F.conv2d(x, w, None, 2, 3)
    got = ops.stem7x7s2_fwd(x, w)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-5)
    assert float((got.double() - F.conv2d(x.double(), w.double(), None, 2, 3)).abs().max()) < 1e-5  # exact-f32 fma chains over K = 147
    wp = w.clone().requires_grad_(True)
    out = StemConvFn.apply(x, wp)
    assert torch.equal(out, got)
    dy = torch.randn(ref.shape, generator=g).to(DEV)
    gw, = torch.autograd.grad(out, [wp], dy)
    ew, = torch.autograd.grad(F.conv2d(x, wp, None, 2, 3), [wp], dy)
    torch.testing.assert_close(gw, ew, rtol=1e-4, atol=1e-4 * (B * H * W / 4) ** 0.5)
    assert not ops.stem7x7s2_fwd_supported(torch.empty(1, 3, 224, 224, device=DEV), w)  # OW = 112: MIOpen keeps ImageNet's stem
    # the epilogue's per-workgroup moments of y (for bn1): same y, and merged they are the batch statistics of y
    y2, st = ops.stem7x7s2_fwd(x, w, True)
    assert torch.equal(y2, got) and st.shape[0] == K and st.shape[2] == 3
    n = st[:, :, 2].double().sum(1)
    mean = st[:, :, 0].double().sum(1) / n
    tile_mean = st[:, :, 0].double() / st[:, :, 2].double().clamp_min(1)
    m2 = (st[:, :, 1].double() + st[:, :, 2].double() * (tile_mean - mean[:, None]) ** 2).sum(1)
    assert torch.all(n == B * (H // 2) * (W // 2))
    ref64 = got.double().transpose(0, 1).reshape(K, -1)
    torch.testing.assert_close(mean, ref64.mean(1), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(m2 / n, ref64.var(1, unbiased=False), rtol=1e-5, atol=1e-9)
    if H * W // 4 <= 16000:
        from eeadv.functional import BnReluPoolFn
        gam, bet = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV)
        outs = []
        for cs in (None, st):
            rm, rv = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
            outs.append((BnReluPoolFn.apply(got, gam, bet, rm, rv, 0.1, 1e-5, True, False, cs), rm, rv))
        for a, e in zip(outs[1], outs[0]):  # statistics merged from the convolution's tiles vs bn_split_stats_kernel's own two passes
            torch.testing.assert_close(a, e, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("B,Cin,Cout", [(100, 512, 512), (3, 5, 7), (1, 64, 32)])
def test_conv3x3_on_2x2_map_as_dense_product(ops, B, Cin, Cout):
    """Conv2d(3x3, stride 1, padding 1) on a 2x2 map (ResNet layer4 at 64x64 inputs) as one GEMM vs ATen, and the rearranged
    weight matrix follows in-place weight updates (version counter) without changing its address (HIP graphs)."""
    import torch.nn.functional as F
    from eeadv import functional as EF
    g = torch.Generator(device="cpu").manual_seed(B + Cin)
    x = torch.randn(B, Cin, 2, 2, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(DEV).requires_grad_(True)
    dy = torch.randn(B, Cout, 2, 2, generator=g).to(DEV)
    for round_ in range(2):
        ref = F.conv2d(x, w, None, 1, 1)
        got = EF.Conv3x3Map2Fn.apply(x, w)
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
        (gx, gw), (ex, ew) = torch.autograd.grad(got, [x, w], dy), torch.autograd.grad(ref, [x, w], dy)
        torch.testing.assert_close(gx, ex, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(gw, ew, rtol=1e-4, atol=1e-3)
        ptr = EF._DENSE_W[(id(w), "s1")][2].data_ptr()
        with torch.no_grad():
            w.mul_(0.5).add_(0.01)  # an optimiser step: the next forward must see it
        if round_ == 1:
            assert EF._DENSE_W[(id(w), "s1")][2].data_ptr() == ptr
    # a captured forward reads the buffer refreshed by refresh_dense_weights()
    with torch.no_grad():
        xs = x.detach()
        EF.Conv3x3Map2Fn.apply(xs, w)
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            out = EF.Conv3x3Map2Fn.apply(xs, w)
        w.add_(0.25)
        EF.refresh_dense_weights()
        gph.replay()
        torch.testing.assert_close(out, F.conv2d(xs, w, None, 1, 1), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(4, 3, 64, 64), (3, 1, 28, 28), (2, 3, 17, 23), (2, 2, 5, 4), (1, 3, 80, 72), (2, 4, 224, 224)])
def test_frontend_bwd_from_saved_responses_is_bit_identical(ops, shape):
    """The forward that keeps gx, gy + the backward that consumes them (no x, no blur / Sobel recomputation) against the
    recomputing pair: every output bit-identical, NaN gradients included; and against the C oracle."""
    rng = np.random.default_rng(sum(shape) + 3)
    x = rng.random(shape, dtype=np.float32)
    x[0, :, : shape[2] // 2, : shape[3] // 2] = 0.25  # flat patch: zero magnitude -> NaN in the gradient
    xh = (rng.random(shape, dtype=np.float32) * 1.2 - 0.1).astype(np.float32)
    g_in = rng.standard_normal(shape).astype(np.float32)
    wts = ops.EdgeWeights(1.0)
    alpha, high, w = 0.05, 0.2, 0.5
    a_in, a_gate, a_edge = ops.frontend_fwd(dev(x), dev(xh), wts, alpha, high, w, want_edge=True)
    b_in, b_gate, b_edge, gx, gy = ops.frontend_fwd_save(dev(x), dev(xh), wts, alpha, high, w, want_edge=True)
    assert torch.equal(a_in, b_in) and torch.equal(a_gate, b_gate) and torch.equal(a_edge, b_edge)
    _, _, ogx, ogy = O.edge125_fwd(x, alpha, high, want_internals=True)[0:4]
    assert_bitexact(gx.cpu().numpy(), ogx, "saved gx")
    assert_bitexact(gy.cpu().numpy(), ogy, "saved gy")
    r_hfs, r_edge = ops.frontend_bwd(dev(g_in), a_gate, dev(x), wts, alpha, high, w)
    s_hfs, s_edge = ops.frontend_bwd_saved(dev(g_in), b_gate, gx, gy, wts, alpha, high, w)
    assert_bitexact(s_hfs.cpu().numpy(), r_hfs.cpu().numpy(), "g_hfs")
    assert_bitexact(s_edge.cpu().numpy(), r_edge.cpu().numpy(), "g_edge")
    if shape[2] >= 16:
        assert int(torch.isnan(s_edge).sum()) > 0


# ---- Net_2's convolutional half (MNIST/models_mnist/Net2.py:13-14) as two launches each way (ee_net2.hip) ----------------------------
@pytest.mark.parametrize("B,with_drop", [(50, False), (50, True), (3, True), (1, False)])
def test_net2_conv_half_matches_aten(ops, B, with_drop):
    """relu(max_pool2d(drop * conv2(relu(max_pool2d(conv1(x), 2))), 2)): values within 1e-5 of ATen's sequence, the input gradient within
    1e-5 of autograd's (the same pool winners and ReLU masks - compared on inputs without ties), parameter gradients on the hand-written
    kernels too, NaN / inf inputs keep ATen's footprint."""
    import torch.nn.functional as F
    from eeadv.functional import Net2ConvFn
    g = torch.Generator(device="cpu").manual_seed(B + 7 * with_drop)
    x = torch.rand(B, 1, 28, 28, generator=g).to(DEV).requires_grad_(True)
    w1 = (torch.randn(32, 1, 5, 5, generator=g) * 0.2).to(DEV).requires_grad_(True)
    b1 = (torch.randn(32, generator=g) * 0.1).to(DEV).requires_grad_(True)
    w2 = (torch.randn(64, 32, 5, 5, generator=g) * 0.05).to(DEV).requires_grad_(True)
    b2 = (torch.randn(64, generator=g) * 0.1).to(DEV).requires_grad_(True)
    keep = 0.7 if B == 3 else 0.5
    draw = (torch.rand(B, 64, generator=g) < keep).float().to(DEV) if with_drop else None  # Bernoulli(keep): 0 / 1
    drop = draw.div(keep) if with_drop else None                                           # what the stock sequence multiplies by
    assert ops.net2_conv_supported(x, w1, w2)

    def stock(xx):
        h = F.relu(F.max_pool2d(F.conv2d(xx, w1, b1), 2))
        h = F.conv2d(h, w2, b2)
        if drop is not None:
            h = h * drop.view(B, 64, 1, 1)
        return F.relu(F.max_pool2d(h, 2))

    ref = stock(x)
    got = Net2ConvFn.apply(x, w1, b1, w2, b2, draw, keep)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
    assert torch.equal(got == 0, ref == 0)
    dy = torch.randn(ref.shape, generator=g).to(DEV)
    from eeadv.functional import input_grad_only
    with input_grad_only():  # the attack loop: the HIP backward
        (gx,) = torch.autograd.grad(got, [x], dy, retain_graph=True)
    (ex,) = torch.autograd.grad(ref, [x], dy, retain_graph=True)
    torch.testing.assert_close(gx, ex, rtol=1e-4, atol=1e-5 * float(ex.abs().max()))
    # the update: parameter gradients (and x's) - ee_net2.hip's net2_conv*_wrw kernels - against autograd through the stock sequence, against its
    # float64 twin, and bit-identical from call to call (the images are added in order)
    mine = torch.autograd.grad(got, [x, w1, b1, w2, b2], dy, retain_graph=True)
    for a, e in zip(mine, torch.autograd.grad(ref, [x, w1, b1, w2, b2], dy)):
        torch.testing.assert_close(a, e, rtol=1e-4, atol=1e-5 * float(e.abs().max()))
    p64 = [t.detach().double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    h64 = F.relu(F.max_pool2d(F.conv2d(p64[0], p64[1], p64[2]), 2))
    h64 = F.conv2d(h64, p64[3], p64[4])
    if drop is not None:
        h64 = h64 * drop.double().view(B, 64, 1, 1)
    h64 = F.relu(F.max_pool2d(h64, 2))
    for a, e in zip(mine, torch.autograd.grad(h64, p64, dy.double())):
        assert float((a.double() - e).abs().max()) <= 2e-5 * float(e.abs().max()) + 1e-7
    for a, e in zip(mine, torch.autograd.grad(got, [x, w1, b1, w2, b2], dy)):
        assert torch.equal(a, e)
    # NaN / inf in the image: the NaN footprint of a direct convolution (float64 on the host - MIOpen's Winograd spreads a NaN over its
    # whole transform tile and turns inf - inf into NaN, so the stock GPU sequence is no yardstick here), forward and backward
    xn = x.detach().clone()
    xn[0, 0, 5, 7], xn[B - 1, 0, 20, 3] = float("nan"), float("inf")
    xn.requires_grad_(True)
    gotn = Net2ConvFn.apply(xn, w1, b1, w2, b2, draw, keep)
    with input_grad_only():
        (gn,) = torch.autograd.grad(gotn, [xn], dy)
    xc = xn.detach().double().cpu().requires_grad_(True)
    pc = [t.detach().double().cpu() for t in (w1, b1, w2, b2)]
    hc = F.relu(F.max_pool2d(F.conv2d(xc, pc[0], pc[1]), 2))
    hc = F.conv2d(hc, pc[2], pc[3])
    if drop is not None:
        hc = hc * drop.double().cpu().view(B, 64, 1, 1)
    hc = F.relu(F.max_pool2d(hc, 2))
    (gc,) = torch.autograd.grad(hc, [xc], dy.double().cpu())
    assert bool(torch.isnan(gotn).any()) and torch.equal(torch.isnan(gotn).cpu(), torch.isnan(hc))
    assert torch.equal(torch.isnan(gn).cpu(), torch.isnan(gc))


@pytest.mark.parametrize("B,with_drop", [(50, True), (7, False)])
def test_net2_conv2_on_the_matrix_cores_against_the_scalar_kernels(ops, monkeypatch, B, with_drop):
    """round 4: conv2 forward (one image x 16 output channels per workgroup, K = 800 on v_mfma_f32_16x16x4_f32) and backward-data (the
    un-pooled gradient's one non-zero per window, grouped by its position: a 200 x 64 x 64 product per workgroup + a 25-term gather) against the
    scalar kernels of rounds 1-3 (EEADV_NET2_SCALAR=1, read by the library per call): values within 1e-5 (another summation order), the same
    pool winners and ReLU zeros on inputs without ties, the same device-side dropout draws and state."""
    g = torch.Generator(device="cpu").manual_seed(B)
    x = torch.rand(B, 1, 28, 28, generator=g).to(DEV)
    w1 = (torch.randn(32, 1, 5, 5, generator=g) * 0.2).to(DEV)
    b1 = (torch.randn(32, generator=g) * 0.1).to(DEV)
    w2 = (torch.randn(64, 32, 5, 5, generator=g) * 0.05).to(DEV)
    b2 = (torch.randn(64, generator=g) * 0.1).to(DEV)
    da2 = torch.randn(B, 64, 4, 4, generator=g).to(DEV)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("EEADV_NET2_SCALAR", mode)
        state = torch.tensor([5, 12, 0, 0], dtype=torch.int64, device=DEV) if with_drop else None
        a2, saved, mask = ops.net2_conv_fwd(x, w1, b1, w2, b2, None, 0.5 if with_drop else 1.0, state)
        da1 = torch.empty_like(saved[0])
        dx = ops.net2_conv_bwd(da2, a2, saved, w1, w2, mask, 0.5 if with_drop else 1.0, da1_out=da1)
        res[mode] = (a2, saved[2], mask, da1, dx, None if state is None else state.tolist())
    (a_s, c_s, m_s, d_s, x_s, st_s), (a_m, c_m, m_m, d_m, x_m, st_m) = res["1"], res["0"]
    torch.testing.assert_close(a_m, a_s, rtol=1e-5, atol=1e-5)
    assert torch.equal(a_m == 0, a_s == 0) and float((c_m == c_s).float().mean()) > 0.999
    if with_drop:
        assert torch.equal(m_m, m_s) and st_m == st_s and st_m[2] == 0 and st_m[1] == 12 + (B * 64 + 3) // 4
    same = (c_m == c_s).all(dim=(1, 2, 3))  # images whose pool winners agree: the same function is differentiated
    torch.testing.assert_close(d_m[same], d_s[same], rtol=1e-4, atol=1e-5 * float(d_s.abs().max()))
    torch.testing.assert_close(x_m[same], x_s[same], rtol=1e-4, atol=1e-5 * float(x_s.abs().max()))
    assert int(same.sum()) >= B - 1


def test_net2_model_uses_the_fused_half_and_draws_the_stock_dropout_mask(ops):
    """Net_2.body in train mode: Dropout2d's mask is drawn with the calls F.dropout2d makes, so under the same seed the fused path and the
    stock sequence see the same mask (logits within 1e-4) and leave the generator in the same state."""
    import torch.nn.functional as F
    from eeadv.models import Net_2
    torch.manual_seed(3)
    net = Net_2().to(DEV).train()
    x = torch.rand(50, 1, 28, 28, device=DEV)
    torch.manual_seed(11)
    got = net.body(x)
    after_a = torch.cuda.get_rng_state(0)
    torch.manual_seed(11)
    h = F.relu(F.max_pool2d(net.conv1(x), 2))
    h = F.relu(F.max_pool2d(net.conv2_drop(net.conv2(h)), 2))
    ref = net.fc2(F.relu(net.fc1(h.view(-1, 1024))))
    after_b = torch.cuda.get_rng_state(0)
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
    assert torch.equal(after_a, after_b)
    net.eval()
    torch.testing.assert_close(net.body(x), net.fc2(F.relu(net.fc1(F.relu(F.max_pool2d(net.conv2(F.relu(F.max_pool2d(net.conv1(x), 2))), 2)).view(-1, 1024)))),
                               rtol=1e-4, atol=1e-4)


def test_net2_dropout_mask_drawn_on_the_device(ops):
    """ee_net2_conv_fwd_f32 with drop = NULL and a draw state: the second kernel draws Dropout2d's Bernoulli(keep) mask itself (what a
    captured attack iteration uses instead of torch's bernoulli_ launch).  The mask it reports reproduces its output when injected (bit for
    bit), is 0 / 1 with mean ~ keep, differs from launch to launch (the last workgroup advanced the offset, the ticket is back at 0), and a
    replayed HIP graph draws fresh masks too."""
    from eeadv import runtime
    g = torch.Generator(device="cpu").manual_seed(5)
    B, keep = 50, 0.5
    x = torch.rand(B, 1, 28, 28, generator=g).to(DEV)
    w1, b1 = (torch.randn(32, 1, 5, 5, generator=g) * 0.2).to(DEV), (torch.randn(32, generator=g) * 0.1).to(DEV)
    w2, b2 = (torch.randn(64, 32, 5, 5, generator=g) * 0.05).to(DEV), (torch.randn(64, generator=g) * 0.1).to(DEV)
    torch.manual_seed(77)
    runtime.reseed()
    state = runtime.draw_state(torch.device(DEV))
    off0 = int(state[1])
    a2, saved, mask = ops.net2_conv_fwd(x, w1, b1, w2, b2, None, keep, state)
    assert set(torch.unique(mask).tolist()) <= {0.0, 1.0} and 0.4 < float(mask.mean()) < 0.6
    assert int(state[1]) == off0 + (B * 64 + 3) // 4 and int(state[2]) == 0
    again, _, same = ops.net2_conv_fwd(x, w1, b1, w2, b2, mask, keep)
    assert torch.equal(a2, again) and same is mask
    zero = (mask == 0).view(B, 64, 1, 1).expand_as(a2)
    assert float(a2[zero].abs().max()) == 0.0  # dropped channels: relu(max_pool(0)) = 0
    _, _, mask2 = ops.net2_conv_fwd(x, w1, b1, w2, b2, None, keep, state)
    assert not torch.equal(mask, mask2)
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        _, _, mg = ops.net2_conv_fwd(x, w1, b1, w2, b2, None, keep, state)
    seen = []
    for _ in range(3):
        gph.replay()
        seen.append(mg.clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    total = torch.stack(seen + [mask, mask2]).mean()
    assert 0.45 < float(total) < 0.55


@pytest.mark.parametrize("mt", ["222222", "111111"])
@pytest.mark.parametrize("B,Cin,Cout,H", [(100, 128, 256, 8), (100, 256, 512, 4), (100, 64, 128, 16), (3, 32, 32, 8), (1, 64, 96, 8), (7, 32, 64, 4), (9, 96, 32, 4), (8, 64, 64, 4),
                                          (2, 32, 32, 4), (3, 32, 64, 16), (1, 96, 32, 16)])
def test_conv3x3_stride2_small_maps_match_aten(ops, monkeypatch, B, Cin, Cout, H, mt):
    """Conv2d(3x3, stride 2, padding 1) from 16x16 / 8x8 / 4x4 maps on ee_s2.hip (split-reduction MFMA kernel, backward-data by parity classes):
    forward and input gradient against a float64 convolution and against ATen, odd batch sizes (partly filled workgroups) included,
    with 32 and with 16 result channels per workgroup (EEADV_S2_MT); the rearranged filters follow in-place weight updates."""
    import torch.nn.functional as F
    from eeadv import functional as EF
    monkeypatch.setenv("EEADV_S2_MT", mt)
    g = torch.Generator(device="cpu").manual_seed(B + Cin + Cout + H)
    x = torch.randn(B, Cin, H, H, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(DEV).requires_grad_(True)
    dy = torch.randn(B, Cout, H // 2, H // 2, generator=g).to(DEV)
    for round_ in range(2):
        ref = F.conv2d(x, w, None, 2, 1)
        got = EF.Conv3x3S2SmallFn.apply(x, w)
        x64, w64 = x.detach().double().requires_grad_(True), w.detach().double()
        ref64 = F.conv2d(x64, w64, None, 2, 1)
        assert float((got.double() - ref64).abs().max()) < 1e-6 * float(ref64.abs().max())
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
        (gx, gw), (ex, ew) = torch.autograd.grad(got, [x, w], dy), torch.autograd.grad(ref, [x, w], dy)
        (e64,) = torch.autograd.grad(ref64, [x64], dy.double())
        assert float((gx.double() - e64).abs().max()) < 1e-6 * float(e64.abs().max())
        torch.testing.assert_close(gx, ex, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(gw, ew, rtol=1e-4, atol=1e-4 * (H * H * B) ** 0.5)
        with torch.no_grad():
            w.mul_(1.25)


@pytest.mark.parametrize("B,Hd,K,reduction", [(50, 1024, 10, "sum"), (7, 1024, 10, "mean"), (3, 100, 64, "sum"), (1, 33, 2, "mean"), (4, 1500, 10, "sum"), (5, 1000, 16, "mean")])
def test_fc_ce_grad_matches_autograd(ops, B, Hd, K, reduction):
    """fc2(relu(z1)) + CrossEntropyLoss + both backward steps in one launch (the tail of MNIST's Net_2 inside the attack loop) against
    float64 autograd; logits too; the ReLU gate at exact zeros and NaN as torch's threshold_backward has it."""
    import torch.nn.functional as F
    g = torch.Generator(device="cpu").manual_seed(B * Hd + K)
    z1 = torch.randn(B, Hd, generator=g)
    z1[0, :5] = 0.0  # relu'(0) = 0
    w2 = torch.randn(K, Hd, generator=g) / Hd ** 0.5
    b2 = torch.randn(K, generator=g)
    y = torch.randint(0, K, (B,), generator=g)
    z64 = z1.double().requires_grad_(True)
    loss = F.cross_entropy(F.linear(F.relu(z64), w2.double(), b2.double()), y, reduction=reduction)
    (want,) = torch.autograd.grad(loss, [z64])
    dz, lg = ops.fc_ce_grad(z1.to(DEV), w2.to(DEV), b2.to(DEV), y.to(DEV), reduction, want_logits=True)
    assert float((dz.cpu().double() - want).abs().max()) < 2e-6 * max(float(want.abs().max()), 1e-3)
    torch.testing.assert_close(lg.cpu().double(), F.linear(F.relu(z1.double()), w2.double(), b2.double()), rtol=1e-5, atol=1e-5)
    assert float(dz[0, :5].abs().max()) == 0.0
    dz2 = ops.fc_ce_grad(z1.to(DEV), w2.to(DEV), None, y.to(DEV), reduction)  # no bias
    loss = F.cross_entropy(F.linear(F.relu(z64), w2.double()), y, reduction=reduction)
    (want,) = torch.autograd.grad(loss, [z64])
    assert float((dz2.cpu().double() - want).abs().max()) < 2e-6 * max(float(want.abs().max()), 1e-3)
    zn = z1.clone()
    zn[0, 7] = float("nan")  # relu keeps the NaN (the row's logits and gradient are NaN); ATen's threshold_backward zeroes where z <= 0,
    # which a NaN is not: the gate of the NaN entry itself is OPEN and its gradient is the row's NaN
    dzn = ops.fc_ce_grad(zn.to(DEV), w2.to(DEV), b2.to(DEV), y.to(DEV), reduction).cpu()
    open_gate = int((zn[0, 8:] > 0).nonzero()[0]) + 8
    assert bool(torch.isnan(dzn[0, 7])) and bool(torch.isnan(dzn[0, open_gate])) and (B == 1 or bool(torch.isfinite(dzn[1:]).all()))
    closed = (zn[0] <= 0).nonzero().flatten()
    assert float(dzn[0, closed].abs().max()) == 0.0  # closed gates stay closed next to the NaN


@pytest.mark.parametrize("B,C,H,W,two", [(100, 64, 32, 32, True), (3, 64, 32, 32, False), (5, 16, 14, 20, True), (2, 8, 7, 12, False)])
def test_stem_batchnorm_backward_sums_from_the_pooled_tensors(ops, B, C, H, W, two):
    """ee_bn_relu_pool_fwd_xa_f32 / _bwd_xa_f32 (resnet.py:113-117 backwards, training mode): x at every window's argmax is what the forward
    recorded, and the backward's batch sums taken from the pooled gradient and that tensor alone give the dx / dgamma / dbeta of the form that
    reads the full-resolution map twice - to rounding (the sums run in another order), with a NaN input gradient staying where it was."""
    g = torch.Generator(device="cpu").manual_seed(B * C + H)
    x = torch.randn(B, C, H, W, generator=g).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    y, code, sm, si, xa = ops.bn_relu_pool_fwd(x, gamma, beta, rm.clone(), rv.clone(), 0.1, 1e-5, True, None, True)
    y0, code0, sm0, si0 = ops.bn_relu_pool_fwd(x, gamma, beta, rm.clone(), rv.clone(), 0.1, 1e-5, True)
    assert torch.equal(y, y0) and torch.equal(code, code0) and torch.equal(sm, sm0) and torch.equal(si, si0)
    # x_argmax is x at the position the code names
    OH, OW = y.shape[2], y.shape[3]
    oh, ow = torch.meshgrid(torch.arange(OH, device=DEV), torch.arange(OW, device=DEV), indexing="ij")
    hh = (2 * oh - 1)[None, None] + (code.long() // 3)
    ww = (2 * ow - 1)[None, None] + (code.long() % 3)
    assert torch.equal(xa, x.flatten(2).gather(2, (hh * W + ww).flatten(2)).view_as(xa))
    dyp = torch.randn(y.shape, generator=g).to(DEV)
    dyp2 = torch.randn(y.shape, generator=g).to(DEV) if two else None
    want = ops.bn_relu_pool_bwd(dyp, code, x, gamma, beta, sm, si, rm, rv, 1e-5, True, True, True, dyp2)
    got = ops.bn_relu_pool_bwd(dyp, code, x, gamma, beta, sm, si, rm, rv, 1e-5, True, True, True, dyp2, xa)
    for a_, b_, name in zip(got, want, ("dx", "dgamma", "dbeta")):
        scale = float(b_.abs().max())
        assert float((a_ - b_).abs().max()) <= 2e-5 * scale + 1e-6, name
    # eval mode takes no sums: the same bits with or without x_argmax
    e0 = ops.bn_relu_pool_bwd(dyp, code, x, gamma, beta, None, None, rm, rv, 1e-5, False, True, False, dyp2)
    e1 = ops.bn_relu_pool_bwd(dyp, code, x, gamma, beta, None, None, rm, rv, 1e-5, False, True, False, dyp2, xa)
    assert torch.equal(e0[0], e1[0])


@pytest.mark.parametrize("B,C,HW,K,reduction", [(100, 512, 4, 200, "sum"), (3, 2048, 49, 1000, "mean"), (1, 64, 1, 10, "sum"), (7, 300, 4, 33, "mean")])
def test_cross_entropy_gradient_inside_the_head_backward(ops, B, C, HW, K, reduction):
    """ee_ce_pool_linear_bwd_f32 (resnet.py:157-160 + attacks.py:23 / :255): the loss gradient formed inside the head's backward launch gives the
    bits of ee_ce_f32 followed by ee_pool_linear_bwd_f32; a NaN logit stays in its own image."""
    g = torch.Generator(device="cpu").manual_seed(B * C + K)
    side = int(HW ** 0.5)
    feat = torch.randn(B, C, side, side, generator=g).to(DEV)
    w = (torch.randn(K, C, generator=g) / C ** 0.5).to(DEV)
    bias = torch.randn(K, generator=g).to(DEV)
    y = torch.randint(0, K, (B,), generator=g).to(DEV)
    logits, _ = ops.pool_linear_fwd(feat, w, bias)
    _, d = ops.ce(logits, y, reduction, 0.0, False, True)
    want = ops.pool_linear_bwd(d, w, tuple(feat.shape))
    got = ops.ce_pool_linear_bwd(logits, y, w, tuple(feat.shape), reduction)
    assert torch.equal(got, want)
    if B > 1:
        ln = logits.clone()
        ln[0, 0] = float("nan")
        gn = ops.ce_pool_linear_bwd(ln, y, w, tuple(feat.shape), reduction)
        assert bool(torch.isnan(gn[0]).all()) and torch.equal(gn[1:], want[1:])


def test_resnet_attack_gradient_with_the_loss_gradient_inside_the_head(ops, monkeypatch):
    """engine's input gradient of ResNet-18 (64 x 64) with models.ResNet.head_grad on (two head launches) and off (three): the same bits"""
    from eeadv import engine, models
    torch.manual_seed(4)
    net = models.make_resnet(18, "tiny").to(DEV).eval()
    x = torch.rand(6, 3, 64, 64, device=DEV)
    y = torch.randint(0, 200, (6,), device=DEV)
    outs = []
    for on in (True, False):
        monkeypatch.setattr(models, "_HEAD_CE", on)
        for kind in (engine.CE_SUM, engine.CE_MEAN):
            xi = x.clone().requires_grad_(True)
            outs.append(engine.input_gradient(net, xi, engine.LossSpec(kind, y)).clone())
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])
    assert float(outs[0].abs().max()) > 0


@pytest.mark.parametrize("co,ci", [(64, 32), (32, 96), (128, 128)])
def test_weight_preparation_kernels_match_their_torch_restatement(ops, monkeypatch, co, ci):
    """ee_wprep.hip (one launch per weight and kind) against functional._rearranged's torch expressions: the permutation kinds bit for bit,
    the Winograd filter transforms within rounding (the einsum sums in another order) and against float64; and the cache rebuilds in place."""
    from eeadv import functional as EF
    g = torch.Generator(device="cpu").manual_seed(co + ci)
    w = torch.randn(co, ci, 3, 3, generator=g).to(DEV)
    w1 = torch.randn(co, ci, 1, 1, generator=g).to(DEV)
    for kind in ("s2m_f", "s2m_b", "s2p_f", "s2p_b", "s1"):
        extra = w1 if kind.startswith("s2p") else None
        want = EF._rearranged(w, kind, extra).contiguous()
        got = torch.empty(EF._rearranged_shape(w, kind), device=DEV)
        ops.conv_weight_prep(EF._NATIVE_KIND[kind], w, extra, got)
        assert got.shape == want.shape and torch.equal(got, want), kind
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=DEV)
    for kind in ("wino_f", "wino_b"):
        got = torch.empty(EF._rearranged_shape(w, kind), device=DEV)
        ops.conv_weight_prep(EF._NATIVE_KIND[kind], w, None, got)
        torch.testing.assert_close(got, EF._rearranged(w, kind).contiguous(), rtol=1e-6, atol=1e-6)
        w64 = w.double() if kind == "wino_f" else w.double().flip(2, 3)
        want = torch.einsum("ia,jb,rkab->ijkr" if kind == "wino_f" else "ia,jb,krab->ijkr", G, G, w64).reshape(got.shape)
        assert float((got.double() - want).abs().max()) < 1e-6 * float(want.abs().max())
    if co % 32 == 0 and ci % 32 == 0:  # both sets in one launch (what Conv3x3WinoFn's cache uses): the per-set launches' bits
        both = torch.empty(EF._rearranged_shape(w, "wino_fb"), device=DEV)
        ops.conv_weight_prep(EF._NATIVE_KIND["wino_fb"], w, None, both)
        for half, kind in enumerate(("wino_f", "wino_b")):
            one = torch.empty(EF._rearranged_shape(w, kind), device=DEV)
            ops.conv_weight_prep(EF._NATIVE_KIND[kind], w, None, one)
            assert torch.equal(both[half].view(one.shape), one), kind
        pf = torch.nn.Parameter(w.clone())
        uf, ub = EF.wino_sets(pf)
        assert uf.data_ptr() + 4 * uf.numel() == ub.data_ptr() and torch.equal(uf, both[0].view(uf.shape))
    p = torch.nn.Parameter(w.clone())
    buf = EF._dense_weight(p, "wino_f")
    ptr = buf.data_ptr()
    with torch.no_grad():
        p.mul_(2.0)
    again = EF._dense_weight(p, "wino_f")
    assert again.data_ptr() == ptr
    torch.testing.assert_close(again, EF._rearranged(p, "wino_f").contiguous(), rtol=1e-6, atol=1e-6)
    # the in-place rebuild of every cached item of a model (what a captured optimiser step ends with) gives the per-item launches' bits
    m = torch.nn.Module()
    m.a, m.b = torch.nn.Parameter(w.clone()), torch.nn.Parameter(w1.clone())
    kinds = ("wino_f", "wino_b", "s2p_f", "s2p_b", "s2m_f", "s1") + (("wino_fb",) if co % 32 == 0 and ci % 32 == 0 else ())
    bufs = {k: EF._dense_weight(m.a, k, m.b if k.startswith("s2p") else None) for k in kinds}  # (round 4: the rebuild below is ONE batched launch)
    with torch.no_grad():
        m.a.mul_(0.5)
        m.b.add_(1.0)
    assert set(EF.rebuild_dense_weights(m)) == {(id(m.a), k) for k in bufs}
    for k, bufk in bufs.items():
        want = torch.empty_like(bufk)
        ops.conv_weight_prep(EF._NATIVE_KIND[k], m.a.detach(), m.b.detach() if k.startswith("s2p") else None, want)
        assert torch.equal(bufk, want), k


@pytest.mark.parametrize("H", [4, 8, 16])
@pytest.mark.parametrize("KC", [16, 48])
def test_mfma_convs_with_an_odd_number_of_rounds(ops, H, KC):
    """The producer / consumer kernels run their rounds in pairs (two register sets, two LDS buffers): channel counts that give an ODD
    number of 16-channel rounds - below what the models use, but what the C ABI admits - through ops directly: Winograd forward with 16 / 48
    reduction channels, the stride-2 kernel forward (Cin = KC) and backward-data (Cout = KC), against float64."""
    import torch.nn.functional as F
    from eeadv import functional as EF
    g = torch.Generator(device="cpu").manual_seed(H + KC)
    B = 5
    x = torch.randn(B, KC, H, H, generator=g).to(DEV)
    w = (torch.randn(32, KC, 3, 3, generator=g) / (3 * KC ** 0.5)).to(DEV)
    got = ops.wino3x3(x, EF._rearranged(w, "wino_f").contiguous())
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    assert float((got.double() - ref).abs().max()) < 2e-6 * float(ref.abs().max())
    got = ops.conv3x3s2_small_fwd(x, EF._rearranged(w, "s2m_f").contiguous(), 32)
    ref = F.conv2d(x.double(), w.double(), None, 2, 1)
    assert float((got.double() - ref).abs().max()) < 1e-6 * float(ref.abs().max())
    wb = (torch.randn(KC, 32, 3, 3, generator=g) / (3 * 32 ** 0.5)).to(DEV)  # Cout = KC reduction channels, Cin = 32
    dy = torch.randn(B, KC, H // 2, H // 2, generator=g).to(DEV)
    got = ops.conv3x3s2_small_bwd_data(dy, EF._rearranged(wb, "s2m_b").contiguous(), 32)
    xr = torch.zeros(B, 32, H, H, dtype=torch.float64, device=DEV, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(xr, wb.double(), None, 2, 1), [xr], dy.double())
    assert float((got.double() - ref).abs().max()) < 1e-6 * float(ref.abs().max())


@pytest.mark.parametrize("mt", ["222222", "111111"])
@pytest.mark.parametrize("B,Cin,Cout,H", [(100, 64, 128, 16), (100, 128, 256, 8), (100, 256, 512, 4), (3, 32, 64, 16), (5, 64, 32, 8), (9, 32, 32, 4), (1, 96, 64, 4)])
def test_conv3x3_stride2_with_shortcut_conv_matches_aten(ops, monkeypatch, B, Cin, Cout, H, mt):
    """A down-sampling BasicBlock's conv1 (3x3, stride 2) and shortcut convolution (1x1, stride 2) of the same input in one launch each way
    (functional.Conv3x3S2PairFn): both outputs, the summed input gradient and both weight gradients against float64 / ATen; the
    rearranged filter pair follows in-place updates of either weight."""
    import torch.nn.functional as F
    from eeadv import functional as EF
    monkeypatch.setenv("EEADV_S2_MT", mt)
    g = torch.Generator(device="cpu").manual_seed(B + Cin + Cout + H)
    x = torch.randn(B, Cin, H, H, generator=g).to(DEV).requires_grad_(True)
    w3 = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(DEV).requires_grad_(True)
    w1 = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).to(DEV).requires_grad_(True)
    dy3 = torch.randn(B, Cout, H // 2, H // 2, generator=g).to(DEV)
    dy1 = torch.randn(B, Cout, H // 2, H // 2, generator=g).to(DEV)
    for round_ in range(3):
        r3, r1 = F.conv2d(x, w3, None, 2, 1), F.conv2d(x, w1, None, 2, 0)
        g3, g1 = EF.Conv3x3S2PairFn.apply(x, w3, w1)
        x64 = x.detach().double().requires_grad_(True)
        q3, q1 = F.conv2d(x64, w3.detach().double(), None, 2, 1), F.conv2d(x64, w1.detach().double(), None, 2, 0)
        assert float((g3.detach().double() - q3).abs().max()) < 1e-6 * float(q3.abs().max())
        assert float((g1.detach().double() - q1).abs().max()) < 1e-6 * float(q1.abs().max())
        torch.testing.assert_close(g3, r3, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(g1, r1, rtol=1e-4, atol=1e-4)
        got = torch.autograd.grad([g3, g1], [x, w3, w1], [dy3, dy1])
        exp = torch.autograd.grad([r3, r1], [x, w3, w1], [dy3, dy1])
        (e64,) = torch.autograd.grad([q3, q1], [x64], [dy3.double(), dy1.double()])
        assert float((got[0].double() - e64).abs().max()) < 1e-6 * float(e64.abs().max())
        torch.testing.assert_close(got[0], exp[0], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(got[1], exp[1], rtol=1e-4, atol=1e-4 * (H * H * B) ** 0.5)
        torch.testing.assert_close(got[2], exp[2], rtol=1e-4, atol=1e-4 * (H * H * B) ** 0.5)
        with torch.no_grad():
            (w3 if round_ == 0 else w1).mul_(1.25)


@pytest.mark.parametrize("B,Cin,Cout,H", [(100, 128, 128, 8), (3, 32, 32, 8), (2, 64, 96, 8), (1, 128, 64, 8), (100, 64, 64, 16), (3, 32, 96, 16), (100, 256, 256, 4), (7, 32, 64, 4), (1, 64, 32, 4)])
def test_conv3x3_winograd_on_8x8_maps_matches_aten(ops, B, Cin, Cout, H):
    """Conv2d(3x3, stride 1, padding 1) on 8x8 maps as Winograd F(2x2, 3x3) on the f32 matrix cores (ee_wino.hip): forward and input
    gradient against float64 direct convolution (error of the order of MIOpen's own Winograd solver) and against ATen; the transformed
    filters follow in-place weight updates."""
    import torch.nn.functional as F
    from eeadv import functional as EF
    g = torch.Generator(device="cpu").manual_seed(B + Cin + Cout)
    x = torch.randn(B, Cin, H, H, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(DEV).requires_grad_(True)
    dy = torch.randn(B, Cout, H, H, generator=g).to(DEV)
    for round_ in range(2):
        ref = F.conv2d(x, w, None, 1, 1)
        got = EF.Conv3x3WinoFn.apply(x, w)
        x64, w64 = x.detach().double().requires_grad_(True), w.detach().double()
        ref64 = F.conv2d(x64, w64, None, 1, 1)
        scale = float(ref64.abs().max())
        assert float((got.double() - ref64).abs().max()) < 2e-6 * scale
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
        (gx, gw), (ex, ew) = torch.autograd.grad(got, [x, w], dy), torch.autograd.grad(ref, [x, w], dy)
        (e64,) = torch.autograd.grad(ref64, [x64], dy.double())
        assert float((gx.double() - e64).abs().max()) < 2e-6 * float(e64.abs().max())
        torch.testing.assert_close(gx, ex, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(gw, ew, rtol=1e-4, atol=1e-4 * (H * H * B) ** 0.5)
        with torch.no_grad():
            w.mul_(1.25)


@pytest.mark.parametrize("B,Cin,Cout,H", [(100, 64, 64, 16), (100, 128, 128, 8), (100, 256, 256, 4), (100, 512, 512, 2), (1, 32, 32, 16), (3, 32, 96, 16),
                                          (7, 64, 32, 8), (5, 32, 64, 4), (37, 96, 32, 4), (17, 64, 96, 2), (1, 32, 32, 2), (33, 32, 32, 2)])
def test_conv3x3_weight_gradient_winograd_matches_float64(ops, B, Cin, Cout, H):
    """d loss / d weight of Conv2d(3x3, stride 1, padding 1) (`loss.backward()`, experiments_tinyimagenet.py:304-306; resnet.py:26-31) as Winograd
    F(3x3, 2x2) on the f32 matrix cores with the reduction over images split over workgroups (ee_wrw.hip): against float64 direct
    correlation (2e-6 of the largest entry per sqrt of the reduction length ... in practice below MIOpen's own error), against ATen, and
    bit-identical from call to call (partial sums are added in a fixed order; MIOpen's solvers use atomics)."""
    g = torch.Generator(device="cpu").manual_seed(B + Cin + Cout + H)
    x = torch.randn(B, Cin, H, H, generator=g).to(DEV)
    dy = torch.randn(B, Cout, H, H, generator=g).to(DEV)
    assert ops.wrw3x3_supported(x, dy)
    got = ops.wrw3x3(x, dy)
    w = torch.zeros(Cout, Cin, 3, 3, device=DEV)
    ref = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    ref64 = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    scale = float(ref64.abs().max())
    err, err_aten = float((got.double() - ref64).abs().max()), float((ref.double() - ref64).abs().max())
    assert err < 3e-6 * scale, (err / scale, err_aten / scale)
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4 * (H * H * B) ** 0.5)
    again = ops.wrw3x3(x, dy)
    assert torch.equal(got, again)
    # structured inputs: a single bright pixel and a single gradient pixel give exactly one tap (exact in fp32: products with 0, 1, .5, .25)
    x.zero_(), dy.zero_()
    b, ci, co = B - 1, Cin - 1, Cout // 2
    x[b, ci, H - 1, 0] = 1.0
    dy[b, co, H - 2, 1] = 1.0  # output (H-2, 1) sees input (H-1, 0) through tap (2, 0)
    one = ops.wrw3x3(x, dy)
    exp = torch.zeros_like(one)
    exp[co, ci, 2, 0] = 1.0
    assert torch.equal(one, exp)


@pytest.mark.parametrize("B,Cin,Cout,H", [(100, 64, 128, 16), (100, 128, 256, 8), (100, 256, 512, 4), (1, 32, 32, 16), (3, 32, 64, 16), (7, 64, 32, 8),
                                          (5, 32, 96, 4), (37, 96, 32, 4), (2, 32, 32, 8)])
@pytest.mark.parametrize("pair", [True, False])
def test_conv3x3s2_weight_gradient_matches_float64(ops, B, Cin, Cout, H, pair):
    """Weight gradients of a down-sampling block's Conv2d(3x3, stride 2, padding 1) and of its shortcut Conv2d(1x1, stride 2) of the same input
    (resnet.py:50-59, :132-142) in one launch of ee_wrw.hip: against float64, against ATen, bit-identical from call to call, and exact on a
    one-pixel input."""
    g = torch.Generator(device="cpu").manual_seed(B + Cin + Cout + H)
    x = torch.randn(B, Cin, H, H, generator=g).to(DEV)
    dy3 = torch.randn(B, Cout, H // 2, H // 2, generator=g).to(DEV)
    dy1 = torch.randn(B, Cout, H // 2, H // 2, generator=g).to(DEV) if pair else None
    assert ops.wrw3x3s2_supported(x, dy3, dy1)
    dw3, dw1 = ops.wrw3x3s2(x, dy3, dy1)
    assert (dw1 is None) == (not pair)
    w3, w1 = torch.zeros(Cout, Cin, 3, 3, device=DEV), torch.zeros(Cout, Cin, 1, 1, device=DEV)

    def ref(dy, w, pad, dt):
        return torch.ops.aten.convolution_backward(dy.to(dt), x.to(dt), w.to(dt), None, [2, 2], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    for got, dy, w, pad in ((dw3, dy3, w3, 1),) + (((dw1, dy1, w1, 0),) if pair else ()):
        r64 = ref(dy, w, pad, torch.float64)
        assert float((got.double() - r64).abs().max()) < 3e-6 * float(r64.abs().max())
        torch.testing.assert_close(got, ref(dy, w, pad, torch.float32), rtol=1e-4, atol=1e-4 * (H * H * B) ** 0.5)
    a3, a1 = ops.wrw3x3s2(x, dy3, dy1)
    assert torch.equal(a3, dw3) and (not pair or torch.equal(a1, dw1))
    x.zero_(), dy3.zero_()
    b, ci, co = B - 1, Cin - 1, Cout // 2
    x[b, ci, H - 1, 0] = 1.0
    dy3[b, co, H // 2 - 1, 0] = 1.0  # output (H/2-1, 0) sees input (H-1, 0) = (2 oh + 1, 2 ow + 0) through tap (2, 1)
    if pair:
        dy1.zero_()
        x[b, 0, 2, 2] = 1.0
        dy1[b, 1, 1, 1] = 1.0  # the shortcut's pixel (1, 1) is input (2, 2)
    o3, o1 = ops.wrw3x3s2(x, dy3, dy1)
    e3 = torch.zeros_like(o3)
    e3[co, ci, 2, 1] = 1.0
    assert torch.equal(o3, e3)
    if pair:
        e1 = torch.zeros_like(o1)
        e1[1, 0, 0, 0] = 1.0
        assert torch.equal(o1, e1)


@pytest.mark.parametrize("B,H,W", [(100, 64, 64), (3, 64, 64), (1, 32, 32), (5, 34, 96), (2, 224, 224)])
def test_stem_weight_gradient_matches_float64(ops, B, H, W):
    """d loss / d weight of the stem Conv2d(3, 64, 7, stride 2, padding 3) (resnet.py:112) on ee_wrw.hip: against float64 and ATen, bit-identical
    from call to call, exact on a one-pixel input at the image border (the zero padding)."""
    g = torch.Generator(device="cpu").manual_seed(B + H + W)
    x = torch.randn(B, 3, H, W, generator=g).to(DEV)
    dy = torch.randn(B, 64, H // 2, W // 2, generator=g).to(DEV)
    assert ops.wrw_stem7x7s2_supported(x, dy)
    got = ops.wrw_stem7x7s2(x, dy)
    w = torch.zeros(64, 3, 7, 7, device=DEV)
    ref = lambda dt: torch.ops.aten.convolution_backward(dy.to(dt), x.to(dt), w.to(dt), None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    r64 = ref(torch.float64)
    assert float((got.double() - r64).abs().max()) < 3e-6 * float(r64.abs().max())
    torch.testing.assert_close(got, ref(torch.float32), rtol=1e-4, atol=1e-4 * (H * W * B) ** 0.5)
    assert torch.equal(got, ops.wrw_stem7x7s2(x, dy))
    x.zero_(), dy.zero_()
    x[B - 1, 2, H - 1, 0] = 1.0
    dy[B - 1, 5, H // 2 - 1, 1] = 1.0  # output (H/2-1, 1) sees input (H-1, 0) = (2 oh - 3 + kh, 2 ow - 3 + kw) through tap (4, 1)
    exp = torch.zeros_like(got)
    exp[5, 2, 4, 1] = 1.0
    assert torch.equal(ops.wrw_stem7x7s2(x, dy), exp)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(32, 64, 256, 56, 56), (32, 256, 64, 56, 56), (32, 512, 128, 28, 28), (32, 256, 1024, 14, 14), (3, 64, 64, 6, 6),
                                            (1, 128, 64, 2, 2), (5, 64, 192, 10, 14), (2, 64, 64, 30, 34)])
def test_conv1x1_weight_gradient_matches_float64(ops, B, Cin, Cout, H, W):
    """d loss / d weight of the bottleneck blocks' Conv2d(1x1, stride 1) (resnet.py:75-100) as an NCHW product on the f32 matrix cores (ee_wrw.hip):
    against float64 (reduction lengths up to 100 352), against ATen, bit-identical from call to call, exact on a one-pixel input."""
    g = torch.Generator(device="cpu").manual_seed(B + Cin + Cout + H * W)
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    dy = torch.randn(B, Cout, H, W, generator=g).to(DEV)
    assert ops.wrw1x1_supported(x, dy)
    got = ops.wrw1x1(x, dy)
    r64 = torch.einsum("bohw,bihw->oi", dy.double(), x.double())
    assert float((got[:, :, 0, 0].double() - r64).abs().max()) < 3e-6 * float(r64.abs().max())
    w = torch.zeros(Cout, Cin, 1, 1, device=DEV)
    ref = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4 * (H * W * B) ** 0.5)
    assert torch.equal(got, ops.wrw1x1(x, dy))
    x.zero_(), dy.zero_()
    x[B - 1, Cin - 1, H - 1, W - 1] = 1.0
    dy[B - 1, 3, H - 1, W - 1] = 1.0
    exp = torch.zeros_like(got)
    exp[3, Cin - 1, 0, 0] = 1.0
    assert torch.equal(ops.wrw1x1(x, dy), exp)
