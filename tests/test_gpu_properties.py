"""Full-size runs of the hot path (BASELINE.json's per-process shapes and beyond) checked through size-independent
properties and strided samples against the oracle - the oracle itself is only asked for what it finishes in seconds.
All through the C ABI (eeadv.ops)."""
import numpy as np
import pytest
import torch

from oracle import ee_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from eeadv import ops as _ops
    return _ops


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("shape", [(1600, 3, 64, 64), (256, 3, 224, 224), (5000, 1, 28, 28)])
def test_pgd_step_full_size_stays_in_the_ball_and_matches_oracle_on_samples(ops, shape):
    """attacks.py:25-27 at the Tiny / ImageNet / MNIST per-process batch sizes (x16, x1, x100): every element inside
    [x0-eps, x0+eps] and [0,1], untouched where the gradient is 0 or NaN and x already feasible, and bit-equal to the oracle
    on a strided sample of images."""
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    eps, alpha = 16 / 255, 2 / 255
    x0 = torch.rand(shape, device=DEV, generator=g)
    x = (x0 + (torch.rand(shape, device=DEV, generator=g) * 2 - 1) * eps).clamp_(0, 1)
    grad = torch.randn(shape, device=DEV, generator=g)
    grad[::7] = 0.0
    grad[3::11, :, ::5] = float("nan")
    before = x.clone()
    ops.pgd_step_(x, grad, x0, alpha, eps)
    assert bool(((x >= 0) & (x <= 1)).all())
    assert bool(((x - x0).abs() <= eps + 1e-7).all())
    still = (grad == 0) | torch.isnan(grad)
    assert torch.equal(x[still], before[still])  # sign(0) = sign(NaN) = 0 and the old x was feasible
    moved = (x - before).abs()
    assert float(moved.max()) <= alpha + 1e-7
    for i in range(0, shape[0], max(1, shape[0] // 5)):
        want = O.pgd_step(before[i:i + 1].cpu().numpy(), grad[i:i + 1].cpu().numpy(), x0[i:i + 1].cpu().numpy(), alpha, eps)
        assert np.array_equal(bits(x[i:i + 1].cpu().numpy()), bits(want)), i


@pytest.mark.parametrize("shape", [(1600, 3, 64, 64), (256, 3, 224, 224), (5000, 1, 28, 28)])
def test_front_end_full_size_properties(ops, shape):
    """resnet_EE.py:176-191 at full size: x_in in [0,1]; the gate is exactly 1[0 <= x_hfs + w*e <= 1]; the edge map is
    binary; backward: g_hfs = gate * g_in exactly, the edge gradient is NaN only where the forward magnitude is 0 inside
    the 5x5 footprint; and sampled images are bit-equal to the oracle (forward) in both directions."""
    g = torch.Generator(device=DEV).manual_seed(sum(shape) + 1)
    B, C, H, W = shape
    x = torch.rand(shape, device=DEV, generator=g)
    x_hfs = torch.rand(shape, device=DEV, generator=g) * 1.2 - 0.1
    wts = ops.EdgeWeights(1.0)
    alpha, high, w = 0.0, 0.2, 0.5  # uniform noise: magnitudes around 0.2, so the edge map is mixed
    x_in, gate, edge = ops.frontend_fwd(x, x_hfs, wts, alpha, high, w, want_edge=True)
    assert bool(((x_in >= 0) & (x_in <= 1)).all())
    assert bool(((edge == 0) | (edge == 1)).all())
    assert 0.05 < float(edge.mean()) < 0.95
    s = x_hfs + w * edge
    assert torch.equal(gate.bool(), (s >= 0) & (s <= 1))
    assert torch.equal(x_in, s.clamp(0, 1))
    g_in = torch.randn(shape, device=DEV, generator=g)
    g_hfs, g_edge = ops.frontend_bwd(g_in, gate, x, wts, alpha, high, w)
    assert torch.equal(g_hfs, torch.where(gate.bool(), g_in, torch.zeros_like(g_in)))
    assert g_edge.shape == (B, 1, H, W)
    for i in range(0, B, max(1, B // 4)):
        xi, hi_ = x[i:i + 1].cpu().numpy(), x_hfs[i:i + 1].cpu().numpy()
        oe = O.edge125_fwd(xi, alpha, high)
        assert np.array_equal(bits(edge[i:i + 1].cpu().numpy()), bits(oe)), i
        u = (w * (g_in[i:i + 1] * gate[i:i + 1].float()).sum(1, keepdim=True)).cpu().numpy()  # sum over channels, then * w
        og = O.edge125_bwd(xi, u, alpha, high)
        got = g_edge[i:i + 1].cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(og)), i
        fin = ~np.isnan(og)
        np.testing.assert_allclose(got[fin], og[fin], rtol=0, atol=2e-6)  # channel-sum order of u differs by an ulp


@pytest.mark.parametrize("B,size,r", [(1600, 64, 16), (5000, 28, 8)])
def test_hfs_full_size_is_linear_self_adjoint_and_mean_preserving(ops, B, size, r):
    """core.py:15-55 as an operator at 16x / 100x the reference batch: linear, self-adjoint, the identity on its symmetric pass band, and it keeps every plane's mean (the DC bin is inside the mask)."""
    from eeadv.hfs import HFSOperator
    C = 3 if size == 64 else 1
    g = torch.Generator(device=DEV).manual_seed(B + size)
    op = HFSOperator(size, size, r, torch.device(DEV))
    x = torch.rand(B, C, size, size, device=DEV, generator=g)
    z = torch.rand(B, C, size, size, device=DEV, generator=g)
    fx, fz = op.forward(x), op.forward(z)
    torch.testing.assert_close(op.forward(2.0 * x - 0.5 * z), 2.0 * fx - 0.5 * fz, rtol=0, atol=2e-5)
    # not a projection in general: irfft of the masked half spectrum symmetrises the unpaired -r row / column of the mask
    # (DESIGN.md section 2), so F(F(x)) = F(x) only on inputs inside the symmetric pass band
    lowpass = torch.cos(2 * np.pi * 3 * torch.arange(size, device=DEV) / size)[None, :] * torch.ones(size, 1, device=DEV)
    torch.testing.assert_close(op.forward(lowpass.expand(2, C, size, size).contiguous())[0, 0], lowpass, rtol=0, atol=1e-5)
    torch.testing.assert_close(fx.mean((-1, -2)), x.mean((-1, -2)), rtol=0, atol=2e-6)
    lhs, rhs = (fx * z).sum((-1, -2)), (x * fz).sum((-1, -2))  # <F x, z> = <x, F z>
    torch.testing.assert_close(lhs, rhs, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(op.adjoint(z), fz, rtol=0, atol=1e-6)


def test_losses_and_topk_full_size(ops):
    """attacks.py:23 / :412 / helper.py:39-55 at B = 4096, K = 1000 (ImageNet classes): CE and KL against torch's own fp32
    ops within 1e-5 relative, gradients within 1e-6, top-k hit counts equal to torch.topk's."""
    import torch.nn.functional as F
    g = torch.Generator(device=DEV).manual_seed(9)
    B, K = 4096, 1000
    z = torch.randn(B, K, device=DEV, generator=g) * 3
    zp = torch.randn(B, K, device=DEV, generator=g) * 3
    y = torch.randint(0, K, (B,), device=DEV, generator=g)
    loss, d = ops.ce(z, y, "sum", 0.0, True, True)
    zr = z.clone().requires_grad_(True)
    ref = F.cross_entropy(zr, y, reduction="sum")
    ref.backward()
    np.testing.assert_allclose(float(loss), float(ref.detach()), rtol=1e-5)
    torch.testing.assert_close(d, zr.grad, rtol=0, atol=1e-6)
    kl, dq, _ = ops.kl_batchmean(z, zp, True, True, False)
    zq = z.clone().requires_grad_(True)
    kref = F.kl_div(F.log_softmax(zq, 1), F.softmax(zp, 1), reduction="batchmean")
    kref.backward()
    np.testing.assert_allclose(float(kl), float(kref.detach()), rtol=1e-5)
    torch.testing.assert_close(dq, zq.grad, rtol=0, atol=1e-7)
    idx, correct = ops.topk(z, y, 5)
    tk = z.topk(5, 1).indices
    assert torch.equal(idx, tk)
    hits = (tk == y[:, None]).cumsum(1).clamp(max=1).sum(0)
    assert correct.tolist() == hits.tolist()


def test_empty_batches_are_no_ops(ops):
    """B = 0 through the ABI: nothing is launched, shapes come back right."""
    wts = ops.EdgeWeights(1.0)
    x = torch.empty(0, 3, 64, 64, device=DEV)
    assert ops.edge125_fwd(x, wts, 0.0, 0.4).shape == (0, 1, 64, 64)
    x_in, gate, _ = ops.frontend_fwd(x, x, wts, 0.0, 0.4, 0.5)
    assert x_in.shape == x.shape and gate.shape == x.shape
    ops.pgd_step_(x, x, x, 0.01, 0.1)
    z = torch.empty(0, 10, device=DEV)
    loss, d = ops.ce(z, torch.empty(0, dtype=torch.int64, device=DEV), "sum", 0.0, True, True)
    assert float(loss) == 0.0 and d.shape == (0, 10)
    kl, dq, _ = ops.kl_batchmean(z, z, True, True, False)
    assert dq.shape == (0, 10) and bool(torch.isnan(kl))  # torch: the batch mean of nothing is NaN
    idx, correct = ops.topk(z, torch.empty(0, dtype=torch.int64, device=DEV), 3)
    assert idx.shape == (0, 3) and correct.tolist() == [0, 0, 0]
    from eeadv.hfs import HFSOperator
    assert HFSOperator(64, 64, 16, torch.device(DEV)).forward(x).shape == x.shape
    from eeadv.functional import BnActFn, MaxPool3s2Fn, PoolLinearFn
    f = torch.empty(0, 8, 4, 4, device=DEV)
    assert MaxPool3s2Fn.apply(f).shape == (0, 8, 2, 2)
    assert PoolLinearFn.apply(f, torch.ones(5, 8, device=DEV), None).shape == (0, 5)
