"""GPU parity of the fused front-end kernels (csrc/ee_chain.hip: one workgroup per image, two launches per PGD iteration)
against the separate kernels they replace (ee_square_draw / ee_hfs / ee_frontend_fwd_save | ee_frontend_bwd_saved / ee_hfs /
ee_pgd_step_bcast), which are themselves pinned to the oracle in test_gpu_kernels.py / test_gpu_path.py, and against the C
oracle directly.

Bit-exact: edge map, saved Sobel responses, derivative code of Add_Square, the draws, the update where the gradient sign is
decided, NaN handling.  Tolerance (2e-6): the low-pass values - the MFMA chain sums in a different order than ee_hfs.hip; both
are ~3e-7 from the float64 operator.
"""
import numpy as np
import pytest
import torch

from oracle import ee_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(3, 64, 8), (1, 28, 4), (3, 32, 4), (1, 64, 8)]  # (C, H = W, r)


@pytest.fixture(scope="module")
def ops():
    from eeadv import ops as _ops
    return _ops


def _setup(C, n, r, B, seed):
    from eeadv import hfs as HF
    torch.manual_seed(seed)
    x = torch.rand(B, C, n, n, device=DEV)
    x[0, :, 3:12, 5:20] = 0.5                      # flat patch: zero edge magnitude -> NaN edge gradients
    x[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 0.02, 0.98], device=DEV)
    op = HF.HFSOperator(n, n, r, DEV)
    assert op.chain is not None and op.kernel is not None
    return x, op


def _draws(ops, B, C, n, s, seed=5):
    state = torch.tensor([seed, 0, 0, 0], dtype=torch.int64, device=DEV)
    sizes = torch.tensor([s], dtype=torch.int32, device=DEV)
    stripe, pos, sign = ops.square_draw(B, C, n, sizes, state)
    return {"stripe": stripe, "sq_pos": pos, "sq_sign": sign, "sq_size": sizes}


@pytest.mark.parametrize("C,n,r", SHAPES)
def test_lowpass_mfma_chain_vs_float64_operator(ops, C, n, r):
    """w = 0, no Add_Square: x_in = clamp(hfs(x)): the four chained MFMA products against the dense operator in float64."""
    from eeadv import hfs as HF
    x, op = _setup(C, n, r, 5, 1)
    wts = ops.EdgeWeights(1.0)
    x_in, gate, gx, gy, edge = ops.chain_fwd(x, op.chain, wts, 0.0, 76 / 255, 0.0, want_edge=True)
    Ar, Ai, B1, B2 = HF.hfs_matrices(n, n, r)
    xd = x.cpu().double().numpy()
    want = np.einsum("hk,bckw->bchw", Ar, xd @ B1) + np.einsum("hk,bckw->bchw", Ai, xd @ B2)
    got = x_in.cpu().numpy()
    inside = (want > 1e-5) & (want < 1 - 1e-5)
    assert np.abs(got - np.clip(want, 0, 1)).max() < 2e-6
    assert np.array_equal((gate.cpu().numpy() & 1)[inside], np.ones(inside.sum(), np.uint8))
    assert np.array_equal(gate.cpu().numpy() >> 1, np.full(gate.shape, 3, np.uint8))  # no Add_Square: derivative 1
    np.testing.assert_allclose(got, op.forward(x).clamp(0, 1).cpu().numpy(), atol=2e-6)  # and against ee_hfs_f32
    oe, omag, ogx, ogy = O.edge125_fwd(x.cpu().numpy(), 0.0, 76 / 255, want_internals=True)
    assert np.array_equal(edge.cpu().numpy(), oe)
    assert np.array_equal(gx.cpu().numpy(), ogx) and np.array_equal(gy.cpu().numpy(), ogy)


@pytest.mark.parametrize("C,n,r", SHAPES)
@pytest.mark.parametrize("B", [1, 7, 100])
def test_chain_forward_vs_separate_kernels(ops, C, n, r, B):
    x, op = _setup(C, n, r, B, 2)
    eps = 16 / 255 if C == 3 else 0.3
    alpha, high, w = (0.0, 76 / 255, 1.0) if C == 3 else (0.3, 51 / 255, 1.0)
    s = max(int(round((0.8 * n * n) ** 0.5)), 1)
    d = _draws(ops, B, C, n, s)
    wts = ops.EdgeWeights(1.0)
    x_lp = op.forward_square(x, eps, d)
    ref_in, ref_gate, ref_edge, ref_gx, ref_gy = ops.frontend_fwd_save(x, x_lp, wts, alpha, high, w, want_edge=True)
    x_in, gate, gx, gy, edge = ops.chain_fwd(x, op.chain, wts, alpha, high, w, True, eps, s, None, d, want_edge=True)
    assert torch.equal(edge, ref_edge) and torch.equal(gx, ref_gx) and torch.equal(gy, ref_gy)  # the edge filter: bit for bit
    np.testing.assert_allclose(x_in.cpu().numpy(), ref_in.cpu().numpy(), atol=2e-6)
    ssum = (x_lp + w * ref_edge).cpu().numpy()
    decided = (np.abs(ssum) > 2e-6) & (np.abs(ssum - 1) > 2e-6)  # the clamp gate can only differ where the sum sits on a bound
    g0, r0 = gate.cpu().numpy() & 1, ref_gate.cpu().numpy()
    assert np.array_equal(g0[decided], r0[decided]) and decided.mean() > 0.5
    # bits 1-2: d add_square / dx, the value ee_add_square_bwd_f32 multiplies by
    mask = ops.add_square_bwd(torch.ones_like(x), x, eps, d["stripe"], d["sq_sign"], d["sq_pos"], d["sq_size"]).cpu().numpy()
    code = gate.cpu().numpy() >> 1
    assert set(np.unique(mask).tolist()) <= {0.0, 0.5, 0.75, 1.0}
    assert np.array_equal(np.where(code == 0, 0.0, (code + 1) * 0.25).astype(np.float32), mask)


@pytest.mark.parametrize("C,n,r", [(3, 64, 8), (1, 28, 4)])
def test_chain_forward_draws_on_the_device(ops, C, n, r):
    """Philox mode: same element <-> counter mapping as ee_square_draw_f32, the last workgroup advances the state, the ticket
    returns to 0, and a replayed HIP graph draws new numbers."""
    B = 33
    x, op = _setup(C, n, r, B, 3)
    eps, s = 16 / 255, max(int(round((0.8 * n * n) ** 0.5)), 1)
    wts = ops.EdgeWeights(1.0)
    st_a = torch.tensor([99, 40, 0, 0], dtype=torch.int64, device=DEV)
    st_b = st_a.clone()
    sizes = torch.tensor([s], dtype=torch.int32, device=DEV)
    stripe, pos, sign = ops.square_draw(B, C, n, sizes, st_a)
    d = {"stripe": stripe, "sq_pos": pos, "sq_sign": sign, "sq_size": sizes}
    want = ops.chain_fwd(x, op.chain, wts, 0.0, 76 / 255, 1.0, True, eps, s, None, d)
    got = ops.chain_fwd(x, op.chain, wts, 0.0, 76 / 255, 1.0, True, eps, s, st_b, None)
    for a, b in zip(got[:4], want[:4]):
        assert torch.equal(a, b)
    assert st_b.tolist() == st_a.tolist() and st_b[2].item() == 0 and st_b[1].item() == 40 + (B * C * n + 1 + C + 3) // 4
    nxt = ops.chain_fwd(x, op.chain, wts, 0.0, 76 / 255, 1.0, True, eps, s, st_b, None)
    assert not torch.equal(nxt[0], got[0])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.chain_fwd(x, op.chain, wts, 0.0, 76 / 255, 1.0, True, eps, s, st_b, None)
    g.replay()
    first = out[0].clone()
    g.replay()
    assert not torch.equal(out[0], first) and st_b[2].item() == 0


@pytest.mark.parametrize("C,n,r", SHAPES)
@pytest.mark.parametrize("B,direction", [(1, 1), (7, -1), (100, 1)])
def test_chain_backward_update_vs_separate_kernels(ops, C, n, r, B, direction):
    x, op = _setup(C, n, r, B, 4)
    eps_sq = 16 / 255 if C == 3 else 0.3
    alpha, high, w = (0.0, 76 / 255, 1.0) if C == 3 else (0.3, 51 / 255, 1.0)
    s = max(int(round((0.8 * n * n) ** 0.5)), 1)
    d = _draws(ops, B, C, n, s)
    wts = ops.EdgeWeights(1.0)
    x_in, gate, gx, gy, _ = ops.chain_fwd(x, op.chain, wts, alpha, high, w, True, eps_sq, s, None, d)
    torch.manual_seed(9)
    g_in = torch.randn(B, C, n, n, device=DEV)
    g_in[:, :, 1, :8] = 0.0
    x0 = (x + (torch.rand_like(x) - 0.5) * 0.1).clamp(0, 1)
    step, ball = 2 / 255, 16 / 255
    # the separate kernels on the same saved state (their gate is a boolean: bit 0)
    gate0 = (gate & 1).contiguous()
    g_hfs, g_edge = ops.frontend_bwd_saved(g_in, gate0, gx, gy, wts, alpha, high, w)
    g_lp = op.backward_square(g_hfs, x, eps_sq, d)
    want = x.clone()
    ops.pgd_step_bcast_(want, g_lp, g_edge, x0, step, ball, 0.0, 1.0, direction)
    got = x.clone()
    ops.chain_bwd_(got, g_in, gate, gx, gy, x0, op.chain, wts, alpha, high, w, step, ball, 0.0, 1.0, direction)
    g_ref = (g_lp + g_edge).cpu().numpy()
    assert np.isnan(g_ref).sum() > 0  # the flat patch
    scale = np.nanmax(np.abs(g_ref))
    # low-pass rounding (2e-6 of the largest entry) can flip a smaller gradient's sign; where d add_square / dx = 0 the low-pass
    # part is an exact 0 on both sides
    decided = np.isnan(g_ref) | (np.abs(g_ref) > 2e-6 * scale) | (g_lp.cpu().numpy() == 0)
    gw, ww = got.cpu().numpy(), want.cpu().numpy()
    assert np.array_equal(gw[decided], ww[decided])
    assert decided.mean() > 0.99 and (gw == ww).mean() > 0.999
    assert np.array_equal(gw[np.isnan(g_ref)], np.clip(np.minimum(np.maximum(x.cpu().numpy(), x0.cpu().numpy() - np.float32(ball)),
                                                                  x0.cpu().numpy() + np.float32(ball)), 0, 1)[np.isnan(g_ref)])
    # against the C oracle: same formula on the reference gradient
    ow = O.pgd_step(x.cpu().numpy(), g_ref, x0.cpu().numpy(), step, ball, direction=direction)
    assert np.array_equal(ww, ow)


def test_engine_uses_the_chain_and_matches_the_manual_path(monkeypatch):
    """engine.attack_step_ on an EE_square model: two front-end launches per iteration (ee_chain) against the six-launch manual
    path, same draws: the iterate agrees except where the gradient's sign is inside rounding noise."""
    from eeadv import engine, models, ops, runtime
    torch.manual_seed(11)
    m = models.make_resnet_ee(18, "tiny", True, cize=64, r=8, w=1.0, with_gf=False, low=38.0, high=76.0, alpha=0.0, sigma=1.0,
                              type_canny="CannyFilter_step125_1", epsilon=16 / 255, n_queries=1).to(DEV).eval()
    x = torch.rand(6, 3, 64, 64, device=DEV)
    x[1, :, 20:40, 8:30] = 0.25
    y = torch.randint(0, 200, (6,), device=DEV)
    assert m.chain_ok(x)
    spec = engine.LossSpec(engine.CE_SUM, y)
    runtime.reseed()
    torch.manual_seed(3)
    state = runtime.draw_state(x.device)
    cur = x.clone()
    for it in range(3):  # both paths from the same iterate and the same Philox state, step by step (the CNN in between is MIOpen's:
        saved = state.clone()  # two runs of it are not bit-identical, and a flipped sign moves a pixel by 2 alpha from then on)
        res = {}
        for chain in (True, False):
            monkeypatch.setattr(models, "_CHAIN", chain)
            state.copy_(saved)
            xa = cur.clone()
            engine.attack_step_(m, xa, x, spec, 2 / 255, 16 / 255, 1, 0.0, 1.0)
            res[chain] = xa
        assert state[1].item() > saved[1].item() and state[2].item() == 0  # the draws advanced, by the same amount on both paths
        a, b = res[True].cpu().numpy(), res[False].cpu().numpy()
        # the two back ends of the front end round differently, so sign(g) flips where |g| is inside that noise (0.05-0.15 % of the pixels,
        # depending on which kernels the CNN in between runs on): such a pixel moves by one step in the other direction, nothing else may differ
        assert float(np.abs(a - cur.cpu().numpy()).max()) > 0 and (a == b).mean() > 0.998, (it, (a == b).mean())
        assert float(np.abs(a - b).max()) <= 2 * (2 / 255) + 1e-6
        cur = res[True]
    a = cur.cpu().numpy()
    assert np.array_equal(a[1, :, 24:36, 12:26], x.cpu().numpy()[1, :, 24:36, 12:26])  # NaN gradient inside the flat patch: no update


@pytest.mark.parametrize("train", [False, True])
def test_whole_attack_is_reproducible_bit_for_bit(train):
    """PGD-10 on resnet18_EE_square (the bench's model, graph replays + one eager probe iteration) twice from the same iterate and the same
    Philox state: no kernel of an iteration is MIOpen's any more (whose backward-data used atomics: two runs of the stock network differ,
    profiles/round2_c_stock_miopen_nondeterminism.txt), every reduction here runs in a fixed order, so the adversarial batch comes out
    IDENTICAL - eval mode and train mode (batch-statistics BatchNorm, as the reference's attack runs it)."""
    from eeadv import engine, models, runtime
    torch.manual_seed(5)
    m = models.make_resnet_ee(18, "tiny", True, cize=64, r=8, w=1.0, with_gf=False, low=38.0, high=76.0, alpha=0.0, sigma=1.0,
                              type_canny="CannyFilter_step125_1", epsilon=16 / 255, n_queries=1).to(DEV)
    m.train(train)
    x = torch.rand(20, 3, 64, 64, device=DEV)
    y = torch.randint(0, 200, (20,), device=DEV)
    spec = engine.LossSpec(engine.CE_SUM, y)
    runs = []
    runtime.reseed()
    torch.manual_seed(9)
    state = runtime.draw_state(x.device)
    saved = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}
    engine.pgd_loop(m, x, x.clone(), spec, 10, 2 / 255, 16 / 255, use_graph=True)  # builds the graph (its warm-up passes draw too)
    state0 = state.clone()
    for _ in range(2):
        state.copy_(state0)  # the same Add_Square draws
        live = m.state_dict()
        for k, v in saved.items():  # the same BatchNorm buffers
            live[k].data.copy_(v)
        runs.append(engine.pgd_loop(m, x, x.clone(), spec, 10, 2 / 255, 16 / 255, use_graph=True).clone())
    assert float((runs[0] - x).abs().max()) > 1e-3
    assert torch.equal(runs[0], runs[1])


@pytest.mark.parametrize("H,W,r,C,B", [(224, 224, 16, 3, 4), (64, 64, 8, 3, 5), (28, 28, 4, 1, 3), (7, 9, 2, 2, 3), (96, 128, 12, 1, 2), (256, 256, 16, 1, 2),
                                       (50, 70, 6, 2, 2)])
def test_band_lowpass_kernel_vs_float64_operator_UNPINNED(ops, H, W, r, C, B):
    """ee_hfs_mfma_f32 (planes up to 256 x 256 on the matrix cores, one wavefront per 16-row band) against the dense operator in
    float64 and the FFT restatement of the oracle; linear, self-adjoint, reproducible bit for bit from run to run."""
    from eeadv import hfs as HF
    from oracle import ref_path as R
    torch.manual_seed(H * W + r)
    x = torch.rand(B, C, H, W, device=DEV)
    flat, nu_pad = HF.band_tables(H, W, r)
    tab = torch.from_numpy(flat).to(DEV)
    y = ops.hfs_mfma(x, tab, nu_pad)
    Ar, Ai, B1, B2 = HF.hfs_matrices(H, W, r)
    xd = x.cpu().double().numpy()
    want = np.einsum("hk,bckw->bchw", Ar, xd @ B1) + np.einsum("hk,bckw->bchw", Ai, xd @ B2)
    assert np.abs(y.cpu().numpy() - want).max() < 2e-6
    np.testing.assert_allclose(y.cpu().numpy(), R.HighFreqSuppress(H, W, r)(x.cpu()).numpy(), atol=3e-6)
    assert torch.equal(y, ops.hfs_mfma(x, tab, nu_pad))  # band-ordered reduction: reproducible
    u = torch.randn_like(x)
    lhs, rhs = float((ops.hfs_mfma(x, tab, nu_pad) * u).sum()), float((x * ops.hfs_mfma(u, tab, nu_pad)).sum())
    assert abs(lhs - rhs) < 1e-3 * (abs(lhs) + 1)  # self-adjoint


@pytest.mark.parametrize("n,r,C", [(224, 16, 3), (64, 8, 3), (96, 12, 1)])
def test_band_lowpass_fused_add_square(ops, n, r, C):
    """sq_mode 1 / 2 of ee_hfs_mfma_f32 against Add_Square (its own kernels, pinned to the reference) around the plain operator."""
    import utils.core as core
    from eeadv import hfs as HF
    torch.manual_seed(n)
    B, eps = 3, 16 / 255
    x = torch.rand(B, C, n, n, device=DEV)
    x[0, 0, 0, :4] = torch.tensor([0.0, 1.0, eps, 1 - eps], device=DEV)
    sq = core.Add_Square(C, n, eps, n_queries=1)
    d = sq.prepare(x)
    flat, nu_pad = HF.band_tables(n, n, r)
    tab = torch.from_numpy(flat).to(DEV)
    xs = ops.add_square_fwd(x, eps, d["stripe"], d["sq_sign"], d["sq_pos"], d["sq_size"])
    fused = ops.hfs_mfma(x, tab, nu_pad, 1, None, eps, d["stripe"], d["sq_sign"], d["sq_pos"], d["sq_size"])
    assert torch.equal(fused, ops.hfs_mfma(xs, tab, nu_pad))  # same kernel, same staged values
    g = torch.randn_like(x)
    mask = ops.add_square_bwd(torch.ones_like(x), x, eps, d["stripe"], d["sq_sign"], d["sq_pos"], d["sq_size"])
    back = ops.hfs_mfma(g, tab, nu_pad, 2, x, eps, d["stripe"], d["sq_sign"], d["sq_pos"], d["sq_size"])
    assert torch.equal(back, ops.hfs_mfma(g, tab, nu_pad) * mask)


def test_imagenet_ee_square_front_end_runs_on_the_band_kernel():
    """ImageNet/configs_imagenet/ee_at_bpda3_square.yml (cize 224, r 16): HighFreqSuppress no longer falls back to three rocBLAS
    launches - forward and input gradient of the front end against the oracle's FFT restatement."""
    from eeadv import models
    from oracle import ref_path as R
    torch.manual_seed(5)
    m = models.make_resnet_ee(18, "imagenet", True, cize=224, r=16, w=1.0, with_gf=False, low=38.0, high=76.0, alpha=0.0, sigma=1.0,
                              type_canny="CannyFilter_step125_1", epsilon=16 / 255, n_queries=1).to(DEV).eval()
    op = m.hfs.operator(torch.device(DEV))
    assert op.kernel is None and op.mfma is not None and op.fused_square
    x = torch.rand(2, 3, 224, 224)
    front = R.EEFront(224, 3, 16, 1.0, 38.0, 76.0, 0.0, 1.0, "CannyFilter_step125_1", False, True, 16 / 255, 1)
    draws = front.add_square.draw(2)
    ddev = {"stripe": draws["stripe"].to(DEV), "sq_pos": draws["sq_pos"].to(DEV), "sq_sign": draws["sq_sign"].reshape(1, 3).to(DEV)}
    xr = x.clone().requires_grad_(True)
    want = front(xr, draws)
    xd = x.to(DEV).requires_grad_(True)
    got = m.front(xd, ddev)
    assert float(((got.detach().cpu() - want.detach()).abs() > 1e-5).float().mean()) == 0.0
    u = torch.randn_like(x)
    (want * u).sum().backward()
    (got * u.to(DEV)).sum().backward()
    g, gr = xd.grad.cpu().numpy(), xr.grad.numpy()
    assert np.array_equal(np.isnan(g), np.isnan(gr))
    fin = ~np.isnan(gr)
    assert np.abs(g[fin] - gr[fin]).max() < 2e-5 * np.abs(gr[fin]).max() + 1e-6


def test_whole_training_step_is_reproducible_bit_for_bit(monkeypatch):
    """Five adversarial-training steps of resnet18_EE_square (random start, PGD-4 in train mode, the update: two eager steps, the captures,
    graph replays) twice from the same seeds: since round 3 every weight gradient of this network comes from ee_wrw.hip, whose partial sums
    are added in a fixed order - MIOpen's weight-gradient solvers used atomics, so two runs of the same step differed in the last bits and a
    deep ReLU network amplified that from step to step.  Now the parameters, the BatchNorm buffers and the losses come out IDENTICAL."""
    from eeadv import engine, models, runtime, trainer
    from tiny_models import Args
    monkeypatch.setenv("EEADV_GRAPH", "1")
    g = torch.Generator().manual_seed(17)
    batches = [(torch.rand(16, 3, 64, 64, generator=g).to(DEV), torch.randint(0, 200, (16,), generator=g).to(DEV)) for _ in range(5)]
    args = Args(method_name="EE_BPDA3_AT_square", random=True, epsilon=16 / 255, num_steps_1=4, step_size_1=2 / 255, num_classes=200, beta=6.0)
    runs = []
    for _ in range(2):
        engine.clear_graphs()
        trainer.clear_update_graphs()
        torch.manual_seed(23)
        runtime.reseed()
        model = models.make_resnet_ee(18, "tiny", True, cize=64, r=8, w=1.0, with_gf=False, low=38.0, high=76.0, alpha=0.0, sigma=1.0,
                                      type_canny="CannyFilter_step125_1", epsilon=16 / 255, n_queries=1).to(DEV).train()
        opt = trainer.make_sgd(model.parameters(), lr=0.05, momentum=0.9, weight_decay=2e-4)
        crit = trainer.make_criterion(args)
        losses = []
        for x, y in batches:
            loss, _ = trainer.train_batch(model, crit, opt, args, x, y, torch.device(DEV))
            losses.append(float(loss))
        state = torch.cat([t.detach().flatten().float() for t in list(model.parameters()) + [b for n, b in model.named_buffers() if "running" in n]])
        runs.append((losses, state.clone()))
    engine.clear_graphs()
    trainer.clear_update_graphs()
    assert len(set(runs[0][0])) > 1 and all(np.isfinite(runs[0][0]))
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1])
