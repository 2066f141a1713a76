#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the REFERENCE itself.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py

How the reference is imported
-----------------------------
`utils/attacks.py` and `utils/core.py` import three modules this image lacks
(`torch._six` via utils/_jit_internal.py:9, `cv2` at attacks.py:8 / core.py:10,
`torchvision` in the model files).  They are registered in `sys.modules` as
EMPTY `types.ModuleType` objects (only `torch._six.builtins`, needed by the
import statement itself, is set).  Nothing computed below touches them, with
ONE exception that is handled explicitly:

* `get_thin_kernels()` (core.py:87-112) needs real `cv2.getRotationMatrix2D`
  / `cv2.warpAffine`.  Every Canny class calls it in `__init__`.  It is
  replaced by the analytically derived table `THIN_TABLE` below.
  - `CannyFilter_step125_1.forward` (core.py:549-585) never reads
    `weight_directional`, so its fixtures are unaffected by that table and are
    genuinely pinned to the reference.
  - `CannyFilter` / `CannyFilter_BPDA` fixtures DO depend on the table; they
    are written to `canny_full_unpinned.npz` and are labelled
    "parity unpinned" wherever they are used.

The fixtures hold data only (inputs, seeds, outputs) - no reference source.
"""
import builtins
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
_six = types.ModuleType("torch._six")
_six.builtins = builtins
sys.modules["torch._six"] = _six
sys.modules["cv2"] = types.ModuleType("cv2")
_tv = types.ModuleType("torchvision")
_tvm = types.ModuleType("torchvision.models")
_tv.models = _tvm
sys.modules["torchvision"] = _tv
sys.modules["torchvision.models"] = _tvm

REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "MNIST"))
sys.path.insert(0, os.path.join(REF, "Tiny_ImageNet"))

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import utils.attacks as rattacks  # noqa: E402
import utils.core as rcore  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# k*45 degrees -> (row, col) of the -1 neighbour in the 3x3 directional kernel
# (centre is +1).  Derived from core.py:87-112 (rotation of [0,0,1,-1,-1] about
# the centre, counter-clockwise in image coordinates); real cv2 unavailable.
THIN_TABLE = {0: (1, 2), 1: (0, 2), 2: (0, 1), 3: (0, 0), 4: (1, 0), 5: (2, 0), 6: (2, 1), 7: (2, 2)}


def derived_thin_kernels(start=0, end=360, step=45):
    ks = []
    for angle in range(start, end, step):
        k = np.zeros((3, 3))
        k[1, 1] = 1
        r, c = THIN_TABLE[(angle // 45) % 8]
        k[r, c] = -1
        ks.append(k)
    return ks


rcore.get_thin_kernels = derived_thin_kernels


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


class Args:
    def __init__(self, **kw):
        self.__dict__.update(kw)


# --------------------------------------------------------------------------
# 1. fixed weights and masks
# --------------------------------------------------------------------------
def gen_kernels():
    out = {
        "gauss_k3_mu0_s1": rcore.get_gaussian_kernel(3, 0, 1),
        "gauss_k3_mu0_s2": rcore.get_gaussian_kernel(3, 0, 2.0),
        "gauss_k5_mu0_s1": rcore.get_gaussian_kernel(5, 0, 1),
        "gauss_k3_unnorm": rcore.get_gaussian_kernel(3, 0, 1, normalize=False),
        "sobel_k3": rcore.get_sobel_kernel(3),
        "sobel_k5": rcore.get_sobel_kernel(5),
    }
    for (w, h, r) in [(64, 64, 8), (28, 28, 4), (224, 224, 16), (7, 9, 2), (32, 32, 4)]:
        m = rcore.HighFreqSuppress(w, h, r)
        out["hfs_mask_%d_%d_%d" % (w, h, r)] = m.temp.numpy()[0, 0, :, :, 0].astype(np.uint8)
    save("kernels", **out)


# --------------------------------------------------------------------------
# 2. CannyFilter_step125_1 forward + backward
# --------------------------------------------------------------------------
def rect_image(B, C, H, W, seed):
    """Piece-wise constant image (MNIST-like): exercises mag == 0 -> NaN grads."""
    rng = np.random.RandomState(seed)
    x = np.zeros((B, C, H, W), np.float32)
    for b in range(B):
        for _ in range(3):
            h0, w0 = rng.randint(0, H - 4), rng.randint(0, W - 4)
            h1, w1 = rng.randint(h0 + 2, H), rng.randint(w0 + 2, W)
            for c in range(C):
                x[b, c, h0:h1, w0:w1] = np.float32(rng.choice([0.25, 0.5, 0.75, 1.0]))
    return x


def ramp_images(C, H, W, high):
    """Horizontal ramps whose Sobel magnitude sits within a few ulp of `high`."""
    xs = []
    base = np.float32(high) / np.float32(4.0)
    for k in range(-3, 4):
        a = np.float32(base) * (np.float32(1.0) + np.float32(k) * np.float32(2.0 ** -23))
        row = (np.arange(W, dtype=np.float32) * a).astype(np.float32)
        xs.append(np.broadcast_to(row, (C, H, W)).copy())
    return np.stack(xs).astype(np.float32)


def run_edge125(x_np, alpha, high, seed_u):
    filt = rcore.CannyFilter_step125_1(sigma=1, alpha=alpha)
    x = torch.from_numpy(x_np.copy()).requires_grad_(True)
    e = filt(x, low_threshold=high / 2, high_threshold=high, hysteresis=True)
    g = torch.Generator().manual_seed(seed_u)
    u = torch.randn(e.shape, generator=g)
    (e * u).sum().backward()
    return e.detach().numpy().astype(np.uint8), u.numpy(), x.grad.numpy()


def gen_edge125():
    # NOTE: every case has batch >= 2.  torch routes N == 1 small 3x3 convolutions through its
    # im2col+GEMM path, whose summation order differs (by rounding noise) from the oneDNN direct
    # convolution used for N > 1 - the shape class of every reference config (B = 50 / 100 / 256).
    cases = {}
    torch.manual_seed(0)
    cases["rand_tiny"] = (torch.rand(2, 3, 64, 64).numpy(), 0.0, 76.0 / 255)
    torch.manual_seed(0)
    cases["rand_mnist"] = (torch.rand(2, 1, 28, 28).numpy(), 0.3, 51.0 / 255)
    cases["rect_mnist"] = (rect_image(2, 1, 28, 28, 3), 0.3, 51.0 / 255)
    cases["rect_rgb"] = (rect_image(2, 3, 32, 32, 4), 0.0, 76.0 / 255)
    cases["ramp_thr"] = (ramp_images(3, 16, 16, 76.0 / 255), 0.0, 76.0 / 255)
    torch.manual_seed(5)
    cases["ragged"] = (torch.rand(2, 3, 17, 23).numpy(), 0.0, 76.0 / 255)
    torch.manual_seed(6)
    cases["two_ch"] = (torch.rand(3, 2, 5, 9).numpy(), 0.1, 0.2)
    torch.manual_seed(7)
    cases["one_px"] = (torch.rand(2, 3, 1, 1).numpy(), 0.0, 76.0 / 255)
    torch.manual_seed(8)
    cases["thin"] = (torch.rand(2, 1, 1, 40).numpy(), 0.0, 0.1)
    torch.manual_seed(9)
    # smooth image scaled so that many magnitudes exceed 1.001 (upper gate)
    cases["big_mag"] = ((torch.rand(2, 3, 24, 24) * 6.0).numpy(), 0.0, 76.0 / 255)
    out = {}
    for name, (x, alpha, high) in cases.items():
        e, u, gx = run_edge125(x, alpha, high, 11)
        out[name + "__x"] = x.astype(np.float32)
        out[name + "__alpha_high"] = np.array([alpha, high], np.float64)
        out[name + "__edge"] = e
        out[name + "__u"] = u
        out[name + "__gx"] = gx
        print("  edge125", name, x.shape, "edges", int(e.sum()), "nan grads", int(np.isnan(gx).sum()))
    save("edge125", **out)


# --------------------------------------------------------------------------
# 3. the PGD / FGSM update, recorded step by step
# --------------------------------------------------------------------------
class TinyNet(nn.Module):
    """A small differentiable classifier with exact-zero gradients (ReLU) so
    that sign(0) = 0 is exercised.  Seeded; rebuilt identically in tests."""

    def __init__(self, cin, hw, ncls, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w1 = nn.Parameter(torch.randn(8, cin, 3, 3, generator=g) * 0.5)
        self.w2 = nn.Parameter(torch.randn(ncls, 8 * (hw // 2) * (hw // 2), generator=g) * 0.2)

    def forward(self, x):
        h = F.relu(F.conv2d(x, self.w1, padding=1))
        h = F.avg_pool2d(h, 2)
        return F.linear(h.flatten(1), self.w2)


class Recorder(nn.Module):
    """Wraps a model; records every input it is called with and the gradient
    that later arrives at that input."""

    def __init__(self, m):
        super().__init__()
        self.m = m
        self.xs, self.gs = [], []

    def forward(self, x):
        self.xs.append(x.detach().clone().numpy())
        if x.requires_grad:
            x.register_hook(lambda g: self.gs.append(g.detach().clone().numpy()))
        return self.m(x)


def gen_pgd():
    out = {}
    B, C, HW, K = 3, 2, 8, 10
    torch.manual_seed(21)
    x0 = torch.rand(B, C, HW, HW)
    # make some pixels sit on the [0,1] and eps-box borders
    x0[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 0.001, 0.999])
    y = torch.tensor([1, 0, 4])
    eps, alpha, steps = 0.062745098039216, 0.007843137254902, 6
    out["x0"], out["y"] = x0.numpy(), y.numpy()
    out["eps_alpha"] = np.array([eps, alpha], np.float64)

    # (a) untargeted PGD with random start
    rec = Recorder(TinyNet(C, HW, K, 31))
    torch.manual_seed(100)
    xa = rattacks.PGD(rec, Args(random=True, epsilon=eps), x0, y, steps, alpha)
    torch.manual_seed(100)
    noise = torch.zeros_like(x0).uniform_(-eps, eps)
    out["pgd_noise"] = noise.numpy()
    out["pgd_xs"] = np.stack(rec.xs)
    out["pgd_gs"] = np.stack(rec.gs)
    out["pgd_final"] = xa.numpy()

    # (b) untargeted PGD without random start, MNIST-like parameters
    rec = Recorder(TinyNet(C, HW, K, 32))
    xb = rattacks.PGD(rec, Args(random=False, epsilon=0.3), x0, y, 8, 0.01)
    out["pgdb_xs"], out["pgdb_gs"], out["pgdb_final"] = np.stack(rec.xs), np.stack(rec.gs), xb.numpy()

    # (c) targeted PGD (descent)
    rec = Recorder(TinyNet(C, HW, K, 33))
    torch.manual_seed(101)
    xc, tl = rattacks.targeted_PGD(rec, Args(random=True, epsilon=eps), x0, y, steps, alpha, K, "cpu")
    torch.manual_seed(101)
    off = torch.randint(low=1, high=K, size=y.shape)
    noise_c = torch.zeros_like(x0).uniform_(-eps, eps)
    out["tpgd_offset"], out["tpgd_noise"] = off.numpy(), noise_c.numpy()
    out["tpgd_target"] = tl.numpy()
    out["tpgd_xs"], out["tpgd_gs"], out["tpgd_final"] = np.stack(rec.xs), np.stack(rec.gs), xc.numpy()

    # (d) FGSM, both directions
    for targeted in (False, True):
        rec = Recorder(TinyNet(C, HW, K, 34))
        xf = rattacks.FGSM(rec, x0, y, targeted=targeted, step_size=0.007)
        tag = "fgsm_t" if targeted else "fgsm_u"
        out[tag + "_g"], out[tag + "_final"] = rec.gs[0], xf.numpy()

    # (e) a gradient with NaN / +-0 / inf entries through the PGD update lines
    class FixedGrad(nn.Module):
        def __init__(self, g):
            super().__init__()
            self.g = g

        def forward(self, x):
            # logits whose dCE/dx is dominated by self.g; NaNs are injected by hook
            x.register_hook(lambda gr: self.g)
            return (x * 0).flatten(1)[:, :K] + x.flatten(1)[:, :K]

    gspec = torch.randn(B, C, HW, HW)
    gspec[0, 0, 0, :6] = torch.tensor([float("nan"), 0.0, -0.0, float("inf"), -float("inf"), 1e-40])
    xe = rattacks.PGD(FixedGrad(gspec), Args(random=False, epsilon=eps), x0, y, 1, alpha)
    out["special_g"], out["special_final"] = gspec.numpy(), xe.numpy()
    save("pgd_steps", **out)


# --------------------------------------------------------------------------
# 4. losses
# --------------------------------------------------------------------------
def gen_losses():
    out = {}
    for tag, (B, K, seed) in {"s": (4, 10, 2), "t": (16, 200, 3), "i": (5, 1000, 4)}.items():
        torch.manual_seed(seed)
        la = (torch.randn(B, K) * 2).requires_grad_(True)
        lb = (torch.randn(B, K) * 2).requires_grad_(True)
        y = torch.randint(0, K, (B,))
        out[tag + "_la"], out[tag + "_lb"], out[tag + "_y"] = la.detach().numpy(), lb.detach().numpy(), y.numpy()

        ce_sum = F.cross_entropy(la, y, reduction="sum")
        (g_ce_sum,) = torch.autograd.grad(ce_sum, la)
        ce_mean = F.cross_entropy(la, y)
        (g_ce_mean,) = torch.autograd.grad(ce_mean, la)
        out[tag + "_ce_sum"], out[tag + "_ce_sum_g"] = ce_sum.item(), g_ce_sum.numpy()
        out[tag + "_ce_mean"], out[tag + "_ce_mean_g"] = ce_mean.item(), g_ce_mean.numpy()

        tr = rattacks.Trades(beta=6.0)
        kl = tr.criterion_kl(F.log_softmax(lb, dim=1), F.softmax(la, dim=-1))
        g_kl_b, g_kl_a = torch.autograd.grad(kl, [lb, la])
        out[tag + "_kl"], out[tag + "_kl_gq"], out[tag + "_kl_gp"] = kl.item(), g_kl_b.numpy(), g_kl_a.numpy()

        mse = F.mse_loss(la, lb)
        out[tag + "_mse"] = mse.item()

        class _Opt:
            def zero_grad(self):
                pass

        alp = rattacks.ALP(beta=0.5).loss(nn.Identity(), la, lb, y, _Opt())
        ga, gb = torch.autograd.grad(alp, [la, lb])
        out[tag + "_alp"], out[tag + "_alp_ga"], out[tag + "_alp_gb"] = alp.item(), ga.numpy(), gb.numpy()

        # Trades.loss: model(x_adv) -> use a linear "model" so that logits_adv = lb-like
        lin = nn.Linear(K, K, bias=False)
        with torch.no_grad():
            lin.weight.copy_(torch.eye(K))
        xadv = lb.detach().clone()
        tl = tr.loss(lin, la, xadv, y, _Opt())
        (g_tl,) = torch.autograd.grad(tl, la)
        out[tag + "_trades"], out[tag + "_trades_ga"] = tl.item(), g_tl.numpy()

        ls = rattacks.LabelSmoothLoss(0.1)(la, y)
        (g_ls,) = torch.autograd.grad(ls, la)
        out[tag + "_lsmooth"], out[tag + "_lsmooth_g"] = ls.item(), g_ls.numpy()
        out[tag + "_cle"] = rattacks.compute_loss_and_error(la, y, 0.2).item()
        out[tag + "_pred"] = rattacks.predict_from_logits(la).numpy()

        av = rattacks.AVmixup(Args(random=False, epsilon=0.1), 2.0, 1.0, 0.1, 0.01, 1, num_classes=K, device="cpu")
        onehot = torch.eye(K)[y]
        out[tag + "_smooth_l1"] = av._label_smoothing(onehot, 1.0).numpy()
        out[tag + "_smooth_l2"] = av._label_smoothing(onehot, 0.1).numpy()
        # soft-label loss used inside AVmixup.perturb (attacks.py:462-463) and by the driver
        soft = av._label_smoothing(onehot, 0.1).double()
        sl = -torch.sum(F.log_softmax(la, dim=1) * soft) / B
        (g_sl,) = torch.autograd.grad(sl, la)
        out[tag + "_softce_f64"], out[tag + "_softce_g"] = sl.item(), g_sl.numpy()
        out[tag + "_l2norm"] = rattacks.l2_norm(la.detach().view(B, 1, 1, K)).numpy()
    save("losses", **out)


# --------------------------------------------------------------------------
# 5. AVmixup.perturb / CW (targeted) end to end on TinyNet
# --------------------------------------------------------------------------
def gen_avmix_cw():
    out = {}
    B, C, HW, K = 4, 2, 8, 10
    torch.manual_seed(41)
    x0 = torch.rand(B, C, HW, HW)
    y = torch.tensor([3, 1, 0, 7])
    eps, alpha = 0.062745098039216, 0.003921568627451
    out["x0"], out["y"] = x0.numpy(), y.numpy()
    out["eps_alpha"] = np.array([eps, alpha], np.float64)

    rec = Recorder(TinyNet(C, HW, K, 51))
    av = rattacks.AVmixup(Args(random=True, epsilon=eps), 2.0, 1.0, 0.1, alpha, 5, num_classes=K, device="cpu")
    torch.manual_seed(200)
    np.random.seed(201)
    xm, ym = av.perturb(rec, x0, torch.eye(K)[y])
    torch.manual_seed(200)
    noise = torch.zeros_like(x0).uniform_(-eps, eps)
    np.random.seed(201)
    beta = np.random.beta(1.0, 1.0, [B, 1, 1, 1])
    out["av_noise"], out["av_beta"] = noise.numpy(), beta
    out["av_xs"], out["av_gs"] = np.stack(rec.xs), np.stack(rec.gs)
    out["av_x"], out["av_y"] = xm.numpy(), ym.numpy()
    assert ym.dtype == torch.float64 and xm.dtype == torch.float32

    # CW-Linf, targeted form (the only one that runs in the reference)
    net = TinyNet(C, HW, K, 52)
    tgt = torch.tensor([4, 2, 1, 8])
    torch.manual_seed(202)
    adv, p = rattacks.CWLinfAttack(x0, net(x0).argmax(1), net, eps, None, eps, max_iters=4, target=tgt,
                                   n_class=K, cur_device="cpu")
    torch.manual_seed(202)
    rp = torch.FloatTensor(x0.shape).uniform_(-eps, eps)
    out["cw_noise"], out["cw_target"] = rp.numpy(), tgt.numpy()
    out["cw_adv"], out["cw_p"] = adv.detach().numpy(), p.detach().numpy()
    save("avmix_cw", **out)


# --------------------------------------------------------------------------
# 6. end-to-end: reference models through reference PGD (weights re-seeded in tests)
# --------------------------------------------------------------------------
def state_checksum(model):
    s = 0.0
    for p in model.state_dict().values():
        s += float(p.double().abs().sum())
    return s


def gen_e2e():
    out = {}
    from models_mnist.Net2 import Net_2
    torch.manual_seed(7)
    net = Net_2().eval()
    out["net2_checksum"] = state_checksum(net)
    torch.manual_seed(1)
    x = torch.rand(6, 1, 28, 28)
    y = torch.randint(0, 10, (6,))
    torch.manual_seed(300)
    rec = Recorder(net).eval()
    xa = rattacks.PGD(rec, Args(random=True, epsilon=0.3), x, y, 40, 0.01)
    # every iterate the reference fed to the model and the gradient it got back (the step-by-step replay of
    # tests/test_gpu_path.py feeds these x_k to the GPU model and explains every differing pixel by |g|)
    out["net2_xs"], out["net2_gs"] = np.stack(rec.xs), np.stack(rec.gs)
    torch.manual_seed(300)
    out["net2_noise"] = torch.zeros_like(x).uniform_(-0.3, 0.3).numpy()
    with torch.no_grad():
        out["net2_logits_clean"] = net(x).numpy()
        out["net2_logits_adv"] = net(xa).numpy()
    out["net2_x"], out["net2_y"], out["net2_xadv"] = x.numpy(), y.numpy(), xa.numpy()

    # load resnet.py as a plain file: the package __init__ pulls in resnet_EE*.py,
    # which need torchvision.transforms / turtle and hard-code .cuda()
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_tiny_resnet", os.path.join(REF, "Tiny_ImageNet", "models_tinyimagenet", "resnet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    resnet18 = mod.resnet18
    torch.manual_seed(8)
    rn = resnet18().eval()
    out["rn18_checksum"] = state_checksum(rn)
    torch.manual_seed(2)
    x = torch.rand(2, 3, 64, 64)
    y = torch.randint(0, 200, (2,))
    eps, alpha = 0.062745098039216, 0.007843137254902
    torch.manual_seed(301)
    rec = Recorder(rn).eval()
    xa = rattacks.PGD(rec, Args(random=True, epsilon=eps), x, y, 3, alpha)
    out["rn18_xs"], out["rn18_gs"] = np.stack(rec.xs), np.stack(rec.gs)
    torch.manual_seed(301)
    out["rn18_noise"] = torch.zeros_like(x).uniform_(-eps, eps).numpy()
    with torch.no_grad():
        out["rn18_logits_clean"] = rn(x).numpy()
        out["rn18_logits_adv"] = rn(xa).numpy()
    out["rn18_x"], out["rn18_y"], out["rn18_xadv"] = x.numpy(), y.numpy(), xa.numpy()
    save("e2e", **out)


# --------------------------------------------------------------------------
# 7. full Canny / BPDA (depends on the DERIVED thin-kernel table -> unpinned)
# --------------------------------------------------------------------------
def gen_canny_full():
    out = {}
    torch.manual_seed(12)
    xs = {"rand_rgb": (torch.rand(2, 3, 32, 32).numpy(), 0.0, 38.0 / 255, 76.0 / 255),
          "rand_mnist": (torch.rand(2, 1, 28, 28).numpy(), 0.3, 25.0 / 255, 51.0 / 255),
          "rect_rgb": (rect_image(2, 3, 32, 32, 14), 0.0, 38.0 / 255, 76.0 / 255)}
    for name, (x_np, alpha, low, high) in xs.items():
        for cls_name in ("CannyFilter", "CannyFilter_BPDA"):
            filt = getattr(rcore, cls_name)(sigma=1, alpha=alpha)
            x = torch.from_numpy(x_np.copy()).requires_grad_(True)
            e = filt(x, low_threshold=low, high_threshold=high, hysteresis=True)
            g = torch.Generator().manual_seed(13)
            u = torch.randn(e.shape, generator=g)
            (e * u).sum().backward()
            key = name + "__" + cls_name
            out[key + "__edge"] = e.detach().numpy().astype(np.float32)
            out[key + "__gx"] = x.grad.numpy()
            out[name + "__u"] = u.numpy()
            print("  canny_full", key, "edges", float(e.sum()), "nan", int(np.isnan(x.grad.numpy()).sum()))
        out[name + "__x"] = x_np
        out[name + "__alpha_low_high"] = np.array([alpha, low, high], np.float64)
    save("canny_full_unpinned", **out)


# --------------------------------------------------------------------------
# 8. the rest of the targeted family (attacks.py:59-86, 337-357, 481-518): all three take `device`
# --------------------------------------------------------------------------
def gen_targeted():
    out = {}
    B, C, HW, K = 3, 2, 8, 10
    torch.manual_seed(61)
    x0 = torch.rand(B, C, HW, HW)
    x0[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 0.001, 0.999])
    y = torch.tensor([2, 9, 5])
    eps, alpha, steps = 0.062745098039216, 0.007843137254902, 5
    out["x0"], out["y"] = x0.numpy(), y.numpy()
    out["eps_alpha"] = np.array([eps, alpha], np.float64)

    # (a) targeted_PGD_trick: ONE Bernoulli per batch (attacks.py:69-71); find a seed for each outcome
    want = {True: None, False: None}
    seed = 400
    while any(v is None for v in want.values()):
        torch.manual_seed(seed)
        torch.randint(low=1, high=K, size=y.shape)
        torch.Tensor(x0.shape).uniform_(-eps, eps)
        outcome = bool(torch.gt(torch.rand([]), 0.5))
        if want[outcome] is None:
            want[outcome] = seed
        seed += 1
    for outcome, sd in want.items():
        tag = "trick_noise" if outcome else "trick_clean"
        rec = Recorder(TinyNet(C, HW, K, 71))
        torch.manual_seed(sd)
        xa, tl = rattacks.targeted_PGD_trick(rec, Args(random=True, epsilon=eps, prob_start_from_clean=0.5), x0, y, steps,
                                             alpha, K, "cpu")
        torch.manual_seed(sd)  # replay the generator calls of attacks.py:66, :69, :70 in their order
        off = torch.randint(low=1, high=K, size=y.shape)
        init = torch.Tensor(x0.shape).uniform_(-eps, eps)
        u = torch.rand([])
        assert bool(torch.gt(u, 0.5)) == outcome
        out[tag + "_offset"], out[tag + "_init"], out[tag + "_u"] = off.numpy(), init.numpy(), np.float32(u.item())
        out[tag + "_target"] = tl.numpy()
        out[tag + "_xs"], out[tag + "_gs"], out[tag + "_final"] = np.stack(rec.xs), np.stack(rec.gs), xa.numpy()
    # args.random = False: no generator call besides the label offset
    rec = Recorder(TinyNet(C, HW, K, 71))
    torch.manual_seed(410)
    xa, tl = rattacks.targeted_PGD_trick(rec, Args(random=False, epsilon=eps, prob_start_from_clean=0.5), x0, y, steps, alpha, K, "cpu")
    torch.manual_seed(410)
    out["trick_norand_offset"] = torch.randint(low=1, high=K, size=y.shape).numpy()
    out["trick_norand_target"], out["trick_norand_final"] = tl.numpy(), xa.numpy()

    # (b) tar_alp_imagenet: nclass 1000 hard-coded (attacks.py:341-342), start x + 0.001 * randn, never clamped before step 1
    K2 = 1000
    y2 = torch.tensor([999, 0, 517])
    rec = Recorder(TinyNet(C, HW, K2, 72))
    torch.manual_seed(420)
    xa, tl = rattacks.tar_alp_imagenet(rec, Args(epsilon=eps), x0, y2, steps, alpha, "cpu")
    torch.manual_seed(420)
    off = torch.randint(low=1, high=1000, size=y2.shape)
    nz = torch.randn(x0.shape)
    out["talp_y"], out["talp_offset"], out["talp_randn"], out["talp_target"] = y2.numpy(), off.numpy(), nz.numpy(), tl.numpy()
    out["talp_xs"], out["talp_gs"], out["talp_final"] = np.stack(rec.xs), np.stack(rec.gs), xa.numpy()

    # (c) AVmixup.tar_perturb with what the driver passes: ONE-HOT float targets (experiments_tinyimagenet.py:266-269);
    #     the "labels" fmod(onehot + randint, K) are a [B,K] float matrix that multiplies the log-probabilities (:503)
    rec = Recorder(TinyNet(C, HW, K, 73))
    av = rattacks.AVmixup(Args(random=True, epsilon=eps), 2.0, 1.0, 0.1, alpha, steps, num_classes=K, device="cpu")
    onehot = torch.eye(K)[y]
    torch.manual_seed(430)
    np.random.seed(431)
    xm, ym = av.tar_perturb(rec, x0, onehot)
    torch.manual_seed(430)
    off = torch.randint(low=1, high=K, size=onehot.shape)
    nz = torch.zeros_like(x0).uniform_(-eps, eps)
    np.random.seed(431)
    beta = np.random.beta(1.0, 1.0, [B, 1, 1, 1])
    out["tav_offset"], out["tav_noise"], out["tav_beta"] = off.numpy(), nz.numpy(), beta
    out["tav_xs"], out["tav_gs"] = np.stack(rec.xs), np.stack(rec.gs)
    out["tav_x"], out["tav_y"] = xm.numpy(), ym.numpy()
    assert ym.dtype == torch.float64 and xm.dtype == torch.float32
    save("targeted", **out)


# --------------------------------------------------------------------------
# 9. the *_Linf / L2 loops whose start is `torch.randn(shape, device='cuda')` (attacks.py:250, 291, 311, 383, 406)
# --------------------------------------------------------------------------
class _RandnOnHost:
    """This container has no ROCm device.  While active, `torch.randn(..., device='cuda')` - the ONE device-bound call in
    each of these methods - draws from the CPU generator instead; every other line of the reference runs as written."""

    def __enter__(self):
        self._orig = torch.randn

        def randn(*a, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k = dict(k, device="cpu")
            return self._orig(*a, **k)
        torch.randn = randn

    def __exit__(self, *exc):
        torch.randn = self._orig


def gen_linf():
    out = {}
    B, C, HW, K = 3, 2, 8, 10
    torch.manual_seed(81)
    x0 = torch.rand(B, C, HW, HW)
    x0[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 0.0005, 0.9995])  # x0 + 0.001 * randn leaves [0, 1] here: the start is NOT clamped
    y = torch.tensor([7, 3, 0])
    eps, alpha, steps = 0.062745098039216, 0.003921568627451, 5
    out["x0"], out["y"] = x0.numpy(), y.numpy()
    out["eps_alpha"] = np.array([eps, alpha], np.float64)
    with _RandnOnHost():
        # TRADES (attacks.py:404-418): KL batchmean against softmax of the natural logits
        net = TinyNet(C, HW, K, 91)
        rec = Recorder(net)
        logits = net(x0)
        torch.manual_seed(500)
        xa = rattacks.Trades(alpha, eps, steps, 6.0).PGD_Linf(rec, x0, logits)
        assert not rec.training  # eval() side effect
        torch.manual_seed(500)
        out["trades_randn"] = torch.randn(x0.shape).numpy()
        out["trades_logits"] = logits.detach().numpy()
        out["trades_xs"], out["trades_gs"], out["trades_final"] = np.stack(rec.xs), np.stack(rec.gs), xa.numpy()

        # TRADES L2 (attacks.py:381-401)
        rec = Recorder(TinyNet(C, HW, K, 91))
        torch.manual_seed(501)
        xa = rattacks.Trades(0.05, 0.03, steps, 6.0).PGD_L2(rec, x0, logits)
        torch.manual_seed(501)
        out["tradesl2_randn"] = torch.randn(x0.shape).numpy()
        out["tradesl2_step_eps"] = np.array([0.05, 0.03], np.float64)
        out["tradesl2_xs"], out["tradesl2_gs"], out["tradesl2_final"] = np.stack(rec.xs), np.stack(rec.gs), xa.numpy()

        # ALP (attacks.py:247-261): mean cross-entropy
        rec = Recorder(TinyNet(C, HW, K, 92))
        torch.manual_seed(502)
        xa = rattacks.ALP(alpha, eps, steps, 1.0).PGD_Linf(rec, x0, y)
        torch.manual_seed(502)
        out["alp_randn"] = torch.randn(x0.shape).numpy()
        out["alp_xs"], out["alp_gs"], out["alp_final"] = np.stack(rec.xs), np.stack(rec.gs), xa.numpy()

        # targeted_ALP (attacks.py:288-322): both methods
        rec = Recorder(TinyNet(C, HW, K, 93))
        tal = rattacks.targeted_ALP(alpha, eps, steps, 1.0, n_class=K)
        torch.manual_seed(503)
        xa = tal.tarPGD_Linf(rec, x0, y, "cpu")
        torch.manual_seed(503)
        off = torch.randint(low=1, high=K, size=y.shape)
        out["talpc_offset"], out["talpc_randn"] = off.numpy(), torch.randn(x0.shape).numpy()
        out["talpc_xs"], out["talpc_gs"], out["talpc_final"] = np.stack(rec.xs), np.stack(rec.gs), xa.numpy()
        rec = Recorder(TinyNet(C, HW, K, 93))
        torch.manual_seed(504)
        xa = tal.PGD_Linf(rec, x0, y)
        torch.manual_seed(504)
        out["talpu_randn"] = torch.randn(x0.shape).numpy()
        out["talpu_final"] = xa.numpy()
    save("linf_loops", **out)


# --------------------------------------------------------------------------
# 10. Add_Square (core.py:589-655): the reference's forward with its own draws recorded
# --------------------------------------------------------------------------
class _CudaIsHost:
    """`Add_Square` calls `.cuda()` on freshly drawn CPU tensors (core.py:600, 604, 646).  While active, Tensor.cuda returns
    the tensor itself, so the reference's forward runs on the host with the CPU generator; nothing else is replaced."""

    def __enter__(self):
        self._orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda t, *a, **k: t

    def __exit__(self, *exc):
        torch.Tensor.cuda = self._orig


def gen_add_square():
    out = {}
    cases = {"tiny": (3, 3, 64, 16.0 / 255, 1, 600), "mnist": (2, 1, 28, 0.3, 1, 601), "nq12": (2, 3, 16, 16.0 / 255, 12, 602),
             "nq60_resc": (2, 3, 16, 0.1, 60, 603)}
    for tag, (B, C, n, eps, nq, seed) in cases.items():
        mod = rcore.Add_Square(channels=C, size=n, epsilon=eps, n_queries=nq, rescale_schedule=tag.endswith("resc"))
        choices, ints = [], []
        rc, ri = mod.random_choice, mod.random_int

        def random_choice(shape, rc=rc):
            t = rc(shape)
            choices.append(t.numpy().copy())
            return t

        def random_int(low=0, high=1, shape=[1], ri=ri):
            t = ri(low, high, shape)
            ints.append((int(t.item()), int(high)))
            return t
        mod.random_choice, mod.random_int = random_choice, random_int
        torch.manual_seed(seed)
        x = torch.rand(B, C, n, n)
        x[0, 0, 0, :6] = torch.tensor([0.0, 1.0, eps, 1 - eps, eps / 2, 1 - eps / 2])
        xg = x.clone().requires_grad_(True)
        with _CudaIsHost():
            y = mod(xg)
        g = torch.Generator().manual_seed(seed + 50)
        u = torch.randn(x.shape, generator=g)
        (y * u).sum().backward()
        assert len(choices) == nq + 1 and len(ints) == nq
        out[tag + "__cfg"] = np.array([B, C, n, nq, int(tag.endswith("resc"))], np.int64)
        out[tag + "__eps"] = np.float64(eps)
        out[tag + "__x"], out[tag + "__y"], out[tag + "__u"], out[tag + "__gx"] = x.numpy(), y.detach().numpy(), u.numpy(), xg.grad.numpy()
        out[tag + "__stripe"] = choices[0]                                   # [B,C,1,n]  (core.py:637)
        out[tag + "__sq_sign"] = np.stack(choices[1:]).reshape(nq, C)        # [nq,C]     (core.py:648)
        out[tag + "__sq_pos"] = np.array([v for v, _ in ints], np.int64)     # [nq]       (core.py:645)
        out[tag + "__sq_size"] = np.array([n - hi for _, hi in ints], np.int32)  # s = h - high  (core.py:644-645)
        print("  add_square", tag, "pos", out[tag + "__sq_pos"][:4], "size", out[tag + "__sq_size"][:4])
    save("add_square", **out)


# --------------------------------------------------------------------------
# 11. free-AT (ImageNet/free_imagenet/AT_free_imagenet_ddp.py:263-309): the reference's own train()
# --------------------------------------------------------------------------
class TinyBNNet(nn.Module):
    """TinyNet with a train-mode BatchNorm between the convolution and the ReLU (free-AT keeps model.train(), :277)."""

    def __init__(self, cin, hw, ncls, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w1 = nn.Parameter(torch.randn(8, cin, 3, 3, generator=g) * 0.5)
        self.bn = nn.BatchNorm2d(8)
        self.w2 = nn.Parameter(torch.randn(ncls, 8 * (hw // 2) * (hw // 2), generator=g) * 0.2)

    def forward(self, x):
        h = F.relu(self.bn(F.conv2d(x, self.w1, padding=1)))
        h = F.avg_pool2d(h, 2)
        return F.linear(h.flatten(1), self.w2)


def _reference_free_at_train(ns):
    """The script cannot be imported: it parses argv, asks `managpu` for GPUs and imports a name that does not exist
    (`from utils.core import PGD`, :16) at module level.  Its train() (:263-309) is self-contained, so ONLY that FunctionDef
    is taken from the parsed source and compiled, unmodified, in a namespace holding what it reads: `args`, the module-level
    `global_noise_data` (:261), `Variable`, `time`, and AverageMeter / accuracy of the reference's own utils/helper.py."""
    import ast
    path = os.path.join(REF, "ImageNet", "free_imagenet", "AT_free_imagenet_ddp.py")
    tree = ast.parse(open(path).read(), path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "train"]
    assert len(fn) == 1 and (fn[0].lineno, fn[0].end_lineno) == (263, 324), (fn[0].lineno, fn[0].end_lineno)
    mod = ast.Module(body=fn, type_ignores=[])
    exec(compile(mod, path, "exec"), ns)
    return ns["train"]


def gen_free_at():
    import time
    from torch.autograd import Variable
    sys.modules.setdefault("easydict", types.ModuleType("easydict")).EasyDict = dict  # utils/helper.py:9, never called here
    import utils.helper as rhelper
    out = {}
    B, C, HW, K = 3, 2, 8, 10
    a = e = 4.0 / 255  # AT_free_imagenet_ddp.py:130-131 with the defaults of :91-99
    for tag, net_cls, seed in (("plain", TinyNet, 3), ("bn", TinyBNNet, 4)):
        torch.manual_seed(700 + seed)
        batches = []
        for b in range(2):
            x = torch.rand(B, C, HW, HW)
            x[0, 0, 0, :2] = torch.tensor([0.0, 1.0])  # x + delta leaves [0, 1] here: the in-place clamp masks the gradient
            batches.append((x, torch.randint(0, K, (B,))))
        batches.append((torch.rand(2, C, HW, HW), torch.randint(0, K, (2,))))  # a short last batch: rows 2.. are only clamped
        net = net_cls(C, HW, K, seed)
        rec = Recorder(net)
        noise0 = torch.zeros(B + 1, C, HW, HW)
        noise0[B:] = 0.5  # a row beyond every batch: only the buffer-wide clamp_ (:307) touches it
        ns = {"args": Args(n_repeats=4, fgsm_step=a, clip_eps=e, local_rank=0), "global_noise_data": noise0.clone(),
              "Variable": Variable, "time": time, "torch": torch, "AverageMeter": rhelper.AverageMeter, "accuracy": rhelper.accuracy}
        train = _reference_free_at_train(ns)
        deltas, weights, bn_means = [], [], []

        class RecSGD(torch.optim.SGD):
            def step(self, closure=None):  # :309, the last statement of a repeat: delta already moved (:306-307)
                r = super().step(closure)
                deltas.append(ns["global_noise_data"].clone().numpy())
                weights.append(torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy())
                if hasattr(net, "bn"):
                    bn_means.append(net.bn.running_mean.clone().numpy())
                return r
        opt = RecSGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
        logits = []
        fwd = rec.forward
        rec.forward = lambda x: (lambda z: (logits.append(z.detach().numpy().copy()), z)[1])(fwd(x))
        train(batches, rec, nn.CrossEntropyLoss(), opt, 0, 1, "cpu", None)
        assert rec.training and len(deltas) == 12 and len(rec.gs) == 12
        out[tag + "__noise0"] = noise0.numpy()
        for b, (x, y) in enumerate(batches):
            out[tag + "__x%d" % b], out[tag + "__y%d" % b] = x.numpy(), y.numpy()
        out[tag + "__step_eps"] = np.array([a, e], np.float64)
        out[tag + "__sgd"] = np.array([0.05, 0.9, 1e-4], np.float64)
        out[tag + "__deltas"] = np.stack(deltas)          # [12, B+1, C, H, W]: the buffer after every repeat
        out[tag + "__weights"] = np.stack(weights)        # [12, P]: all parameters after every optimizer.step()
        for i in range(12):  # in1 of each repeat, the gradient that reached it (dL/din1, BEFORE the clamp mask), the logits
            out[tag + "__in1_%d" % i], out[tag + "__gin1_%d" % i], out[tag + "__logits_%d" % i] = rec.xs[i], rec.gs[i], logits[i]
        if bn_means:
            out[tag + "__bn_running_mean"] = np.stack(bn_means)
        print("  free_at", tag, "delta range", float(out[tag + "__deltas"][-1][:B].min()), float(out[tag + "__deltas"][-1][:B].max()))
    save("freeat", **out)


# --------------------------------------------------------------------------
# 12. AWP (AWP/Tiny_imagenet/models_tiny_awp/utils_awp.py + the step of experiments_tiny_awp.py:256-286)
# --------------------------------------------------------------------------
class TinyModuleNet(nn.Module):
    """conv (with bias) -> BatchNorm -> ReLU -> avgpool2 -> linear, built from nn modules so that the state_dict keys read
    `conv.weight`, `bn.weight`, `fc.weight` ...: AWP (utils_awp.py:8-18) perturbs the entries named '*weight*' with more than one dimension."""

    def __init__(self, cin, hw, ncls, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.conv = nn.Conv2d(cin, 8, 3, padding=1)
        self.bn = nn.BatchNorm2d(8)
        self.fc = nn.Linear(8 * (hw // 2) * (hw // 2), ncls)

    def forward(self, x):
        h = F.relu(self.bn(self.conv(x)))
        return self.fc(F.avg_pool2d(h, 2).flatten(1))


def gen_awp():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_utils_awp", os.path.join(REF, "AWP", "Tiny_imagenet", "models_tiny_awp", "utils_awp.py"))
    rawp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rawp)  # torch only
    out = {}
    B, C, HW, K = 4, 2, 8, 10
    torch.manual_seed(90)
    x = torch.rand(B, C, HW, HW)
    y = torch.randint(0, K, (B,))
    eps, alpha, steps, gamma = 0.062745098039216, 0.007843137254902, 3, 0.01
    args = Args(random=False, epsilon=eps)
    net, proxy = TinyModuleNet(C, HW, K, 95), TinyModuleNet(C, HW, K, 96)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=2e-4)
    proxy_opt = torch.optim.SGD(proxy.parameters(), lr=0.01)  # experiments_tiny_awp.py:118
    adv = rawp.AdvWeightPerturb(model=net, proxy=proxy, proxy_optim=proxy_opt, gamma=gamma)
    crit = nn.CrossEntropyLoss()
    flat = lambda m: torch.cat([p.detach().reshape(-1) for p in m.parameters()]).numpy().copy()
    rec = {k: [] for k in ("adv", "diff_w1", "diff_w2", "perturbed", "loss", "logits", "after", "bn_mean")}
    for step in range(3):  # the loop body of :256-286 with the reference's own PGD and AdvWeightPerturb
        net.train()
        data_adv = rattacks.PGD(net, args, x, y, steps, alpha)
        awp = adv.calc_awp(inputs_adv=data_adv, targets=y)
        adv.perturb(awp)
        rec["perturbed"].append(flat(net))
        robust_output = net(data_adv)
        robust_loss = crit(robust_output, y)
        opt.zero_grad()
        robust_loss.backward()
        opt.step()
        adv.restore(awp)
        assert list(awp.keys()) == ["conv.weight", "fc.weight"]  # bn.weight and the biases are 1-d: skipped (:13-14)
        rec["adv"].append(data_adv.numpy().copy())
        rec["diff_w1"].append(awp["conv.weight"].numpy().copy())
        rec["diff_w2"].append(awp["fc.weight"].numpy().copy())
        rec["loss"].append(float(robust_loss.detach()))
        rec["logits"].append(robust_output.detach().numpy().copy())
        rec["after"].append(flat(net))
        rec["bn_mean"].append(net.bn.running_mean.numpy().copy())
    out["x"], out["y"] = x.numpy(), y.numpy()
    out["cfg"] = np.array([eps, alpha, steps, gamma, 0.1, 0.9, 2e-4, 0.01], np.float64)
    for k, v in rec.items():
        out[k] = np.stack(v) if k != "loss" else np.array(v, np.float64)
    save("awp", **out)


if __name__ == "__main__":
    torch.set_num_threads(1)
    gen_kernels()
    gen_edge125()
    gen_pgd()
    gen_losses()
    gen_avmix_cw()
    gen_e2e()
    gen_canny_full()
    gen_targeted()
    gen_linf()
    gen_add_square()
    gen_free_at()
    gen_awp()
