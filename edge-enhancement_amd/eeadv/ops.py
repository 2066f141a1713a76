"""Tensor-level wrappers over the C ABI (include/eeadv.h).

Every function takes ROCm tensors, checks device / dtype / contiguity / shape on the host (a wrong shape
must never reach a kernel), and enqueues ONE library call on torch's current HIP stream.  Nothing here
computes on the CPU and nothing falls back to torch ops: a CPU tensor raises.
"""
import ctypes

import numpy as np
import torch

from . import _native as N

_INF = float("inf")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype, name, shape=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise N.EEError("%s is on %s: the eeadv kernels only run on a ROCm device (no CPU fallback)" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return ctypes.c_void_p(t.data_ptr())


def _opt(t, dtype, name, shape=None):
    return None if t is None else _chk(t, dtype, name, shape)


# ---- fixed 3x3 weights (utils/core.py:58-84, restated in numpy float64 then cast like core.py:164,177,180) ----
def gaussian_kernel_np(k=3, mu=0, sigma=1, normalize=True):
    g1 = np.linspace(-1, 1, k)
    x, y = np.meshgrid(g1, g1)
    d = (x ** 2 + y ** 2) ** 0.5
    g2 = np.exp(-(d - mu) ** 2 / (2 * sigma ** 2))
    g2 = g2 / (2 * np.pi * sigma ** 2)
    if normalize:
        g2 = g2 / np.sum(g2)
    return g2


def sobel_kernel_np(k=3):
    rng = np.linspace(-(k // 2), k // 2, k)
    x, y = np.meshgrid(rng, rng)
    den = x ** 2 + y ** 2
    den[:, k // 2] = 1
    return x / den


# k*45 degrees -> (drow, dcol) of the -1 tap of the directional kernel relative to its centre (core.py:87-112).
# DERIVED (the reference rotates [0,0,1,-1,-1] with cv2, which is unavailable): parity unpinned.
CANNY_DIRS = np.array([(0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1)], dtype=np.int32)


class EdgeWeights:
    """The 27 host floats every edge kernel takes (Gaussian, Sobel-x, Sobel-y) + the full filter's direction table."""

    def __init__(self, sigma=1.0, mu=0.0):
        self.dirs = np.ascontiguousarray(CANNY_DIRS.reshape(-1), dtype=np.int32)
        self.dirs_ptr = self.dirs.ctypes.data_as(ctypes.c_void_p)
        g = gaussian_kernel_np(3, mu, sigma).astype(np.float32).reshape(9)
        s = sobel_kernel_np(3)
        self.host = np.concatenate([g, s.astype(np.float32).reshape(9), s.T.astype(np.float32).reshape(9)])
        self.host = np.ascontiguousarray(self.host, dtype=np.float32)
        self.ptr = self.host.ctypes.data_as(ctypes.c_void_p)


# ---- PGD family ------------------------------------------------------------------------------------------
def pgd_init(x0, noise, lo=0.0, hi=1.0, out=None):
    p0 = _chk(x0, torch.float32, "x0")
    pn = _chk(noise, torch.float32, "noise", x0.shape)
    out = torch.empty_like(x0) if out is None else out
    po = _chk(out, torch.float32, "out", x0.shape)
    N.check(N.lib.ee_pgd_init_f32(po, p0, pn, x0.numel(), lo, hi, _stream()), "ee_pgd_init_f32")
    return out


def pgd_init_rng(x0, scale, dist, seed, offset, lo=0.0, hi=1.0, out=None):
    p0 = _chk(x0, torch.float32, "x0")
    out = torch.empty_like(x0) if out is None else out
    po = _chk(out, torch.float32, "out", x0.shape)
    N.check(N.lib.ee_pgd_init_rng_f32(po, p0, x0.numel(), scale, dist, seed, offset, lo, hi, _stream()),
            "ee_pgd_init_rng_f32")
    return out


def pgd_step_(x, g, x0, alpha, eps, lo=0.0, hi=1.0, direction=1):
    """In place on x (attacks.py:25-27)."""
    px = _chk(x, torch.float32, "x")
    pg = _chk(g, torch.float32, "g", x.shape)
    p0 = _chk(x0, torch.float32, "x0", x.shape)
    N.check(N.lib.ee_pgd_step_f32(px, pg, p0, x.numel(), alpha, eps, lo, hi, direction, _stream()), "ee_pgd_step_f32")
    return x


def pgd_step_bcast_(x, g_lp, g_edge, x0, alpha, eps, lo=0.0, hi=1.0, direction=1):
    B, C, H, W = x.shape
    px = _chk(x, torch.float32, "x")
    pl = _chk(g_lp, torch.float32, "g_lp", x.shape)
    pe = _chk(g_edge, torch.float32, "g_edge", (B, 1, H, W))
    p0 = _chk(x0, torch.float32, "x0", x.shape)
    N.check(N.lib.ee_pgd_step_bcast_f32(px, pl, pe, p0, B, C, H * W, alpha, eps, lo, hi, direction, _stream()),
            "ee_pgd_step_bcast_f32")
    return x


def l2_step_(x, g, x0, step, eps, lo=0.0, hi=1.0):
    """In place on x [B, ...]: the TRADES L2 update (attacks.py:389-400)."""
    B = x.shape[0]
    px = _chk(x, torch.float32, "x")
    pg = _chk(g, torch.float32, "g", x.shape)
    p0 = _chk(x0, torch.float32, "x0", x.shape)
    if B:
        N.check(N.lib.ee_l2_step_f32(px, pg, p0, B, x.numel() // B, step, eps, lo, hi, _stream()), "ee_l2_step_f32")
    return x


def fgsm_step(x, g, alpha, lo=0.0, hi=1.0, direction=1):
    px = _chk(x, torch.float32, "x")
    pg = _chk(g, torch.float32, "g", x.shape)
    out = torch.empty_like(x)
    N.check(N.lib.ee_fgsm_step_f32(_chk(out, torch.float32, "out"), px, pg, x.numel(), alpha, lo, hi, direction, _stream()),
            "ee_fgsm_step_f32")
    return out


def add_clamp(x, delta, lo=0.0, hi=1.0):
    px = _chk(x, torch.float32, "x")
    pd = _chk(delta, torch.float32, "delta", x.shape)
    out = torch.empty_like(x)
    N.check(N.lib.ee_add_clamp_f32(_chk(out, torch.float32, "out"), px, pd, x.numel(), lo, hi, _stream()), "ee_add_clamp_f32")
    return out


def freeat_update_(delta, g, alpha, eps):
    """delta[:n] updated in place, n = g.numel() (the live rows of the persistent noise buffer)."""
    pd = _chk(delta, torch.float32, "delta")
    pg = _chk(g, torch.float32, "g")
    if g.numel() > delta.numel():
        raise ValueError("gradient has more elements than the noise buffer")
    N.check(N.lib.ee_freeat_update_f32(pd, pg, g.numel(), alpha, eps, _stream()), "ee_freeat_update_f32")
    return delta


def freeat_update_masked_(delta, g_in1, x, alpha, eps):
    """delta[:n] += alpha*sign(g_in1 * 1[0 <= x + delta <= 1]); clamp(+-eps); n = x.numel()."""
    pd = _chk(delta, torch.float32, "delta")
    pg = _chk(g_in1, torch.float32, "g_in1", x.shape)
    px = _chk(x, torch.float32, "x")
    if x.numel() > delta.numel():
        raise ValueError("batch has more elements than the noise buffer")
    N.check(N.lib.ee_freeat_update_masked_f32(pd, pg, px, x.numel(), alpha, eps, _stream()), "ee_freeat_update_masked_f32")
    return delta


def avmix(x, x0, wgt, gamma):
    B = x.shape[0]
    px = _chk(x, torch.float32, "x")
    p0 = _chk(x0, torch.float32, "x0", x.shape)
    pw = _chk(wgt, torch.float64, "wgt", (B,))
    out = torch.empty_like(x)
    N.check(N.lib.ee_avmix_f32(_chk(out, torch.float32, "out"), px, p0, pw, B, x.numel() // max(B, 1), gamma, _stream()),
            "ee_avmix_f32")
    return out


def avmix_labels(labels, wgt, K, lambda1, lambda2):
    B = labels.shape[0]
    pl = _chk(labels, torch.int64, "labels", (B,))
    pw = _chk(wgt, torch.float64, "wgt", (B,))
    out = torch.empty((B, K), dtype=torch.float64, device=labels.device)
    N.check(N.lib.ee_avmix_labels_f64(_chk(out, torch.float64, "out"), pl, pw, B, K, lambda1, lambda2, _stream()),
            "ee_avmix_labels_f64")
    return out


# ---- edge filter / front end ---------------------------------------------------------------------------------
def edge125_fwd(x, wts, alpha, high, want_mag=False):
    B, C, H, W = x.shape
    px = _chk(x, torch.float32, "x")
    edge = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    mag = torch.empty_like(edge) if want_mag else None
    N.check(N.lib.ee_edge125_fwd_f32(px, B, C, H, W, wts.ptr, alpha, high, _chk(edge, torch.float32, "edge"),
                                     _opt(mag, torch.float32, "mag"), _stream()), "ee_edge125_fwd_f32")
    return (edge, mag) if want_mag else edge


def edge125_bwd(x, u, wts, alpha, high):
    B, C, H, W = x.shape
    px = _chk(x, torch.float32, "x")
    pu = _chk(u, torch.float32, "u", (B, 1, H, W))
    g = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_edge125_bwd_f32(px, pu, B, C, H, W, wts.ptr, alpha, high, _chk(g, torch.float32, "g"), _stream()),
            "ee_edge125_bwd_f32")
    return g


def frontend_fwd(x, x_hfs, wts, alpha, high, w, want_edge=False):
    B, C, H, W = x.shape
    px = _chk(x, torch.float32, "x")
    ph = _chk(x_hfs, torch.float32, "x_hfs", x.shape)
    x_in = torch.empty_like(x)
    gate = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    edge = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device) if want_edge else None
    N.check(N.lib.ee_frontend_fwd_f32(px, ph, B, C, H, W, wts.ptr, alpha, high, w, _chk(x_in, torch.float32, "x_in"),
                                      _chk(gate, torch.uint8, "gate"), _opt(edge, torch.float32, "edge"), _stream()),
            "ee_frontend_fwd_f32")
    return x_in, gate, edge


def frontend_bwd(g_in, gate, x, wts, alpha, high, w):
    B, C, H, W = x.shape
    pg = _chk(g_in, torch.float32, "g_in", x.shape)
    pt = _chk(gate, torch.uint8, "gate", x.shape)
    px = _chk(x, torch.float32, "x")
    g_hfs = torch.empty_like(x)
    g_edge = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_frontend_bwd_f32(pg, pt, px, B, C, H, W, wts.ptr, alpha, high, w, _chk(g_hfs, torch.float32, "g_hfs"),
                                      _chk(g_edge, torch.float32, "g_edge"), _stream()), "ee_frontend_bwd_f32")
    return g_hfs, g_edge


def frontend_fwd_save(x, x_hfs, wts, alpha, high, w, want_edge=False):
    """frontend_fwd that also keeps the Sobel responses for frontend_bwd_saved: returns (x_in, gate, edge, gx, gy)."""
    B, C, H, W = x.shape
    px = _chk(x, torch.float32, "x")
    ph = _chk(x_hfs, torch.float32, "x_hfs", x.shape)
    x_in = torch.empty_like(x)
    gate = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    edge = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device) if want_edge else None
    gx = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    gy = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_frontend_fwd_save_f32(px, ph, B, C, H, W, wts.ptr, alpha, high, w, _chk(x_in, torch.float32, "x_in"),
                                           _chk(gate, torch.uint8, "gate"), _opt(edge, torch.float32, "edge"), gx.data_ptr(), gy.data_ptr(),
                                           _stream()), "ee_frontend_fwd_save_f32")
    return x_in, gate, edge, gx, gy


def frontend_bwd_saved(g_in, gate, gx, gy, wts, alpha, high, w):
    """Backward of the front end from the saved Sobel responses (no x, no recomputation): (g_hfs, g_edge), bit-identical to
    frontend_bwd."""
    B, C, H, W = g_in.shape
    pg = _chk(g_in, torch.float32, "g_in")
    pt = _chk(gate, torch.uint8, "gate", g_in.shape)
    g_hfs = torch.empty_like(g_in)
    g_edge = torch.empty((B, 1, H, W), dtype=torch.float32, device=g_in.device)
    N.check(N.lib.ee_frontend_bwd_saved_f32(pg, pt, _chk(gx, torch.float32, "gx", (B, 1, H, W)), _chk(gy, torch.float32, "gy", (B, 1, H, W)),
                                            B, C, H, W, wts.ptr, alpha, high, w, g_hfs.data_ptr(), g_edge.data_ptr(), _stream()),
            "ee_frontend_bwd_saved_f32")
    return g_hfs, g_edge


def canny_fwd(x, wts, alpha, low, high):
    B, C, H, W = x.shape
    edge = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_canny_fwd_f32(_chk(x, torch.float32, "x"), None, B, C, H, W, wts.ptr, wts.dirs_ptr, alpha, low, high, 0.0,
                                   _chk(edge, torch.float32, "edge"), None, None, _stream()), "ee_canny_fwd_f32")
    return edge


def canny_bwd(x, u, wts, alpha, low, high):
    B, C, H, W = x.shape
    g = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_canny_bwd_f32(_chk(x, torch.float32, "x"), _chk(u, torch.float32, "u", (B, 1, H, W)), None, None, B, C, H, W, wts.ptr,
                                   wts.dirs_ptr, alpha, low, high, 0.0, _chk(g, torch.float32, "g"), None, _stream()), "ee_canny_bwd_f32")
    return g


def canny_bpda_fwd(x, wts, low, high):
    """CannyFilter_BPDA forward (core.py:426-505) -> (edge, thin, t2); thin / t2 are the backward's saved state."""
    B, C, H, W = x.shape
    edge, thin, t2 = (torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device) for _ in range(3))
    N.check(N.lib.ee_canny_bpda_fwd_f32(_chk(x, torch.float32, "x"), B, C, H, W, wts.ptr, wts.dirs_ptr, low, high, edge.data_ptr(),
                                        thin.data_ptr(), t2.data_ptr(), _stream()), "ee_canny_bpda_fwd_f32")
    return edge, thin, t2


def canny_bpda_bwd(x, u, thin, t2, wts, low, high):
    B, C, H, W = x.shape
    g = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    scratch = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_canny_bpda_bwd_f32(_chk(x, torch.float32, "x"), _chk(u, torch.float32, "u", (B, 1, H, W)),
                                        _chk(thin, torch.float32, "thin", (B, 1, H, W)), _chk(t2, torch.float32, "t2", (B, 1, H, W)), B, C, H,
                                        W, wts.ptr, wts.dirs_ptr, low, high, scratch.data_ptr(), g.data_ptr(), _stream()),
            "ee_canny_bpda_bwd_f32")
    return g


def canny_frontend_fwd(x, x_hfs, wts, alpha, low, high, w, want_edge=False):
    B, C, H, W = x.shape
    x_in = torch.empty_like(x)
    gate = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    edge = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device) if want_edge else None
    N.check(N.lib.ee_canny_fwd_f32(_chk(x, torch.float32, "x"), _chk(x_hfs, torch.float32, "x_hfs", x.shape), B, C, H, W, wts.ptr,
                                   wts.dirs_ptr, alpha, low, high, w, _opt(edge, torch.float32, "edge"), _chk(x_in, torch.float32, "x_in"),
                                   _chk(gate, torch.uint8, "gate"), _stream()), "ee_canny_fwd_f32")
    return x_in, gate, edge


def canny_frontend_bwd(g_in, gate, x, wts, alpha, low, high, w):
    B, C, H, W = x.shape
    g_hfs = torch.empty_like(x)
    g_edge = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_canny_bwd_f32(_chk(x, torch.float32, "x"), None, _chk(g_in, torch.float32, "g_in", x.shape),
                                   _chk(gate, torch.uint8, "gate", x.shape), B, C, H, W, wts.ptr, wts.dirs_ptr, alpha, low, high, w,
                                   _chk(g_edge, torch.float32, "g_edge"), _chk(g_hfs, torch.float32, "g_hfs"), _stream()),
            "ee_canny_bwd_f32")
    return g_hfs, g_edge


# ---- losses ------------------------------------------------------------------------------------------------------
def _inv(n):
    """1/n with torch's semantics for an empty reduction: the mean of nothing is NaN (0 * NaN in the reduce kernel)."""
    return 1.0 / n if n else float("nan")


def _reduce_rows(rows, scale):
    out = torch.empty(1, dtype=torch.float64, device=rows.device)
    N.check(N.lib.ee_reduce_rows_f64(_chk(rows, torch.float64, "rows"), rows.numel(), scale, _chk(out, torch.float64, "out"),
                                     _stream()), "ee_reduce_rows_f64")
    return out


def ce(logits, labels, reduction="mean", smoothing=0.0, want_loss=True, want_grad=True):
    """Returns (loss: 0-dim fp32 tensor or None, dlogits or None)."""
    B, K = logits.shape
    pz = _chk(logits, torch.float32, "logits")
    py = _chk(labels, torch.int64, "labels", (B,))
    gscale = _inv(B) if reduction == "mean" else 1.0
    rows = torch.empty(B, dtype=torch.float64, device=logits.device) if want_loss else None
    d = torch.empty_like(logits) if want_grad else None
    N.check(N.lib.ee_ce_f32(pz, py, B, K, smoothing, gscale, _opt(rows, torch.float64, "rows"), _opt(d, torch.float32, "d"),
                            _stream()), "ee_ce_f32")
    loss = _reduce_rows(rows, gscale)[0].to(torch.float32) if want_loss else None
    return loss, d


def fc_ce_grad_supported(z1, w2):
    return z1.dim() == 2 and w2.dim() == 2 and w2.shape[1] == z1.shape[1] and w2.shape[0] <= 64 and z1.shape[1] <= 8192


def fc_ce_grad(z1, w2, b2, labels, reduction="sum", want_logits=False):
    """d CrossEntropyLoss(fc2(relu(z1)), labels) / d z1 in one launch (ee_loss.hip: fc_ce_grad_kernel): z1 [B,Hd], w2 [K,Hd], b2 [K] or None.
    Returns dz1, or (dz1, logits)."""
    B, Hd = z1.shape
    K = w2.shape[0]
    dz = torch.empty_like(z1)
    lg = torch.empty((B, K), dtype=torch.float32, device=z1.device) if want_logits else None
    N.check(N.lib.ee_fc_ce_grad_f32(_chk(z1, torch.float32, "z1"), _chk(w2, torch.float32, "w2", (K, Hd)), _opt(b2, torch.float32, "b2"),
                                    _chk(labels, torch.int64, "labels", (B,)), dz.data_ptr(), _opt(lg, torch.float32, "logits"), B, Hd, K,
                                    _inv(B) if reduction == "mean" else 1.0, _stream()), "ee_fc_ce_grad_f32")
    return (dz, lg) if want_logits else dz


def kl_batchmean(zq, zp, want_loss=True, want_dq=True, want_dp=False):
    B, K = zq.shape
    pq = _chk(zq, torch.float32, "zq")
    pp = _chk(zp, torch.float32, "zp", zq.shape)
    rows = torch.empty(B, dtype=torch.float64, device=zq.device) if want_loss else None
    dq = torch.empty_like(zq) if want_dq else None
    dp = torch.empty_like(zq) if want_dp else None
    N.check(N.lib.ee_kl_f32(pq, pp, B, K, _inv(B), _opt(rows, torch.float64, "rows"), _opt(dq, torch.float32, "dq"),
                            _opt(dp, torch.float32, "dp"), _stream()), "ee_kl_f32")
    loss = _reduce_rows(rows, _inv(B))[0].to(torch.float32) if want_loss else None
    return loss, dq, dp


def softce(z, t, scale, want_loss=True, want_grad=True):
    """-sum(log_softmax(z) * t) * scale with float64 targets; loss and gradient are float64."""
    B, K = z.shape
    pz = _chk(z, torch.float32, "z")
    pt = _chk(t, torch.float64, "t", z.shape)
    rows = torch.empty(B, dtype=torch.float64, device=z.device) if want_loss else None
    dz = torch.empty((B, K), dtype=torch.float64, device=z.device) if want_grad else None
    N.check(N.lib.ee_softce_f64(pz, pt, B, K, scale, _opt(rows, torch.float64, "rows"), _opt(dz, torch.float64, "dz"),
                                _stream()), "ee_softce_f64")
    loss = _reduce_rows(rows, scale)[0] if want_loss else None
    return loss, dz


def mse(a, b, want_loss=True, want_grad=True):
    pa = _chk(a, torch.float32, "a")
    pb = _chk(b, torch.float32, "b", a.shape)
    n = a.numel()
    nb = N.lib.ee_mse_num_partials(n)
    part = torch.empty(nb, dtype=torch.float64, device=a.device) if want_loss else None
    da = torch.empty_like(a) if want_grad else None
    N.check(N.lib.ee_mse_f32(pa, pb, n, 2.0 * _inv(n), _opt(part, torch.float64, "part"), _opt(da, torch.float32, "da"), _stream()),
            "ee_mse_f32")
    loss = _reduce_rows(part, _inv(n))[0].to(torch.float32) if want_loss else None
    return loss, da


def topk(logits, labels, k):
    """(idx [B,k] int64, correct [k] int64 or None)."""
    B, K = logits.shape
    pz = _chk(logits, torch.float32, "logits")
    idx = torch.empty((B, k), dtype=torch.int64, device=logits.device)
    correct = torch.empty(k, dtype=torch.int64, device=logits.device) if labels is not None else None
    N.check(N.lib.ee_topk_i64(pz, _opt(labels, torch.int64, "labels", (B,)), B, K, k, _chk(idx, torch.int64, "idx"),
                              _opt(correct, torch.int64, "correct"), _stream()), "ee_topk_i64")
    return idx, correct


# ---- Add_Square ------------------------------------------------------------------------------------------------------
def add_square_fwd(x, eps, stripe, sq_sign, sq_pos, sq_size):
    B, C, H, W = x.shape
    nq = int(sq_size.numel())
    out = torch.empty_like(x)
    N.check(N.lib.ee_add_square_fwd_f32(_chk(x, torch.float32, "x"), B, C, H, W, eps, _chk(stripe, torch.float32, "stripe", (B, C, 1, W)),
                                        _chk(sq_sign, torch.float32, "sq_sign", (nq, C)), _chk(sq_pos, torch.int64, "sq_pos", (nq,)),
                                        _chk(sq_size, torch.int32, "sq_size", (nq,)), nq, _chk(out, torch.float32, "out"), _stream()),
            "ee_add_square_fwd_f32")
    return out


def add_square_bwd(g_out, x, eps, stripe, sq_sign, sq_pos, sq_size):
    B, C, H, W = x.shape
    nq = int(sq_size.numel())
    g_x = torch.empty_like(x)
    N.check(N.lib.ee_add_square_bwd_f32(_chk(g_out, torch.float32, "g_out", x.shape), _chk(x, torch.float32, "x"), B, C, H, W, eps,
                                        _chk(stripe, torch.float32, "stripe", (B, C, 1, W)),
                                        _chk(sq_sign, torch.float32, "sq_sign", (nq, C)), _chk(sq_pos, torch.int64, "sq_pos", (nq,)),
                                        _chk(sq_size, torch.int32, "sq_size", (nq,)), nq, _chk(g_x, torch.float32, "g_x"), _stream()),
            "ee_add_square_bwd_f32")
    return g_x


# ---- HighFreqSuppress -----------------------------------------------------------------------------------------------
def square_draw(batch, C, h, sq_size, state):
    """The random draws of one Add_Square forward in one launch (core.py:637, :645, :648): returns stripe [B,C,1,h] fp32,
    sq_pos [nq] int64, sq_sign [nq,C] fp32.  state: int64[2] device tensor {seed, offset}, advanced by the kernel."""
    dev = sq_size.device
    nq = int(sq_size.numel())
    stripe = torch.empty((batch, C, 1, h), dtype=torch.float32, device=dev)
    sq_pos = torch.empty((nq,), dtype=torch.int64, device=dev)
    sq_sign = torch.empty((nq, C), dtype=torch.float32, device=dev)
    N.check(N.lib.ee_square_draw_f32(stripe.data_ptr(), stripe.numel(), sq_pos.data_ptr(), sq_sign.data_ptr(),
                                     _chk(sq_size, torch.int32, "sq_size"), nq, C, h, _chk(state, torch.int64, "state", (4,)), _stream()),
            "ee_square_draw_f32")
    return stripe, sq_pos, sq_sign


def hfs(x, tables, NU, NV, sq_mode=0, sq_x=None, eps=0.0, stripe=None, sq_sign=None, sq_pos=None, sq_size=None):
    """y = F(x) (sq_mode 0), F(add_square(x)) (1) or F(x) * d add_square/dx at sq_x (2); F = the low-pass operator."""
    B, C, H, W = x.shape
    px = _chk(x, torch.float32, "x")
    need = N.lib.ee_hfs_table_floats(H, W, NU, NV)
    pt = _chk(tables, torch.float32, "tables", (need,))
    out = torch.empty_like(x)
    nq = 0
    ps = pg = pp = pz = pxo = None
    if sq_mode:
        nq = int(sq_size.numel())
        ps = _chk(stripe, torch.float32, "stripe", (B, C, 1, W))
        pg = _chk(sq_sign, torch.float32, "sq_sign", (nq, C))
        pp = _chk(sq_pos, torch.int64, "sq_pos", (nq,))
        pz = _chk(sq_size, torch.int32, "sq_size", (nq,))
        if sq_mode == 2:
            pxo = _chk(sq_x, torch.float32, "sq_x", x.shape)
    N.check(N.lib.ee_hfs_f32(px, _chk(out, torch.float32, "out"), B, C, H, W, pt, NU, NV, 1.0 / H, sq_mode, pxo, eps, ps, pg, pp, pz, nq,
                             _stream()), "ee_hfs_f32")
    return out


def hfs_mfma(x, tables, nu_pad, sq_mode=0, sq_x=None, eps=0.0, stripe=None, sq_sign=None, sq_pos=None, sq_size=None):
    """The same operator as `hfs` for planes up to 256 x 256, on the matrix cores (ee_hfs_mfma.hip)."""
    B, C, H, W = x.shape
    px = _chk(x, torch.float32, "x")
    pt = _chk(tables, torch.float32, "tables", (N.lib.ee_hfs_mfma_table_floats(H, W, nu_pad),))
    out = torch.empty_like(x)
    nq = 0
    ps = pg = pp = pz = pxo = None
    if sq_mode:
        nq = int(sq_size.numel())
        ps = _chk(stripe, torch.float32, "stripe", (B, C, 1, W))
        pg = _chk(sq_sign, torch.float32, "sq_sign", (nq, C))
        pp = _chk(sq_pos, torch.int64, "sq_pos", (nq,))
        pz = _chk(sq_size, torch.int32, "sq_size", (nq,))
        if sq_mode == 2:
            pxo = _chk(sq_x, torch.float32, "sq_x", x.shape)
    N.check(N.lib.ee_hfs_mfma_f32(px, _chk(out, torch.float32, "out"), B, C, H, W, pt, nu_pad, sq_mode, pxo, eps, ps, pg, pp, pz, nq, _stream()),
            "ee_hfs_mfma_f32")
    return out


# ---- the fused front end of one PGD iteration (ee_chain.hip) ------------------------------------------------------------------------
def chain_supported(C, H, W):
    return bool(N.lib.ee_chain_supported(int(C), int(H), int(W)))


def chain_fwd(x, tables, wts, alpha, high, w, square=False, eps=0.0, sq_size=0, state=None, draws=None, want_edge=False):
    """x [B,C,H,W] -> (x_in, gate, gx, gy, edge): clamp(hfs(add_square(x)) + w * edge125(x), 0, 1) in ONE launch.  `draws`
    (dict stripe [B,C,1,W], sq_pos [1], sq_sign [1,C]) injects the Add_Square draws; otherwise they come from the Philox `state`
    (int64[4] device tensor {seed, offset, ticket, -}, advanced by the kernel)."""
    B, C, H, W = x.shape
    px = _chk(x, torch.float32, "x")
    pt = _chk(tables, torch.float32, "tables", (N.lib.ee_chain_table_floats(H, W),))
    x_in = torch.empty_like(x)
    gate = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    gx = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    gy = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    edge = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device) if want_edge else None
    ps = pp = pg = pst = None
    if square:
        if draws is not None:
            ps = _chk(draws["stripe"], torch.float32, "stripe", (B, C, 1, W))
            pp = _chk(draws["sq_pos"], torch.int64, "sq_pos", (1,))
            pg = _chk(draws["sq_sign"], torch.float32, "sq_sign", (1, C))
        else:
            pst = _chk(state, torch.int64, "state", (4,))
    N.check(N.lib.ee_chain_fwd_f32(px, B, C, H, W, pt, wts.ptr, alpha, high, w, 1 if square else 0, eps, int(sq_size), pst, ps, pp, pg,
                                   x_in.data_ptr(), gate.data_ptr(), gx.data_ptr(), gy.data_ptr(), _opt(edge, torch.float32, "edge"), _stream()),
            "ee_chain_fwd_f32")
    return x_in, gate, gx, gy, edge


def chain_bwd_(x, g_in, gate, gx, gy, x0, tables, wts, alpha, high, w, step, eps, lo=0.0, hi=1.0, direction=1):
    """In place on x: the front end's backward (edge adjoint from gx / gy, low-pass of gate * g_in times d add_square) and the PGD
    update (attacks.py:25-27) in ONE launch."""
    B, C, H, W = x.shape
    px = _chk(x, torch.float32, "x")
    pg = _chk(g_in, torch.float32, "g_in", x.shape)
    pt = _chk(gate, torch.uint8, "gate", x.shape)
    p0 = _chk(x0, torch.float32, "x0", x.shape)
    N.check(N.lib.ee_chain_bwd_f32(pg, pt, _chk(gx, torch.float32, "gx", (B, 1, H, W)), _chk(gy, torch.float32, "gy", (B, 1, H, W)), px, p0, B, C, H, W,
                                   _chk(tables, torch.float32, "tables", (N.lib.ee_chain_table_floats(H, W),)), wts.ptr, alpha, high, w, step, eps, lo,
                                   hi, direction, _stream()), "ee_chain_bwd_f32")
    return x


# ---- BatchNorm2d (+ residual) (+ ReLU) ----------------------------------------------------------------------------------
def _bn_workspace(x, B, C, HW):
    n = N.lib.ee_bn_workspace_floats(B, C, HW)
    return torch.empty(n, dtype=torch.float32, device=x.device).data_ptr() if n else None


def bn_act_fwd(x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, relu):
    """y = [relu]( bn(x) [+ residual] ), one launch; returns (y, save_mean, save_invstd) (the saves are None in eval mode).
    Updates running_mean / running_var in place in training mode, as nn.BatchNorm2d does (resnet.py:44-59)."""
    B, C = x.shape[0], x.shape[1]
    HW = x[0, 0].numel() if B else 1
    px = _chk(x, torch.float32, "x")
    pr = None if residual is None else _chk(residual, torch.float32, "residual", x.shape)
    pg = None if gamma is None else _chk(gamma, torch.float32, "gamma", (C,))
    pb = None if beta is None else _chk(beta, torch.float32, "beta", (C,))
    prm = None if running_mean is None else _chk(running_mean, torch.float32, "running_mean", (C,))
    prv = None if running_var is None else _chk(running_var, torch.float32, "running_var", (C,))
    y = torch.empty_like(x)
    sm = si = psm = psi = None
    if training:
        sm = torch.empty(C, dtype=torch.float32, device=x.device)
        si = torch.empty(C, dtype=torch.float32, device=x.device)
        psm, psi = sm.data_ptr(), si.data_ptr()
    N.check(N.lib.ee_bn_act_fwd_f32(px, pr, pg, pb, prm, prv, float(momentum), float(eps), 1 if training else 0, 1 if relu else 0,
                                    _chk(y, torch.float32, "y"), psm, psi, _bn_workspace(x, B, C, HW), B, C, HW, _stream()),
            "ee_bn_act_fwd_f32")
    return y, sm, si


def bn_act_bwd(dy, y, x, gamma, save_mean, save_invstd, running_mean, running_var, eps, training, relu, want_dx=True,
               want_dres=False, want_dparams=True, dy2=None, beta=None):
    """Backward of bn_act_fwd: returns (dx, dresidual, dgamma, dbeta), None where not wanted.  `dy2`: a second piece of the incoming
    gradient (the output fed two consumers), added to dy on load.  y=None with relu (forward without residual only): the ReLU mask is
    recomputed from x, gamma and `beta`."""
    B, C = x.shape[0], x.shape[1]
    HW = x[0, 0].numel() if B else 1
    pdy = _chk(dy, torch.float32, "dy", x.shape)
    py = None if y is None else _chk(y, torch.float32, "y", x.shape)
    px = _chk(x, torch.float32, "x")
    pg = None if gamma is None else _chk(gamma, torch.float32, "gamma", (C,))
    dx = torch.empty_like(x) if want_dx else None
    dres = torch.empty_like(x) if want_dres else None
    dg = torch.empty(C, dtype=torch.float32, device=x.device) if want_dparams else None
    db = torch.empty(C, dtype=torch.float32, device=x.device) if want_dparams else None
    ptr = lambda t: None if t is None else t.data_ptr()
    pdy2 = None if dy2 is None else _chk(dy2, torch.float32, "dy2", x.shape)
    N.check(N.lib.ee_bn_act_bwd2_f32(pdy, pdy2, py, px, pg, ptr(beta), ptr(save_mean), ptr(save_invstd), ptr(running_mean), ptr(running_var), float(eps),
                                     1 if training else 0, 1 if relu else 0, ptr(dx), ptr(dres), ptr(dg), ptr(db),
                                     _bn_workspace(x, B, C, HW), B, C, HW, _stream()),
            "ee_bn_act_bwd2_f32")
    return dx, dres, dg, db


# ---- SyncBatchNorm: the local halves around the two exchanges (ee_bn.hip; the collectives live in eeadv/syncbn.py) --------------------
def syncbn_supported(x):
    return x.dim() >= 3 and x.shape[0] > 0 and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and N.lib.ee_syncbn_workspace_floats(
        x.shape[0], x.shape[1], x[0, 0].numel()) > 0


def _syncbn_workspace(x, B, C, HW):
    return torch.empty(N.lib.ee_syncbn_workspace_floats(B, C, HW), dtype=torch.float32, device=x.device)


def syncbn_stats(x):
    """this rank's (mean, M2, count) per channel: [C, 3]"""
    B, C = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    moments = torch.empty((C, 3), dtype=torch.float32, device=x.device)
    ws = _syncbn_workspace(x, B, C, HW)
    N.check(N.lib.ee_syncbn_stats_f32(_chk(x, torch.float32, "x"), ws.data_ptr(), moments.data_ptr(), B, C, HW, _stream()), "ee_syncbn_stats_f32")
    return moments


def syncbn_apply(x, residual, gamma, beta, all_moments, running_mean, running_var, momentum, eps, relu):
    """all_moments [W, C, 3] (rank order) -> (y, save_mean, save_invstd) of the global batch; running statistics updated in place"""
    B, C = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    W = all_moments.shape[0]
    pm = _chk(all_moments, torch.float32, "all_moments", (W, C, 3))
    ptr = lambda t: None if t is None else t.data_ptr()
    y = torch.empty_like(x)
    sm = torch.empty(C, dtype=torch.float32, device=x.device)
    si = torch.empty(C, dtype=torch.float32, device=x.device)
    pr = None if residual is None else _chk(residual, torch.float32, "residual", x.shape)
    N.check(N.lib.ee_syncbn_apply_f32(_chk(x, torch.float32, "x"), pr, ptr(gamma), ptr(beta), pm, W, ptr(running_mean), ptr(running_var), float(momentum),
                                      float(eps), 1 if relu else 0, y.data_ptr(), sm.data_ptr(), si.data_ptr(), B, C, HW, _stream()), "ee_syncbn_apply_f32")
    return y, sm, si


def syncbn_bwd_sums(dy, dy2, y, x, gamma, beta, save_mean, save_invstd, relu):
    """this rank's (sum dz, sum dz * xhat) per channel: [C, 2]"""
    B, C = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    ptr = lambda t: None if t is None else t.data_ptr()
    sums = torch.empty((C, 2), dtype=torch.float32, device=x.device)
    ws = _syncbn_workspace(x, B, C, HW)
    N.check(N.lib.ee_syncbn_bwd_sums_f32(_chk(dy, torch.float32, "dy", x.shape), None if dy2 is None else _chk(dy2, torch.float32, "dy2", x.shape),
                                         None if y is None else _chk(y, torch.float32, "y", x.shape), _chk(x, torch.float32, "x"), ptr(gamma), ptr(beta),
                                         save_mean.data_ptr(), save_invstd.data_ptr(), 1 if relu else 0, ws.data_ptr(), sums.data_ptr(), B, C, HW, _stream()),
            "ee_syncbn_bwd_sums_f32")
    return sums


def syncbn_bwd_apply(dy, dy2, y, x, gamma, beta, save_mean, save_invstd, global_sums, n_global, relu, want_dx=True, want_dres=False):
    B, C = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    ptr = lambda t: None if t is None else t.data_ptr()
    dx = torch.empty_like(x) if want_dx else None
    dres = torch.empty_like(x) if want_dres else None
    N.check(N.lib.ee_syncbn_bwd_apply_f32(_chk(dy, torch.float32, "dy", x.shape), None if dy2 is None else _chk(dy2, torch.float32, "dy2", x.shape),
                                          None if y is None else _chk(y, torch.float32, "y", x.shape), _chk(x, torch.float32, "x"), ptr(gamma), ptr(beta),
                                          save_mean.data_ptr(), save_invstd.data_ptr(), _chk(global_sums, torch.float32, "global_sums", (C, 2)),
                                          float(n_global), 1 if relu else 0, ptr(dx), ptr(dres), B, C, HW, _stream()), "ee_syncbn_bwd_apply_f32")
    return dx, dres


def bn_dual_supported(x):
    return x.dim() == 4 and x.shape[0] > 0 and N.lib.ee_bn_dual_supported(x.shape[0], x.shape[1], x[0, 0].numel()) == 1


def bn_dual_fwd(xa, xb, bna, bnb, training):
    """y = relu(bn_a(xa) + bn_b(xb)) in one launch (the end of a residual block with a down-sampling shortcut, resnet.py:54-59).
    bna / bnb = (gamma, beta, running_mean, running_var, momentum, eps).  Returns (y, (sm_a, si_a, sm_b, si_b)) - Nones in eval mode."""
    B, C = xa.shape[0], xa.shape[1]
    HW = xa[0, 0].numel()
    y = torch.empty_like(xa)
    saves = [None] * 4
    if training:
        saves = [torch.empty(C, dtype=torch.float32, device=xa.device) for _ in range(4)]
    ptr = lambda t: None if t is None else t.data_ptr()
    ga, ba, rma, rva, ma, ea = bna
    gb, bb, rmb, rvb, mb, eb = bnb
    N.check(N.lib.ee_bn_dual_fwd_f32(_chk(xa, torch.float32, "xa"), _chk(xb, torch.float32, "xb", xa.shape), ptr(ga), ptr(ba), ptr(rma), ptr(rva),
                                     float(ma), float(ea), ptr(saves[0]), ptr(saves[1]), ptr(gb), ptr(bb), ptr(rmb), ptr(rvb), float(mb), float(eb),
                                     ptr(saves[2]), ptr(saves[3]), 1 if training else 0, y.data_ptr(), B, C, HW, _stream()), "ee_bn_dual_fwd_f32")
    return y, tuple(saves)


def bn_dual_bwd(dy, dy2, y, xa, xb, ga, gb, saves, rma, rva, rmb, rvb, ea, eb, training, want_dxa=True, want_dxb=True, want_dparams=True):
    """Backward of bn_dual_fwd: (dxa, dxb, dgamma_a, dbeta_a, dgamma_b, dbeta_b), None where not wanted."""
    B, C = xa.shape[0], xa.shape[1]
    HW = xa[0, 0].numel()
    dxa = torch.empty_like(xa) if want_dxa else None
    dxb = torch.empty_like(xb) if want_dxb else None
    dp = [torch.empty(C, dtype=torch.float32, device=xa.device) if want_dparams else None for _ in range(4)]
    ptr = lambda t: None if t is None else t.data_ptr()
    N.check(N.lib.ee_bn_dual_bwd_f32(_chk(dy, torch.float32, "dy", xa.shape), None if dy2 is None else _chk(dy2, torch.float32, "dy2", xa.shape),
                                     _chk(y, torch.float32, "y", xa.shape), _chk(xa, torch.float32, "xa"), _chk(xb, torch.float32, "xb"), ptr(ga),
                                     ptr(saves[0]), ptr(saves[1]), ptr(rma), ptr(rva), float(ea), ptr(gb), ptr(saves[2]), ptr(saves[3]), ptr(rmb),
                                     ptr(rvb), float(eb), 1 if training else 0, ptr(dxa), ptr(dxb), ptr(dp[0]), ptr(dp[1]), ptr(dp[2]), ptr(dp[3]),
                                     B, C, HW, _stream()), "ee_bn_dual_bwd_f32")
    return (dxa, dxb) + tuple(dp)


def bn_relu_pool_supported(x):
    """relu(bn(x)) -> MaxPool2d(3, 2, 1) as one pass each way (the ResNet stem): shapes ee_bn_relu_pool_*_f32 take."""
    return x.dim() == 4 and x.shape[0] > 0 and N.lib.ee_bn_relu_pool_workspace_floats(x.shape[0], x.shape[1], x.shape[2], x.shape[3]) > 0


def _pool_workspace(x):
    n = N.lib.ee_bn_relu_pool_workspace_floats(x.shape[0], x.shape[1], x.shape[2], x.shape[3])
    return torch.empty(n, dtype=torch.float32, device=x.device)


def bn_relu_pool_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, training, conv_stats=None, want_x_argmax=False):
    """maxpool3s2(relu(bn(x))) -> (y_pool, code uint8, save_mean, save_invstd) (resnet.py:113-117); the saves are None in eval mode.
    conv_stats [C,S,3]: the producing convolution's per-workgroup moments of x (stem7x7s2_fwd(want_stats=True)): no statistics pass over x.
    want_x_argmax: a fifth result, x at every window's argmax [B,C,OH,OW] - bn_relu_pool_bwd(x_argmax=...) then needs one pass over x less."""
    B, C, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
    code = torch.empty((B, C, OH, OW), dtype=torch.uint8, device=x.device)
    sm = si = None
    if training:
        sm = torch.empty(C, dtype=torch.float32, device=x.device)
        si = torch.empty(C, dtype=torch.float32, device=x.device)
    ptr = lambda t: None if t is None else t.data_ptr()
    ws = _pool_workspace(x)
    cs = None if conv_stats is None else _chk(conv_stats, torch.float32, "conv_stats")
    ncs = 0 if conv_stats is None else conv_stats.shape[1]
    if want_x_argmax:
        xa = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
        N.check(N.lib.ee_bn_relu_pool_fwd_xa_f32(_chk(x, torch.float32, "x"), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(momentum),
                                                 float(eps), 1 if training else 0, y.data_ptr(), code.data_ptr(), xa.data_ptr(), ptr(sm), ptr(si),
                                                 ws.data_ptr(), cs, ncs, B, C, H, W, _stream()), "ee_bn_relu_pool_fwd_xa_f32")
        return y, code, sm, si, xa
    N.check(N.lib.ee_bn_relu_pool_fwd_f32(_chk(x, torch.float32, "x"), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(momentum),
                                          float(eps), 1 if training else 0, y.data_ptr(), code.data_ptr(), ptr(sm), ptr(si), ws.data_ptr(),
                                          cs, ncs, B, C, H, W, _stream()), "ee_bn_relu_pool_fwd_f32")
    return y, code, sm, si


def bn_relu_pool_bwd(dy_pool, code, x, gamma, beta, save_mean, save_invstd, running_mean, running_var, eps, training, want_dx=True,
                     want_dparams=True, dy_pool2=None, x_argmax=None):
    """Backward of bn_relu_pool_fwd: (dx, dgamma, dbeta), None where not wanted; `dy_pool2` = a second piece of the gradient;
    `x_argmax` (bn_relu_pool_fwd(want_x_argmax=True)): training mode then takes its batch sums from the pooled tensors."""
    B, C, H, W = x.shape
    dx = torch.empty_like(x) if want_dx else None
    dg = torch.empty(C, dtype=torch.float32, device=x.device) if want_dparams else None
    db = torch.empty(C, dtype=torch.float32, device=x.device) if want_dparams else None
    ptr = lambda t: None if t is None else t.data_ptr()
    ws = _pool_workspace(x)
    p2 = None if dy_pool2 is None else _chk(dy_pool2, torch.float32, "dy_pool2", code.shape)
    if x_argmax is not None:
        N.check(N.lib.ee_bn_relu_pool_bwd_xa_f32(_chk(dy_pool, torch.float32, "dy_pool", code.shape), p2, _chk(code, torch.uint8, "code"),
                                                 _chk(x, torch.float32, "x"), _chk(x_argmax, torch.float32, "x_argmax", code.shape), ptr(gamma), ptr(beta),
                                                 ptr(save_mean), ptr(save_invstd), ptr(running_mean), ptr(running_var), float(eps), 1 if training else 0,
                                                 ptr(dx), ptr(dg), ptr(db), ws.data_ptr(), B, C, H, W, _stream()), "ee_bn_relu_pool_bwd_xa_f32")
        return dx, dg, db
    N.check(N.lib.ee_bn_relu_pool_bwd_f32(_chk(dy_pool, torch.float32, "dy_pool", code.shape), p2, _chk(code, torch.uint8, "code"),
                                          _chk(x, torch.float32, "x"), ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_invstd), ptr(running_mean),
                                          ptr(running_var), float(eps), 1 if training else 0, ptr(dx), ptr(dg), ptr(db), ws.data_ptr(),
                                          B, C, H, W, _stream()), "ee_bn_relu_pool_bwd_f32")
    return dx, dg, db


# ---- stem max-pool and classifier head -----------------------------------------------------------------------------------
def maxpool3s2_fwd(x):
    """MaxPool2d(3, 2, 1) of x [B,C,H,W] -> (y, code uint8) (resnet.py:117)."""
    B, C, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
    code = torch.empty((B, C, OH, OW), dtype=torch.uint8, device=x.device)
    N.check(N.lib.ee_maxpool3s2_fwd_f32(_chk(x, torch.float32, "x"), _chk(y, torch.float32, "y"), _chk(code, torch.uint8, "code"), B * C, H, W,
                                        _stream()), "ee_maxpool3s2_fwd_f32")
    return y, code


def maxpool3s2_bwd(dy, code, H, W):
    B, C, OH, OW = dy.shape
    if (OH, OW) != ((H - 1) // 2 + 1, (W - 1) // 2 + 1):
        raise ValueError("maxpool3s2_bwd: dy %s does not belong to a %dx%d input" % (tuple(dy.shape), H, W))
    dx = torch.empty((B, C, H, W), dtype=torch.float32, device=dy.device)
    N.check(N.lib.ee_maxpool3s2_bwd_f32(_chk(dy, torch.float32, "dy"), _chk(code, torch.uint8, "code", dy.shape), _chk(dx, torch.float32, "dx"),
                                        B * C, H, W, _stream()), "ee_maxpool3s2_bwd_f32")
    return dx


def conv1x1s2_fwd(x, weight):
    """Conv2d(Cin, Cout, 1, stride=2, bias=False) (resnet.py:137-142): x [B,Cin,H,W], weight [Cout,Cin,1,1]."""
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    y = torch.empty((B, Cout, H // 2, W // 2), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_conv1x1s2_fwd_f32(_chk(x, torch.float32, "x"), _chk(weight, torch.float32, "weight", (Cout, Cin, 1, 1)), y.data_ptr(), B,
                                       Cin, Cout, H, W, _stream()), "ee_conv1x1s2_fwd_f32")
    return y


def conv1x1s2_bwd(dy, weight, H, W):
    B, Cout = dy.shape[0], dy.shape[1]
    Cin = weight.shape[1]
    dx = torch.empty((B, Cin, H, W), dtype=torch.float32, device=dy.device)
    N.check(N.lib.ee_conv1x1s2_bwd_f32(_chk(dy, torch.float32, "dy", (B, Cout, H // 2, W // 2)),
                                       _chk(weight, torch.float32, "weight", (Cout, Cin, 1, 1)), dx.data_ptr(), B, Cin, Cout, H, W, _stream()),
            "ee_conv1x1s2_bwd_f32")
    return dx


def wino3x3_supported(x, cin, cout):
    return x.dim() == 4 and x.shape[2] == x.shape[3] and x.shape[2] in (4, 8, 16) and cin % 32 == 0 and cout % 32 == 0


def wino3x3(x, u):
    """3x3 / stride 1 / padding 1 convolution of 4x4, 8x8 or 16x16 maps, filters in the Winograd domain: x [B,KC,H,H], u [16,KC,RC] -> [B,RC,H,H]."""
    B, KC, H = x.shape[0], x.shape[1], x.shape[2]
    RC = u.shape[2]
    y = torch.empty((B, RC, H, H), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_wino3x3_f32(_chk(x, torch.float32, "x", (B, KC, H, H)), _chk(u, torch.float32, "u", (16, KC, RC)), y.data_ptr(), B, KC, RC, H,
                                 _stream()), "ee_wino3x3_f32")
    return y


def _optf(t, name, shape=None):
    return None if t is None else _chk(t, torch.float32, name, shape)


def wino3x3_bn_eval_fwd(x, u, bn, res, relu):
    """[relu]( bn(conv3x3(x)) [+ res] ) with bn in eval mode (running statistics) in ONE launch: bn = (mean, var, gamma, beta, eps)"""
    B, KC, H = x.shape[0], x.shape[1], x.shape[2]
    RC = u.shape[2]
    mean, var, gamma, beta, eps = bn
    y = torch.empty((B, RC, H, H), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_wino3x3_bn_eval_fwd_f32(_chk(x, torch.float32, "x", (B, KC, H, H)), _chk(u, torch.float32, "u", (16, KC, RC)),
                                             _chk(mean, torch.float32, "running_mean", (RC,)), _chk(var, torch.float32, "running_var", (RC,)),
                                             _optf(gamma, "gamma", (RC,)), _optf(beta, "beta", (RC,)), float(eps), _optf(res, "res", (B, RC, H, H)),
                                             1 if relu else 0, y.data_ptr(), B, KC, RC, H, _stream()), "ee_wino3x3_bn_eval_fwd_f32")
    return y


def wino3x3_bn_eval_bwd(dy, dy2, y, u_b, bn, want_dres, dx_add=None):
    """dz = (y > 0) * (dy [+ dy2]); dx = conv3x3^T(gamma / sqrt(var + eps) * dz) [+ dx_add] in ONE launch -> (dx, dz or None).
    bn = (var, gamma, eps) of the BatchNorm behind the convolution; u_b [16, Cout, Cin]."""
    B, Cout, H = dy.shape[0], dy.shape[1], dy.shape[2]
    Cin = u_b.shape[2]
    var, gamma, eps = bn
    dx = torch.empty((B, Cin, H, H), dtype=torch.float32, device=dy.device)
    dres = torch.empty_like(dy) if want_dres else None
    N.check(N.lib.ee_wino3x3_bn_eval_bwd_f32(_chk(dy, torch.float32, "dy", (B, Cout, H, H)), _optf(dy2, "dy2", (B, Cout, H, H)),
                                             _chk(y, torch.float32, "y", (B, Cout, H, H)), _chk(u_b, torch.float32, "u_b", (16, Cout, Cin)),
                                             _chk(var, torch.float32, "running_var", (Cout,)), _optf(gamma, "gamma", (Cout,)), float(eps),
                                             None if dres is None else dres.data_ptr(), _optf(dx_add, "dx_add", (B, Cin, H, H)), dx.data_ptr(),
                                             B, Cin, Cout, H, _stream()), "ee_wino3x3_bn_eval_bwd_f32")
    return dx, dres


def conv3x3s2_pair_bn_eval_fwd(x, w10, cout, bn3, bn1):
    """(relu(bn3(conv3x3s2(x))), bn1(conv1x1s2(x))) with both BatchNorms in eval mode, ONE launch: bn = (mean, var, gamma, beta, eps)"""
    B, Cin, H = x.shape[0], x.shape[1], x.shape[2]
    y3 = torch.empty((B, cout, H // 2, H // 2), dtype=torch.float32, device=x.device)
    y1 = torch.empty_like(y3)
    N.check(N.lib.ee_conv3x3s2_pair_bn_eval_fwd_f32(
        _chk(x, torch.float32, "x", (B, Cin, H, H)), _chk(w10, torch.float32, "w10", (cout // 32, Cin // 16, 10, 4, 2, 16, 4)),
        _chk(bn3[0], torch.float32, "mean3", (cout,)), _chk(bn3[1], torch.float32, "var3", (cout,)), _optf(bn3[2], "gamma3", (cout,)), _optf(bn3[3], "beta3", (cout,)), float(bn3[4]),
        _chk(bn1[0], torch.float32, "mean1", (cout,)), _chk(bn1[1], torch.float32, "var1", (cout,)), _optf(bn1[2], "gamma1", (cout,)), _optf(bn1[3], "beta1", (cout,)), float(bn1[4]),
        y3.data_ptr(), y1.data_ptr(), B, Cin, cout, H, _stream()), "ee_conv3x3s2_pair_bn_eval_fwd_f32")
    return y3, y1


def conv3x3s2_pair_bn_eval_bwd(dy3, y3, dy1, w10, cin, bn3, bn1):
    """dx = conv3x3s2^T(gamma3 / sqrt(var3 + eps3) * (y3 > 0) * dy3) + conv1x1s2^T(gamma1 / sqrt(var1 + eps1) * dy1); bn = (var, gamma, eps)"""
    B, Cout, OH = dy3.shape[0], dy3.shape[1], dy3.shape[2]
    dx = torch.empty((B, cin, 2 * OH, 2 * OH), dtype=torch.float32, device=dy3.device)
    N.check(N.lib.ee_conv3x3s2_pair_bn_eval_bwd_f32(
        _chk(dy3, torch.float32, "dy3", (B, Cout, OH, OH)), _chk(y3, torch.float32, "y3", (B, Cout, OH, OH)), _chk(dy1, torch.float32, "dy1", (B, Cout, OH, OH)),
        _chk(w10, torch.float32, "w10", (cin // 32, Cout // 16, 10, 4, 2, 16, 4)), _chk(bn3[0], torch.float32, "var3", (Cout,)), _optf(bn3[1], "gamma3", (Cout,)),
        float(bn3[2]), _chk(bn1[0], torch.float32, "var1", (Cout,)), _optf(bn1[1], "gamma1", (Cout,)), float(bn1[2]), dx.data_ptr(), B, cin, Cout, 2 * OH,
        _stream()), "ee_conv3x3s2_pair_bn_eval_bwd_f32")
    return dx


def wino3x3_stats(x, u):
    """conv3x3(x) and, per (result channel, image), the plane's (mean, M2): (y, stats [RC, B, 2]) - the producer half of a train-mode
    BatchNorm exchanged across the kernel boundary (ee_wino3x3_stats_f32)"""
    B, KC, H = x.shape[0], x.shape[1], x.shape[2]
    RC = u.shape[2]
    y = torch.empty((B, RC, H, H), dtype=torch.float32, device=x.device)
    stats = torch.empty((RC, B, 2), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_wino3x3_stats_f32(_chk(x, torch.float32, "x", (B, KC, H, H)), _chk(u, torch.float32, "u", (16, KC, RC)), y.data_ptr(), stats.data_ptr(),
                                       B, KC, RC, H, _stream()), "ee_wino3x3_stats_f32")
    return y, stats


def wino3x3_bn_train_pre(x, stats, cnt, gamma, beta, eps, momentum, running_mean, running_var, u):
    """conv3x3(relu(batch_norm(x))) in TRAIN mode, the batch statistics merged from `stats` [KC, S, 2] (S partials of `cnt` values each):
    (y, save_mean, save_invstd); running_mean / running_var (or None) move by `momentum` (ee_wino3x3_bn_train_pre_f32)"""
    B, KC, H = x.shape[0], x.shape[1], x.shape[2]
    RC = u.shape[2]
    S = stats.shape[1]
    y = torch.empty((B, RC, H, H), dtype=torch.float32, device=x.device)
    sm = torch.empty(KC, dtype=torch.float32, device=x.device)
    si = torch.empty(KC, dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_wino3x3_bn_train_pre_f32(_chk(x, torch.float32, "x", (B, KC, H, H)), _chk(stats, torch.float32, "stats", (KC, S, 2)), S, int(cnt),
                                              _optf(gamma, "gamma", (KC,)), _optf(beta, "beta", (KC,)), float(eps), float(momentum),
                                              _optf(running_mean, "running_mean", (KC,)), _optf(running_var, "running_var", (KC,)), sm.data_ptr(), si.data_ptr(),
                                              _chk(u, torch.float32, "u", (16, KC, RC)), y.data_ptr(), B, KC, RC, H, _stream()), "ee_wino3x3_bn_train_pre_f32")
    return y, sm, si


def wino3x3_bwd_sums(dc, u_b, x, save_mean, save_invstd, gamma, beta):
    """dy = conv3x3^T(dc) (the backward-data of the layer behind a train-mode BatchNorm) plus, per (channel, image), (sum dz, sum dz * xhat) with
    dz = (bn(x) > 0) * dy: (dy, sums [Cin, B, 2]) - the producer half of the BatchNorm backward across the kernel boundary (16x16 maps)"""
    B, Cout, H = dc.shape[0], dc.shape[1], dc.shape[2]
    Cin = u_b.shape[2]
    dy = torch.empty((B, Cin, H, H), dtype=torch.float32, device=dc.device)
    sums = torch.empty((Cin, B, 2), dtype=torch.float32, device=dc.device)
    N.check(N.lib.ee_wino3x3_bwd_sums_f32(_chk(dc, torch.float32, "dc", (B, Cout, H, H)), _chk(u_b, torch.float32, "u_b", (16, Cout, Cin)),
                                          _chk(x, torch.float32, "x", (B, Cin, H, H)), _chk(save_mean, torch.float32, "save_mean", (Cin,)),
                                          _chk(save_invstd, torch.float32, "save_invstd", (Cin,)), _optf(gamma, "gamma", (Cin,)), _optf(beta, "beta", (Cin,)),
                                          dy.data_ptr(), sums.data_ptr(), B, Cin, Cout, H, _stream()), "ee_wino3x3_bwd_sums_f32")
    return dy, sums


def wino3x3_bn_train_bwd_pre(dy, x, sums, cnt, save_mean, save_invstd, gamma, beta, u_b):
    """dx = conv3x3^T(batch_norm_relu_backward(dy; x)) in TRAIN mode, the batch means of dz and dz * xhat merged from `sums` [Cout, S, 2]"""
    B, Cout, H = dy.shape[0], dy.shape[1], dy.shape[2]
    Cin = u_b.shape[2]
    dx = torch.empty((B, Cin, H, H), dtype=torch.float32, device=dy.device)
    N.check(N.lib.ee_wino3x3_bn_train_bwd_pre_f32(_chk(dy, torch.float32, "dy", (B, Cout, H, H)), _chk(x, torch.float32, "x", (B, Cout, H, H)),
                                                  _chk(sums, torch.float32, "sums", (Cout, sums.shape[1], 2)), sums.shape[1], int(cnt),
                                                  _chk(save_mean, torch.float32, "save_mean", (Cout,)), _chk(save_invstd, torch.float32, "save_invstd", (Cout,)),
                                                  _optf(gamma, "gamma", (Cout,)), _optf(beta, "beta", (Cout,)), _chk(u_b, torch.float32, "u_b", (16, Cout, Cin)),
                                                  dx.data_ptr(), B, Cin, Cout, H, _stream()), "ee_wino3x3_bn_train_bwd_pre_f32")
    return dx


def conv3x3s2_pair_stats_fwd(x, w10, cout):
    """conv3x3s2_pair_fwd plus the statistics of y3 for the train-mode BatchNorm behind it: (y3, y1, stats [Cout, S, 2], cnt)"""
    B, Cin, H = x.shape[0], x.shape[1], x.shape[2]
    y3 = torch.empty((B, cout, H // 2, H // 2), dtype=torch.float32, device=x.device)
    y1 = torch.empty_like(y3)
    S, cnt = (2 * B, 32) if H == 16 else (B, 16)
    stats = torch.empty((cout, S, 2), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_conv3x3s2_pair_stats_fwd_f32(_chk(x, torch.float32, "x", (B, Cin, H, H)), _chk(w10, torch.float32, "w10", (cout // 32, Cin // 16, 10, 4, 2, 16, 4)),
                                                  y3.data_ptr(), y1.data_ptr(), stats.data_ptr(), B, Cin, cout, H, _stream()), "ee_conv3x3s2_pair_stats_fwd_f32")
    return y3, y1, stats, cnt


def dense2x2_supported(cin, cout):
    return cin % 128 == 0 and cout % 8 == 0


def dense2x2(x, w2):
    """conv3x3 on a 2x2 map as one dense product (ee_dense.hip): x [B, Cin, 2, 2], w2 [4 Cin, 4 Cout] -> [B, Cout, 2, 2]"""
    B, Cin = x.shape[0], x.shape[1]
    Cout = w2.shape[1] // 4
    y = torch.empty((B, Cout, 2, 2), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_dense2x2_f32(_chk(x, torch.float32, "x", (B, Cin, 2, 2)), _chk(w2, torch.float32, "w2", (4 * Cin, 4 * Cout)), y.data_ptr(), B, Cin, Cout,
                                  _stream()), "ee_dense2x2_f32")
    return y


def dense2x2_bn_eval_fwd(x, w2, bn, res, relu):
    """[relu]( bn(conv3x3(x)) [+ res] ) on a 2x2 map with bn in eval mode, ONE launch: bn = (mean, var, gamma, beta, eps)"""
    B, Cin = x.shape[0], x.shape[1]
    Cout = w2.shape[1] // 4
    mean, var, gamma, beta, eps = bn
    y = torch.empty((B, Cout, 2, 2), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_dense2x2_bn_eval_fwd_f32(_chk(x, torch.float32, "x", (B, Cin, 2, 2)), _chk(w2, torch.float32, "w2", (4 * Cin, 4 * Cout)),
                                              _chk(mean, torch.float32, "running_mean", (Cout,)), _chk(var, torch.float32, "running_var", (Cout,)),
                                              _optf(gamma, "gamma", (Cout,)), _optf(beta, "beta", (Cout,)), float(eps), _optf(res, "res", (B, Cout, 2, 2)),
                                              1 if relu else 0, y.data_ptr(), B, Cin, Cout, _stream()), "ee_dense2x2_bn_eval_fwd_f32")
    return y


def dense2x2_bn_eval_bwd(dy, dy2, y, w2t, bn, want_dres, dx_add=None):
    """dz = (y > 0) * (dy [+ dy2]); dx = (gamma / sqrt(var + eps) * dz) . w2t [+ dx_add] -> (dx, dz or None); w2t [4 Cout, 4 Cin]; bn = (var, gamma, eps)"""
    B, Cout = dy.shape[0], dy.shape[1]
    Cin = w2t.shape[1] // 4
    var, gamma, eps = bn
    dx = torch.empty((B, Cin, 2, 2), dtype=torch.float32, device=dy.device)
    dres = torch.empty_like(dy) if want_dres else None
    N.check(N.lib.ee_dense2x2_bn_eval_bwd_f32(_chk(dy, torch.float32, "dy", (B, Cout, 2, 2)), _optf(dy2, "dy2", (B, Cout, 2, 2)),
                                              _chk(y, torch.float32, "y", (B, Cout, 2, 2)), _chk(w2t, torch.float32, "w2t", (4 * Cout, 4 * Cin)),
                                              _chk(var, torch.float32, "running_var", (Cout,)), _optf(gamma, "gamma", (Cout,)), float(eps),
                                              None if dres is None else dres.data_ptr(), _optf(dx_add, "dx_add", (B, Cin, 2, 2)), dx.data_ptr(),
                                              B, Cin, Cout, _stream()), "ee_dense2x2_bn_eval_bwd_f32")
    return dx, dres


def wrw3x3_supported(x, dy):
    """ee_wrw.hip: weight gradient of a 3x3 / stride 1 / padding 1 convolution on 2x2, 4x4, 8x8 or 16x16 maps"""
    return (x.dim() == 4 and dy.dim() == 4 and x.shape[2] == x.shape[3] and x.shape[2] in (2, 4, 8, 16) and x.shape[2:] == dy.shape[2:]
            and x.shape[0] == dy.shape[0] and x.shape[1] % 32 == 0 and dy.shape[1] % 32 == 0 and x.is_cuda and x.dtype == torch.float32
            and dy.dtype == torch.float32 and x.is_contiguous() and dy.is_contiguous())


def wrw3x3(x, dy):
    """d loss / d weight [Cout,Cin,3,3] of conv3x3(x, weight) (stride 1, padding 1) from its input x [B,Cin,H,H] and output gradient dy [B,Cout,H,H]:
    Winograd F(3x3, 2x2) on the matrix cores, partial sums added in a fixed order (bit-reproducible)."""
    B, Cin, H = x.shape[0], x.shape[1], x.shape[2]
    Cout = dy.shape[1]
    dw = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=x.device)
    n = int(N.lib.ee_wrw3x3_workspace_floats(B, Cin, Cout, H))
    ws = torch.empty(max(n, 4), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_wrw3x3_f32(_chk(x, torch.float32, "x", (B, Cin, H, H)), _chk(dy, torch.float32, "dy", (B, Cout, H, H)), dw.data_ptr(), ws.data_ptr(),
                                B, Cin, Cout, H, _stream()), "ee_wrw3x3_f32")
    return dw


def wrw3x3s2_supported(x, dy3, dy1=None):
    """ee_wrw.hip: weight gradient of a 3x3 / stride 2 / padding 1 convolution from a 16x16, 8x8 or 4x4 map (and of the 1x1 / stride 2 shortcut)"""
    ok = (x.dim() == 4 and dy3.dim() == 4 and x.shape[2] == x.shape[3] and x.shape[2] in (4, 8, 16) and dy3.shape[2] == dy3.shape[3] == x.shape[2] // 2
          and x.shape[0] == dy3.shape[0] and x.shape[1] % 32 == 0 and dy3.shape[1] % 32 == 0 and x.is_cuda and x.dtype == torch.float32
          and dy3.dtype == torch.float32 and x.is_contiguous() and dy3.is_contiguous())
    return ok and (dy1 is None or (dy1.shape == dy3.shape and dy1.dtype == torch.float32 and dy1.is_contiguous()))


def wrw3x3s2(x, dy3, dy1=None):
    """(dw3 [Cout,Cin,3,3], dw1 [Cout,Cin,1,1] or None): weight gradients of conv3x3(x) (stride 2, padding 1) and, with dy1, of the 1x1 / stride 2
    shortcut of the same x, in one launch + the fixed-order sum (two views of one buffer)"""
    B, Cin, H = x.shape[0], x.shape[1], x.shape[2]
    Cout = dy3.shape[1]
    taps = 9 if dy1 is None else 10
    dw = torch.empty(taps * Cout * Cin, dtype=torch.float32, device=x.device)
    n = int(N.lib.ee_wrw3x3s2_workspace_floats(B, Cin, Cout, H, 0 if dy1 is None else 1))
    ws = torch.empty(max(n, 4), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_wrw3x3s2_f32(_chk(x, torch.float32, "x", (B, Cin, H, H)), _chk(dy3, torch.float32, "dy3", (B, Cout, H // 2, H // 2)),
                                  None if dy1 is None else _chk(dy1, torch.float32, "dy1", (B, Cout, H // 2, H // 2)), dw.data_ptr(), ws.data_ptr(),
                                  B, Cin, Cout, H, _stream()), "ee_wrw3x3s2_f32")
    dw3 = dw[:9 * Cout * Cin].view(Cout, Cin, 3, 3)
    return dw3, (None if dy1 is None else dw[9 * Cout * Cin:].view(Cout, Cin, 1, 1))


def wrw_stem7x7s2_supported(x, dy):
    """ee_wrw.hip: weight gradient of the stem Conv2d(3, 64, 7, stride 2, padding 3)"""
    return (x.dim() == 4 and dy.dim() == 4 and x.shape[1] == 3 and dy.shape[1] == 64 and x.shape[2] % 2 == 0 and x.shape[3] % 32 == 0
            and tuple(dy.shape[2:]) == (x.shape[2] // 2, x.shape[3] // 2) and x.shape[0] == dy.shape[0] and x.is_cuda and x.dtype == torch.float32
            and dy.dtype == torch.float32 and x.is_contiguous() and dy.is_contiguous())


def wrw_stem7x7s2(x, dy):
    """d loss / d weight [64,3,7,7] of the stem convolution from its input x [B,3,H,W] and output gradient dy [B,64,H/2,W/2] (bit-reproducible)"""
    B, H, W = x.shape[0], x.shape[2], x.shape[3]
    dw = torch.empty((64, 3, 7, 7), dtype=torch.float32, device=x.device)
    n = int(N.lib.ee_wrw_stem7x7s2_workspace_floats(B, H, W))
    ws = torch.empty(max(n, 4), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_wrw_stem7x7s2_f32(_chk(x, torch.float32, "x", (B, 3, H, W)), _chk(dy, torch.float32, "dy", (B, 64, H // 2, W // 2)), dw.data_ptr(),
                                       ws.data_ptr(), B, H, W, _stream()), "ee_wrw_stem7x7s2_f32")
    return dw


def wrw1x1_supported(x, dy):
    """ee_wrw.hip: weight gradient of a 1x1 / stride 1 convolution, NCHW operands"""
    return (x.dim() == 4 and dy.dim() == 4 and x.shape[0] == dy.shape[0] and x.shape[2:] == dy.shape[2:] and x.shape[1] % 64 == 0 and dy.shape[1] % 64 == 0
            and (x.shape[2] * x.shape[3]) % 4 == 0 and x.is_cuda and x.dtype == torch.float32 and dy.dtype == torch.float32 and x.is_contiguous()
            and dy.is_contiguous())


def wrw1x1(x, dy):
    """d loss / d weight [Cout,Cin,1,1] of conv1x1(x, weight) (stride 1) from its input x [B,Cin,H,W] and output gradient dy [B,Cout,H,W] (bit-reproducible)"""
    B, Cin, HW = x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
    Cout = dy.shape[1]
    dw = torch.empty((Cout, Cin, 1, 1), dtype=torch.float32, device=x.device)
    n = int(N.lib.ee_wrw1x1_workspace_floats(B, Cin, Cout, HW))
    ws = torch.empty(max(n, 4), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_wrw1x1_f32(_chk(x, torch.float32, "x"), _chk(dy, torch.float32, "dy"), dw.data_ptr(), ws.data_ptr(), B, Cin, Cout, HW, _stream()),
            "ee_wrw1x1_f32")
    return dw


def conv3x3s2_small_supported(x, cin, cout):
    """ee_s2.hip: 3x3 / stride 2 / padding 1 from a 16x16, 8x8 or 4x4 map"""
    return x.dim() == 4 and x.shape[2] == x.shape[3] and x.shape[2] in (4, 8, 16) and cin % 32 == 0 and cout % 32 == 0


def conv3x3s2_small_fwd(x, w9, cout):
    """x [B,Cin,H,H] (H = 16, 8 or 4), w9 = the filters rearranged (functional._rearranged kind "s2m_f") -> [B,Cout,H/2,H/2]"""
    B, Cin, H = x.shape[0], x.shape[1], x.shape[2]
    y = torch.empty((B, cout, H // 2, H // 2), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_conv3x3s2_small_fwd_f32(_chk(x, torch.float32, "x", (B, Cin, H, H)), _chk(w9, torch.float32, "w9", (cout // 32, Cin // 16, 9, 4, 2, 16, 4)),
                                             y.data_ptr(), B, Cin, cout, H, _stream()), "ee_conv3x3s2_small_fwd_f32")
    return y


def conv3x3s2_small_bwd_data(dy, w9, cin):
    """dy [B,Cout,H/2,H/2], w9 = the filters rearranged (kind "s2m_b") -> dx [B,Cin,H,H] (H = 16, 8 or 4)"""
    B, Cout, OH = dy.shape[0], dy.shape[1], dy.shape[2]
    dx = torch.empty((B, cin, 2 * OH, 2 * OH), dtype=torch.float32, device=dy.device)
    N.check(N.lib.ee_conv3x3s2_small_bwd_data_f32(_chk(dy, torch.float32, "dy", (B, Cout, OH, OH)),
                                                  _chk(w9, torch.float32, "w9", (cin // 32, Cout // 16, 9, 4, 2, 16, 4)), dx.data_ptr(), B, cin, Cout, 2 * OH,
                                                  _stream()), "ee_conv3x3s2_small_bwd_data_f32")
    return dx


def conv_weight_prep(kind, weight, extra, out):
    """ee_wprep.hip: the filters of one convolution in the order its kernel reads them (kind = EE_WPREP_* of eeadv.h), written into `out`
    in one launch; `extra` = the shortcut's 1x1 weight for the paired stride-2 kinds."""
    co, ci = weight.shape[0], weight.shape[1]
    N.check(N.lib.ee_conv_weight_prep_f32(kind, _chk(weight, torch.float32, "weight", (co, ci, 3, 3)),
                                          None if extra is None else _chk(extra, torch.float32, "extra", (co, ci, 1, 1)),
                                          _chk(out, torch.float32, "out"), co, ci, _stream()), "ee_conv_weight_prep_f32")
    return out


def conv_weight_prep_batch(items):
    """ee_wprep.hip: `items` = [(kind, weight, extra or None, out)] - every rearranged copy in ONE launch per 64 items (ee_conv_weight_prep_batch_f32)"""
    n = len(items)
    if not n:
        return
    kinds = (ctypes.c_int * n)(*[int(k) for k, _, _, _ in items])
    cout = (ctypes.c_int * n)(*[w.shape[0] for _, w, _, _ in items])
    cin = (ctypes.c_int * n)(*[w.shape[1] for _, w, _, _ in items])
    pw = (ctypes.c_void_p * n)(*[_chk(w, torch.float32, "weight", (w.shape[0], w.shape[1], 3, 3)).value for _, w, _, _ in items])
    pe = (ctypes.c_void_p * n)(*[None if e is None else _chk(e, torch.float32, "extra", (w.shape[0], w.shape[1], 1, 1)).value for _, w, e, _ in items])
    po = (ctypes.c_void_p * n)(*[_chk(o, torch.float32, "out").value for _, _, _, o in items])
    N.check(N.lib.ee_conv_weight_prep_batch_f32(n, kinds, pw, pe, po, cout, cin, _stream()), "ee_conv_weight_prep_batch_f32")


def conv3x3s2_pair_fwd(x, w10, cout):
    """conv3x3 / stride 2 / padding 1 AND conv1x1 / stride 2 of the same x in one launch (a down-sampling BasicBlock's conv1 and shortcut):
    x [B,Cin,H,H], w10 = both filter sets rearranged (functional._rearranged kind "s2p_f") -> (y3, y1), each [B,Cout,H/2,H/2]"""
    B, Cin, H = x.shape[0], x.shape[1], x.shape[2]
    y3 = torch.empty((B, cout, H // 2, H // 2), dtype=torch.float32, device=x.device)
    y1 = torch.empty_like(y3)
    N.check(N.lib.ee_conv3x3s2_pair_fwd_f32(_chk(x, torch.float32, "x", (B, Cin, H, H)), _chk(w10, torch.float32, "w10", (cout // 32, Cin // 16, 10, 4, 2, 16, 4)),
                                            y3.data_ptr(), y1.data_ptr(), B, Cin, cout, H, _stream()), "ee_conv3x3s2_pair_fwd_f32")
    return y3, y1


def conv3x3s2_pair_bwd_data(dy3, dy1, w10, cin):
    """dx = conv3x3s2^T(dy3) + conv1x1s2^T(dy1): dy3, dy1 [B,Cout,H/2,H/2], w10 (kind "s2p_b") -> dx [B,Cin,H,H]"""
    B, Cout, OH = dy3.shape[0], dy3.shape[1], dy3.shape[2]
    dx = torch.empty((B, cin, 2 * OH, 2 * OH), dtype=torch.float32, device=dy3.device)
    N.check(N.lib.ee_conv3x3s2_pair_bwd_data_f32(_chk(dy3, torch.float32, "dy3", (B, Cout, OH, OH)), _chk(dy1, torch.float32, "dy1", (B, Cout, OH, OH)),
                                                 _chk(w10, torch.float32, "w10", (cin // 32, Cout // 16, 10, 4, 2, 16, 4)), dx.data_ptr(), B, cin, Cout, 2 * OH,
                                                 _stream()), "ee_conv3x3s2_pair_bwd_data_f32")
    return dx


def stem7x7s2_bwd_data(dy, weight, H, W):
    """d loss / d image through Conv2d(3, K, 7, stride 2, padding 3) (resnet.py:112): dy [B,K,H/2,W/2] -> [B,3,H,W]."""
    B, K = dy.shape[0], dy.shape[1]
    dx = torch.empty((B, 3, H, W), dtype=torch.float32, device=dy.device)
    N.check(N.lib.ee_stem7x7s2_bwd_data_f32(_chk(dy, torch.float32, "dy", (B, K, H // 2, W // 2)),
                                            _chk(weight, torch.float32, "weight", (K, 3, 7, 7)), dx.data_ptr(), B, K, H, W, _stream()),
            "ee_stem7x7s2_bwd_data_f32")
    return dx


def stem7x7s2_fwd_supported(x, weight):
    return (x.dim() == 4 and x.shape[1] == 3 and x.shape[2] % 2 == 0 and x.shape[3] % 64 == 0 and weight.shape[0] % 64 == 0
            and tuple(weight.shape[1:]) == (3, 7, 7) and weight.data_ptr() % 16 == 0)


def stem7x7s2_fwd(x, weight, want_stats=False):
    """Conv2d(3, K, 7, stride 2, padding 3) forward (resnet.py:112): x [B,3,H,W] -> [B,K,H/2,W/2] on the f32 matrix cores.
    want_stats: also (y, stats [K,S,3]) - per channel and workgroup (sum, M2, count) of y for the BatchNorm that follows."""
    B, _, H, W = x.shape
    K = weight.shape[0]
    y = torch.empty((B, K, H // 2, W // 2), dtype=torch.float32, device=x.device)
    stats = None
    if want_stats:
        n = N.lib.ee_stem7x7s2_fwd_stats_floats(B, K, H, W)
        stats = torch.empty((K, n // (3 * K), 3), dtype=torch.float32, device=x.device)
    N.check(N.lib.ee_stem7x7s2_fwd_stats_f32(_chk(x, torch.float32, "x", (B, 3, H, W)), _chk(weight, torch.float32, "weight", (K, 3, 7, 7)),
                                             y.data_ptr(), None if stats is None else stats.data_ptr(), B, K, H, W, _stream()),
            "ee_stem7x7s2_fwd_stats_f32")
    return (y, stats) if want_stats else y


def pool_linear_fwd(feat, weight, bias):
    """logits = fc(global_avgpool(feat)) (resnet.py:157-160) -> (logits [B,K], pooled [B,C])."""
    B, C = feat.shape[0], feat.shape[1]
    HW = feat[0, 0].numel() if B else 1
    K = weight.shape[0]
    pooled = torch.empty((B, C), dtype=torch.float32, device=feat.device)
    logits = torch.empty((B, K), dtype=torch.float32, device=feat.device)
    pb = None if bias is None else _chk(bias, torch.float32, "bias", (K,))
    N.check(N.lib.ee_pool_linear_fwd_f32(_chk(feat, torch.float32, "feat"), _chk(weight, torch.float32, "weight", (K, C)), pb,
                                         _chk(pooled, torch.float32, "pooled"), _chk(logits, torch.float32, "logits"), B, C, HW, K, _stream()),
            "ee_pool_linear_fwd_f32")
    return logits, pooled


def pool_linear_bwd(dlogits, weight, feat_shape):
    B, C = feat_shape[0], feat_shape[1]
    HW = 1
    for d in feat_shape[2:]:
        HW *= d
    K = weight.shape[0]
    dfeat = torch.empty(feat_shape, dtype=torch.float32, device=dlogits.device)
    N.check(N.lib.ee_pool_linear_bwd_f32(_chk(dlogits, torch.float32, "dlogits", (B, K)), _chk(weight, torch.float32, "weight", (K, C)),
                                         _chk(dfeat, torch.float32, "dfeat"), B, C, HW, K, _stream()), "ee_pool_linear_bwd_f32")
    return dfeat


def ce_pool_linear_bwd(logits, labels, weight, feat_shape, reduction="sum"):
    """d CrossEntropyLoss(logits, labels) / d feat for logits = fc(global_avgpool(feat)): the loss gradient is formed inside the head's backward
    launch (ee_head.hip) - the bits of ce(...) followed by pool_linear_bwd(...), one launch less"""
    B, C = feat_shape[0], feat_shape[1]
    HW = 1
    for d in feat_shape[2:]:
        HW *= d
    K = weight.shape[0]
    dfeat = torch.empty(feat_shape, dtype=torch.float32, device=logits.device)
    N.check(N.lib.ee_ce_pool_linear_bwd_f32(_chk(logits, torch.float32, "logits", (B, K)), _chk(labels, torch.int64, "labels", (B,)),
                                            _inv(B) if reduction == "mean" else 1.0, _chk(weight, torch.float32, "weight", (K, C)),
                                            _chk(dfeat, torch.float32, "dfeat"), B, C, HW, K, _stream()), "ee_ce_pool_linear_bwd_f32")
    return dfeat


# ---- Net_2's convolutional half (MNIST) -------------------------------------------------------------------------------------------
def net2_conv_supported(x, w1, w2):
    return (x.dim() == 4 and tuple(x.shape[1:]) == (1, 28, 28) and tuple(w1.shape) == (32, 1, 5, 5) and tuple(w2.shape) == (64, 32, 5, 5)
            and x.data_ptr() % 16 == 0)


def net2_conv_fwd(x, w1, b1, w2, b2, drop=None, keep=1.0, draw_state=None):
    """a2 = relu(max_pool2d(drop * conv2(relu(max_pool2d(conv1(x), 2))), 2)) (MNIST/models_mnist/Net2.py:13-14), two launches.
    Returns (a2 [B,64,4,4], saved, drop) - `saved` = (a1, code1, code2) for net2_conv_bwd.  drop: Dropout2d's [B,64] Bernoulli(keep) draw (0 / 1) or None;
    the kernels scale by drop / keep (what noise.div_(1 - p) does).  drop None + draw_state (runtime.draw_state): the kernel draws the mask itself and
    the returned `drop` is what it drew."""
    B = x.shape[0]
    dev = x.device
    a1 = torch.empty((B, 32, 12, 12), dtype=torch.float32, device=dev)
    c1 = torch.empty((B, 32, 12, 12), dtype=torch.uint8, device=dev)
    a2 = torch.empty((B, 64, 4, 4), dtype=torch.float32, device=dev)
    c2 = torch.empty((B, 64, 4, 4), dtype=torch.uint8, device=dev)
    ptr = lambda t: None if t is None else t.data_ptr()
    rng = drop is None and draw_state is not None
    drop_out = torch.empty((B, 64), dtype=torch.float32, device=dev) if rng else None
    N.check(N.lib.ee_net2_conv_fwd_f32(_chk(x, torch.float32, "x", (B, 1, 28, 28)), _chk(w1, torch.float32, "w1", (32, 1, 5, 5)),
                                       None if b1 is None else _chk(b1, torch.float32, "b1", (32,)), _chk(w2, torch.float32, "w2", (64, 32, 5, 5)),
                                       None if b2 is None else _chk(b2, torch.float32, "b2", (64,)),
                                       None if drop is None else _chk(drop, torch.float32, "drop", (B, 64)), float(keep),
                                       _chk(draw_state, torch.int64, "draw_state", (4,)) if rng else None, ptr(drop_out), ptr(a1), ptr(c1), ptr(a2),
                                       ptr(c2), B, _stream()), "ee_net2_conv_fwd_f32")
    return a2, (a1, c1, c2), (drop_out if rng else drop)


def net2_conv_bwd(da2, a2, saved, w1, w2, drop=None, keep=1.0, da1_out=None, need_dx=True):
    """d loss / d x [B,1,28,28] of net2_conv_fwd (input gradient only), two launches; da1_out: keeps the gradient of a1 for net2_conv_wrw."""
    a1, c1, c2 = saved
    B = a2.shape[0]
    da1 = torch.empty_like(a1) if da1_out is None else da1_out
    dx = torch.empty((B, 1, 28, 28), dtype=torch.float32, device=a2.device) if need_dx else None
    N.check(N.lib.ee_net2_conv_bwd_f32(_chk(da2, torch.float32, "da2", (B, 64, 4, 4)), _chk(a2, torch.float32, "a2"), _chk(c2, torch.uint8, "code2"),
                                       None if drop is None else _chk(drop, torch.float32, "drop", (B, 64)), float(keep), _chk(w2, torch.float32, "w2"),
                                       _chk(a1, torch.float32, "a1"), _chk(c1, torch.uint8, "code1"), _chk(w1, torch.float32, "w1"),
                                       da1.data_ptr(), None if dx is None else dx.data_ptr(), B, _stream()), "ee_net2_conv_bwd_f32")
    return dx


def net2_conv_wrw(x, da2, a2, saved, da1, drop=None, keep=1.0):
    """(dw1 [32,1,5,5], db1 [32], dw2 [64,32,5,5], db2 [64]) of net2_conv_fwd - views of one buffer: two launches + the fixed-order sum over the
    groups of ten images (bit-reproducible).  da1 = the gradient of a1 (net2_conv_bwd's da1_out)."""
    a1, c1, c2 = saved
    B = a2.shape[0]
    out = torch.empty(64 * 32 * 25 + 32 * 25 + 64 + 32, dtype=torch.float32, device=a2.device)
    n = int(N.lib.ee_net2_conv_wrw_workspace_floats(B))
    ws = torch.empty(max(n, 4), dtype=torch.float32, device=a2.device)
    N.check(N.lib.ee_net2_conv_wrw_f32(_chk(x, torch.float32, "x", (B, 1, 28, 28)), _chk(a1, torch.float32, "a1", (B, 32, 12, 12)), _chk(c1, torch.uint8, "code1"),
                                       _chk(da1, torch.float32, "da1", (B, 32, 12, 12)), _chk(a2, torch.float32, "a2", (B, 64, 4, 4)), _chk(c2, torch.uint8, "code2"),
                                       _chk(da2, torch.float32, "da2", (B, 64, 4, 4)), None if drop is None else _chk(drop, torch.float32, "drop", (B, 64)),
                                       float(keep), out.data_ptr(), ws.data_ptr(), B, _stream()), "ee_net2_conv_wrw_f32")
    n2, n1 = 64 * 32 * 25, 32 * 25
    return out[n2:n2 + n1].view(32, 1, 5, 5), out[n2 + n1 + 64:], out[:n2].view(64, 32, 5, 5), out[n2 + n1:n2 + n1 + 64]


# ---- timing hooks ------------------------------------------------------------------------------------------------------
def prof_enable(on=True):
    N.check(N.lib.ee_prof_enable(1 if on else 0), "ee_prof_enable")


def prof_mark_empty():
    N.check(N.lib.ee_prof_mark_empty(_stream()), "ee_prof_mark_empty")


def prof_reset():
    N.check(N.lib.ee_prof_reset(), "ee_prof_reset")


def prof_read(kernel_id):
    ms, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
    N.check(N.lib.ee_prof_read(kernel_id, ctypes.byref(ms), ctypes.byref(cnt)), "ee_prof_read")
    return ms.value, cnt.value


def prof_read_work(kernel_id):
    """Floating-point operations the timed launches of a matrix-core family declared (0 for the HBM-bound families)."""
    w = ctypes.c_double(0.0)
    N.check(N.lib.ee_prof_read_work(kernel_id, ctypes.byref(w)), "ee_prof_read_work")
    return w.value
