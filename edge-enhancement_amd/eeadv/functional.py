"""torch.autograd bridges onto the HIP kernels (eeadv.ops).  Everything here requires ROCm tensors."""
import weakref

import os

import torch

from . import ops


# ctx.needs_input_grad says whether an input REQUIRES grad, not whether this particular backward pass wants its gradient:
# a PGD iteration (torch.autograd.grad w.r.t. the input only, attacks.py:24) would still make the Functions below produce
# weight / bias gradients (a GEMM and a reduction per iteration for the classifier head).  The attack loop brackets its
# backward pass with input_grad_only(); the flag is a plain module global because autograd runs Function.backward on its own
# device thread while the calling thread blocks inside torch.autograd.grad.
_INPUT_GRAD_ONLY = False


class input_grad_only:
    def __enter__(self):
        global _INPUT_GRAD_ONLY
        self._prev, _INPUT_GRAD_ONLY = _INPUT_GRAD_ONLY, True

    def __exit__(self, *exc):
        global _INPUT_GRAD_ONLY
        _INPUT_GRAD_ONLY = self._prev


# The attack loop's forward passes (engine._body_input_grad) are differentiated with respect to the INPUT only.  Inside attack_forward()
# an eval-mode residual block may therefore take the kernels with BatchNorm folded in (EvalBasicBlockFn / EvalDownBlockFn below), which
# produce no parameter gradients at all; any other eval-mode forward with autograd on keeps the per-layer Functions.
_ATTACK_FORWARD = False


class attack_forward:
    def __enter__(self):
        global _ATTACK_FORWARD
        self._prev, _ATTACK_FORWARD = _ATTACK_FORWARD, True

    def __exit__(self, *exc):
        global _ATTACK_FORWARD
        _ATTACK_FORWARD = self._prev


def attack_forward_active():
    return _ATTACK_FORWARD


def input_only_forward():
    """True where a forward pass will never be asked for parameter gradients: inside the attack loop, or with autograd off"""
    return _ATTACK_FORWARD or not torch.is_grad_enabled()


# EEADV_STOCK_WRW=1: the weight gradients of the 3x3 layers (and the stride-2 shortcuts) from ATen / MIOpen instead of ee_wrw.hip (an A/B switch for measurements)
_STOCK_WRW = os.environ.get("EEADV_STOCK_WRW", "0") == "1"


# ee_wrw.hip's stem kernel wins on the 64x64 inputs of the Tiny-ImageNet configs (46 us against MIOpen's 63); on 224x224 (ResNet-50 free-AT, batch
# 32) MIOpen's searched solver is the faster one (559 against 557 img/s end to end), so wider images keep it
_STEM_WRW_MAXW = 64


def conv3x3_weight_grad(x, dy, weight):
    """d loss / d weight of conv3x3(x, weight) (stride 1, padding 1): ee_wrw.hip on the maps it takes (2 / 4 / 8 / 16 wide, channels % 32 == 0;
    bit-reproducible), ATen's convolution_backward otherwise."""
    if not _STOCK_WRW and ops.wrw3x3_supported(x, dy):
        return ops.wrw3x3(x, dy)
    return torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]


class Edge125Fn(torch.autograd.Function):
    """CannyFilter_step125_1 forward/backward (utils/core.py:549-585, To_compare :329-358)."""

    @staticmethod
    def forward(ctx, img, wts, alpha, high):
        img = img.contiguous()
        ctx.save_for_backward(img)
        ctx.cfg = (wts, alpha, high)
        return ops.edge125_fwd(img, wts, alpha, high)

    @staticmethod
    def backward(ctx, u):
        (img,) = ctx.saved_tensors
        wts, alpha, high = ctx.cfg
        g = ops.edge125_bwd(img, u.contiguous(), wts, alpha, high)
        return g.expand_as(img), None, None, None


class FrontEndFn(torch.autograd.Function):
    """x_in = clamp(x_hfs + w * edge125(x), 0, 1) in one kernel each way (resnet_EE.py:176-191)."""

    @staticmethod
    def forward(ctx, x, x_hfs, wts, alpha, high, w):
        x, x_hfs = x.contiguous(), x_hfs.contiguous()
        x_in, gate, _, sx, sy = ops.frontend_fwd_save(x, x_hfs, wts, alpha, high, w)
        ctx.save_for_backward(gate, sx, sy)  # the Sobel responses instead of x: the backward recomputes nothing
        ctx.cfg = (wts, alpha, high, w)
        return x_in

    @staticmethod
    def backward(ctx, g_in):
        gate, sx, sy = ctx.saved_tensors
        wts, alpha, high, w = ctx.cfg
        g_hfs, g_edge = ops.frontend_bwd_saved(g_in.contiguous(), gate, sx, sy, wts, alpha, high, w)
        gx = g_edge.expand_as(g_hfs) if ctx.needs_input_grad[0] else None
        return gx, g_hfs, None, None, None, None


class CannyFn(torch.autograd.Function):
    """Full CannyFilter forward/backward (utils/core.py:222-326; thresholds given, hysteresis=True)."""

    @staticmethod
    def forward(ctx, img, wts, alpha, low, high):
        img = img.contiguous()
        ctx.save_for_backward(img)
        ctx.cfg = (wts, alpha, low, high)
        return ops.canny_fwd(img, wts, alpha, low, high)

    @staticmethod
    def backward(ctx, u):
        (img,) = ctx.saved_tensors
        wts, alpha, low, high = ctx.cfg
        return ops.canny_bwd(img, u.contiguous(), wts, alpha, low, high).expand_as(img), None, None, None, None


class CannyBPDAFn(torch.autograd.Function):
    """CannyFilter_BPDA forward/backward (utils/core.py:386-505; thresholds given, hysteresis=True)."""

    @staticmethod
    def forward(ctx, img, wts, low, high):
        img = img.contiguous()
        edge, thin, t2 = ops.canny_bpda_fwd(img, wts, low, high)
        ctx.save_for_backward(img, thin, t2)
        ctx.cfg = (wts, low, high)
        return edge

    @staticmethod
    def backward(ctx, u):
        img, thin, t2 = ctx.saved_tensors
        wts, low, high = ctx.cfg
        return ops.canny_bpda_bwd(img, u.contiguous(), thin, t2, wts, low, high).expand_as(img), None, None, None


class CannyFrontEndFn(torch.autograd.Function):
    """x_in = clamp(x_hfs + w * CannyFilter(x), 0, 1) in one kernel each way (Net2_EE.py:36-49, resnet_EE.py:176-191)."""

    @staticmethod
    def forward(ctx, x, x_hfs, wts, alpha, low, high, w):
        x, x_hfs = x.contiguous(), x_hfs.contiguous()
        x_in, gate, _ = ops.canny_frontend_fwd(x, x_hfs, wts, alpha, low, high, w)
        ctx.save_for_backward(x, gate)
        ctx.cfg = (wts, alpha, low, high, w)
        return x_in

    @staticmethod
    def backward(ctx, g_in):
        x, gate = ctx.saved_tensors
        wts, alpha, low, high, w = ctx.cfg
        g_hfs, g_edge = ops.canny_frontend_bwd(g_in.contiguous(), gate, x, wts, alpha, low, high, w)
        gx = g_edge.expand_as(x) if ctx.needs_input_grad[0] else None
        return gx, g_hfs, None, None, None, None, None


class AddSquareFn(torch.autograd.Function):
    """Add_Square (utils/core.py:636-655) given its random draws."""

    @staticmethod
    def forward(ctx, x, eps, stripe, sq_sign, sq_pos, sq_size):
        x = x.contiguous()
        ctx.save_for_backward(x, stripe, sq_sign, sq_pos, sq_size)
        ctx.eps = eps
        return ops.add_square_fwd(x, eps, stripe, sq_sign, sq_pos, sq_size)

    @staticmethod
    def backward(ctx, g):
        x, stripe, sq_sign, sq_pos, sq_size = ctx.saved_tensors
        return ops.add_square_bwd(g.contiguous(), x, ctx.eps, stripe, sq_sign, sq_pos, sq_size), None, None, None, None, None




def _two_pieces(grads):
    """(dy, dy2) of a forked output: either piece may be missing (its consumer needed no gradient)."""
    g = [t.contiguous() for t in grads if t is not None]
    return (g[0] if g else None), (g[1] if len(g) > 1 else None)


class BnActFn(torch.autograd.Function):
    """[relu]( batch_norm(x) [+ residual] ) in one launch each way (resnet.py:44-59 / :90-110; ee_bn.hip).  `fork=True` returns the
    output TWICE (two tensors over one buffer): a residual block's output feeds the next block's convolution and its identity branch,
    autograd then hands the two gradients over separately and the backward kernel adds them on load - otherwise the engine sums them in
    a launch of its own (8 per ResNet-18 pass)."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, relu, fork=False):
        y, sm, si = ops.bn_act_fwd(x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, relu)
        # the backward's ReLU mask is y > 0; without a residual it is recomputed from x (the forward's expression), and y is not kept for it
        keep_y = relu and residual is not None
        ctx.save_for_backward(x, y if keep_y else None, gamma, sm, si, None if training else running_mean, None if training else running_var,
                              beta if relu and not keep_y else None)
        ctx.cfg = (eps, training, relu, residual is not None)
        ctx.set_materialize_grads(False)
        return (y, y.view_as(y)) if fork else y

    @staticmethod
    def backward(ctx, *grads):
        x, y, gamma, sm, si, rm, rv, beta = ctx.saved_tensors
        eps, training, relu, has_res = ctx.cfg
        want = list(ctx.needs_input_grad)
        if _INPUT_GRAD_ONLY:
            want[2] = want[3] = False
        none = (None,) * 11
        dy, dy2 = _two_pieces(grads)
        if dy is None:
            return none
        if not relu and not want[0] and not want[2] and not want[3]:
            dres = None
            if has_res and want[1]:
                dres = dy if dy2 is None else dy + dy2
            return (None, dres) + none[2:]
        want_dres = has_res and want[1] and (relu or dy2 is not None)  # without the ReLU (and in one piece) the residual's gradient IS dy: no copy needed
        dx, dres, dg, db = ops.bn_act_bwd(dy, y, x, gamma, sm, si, rm, rv, eps, training, relu, want[0], want_dres,
                                          want[2] or want[3], dy2, beta)
        if has_res and want[1] and not want_dres:
            dres = dy
        return (dx, dres, (dg if want[2] else None), (db if want[3] else None)) + none[4:]


class BnDualFn(torch.autograd.Function):
    """relu(bn_a(xa) + bn_b(xb)): the end of a residual block with a down-sampling shortcut (resnet.py:54-59, :137-142) - the block's
    second BatchNorm and the shortcut's in ONE launch each way (ee_bn.hip, bn_dual_*), bit-identical to the two BnActFn calls.
    fork: see BnActFn."""

    @staticmethod
    def forward(ctx, xa, xb, ga, ba, rma, rva, mom_a, eps_a, gb, bb, rmb, rvb, mom_b, eps_b, training, fork=False):
        y, saves = ops.bn_dual_fwd(xa, xb, (ga, ba, rma, rva, mom_a, eps_a), (gb, bb, rmb, rvb, mom_b, eps_b), training)
        ctx.save_for_backward(xa, xb, y, ga, gb, *saves, *((None,) * 4 if training else (rma, rva, rmb, rvb)))
        ctx.cfg = (eps_a, eps_b, training)
        ctx.set_materialize_grads(False)
        return (y, y.view_as(y)) if fork else y

    @staticmethod
    def backward(ctx, *grads):
        xa, xb, y, ga, gb, sma, sia, smb, sib, rma, rva, rmb, rvb = ctx.saved_tensors
        eps_a, eps_b, training = ctx.cfg
        need = ctx.needs_input_grad
        none = (None,) * 16
        dy, dy2 = _two_pieces(grads)
        want_params = (need[2] or need[3] or need[8] or need[9]) and not _INPUT_GRAD_ONLY
        if dy is None or not (need[0] or need[1] or want_params):
            return none
        dxa, dxb, dga, dba, dgb, dbb = ops.bn_dual_bwd(dy, dy2, y, xa, xb, ga, gb, (sma, sia, smb, sib), rma, rva, rmb, rvb, eps_a, eps_b, training,
                                                       need[0], need[1], want_params)
        keep = lambda t, i: t if (need[i] and want_params) else None
        return (dxa, dxb, keep(dga, 2), keep(dba, 3), None, None, None, None, keep(dgb, 8), keep(dbb, 9)) + none[10:]


_POOL_XA = os.environ.get("EEADV_POOL_XA", "1") != "0"  # the stem's BatchNorm backward sums from pooled tensors (A/B switch)


class BnReluPoolFn(torch.autograd.Function):
    """maxpool3s2(relu(batch_norm(x))) - the ResNet stem (resnet.py:113-117) - in one pass each way: the full-resolution activation and its
    gradient never exist (ee_bn.hip: bn_pool_*)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, training, fork=False, conv_stats=None):
        if conv_stats is not None and conv_stats.numel() == 0:
            conv_stats = None
        xa = None
        if training and _POOL_XA:  # x at every window's argmax: the backward's batch sums then come from the pooled tensors (one pass over x less)
            y, code, sm, si, xa = ops.bn_relu_pool_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, training, conv_stats, True)
        else:
            y, code, sm, si = ops.bn_relu_pool_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, training, conv_stats)
        ctx.save_for_backward(x, code, gamma, beta, sm, si, None if training else running_mean, None if training else running_var, xa)
        ctx.cfg = (eps, training)
        ctx.set_materialize_grads(False)
        return (y, y.view_as(y)) if fork else y  # fork: see BnActFn

    @staticmethod
    def backward(ctx, *grads):
        x, code, gamma, beta, sm, si, rm, rv, xa = ctx.saved_tensors
        eps, training = ctx.cfg
        want_params = (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not _INPUT_GRAD_ONLY
        dy, dy2 = _two_pieces(grads)
        if dy is None or (not ctx.needs_input_grad[0] and not want_params):
            return (None,) * 10
        dx, dg, db = ops.bn_relu_pool_bwd(dy, code, x, gamma, beta, sm, si, rm, rv, eps, training, ctx.needs_input_grad[0], want_params, dy2, xa)
        return (dx, (dg if ctx.needs_input_grad[1] and want_params else None), (db if ctx.needs_input_grad[2] and want_params else None)) + (None,) * 7


class MaxPool3s2Fn(torch.autograd.Function):
    """MaxPool2d(3, stride 2, padding 1) with a one-byte argmax code (resnet.py:117; ee_pool.hip)."""

    @staticmethod
    def forward(ctx, x):
        y, code = ops.maxpool3s2_fwd(x)
        ctx.save_for_backward(code)
        ctx.hw = (x.shape[2], x.shape[3])
        return y

    @staticmethod
    def backward(ctx, dy):
        (code,) = ctx.saved_tensors
        return ops.maxpool3s2_bwd(dy.contiguous(), code, *ctx.hw)


class Conv1x1S2Fn(torch.autograd.Function):
    """The shortcut Conv2d(Cin, Cout, 1, stride=2, bias=False) (resnet.py:137-142) on ee_conv.hip; the weight gradient
    (once per training step) comes from ATen's convolution_backward."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return ops.conv1x1s2_fwd(x, weight)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.conv1x1s2_bwd(dy, weight, x.shape[2], x.shape[3]) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1] and not _INPUT_GRAD_ONLY:
            dw = torch.ops.aten.convolution_backward(dy, x, weight, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return dx, dw




class StemConvFn(torch.autograd.Function):
    """The stem Conv2d(3, 64, 7, stride 2, padding 3, bias=False) (resnet.py:112-113): the forward (maps whose width is a multiple
    of 64) and the gradient with respect to the image - what the attack loop is after - on ee_conv.hip; the weight gradient: ee_wrw.hip."""

    @staticmethod
    def forward(ctx, x, weight, want_stats=False):
        """want_stats: returns (y, stats) - stats = the per-workgroup moments of y for bn1 (ops.stem7x7s2_fwd), an EMPTY tensor when the
        convolution ran on MIOpen (the BatchNorm then takes its statistics itself)."""
        ctx.save_for_backward(x, weight)
        ctx.set_materialize_grads(False)  # or autograd zero-fills a gradient for `stats` on every backward pass (one launch)
        if ops.stem7x7s2_fwd_supported(x, weight):
            if want_stats:
                y, stats = ops.stem7x7s2_fwd(x, weight, True)
                ctx.mark_non_differentiable(stats)
                return y, stats
            y = ops.stem7x7s2_fwd(x, weight)
        else:
            y = torch.ops.aten.convolution(x, weight, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1)
        if want_stats:
            stats = x.new_empty(0)
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, dy, *_):
        x, weight = ctx.saved_tensors
        if dy is None:
            return None, None, None
        dy = dy.contiguous()
        dx = ops.stem7x7s2_bwd_data(dy, weight, x.shape[2], x.shape[3]) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1] and not _INPUT_GRAD_ONLY:
            if not _STOCK_WRW and ops.wrw_stem7x7s2_supported(x, dy) and x.shape[3] <= _STEM_WRW_MAXW:
                dw = ops.wrw_stem7x7s2(x, dy)
            else:
                dw = torch.ops.aten.convolution_backward(dy, x, weight, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return dx, dw, None


# ---- a 3x3 / stride 1 / padding 1 convolution on a 2x2 map is one dense product ----------------------------------------------
# Every input pixel reaches every output pixel, so y[n, (co,oy,ox)] = sum_{(ci,iy,ix)} x[n, (ci,iy,ix)] * W[co, ci, iy-oy+1, ix-ox+1]:
# a [B, 4 Cin] x [4 Cin, 4 Cout] GEMM with no zero padding in it (5/9 of a 3x3 window falls outside a 2x2 map).  ResNet-18's
# layer4 at 64x64 input: 15 us on the BLAS against 41 us (forward) / 45 us (backward-data) for MIOpen's best solver.
# The rearranged weight matrix is rebuilt IN PLACE when the weight's version counter moves (once per optimiser step), so
# captured HIP graphs keep pointing at the same buffer; callers that replay graphs call refresh_dense_weights() first.
_DENSE_W = {}  # id(weight) -> [weakref(weight), version, W2 buffer]
_DENSE_IDX = {}


def _rearranged(weight, kind="s1", extra=None):
    """kind "s1": 3x3 / stride 1 / padding 1 on a 2x2 map -> [4 Cin, 4 Cout]; kind "s2": 3x3 / stride 2 / padding 1 from a 4x4 map to
    a 2x2 map -> [16 Cin, 4 Cout] (tap ky = iy - 2 oy + 1, zero where it leaves the 3x3 window)."""
    co, ci = weight.shape[0], weight.shape[1]
    if kind == "wino_fb":
        return torch.stack([_rearranged(weight, "wino_f").reshape(-1), _rearranged(weight, "wino_b").reshape(-1)])
    if kind in ("wino_f", "wino_b"):
        # Winograd F(2x2, 3x3) filter transform U = G g G^T, laid out [16][KC][RC] for ee_wino.hip: forward g = w[r][k] (k = input channel),
        # backward-data g = w[k][r] rotated by 180 degrees (k = output channel)
        G = _DENSE_IDX.get((weight.device, "G"))
        if G is None:
            G = _DENSE_IDX[(weight.device, "G")] = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]],
                                                                 dtype=weight.dtype, device=weight.device)
        w = weight.detach()
        if kind == "wino_f":
            return torch.einsum("ia,jb,rkab->ijkr", G, G, w).reshape(16, ci, co)
        return torch.einsum("ia,jb,krab->ijkr", G, G, w.flip(2, 3)).reshape(16, co, ci)
    if kind in ("s2m_f", "s2m_b", "s2p_f", "s2p_b"):
        # ee_s2.hip's A operands in reading order: [result block of 32][round of 16 reduction channels][tap][quad][half][m][k]; "s2p_*": the
        # block's shortcut 1x1 filters (`extra`, [co, ci, 1, 1]) as a tenth tap
        w, taps = weight.detach().reshape(co, ci, 9), 9
        if kind in ("s2p_f", "s2p_b"):
            w, taps = torch.cat([w, extra.detach().reshape(co, ci, 1)], 2), 10
        if kind in ("s2m_f", "s2p_f"):  # result = co = 32 cb + 16 h + m, reduction = ci = 16 rd + 4 q + k
            return w.view(co // 32, 2, 16, ci // 16, 4, 4, taps).permute(0, 3, 6, 4, 1, 2, 5)
        return w.view(co // 16, 4, 4, ci // 32, 2, 16, taps).permute(3, 0, 6, 1, 4, 5, 2)  # result = ci, reduction = co
    if kind == "s1t":  # the dense 2x2 matrix transposed: what ee_dense.hip's backward-data product reads ([4 Cout, 4 Cin])
        return _rearranged(weight, "s1").t().contiguous()
    idx = _DENSE_IDX.get((weight.device, kind))
    if idx is None:
        n_in, stride = (2, 1) if kind == "s1" else (4, 2)
        i, o = torch.arange(n_in, device=weight.device), torch.arange(2, device=weight.device)
        # k[iy, ix, oy, ox] = iy - stride * oy + 1 (rows) / ix - stride * ox + 1 (columns)
        ky = (i.view(n_in, 1, 1, 1) - stride * o.view(1, 1, 2, 1) + 1).expand(n_in, n_in, 2, 2)
        kx = (i.view(1, n_in, 1, 1) - stride * o.view(1, 1, 1, 2) + 1).expand(n_in, n_in, 2, 2)
        ok = ((ky >= 0) & (ky <= 2) & (kx >= 0) & (kx <= 2))
        idx = _DENSE_IDX[(weight.device, kind)] = (ky.clamp(0, 2).contiguous(), kx.clamp(0, 2).contiguous(),
                                                   None if bool(ok.all()) else ok.to(weight.dtype).contiguous(), n_in * n_in)
    g = weight.detach()[:, :, idx[0], idx[1]]  # [co, ci, iy, ix, oy, ox]
    if idx[2] is not None:
        g = g * idx[2]
    return g.permute(1, 2, 3, 0, 4, 5).reshape(idx[3] * ci, 4 * co)


def _versions(weight, extra):
    return weight._version if extra is None else (weight._version, extra._version)


# kinds ee_wprep.hip builds in one launch (EE_WPREP_* of eeadv.h); _rearranged's torch expressions are their restatement (tests) and the
# path of anything the kernel does not take
_NATIVE_KIND = {"wino_f": 0, "wino_b": 1, "s2m_f": 2, "s2m_b": 3, "s2p_f": 4, "s2p_b": 5, "s1": 6, "wino_fb": 7}


def _rearranged_shape(weight, kind):
    co, ci = weight.shape[0], weight.shape[1]
    if kind in ("wino_f", "wino_b"):
        return (16, ci, co) if kind == "wino_f" else (16, co, ci)
    if kind == "wino_fb":  # [forward set | backward-data set], each 16 * ci * co floats (wino_sets() hands out the two views)
        return (2, 16 * ci * co)
    if kind in ("s2m_f", "s2p_f"):
        return (co // 32, ci // 16, 10 if kind == "s2p_f" else 9, 4, 2, 16, 4)
    if kind in ("s2m_b", "s2p_b"):
        return (ci // 32, co // 16, 10 if kind == "s2p_b" else 9, 4, 2, 16, 4)
    return (4 * ci, 4 * co)  # "s1"


def _fill_rearranged(buf, weight, kind, extra):
    """buf <- the rearranged copy, in place (capturable): one hand-written launch where ee_wprep.hip knows the kind"""
    if (_native_kind(weight, kind) and (extra is None or (extra.is_contiguous() and extra.dtype == torch.float32))):
        ops.conv_weight_prep(_NATIVE_KIND[kind], weight.detach(), None if extra is None else extra.detach(), buf)
    else:
        buf.copy_(_rearranged(weight, kind, extra))


def _native_kind(weight, kind):
    return (kind in _NATIVE_KIND and weight.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous()
            and (kind != "wino_fb" or (weight.shape[0] % 32 == 0 and weight.shape[1] % 32 == 0)))


def wino_sets(weight):
    """(u forward [16, Cin, Cout], u backward-data [16, Cout, Cin]) of a 3x3 weight: the two halves of ONE cached buffer that one ee_wprep.hip
    launch rebuilds (round 3; two entries and two launches before)"""
    co, ci = weight.shape[0], weight.shape[1]
    both = _dense_weight(weight, "wino_fb")
    return both[0].view(16, ci, co), both[1].view(16, co, ci)


def _new_rearranged(weight, kind, extra):
    if _native_kind(weight, kind):
        buf = torch.empty(_rearranged_shape(weight, kind), dtype=torch.float32, device=weight.device)
        _fill_rearranged(buf, weight, kind, extra)
        return buf
    return _rearranged(weight, kind, extra).contiguous()


def _dense_weight(weight, kind="s1", extra=None):
    """The rearranged copy of `weight` (and, for the "s2p_*" kinds, of a second parameter `extra` riding along) for `kind`: built on first
    use, rebuilt IN PLACE when a version counter moved; inside a graph capture only handed out."""
    ent = _DENSE_W.get((id(weight), kind))
    if ent is not None and (ent[0]() is not weight or (ent[4]() if ent[4] is not None else None) is not extra):
        ent = None
    if torch.cuda.is_current_stream_capturing():
        if ent is None:
            raise RuntimeError("rearranged convolution weights: first use inside a graph capture (run one eager forward first)")
        return ent[2]
    if ent is None or ent[1] != _versions(weight, extra):
        with torch.no_grad():
            if ent is None:
                ent = _DENSE_W[(id(weight), kind)] = [weakref.ref(weight), _versions(weight, extra), _new_rearranged(weight, kind, extra), kind,
                                                      None if extra is None else weakref.ref(extra)]
            else:
                _fill_rearranged(ent[2], weight, kind, extra)
                ent[1] = _versions(weight, extra)
    return ent[2]


def _dense_entry_params(ent):
    """(weight, extra) of a cache entry, (None, None) once either is gone"""
    w = ent[0]()
    e = ent[4]() if ent[4] is not None else None
    if w is None or (ent[4] is not None and e is None):
        return None, None
    return w, e


def rebuild_dense_weights(model=None):
    """Unconditional in-place rebuild of the rearranged matrices - capturable: a captured optimiser step ends with it, because
    replaying a graph updates the weights without moving their Python-side version counters.  `model`: only ITS weights - a
    captured graph must not bake in copies into the buffers of another live model (tests, A/B scripts), which would write
    freed memory once that model is gone.  Round 4: ONE ee_wprep.hip launch for all (weight, kind) items of the model, their descriptors in a
    device-resident table (ee_conv_weight_prep_batch_f32) where the captured update of ResNet-18 used to end with 19 launches - measured the
    SAME 120 us either way: the gathers (16 MB of rearranged filters written per update), not the launches, are what this costs.
    Returns the cache keys it rebuilt."""
    own = None if model is None else {id(p) for p in model.parameters()}
    rebuilt, batch = set(), []
    with torch.no_grad():
        for key in list(_DENSE_W):
            ent = _DENSE_W[key]
            w, e = _dense_entry_params(ent)
            if w is None:
                del _DENSE_W[key]
            elif own is None or key[0] in own:
                if _native_kind(w, ent[3]) and (e is None or (e.is_contiguous() and e.dtype == torch.float32)):
                    batch.append((_NATIVE_KIND[ent[3]], w.detach(), None if e is None else e.detach(), ent[2]))  # round 4: ONE launch for all of them
                else:
                    _fill_rearranged(ent[2], w, ent[3], e)
                ent[1] = _versions(w, e)
                rebuilt.add(key)
        ops.conv_weight_prep_batch(batch)
    return rebuilt


def invalidate_params(param_ids):
    """Mark the cached rearranged copies of these parameters stale: for updates that move weights without advancing their version
    counters (torch's fused SGD - trainer._FusedSGD calls this after every step)."""
    for key, ent in _DENSE_W.items():
        if key[0] in param_ids:
            ent[1] = None


def invalidate_dense_except(param_ids, keys):
    """After the REPLAY of a captured optimiser step that ends with rebuild_dense_weights: the replay moved the weights without moving
    their version counters and rebuilt only the cache entries that existed at capture (`keys`).  Entries of the same parameters
    (`param_ids`) created later - a first forward at another input size takes another kernel kind - are marked stale here, so that
    the next refresh_dense_weights() / _dense_weight() rebuilds them instead of trusting an unchanged version counter."""
    for key, ent in _DENSE_W.items():
        if key[0] in param_ids and key not in keys:
            ent[1] = None


def refresh_dense_weights():
    """Bring every rearranged weight matrix up to date (eagerly, outside any capture): call before replaying a HIP graph
    that contains Conv3x3Map2Fn."""
    for key in list(_DENSE_W):
        ent = _DENSE_W[key]
        w, e = _dense_entry_params(ent)
        if w is None:
            del _DENSE_W[key]
        elif ent[1] != _versions(w, e):
            _dense_weight(w, ent[3], e)


class Conv1x1Fn(torch.autograd.Function):
    """Conv2d(1x1, stride 1, bias=False) of the bottleneck blocks (resnet.py:75-100): forward and backward-data stay ATen's (MIOpen / rocBLAS run
    them as plain GEMMs), the WEIGHT gradient is ee_wrw.hip's NCHW product (MIOpen's searched solvers for it are NHWC kernels behind layout
    transposes and zero fills: 31 % of a free-AT repeat on ResNet-50)."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return torch.ops.aten.convolution(x, weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        want_w = ctx.needs_input_grad[1] and not _INPUT_GRAD_ONLY
        own = want_w and not _STOCK_WRW and ops.wrw1x1_supported(x, dy)
        dx, dw, _ = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                        [ctx.needs_input_grad[0], want_w and not own, False])
        if own:
            dw = ops.wrw1x1(x, dy)
        return dx, dw


class Conv3x3WinoFn(torch.autograd.Function):
    """Conv2d(3x3, stride 1, padding 1, bias=False) on 8x8 maps (ResNet-18 layer2 at 64x64 inputs, resnet.py:26-31): forward and
    backward-data as Winograd F(2x2, 3x3) around the f32 matrix cores (ee_wino.hip); the transformed filters follow the weight's version
    counter like the dense matrices above (rebuilt inside a captured optimiser step); weight gradient: ee_wrw.hip."""

    @staticmethod
    def forward(ctx, x, weight):
        u = wino_sets(weight)[0]  # (both sets are created here, outside any capture; the backward only reads its half)
        ctx.save_for_backward(x, weight)
        return ops.wino3x3(x, u)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.wino3x3(dy, wino_sets(weight)[1]) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1] and not _INPUT_GRAD_ONLY:
            dw = conv3x3_weight_grad(x, dy, weight)
        return dx, dw


def _bn_fwd_consts(g, b, rm, rv, eps):
    return (rm, rv, g, b, eps)


def _eval_conv_fwd(x, w, bn, res):
    """relu(bn(conv3x3(x)) [+ res]) with bn in eval mode, one launch: Winograd on 4 / 8 / 16-wide maps, the dense product on 2x2 maps"""
    if x.shape[2] == 2:
        if _DENSE_FOLD_BWD:
            _dense_weight(w, "s1t")  # created outside any capture; the backward only reads it
        return ops.dense2x2_bn_eval_fwd(x, _dense_weight(w, "s1"), bn, res, True)
    return ops.wino3x3_bn_eval_fwd(x, wino_sets(w)[0], bn, res, True)


# 2x2 maps, backward: ee_dense2x2_bn_eval_bwd_f32 is built and tested, but folding the BatchNorm / ReLU backward into the product's A staging makes
# each of its 64 column-tile workgroups read the gradient AND the mask (196 MB through L2 per launch instead of 128): 26.4 us against 21.4 for the
# BatchNorm launch + Tensile's product (profiles/round4_d_dense_probe.txt).  The forward fold wins (17.2 against 20.0 us) and is what runs.
_DENSE_FOLD_BWD = os.environ.get("EEADV_DENSE_FOLD_BWD", "0") == "1"


def _eval_conv_bwd(dy, dy2, y, w, bn, want_dres, dx_add=None):
    """bn = (running_var, gamma, eps, running_mean): the backward of _eval_conv_fwd with respect to x -> (dx [+ dx_add], dz or None)"""
    if dy.shape[2] == 2:
        if _DENSE_FOLD_BWD:
            return ops.dense2x2_bn_eval_bwd(dy, dy2, y, _dense_weight(w, "s1t"), bn[:3], want_dres, dx_add)
        # the BatchNorm / ReLU backward launch (x only enters as xhat * 0: y stands in for it), then Tensile's product with dx_add as its C operand
        dzs, dz, _, _ = ops.bn_act_bwd(dy, y, y, bn[1], None, None, bn[3], bn[0], bn[2], False, True, True, want_dres, False, dy2)
        B = dy.shape[0]
        w2 = _dense_weight(w, "s1")
        dx = torch.mm(dzs.reshape(B, -1), w2.t()) if dx_add is None else torch.addmm(dx_add.reshape(B, -1), dzs.reshape(B, -1), w2.t())
        return dx.view(B, w.shape[1], 2, 2), dz
    return ops.wino3x3_bn_eval_bwd(dy, dy2, y, wino_sets(w)[1], bn[:3], want_dres, dx_add)


class EvalBasicBlockFn(torch.autograd.Function):
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + x): a BasicBlock without shortcut convolution (resnet.py:44-59) under model.eval(), TWO launches
    each way - the Winograd kernels with the running-statistics BatchNorm, the residual and the ReLU in their output transform (forward)
    and the BatchNorm / ReLU backward in their input staging (backward-data): ee_wino3x3_bn_eval_*.  Bit-identical to the per-layer
    sequence (Conv3x3WinoFn, BnActFn) it replaces: four launches each way.  Input gradient only: the parameters get None
    (models.BasicBlock takes this path where functional.input_only_forward() holds).  x arrives as the two tensors of a forked output;
    ONE summed gradient goes back (to the first)."""

    @staticmethod
    def forward(ctx, xa, xb, w1, w2, g1, b1, rm1, rv1, eps1, g2, b2, rm2, rv2, eps2, fork):
        out1 = _eval_conv_fwd(xa, w1, _bn_fwd_consts(g1, b1, rm1, rv1, eps1), None)
        out2 = _eval_conv_fwd(out1, w2, _bn_fwd_consts(g2, b2, rm2, rv2, eps2), xa)
        ctx.save_for_backward(out1, out2, w1, w2, g1, rv1, g2, rv2, rm1, rm2)
        ctx.eps = (eps1, eps2)
        ctx.set_materialize_grads(False)
        return (out2, out2.view_as(out2)) if fork else out2

    @staticmethod
    def backward(ctx, *grads):
        out1, out2, w1, w2, g1, rv1, g2, rv2, rm1, rm2 = ctx.saved_tensors
        none = (None,) * 15
        dy, dy2 = _two_pieces(grads)
        if dy is None or not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            return none
        d1, dres = _eval_conv_bwd(dy, dy2, out2, w2, (rv2, g2, ctx.eps[1], rm2), True)
        dx, _ = _eval_conv_bwd(d1, None, out1, w1, (rv1, g1, ctx.eps[0], rm1), False, dx_add=dres)
        return (dx,) + none[1:]


class EvalDownBlockFn(torch.autograd.Function):
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + bn_ds(conv_ds(x))): a down-sampling BasicBlock (resnet.py:44-59, :137-142) under model.eval(),
    TWO launches each way: ee_conv3x3s2_pair_bn_eval_* (both stride-2 convolutions with their BatchNorms) and ee_wino3x3_bn_eval_* (conv2,
    bn2, the shortcut's sum, the ReLU).  Bit-identical to Conv3x3S2PairFn, BnActFn, Conv3x3WinoFn, BnDualFn - four launches each way."""

    @staticmethod
    def forward(ctx, xa, xb, w3, wd, w2, g1, b1, rm1, rv1, eps1, gd, bd, rmd, rvd, epsd, g2, b2, rm2, rv2, eps2, fork):
        w10 = _dense_weight(w3, "s2p_f", wd)
        _dense_weight(w3, "s2p_b", wd)  # created outside any capture; the backward only reads it
        out1, sc = ops.conv3x3s2_pair_bn_eval_fwd(xa, w10, w3.shape[0], _bn_fwd_consts(g1, b1, rm1, rv1, eps1), _bn_fwd_consts(gd, bd, rmd, rvd, epsd))
        out2 = _eval_conv_fwd(out1, w2, _bn_fwd_consts(g2, b2, rm2, rv2, eps2), sc)
        ctx.save_for_backward(out1, out2, w3, wd, w2, g1, rv1, gd, rvd, g2, rv2, rm2)
        ctx.eps = (eps1, epsd, eps2)
        ctx.set_materialize_grads(False)
        return (out2, out2.view_as(out2)) if fork else out2

    @staticmethod
    def backward(ctx, *grads):
        out1, out2, w3, wd, w2, g1, rv1, gd, rvd, g2, rv2, rm2 = ctx.saved_tensors
        none = (None,) * 21
        dy, dy2 = _two_pieces(grads)
        if dy is None or not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            return none
        d1, dsc = _eval_conv_bwd(dy, dy2, out2, w2, (rv2, g2, ctx.eps[2], rm2), True)
        dx = ops.conv3x3s2_pair_bn_eval_bwd(d1, out1, dsc, _dense_weight(w3, "s2p_b", wd), w3.shape[1], (rv1, g1, ctx.eps[0]), (rvd, gd, ctx.eps[1]))
        return (dx,) + none[1:]


_TRAIN_BWD_BOUNDARY = os.environ.get("EEADV_TRAIN_BWD_BOUNDARY", "1") == "1"  # 0: the BatchNorm backward launch of round 4's first form (A/B)


class TrainConvBnConvFn(torch.autograd.Function):
    """conv2(relu(bn1(conv1(x)))) of a BasicBlock (resnet.py:44-49) in TRAIN mode inside the attack loop, TWO launches forward: conv1's output
    transform also writes per-image moments, conv2's prologue merges them (the batch statistics: a grid-wide exchange through the kernel
    boundary instead of a BatchNorm launch), normalises and applies the ReLU while it stages its input (ee_wino3x3_stats_f32 /
    ee_wino3x3_bn_train_pre_f32).  The statistics are summed in another order than ee_bn.hip's: rounding-level difference, bit-reproducible.
    Backward (input gradient only; the parameters get None): conv2^T, the BatchNorm / ReLU backward kernel on the saved statistics, conv1^T."""

    @staticmethod
    def forward(ctx, x, w1, w2, gamma, beta, running_mean, running_var, momentum, eps):
        c1, stats = ops.wino3x3_stats(x, wino_sets(w1)[0])
        c2, sm, si = ops.wino3x3_bn_train_pre(c1, stats, x.shape[2] * x.shape[3], gamma, beta, eps, momentum, running_mean, running_var, wino_sets(w2)[0])
        ctx.save_for_backward(c1, w1, w2, gamma, beta, sm, si)
        ctx.eps = eps
        return c2

    @staticmethod
    def backward(ctx, dc2):
        c1, w1, w2, gamma, beta, sm, si = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return (None,) * 9
        if c1.shape[2] == 16 and c1.shape[1] <= 128 and _TRAIN_BWD_BOUNDARY:
            # the BatchNorm's backward crosses the kernel boundary too: conv2^T writes the per-image sums of dz and dz * xhat next to its output,
            # conv1^T merges them and forms the BatchNorm's input gradient while it stages (two launches instead of three)
            d_a1, sums = ops.wino3x3_bwd_sums(dc2.contiguous(), wino_sets(w2)[1], c1, sm, si, gamma, beta)
            return (ops.wino3x3_bn_train_bwd_pre(d_a1, c1, sums, 256, sm, si, gamma, beta, wino_sets(w1)[1]),) + (None,) * 8
        d_a1 = ops.wino3x3(dc2.contiguous(), wino_sets(w2)[1])
        d_c1 = ops.bn_act_bwd(d_a1, None, c1, gamma, sm, si, None, None, ctx.eps, True, True, True, False, False, None, beta)[0]
        return (ops.wino3x3(d_c1, wino_sets(w1)[1]),) + (None,) * 8


class TrainPairBnConvFn(torch.autograd.Function):
    """(conv2(relu(bn1(conv1(x)))), downsample[0](x)) of a down-sampling BasicBlock in TRAIN mode inside the attack loop: the stride-2 pair
    kernel with the moments of its 3x3 output, then conv2 with the merge / normalisation / ReLU in its prologue - see TrainConvBnConvFn."""

    @staticmethod
    def forward(ctx, x, w3, wd, w2, gamma, beta, running_mean, running_var, momentum, eps):
        w10 = _dense_weight(w3, "s2p_f", wd)
        _dense_weight(w3, "s2p_b", wd)  # created outside any capture; the backward only reads it
        y3, y1, stats, cnt = ops.conv3x3s2_pair_stats_fwd(x, w10, w3.shape[0])
        c2, sm, si = ops.wino3x3_bn_train_pre(y3, stats, cnt, gamma, beta, eps, momentum, running_mean, running_var, wino_sets(w2)[0])
        ctx.save_for_backward(y3, w3, wd, w2, gamma, beta, sm, si)
        ctx.eps = eps
        ctx.set_materialize_grads(False)
        return c2, y1

    @staticmethod
    def backward(ctx, dc2, dy1):
        y3, w3, wd, w2, gamma, beta, sm, si = ctx.saved_tensors
        if not ctx.needs_input_grad[0] or (dc2 is None and dy1 is None):
            return (None,) * 10
        if dc2 is None:
            d_y3 = torch.zeros_like(y3)
        else:
            d_a1 = ops.wino3x3(dc2.contiguous(), wino_sets(w2)[1])
            d_y3 = ops.bn_act_bwd(d_a1, None, y3, gamma, sm, si, None, None, ctx.eps, True, True, True, False, False, None, beta)[0]
        dy1 = torch.zeros_like(y3) if dy1 is None else dy1.contiguous()
        return (ops.conv3x3s2_pair_bwd_data(d_y3, dy1, _dense_weight(w3, "s2p_b", wd), w3.shape[1]),) + (None,) * 9


class Conv3x3S2SmallFn(torch.autograd.Function):
    """Conv2d(3x3, stride 2, padding 1, bias=False) from an 8x8 or a 4x4 map (ResNet-18 layer3.0 / layer4.0 conv1 at 64x64 inputs,
    resnet.py:26-31): forward and backward-data on ee_s2.hip (reduction split over the wavefronts, backward by parity classes); the
    rearranged filters follow the weight's version counter like the Winograd ones; weight gradient: ee_wrw.hip."""

    @staticmethod
    def forward(ctx, x, weight):
        w9 = _dense_weight(weight, "s2m_f")
        _dense_weight(weight, "s2m_b")  # created outside any capture; the backward only reads it
        ctx.save_for_backward(x, weight)
        return ops.conv3x3s2_small_fwd(x, w9, weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.conv3x3s2_small_bwd_data(dy, _dense_weight(weight, "s2m_b"), weight.shape[1]) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1] and not _INPUT_GRAD_ONLY:
            if not _STOCK_WRW and ops.wrw3x3s2_supported(x, dy):
                dw = ops.wrw3x3s2(x, dy)[0]
            else:
                dw = torch.ops.aten.convolution_backward(dy, x, weight, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return dx, dw


class Conv3x3S2PairFn(torch.autograd.Function):
    """(conv1(x), downsample[0](x)) of a down-sampling BasicBlock (resnet.py:50-59, :137-142) - Conv2d(3x3, stride 2, padding 1) and
    Conv2d(1x1, stride 2) of the same input - as ONE launch each way on ee_s2.hip: the 1x1 filter rides along as a tenth tap over the 3x3's
    centre plane, and the backward pass returns the block's input gradient already summed.  Weight gradients: one launch of ee_wrw.hip."""

    @staticmethod
    def forward(ctx, x, w3, w1):
        w10 = _dense_weight(w3, "s2p_f", w1)
        _dense_weight(w3, "s2p_b", w1)  # created outside any capture; the backward only reads it
        ctx.save_for_backward(x, w3, w1)
        return ops.conv3x3s2_pair_fwd(x, w10, w3.shape[0])

    @staticmethod
    def backward(ctx, dy3, dy1):
        x, w3, w1 = ctx.saved_tensors
        dy3 = torch.zeros_like(dy1) if dy3 is None else dy3.contiguous()
        dy1 = torch.zeros_like(dy3) if dy1 is None else dy1.contiguous()
        dx = ops.conv3x3s2_pair_bwd_data(dy3, dy1, _dense_weight(w3, "s2p_b", w1), w3.shape[1]) if ctx.needs_input_grad[0] else None
        dw3 = dw1 = None
        if not _INPUT_GRAD_ONLY and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            if not _STOCK_WRW and ops.wrw3x3s2_supported(x, dy3, dy1):  # both weight gradients in one launch (ee_wrw.hip)
                dw3, dw1 = ops.wrw3x3s2(x, dy3, dy1)
                return dx, (dw3 if ctx.needs_input_grad[1] else None), (dw1 if ctx.needs_input_grad[2] else None)
            if ctx.needs_input_grad[1]:
                dw3 = torch.ops.aten.convolution_backward(dy3, x, w3, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
            if ctx.needs_input_grad[2]:
                dw1 = torch.ops.aten.convolution_backward(dy1, x, w1, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return dx, dw3, dw1


class Conv3x3Map2Fn(torch.autograd.Function):
    """Conv2d(Cin, Cout, 3, stride 1, padding 1, bias=False) on a 2x2 map (resnet.py:31, layer4 at 64x64 inputs) as one GEMM
    each way; the weight gradient (once per training step): ee_wrw.hip."""

    @staticmethod
    def forward(ctx, x, weight):
        B = x.shape[0]
        w2 = _dense_weight(weight)
        ctx.save_for_backward(x, weight)
        return torch.mm(x.reshape(B, -1), w2).view(B, weight.shape[0], 2, 2)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        B = dy.shape[0]
        dx = torch.mm(dy.reshape(B, -1), _dense_weight(weight).t()).view_as(x) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1] and not _INPUT_GRAD_ONLY:
            dw = conv3x3_weight_grad(x, dy, weight)
        return dx, dw


class Net2ConvFn(torch.autograd.Function):
    """relu(max_pool2d(conv2_drop(conv2(relu(max_pool2d(conv1(x), 2)))), 2)) of Net_2 (MNIST/models_mnist/Net2.py:13-14) in two launches
    each way (ee_net2.hip): the gradient w.r.t. the image - what the attack loop asks for, 40 times per training step - and, for a backward
    that needs PARAMETER gradients (the update, once per step), two more launches (net2_conv*_wrw_kernel; before round 3 that pass re-ran the
    stock sequence through ATen)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, drop, keep=1.0, draw_state=None):
        """drop: the Bernoulli(keep) draw [B,64] of Dropout2d (0 / 1), or None; scaled by 1 / keep inside the kernels.  drop None with a
        draw_state: the kernel draws the mask on the device (no random launch from the host: the captured attack iteration)"""
        a2, saved, drop = ops.net2_conv_fwd(x, w1, b1, w2, b2, drop, keep, draw_state)
        ctx.save_for_backward(x, w1, b1, w2, b2, drop, a2, *saved)
        ctx.keep = keep
        return a2

    @staticmethod
    def backward(ctx, da2):
        x, w1, b1, w2, b2, drop, a2, a1, c1, c2 = ctx.saved_tensors
        need = ctx.needs_input_grad
        if any(need[1:5]) and not _INPUT_GRAD_ONLY and not _STOCK_WRW:
            # the training step's backward: the input gradient as in the attack loop (its scratch da1 kept), then the parameter gradients in
            # two launches of ee_net2.hip (images added in order: bit-reproducible)
            da2 = da2.contiguous()
            da1 = torch.empty_like(a1)
            dx = ops.net2_conv_bwd(da2, a2, (a1, c1, c2), w1, w2, drop, ctx.keep, da1_out=da1, need_dx=need[0])
            dw1, db1, dw2, db2 = ops.net2_conv_wrw(x, da2, a2, (a1, c1, c2), da1, drop, ctx.keep)
            return ((dx if need[0] else None), (dw1 if need[1] else None), (db1 if need[2] and b1 is not None else None), (dw2 if need[3] else None),
                    (db2 if need[4] and b2 is not None else None), None, None, None)
        if any(need[1:5]) and not _INPUT_GRAD_ONLY:  # EEADV_STOCK_WRW=1: recompute the stock sequence with the same dropout mask and let ATen differentiate it
            import torch.nn.functional as F
            with torch.enable_grad():
                xx = x.detach().requires_grad_(need[0])
                h = F.relu(F.max_pool2d(F.conv2d(xx, w1, b1), 2))
                h = F.conv2d(h, w2, b2)
                if drop is not None:
                    h = h * drop.div(ctx.keep).view(drop.shape[0], drop.shape[1], 1, 1)
                h = F.relu(F.max_pool2d(h, 2))
                wanted = [t for t, n in zip((xx, w1, b1, w2, b2), need[:5]) if n and t is not None]
                got = iter(torch.autograd.grad(h, wanted, da2))
            return tuple(next(got) if (n and t is not None) else None for t, n in zip((xx, w1, b1, w2, b2), need[:5])) + (None, None, None)
        dx = ops.net2_conv_bwd(da2.contiguous(), a2, (a1, c1, c2), w1, w2, drop, ctx.keep) if need[0] else None
        return dx, None, None, None, None, None, None, None


class PoolLinearFn(torch.autograd.Function):
    """fc(global_avgpool(feat).view(B, -1)) in one launch each way (resnet.py:157-160; ee_head.hip).  The weight and
    bias gradients (once per training step) are two BLAS calls on the saved pooled features."""

    @staticmethod
    def forward(ctx, feat, weight, bias):
        logits, pooled = ops.pool_linear_fwd(feat, weight, bias)
        ctx.save_for_backward(pooled, weight)
        ctx.feat_shape = tuple(feat.shape)
        ctx.has_bias = bias is not None
        return logits

    @staticmethod
    def backward(ctx, dl):
        pooled, weight = ctx.saved_tensors
        dl = dl.contiguous()
        dfeat = ops.pool_linear_bwd(dl, weight, ctx.feat_shape) if ctx.needs_input_grad[0] else None
        dw = dl.t().mm(pooled) if ctx.needs_input_grad[1] and not _INPUT_GRAD_ONLY else None
        db = dl.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] and not _INPUT_GRAD_ONLY else None
        return dfeat, dw, db


class _ScalarLossFn(torch.autograd.Function):
    """A scalar loss whose gradients were produced by the same kernel launch as its value."""

    @staticmethod
    def forward(ctx, loss, *pairs):
        # pairs = (input_0, grad_0, input_1, grad_1, ...): inputs keep the graph, grads are constants
        ctx.grads = pairs[1::2]
        return loss.clone()

    @staticmethod
    def backward(ctx, go):
        out = [None]
        for g in ctx.grads:
            out += [None if g is None else (g * go.to(g.dtype)), None]
        return tuple(out)


def _attach(loss, *pairs):
    if any(t is not None and t.requires_grad for t in pairs[0::2]):
        flat = []
        for t, g in zip(pairs[0::2], pairs[1::2]):
            flat += [t, g]
        return _ScalarLossFn.apply(loss, *flat)
    return loss


def cross_entropy(logits, labels, reduction="mean", smoothing=0.0):
    """F.cross_entropy / LabelSmoothLoss (attacks.py:23, :255, :89-99) with the gradient from the same kernel."""
    loss, d = ops.ce(logits.contiguous(), labels.contiguous(), reduction, smoothing, True, logits.requires_grad)
    return _attach(loss, logits, d)


def kl_div_batchmean(logits_q, logits_p):
    """nn.KLDivLoss('batchmean')(log_softmax(logits_q), softmax(logits_p)) (attacks.py:375, :412, :426);
    the gradient flows into BOTH arguments, as in Trades.loss where `logits` is not detached."""
    loss, dq, dp = ops.kl_batchmean(logits_q.contiguous(), logits_p.contiguous(), True, logits_q.requires_grad,
                                    logits_p.requires_grad)
    return _attach(loss, logits_q, dq, logits_p, dp)


def mse_loss(a, b):
    """F.mse_loss (attacks.py:269)."""
    loss, da = ops.mse(a.contiguous(), b.contiguous(), True, a.requires_grad or b.requires_grad)
    return _attach(loss, a, da, b, None if da is None else -da)


def soft_cross_entropy(logits, soft_targets, scale):
    """-sum(log_softmax(logits) * soft_targets) * scale with float64 targets -> float64 scalar
    (attacks.py:462-463; Tiny_ImageNet/experiments_tinyimagenet.py:292-293)."""
    loss, dz = ops.softce(logits.contiguous(), soft_targets.contiguous(), scale, True, logits.requires_grad)
    return _attach(loss, logits, None if dz is None else dz.to(torch.float32))
