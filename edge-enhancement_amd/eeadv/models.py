"""The classifiers the attack loop differentiates, restated for PyTorch-ROCm.

These are the WORKLOAD, not the product: convolutions / BatchNorm / pooling stay on MIOpen and hipBLASLt.
Parameter names and construction order follow the reference so that (a) the same seed gives the same
initial weights as the reference's plain models and (b) reference checkpoints load (EE checkpoints carry
extra `sobel.*` / `u2netp.*` keys of modules the reference constructs but never calls - SURVEY 2.1 #12 -
load those with strict=False).

  Net_2 / Net2_EE / Net2_EE_square : MNIST/models_mnist/Net2.py:6-20, Net2_EE.py:7-54, Net2_EE_square.py:7-68
  ResNet / ResNet_EE(_square)      : Tiny_ImageNet/models_tinyimagenet/resnet.py:26-162, resnet_EE.py:107-204,
                                     resnet_EE_square.py:108-219; ImageNet twins differ in pooling / classes
                                     (ImageNet/models_imagenet/resnet.py:103,114)
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from utils.core import (Add_Square, CannyFilter, CannyFilter_BPDA, CannyFilter_step125_1, HighFreqSuppress, ee_front_end,
                        get_gaussian_kernel)

from . import hfs as _hfs, ops, runtime, syncbn as _syncbn
from . import functional as _fn
from .functional import EvalBasicBlockFn, EvalDownBlockFn, TrainConvBnConvFn, TrainPairBnConvFn, Net2ConvFn, BnActFn, BnDualFn, BnReluPoolFn, Conv1x1Fn, Conv1x1S2Fn, Conv3x3Map2Fn, Conv3x3S2PairFn, Conv3x3S2SmallFn, Conv3x3WinoFn, MaxPool3s2Fn, PoolLinearFn, StemConvFn

_CANNY = {"CannyFilter": CannyFilter, "CannyFilter_step125_1": CannyFilter_step125_1, "CannyFilter_BPDA": CannyFilter_BPDA}


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d whose `num_batches_tracked += 1` is NOT launched per layer: stock PyTorch issues one int64 add
    kernel per BatchNorm per training forward (20 launches per ResNet-18 forward, ~5 us each on MI355X, 3 % of a
    PGD-10 step).  The owning model bumps all counters with ONE multi-tensor add after its forward (`_bump_bn_counters`);
    state_dict keys, running statistics (momentum is never None here) and numerics are unchanged."""

    def forward(self, input):
        if self.training and self.track_running_stats:
            return F.batch_norm(input, self.running_mean, self.running_var, self.weight, self.bias, True, self.momentum, self.eps)
        return F.batch_norm(input, self.running_mean, self.running_var, self.weight, self.bias, not self.track_running_stats,
                            0.0 if self.momentum is None else self.momentum, self.eps)


def bn_act(bn, x, residual=None, relu=True, fork=False):
    """[relu]( bn(x) [+ residual] ): ONE HIP launch each way (ee_bn.hip) for our BatchNorm2d on dense NCHW fp32 ROCm
    tensors; the stock three-op sequence for anything else (SyncBatchNorm after convert_sync_batchnorm, channels_last,
    CPU plumbing in the host tests).  fork=True: the fused path returns the output as a PAIR of tensors over one buffer, one per
    consumer (functional.BnActFn); the stock path returns the plain tensor - callers take both (`_pair`)."""
    if ("bn" not in _STOCK and type(bn) is BatchNorm2d and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and bn.affine
            and bn.track_running_stats and (residual is None or (residual.is_contiguous() and residual.dtype == torch.float32))):
        return BnActFn.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                             0.0 if bn.momentum is None else bn.momentum, bn.eps, bn.training, relu, fork and _FORK)
    if isinstance(bn, _syncbn.SyncBatchNorm2d):  # multi-rank ImageNet scripts: fused kernels around one collective each way (eeadv.syncbn)
        return _syncbn.sync_bn_act(bn, x, residual, relu)
    out = bn(x)
    if residual is not None:
        out = out + residual
    return F.relu(out) if relu else out


def _pair(x):
    """(for the main branch, for the identity branch) of a block input: the two tensors of a forked output, or the tensor twice"""
    return x if isinstance(x, tuple) else (x, x)


_FORK = True  # block outputs as a pair of tensors over one buffer (False: autograd adds the two consumers' gradients in a launch of its own)


# EEADV_STOCK_GLUE=bn,pool,head,conv,stem,dense,conv3 (any subset) routes that piece of the CNN body through the stock ATen / MIOpen ops instead of
# ee_bn.hip / ee_pool.hip / ee_head.hip / ee_conv.hip: an A/B switch for measurements, never needed for correctness.
_STOCK = frozenset(t for t in os.environ.get("EEADV_STOCK_GLUE", "").split(",") if t)


_CHAIN = os.environ.get("EEADV_CHAIN", "1") == "1"  # the front end of a PGD iteration as two launches (ee_chain.hip) instead of six


# ee_conv.hip's stride-2 1x1 shortcut kernel reads its operands straight from L2, one wavefront per 32x32 tile: built for the 16 / 8 / 4-wide
# maps of the 64x64 configs (8 - 12 us against MIOpen's 25).  On ImageNet-size maps it is 4x SLOWER than MIOpen (profiles/
# round3_c_resnet50_conv_probe.txt: 424 / 462 / 483 us against 102 / 91 / 91 on the three ResNet-50 shortcuts at batch 32), so those go to MIOpen.
# ... and so does ResNet-50's last shortcut (1024 -> 2048 channels on a 14x14 map: 480 us): the reduction is split over a workgroup's four wavefronts only
_CONV1X1S2_MAXW, _CONV1X1S2_MAXC = 16, 256


def _dense_f32(x):
    return x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()


def _route(conv, how):
    """remember which implementation the last forward of this convolution took (bench.py: config.fallback_layers)"""
    conv.__dict__["_ee_route"] = how
    return conv


# routes whose forward / backward-data run on a vendor library (MIOpen solvers, Tensile GEMMs) rather than on a kernel of csrc/
_VENDOR_ROUTES = ("miopen", "tensile")


def fallback_report(model):
    """{"count", "of", "layers"}: the convolutions of `model` whose last forward went to MIOpen / Tensile, by module name and route
    ("miopen": ATen's convolution both ways; "miopen+ee_wrw": only the weight gradient is ours; "tensile-dense2x2": the 3x3 on a 2x2 map
    as one BLAS product).  A convolution that never ran is not counted."""
    layers, total = [], 0
    for name, m in model.named_modules():
        if isinstance(m, nn.Conv2d) and "_ee_route" in m.__dict__:
            total += 1
            how = m.__dict__["_ee_route"]
            if how.startswith(_VENDOR_ROUTES):
                layers.append("%s:%s" % (name, how))
    return {"count": len(layers), "of": total, "layers": layers}


def stem_bn_pool(bn, pool, x, fork=False, conv_stats=None):
    """maxpool(relu(bn1(x))) of the ResNet stem (resnet.py:113-117): one fused pass each way when the shapes allow (ee_bn.hip, bn_pool_*),
    the two separate kernels - or the stock modules - otherwise."""
    if ("bn" not in _STOCK and "pool" not in _STOCK and "bnpool" not in _STOCK and type(bn) is BatchNorm2d and type(pool) is nn.MaxPool2d
            and _dense_f32(x) and bn.affine and bn.track_running_stats and pool.kernel_size == 3 and pool.stride == 2 and pool.padding == 1
            and pool.dilation == 1 and not pool.ceil_mode and not pool.return_indices and ops.bn_relu_pool_supported(x)):
        return BnReluPoolFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, 0.0 if bn.momentum is None else bn.momentum, bn.eps,
                                  bn.training, fork and _FORK, conv_stats)
    return stem_pool(pool, bn_act(bn, x))


def stem_pool(pool, x):
    """The stem's MaxPool2d(3, 2, 1) through ee_pool.hip (bit-identical to ATen's, one-byte argmax); anything else stock."""
    if ("pool" not in _STOCK and type(pool) is nn.MaxPool2d and _dense_f32(x) and pool.kernel_size == 3 and pool.stride == 2 and pool.padding == 1
            and pool.dilation == 1 and not pool.ceil_mode and not pool.return_indices):
        return MaxPool3s2Fn.apply(x)
    return pool(x)


def _head_is_fused(avgpool, fc, x):
    is_global = (isinstance(avgpool, nn.AdaptiveAvgPool2d) and avgpool.output_size in (1, (1, 1))) or (
        isinstance(avgpool, nn.AvgPool2d) and avgpool.kernel_size in (x.shape[2], (x.shape[2], x.shape[3])) and x.shape[2] == x.shape[3]
        and avgpool.padding == 0)
    return ("head" not in _STOCK and is_global and type(fc) is nn.Linear and _dense_f32(x) and x.shape[1] <= 4096 and fc.out_features <= 8192
            and fc.weight.is_cuda and fc.weight.dtype == torch.float32 and fc.weight.is_contiguous())


_HEAD_CE = os.environ.get("EEADV_HEAD_CE", "1") != "0"  # the attack loop's loss gradient inside the head's backward launch


def head(avgpool, fc, x):
    """x = avgpool(x); x = x.view(B, -1); x = fc(x) (resnet.py:157-160): ONE launch each way (ee_head.hip) when the pool
    is global (AdaptiveAvgPool2d(1), or AvgPool2d(7) on a 7x7 map) and the tensors are dense fp32 ROCm."""
    if _head_is_fused(avgpool, fc, x):
        return PoolLinearFn.apply(x, fc.weight, fc.bias)
    x = avgpool(x)
    return fc(x.view(x.size(0), -1))


def s2_pair(block, x):
    """(conv1(x), downsample[0](x)) of a down-sampling BasicBlock as one launch each way (functional.Conv3x3S2PairFn, ee_s2.hip), or None:
    both convolutions read the block's input, and the shortcut's 1x1 / stride 2 filter sees exactly the 3x3's centre tap."""
    ds, c3 = block.downsample, block.conv1
    if ("conv3" in _STOCK or "conv" in _STOCK or "s2small" in _STOCK or "s2pair" in _STOCK or ds is None or not isinstance(ds, nn.Sequential) or len(ds) != 2
            or type(ds[0]) is not nn.Conv2d or type(c3) is not nn.Conv2d or not _dense_f32(x) or x.shape[2] != x.shape[3] or x.shape[2] not in (4, 8, 16)):
        return None
    c1 = ds[0]
    if (c3.kernel_size != (3, 3) or c3.stride != (2, 2) or c3.padding != (1, 1) or c3.dilation != (1, 1) or c3.groups != 1 or c3.bias is not None
            or c3.padding_mode != "zeros" or c1.kernel_size != (1, 1) or c1.stride != (2, 2) or c1.padding != (0, 0) or c1.groups != 1 or c1.bias is not None
            or c1.in_channels != c3.in_channels or c1.out_channels != c3.out_channels or c3.in_channels % 32 or c3.out_channels % 32
            or not c3.weight.is_contiguous() or not c1.weight.is_contiguous()):
        return None
    _route(c3, "ee_s2.pair"), _route(c1, "ee_s2.pair")
    return Conv3x3S2PairFn.apply(x, c3.weight, c1.weight)


def block_tail(block, bn, out, x, fork, sc=None):
    """relu(bn(out) + shortcut(x)) - the last line of a residual block (resnet.py:54-59 / :105-110).  With a down-sampling shortcut whose
    1x1 convolution runs on ee_conv.hip, its BatchNorm and the block's last one are ONE launch each way (functional.BnDualFn).
    sc: the shortcut's convolution if it is already done (s2_pair)."""
    ds = block.downsample
    if sc is not None:
        b2 = ds[1]
        if ("bn" not in _STOCK and "bndual" not in _STOCK and type(b2) is BatchNorm2d and type(bn) is BatchNorm2d and _dense_f32(out) and bn.affine and b2.affine
                and bn.track_running_stats and b2.track_running_stats and bn.training == b2.training and ops.bn_dual_supported(out) and sc.shape == out.shape):
            return BnDualFn.apply(out, sc, bn.weight, bn.bias, bn.running_mean, bn.running_var, 0.0 if bn.momentum is None else bn.momentum, bn.eps,
                                  b2.weight, b2.bias, b2.running_mean, b2.running_var, 0.0 if b2.momentum is None else b2.momentum, b2.eps,
                                  bn.training, fork and _FORK)
        return bn_act(bn, out, bn_act(b2, sc, relu=False), fork=fork)
    if ("bn" not in _STOCK and "bndual" not in _STOCK and ds is not None and isinstance(ds, nn.Sequential) and len(ds) == 2 and type(ds[0]) is nn.Conv2d
            and type(ds[1]) is BatchNorm2d and type(bn) is BatchNorm2d and _dense_f32(x) and _dense_f32(out) and bn.affine and ds[1].affine
            and bn.track_running_stats and ds[1].track_running_stats and bn.training == ds[1].training and ops.bn_dual_supported(out)):
        cv = ds[0]
        if ("conv" not in _STOCK and cv.kernel_size == (1, 1) and cv.stride == (2, 2) and cv.padding == (0, 0) and cv.bias is None and cv.groups == 1
                and cv.in_channels % 2 == 0 and cv.out_channels % 2 == 0 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                and x.shape[3] <= _CONV1X1S2_MAXW and cv.in_channels <= _CONV1X1S2_MAXC and cv.weight.is_contiguous()):
            _route(cv, "ee_conv.1x1s2")
            sc = Conv1x1S2Fn.apply(x, cv.weight)
            if sc.shape == out.shape:
                b2 = ds[1]
                return BnDualFn.apply(out, sc, bn.weight, bn.bias, bn.running_mean, bn.running_var, 0.0 if bn.momentum is None else bn.momentum, bn.eps,
                                      b2.weight, b2.bias, b2.running_mean, b2.running_var, 0.0 if b2.momentum is None else b2.momentum, b2.eps,
                                      bn.training, fork and _FORK)
    return bn_act(bn, out, shortcut(block, x), fork=fork)


def shortcut(block, x):
    """identity, or downsample(x) = BatchNorm(Conv2d(1x1, stride 2)) (resnet.py:137-142) with the convolution on ee_conv.hip
    and the BatchNorm on ee_bn.hip when the shapes allow; the stock modules otherwise."""
    ds = block.downsample
    if ds is None:
        return x
    if ("conv" not in _STOCK and isinstance(ds, nn.Sequential) and len(ds) == 2 and type(ds[0]) is nn.Conv2d and _dense_f32(x)):
        cv = ds[0]
        if (cv.kernel_size == (1, 1) and cv.stride == (2, 2) and cv.padding == (0, 0) and cv.bias is None and cv.groups == 1
                and cv.in_channels % 2 == 0 and cv.out_channels % 2 == 0 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                and x.shape[3] <= _CONV1X1S2_MAXW and cv.in_channels <= _CONV1X1S2_MAXC and cv.weight.is_contiguous()):
            _route(cv, "ee_conv.1x1s2")
            return bn_act(ds[1], Conv1x1S2Fn.apply(x, cv.weight), relu=False)
    if isinstance(ds, nn.Sequential) and len(ds) and isinstance(ds[0], nn.Conv2d):
        _route(ds[0], "miopen")
    return ds(x)


def conv3(conv, x):
    """A block's 3x3 convolution.  stride 1 on 16x16 / 8x8 / 4x4 maps (ResNet-18 layers 1-3 at 64x64 inputs): Winograd F(2x2,3x3) on the matrix
    cores (ee_wino.hip); stride 1 on a 2x2 map (layer 4): one dense product (functional.Conv3x3Map2Fn); stride 2 from a 16 / 8 / 4-wide map: the
    split-reduction kernel of ee_s2.hip (backward by parity classes); every other shape - the ImageNet-size maps - is MIOpen's."""
    plain = (type(conv) is nn.Conv2d and _dense_f32(x) and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.dilation == (1, 1)
             and conv.groups == 1 and conv.bias is None and conv.padding_mode == "zeros" and x.shape[2] == x.shape[3] and conv.weight.is_contiguous())
    if not plain:
        return _route(conv, "miopen")(x)
    hw, by32 = x.shape[2], conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0
    if conv.stride == (1, 1):
        if hw == 2 and "dense" not in _STOCK:
            _route(conv, "tensile-dense2x2")
            return Conv3x3Map2Fn.apply(x, conv.weight)
        if hw in (4, 8, 16) and by32 and "conv3" not in _STOCK and "wino" not in _STOCK:
            _route(conv, "ee_wino")
            return Conv3x3WinoFn.apply(x, conv.weight)
    elif conv.stride == (2, 2) and hw in (4, 8, 16) and by32 and "conv3" not in _STOCK and "s2small" not in _STOCK:
        _route(conv, "ee_s2.small")
        return Conv3x3S2SmallFn.apply(x, conv.weight)
    return _route(conv, "miopen")(x)


def conv1(conv, x):
    """A bottleneck block's 1x1 / stride 1 convolution (resnet.py:75-100): ATen's forward and backward-data, ee_wrw.hip's weight gradient where it
    takes the shape (channels % 64 == 0, H * W % 4 == 0 - every 1x1 of ResNet-50 at 224x224 but the 7x7 maps of layer 4)."""
    if ("conv" not in _STOCK and "wrw1x1" not in _STOCK and type(conv) is nn.Conv2d and _dense_f32(x) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and conv.in_channels % 64 == 0
            and conv.out_channels % 64 == 0 and (x.shape[2] * x.shape[3]) % 4 == 0 and conv.weight.is_contiguous()):
        _route(conv, "miopen+ee_wrw")
        return Conv1x1Fn.apply(x, conv.weight)
    return _route(conv, "miopen")(x)


def stem_conv(conv, x, want_stats=False):
    """conv1 of the ResNets (resnet.py:112): forward on ee_conv.hip where the shape allows; when x needs a gradient (the attack loop),
    its backward-data too."""
    if ("stem" not in _STOCK and type(conv) is nn.Conv2d and _dense_f32(x) and conv.in_channels == 3
            and (x.requires_grad or ops.stem7x7s2_fwd_supported(x, conv.weight))
            and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.bias is None and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and conv.weight.is_contiguous()):
        _route(conv, "ee_conv.stem" if ops.stem7x7s2_fwd_supported(x, conv.weight) else "miopen(forward)+ee_conv.stem(backward-data)")
        return StemConvFn.apply(x, conv.weight, want_stats)
    _route(conv, "miopen")
    return (conv(x), None) if want_stats else conv(x)


class deferred_bn_counters:
    """Inside this context the per-forward `num_batches_tracked += 1` of the ResNets is only COUNTED; leaving it adds the total in one
    multi-tensor launch per model.  engine wraps the capture of a multi-iteration attack graph in it: one 5 us launch per replay instead
    of one per iteration (nothing reads the counters in between: momentum is never None in these models)."""
    active = None

    def __enter__(self):
        self._prev, deferred_bn_counters.active = deferred_bn_counters.active, {}
        return self

    def __exit__(self, *exc):
        pending, deferred_bn_counters.active = deferred_bn_counters.active, self._prev
        for model, n in pending.values():
            _bump_bn_counters(model, n)


def _bump_bn_counters(model, n=1):
    if not model.training and n == 1:
        return
    if deferred_bn_counters.active is not None and model.training:
        ent = deferred_bn_counters.active.setdefault(id(model), [model, 0])
        ent[1] += n
        return
    counters = getattr(model, "_bn_counters", None)
    if counters is None or (counters and counters[0].device != next(model.parameters()).device):
        counters = [m.num_batches_tracked for m in model.modules() if isinstance(m, BatchNorm2d) and m.num_batches_tracked is not None]
        model._bn_counters = counters
    if counters:
        torch._foreach_add_(counters, n)


class _EEFrontMixin:
    """x_in = clamp(hfs(x | add_square(x)) + w * canny(x), 0, 1): the first lines of every EE forward."""

    def _build_front(self, size, channels, r, w, with_gf, low, high, alpha, sigma, type_canny, square, epsilon, n_queries):
        self.w = w
        self.with_gf = with_gf
        self.hfs = HighFreqSuppress(size, size, r)
        if type_canny not in _CANNY:
            raise NotImplementedError
        self.canny = _CANNY[type_canny](sigma=sigma, use_cuda=True, alpha=alpha)
        self.add_square = Add_Square(channels=channels, size=size, epsilon=epsilon, n_queries=n_queries) if square else None
        self.low = low / 255
        self.high = high / 255
        g = torch.from_numpy(get_gaussian_kernel(3, 0., 1.)).unsqueeze(0).unsqueeze(0).type(torch.float)
        self.weight_gaussian = nn.Parameter(data=g, requires_grad=False)

    def front(self, x, draws=None):
        """Autograd-capable front end (used by every ordinary forward)."""
        if self.add_square is None:
            x_lp = self.hfs(x)
        else:
            op = self.hfs.operator(x.device) if x.is_cuda else None
            if op is not None and op.fused_square:  # Add_Square fused into the low-pass kernel's load / store
                x_lp = _hfs.square_hfs_apply(x, op, self.add_square.eps, self.add_square.prepare(x, draws))
            else:
                x_lp = self.hfs(self.add_square(x, draws))
        return ee_front_end(x, x_lp, self.canny, self.w, self.low, self.high, self.with_gf, self.weight_gaussian)

    # ---- the same front end without autograd, for the attack loop (eeadv.engine): two launches forward, two backward,
    # and the input gradient stays split into (low-pass part [B,C,H,W], edge part [B,1,H,W]) so that the PGD update
    # kernel can add them in registers --------------------------------------------------------------------------------
    def manual_ok(self, x):
        return x.is_cuda and isinstance(self.canny, CannyFilter_step125_1) and not self.with_gf

    # ---- ... and the whole thing in ONE launch each way (ee_chain.hip): draws -> Add_Square -> low-pass -> edge filter -> combine
    # forward; gate -> edge adjoint -> low-pass -> d Add_Square -> PGD update backward.  One workgroup per image; shapes of the
    # reference configs (ops.chain_supported), CannyFilter_step125_1, at most one Add_Square query.  EEADV_CHAIN=0 switches it off.
    def chain_ok(self, x):
        if not _CHAIN or not self.manual_ok(x) or x.dtype != torch.float32 or x.dim() != 4:
            return False
        if not ops.chain_supported(x.shape[1], x.shape[2], x.shape[3]):
            return False
        if self.add_square is not None and (self.add_square.n_queries != 1 or self.add_square.h != x.shape[2]):
            return False
        return self.hfs.operator(x.device).chain is not None

    def front_chain(self, x, draws=None):
        """x -> (x_in, ctx) through ee_chain_fwd_f32."""
        op = self.hfs.operator(x.device)
        sq = self.add_square
        if sq is None:
            x_in, gate, gx, gy, _ = ops.chain_fwd(x, op.chain, self.canny.edge_weights, float(self.canny.alpha), float(self.high), float(self.w))
        else:
            d = None if draws is None else sq.prepare(x, draws)
            x_in, gate, gx, gy, _ = ops.chain_fwd(x, op.chain, self.canny.edge_weights, float(self.canny.alpha), float(self.high), float(self.w),
                                                  True, float(sq.eps), sq.square_sizes(x.device)[0][0],
                                                  None if d is not None else runtime.draw_state(x.device), d)
        return x_in, (gate, gx, gy)

    def front_chain_update_(self, x, g_in, ctx, x0, step_size, eps, lo, hi, direction):
        """In place on x: backward of the front end + the PGD update through ee_chain_bwd_f32."""
        gate, gx, gy = ctx
        op = self.hfs.operator(x.device)
        ops.chain_bwd_(x, g_in, gate, gx, gy, x0, op.chain, self.canny.edge_weights, float(self.canny.alpha), float(self.high), float(self.w),
                       float(step_size), float(eps), float(lo), float(hi), direction)

    def front_manual(self, x, draws=None):
        op = self.hfs.operator(x.device)
        d = None
        if self.add_square is not None:
            d = self.add_square.prepare(x, draws)
            if op.fused_square:
                x_lp = op.forward_square(x, self.add_square.eps, d)
            else:
                x_lp = op.forward(ops.add_square_fwd(x, float(self.add_square.eps), d["stripe"], d["sq_sign"], d["sq_pos"], d["sq_size"]))
        else:
            x_lp = op.forward(x)
        # the forward keeps the Sobel responses (8 B per pixel): the backward then skips x, blur and Sobel (ee_edge.hip)
        x_in, gate, _, gx, gy = ops.frontend_fwd_save(x, x_lp, self.canny.edge_weights, float(self.canny.alpha), float(self.high), float(self.w))
        return x_in, (x, gate, d, gx, gy)

    def front_manual_backward(self, g_in, ctx):
        x, gate, d, gx, gy = ctx
        op = self.hfs.operator(x.device)
        g_hfs, g_edge = ops.frontend_bwd_saved(g_in, gate, gx, gy, self.canny.edge_weights, float(self.canny.alpha), float(self.high),
                                               float(self.w))
        if d is None:
            g_lp = op.adjoint(g_hfs)
        elif op.fused_square:
            g_lp = op.backward_square(g_hfs, x, self.add_square.eps, d)
        else:
            g_lp = ops.add_square_bwd(op.adjoint(g_hfs), x, float(self.add_square.eps), d["stripe"], d["sq_sign"], d["sq_pos"], d["sq_size"])
        return g_lp, g_edge


# ---- MNIST ---------------------------------------------------------------------------------------------------
class Net_2(nn.Module):
    def __init__(self):
        super(Net_2, self).__init__()
        self.conv1 = nn.Conv2d(1, 32, kernel_size=5)
        self.conv2 = nn.Conv2d(32, 64, kernel_size=5)
        self.conv2_drop = nn.Dropout2d()
        self.fc1 = nn.Linear(4 * 4 * 64, 1024)
        self.fc2 = nn.Linear(1024, 10)

    def body(self, x):
        return self.head_from_pre(self.body_pre(x))

    def head_from_pre(self, z1):
        return self.fc2(F.relu(z1))

    def head_grad(self, z1, labels, reduction):
        """d CrossEntropyLoss(head_from_pre(z1), labels) / d z1 in one launch, or None where the fused kernel does not apply"""
        if type(self.fc2) is nn.Linear and z1.is_cuda and z1.dtype == torch.float32 and self.fc2.weight.dtype == torch.float32 and ops.fc_ce_grad_supported(z1, self.fc2.weight):
            return ops.fc_ce_grad(z1.contiguous(), self.fc2.weight.detach().contiguous(), None if self.fc2.bias is None else self.fc2.bias.detach(), labels, reduction)
        return None

    def body_pre(self, x):
        """fc1's output, i.e. the body without `fc2(relu(.))`: engine fuses those two layers with the cross-entropy gradient
        (ops.fc_ce_grad) inside the attack loop"""
        if ("net2" not in _STOCK and _dense_f32(x) and type(self.conv1) is nn.Conv2d and type(self.conv2) is nn.Conv2d
                and ops.net2_conv_supported(x, self.conv1.weight, self.conv2.weight) and self.conv1.weight.is_contiguous()
                and self.conv2.weight.is_contiguous()):
            # both convolution + pool + ReLU halves as one launch each (ee_net2.hip); Dropout2d's per-(image, channel) mask is drawn here
            # with the call F.dropout2d makes (noise.bernoulli_(1 - p)), so the generator advances as in the stock sequence; its
            # noise.div_(1 - p) happens inside the kernels
            # ... except while a HIP graph is being captured (the attack loop: 40 forwards per training step): there the second kernel
            # draws the mask itself from the device-resident Philox state (a 5 us launch less per iteration; replays draw fresh masks)
            drop, keep, state = None, 1.0, None
            if self.training and self.conv2_drop.p > 0:
                keep = 1.0 - self.conv2_drop.p
                if torch.cuda.is_current_stream_capturing():
                    state = runtime.draw_state(x.device)  # (exists: whoever captures creates it first - engine / trainer)
                else:
                    drop = torch.empty((x.shape[0], 64), dtype=x.dtype, device=x.device).bernoulli_(keep)
            _route(self.conv1, "ee_net2"), _route(self.conv2, "ee_net2")
            x = Net2ConvFn.apply(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias, drop, keep, state)
        else:
            _route(self.conv1, "miopen"), _route(self.conv2, "miopen")
            x = F.relu(F.max_pool2d(self.conv1(x), 2))
            x = F.relu(F.max_pool2d(self.conv2_drop(self.conv2(x)), 2))
        x = x.view(-1, 4 * 4 * 64)
        return self.fc1(x)

    def forward(self, x):
        return self.body(x)


class Net2_EE(_EEFrontMixin, Net_2):
    def __init__(self, r=8, w=1, with_gf=False, low=60.0, high=120.0, alpha=0.0, sigma=1, type_canny='CannyFilter'):
        nn.Module.__init__(self)
        self._build_front(28, 1, r, w, with_gf, low, high, alpha, sigma, type_canny, False, 0.05, 1)
        self.conv1 = nn.Conv2d(1, 32, kernel_size=5)
        self.conv2 = nn.Conv2d(32, 64, kernel_size=5)
        self.conv2_drop = nn.Dropout2d()
        self.fc1 = nn.Linear(4 * 4 * 64, 1024)
        self.fc2 = nn.Linear(1024, 10)

    def forward(self, x, draws=None):
        return self.body(self.front(x, draws))


class Net2_EE_square(_EEFrontMixin, Net_2):
    def __init__(self, r=8, w=1, with_gf=False, low=60.0, high=120.0, alpha=0.0, sigma=1, type_canny='CannyFilter',
                 epsilon=0.05, n_queries=5000):
        nn.Module.__init__(self)
        self._build_front(28, 1, r, w, with_gf, low, high, alpha, sigma, type_canny, True, epsilon, n_queries)
        self.conv1 = nn.Conv2d(1, 32, kernel_size=5)
        self.conv2 = nn.Conv2d(32, 64, kernel_size=5)
        self.conv2_drop = nn.Dropout2d()
        self.fc1 = nn.Linear(4 * 4 * 64, 1024)
        self.fc2 = nn.Linear(1024, 10)

    def forward(self, x, draws=None):
        return self.body(self.front(x, draws))


# ---- ResNets ---------------------------------------------------------------------------------------------------
def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def _plain_bn(bn):
    return type(bn) is BatchNorm2d and not bn.training and bn.affine and bn.track_running_stats and bn.weight.dtype == torch.float32


def _plain_conv3(cv, stride):
    return (type(cv) is nn.Conv2d and cv.kernel_size == (3, 3) and cv.stride == (stride, stride) and cv.padding == (1, 1) and cv.dilation == (1, 1)
            and cv.groups == 1 and cv.bias is None and cv.padding_mode == "zeros" and cv.in_channels % 32 == 0 and cv.out_channels % 32 == 0
            and cv.weight.is_contiguous())


def _bn_args(bn):
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


def eval_block(block, xm, xs, fork):
    """A BasicBlock under model.eval() where only the input gradient can be asked for (the attack loop, or autograd off): BatchNorm with
    running statistics is folded into the convolution kernels - two launches each way instead of four (functional.EvalBasicBlockFn /
    EvalDownBlockFn; EEADV_STOCK_GLUE=evalfuse switches it off).  Returns None where it does not apply (train mode, SyncBatchNorm, maps
    other than 4 / 8 / 16 wide after the block's stride, channel counts the kernels do not take, parameter gradients wanted)."""
    if ("evalfuse" in _STOCK or "bn" in _STOCK or "conv3" in _STOCK or "wino" in _STOCK or block.training or not _fn.input_only_forward()
            or not _dense_f32(xm) or xm.dim() != 4 or xm.shape[2] != xm.shape[3] or not _plain_bn(block.bn1) or not _plain_bn(block.bn2)
            or not _plain_conv3(block.conv2, 1) or block.conv2.out_channels > 512):
        return None
    ds, hw = block.downsample, xm.shape[2]
    dense_ok = "dense" not in _STOCK and "densefuse" not in _STOCK and ops.dense2x2_supported(block.conv2.in_channels, block.conv2.out_channels)
    if ds is None:
        if (hw not in (4, 8, 16) and not (hw == 2 and dense_ok)) or not _plain_conv3(block.conv1, 1) or block.conv1.in_channels != block.conv2.out_channels:
            return None
        how = "ee_dense+bn" if hw == 2 else "ee_wino+bn"
        _route(block.conv1, how), _route(block.conv2, how)
        return EvalBasicBlockFn.apply(xm, xs, block.conv1.weight, block.conv2.weight, *_bn_args(block.bn1), *_bn_args(block.bn2), fork and _FORK)
    if ("s2pair" in _STOCK or "s2small" in _STOCK or (hw not in (8, 16) and not (hw == 4 and dense_ok)) or not _plain_conv3(block.conv1, 2) or not isinstance(ds, nn.Sequential) or len(ds) != 2
            or type(ds[0]) is not nn.Conv2d or not _plain_bn(ds[1])):
        return None
    c1 = ds[0]
    if (c1.kernel_size != (1, 1) or c1.stride != (2, 2) or c1.padding != (0, 0) or c1.groups != 1 or c1.bias is not None or c1.in_channels != block.conv1.in_channels
            or c1.out_channels != block.conv1.out_channels or not c1.weight.is_contiguous()):
        return None
    _route(block.conv1, "ee_s2.pair+bn"), _route(c1, "ee_s2.pair+bn"), _route(block.conv2, "ee_dense+bn" if hw == 4 else "ee_wino+bn")
    return EvalDownBlockFn.apply(xm, xs, block.conv1.weight, c1.weight, block.conv2.weight, *_bn_args(block.bn1), *_bn_args(ds[1]), *_bn_args(block.bn2),
                                 fork and _FORK)


# Widths of conv2's map (the consumer) that take the boundary path.  The consumer's prologue merges S partials for EVERY reduction channel in EVERY workgroup:
# on 16x16 maps that is 6.4 K pairs against 1.6 M activations (layer1: the BatchNorm launch costs 10.3 us, the exchange 5.2 us,
# profiles/round4_a_bn_boundary_probe.txt); on the 8x8 / 4x4 maps of layers 2-3 it is 12.8 K / 25.6 K pairs per workgroup - more traffic than the
# 5 us BatchNorm launch it would replace (all six blocks on the path: tiny_ee_at 8452 -> 8069 img/s).  EEADV_TRAINFUSE_MAPS=16,8,4 overrides (A/B).
_TRAINFUSE_MAPS = frozenset(int(t) for t in os.environ.get("EEADV_TRAINFUSE_MAPS", "16").split(",") if t)


def train_mid_bn(block, xm):
    """conv2(relu(bn1(conv1(x)))) of a BasicBlock in TRAIN mode where only the input gradient can be asked for (the attack loop of the
    training drivers runs in train mode): bn1's batch statistics cross the kernel boundary between the two convolutions instead of a
    BatchNorm launch (functional.TrainConvBnConvFn / TrainPairBnConvFn; EEADV_STOCK_GLUE=trainfuse switches it off).  Returns conv2's raw
    output (and the shortcut convolution's for a down-sampling block), or None where it does not apply."""
    bn = block.bn1
    if ("trainfuse" in _STOCK or "bn" in _STOCK or "conv3" in _STOCK or "wino" in _STOCK or not block.training or not _fn.attack_forward_active()
            or not _dense_f32(xm) or xm.dim() != 4 or xm.shape[2] != xm.shape[3] or type(bn) is not BatchNorm2d or not bn.training or not bn.affine
            or not bn.track_running_stats or bn.weight.dtype != torch.float32 or not _plain_conv3(block.conv2, 1) or block.conv2.in_channels > 256):
        return None
    ds, hw = block.downsample, xm.shape[2]
    mom = 0.0 if bn.momentum is None else bn.momentum
    if (hw if ds is None else hw // 2) not in _TRAINFUSE_MAPS:  # the CONSUMER's map: conv2 runs behind the block's stride
        return None
    if ds is None:
        if hw not in (4, 8, 16) or not _plain_conv3(block.conv1, 1):
            return None
        _route(block.conv1, "ee_wino+stats"), _route(block.conv2, "ee_wino+bn(train)")
        return TrainConvBnConvFn.apply(xm, block.conv1.weight, block.conv2.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps), None
    if ("s2pair" in _STOCK or "s2small" in _STOCK or hw not in (8, 16) or not _plain_conv3(block.conv1, 2) or not isinstance(ds, nn.Sequential) or len(ds) != 2
            or type(ds[0]) is not nn.Conv2d):
        return None
    c1 = ds[0]
    if (c1.kernel_size != (1, 1) or c1.stride != (2, 2) or c1.padding != (0, 0) or c1.groups != 1 or c1.bias is not None or c1.in_channels != block.conv1.in_channels
            or c1.out_channels != block.conv1.out_channels or not c1.weight.is_contiguous()):
        return None
    _route(block.conv1, "ee_s2.pair+stats"), _route(c1, "ee_s2.pair+stats"), _route(block.conv2, "ee_wino+bn(train)")
    return TrainPairBnConvFn.apply(xm, block.conv1.weight, c1.weight, block.conv2.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, fork=False):
        """x: a tensor, or the two tensors of the previous block's forked output; fork: hand this block's output on the same way"""
        xm, xs = _pair(x)
        if not self.training:
            out = eval_block(self, xm, xs, fork)
            if out is not None:
                return out
        if self.training:
            mid = train_mid_bn(self, xm)
            if mid is not None:
                return block_tail(self, self.bn2, mid[0], xs, fork, sc=mid[1])
        both = s2_pair(self, xm) if self.downsample is not None else None
        if both is not None:  # conv1 and the shortcut's convolution in one launch; the identity piece xs gets no gradient of its own
            return block_tail(self, self.bn2, conv3(self.conv2, bn_act(self.bn1, both[0])), xs, fork, sc=both[1])
        out = bn_act(self.bn1, conv3(self.conv1, xm))
        return block_tail(self, self.bn2, conv3(self.conv2, out), xs, fork)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, fork=False):
        xm, xs = _pair(x)
        out = bn_act(self.bn1, conv1(self.conv1, xm))
        out = bn_act(self.bn2, conv3(self.conv2, out))
        return block_tail(self, self.bn3, conv1(self.conv3, out), xs, fork)


class ResNet(nn.Module):
    """dataset='tiny': 200 classes + AdaptiveAvgPool2d(1); dataset='imagenet': 1000 classes + AvgPool2d(7)."""

    def __init__(self, block, layers, num_classes=200, dataset="tiny"):
        super(ResNet, self).__init__()
        self._build_cnn(block, layers, num_classes, dataset)
        self._init_weights()

    def _build_cnn(self, block, layers, num_classes, dataset):
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AvgPool2d(7, stride=1) if dataset == "imagenet" else nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def body_pre(self, x):
        """the classifier up to the last block's output (engine fuses the head with the cross-entropy gradient inside the attack loop)"""
        # every block output but the last feeds two consumers (the next block's convolution and its identity branch): forked outputs
        x, moments = stem_conv(self.conv1, x, want_stats=True)  # the convolution's epilogue collects bn1's batch statistics
        x = stem_bn_pool(self.bn1, self.maxpool, x, fork=True, conv_stats=moments)
        blocks = [blk for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for blk in layer]
        for i, blk in enumerate(blocks):
            x = blk(x, fork=i + 1 < len(blocks))
        _bump_bn_counters(self)
        return x

    def body(self, x):
        return self.head_from_pre(self.body_pre(x))

    # ---- the same forward cut into three pieces at layer boundaries, for the data-parallel update (trainer._GraphedUpdate with a
    # ddp.FlatGradSync): the backward of each piece is a captured graph of its own, and the all-reduce of a piece's gradients is issued
    # while the next piece's backward runs.  A piece maps the previous piece's output (a tensor, or the pair of a forked block output)
    # to its own; together they are exactly forward().
    def segment_fns(self):
        def first(x):
            if hasattr(self, "front"):
                x = self.front(x)
            x, moments = stem_conv(self.conv1, x, want_stats=True)
            x = stem_bn_pool(self.bn1, self.maxpool, x, fork=True, conv_stats=moments)
            for blk in list(self.layer1) + list(self.layer2):
                x = blk(x, fork=True)
            return x

        def middle(x):
            for blk in self.layer3:
                x = blk(x, fork=True)
            return x

        def last(x):
            blocks = list(self.layer4)
            for i, blk in enumerate(blocks):
                x = blk(x, fork=i + 1 < len(blocks))
            _bump_bn_counters(self)
            return self.head_from_pre(x)
        return [first, middle, last]

    def grad_segments(self):
        """Parameter lists in the order their gradients become complete during backward (ddp.FlatGradSync lays its buffer out so)."""
        own = lambda *mods: [p for m in mods for p in m.parameters()]
        return [own(self.layer4, self.fc), own(self.layer3), own(self.conv1, self.bn1, self.layer1, self.layer2)]

    def head_from_pre(self, feat):
        return head(self.avgpool, self.fc, feat)

    def head_grad(self, feat, labels, reduction):
        """d CrossEntropyLoss(head(feat), labels) / d feat in TWO launches instead of three: ee_head.hip's forward (outside autograd), then the
        loss gradient formed inside the head's backward launch (ops.ce_pool_linear_bwd: the same bits).  (Pool + fc + cross-entropy + both
        backward steps as ONE launch - one workgroup per image, 100 workgroups - was built in round 2 and measured 1.5 % SLOWER end to end.)
        None where ee_head.hip does not take the head: engine falls back to head_from_pre."""
        if not _HEAD_CE or not _head_is_fused(self.avgpool, self.fc, feat):
            return None
        feat = feat.contiguous()
        logits, _ = ops.pool_linear_fwd(feat, self.fc.weight.detach(), None if self.fc.bias is None else self.fc.bias.detach())
        return ops.ce_pool_linear_bwd(logits, labels, self.fc.weight.detach(), tuple(feat.shape), reduction)

    def forward(self, x):
        return self.body(x)


class ResNet_EE(_EEFrontMixin, ResNet):
    def __init__(self, block, layers, num_classes=200, cize=224, r=16, w=0.5, with_gf=False, low=60.0, high=120.0, alpha=0.0,
                 sigma=1, type_canny='CannyFilter', dataset="tiny", square=False, epsilon=0.05, n_queries=5000):
        nn.Module.__init__(self)
        self._build_front(cize, 3, r, w, with_gf, low, high, alpha, sigma, type_canny, square, epsilon, n_queries)
        self._build_cnn(block, layers, num_classes, dataset)
        self._init_weights()

    def forward(self, x, draws=None):
        return self.body(self.front(x, draws))


_LAYERS = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
           101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


def make_resnet(depth, dataset="tiny", pretrained=False, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained weights are read from ~/.torch/models in the reference; none ship offline")
    block, layers = _LAYERS[depth]
    kwargs.setdefault("num_classes", 1000 if dataset == "imagenet" else 200)
    return ResNet(block, layers, dataset=dataset, **kwargs)


def make_resnet_ee(depth, dataset="tiny", square=False, pretrained=False, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained weights are read from ~/.torch/models in the reference; none ship offline")
    block, layers = _LAYERS[depth]
    kwargs.setdefault("num_classes", 1000 if dataset == "imagenet" else 200)
    return ResNet_EE(block, layers, dataset=dataset, square=square, **kwargs)
