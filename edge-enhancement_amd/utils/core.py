#!/usr/bin/env python
# -*- encoding: utf-8 -*-
"""Drop-in for the reference's utils/core.py (edge-enhancement modules): same class names, constructor
arguments, forward signatures and constructor prints; the primary filter (CannyFilter_step125_1), the fused
front end and Add_Square run as HIP kernels.

Differences from the reference: `use_cuda` no longer pins tensors to 'cuda:0' - modules follow
their input's device; fixed weights are non-persistent buffers unless the reference registered them
(`CannyFilter`) so checkpoints keep the reference's key sets; torch.rfft is gone, HighFreqSuppress applies
the same linear operator as two real contractions (eeadv/hfs.py).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from eeadv import _native  # noqa: F401  (fails loudly at import when libeeadv.so is missing)
from eeadv import functional as EF, hfs as _hfs, ops, runtime


# function to suppress high freqency components (core.py:15-55)
class HighFreqSuppress(torch.nn.Module):
    def __init__(self, w, h, r):
        super(HighFreqSuppress, self).__init__()
        self.w = w
        self.h = h
        self.r = r
        self.templete()
        self._ops = {}

    def templete(self):
        """core.py:23-42: ones on frequencies -r .. r-1 (rolled to FFT order); kept for interface parity."""
        temp = np.zeros((self.w, self.h), "float32")
        cw = self.w // 2
        ch = self.h // 2
        dw = self.r if self.w % 2 == 0 else self.r + 1
        dh = self.r if self.h % 2 == 0 else self.r + 1
        temp[cw - self.r:cw + dw, ch - self.r:ch + dh] = 1.0
        temp = np.roll(temp, -cw, axis=0)
        temp = np.roll(temp, -ch, axis=1)
        self.temp = torch.tensor(temp).unsqueeze(0).unsqueeze(0).unsqueeze(-1)

    def operator(self, device):
        key = str(device)
        if key not in self._ops:
            self._ops[key] = _hfs.HFSOperator(self.w, self.h, self.r, device)
        return self._ops[key]

    def forward(self, x):
        runtime.require_device(x, "HighFreqSuppress")
        return _hfs.hfs_apply(x.contiguous(), self.operator(x.device))

    def extra_repr(self):
        return 'feature_width={}, feature_height={}, radius={}'.format(self.w, self.h, self.r)


get_gaussian_kernel = ops.gaussian_kernel_np  # core.py:58-72
get_sobel_kernel = ops.sobel_kernel_np  # core.py:75-84

# k*45 degrees -> (row, col) of the -1 neighbour (centre +1).  core.py:87-112 builds these with
# cv2.getRotationMatrix2D / warpAffine; cv2 is not available, the table is DERIVED (parity unpinned).
_THIN_TABLE = {0: (1, 2), 1: (0, 2), 2: (0, 1), 3: (0, 0), 4: (1, 0), 5: (2, 0), 6: (2, 1), 7: (2, 2)}


def get_thin_kernels(start=0, end=360, step=45):
    thin_kernels = []
    for angle in range(start, end, step):
        k = np.zeros((3, 3))
        k[1, 1] = 1
        r, c = _THIN_TABLE[(angle // 45) % 8]
        k[r, c] = -1
        thin_kernels.append(k)
    return thin_kernels


def safeSign(tensor):  # core.py:115-118
    result = torch.sign(tensor)
    result[result == 0] = -1
    return result


class BinaryConnectDeterministic(torch.autograd.Function):
    """core.py:121-145: sign with sign(0) = -1; straight-through where |input| <= 1.001."""

    @staticmethod
    def forward(ctx, input):
        ctx.save_for_backward(input)
        return safeSign(input)

    @staticmethod
    def backward(ctx, grad_output):
        input, = ctx.saved_tensors
        grad_input = grad_output.clone()
        grad_input[torch.abs(input) > 1.001] = 0
        return grad_input


class To_compare(torch.autograd.Function):
    """core.py:329-358: 1 where input > threshold else 0; gradient passes where threshold < input <= 1.001."""

    @staticmethod
    def forward(ctx, input, threshold):
        ctx.save_for_backward(input, threshold)
        output = input.clone()
        output[output <= threshold] = 0
        output[output > threshold] = 1
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input, threshold = ctx.saved_tensors
        grad_input = grad_output.clone()
        grad_input[input <= threshold] = 0
        grad_input[input > 1.001] = 0
        return grad_input, None


class To_eq(torch.autograd.Function):
    """core.py:361-382."""

    @staticmethod
    def forward(ctx, input):
        ctx.save_for_backward(input)
        output = input.clone()
        output[input != 0.5] = 0
        output[input == 0.5] = 1
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input, = ctx.saved_tensors
        grad_input = grad_output.clone()
        grad_input[input != 0.5] = 0
        return grad_input


class _CannyWeights(nn.Module):
    """Fixed weights shared by the three Canny classes (core.py:161-214 / :399-424 / :522-547)."""

    def __init__(self, k_gaussian, mu, sigma, k_sobel, registered):
        super().__init__()
        if k_gaussian != 3 or k_sobel != 3:
            raise NotImplementedError("the HIP stencils are written for 3x3 Gaussian / Sobel kernels (all reference configs)")
        self.pad_gaussian = nn.ReplicationPad2d(k_gaussian // 2)
        self.reflect_pad = nn.ReplicationPad2d(k_sobel // 2)
        sobel_2D = get_sobel_kernel(k_sobel)
        tensors = {
            "weight_gaussian": torch.from_numpy(get_gaussian_kernel(k_gaussian, mu, sigma)).unsqueeze(0).unsqueeze(0).type(torch.float),
            "weight_sobel_x": torch.from_numpy(sobel_2D).unsqueeze(0).unsqueeze(0).type(torch.float),
            "weight_sobel_y": torch.from_numpy(sobel_2D.T.copy()).unsqueeze(0).unsqueeze(0).type(torch.float),
            "weight_directional": torch.from_numpy(np.stack(get_thin_kernels())).unsqueeze(1).type(torch.float),
            "weight_hysteresis": torch.from_numpy(np.ones((3, 3)) + 0.25).unsqueeze(0).unsqueeze(0).type(torch.float),
        }
        for name, t in tensors.items():
            if registered:  # appears in state_dict() as canny.weight_* like the reference's nn.Parameter
                self.register_parameter(name, nn.Parameter(data=t, requires_grad=False))
            else:  # reference: nn.Parameter(...).to('cuda') -> plain tensor, absent from state_dict()
                self.register_buffer(name, t, persistent=False)
        self.padding_directional = 1
        self.edge_weights = ops.EdgeWeights(sigma, mu)

    def _grads(self, img):
        """Blur per channel, Sobel summed over channels, /C, magnitude (core.py:233-257) in torch ops."""
        C = img.shape[1]
        blurred = torch.cat([F.conv2d(self.pad_gaussian(img[:, c:c + 1]), self.weight_gaussian) for c in range(C)], 1)
        pad_blurred = self.reflect_pad(blurred)
        grad_x = F.conv2d(pad_blurred, self.weight_sobel_x.repeat(1, C, 1, 1)) / C
        grad_y = F.conv2d(pad_blurred, self.weight_sobel_y.repeat(1, C, 1, 1)) / C
        return grad_x, grad_y, (grad_x ** 2 + grad_y ** 2) ** 0.5


def _nms(mag, gx, gy, weight_directional, assign):
    """core.py:258-290 / :448-480."""
    ori = torch.atan(gy / gx) * (360 / np.pi) + 180
    ori = torch.round(ori / 45) * 45
    directional = F.conv2d(mag, weight_directional, padding=1)
    positive_idx = (ori / 45) % 8
    thin_edges = mag.clone()
    for pos_i in range(4):
        neg_i = pos_i + 4
        is_oriented = (positive_idx == pos_i) * 1 + (positive_idx == neg_i) * 1
        is_max = (torch.stack([directional[:, pos_i], directional[:, neg_i]]).min(dim=0)[0] > 0.0).unsqueeze(1)
        to_remove = (is_max == 0) * 1 * is_oriented > 0
        if assign:
            thin_edges[to_remove] = 0.0
        else:
            thin_edges = torch.mul(thin_edges, ~to_remove)
    return thin_edges


class CannyFilter(_CannyWeights):
    """core.py:148-326: full Canny (NMS, STE double threshold, hysteresis).  The path every model takes (both thresholds
    given, hysteresis=True) is ONE HIP kernel each way on ROCm tensors; other argument combinations and the opt-in CPU
    plumbing path run the same expressions in torch ops.  Thin kernels derived (cv2 unavailable) -> parity unpinned."""

    def __init__(self, k_gaussian=3, mu=0, sigma=1, k_sobel=3, use_cuda=False, alpha=0.0):
        super(CannyFilter, self).__init__(k_gaussian, mu, sigma, k_sobel, registered=True)
        self.device = 'cuda' if use_cuda else 'cpu'
        self.alpha = alpha
        print('CannyFilter; sigma:{}, alpha:{}'.format(sigma, alpha))

    def forward(self, img, low_threshold=None, high_threshold=None, hysteresis=False):
        if runtime.require_device(img, "CannyFilter") and low_threshold is not None and high_threshold is not None and hysteresis:
            return EF.CannyFn.apply(img, self.edge_weights, float(self.alpha), float(low_threshold), float(high_threshold))
        grad_x, grad_y, mag = self._grads(img)
        mag = torch.where(mag < self.alpha, torch.zeros_like(mag), mag)
        thin_edges = _nms(mag, grad_x, grad_y, self.weight_directional, assign=True)
        if low_threshold is not None:
            sign = BinaryConnectDeterministic.apply
            low = (sign(thin_edges - low_threshold) + 1) / 2
            if high_threshold is not None:
                high = (sign(thin_edges - high_threshold) + 1) / 2
                thin_edges = low * 0.5 + high * 0.5
                if hysteresis:
                    weak = (thin_edges == 0.5) * 1
                    weak_is_high = (F.conv2d(thin_edges, self.weight_hysteresis, padding=1) > 1) * weak
                    thin_edges = high * 1 + weak_is_high * 1
            else:
                thin_edges = low * 1
        return thin_edges


class CannyFilter_BPDA(_CannyWeights):
    """core.py:386-505 (AWP configs only): no alpha mask, NMS by multiplication, thresholds through To_compare, hysteresis
    through To_eq.  The path the models take (both thresholds given, hysteresis=True) runs on the HIP kernels of
    ee_canny.hip (one launch forward, two backward); other argument combinations and the opt-in CPU plumbing path run the
    same expressions in torch ops.  Thin kernels derived (cv2 unavailable) -> parity unpinned."""

    def __init__(self, k_gaussian=3, mu=0, sigma=1, k_sobel=3, use_cuda=False, alpha=0.0):
        super(CannyFilter_BPDA, self).__init__(k_gaussian, mu, sigma, k_sobel, registered=not use_cuda)
        self.device = 'cuda' if use_cuda else 'cpu'
        self.alpha = torch.tensor(alpha)
        print('CannyFilter; sigma:{}, alpha:{}'.format(sigma, alpha))

    def forward(self, img, low_threshold=None, high_threshold=None, hysteresis=False):
        if runtime.require_device(img, "CannyFilter_BPDA") and low_threshold is not None and high_threshold is not None and hysteresis:
            return EF.CannyBPDAFn.apply(img, self.edge_weights, float(low_threshold), float(high_threshold))
        grad_x, grad_y, mag = self._grads(img)
        thin_edges = _nms(mag, grad_x, grad_y, self.weight_directional, assign=False)
        if low_threshold is not None:
            dev = img.device
            low = To_compare.apply(thin_edges, torch.tensor(low_threshold, device=dev))
            if high_threshold is not None:
                high = To_compare.apply(thin_edges, torch.tensor(high_threshold, device=dev))
                thin_edges = low * 0.5 + high * 0.5
                if hysteresis:
                    weak = To_eq.apply(thin_edges)
                    weak_0 = F.conv2d(thin_edges, self.weight_hysteresis, padding=1)
                    weak_1 = To_compare.apply(weak_0, torch.tensor(1., device=dev))
                    thin_edges = high * 1 + weak_1 * weak * 1
        return thin_edges


#### BPDA: min (max (edge - high_threshold, 0), 1)
class CannyFilter_step125_1(_CannyWeights):
    """core.py:509-585: Gaussian -> Sobel -> magnitude -> alpha mask -> 1[mag > high].  `low_threshold` and
    `hysteresis` are accepted and ignored, exactly as in the reference (:578-583).  ONE HIP kernel each way."""

    def __init__(self, k_gaussian=3, mu=0, sigma=1, k_sobel=3, use_cuda=False, alpha=0.0):
        super(CannyFilter_step125_1, self).__init__(k_gaussian, mu, sigma, k_sobel, registered=not use_cuda)
        self.device = 'cuda' if use_cuda else 'cpu'
        self.alpha = torch.tensor(alpha)
        print('CannyFilter; sigma:{}, alpha:{}'.format(sigma, alpha))

    def forward(self, img, low_threshold=None, high_threshold=None, hysteresis=False):
        if high_threshold is None:
            raise NameError("high_threshold is required (the reference fails with NameError at core.py:583)")
        if runtime.require_device(img, "CannyFilter_step125_1"):
            return EF.Edge125Fn.apply(img, self.edge_weights, float(self.alpha), float(high_threshold))
        _, _, mag = self._grads(img)
        mag = torch.where(mag < self.alpha, torch.zeros_like(mag), mag)
        return To_compare.apply(mag.clone(), torch.tensor(high_threshold)) * 1


def ee_front_end(x, x_hfs, canny, w, low, high, with_gf=False, weight_gaussian=None):
    """The combine lines of every EE model forward (Tiny_ImageNet/models_tinyimagenet/resnet_EE.py:182-191,
    MNIST/models_mnist/Net2_EE.py:40-49):  clamp(x_hfs + w * canny(x, low, high, hysteresis=True), 0, 1).
    Fused into one kernel each way for the primary filter; other filters compose."""
    if isinstance(canny, CannyFilter_step125_1) and not with_gf and x.is_cuda:
        return EF.FrontEndFn.apply(x, x_hfs, canny.edge_weights, float(canny.alpha), float(high), float(w))
    if isinstance(canny, CannyFilter) and not with_gf and x.is_cuda:
        return EF.CannyFrontEndFn.apply(x, x_hfs, canny.edge_weights, float(canny.alpha), float(low), float(high), float(w))
    x_canny = canny(x, low_threshold=low, high_threshold=high, hysteresis=True)
    if with_gf:
        x_canny = F.conv2d(x_canny.type(torch.float), weight_gaussian, padding=1)
    return torch.clamp(x_hfs + w * x_canny, 0.0, 1.0)


# add square to x (core.py:589-655)
class Add_Square(nn.Module):
    def __init__(self, channels=3, size=224, epsilon=0.05, p_init=0.8, n_queries=5000, rescale_schedule=False):
        super(Add_Square, self).__init__()
        self.c = channels
        self.h = size
        self.eps = epsilon
        self.p_init = p_init
        self.n_queries = n_queries
        self.rescale_schedule = rescale_schedule
        self._sizes = {}

    def random_choice(self, shape, device):
        t = 2 * torch.rand(shape, device=device) - 1
        return torch.sign(t)

    def random_int(self, low=0, high=1, shape=[1], device=None):
        t = low + (high - low) * torch.rand(shape, device=device)
        return t.long()

    def p_selection(self, it):
        """ schedule to decrease the parameter p (core.py:607-634)"""
        if self.rescale_schedule:
            it = int(it / self.n_queries * 10000)
        if 10 < it <= 50:
            p = self.p_init / 2
        elif 50 < it <= 200:
            p = self.p_init / 4
        elif 200 < it <= 500:
            p = self.p_init / 8
        elif 500 < it <= 1000:
            p = self.p_init / 16
        elif 1000 < it <= 2000:
            p = self.p_init / 32
        elif 2000 < it <= 4000:
            p = self.p_init / 64
        elif 4000 < it <= 6000:
            p = self.p_init / 128
        elif 6000 < it <= 8000:
            p = self.p_init / 256
        elif 8000 < it:
            p = self.p_init / 512
        else:
            p = self.p_init
        return p

    def square_sizes(self, device):
        """s of core.py:644 for every query: deterministic, built once per device."""
        key = str(device)
        if key not in self._sizes:
            n_features = self.c * self.h * self.h
            s = [max(int(round(math.sqrt(self.p_selection(i) * n_features / self.c))), 1) for i in range(self.n_queries)]
            self._sizes[key] = (s, torch.tensor(s, dtype=torch.int32, device=device))
        return self._sizes[key]

    def draw(self, batch, device):
        """The random numbers of one forward (core.py:637, :645, :648), all left on `device` - no host sync."""
        sizes, sizes_dev = self.square_sizes(device)
        if torch.device(device).type == "cuda":  # one launch instead of ~15 (ee_square.hip), graph-replay safe
            stripe, sq_pos, sq_sign = ops.square_draw(batch, self.c, self.h, sizes_dev, runtime.draw_state(torch.device(device)))
            return {"stripe": stripe, "sq_pos": sq_pos, "sq_sign": sq_sign, "sq_size": sizes_dev}
        stripe = self.random_choice([batch, self.c, 1, self.h], device)
        nq = len(sizes)
        u = torch.rand([nq], device=device)
        span = (self.h - sizes_dev.to(torch.float32))
        sq_pos = (0 + (span - 0) * u).long()
        sq_sign = self.random_choice([nq, self.c], device)
        return {"stripe": stripe, "sq_pos": sq_pos, "sq_sign": sq_sign, "sq_size": sizes_dev}

    def prepare(self, x, draws=None):
        """Draws in the layout the kernels take: stripe [B,C,1,W] fp32, sq_sign [nq,C] fp32, sq_pos [nq] int64,
        sq_size [nq] int32, all on x's device (fresh device-side draws when `draws` is None)."""
        d = self.draw(x.shape[0], x.device) if draws is None else draws
        return {"stripe": d["stripe"].to(x.device, torch.float32).contiguous(),
                "sq_sign": d["sq_sign"].reshape(-1, self.c).to(x.device, torch.float32).contiguous(),
                "sq_pos": d["sq_pos"].to(x.device, torch.int64).contiguous(),
                "sq_size": d["sq_size"].to(x.device) if "sq_size" in d else self.square_sizes(x.device)[1]}

    def forward(self, x, draws=None):
        runtime.require_device(x, "Add_Square")
        if x.is_cuda:
            d = self.prepare(x, draws)
            return EF.AddSquareFn.apply(x, float(self.eps), d["stripe"], d["sq_sign"], d["sq_pos"], d["sq_size"])
        d = self.draw(x.shape[0], x.device) if draws is None else draws
        x_best = torch.clamp(x + self.eps * d["stripe"], 0., 1.)
        for q, s in enumerate(self.square_sizes(x.device)[0]):
            vh = int(d["sq_pos"][q])
            new_deltas = torch.zeros([self.c, self.h, self.h], device=x.device)
            new_deltas[:, vh:vh + s, vh:vh + s] = 2. * self.eps * d["sq_sign"].reshape(-1, self.c)[q].view(self.c, 1, 1)
            x_best = x_best + new_deltas
            x_best = torch.min(torch.max(x_best, x - self.eps), x + self.eps)
            x_best = torch.clamp(x_best, 0., 1.)
        return x_best
