#!/usr/bin/env python
# -*- encoding: utf-8 -*-
"""Drop-in for the reference's utils/attacks.py: same names, argument order, defaults and side effects,
with the per-step element-wise work and the losses running as HIP kernels (libeeadv.so).

Reference lines are cited per function (paths relative to the reference root).  Differences, all
deliberate and documented in DESIGN.md:
  * `device='cuda'` strings hard-coded in the reference (attacks.py:250,291,311,383,406) follow the input
    tensor's device instead;
  * random starts are drawn on the device (Philox inside the init kernel) unless `noise=` is injected -
    device and host generators differ anyway, parity tests inject the noise;
  * CWLinfAttack(target=None) works (the reference raises TypeError at :152);
  * CPU tensors are refused unless eeadv.runtime.allow_cpu_plumbing(True) was called (--no-cuda drivers).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from eeadv import _native  # noqa: F401  (fails loudly at import when libeeadv.so is missing)
from eeadv import engine, functional as EF, ops, runtime

_INF = float("inf")


# ---------------------------------------------------------------------------------------------------------
# shared pieces
# ---------------------------------------------------------------------------------------------------------
def _uniform_start(x0, eps, noise=None):
    """attacks.py:15-17: clamp(x0 + U(-eps, eps), 0, 1)."""
    if runtime.require_device(x0, "random start"):
        x0c = x0.contiguous()
        if noise is not None:
            return ops.pgd_init(x0c, noise.to(x0c.device, torch.float32).contiguous(), 0.0, 1.0)
        seed, off = runtime.philox_ticket(x0c.device, x0c.numel())
        return ops.pgd_init_rng(x0c, float(eps), 0, seed, off, 0.0, 1.0)
    nz = torch.zeros_like(x0).uniform_(-eps, eps) if noise is None else noise
    return torch.clamp(x0 + nz, 0, 1)


def _randn_start(x0, noise=None, scale=0.001):
    """attacks.py:250 / :406: x_natural + 0.001 * randn (NOT clamped)."""
    if runtime.require_device(x0, "random start"):
        x0c = x0.contiguous()
        if noise is not None:
            nz = (scale * noise.to(x0c.device, torch.float32)).contiguous()
            return ops.pgd_init(x0c, nz, -_INF, _INF)
        seed, off = runtime.philox_ticket(x0c.device, x0c.numel())
        return ops.pgd_init_rng(x0c, scale, 1, seed, off, -_INF, _INF)
    nz = torch.randn(x0.shape, device=x0.device) if noise is None else noise
    return x0 + scale * nz


def _host_loop(model, x0, x, loss_fn, num_steps, step_size, eps, direction):
    """Plumbing path for CPU tensors (opt-in): the reference's own expressions, attacks.py:19-27."""
    for _ in range(num_steps):
        x.requires_grad_()
        with torch.enable_grad():
            loss = loss_fn(model(x))
        grad = torch.autograd.grad(loss, [x])[0]
        x = x.detach() + direction * step_size * torch.sign(grad.detach())
        x = torch.min(torch.max(x, x0 - eps), x0 + eps)
        x = torch.clamp(x, 0, 1)
    return x


def _loop(model, x0, x, spec, host_loss, num_steps, step_size, eps, direction=1):
    if runtime.require_device(x0, "PGD loop"):
        return engine.pgd_loop(model, x0, x, spec, num_steps, float(step_size), float(eps), direction)
    return _host_loop(model, x0, x, host_loss, num_steps, step_size, eps, direction)


def _random_targets(labels, nclass, device, label_offset=None):
    """attacks.py:38-40: target = fmod(y + randint(1, nclass), nclass)."""
    if label_offset is None:
        label_offset = torch.randint(low=1, high=nclass, size=labels.shape).to(device)
    return torch.fmod(labels + label_offset.to(labels.device), nclass)


# ---------------------------------------------------------------------------------------------------------
# Projected Gradient Descent (attacks.py:12-29)
# ---------------------------------------------------------------------------------------------------------
def PGD(model, args, inputs, targets, num_steps, step_size, noise=None):
    x0 = inputs.detach()
    x = _uniform_start(x0, args.epsilon, noise) if args.random else x0.clone()
    return _loop(model, x0, x, engine.LossSpec(engine.CE_SUM, targets),
                 lambda z: F.cross_entropy(z, targets, reduction='sum'), num_steps, step_size, args.epsilon)


# A targeted_PGD white-box attacker with random target label (attacks.py:33-56)
def targeted_PGD(model, args, inputs, labels, num_steps, step_size, nclass, device, noise=None, label_offset=None):
    x0 = inputs.detach()
    target_labels = _random_targets(labels, nclass, device, label_offset)
    x = _uniform_start(x0, args.epsilon, noise) if args.random else x0.clone()
    x = _loop(model, x0, x, engine.LossSpec(engine.CE_SUM, target_labels),
              lambda z: F.cross_entropy(z, target_labels, reduction='sum'), num_steps, step_size, args.epsilon, -1)
    return x, target_labels


def targeted_PGD_trick(model, args, inputs, labels, num_steps, step_size, nclass, device, noise=None, label_offset=None,
                       start_from_noise=None):
    """attacks.py:59-86: one Bernoulli per BATCH decides whether the random start is used (:69-71)."""
    x0 = inputs.detach()
    target_labels = _random_targets(labels, nclass, device, label_offset)
    x = x0.clone()
    if args.random:
        if start_from_noise is None:
            start_from_noise = bool(torch.gt(torch.rand([]), args.prob_start_from_clean))
        # x + 0 * noise then clamp == clamp(x): the clamp at :73 runs either way
        x = _uniform_start(x0, args.epsilon, noise) if start_from_noise else torch.clamp(x0, 0.0, 1.0)
    x = _loop(model, x0, x, engine.LossSpec(engine.CE_SUM, target_labels),
              lambda z: F.cross_entropy(z, target_labels, reduction='sum'), num_steps, step_size, args.epsilon, -1)
    return x, target_labels


class LabelSmoothLoss(torch.nn.Module):
    """attacks.py:89-99."""

    def __init__(self, smoothing=0.0):
        super(LabelSmoothLoss, self).__init__()
        self.smoothing = smoothing

    def forward(self, input, target):
        if runtime.require_device(input, "LabelSmoothLoss"):
            return EF.cross_entropy(input, target, "mean", float(self.smoothing))
        log_prob = F.log_softmax(input, dim=-1)
        weight = input.new_ones(input.size()) * self.smoothing / (input.size(-1) - 1.)
        weight.scatter_(-1, target.unsqueeze(-1), (1. - self.smoothing))
        return (-weight * log_prob).sum(dim=-1).mean()


# Compute loss for trick model (attacks.py:103-106)
def compute_loss_and_error(logits, label, label_smoothing=0.):
    return LabelSmoothLoss(label_smoothing)(logits, label.long())


# FGSM (attacks.py:110-128)
def FGSM(model, inputs, target, targeted=False, step_size=0.007):
    x = inputs.detach().clone()
    if runtime.require_device(x, "FGSM"):
        g = engine.input_gradient(engine._unwrap(model), x.contiguous(), engine.LossSpec(engine.CE_SUM, target))
        return ops.fgsm_step(x.detach().contiguous(), g.contiguous(), float(step_size), 0.0, 1.0, -1 if targeted else 1)
    x.requires_grad_()
    with torch.enable_grad():
        loss = F.cross_entropy(model(x), target, reduction='sum')
    grad = torch.autograd.grad(loss, [x])[0]
    x = x.detach() + (-step_size if targeted else step_size) * torch.sign(grad.detach())
    return torch.clamp(x, 0.0, 1.0)


def predict_from_logits(logits, dim=1):
    """attacks.py:131-132."""
    if logits.is_cuda and logits.dim() == 2 and dim in (1, -1) and logits.dtype == torch.float32:
        return ops.topk(logits.contiguous(), None, 1)[0][:, 0]
    return logits.max(dim=dim, keepdim=False)[1]


# CW with Linf norm (attacks.py:136-232)
def CWLinfAttack(x, y, model, magnitude, previous_p, max_eps, max_iters=20, target=None, _type='linf', n_class=10,
                 cur_device=None, noise=None):
    """Evaluation-only attack.  Margin loss -sum(relu(correct - wrong + 50)) (:195-206), fixed step 0.00392
    (:212), three projections per iteration (:218-222); only currently-correct samples are attacked (:146-151).
    The update/projection chain runs through the HIP PGD-step kernel twice (box `magnitude`, then box `max_eps`
    around x - previous_p), which is the same min/max/clamp sequence."""
    model.eval()
    device = cur_device if cur_device is not None else x.device
    x, y = x.to(device), y.to(device)
    if target is not None:
        target = target.to(device)
    adv = x.clone()
    with torch.no_grad():
        pred = predict_from_logits(model(x))
    if torch.sum((pred == y)).item() == 0:
        return adv, previous_p
    ind = (pred == y).nonzero().squeeze()
    x, y = x[ind], y[ind]
    x = x if len(x.shape) == 4 else x.unsqueeze(0)
    y = y if len(y.shape) == 1 else y.unsqueeze(0)
    if target is not None:
        target = target[ind]
        target = target if len(target.shape) == 1 else target.unsqueeze(0)
    previous_p_c = None
    if previous_p is not None:
        previous_p = previous_p.to(device)
        previous_p_c = previous_p.clone()
        previous_p = previous_p[ind]
        previous_p = previous_p if len(previous_p.shape) == 4 else previous_p.unsqueeze(0)
    one_hot_y = torch.zeros(y.size(0), n_class, device=device)
    one_hot_y[torch.arange(y.size(0)), y] = 1
    mag = magnitude.item() if isinstance(magnitude, torch.Tensor) else magnitude
    if noise is None:
        rand_perturb = torch.FloatTensor(x.shape).uniform_(-mag, mag).to(device)
    else:
        rand_perturb = noise.to(device)[ind].reshape(x.shape)
    x = x.contiguous()
    centre2 = (x - previous_p).contiguous() if previous_p is not None else x
    on_dev = runtime.require_device(x, "CWLinfAttack")
    adv_imgs = ops.pgd_init(x, rand_perturb.contiguous(), 0.0, 1.0) if on_dev else torch.clamp(x + rand_perturb, 0, 1)
    for _iter in range(int(max_iters)):
        adv_imgs.requires_grad_(True)
        with torch.enable_grad():
            outputs = model(adv_imgs)
            correct_logit = torch.sum(one_hot_y * outputs, dim=1)
            if target is not None:
                wrong = torch.zeros(target.size(0), n_class, device=device)
                wrong[torch.arange(target.size(0)), target] = 1
                wrong_logit = torch.sum(wrong * outputs, dim=1)
            else:
                wrong_logit, _ = torch.max((1 - one_hot_y) * outputs - 1e4 * one_hot_y, dim=1)
            loss = -torch.sum(F.relu(correct_logit - wrong_logit + 50))
        grads = torch.autograd.grad(loss, adv_imgs)[0]
        adv_imgs = adv_imgs.detach()
        if on_dev:
            # :213 + :218 + :220 : step, box `magnitude` around x, clamp  (max/min commute inside one box)
            ops.pgd_step_(adv_imgs, grads.contiguous(), x, 0.00392, float(mag), 0.0, 1.0, 1)
            # :222 : box `max_eps` around x - previous_p, no clamp inside the loop
            ops.pgd_step_(adv_imgs, torch.zeros_like(adv_imgs), centre2, 0.0, float(max_eps), -_INF, _INF, 1)
        else:
            adv_imgs = adv_imgs + 0.00392 * torch.sign(grads)
            adv_imgs = torch.max(torch.min(adv_imgs, x + mag), x - mag).clamp_(0, 1)
            adv_imgs = torch.max(torch.min(adv_imgs, centre2 + max_eps), centre2 - max_eps)
    adv_imgs = adv_imgs.clamp_(0, 1)
    now_p = adv_imgs - x
    adv[ind] = adv_imgs
    if previous_p is not None:
        previous_p_c[ind] = previous_p + now_p
        return adv, previous_p_c
    return adv, now_p


# ALP (attacks.py:236-272)
class ALP:
    def __init__(self, step_size=0.003, epsilon=0.047, perturb_steps=5, beta=1.0):
        self.step_size = step_size
        self.epsilon = epsilon
        self.perturb_steps = perturb_steps
        self.beta = beta

    def reset_steps(self, k):
        self.perturb_steps = k

    def PGD_Linf(self, model, x_natural, y, noise=None):
        model.eval()  # side effect kept (attacks.py:249)
        x0 = x_natural.detach()
        x = _randn_start(x0, noise)
        return _loop(model, x0, x, engine.LossSpec(engine.CE_MEAN, y), lambda z: F.cross_entropy(z, y),
                     self.perturb_steps, self.step_size, self.epsilon)

    def loss(self, model, logits, logits_adv, y, optimizer):
        model.train()  # side effects kept (attacks.py:265-266)
        optimizer.zero_grad()
        if runtime.require_device(logits, "ALP.loss"):
            loss_robust = 0.5 * EF.cross_entropy(logits, y) + 0.5 * EF.cross_entropy(logits_adv, y)
            return loss_robust + self.beta * EF.mse_loss(logits, logits_adv)
        loss_robust = 0.5 * F.cross_entropy(logits, y) + 0.5 * F.cross_entropy(logits_adv, y)
        return loss_robust + self.beta * F.mse_loss(logits, logits_adv)


# Targeted ALP for Tiny ImageNet (attacks.py:276-333)
class targeted_ALP(ALP):
    def __init__(self, step_size=0.003, epsilon=0.047, perturb_steps=5, beta=1.0, n_class=200):
        ALP.__init__(self, step_size, epsilon, perturb_steps, beta)
        self.n_class = n_class

    def tarPGD_Linf(self, model, x_natural, y, device, noise=None, label_offset=None):
        model.eval()
        target_labels = _random_targets(y, self.n_class, device, label_offset)
        x0 = x_natural.detach()
        x = _randn_start(x0, noise)
        return _loop(model, x0, x, engine.LossSpec(engine.CE_MEAN, target_labels),
                     lambda z: F.cross_entropy(z, target_labels), self.perturb_steps, self.step_size, self.epsilon, -1)


# Targeted ALP for ImageNet (attacks.py:337-357)
def tar_alp_imagenet(model, args, inputs, labels, num_steps, step_size, device, noise=None, label_offset=None):
    x0 = inputs.detach()
    target_labels = _random_targets(labels, 1000, device, label_offset)
    x = _randn_start(x0, noise)
    x = _loop(model, x0, x, engine.LossSpec(engine.CE_SUM, target_labels),
              lambda z: F.cross_entropy(z, target_labels, reduction='sum'), num_steps, step_size, args.epsilon, -1)
    return x, target_labels


def squared_l2_norm(x):
    """attacks.py:360-362 (note: MEAN of squares)."""
    flattened = x.view(x.shape[0], -1)
    return (flattened ** 2).mean(1)


def l2_norm(x):
    return squared_l2_norm(x).sqrt()


class _KLBatchMean(nn.Module):
    """Stands in for nn.KLDivLoss(reduction='batchmean') as `Trades.criterion_kl`: called with
    (log_softmax(q_logits), p) it defers to torch; the loops and `loss` call the fused kernel on logits."""

    def __init__(self):
        super().__init__()
        self._torch = nn.KLDivLoss(reduction="batchmean")

    def forward(self, log_q, p):
        return self._torch(log_q, p)


# TRADES (attacks.py:369-429)
class Trades:
    def __init__(self, step_size=0.003, epsilon=0.047, perturb_steps=5, beta=1.0):
        self.step_size = step_size
        self.epsilon = epsilon
        self.perturb_steps = perturb_steps
        self.beta = beta
        self.criterion_kl = _KLBatchMean()

    def reset_steps(self, k):
        self.perturb_steps = k

    def PGD_L2(self, model, x_natural, logits, noise=None):
        """attacks.py:381-401: gradient normalised by its per-sample RMS, step alpha*g, RMS-ball projection, clamp -
        one kernel per iteration (ee_l2_step_f32) behind the KL loss-gradient kernel."""
        model.eval()
        x0 = x_natural.detach()
        if runtime.require_device(x0, "Trades.PGD_L2"):
            x0 = x0.contiguous()
            x = _randn_start(x0, noise)
            spec = engine.LossSpec(engine.KL, logits.detach().contiguous())
            net = engine._unwrap(model)
            for _ in range(self.perturb_steps):
                g = engine.input_gradient(net, x, spec)
                x = ops.l2_step_(x.detach(), g.contiguous(), x0, float(self.step_size), float(self.epsilon), 0.0, 1.0)
            return x.detach()
        nz = torch.randn(x_natural.shape, device=x_natural.device) if noise is None else noise.to(x_natural.device)
        x_adv = x_natural.detach() + 0.001 * nz.detach()
        prob = F.softmax(logits, dim=-1)
        for _ in range(self.perturb_steps):
            with torch.enable_grad():
                x_adv.requires_grad_()
                loss_kl = self.criterion_kl(F.log_softmax(model(x_adv), dim=1), prob)
            grad = torch.autograd.grad(loss_kl, [x_adv])[0].detach()
            grad /= l2_norm(grad).unsqueeze(-1).unsqueeze(-1).unsqueeze(-1) + 1e-8
            x_adv = x_adv.detach() + self.step_size * grad
            delta = x_adv - x_natural
            delta_norm = l2_norm(delta)
            cond = delta_norm > self.epsilon
            delta[cond] *= self.epsilon / delta_norm[cond].unsqueeze(-1).unsqueeze(-1).unsqueeze(-1)
            x_adv = torch.clamp(x_natural + delta, 0.0, 1.0)
        return x_adv

    def PGD_Linf(self, model, x_natural, logits, noise=None):
        model.eval()  # side effect kept (attacks.py:405), never restored here
        x0 = x_natural.detach()
        x = _randn_start(x0, noise)
        nat = logits.detach().contiguous()  # softmax(logits) is a constant w.r.t. x_adv (attacks.py:407)
        prob = F.softmax(nat, dim=-1)
        return _loop(model, x0, x, engine.LossSpec(engine.KL, nat),
                     lambda z: self.criterion_kl(F.log_softmax(z, dim=1), prob), self.perturb_steps, self.step_size,
                     self.epsilon)

    def loss(self, model, logits, x_adv, labels, optimizer):
        model.train()  # side effects kept (attacks.py:422-423)
        optimizer.zero_grad()
        if runtime.require_device(logits, "Trades.loss"):
            loss_natural = EF.cross_entropy(logits, labels)
            self.last_logits_adv = model(x_adv)  # kept for eeadv.trainer.two_branch_backward (one backward per forward pass)
            loss_robust = EF.kl_div_batchmean(self.last_logits_adv, logits)  # gradient flows into both arguments
            return loss_natural + self.beta * loss_robust
        prob = F.softmax(logits, dim=-1)
        loss_natural = F.cross_entropy(logits, labels)
        loss_robust = self.criterion_kl(F.log_softmax(model(x_adv), dim=1), prob)
        return loss_natural + self.beta * loss_robust


# AVmixup (attacks.py:433-518)
class AVmixup:
    def __init__(self, args, gamma, lambda1, lambda2, step_size, num_steps, num_classes=200, device='cuda'):
        self.args = args
        self.gamma = gamma
        self.lambda1 = lambda1
        self.lambda2 = lambda2
        self.step_size = step_size
        self.num_steps = num_steps
        self.num_classes = num_classes
        self.device = device

    def _label_smoothing(self, one_hot, factor):
        return one_hot * factor + (one_hot - 1.) * ((factor - 1) / float(self.num_classes - 1))

    def _vertex_mix(self, inputs, x, targets, beta):
        """attacks.py:469-479.  The mixing weight is numpy float64, so the labels come back float64."""
        x_weight = np.random.beta(1.0, 1.0, [x.shape[0], 1, 1, 1]) if beta is None else beta
        if runtime.require_device(x, "AVmixup"):
            w = torch.from_numpy(np.ascontiguousarray(x_weight, dtype=np.float64).reshape(-1)).to(x.device)
            x_mix = ops.avmix(x.contiguous(), inputs.contiguous(), w, float(self.gamma))
            y_nat = self._label_smoothing(targets, self.lambda1)
            y_vertex = self._label_smoothing(targets, self.lambda2)
            yw = w.view(-1, 1)
            return x_mix, y_nat * yw + y_vertex * (1 - yw)
        perturb = (x - inputs) * self.gamma
        vertex = torch.clamp(inputs + perturb, 0, 1)
        y_nat = self._label_smoothing(targets, self.lambda1)
        y_vertex = self._label_smoothing(targets, self.lambda2)
        xw = torch.from_numpy(x_weight).to(self.device)
        yw = torch.from_numpy(np.reshape(x_weight, [-1, 1])).to(self.device)
        return (inputs * xw + vertex * (1 - xw)).to(torch.float), y_nat * yw + y_vertex * (1 - yw)

    def perturb(self, model, inputs, targets, noise=None, beta=None):
        """Given (inputs, one-hot targets) returns (mixed adversarial-vertex inputs, mixed soft labels)."""
        x0 = inputs.detach()
        x = _uniform_start(x0, self.args.epsilon, noise) if self.args.random else x0.clone()
        soft = targets.detach()
        x = _loop(model, x0, x, engine.LossSpec(engine.SOFTCE, soft.to(torch.float64).contiguous()),
                  lambda z: -torch.sum(F.log_softmax(z, dim=1) * soft), self.num_steps, self.step_size, self.args.epsilon)
        return self._vertex_mix(x0, x, targets, beta)

    def tar_perturb(self, model, inputs, targets, noise=None, beta=None, label_offset=None):
        """attacks.py:481-518.  As in the reference the loss multiplies log-probabilities [B,K] by the integer
        target LABELS `fmod(targets + offset, K)`; `targets` is whatever the driver passes (one-hot there)."""
        x0 = inputs.detach()
        if label_offset is None:
            label_offset = torch.randint(low=1, high=self.num_classes, size=targets.shape).to(self.device)
        target_labels = torch.fmod(targets + label_offset.to(targets.device), self.num_classes)
        x = _uniform_start(x0, self.args.epsilon, noise) if self.args.random else x0.clone()
        tl = target_labels.detach()
        x = _loop(model, x0, x, engine.LossSpec(engine.SOFTCE, tl.to(torch.float64).contiguous()),
                  lambda z: -torch.sum(F.log_softmax(z, dim=1) * tl), self.num_steps, self.step_size, self.args.epsilon, -1)
        return self._vertex_mix(x0, x, targets, beta)
