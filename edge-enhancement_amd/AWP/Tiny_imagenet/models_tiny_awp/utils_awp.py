"""Adversarial Weight Perturbation around the hot path (reference: AWP/Tiny_imagenet/models_tiny_awp/utils_awp.py, identical twin under
AWP/Cifar100/models_cifar100_awp/; called from experiments_tiny_awp.py:256-286 around `PGD(model, args, input, target, ...)`).

SURVEY 2.1 #11 / 8(f5): AWP is a weight-space method that only CALLS the path; what is kept here is its interface - `diff_in_weights`,
`add_into_weights`, `AdvWeightPerturb(model, proxy, proxy_optim, gamma)` with `.calc_awp(inputs_adv, targets)`, `.perturb(diff)`,
`.restore(diff)` - so that the reference's AWP train loop runs on top of utils.attacks.PGD unchanged.  The PreActResNet model zoo of
that sub-project is out of scope (any model of eeadv.models works as `model` / `proxy`).

Arithmetic as the reference's, tensor by tensor: for every state_dict entry with more than one dimension whose key contains 'weight',
diff = ||w_model|| / (||w_proxy - w_model|| + 1e-20) * (w_proxy - w_model); perturb / restore add +-gamma * diff to the parameters of
those names.  On a ROCm device the per-tensor norms and updates run as multi-tensor (foreach) launches - a handful per call instead of
four per tensor; the values are the same fp32 operations in the same order per tensor.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

EPS = 1E-20


def diff_in_weights(model, proxy):
    """utils_awp.py:8-18"""
    names, olds, news = [], [], []
    for (old_k, old_w), (new_k, new_w) in zip(model.state_dict().items(), proxy.state_dict().items()):
        if len(old_w.size()) <= 1:
            continue
        if 'weight' in old_k:
            names.append(old_k)
            olds.append(old_w)
            news.append(new_w)
    diffs = torch._foreach_sub(news, olds)
    old_norms = torch._foreach_norm(olds)
    diff_norms = torch._foreach_norm(diffs)
    torch._foreach_add_(diff_norms, EPS)
    torch._foreach_div_(old_norms, diff_norms)  # old_w.norm() / (diff_w.norm() + EPS)
    return OrderedDict(zip(names, torch._foreach_mul(diffs, old_norms)))  # ... * diff_w


def add_into_weights(model, diff, coeff=1.0):
    """utils_awp.py:21-26"""
    names_in_diff = diff.keys()
    with torch.no_grad():
        params, deltas = [], []
        for name, param in model.named_parameters():
            if name in names_in_diff:
                params.append(param)
                deltas.append(diff[name])
        if params:
            torch._foreach_add_(params, torch._foreach_mul(deltas, coeff))  # param.add_(coeff * diff[name])


class AdvWeightPerturb(object):
    """utils_awp.py:29-54"""

    def __init__(self, model, proxy, proxy_optim, gamma):
        super(AdvWeightPerturb, self).__init__()
        self.model = model
        self.proxy = proxy
        self.proxy_optim = proxy_optim
        self.gamma = gamma

    def calc_awp(self, inputs_adv, targets):
        self.proxy.load_state_dict(self.model.state_dict())
        self.proxy.train()
        output = self.proxy(inputs_adv)
        if output.is_cuda:
            from eeadv import functional as EF
            loss = - EF.cross_entropy(output, targets, "mean")  # the HIP loss kernel (ee_ce_f32), as the drivers' criterion
        else:
            loss = - F.cross_entropy(output, targets)
        self.proxy_optim.zero_grad()
        loss.backward()
        self.proxy_optim.step()
        # the adversary weight perturb
        return diff_in_weights(self.model, self.proxy)

    def perturb(self, diff):
        add_into_weights(self.model, diff, coeff=1.0 * self.gamma)

    def restore(self, diff):
        add_into_weights(self.model, diff, coeff=-1.0 * self.gamma)
