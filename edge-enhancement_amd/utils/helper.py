#!/usr/bin/env python
# -*- encoding: utf-8 -*-
"""Drop-in for the reference's utils/helper.py: seeds, meters, top-k accuracy, checkpoints, LR schedules and
the YAML+CLI config merge.  `accuracy` runs the HIP top-k kernel for ROCm logits."""
import math
import random
import shutil

import numpy as np
import torch
import yaml

from eeadv import ops


class EasyDict(dict):
    """Attribute-style dict (the reference depends on the `easydict` package for this, helper.py:9)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def set_seed(seed):  # helper.py:11-17
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


class AverageMeter(object):
    """Computes and stores the average and current value (helper.py:20-36)"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def accuracy(output, target, topk=(1,)):
    """Computes the precision@k for the specified values of k (helper.py:39-55).  Soft targets ([B,K]) are
    reduced to their arg-max first (:45-46).  Returns a list of 1-element float tensors, like the reference."""
    maxk = max(topk)
    batch_size = target.size(0)
    if target.shape == output.shape:
        _, target = target.topk(1, 1, largest=True, sorted=True)
        target = target.view(-1)
    if output.is_cuda and output.dtype == torch.float32 and output.dim() == 2 and maxk <= 16:
        _, correct = ops.topk(output.detach().contiguous(), target.contiguous().to(torch.int64), maxk)
        return [correct[k - 1:k].to(torch.float32).mul_(100.0 / batch_size) for k in topk]
    _, pred = output.topk(maxk, 1, largest=True, sorted=True)
    pred = pred.t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / batch_size) for k in topk]


def save_checkpoint(state, is_best, filename, bestfilename):  # helper.py:58-61
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, bestfilename)


def adjust_learning_rate(optimizer, epoch, init_lr):
    """Sets the learning rate to the initial LR decayed by 10 every 30 epochs (helper.py:64-68)"""
    lr = init_lr * (0.1 ** (epoch // 30))
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr


def adjust_learning_rate_free(optimizer, epoch, init_lr, n_repeats):
    """helper.py:71-75"""
    lr = init_lr * (0.1 ** (epoch // int(math.ceil(30. / n_repeats))))
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr


def adjust_learning_rate_1(optimizer, epoch, init_lr, epochs):
    """Decay 0.1 at 50% and 75% of total epochs (helper.py:78-88)"""
    if epoch > epochs * 0.75:
        lr = init_lr * (0.1 ** 2)
    elif epoch > epochs * 0.5:
        lr = init_lr * 0.1
    else:
        lr = init_lr
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr


def compute_attack_success(logits, target_label):
    """helper.py:103-112"""
    _, predicted_clean = logits.max(1)
    return predicted_clean.eq(target_label).sum().item()


def parse_config_file(args):
    """Flat YAML -> EasyDict, CLI arguments merged on top (helper.py:115-127)."""
    with open(args.config) as f:
        config = EasyDict(yaml.safe_load(f))
    for k, v in vars(args).items():
        config[k] = v
    return config
