#!/usr/bin/env python
# -*- encoding: utf-8 -*-
"""Tiny-ImageNet driver (reference: Tiny_ImageNet/experiments_tinyimagenet.py).  Same CLI, YAML keys, `arch` /
`method_name` strings (ST, AT, ALP, tarALP, TRADES, AVmixup, tarAVmixup, tarAT, tarEE, tarEE_BPDA3_AT_square,
*pre_square*), log lines and checkpoint names.

    python experiments_tinyimagenet.py -c configs_tinyimagenet/trades_training.yml              # BASELINE config 3
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        experiments_tinyimagenet.py -c configs_tinyimagenet/ee_at_bpda3_square.yml              # BASELINE config 4 (DDP over RCCL)
"""
import os
import sys

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import models_tinyimagenet as zoo  # noqa: E402
from eeadv import driver  # noqa: E402

SPEC = {"description": "PyTorch Tiny ImageNet Training", "ckpt_dir": "checkpoint_Tiny_ImageNet", "shape": (3, 64, 64), "num_classes": 200,
        "ckpt_module_prefix": True}  # the reference saves nn.DataParallel(model).state_dict() (:110,:196): `module.`-prefixed keys


def build_model(args):
    """experiments_tinyimagenet.py:65-105."""
    arch = args.arch
    if arch in ('resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152'):
        return getattr(zoo, arch)(pretrained=args.pretrained)
    ee = dict(pretrained=args.pretrained, cize=args.cize, r=args.r, w=args.w, with_gf=args.gf, low=args.low, high=args.high,
              alpha=args.alpha, sigma=args.sigma, type_canny=args.type_canny)
    if arch in ('resnet18_EE', 'resnet34_EE', 'resnet50_EE', 'resnet101_EE', 'resnet152_EE'):
        print('r:{},w:{},gf:{},low:{},high:{}'.format(args.r, args.w, args.gf, args.low, args.high))
        return getattr(zoo, arch)(**ee)
    if arch == 'resnet18_EE_square':
        return zoo.resnet18_EE_square(epsilon=args.epsilon, n_queries=args.n_queries, **ee)
    raise NotImplementedError


if __name__ == '__main__':
    driver.run(SPEC, build_model)
