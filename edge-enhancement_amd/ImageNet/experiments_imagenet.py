#!/usr/bin/env python
# -*- encoding: utf-8 -*-
"""ImageNet DDP driver (reference: ImageNet/experiments_imagenet.py): one process per GPU over RCCL, SyncBatchNorm,
per-rank batch = batch_size / world, seeds seed + rank, 30-epoch step LR.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 experiments_imagenet.py \
        -c configs_imagenet/ee_at_bpda3_square.yml
"""
import os
import sys

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import models_imagenet as zoo  # noqa: E402
from eeadv import driver  # noqa: E402

SPEC = {"description": "PyTorch ImageNet Training", "ckpt_dir": "checkpoint_ImageNet", "shape": (3, 224, 224), "num_classes": 1000,
        "ddp": True, "sync_bn": True, "lr_schedule": "step30"}


def build_model(args):
    """experiments_imagenet.py:66-122."""
    arch = args.arch
    if arch in ('resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152'):
        return getattr(zoo, arch)(pretrained=args.pretrained)
    ee = dict(pretrained=args.pretrained, cize=args.cize, r=args.r, w=args.w, with_gf=args.gf, low=args.low, high=args.high,
              alpha=args.alpha, sigma=args.sigma, type_canny=args.type_canny if args.type_canny not in (None, "None") else 'CannyFilter')
    if arch.endswith('_EE_square'):
        return getattr(zoo, arch)(epsilon=args.epsilon, n_queries=args.n_queries, **ee)
    if arch.endswith('_EE'):
        return getattr(zoo, arch)(**ee)
    raise NotImplementedError


if __name__ == '__main__':
    driver.run(SPEC, build_model)
