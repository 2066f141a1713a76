#!/usr/bin/env python
# -*- encoding: utf-8 -*-
"""MNIST driver (reference: MNIST/experiments_mnist.py).  Same CLI, YAML keys, `arch` / `method_name` strings,
log lines and checkpoint names; the loop bodies live in eeadv.driver / eeadv.trainer.

    python experiments_mnist.py -c configs_mnist/standard_training.yml --no-cuda        # BASELINE config 1 (CPU plumbing)
    python experiments_mnist.py -c configs_mnist/ee_at_bpda3_square.yml                 # BASELINE config 2 (1x MI355X)
"""
import os
import sys

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from eeadv import driver  # noqa: E402
from models_mnist import Net_2, Net2_EE, Net2_EE_square  # noqa: E402

SPEC = {"description": "PyTorch Mnist Training", "ckpt_dir": "checkpoint_MNIST", "shape": (1, 28, 28), "num_classes": 10,
        "mnist_top5_quirk": True, "ckpt_module_prefix": True}  # checkpoints of nn.DataParallel(model): `module.` keys (:77,:168)


def build_model(args):
    """experiments_mnist.py:60-72."""
    if args.arch == 'Net2':
        return Net_2()
    if args.arch == 'Net2_EE':
        print('r:{},w:{},gf:{},low:{},high:{}'.format(args.r, args.w, args.gf, args.low, args.high))
        return Net2_EE(r=args.r, w=args.w, with_gf=args.gf, low=args.low, high=args.high, alpha=args.alpha, sigma=args.sigma,
                       type_canny=args.type_canny if args.type_canny not in (None, "None") else 'CannyFilter')
    if args.arch == 'Net2_EE_square':
        return Net2_EE_square(r=args.r, w=args.w, with_gf=args.gf, low=args.low, high=args.high, alpha=args.alpha, sigma=args.sigma,
                              type_canny=args.type_canny, epsilon=args.epsilon, n_queries=args.n_queries)
    raise NotImplementedError


if __name__ == '__main__':
    driver.run(SPEC, build_model)
