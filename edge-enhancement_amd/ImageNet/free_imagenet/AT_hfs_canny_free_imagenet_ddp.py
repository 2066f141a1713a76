#!/usr/bin/env python
# -*- encoding: utf-8 -*-
""""Free" adversarial training of the edge-enhanced ImageNet models, DDP (reference:
ImageNet/free_imagenet/AT_hfs_canny_free_imagenet_ddp.py).

The loop is AT_free_imagenet_ddp.py's (same file, :286-309 there = :311-334 here): one forward/backward per repeat gives
the weight gradient and the input gradient; the noise update, projection and `clamp(x + noise)` are HIP kernels
(eeadv.trainer.free_at_repeat).  What this script changes, as the reference does:
  * the models: `resnet50` or the EE front end on 224 x 224 (HighFreqSuppress r = 16 on the matrix cores - ee_hfs_mfma_f32 -,
    Add_Square, CannyFilter_step125_1 fused with the combine - ee_frontend_fwd/bwd_f32 -, utils/core.py);
  * its defaults (:42, :75-115): arch resnet50_EE_square, sigma 1, w 1, r 16, low 38, high 76, --type_canny, --n_queries,
    --num-steps-2/3, --step-size-2/3;
  * the evaluation attack: PGD with num_steps_3 / step_size_3 (:377);
  * the output directory `<arch>/<type_canny>_clip-eps<e>/` (:197) and the argument dump at the head of log.txt (:209-211).
Reference quirk kept, on purpose and visibly: `-a resnet50_EE_square` builds **resnet18**_EE_square (:162-165); the depths the
reference's constructor chain offers but its `choices` list rejects (`resnet{18,50,101,152}_EE`, :148-161) are selectable here.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 AT_hfs_canny_free_imagenet_ddp.py --data synthetic
"""
import os
import sys

sys.path.append(os.path.dirname(os.path.abspath(__file__)))
import AT_free_imagenet_ddp as base  # noqa: E402
from AT_free_imagenet_ddp import ddp, zoo  # noqa: E402

ARCHS = ('resnet50', 'resnet18_EE', 'resnet50_EE', 'resnet101_EE', 'resnet152_EE', 'resnet50_EE_square')


def make_parser():
    """:38-115: the base flags with this script's defaults, plus its own."""
    p = base.make_parser(ARCHS, 'resnet50_EE_square')
    p.set_defaults(num_steps_1=10, sigma=1, w=1, r=16, low=38, high=76)  # integer defaults, as the reference's (they reach the file names)
    p.add_argument('--num-steps-2', type=int, default=50)
    p.add_argument('--step-size-2', type=float, default=1.0 / 255)
    p.add_argument('--num-steps-3', type=int, default=100)
    p.add_argument('--step-size-3', type=float, default=1.0 / 255)
    p.add_argument('--type_canny', '--type-canny', default='CannyFilter_step125_1', type=str)
    p.add_argument('--n_queries', '--n-queries', default=1, type=int)
    return p


def build_model(args):
    """:143-165"""
    ee = dict(cize=args.cize, r=args.r, w=args.w, with_gf=args.gf, low=args.low, high=args.high, alpha=args.alpha, sigma=args.sigma,
              num_classes=args.num_classes)
    if args.arch == 'resnet50':
        return zoo.resnet50(num_classes=args.num_classes)
    if args.arch == 'resnet50_EE_square':  # :162-165 - resnet18_EE_square, whatever the name says
        return zoo.resnet18_EE_square(type_canny=args.type_canny, epsilon=args.epsilon, n_queries=args.n_queries, **ee)
    return getattr(zoo, args.arch)(**ee)


def output_dirs(args):
    """:197-207: cwd/checkpoint_free_imagenet/free_AT_ddp/<arch>/<type_canny>_clip-eps<e>/{model_pth,best_model_pth,log}/"""
    root = args.output_root or os.getcwd()
    d = root + '/checkpoint_free_imagenet/free_AT_ddp/' + str(args.arch) + '/' + str(args.type_canny) + '_clip-eps' + \
        str(int(round(args.clip_eps * 255))) + '/'
    dirs = {"root": d, "model": d + 'model_pth/', "best": d + 'best_model_pth/', "log": d + 'log/'}
    if ddp.rank() == 0:
        for k in ("log", "model", "best"):
            os.makedirs(dirs[k], exist_ok=True)
    return dirs


def main(argv=None):
    return base.main(argv, parser=make_parser(), build=build_model, dirs_of=output_dirs,
                     eval_attack=lambda args: (args.num_steps_3, args.step_size_3), log_args=True)


if __name__ == '__main__':
    main()
