#!/usr/bin/env python
# -*- encoding: utf-8 -*-
""""Free" adversarial training, DDP (reference: ImageNet/free_imagenet/AT_free_imagenet_ddp.py; BASELINE config 5).

One forward/backward per repeat yields both the weight gradient and the input gradient (:286-309):
    in1 = clamp(x + delta[:B], 0, 1)               -> ee_add_clamp_f32        (:289-290)
    loss.backward()                                   (DDP all-reduce over RCCL overlaps this backward)
    delta[:B] += fgsm_step * sign(grad); clamp_(+-eps) -> ee_freeat_update_f32   (:305-307)
The noise buffer persists across batches and epochs and is never reset or checkpointed, as in the reference.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 AT_free_imagenet_ddp.py -a resnet50 --data synthetic
"""
import argparse
import math
import os
import sys
import time

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import models_imagenet as zoo  # noqa: E402
from eeadv import ddp, driver, ops, trainer  # noqa: E402
from utils.helper import AverageMeter, accuracy, adjust_learning_rate_free, set_seed  # noqa: E402


def parse_args(argv=None):
    """AT_free_imagenet_ddp.py:38-108 (argparse only, no YAML)."""
    p = argparse.ArgumentParser(description='PyTorch ImageNet free adversarial training')
    p.add_argument('--data', default='synthetic')
    p.add_argument('-a', '--arch', default='resnet152')
    p.add_argument('--epochs', default=90, type=int)
    p.add_argument('--start-epoch', default=0, type=int)
    p.add_argument('-b', '--batch_size', default=256, type=int)
    p.add_argument('--lr', default=0.1, type=float)
    p.add_argument('--momentum', default=0.9, type=float)
    p.add_argument('--weight_decay', default=1e-4, type=float)
    p.add_argument('-p', '--print_freq', default=10, type=int)
    p.add_argument('--seed', default=1, type=int)
    p.add_argument('--local_rank', default=int(os.environ.get("LOCAL_RANK", "0")), type=int)
    p.add_argument('--n_repeats', default=4, type=int)
    p.add_argument('--fgsm_step', default=4.0, type=float)
    p.add_argument('--clip_eps', default=4.0, type=float)
    p.add_argument('--crop_size', default=224, type=int)
    p.add_argument('--max-batches', default=None, type=int, help='stop an epoch early (smoke runs)')
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    torch.cuda.set_device(ddp.local_rank())
    device = torch.device("cuda", ddp.local_rank())
    ddp.setup(device)
    set_seed(ddp.rank_seed(args.seed))
    args.epochs = int(math.ceil(args.epochs / args.n_repeats))  # :129-131
    args.fgsm_step /= 255.0
    args.clip_eps /= 255.0
    model = getattr(zoo, args.arch)().to(device)
    net = ddp.wrap(model, device, sync_bn=True)
    criterion = trainer.Criterion()
    optimizer = torch.optim.SGD(net.parameters(), args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
    B = ddp.per_rank_batch(args.batch_size)
    spec = {"shape": (3, args.crop_size, args.crop_size), "num_classes": 1000}
    args.data = args.data if str(args.data).startswith("synthetic") else "synthetic"
    loader, _ = driver.make_loaders(args, spec, device, B)
    noise = torch.zeros([args.batch_size, 3, args.crop_size, args.crop_size], device=device)  # :261, global batch size on every rank
    for epoch in range(args.start_epoch, args.epochs):
        adjust_learning_rate_free(optimizer, epoch, args.lr, args.n_repeats)
        net.train()
        batch_time, losses, top1 = AverageMeter(), AverageMeter(), AverageMeter()
        end = time.time()
        for i, (input, target) in enumerate(loader):
            if args.max_batches is not None and i >= args.max_batches:
                break
            n = input.size(0)
            for _ in range(args.n_repeats):
                delta = noise[0:n]
                in1 = ops.add_clamp(input.contiguous(), delta, 0.0, 1.0).requires_grad_(True)
                output = net(in1)
                loss = criterion(output, target)
                optimizer.zero_grad()
                loss.backward()
                # Variable(noise).grad of the reference = in1.grad masked by the in-place clamp; the kernel applies the mask
                ops.freeat_update_masked_(noise, in1.grad.contiguous(), input.contiguous(), args.fgsm_step, args.clip_eps)
                optimizer.step()
            batch_time.update(time.time() - end)
            end = time.time()
            if i % args.print_freq == 0 and ddp.rank() == 0:
                prec1, _ = accuracy(output.detach(), target, topk=(1, 5))
                losses.update(loss.item(), n)
                top1.update(prec1.item(), n)
                print('Epoch: [{0}][{1}/{2}]\tTime {bt.val:.3f} ({bt.avg:.3f})\tLoss {l.val:.4f} ({l.avg:.4f})\tPrec@1 {t.val:.3f} ({t.avg:.3f})'
                      .format(epoch, i, len(loader), bt=batch_time, l=losses, t=top1))
    ddp.teardown()


if __name__ == '__main__':
    main()
