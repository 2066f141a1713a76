#!/usr/bin/env python
# -*- encoding: utf-8 -*-
""""Free" adversarial training, DDP (reference: ImageNet/free_imagenet/AT_free_imagenet_ddp.py; BASELINE config 5).

One forward/backward per repeat yields both the weight gradient and the input gradient (:286-309); the three
element-wise groups around it are HIP kernels (eeadv.trainer.free_at_repeat).  The noise buffer persists across
batches and epochs and is never reset or checkpointed, as in the reference.  Kept from the reference: the argparse
surface (:36-108, no YAML), epochs / n_repeats and the /255 scalings (:129-131), SyncBatchNorm + DDP (:149-152), per-rank
batch = batch_size / world, PGD evaluation after every epoch (:326-331), best_prec1 on the ADVERSARIAL top-1 (:403),
checkpoint dict keys / file names / directory layout (:176-256, `state_dict` of the WRAPPED model, i.e. `module.`-prefixed
keys), --resume / --evaluate, the print formats.  Fixed: `PGD` is imported from utils.attacks (the reference imports it from
utils.core, :16, and cannot start); the per-rank batch uses the real world size instead of --nGPU.

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
from eeadv import ddp, driver, trainer  # noqa: E402
from utils.attacks import PGD  # noqa: E402
from utils.helper import AverageMeter, accuracy, adjust_learning_rate_free, save_checkpoint, set_seed  # noqa: E402

ARCHS = ('resnet18', 'resnet50', 'resnet101', 'resnet152')  # :137-146


def make_parser(archs=ARCHS, default_arch='resnet152'):
    """AT_free_imagenet_ddp.py:36-108, flag for flag."""
    p = argparse.ArgumentParser(description='PyTorch ImageNet Training')
    p.add_argument('--data', metavar='DIR', default='synthetic', help='path to dataset, or synthetic[:train_batches[:val_batches]]')
    p.add_argument('-a', '--arch', metavar='ARCH', default=default_arch, choices=archs)
    p.add_argument('--epochs', default=90, type=int, metavar='N')
    p.add_argument('--start-epoch', default=0, type=int, metavar='N')
    p.add_argument('-b', '--batch-size', '--batch_size', default=256, type=int, metavar='N')
    p.add_argument('--lr', '--learning-rate', default=0.1, type=float, metavar='LR')
    p.add_argument('--momentum', default=0.9, type=float, metavar='M')
    p.add_argument('--weight-decay', '--wd', '--weight_decay', default=1e-4, type=float, metavar='W')
    p.add_argument('-j', '--workers', default=4, type=int, metavar='N')
    p.add_argument('-m', '--pin-memory', default=True, dest='pin_memory', action='store_true')
    p.add_argument('-p', '--pretrained', default=False, dest='pretrained', action='store_true')
    p.add_argument('--print-freq', '--print_freq', '-f', default=100, type=int, metavar='N')
    p.add_argument('--resume', default='', type=str, metavar='PATH')
    p.add_argument('-e', '--evaluate', dest='evaluate', action='store_true')
    p.add_argument('--no-cuda', action='store_true', default=False)
    p.add_argument('--seed', type=int, default=1, metavar='S')
    p.add_argument('--epsilon', type=float, default=4.0 / 255)
    p.add_argument('--num-steps-1', type=int, default=50)
    p.add_argument('--step-size-1', type=float, default=1.0 / 255)
    p.add_argument('--random', default=True)
    p.add_argument('--cize', default=224, type=int)
    p.add_argument('--alpha', type=float, default=0)
    p.add_argument('--sigma', type=float, default=0)
    p.add_argument('--clip-eps', '--clip_eps', default=4.0, type=float)
    p.add_argument('--fgsm-step', '--fgsm_step', default=4.0, type=float)
    p.add_argument('--max-color-value', default=255.0, type=float)
    p.add_argument('--crop-size', '--crop_size', default=224, type=int)
    p.add_argument('--n-repeats', '--n_repeats', default=4, type=int)
    p.add_argument('--w', default=0, type=float)
    p.add_argument('--r', default=0, type=int)
    p.add_argument('--gf', default=False, action='store_true')
    p.add_argument('--low', default=0, type=float)
    p.add_argument('--high', default=0, type=float)
    p.add_argument('--local_rank', default=int(os.environ.get("LOCAL_RANK", "0")), type=int)
    p.add_argument('--nGPU', default=4, type=int, help='kept for the command line; the world size is what counts')
    # additions (smoke runs / tests)
    p.add_argument('--max-batches', default=None, type=int, help='stop an epoch early')
    p.add_argument('--max-epochs', default=None, type=int, help='stop after this many epochs')
    p.add_argument('--num-classes', default=1000, type=int, help='classes of the synthetic labels and the classifier')
    p.add_argument('--output-root', default=None, help='where checkpoint_free_imagenet/ is created (default: cwd, as the reference)')
    return p


def output_dirs(args):
    """:176-191: cwd/checkpoint_free_imagenet/free_AT_ddp/<arch>Baseline_clip-eps<e>/{model_pth,best_model_pth,log}/"""
    root = args.output_root or os.getcwd()
    d = root + '/checkpoint_free_imagenet/free_AT_ddp/' + str(args.arch) + 'Baseline' + '_clip-eps' + str(int(round(args.clip_eps * 255))) + '/'
    dirs = {"root": d, "model": d + 'model_pth/', "best": d + 'best_model_pth/', "log": d + 'log/'}
    if ddp.rank() == 0:
        for k in ("log", "model", "best"):
            os.makedirs(dirs[k], exist_ok=True)
    return dirs


def checkpoint_names(args, dirs, epoch):
    """:241-254 (int(clip_eps*255) of the reference truncates 3.9999 to 3 for eps = 4/255 on some hosts; rounded here)."""
    stem = ('at_clip-eps' + str(int(round(args.clip_eps * 255))) + '_fgsm-step' + str(int(round(args.fgsm_step * 255))) + '_n-repeats' +
            str(args.n_repeats) + '_r' + str(args.r) + '_canny_sigma' + str(args.sigma) + '_alpha' + str(args.alpha) + '-bs' +
            str(args.batch_size) + '-lr_' + str(args.lr) + '-w' + str(args.w) + '-gf' + str(args.gf) + '-l' + str(args.low) + '-h' +
            str(args.high) + '-ty1_')
    return dirs["model"] + stem + str(epoch) + '.pth', dirs["best"] + stem + '.pth'


def _say(line, log_dir):
    """The reference prints (its log.txt lines are commented out, :326-327); the file is written too, like the other drivers."""
    if ddp.rank() == 0:
        print(line)
        with open(log_dir + 'log.txt', 'a') as f:
            print(line, file=f)


_STEPS = {}  # id(model) -> trainer.FreeAtStep (its captured repeat lives across epochs; a new learning rate re-captures)


def train(train_loader, net, criterion, optimizer, epoch, args, device, log_dir, noise, sync=None):
    """:263-327."""
    batch_time, data_time, losses, top1, top5 = (AverageMeter() for _ in range(5))
    net.train()
    step = _STEPS.get(id(net))
    if step is None or step.noise is not noise or step.optimizer is not optimizer:
        step = _STEPS[id(net)] = trainer.FreeAtStep(net, criterion, optimizer, noise, args.fgsm_step, args.clip_eps, args.n_repeats, sync=sync)
    end = time.time()
    for i, (input, target) in enumerate(train_loader):
        if args.max_batches is not None and i >= args.max_batches:
            break
        target, input = target.to(device), input.to(device)
        data_time.update(time.time() - end)
        loss, output = step(input, target)  # the n_repeats repeats of :286-309
        batch_time.update((time.time() - end) / args.n_repeats, args.n_repeats)
        end = time.time()
        if i % args.print_freq == 0:  # the reference syncs with .item() on every repeat (:296); here only when it prints
            prec1, prec5 = accuracy(output, target, topk=(1, min(5, args.num_classes)))
            losses.update(loss.item(), input.size(0))
            top1.update(prec1.item(), input.size(0))
            top5.update(prec5.item(), input.size(0))
            _say('Epoch: [{0}][{1}/{2}]\t'
                 'Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t'
                 'Data {data_time.val:.3f} ({data_time.avg:.3f})\t'
                 'Loss {loss.val:.4f} ({loss.avg:.4f})\t'
                 'Prec@1 {top1.val:.3f} ({top1.avg:.3f})\t'
                 'Prec@5 {top5.val:.3f} ({top5.avg:.3f})\t'.format(epoch, i, len(train_loader), batch_time=batch_time, data_time=data_time,
                                                                  loss=losses, top1=top1, top5=top5), log_dir)


def validate(val_loader, net, criterion, args, device, log_dir, num_steps=None, step_size=None):
    """:329-403: PGD(num_steps_1, step_size_1) in eval mode, clean + adversarial forward, metrics averaged over ranks;
    returns the LOCAL adversarial (top-1, top-5) like the reference (:403)."""
    spec = {"shape": (3, args.crop_size, args.crop_size), "num_classes": args.num_classes}
    vargs = argparse.Namespace(**vars(args))
    vargs.method_name, vargs.attack_method, vargs.print_freq = "AT", "PGD", args.print_freq
    vargs.get = lambda k, d=None: getattr(vargs, k, d)
    return driver.validate(val_loader, net, criterion, vargs, device, args.num_steps_1 if num_steps is None else num_steps,
                           args.step_size_1 if step_size is None else step_size, log_dir, spec, local_result=True)


def build_model(args):
    """:137-146"""
    return getattr(zoo, args.arch)(num_classes=args.num_classes)


def main(argv=None, parser=None, build=build_model, dirs_of=output_dirs, eval_attack=lambda args: (args.num_steps_1, args.step_size_1),
         log_args=False):
    """`parser`, `build`, `dirs_of`, `eval_attack`, `log_args`: what AT_hfs_canny_free_imagenet_ddp.py (the same loop over the
    edge-enhanced models) changes."""
    args = (parser or make_parser()).parse_args(argv)
    torch.cuda.set_device(ddp.local_rank())
    device = torch.device("cuda", ddp.local_rank())
    ddp.setup(device)
    set_seed(ddp.rank_seed(args.seed))  # :126
    args.epochs = int(math.ceil(args.epochs / args.n_repeats))  # :129-131
    args.fgsm_step /= args.max_color_value
    args.clip_eps /= args.max_color_value
    print("=> creating model '{}'".format(args.arch))
    model = build(args).to(device)
    # SyncBatchNorm issues collectives in every forward: a captured attack graph would have to contain them, which has never run
    # on a multi-GPU node - the PGD evaluation of a multi-rank SyncBatchNorm job therefore runs eagerly unless the user insists
    # one rank: the repeat (trainer.FreeAtStep) and the evaluation attack replay captured HIP graphs, as in the other drivers
    os.environ.setdefault("EEADV_GRAPH", "0" if ddp.world() > 1 else "1")
    # :149-152: SyncBatchNorm, and instead of DistributedDataParallel one flat gradient buffer all-reduced per repeat (ddp.FlatGradSync)
    if ddp.world() > 1:
        model = ddp.convert_sync_batchnorm(model)
    # (EEADV_GRAD_SYNC=ddp: DistributedDataParallel as in the reference, :151-152)
    net, sync = ddp.make_grad_sync(model, device, find_unused_parameters=True)
    criterion = trainer.Criterion()
    optimizer = trainer.make_sgd(net.parameters(), lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
    if ddp.rank() == 0:
        print('arch:{},bs:{},lr:{},wd:{},momentum:{},epochs:{}'.format(args.arch, args.batch_size, args.lr, args.weight_decay, args.momentum, args.epochs))
        print('clip-eps:{},fgsm-step:{},n-repeats:{},world:{}'.format(int(round(args.clip_eps * 255)), int(round(args.fgsm_step * 255)),
                                                                     args.n_repeats, ddp.world()))
    dirs = dirs_of(args)
    if log_args and ddp.rank() == 0:
        with open(dirs["log"] + 'log.txt', 'a') as f:
            print(args, file=f)
    best_prec1 = 0.0
    if args.resume:  # :194-206
        if os.path.isfile(args.resume):
            print("=> loading checkpoint '{}'".format(args.resume))
            ckpt = torch.load(args.resume, map_location=device, weights_only=True)
            args.start_epoch, best_prec1 = ckpt['epoch'], ckpt['best_prec1']
            model.load_state_dict(driver.strip_module_prefix(ckpt['state_dict']))
            optimizer.load_state_dict(ckpt['optimizer'])
            print("=> loaded checkpoint '{}' (epoch {})".format(args.resume, ckpt['epoch']))
        else:
            print("=> no checkpoint found at '{}'".format(args.resume))
    B = ddp.per_rank_batch(args.batch_size)  # :208 (batch_size / nGPU)
    spec = {"shape": (3, args.crop_size, args.crop_size), "num_classes": args.num_classes}
    train_loader, val_loader = driver.make_loaders(args, spec, device, B)
    if args.evaluate:
        validate(val_loader, net, criterion, args, device, dirs["log"], *eval_attack(args))
        ddp.teardown()
        return best_prec1
    noise = torch.zeros([args.batch_size, 3, args.crop_size, args.crop_size], device=device)  # :261, global batch size on every rank
    last = args.epochs if args.max_epochs is None else min(args.epochs, args.start_epoch + args.max_epochs)
    for epoch in range(args.start_epoch, last):
        adjust_learning_rate_free(optimizer, epoch, args.lr, args.n_repeats)
        train(train_loader, net, criterion, optimizer, epoch, args, device, dirs["log"], noise, sync)
        if sync is not None:
            sync.broadcast_buffers()
        prec1, _ = validate(val_loader, net, criterion, args, device, dirs["log"], *eval_attack(args))
        is_best = prec1 > best_prec1
        best_prec1 = max(prec1, best_prec1)
        if ddp.rank() == 0:
            fname, best = checkpoint_names(args, dirs, epoch)
            save_checkpoint({'epoch': epoch + 1, 'arch': args.arch, 'state_dict': driver.with_module_prefix(model.state_dict()),
                             'best_prec1': best_prec1, 'optimizer': optimizer.state_dict()}, is_best, fname, best)
    ddp.teardown()
    return best_prec1


if __name__ == '__main__':
    main()
