"""Shared machinery of the experiments_*.py drivers: the reference has three near-identical
400-line scripts (MNIST/experiments_mnist.py, Tiny_ImageNet/experiments_tinyimagenet.py,
ImageNet/experiments_imagenet.py); here each dataset script only declares what differs and calls `run`.

Kept surface: CLI flags, flat YAML keys merged under the CLI, `method_name` / `arch`
strings, the SGD + step-LR recipe, the checkpoint dict keys and file names, the output directory layout and the
exact log.txt line formats that utils/read_log.py parses.  Fixed while keeping the surface: missing
`type_canny` / `step_size_3` keys are tolerated, `--evaluate` runs the PGD evaluations instead of calling an
undefined function, DataParallel is replaced by one process per GPU (eeadv.ddp).

Data: the benchmark and the tests use seeded synthetic NCHW batches generated on the device (`--data synthetic`,
optionally `synthetic:<n_train_batches>:<n_val_batches>`); the torchvision loaders are out of scope.
"""
import argparse
import os
import time

import torch
import torch.optim as optim

from utils.helper import AverageMeter, adjust_learning_rate, adjust_learning_rate_1, parse_config_file, save_checkpoint, set_seed

from . import ddp, runtime, trainer


def make_parser(description, default_data="synthetic", with_local_rank=False):
    """The reference's argparse block (experiments_tinyimagenet.py:28-46, experiments_imagenet.py:29-46)."""
    parser = argparse.ArgumentParser(description=description)
    parser.add_argument('--data', metavar='DIR', default=default_data, help='path to dataset, or synthetic[:train_batches[:val_batches]]')
    parser.add_argument('-c', '--config', default='configs.yml', type=str, metavar='Path', help='path to the config file (default: configs.yml)')
    parser.add_argument('--pretrained', dest='pretrained', action='store_true', help='use pre-trained model')
    parser.add_argument('--resume', default='', type=str, metavar='PATH', help='path to latest checkpoint, (default: None)')
    parser.add_argument('-e', '--evaluate', dest='evaluate', action='store_true', help='evaluate model on validation set')
    parser.add_argument('--attack_method', default='PGD', type=str, metavar='PATH', help='attack method in validation, (default: PGD)')
    parser.add_argument('--no-cuda', action='store_true', default=False, help='disables CUDA training')
    parser.add_argument('--max-epochs', type=int, default=None, help='stop after this many epochs (smoke runs)')
    parser.add_argument('--output-root', default=None, help='where checkpoint_<DS>/ is created (default: cwd, as the reference)')
    if with_local_rank:
        parser.add_argument('--local_rank', type=int, default=int(os.environ.get("LOCAL_RANK", "0")))
    return parser


class SyntheticLoader:
    """len() batches of seeded U[0,1) images and uniform labels, generated once on the device."""

    def __init__(self, n_batches, batch_size, shape, num_classes, device, seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.batches = []
        for _ in range(n_batches):
            x = torch.rand((batch_size,) + tuple(shape), generator=g)
            y = torch.randint(0, num_classes, (batch_size,), generator=g)
            self.batches.append((x.to(device), y.to(device)))

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def make_loaders(args, spec, device, batch_size):
    if not str(args.data).startswith("synthetic"):
        raise NotImplementedError(
            "only --data synthetic[:train_batches[:val_batches]] is supported: the torchvision dataset loaders of the "
            "reference (utils/data_loader.py) are outside the hot path and torchvision is not available offline")
    parts = str(args.data).split(":")
    n_train = int(parts[1]) if len(parts) > 1 else 4
    n_val = int(parts[2]) if len(parts) > 2 else 2
    seed = ddp.rank_seed(args.seed)
    return (SyntheticLoader(n_train, batch_size, spec["shape"], spec["num_classes"], device, 1000 + seed),
            SyntheticLoader(n_val, batch_size, spec["shape"], spec["num_classes"], device, 2000 + seed))


def output_dirs(args, spec):
    """experiments_tinyimagenet.py:146-159: cwd/checkpoint_<DS>/<method>/<arch>/<type_canny>-bs..-lr..-momentum..-wd..-seed../"""
    root = args.output_root or os.getcwd()
    d = (root + '/' + spec["ckpt_dir"] + '/' + str(args.method_name) + '/' + str(args.arch) + '/' + str(args.get("type_canny")) +
         '-bs' + str(args.batch_size) + '-lr' + str(args.lr) + '-momentum' + str(args.momentum) + '-wd' + str(args.weight_decay) +
         '-seed' + str(args.seed) + '/')
    dirs = {"root": d, "model": d + 'model_pth/', "best": d + 'best_model_pth/', "log": d + 'log/'}
    if ddp.rank() == 0:
        for k in ("log", "model", "best"):
            os.makedirs(dirs[k], exist_ok=True)
    return dirs


def checkpoint_names(args, dirs, epoch):
    """experiments_tinyimagenet.py:200-211 (the best-model name really lacks the '_' before 'r')."""
    tail = ('_canny_sigma' + str(args.sigma) + '_alpha' + str(args.alpha) + '-bs' + str(args.batch_size) + '-lr_' + str(args.lr) +
            '-w' + str(args.w) + '-gf' + str(args.gf) + '-l' + str(args.low) + '-h' + str(args.high))
    head = 'at_numstep' + str(args.num_steps_1) + '_epsilon' + str(int(args.epsilon * 255))
    return (dirs["model"] + head + '_r' + str(args.r) + tail + '_' + str(epoch) + '.pth', dirs["best"] + head + 'r' + str(args.r) + tail + '.pth')


def _log(line, log_dir):
    if ddp.rank() != 0:
        return
    print(line)
    with open(log_dir + 'log.txt', 'a') as f:
        print(line, file=f)


class _DeviceMeters:
    """AverageMeter semantics (val = last batch, avg = running mean) kept on the device: the reference calls
    .item() on every batch (experiments_tinyimagenet.py:299), here the host only syncs when it prints."""

    def __init__(self, n, device):
        self.sum = torch.zeros(n, dtype=torch.float64, device=device)
        self.last = torch.zeros(n, dtype=torch.float64, device=device)
        self.count = 0

    def update(self, values, n):
        self.last = torch.stack([v.reshape(()).to(torch.float64) for v in values])
        self.sum += self.last * n
        self.count += n

    def read(self):
        last, avg = self.last.tolist(), (self.sum / max(self.count, 1)).tolist()
        out = []
        for v, a in zip(last, avg):
            m = AverageMeter()
            m.val, m.avg = v, a
            out.append(m)
        return out


def train(train_loader, model, criterion, optimizer, epoch, args, device, log_dir, spec, sync=None):
    """experiments_tinyimagenet.py:215-323."""
    batch_time, data_time = AverageMeter(), AverageMeter()
    meters = _DeviceMeters(3, device)
    model.train()
    avmixup = None
    if args.method_name in ('AVmixup', 'tarAVmixup'):
        avmixup = trainer.A.AVmixup(args, gamma=2.0, lambda1=1.0, lambda2=0.1, step_size=args.step_size_1, num_steps=args.num_steps_1,
                                    num_classes=spec["num_classes"], device=device)
    add_square = None
    if "pre_square" in args.method_name:
        from utils.core import Add_Square
        add_square = Add_Square(channels=spec["shape"][0], size=args.cize, epsilon=args.epsilon, n_queries=args.n_queries)
    end = time.time()
    for i, (input, target) in enumerate(train_loader):
        target, input = target.to(device), input.to(device)
        data_time.update(time.time() - end)
        if add_square is not None:
            input = add_square(input).detach()
        loss, output = trainer.train_batch(model, criterion, optimizer, args, input, target, device, avmixup, sync=sync)
        prec1, prec5 = trainer.accuracy(output, target, topk=(1, min(5, spec["num_classes"])))
        if spec.get("mnist_top5_quirk"):
            prec5 = prec1  # MNIST/experiments_mnist.py:246 logs prec1 as top5 (visible in the shipped MNIST log)
        meters.update([loss, prec1, prec5], input.size(0))
        batch_time.update(time.time() - end)
        end = time.time()
        if i % args.print_freq == 0:
            losses, top1, top5 = meters.read()
            _log('Epoch: [{0}][{1}/{2}]\t'
                 'Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t'
                 'Data {data_time.val:.3f} ({data_time.avg:.3f})\t'
                 'Loss {loss.val:.4f} ({loss.avg:.4f})\t'
                 'Prec@1 {top1.val:.3f} ({top1.avg:.3f})\t'
                 'Prec@5 {top5.val:.3f} ({top5.avg:.3f})\t'.format(epoch, i, len(train_loader), batch_time=batch_time, data_time=data_time,
                                                                  loss=losses, top1=top1, top5=top5), log_dir)


def strip_module_prefix(state):
    """DataParallel / DDP checkpoints (`module.`-prefixed keys, what the reference's MNIST / Tiny / free-AT scripts save) -> bare keys."""
    return {k[len("module."):] if k.startswith("module.") else k: v for k, v in state.items()}


def with_module_prefix(state):
    """Bare keys -> the `module.`-prefixed keys of a wrapped model's state_dict(): the reference's MNIST and Tiny drivers save
    nn.DataParallel(model).state_dict() (experiments_tinyimagenet.py:110,196) and its free-AT script the DDP model's
    (AT_free_imagenet_ddp.py:236); their --resume loads into the wrapped model and needs these names.  The ImageNet driver
    saves model.module.state_dict() (experiments_imagenet.py:206): bare keys."""
    return {("module." + k if not k.startswith("module.") else k): v for k, v in state.items()}


def validate(val_loader, model, criterion, args, device, num_steps, step_size, log_dir, spec, local_result=False):
    """experiments_tinyimagenet.py:326-432.  Returns (adv top-1, adv top-5) averaged over ranks (the free-AT script returns
    its rank-local averages, AT_free_imagenet_ddp.py:403: local_result=True)."""
    batch_time = AverageMeter()
    meters = _DeviceMeters(6, device)
    model.eval()
    add_square = None
    if "pre_square" in args.method_name:
        from utils.core import Add_Square
        add_square = Add_Square(channels=spec["shape"][0], size=args.cize, epsilon=args.epsilon, n_queries=args.n_queries)
    end = time.time()
    for i, (input, target) in enumerate(val_loader):
        target, input = target.to(device), input.to(device)
        if add_square is not None:
            input = add_square(input).detach()
        vals = trainer.validate_batch(model, criterion, args, input, target, device, num_steps, step_size, spec["num_classes"])
        meters.update(list(vals), input.size(0))
        batch_time.update(time.time() - end)
        end = time.time()
        if i % args.print_freq == 0:
            lc, la, t1c, t5c, t1a, t5a = meters.read()
            fmt = ('{tag}: [{0}/{1}]\tTime {batch_time.val:.3f} ({batch_time.avg:.3f})\tLoss {loss.val:.4f} ({loss.avg:.4f})\t'
                   'Prec@1 {top1.val:.3f} ({top1.avg:.3f})\tPrec@5 {top5.val:.3f} ({top5.avg:.3f})')
            _log(fmt.format(i, len(val_loader), tag='Test_clean', batch_time=batch_time, loss=lc, top1=t1c, top5=t5c), log_dir)
            _log(fmt.format(i, len(val_loader), tag='Test_adv', batch_time=batch_time, loss=la, top1=t1a, top5=t5a), log_dir)
    lc, la, t1c, t5c, t1a, t5a = meters.read()
    c1, c5, a1, a5 = ddp.gather_mean(t1c.avg, t5c.avg, t1a.avg, t5a.avg)  # experiments_imagenet.py:369-384
    _log(' * Clean Prec@1 {0:.3f} Prec@5 {1:.3f}'.format(c1, c5), log_dir)
    _log(' * Adv Prec@1 {0:.3f} Prec@5 {1:.3f}'.format(a1, a5), log_dir)
    if local_result:
        _log(' * Cl Prec@1 {top1.avg:.3f} Prec@5 {top5.avg:.3f}'.format(top1=t1c, top5=t5c), log_dir)  # AT_free_imagenet_ddp.py:400-401
        _log(' * Ad Prec@1 {top1.avg:.3f} Prec@5 {top5.avg:.3f}'.format(top1=t1a, top5=t5a), log_dir)
        return t1a.avg, t5a.avg
    return a1, a5


def run(spec, build_model, argv=None):
    """main() of the reference drivers (experiments_tinyimagenet.py:50-213)."""
    parser = make_parser(spec["description"], with_local_rank=spec.get("ddp", False))
    args = parse_config_file(parser.parse_args(argv))
    for key, default in (("type_canny", None), ("step_size_3", args.get("step_size_2")), ("num_steps_3", args.get("num_steps_2")),
                         ("n_queries", 1), ("cize", spec["shape"][-1]), ("beta", 1.0)):
        if key not in args:
            args[key] = default  # keys the reference reads unconditionally but several of its YAMLs omit
    args.num_classes = spec["num_classes"]
    use_cuda = not args.no_cuda and torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(ddp.local_rank())
        device = torch.device("cuda", ddp.local_rank())
        # the attack iteration and the parameter update replay captured HIP graphs unless the user says otherwise: eagerly
        # the loop is bound by the host (one launch per ~10 us of Python + ctypes), 3-4x slower than the device can go
        # ... except under multi-rank SyncBatchNorm (the ImageNet scripts): every train-mode forward of the attack loop would put
        # RCCL all_gathers inside the captured graph, a combination that has not run on a multi-GPU node yet
        os.environ.setdefault("EEADV_GRAPH", "0" if (ddp.world() > 1 and spec.get("sync_bn", False)) else "1")
    else:
        device = torch.device("cpu")
        runtime.allow_cpu_plumbing(True)  # --no-cuda: torch-op plumbing run on the host (BASELINE config 1)
    ddp.setup(device)
    set_seed(ddp.rank_seed(args.seed))

    print("=> creating model '{}'".format(args.arch))
    model = build_model(args).to(device)
    # N > 1: the bare model + one flat gradient buffer all-reduced per step (ddp.FlatGradSync) instead of DistributedDataParallel
    # (experiments_imagenet.py:125-129): same averaged gradients, and the update stays capturable as HIP graphs
    if ddp.world() > 1 and spec.get("sync_bn", False):
        model = ddp.convert_sync_batchnorm(model)
    # (EEADV_GRAD_SYNC=ddp: DistributedDataParallel + the eager update instead - ddp.grad_sync_mode)
    net, sync = ddp.make_grad_sync(model, device if use_cuda else None)
    if use_cuda:
        torch.backends.cudnn.benchmark = False  # experiments_tinyimagenet.py:111-112
        torch.backends.cudnn.deterministic = True
    optimizer = trainer.make_sgd(net.parameters(), lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
    criterion = trainer.make_criterion(args)
    best_prec1 = 0.0
    if args.resume:
        if os.path.isfile(args.resume):
            print("=> loading checkpoint '{}'".format(args.resume))
            ckpt = torch.load(args.resume, map_location=device, weights_only=True)
            args.start_epoch, best_prec1 = ckpt['epoch'], ckpt['best_prec1']
            missing = model.load_state_dict(strip_module_prefix(ckpt['state_dict']), strict=False)  # reference EE checkpoints carry dead sobel.* / u2netp.* keys
            print("=> loaded checkpoint '{}' (epoch {}); ignored keys: {}".format(args.resume, ckpt['epoch'], len(missing.unexpected_keys)))
            optimizer.load_state_dict(ckpt['optimizer'])
        else:
            print("=> no checkpoint found at '{}'".format(args.resume))

    batch = ddp.per_rank_batch(args.batch_size) if spec.get("ddp", False) else args.batch_size
    train_loader, val_loader = make_loaders(args, spec, device, batch)
    dirs = output_dirs(args, spec)
    print("Output dir:" + dirs["root"])
    schedule = spec.get("lr_schedule", "half_three_quarters")

    if args.evaluate:
        if sync is not None:
            sync.broadcast_buffers()
        for k, s in ((args.num_steps_1, args.step_size_1), (args.num_steps_2, args.step_size_2), (args.num_steps_3, args.step_size_3)):
            print("=> evaluate.tar_num_step:{},step_size:{}".format(k, s))
            validate(val_loader, net, criterion, args, device, k, s, dirs["log"], spec)
        ddp.teardown()
        return

    last_epoch = args.epochs if args.max_epochs is None else min(args.epochs, args.start_epoch + args.max_epochs)
    for epoch in range(args.start_epoch, last_epoch):
        if schedule == "step30":
            adjust_learning_rate(optimizer, epoch, args.lr)  # experiments_imagenet.py
        else:
            adjust_learning_rate_1(optimizer, epoch, args.lr, args.epochs)
        train(train_loader, net, criterion, optimizer, epoch, args, device, dirs["log"], spec, sync)
        if sync is not None:
            sync.broadcast_buffers()  # rank 0's BatchNorm statistics are the model's, as under DistributedDataParallel
        prec1, _ = validate(val_loader, net, criterion, args, device, args.num_steps_1, args.step_size_1, dirs["log"], spec)
        is_best = prec1 > best_prec1
        best_prec1 = max(prec1, best_prec1)
        if ddp.rank() == 0:
            fname, best = checkpoint_names(args, dirs, epoch)
            state = with_module_prefix(model.state_dict()) if spec.get("ckpt_module_prefix", False) else model.state_dict()
            save_checkpoint({'epoch': epoch + 1, 'arch': args.arch, 'state_dict': state, 'best_prec1': best_prec1,
                             'optimizer': optimizer.state_dict()}, is_best, fname, best)
    ddp.teardown()
    return best_prec1
