"""Data parallelism for the hot path: one process per GPU, torch.distributed over RCCL (backend "nccl" on
ROCm) across the xGMI links of one node; gloo on CPU for the multi-process tests.

What is sharded: the batch dimension only: every kernel of the path is per-sample, so a
rank attacks its own shard with NO collective inside the PGD loop.  The single data-path exchange is the
gradient all-reduce of the outer training step (45.1 MB fp32 for ResNet-18/200): by default FlatGradSync
below - one flat buffer in backward order, one all-reduce per model segment issued while the remaining
segments' backward runs (the update stays a replay of captured graphs); EEADV_GRAD_SYNC=ddp selects
DistributedDataParallel (16 MB buckets, eager update).  Replaces the reference's nn.DataParallel (MNIST / Tiny
drivers, which re-broadcast the weights K+1 times per batch) and mirrors its ImageNet DDP scripts
(ImageNet/experiments_imagenet.py:56,125-129,154-161,369-384).
"""
import os

import torch
import torch.distributed as dist


def world():
    return int(os.environ.get("WORLD_SIZE", "1"))


def rank():
    return int(os.environ.get("RANK", "0"))


def local_rank():
    """LOCAL_RANK; 0 on every rank under EEADV_SHARE_GPU=1 (rehearsals and tests of the N > 1 path on a one-GPU box: the ranks
    time-share cuda:0 and exchange over gloo, see EEADV_DIST_BACKEND)"""
    if os.environ.get("EEADV_SHARE_GPU", "0") == "1":
        return 0
    return int(os.environ.get("LOCAL_RANK", "0"))


def setup(device=None, backend=None):
    """init_process_group from the torchrun environment (experiments_imagenet.py:56); no-op when world == 1."""
    if world() == 1 or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = os.environ.get("EEADV_DIST_BACKEND") or ("nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo")
    kw = {"device_id": torch.device(device)} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank(), world_size=world(), **kw)


def teardown():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def rank_seed(seed):
    """experiments_imagenet.py:61: every rank seeds with seed + rank."""
    return seed + rank()


def per_rank_batch(global_batch):
    """experiments_imagenet.py:155: batch_size / nGPU per process."""
    if global_batch % world():
        raise ValueError("batch size %d is not divisible by the %d ranks" % (global_batch, world()))
    return global_batch // world()


def shard_indices(n, r=None, w=None):
    """DistributedSampler's strided partition (padded by wrap-around so every rank gets the same count)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    per = (n + w - 1) // w
    idx = list(range(n)) + list(range(per * w - n))
    return idx[r:per * w:w]


def grad_sync_mode():
    """EEADV_GRAD_SYNC = flat (default): FlatGradSync below - graph-capturable update, segment-wise all-reduce overlapped with the
    backward.  ddp: torch's DistributedDataParallel (`wrap`) + the eager update - the stock multi-rank path, kept selectable in every
    multi-rank entry point (drivers, free-AT scripts, bench.py) as the known-good reference for FlatGradSync."""
    mode = os.environ.get("EEADV_GRAD_SYNC", "flat").lower()
    if mode not in ("flat", "ddp"):
        raise ValueError("EEADV_GRAD_SYNC must be 'flat' or 'ddp', not %r" % mode)
    return mode


def convert_sync_batchnorm(model):
    """torch.nn.SyncBatchNorm.convert_sync_batchnorm of the reference's ImageNet scripts (experiments_imagenet.py:125,
    AT_free_imagenet_ddp.py:149) onto eeadv.syncbn.SyncBatchNorm2d: the fused BatchNorm kernels around one collective per layer and
    direction.  EEADV_STOCK_SYNCBN=1: torch's own SyncBatchNorm (A/B, and the known-good path)."""
    if os.environ.get("EEADV_STOCK_SYNCBN", "0") == "1":
        return torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    from . import syncbn
    return syncbn.convert_sync_batchnorm(model)


def wrap(model, device=None, sync_bn=False, bucket_cap_mb=16, find_unused_parameters=False):
    """DistributedDataParallel around `model` (identity when world == 1).  SyncBatchNorm only where the
    reference converts (its ImageNet scripts, experiments_imagenet.py:125); the Tiny / MNIST configs keep
    per-rank BatchNorm statistics."""
    if world() == 1:
        return model
    if sync_bn:
        model = convert_sync_batchnorm(model)
    ids = [torch.device(device).index] if device is not None and torch.device(device).type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True,
                                                     find_unused_parameters=find_unused_parameters)


class FlatGradSync:
    """The gradient exchange of the data-parallel training step without DistributedDataParallel's autograd hooks, so that the
    forward + backward and the optimiser step stay CAPTURABLE as HIP graphs at N > 1 (DDP's reducer needs an eager backward: at
    N = 1 the update replays one graph, under DDP it was ~600 eager launches - a step-time gap that has nothing to do with RCCL).

      * every parameter's .grad is a view into ONE flat fp32 buffer (45.1 MB for ResNet-18/200, 102 MB for ResNet-50/1000);
        backward accumulates into the views, `zero_()` is one memset;
      * the buffer is laid out in BACKWARD order: a model that names its segments (`grad_segments()`: parameter lists, the segment
        whose gradients are complete first comes first - models.ResNet: [layer4 + fc], [layer3], [stem + layer1 + layer2]) gets one
        contiguous piece per segment, and `start(i)` issues the all-reduce of piece i as soon as the backward of segment i has been
        enqueued - asynchronous, on RCCL's stream, which waits for the compute stream's work up to that point on the device; the
        backward of the remaining segments is enqueued behind it on the compute stream and runs meanwhile (trainer._GraphedUpdate
        replays one captured graph per segment).  layer4 + fc are 76 % of ResNet-18's parameters and their gradients exist after
        ~15 % of the backward's time, so the bulk of the exchange hides behind the rest of the backward.  `finish()` makes the compute
        stream wait for all outstanding pieces (the host never blocks).  Models without segments: `chunks` even pieces >= 8 MB
        (every xGMI peer link busy), deepest parameters first;
      * the 1 / world scaling is ONE pass over the buffer, done where the caller wants it (`scale_()`, captured with the SGD step);
      * parameters and buffers are broadcast from rank 0 once at construction; buffers (BatchNorm statistics) again on request -
        DistributedDataParallel broadcasts rank 0's buffers before every forward, which amounts to "rank 0's running statistics are
        the model's"; `broadcast_buffers()` before a validation pass gives the same statistics where they are read in eval mode.
        Difference that remains: the eval-mode attack INSIDE training of ALP / TRADES on a non-SyncBN model reads rank-local running
        statistics here, rank 0's under DDP (EEADV_GRAD_SYNC=ddp selects DDP);
      * a parameter that never receives a gradient keeps .grad = None like under the reference's DDP (SGD then skips it: no weight
        decay, no momentum): the first backward is watched through post-accumulate hooks, and parameters it did not reach lose their
        view (`_settle`); their slots in the flat buffer stay zero and travel along.
    ImageNet/experiments_imagenet.py:128-129, free_imagenet/AT_free_imagenet_ddp.py:151-152 (DDP(model)), README.md:21."""

    def __init__(self, model, chunks=3, broadcast=True):
        self.model = model
        trainable = [p for p in model.parameters() if p.requires_grad]
        segs = getattr(model, "grad_segments", None)
        segs = [[p for p in seg if p.requires_grad] for seg in segs()] if segs is not None else None
        if segs is not None:
            segs = [seg for seg in segs if seg]
            seen = {id(p) for seg in segs for p in seg}
            rest = [p for p in trainable if id(p) not in seen]
            if rest or len(seen) != sum(len(seg) for seg in segs) or len(seen) > len(trainable):
                raise ValueError("grad_segments() must partition the trainable parameters (%d left over)" % len(rest))
            self.params = [p for seg in segs for p in seg]
        else:
            self.params = trainable[::-1]  # backward order, roughly: the deepest layer's gradients first in memory
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("FlatGradSync: fp32 parameters only")
        dev = self.params[0].device
        pad = lambda k: (k + 63) // 64 * 64  # every piece starts on a 256-byte boundary
        if segs is not None:
            sizes = [sum(p.numel() for p in seg) for seg in segs]
            starts, n = [], 0
            for sz in sizes:
                starts.append(n)
                n += pad(sz)
            self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
            self.pieces = [self.flat[st:st + pad(sz)] for st, sz in zip(starts, sizes)]
            self.views = []
            for st, seg in zip(starts, segs):
                off = st
                for p in seg:
                    self.views.append(self.flat[off:off + p.numel()].view_as(p))
                    off += p.numel()
            self.segmented = True
        else:
            n = sum(p.numel() for p in self.params)
            self.flat = torch.zeros(pad(n), dtype=torch.float32, device=dev)
            self.views, off = [], 0
            for p in self.params:
                self.views.append(self.flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            k = max(1, min(int(chunks), (n + (1 << 21) - 1) >> 21))  # never below ~8 MB a piece
            step = pad((n + k - 1) // k)
            self.pieces = [self.flat[i:min(i + step, pad(n))] for i in range(0, n, step)]
            self.segmented = False
        self.world = world() if dist.is_initialized() else 1
        # EEADV_FORCE_COLLECTIVES=1: issue the collectives at world size 1 as well (scripts/ddp_same_gpu.py with RCCL on a
        # one-GPU box: communicator set-up, the async all-reduce between the captured graphs, the watchdog next to a capture)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("EEADV_FORCE_COLLECTIVES", "0") == "1")
        self._works = []
        self._live = [True] * len(self.params)
        self._touched, self._hooks = set(), []
        for i, p in enumerate(self.params):
            self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, i=i: self._touched.add(i)))
        self.attach()
        if broadcast and self.active:
            self.broadcast_parameters()
            self.broadcast_buffers()

    def describe(self):
        return ("one piece per model segment (%s MB), each all-reduced on RCCL's stream while the backward of the next segment runs; "
                "SGD graph after the last" % " + ".join("%.1f" % (p.numel() * 4 / 1e6) for p in self.pieces)) if self.segmented else \
            "all-reduced between the captured backward and the captured SGD step"

    def _settle(self):
        """After the first backward: parameters it did not reach get .grad = None for good (see the class docstring)."""
        if not self._hooks:
            return
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for i, p in enumerate(self.params):
            if i not in self._touched:
                self._live[i] = False
                if p.grad is self.views[i]:
                    p.grad = None

    def attach(self):
        """(Re-)install the views as the parameters' .grad: anything that set them to None (optimizer.zero_grad(), the .loss()
        methods of ALP / TRADES, attacks.py:265-266) is undone; gradient VALUES are not touched."""
        for p, v, live in zip(self.params, self.views, self._live):
            if live and p.grad is not v:
                p.grad = v

    def zero_(self):
        self.attach()
        self.flat.zero_()

    def start(self, piece=None):
        """Issue the asynchronous all-reduce of one piece (segment index) or of all of them."""
        if not self.active:
            return
        todo = self.pieces if piece is None else [self.pieces[piece]]
        self._works += [dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True) for t in todo]

    def _check_detached(self):
        """a parameter the FIRST backward did not reach lost its view for good; if a later backward reaches it after all (a branch
        switch, an unfrozen layer) autograd gives it a private .grad that no all-reduce sees - the replicas would drift apart silently."""
        if all(self._live):
            return
        for p, live in zip(self.params, self._live):
            if not live and p.grad is not None:
                raise RuntimeError("ddp.FlatGradSync: a parameter that received no gradient in the first backward pass has one now; its gradient lies "
                                   "outside the exchanged buffer.  Build a new FlatGradSync after changing which parameters train (or use EEADV_GRAD_SYNC=ddp "
                                   "with find_unused_parameters)")

    def finish(self):
        """The current stream waits (on the device) for every piece issued since the last finish()."""
        self._settle()  # the first whole backward has been through autograd by now
        self._check_detached()
        works, self._works = self._works, []
        for w in works:
            w.wait()

    def all_reduce_(self):
        self.start()
        self.finish()

    def scale_(self):
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)

    def broadcast_parameters(self):
        with torch.no_grad():
            buf = torch.cat([p.detach().reshape(-1) for p in self.params])
            dist.broadcast(buf, 0)
            off = 0
            for p in self.params:
                p.copy_(buf[off:off + p.numel()].view_as(p))
                off += p.numel()

    def broadcast_buffers(self):
        if not self.active:
            return
        with torch.no_grad():
            for b in self.model.buffers():
                dist.broadcast(b, 0)


def make_grad_sync(model, device=None, find_unused_parameters=False):
    """(model to call, FlatGradSync or None) of a multi-rank run per EEADV_GRAD_SYNC (grad_sync_mode); (model, None) at world 1."""
    if world() == 1:
        return model, None
    if grad_sync_mode() == "ddp":
        return wrap(model, device, find_unused_parameters=find_unused_parameters), None
    return model, FlatGradSync(model)


def gather_mean(*scalars):
    """experiments_imagenet.py:369-384: all_gather each 1-element metric, then average over ranks."""
    if world() == 1:
        return [float(s) for s in scalars]
    t = torch.tensor([float(s) for s in scalars], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    bufs = [torch.zeros_like(t) for _ in range(world())]
    dist.all_gather(bufs, t)
    return torch.stack(bufs).mean(0).tolist()


def max_over_ranks(value, device=None):
    if world() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
