"""Data parallelism for the hot path: one process per GPU, torch.distributed over RCCL (backend "nccl" on
ROCm) across the xGMI links of one node; gloo on CPU for the multi-process tests.

What is sharded: the batch dimension only: every kernel of the path is per-sample, so a
rank attacks its own shard with NO collective inside the PGD loop.  The single data-path exchange is the
gradient all-reduce of the outer training step (45.1 MB fp32 for ResNet-18/200), issued by DDP in 16 MB
buckets while the backward is still running.  Replaces the reference's nn.DataParallel (MNIST / Tiny
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
    return int(os.environ.get("LOCAL_RANK", "0"))


def setup(device=None, backend=None):
    """init_process_group from the torchrun environment (experiments_imagenet.py:56); no-op when world == 1."""
    if world() == 1 or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
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


def wrap(model, device=None, sync_bn=False, bucket_cap_mb=16, find_unused_parameters=False):
    """DistributedDataParallel around `model` (identity when world == 1).  SyncBatchNorm only where the
    reference converts (its ImageNet scripts, experiments_imagenet.py:125); the Tiny / MNIST configs keep
    per-rank BatchNorm statistics."""
    if world() == 1:
        return model
    if sync_bn:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    ids = [torch.device(device).index] if device is not None and torch.device(device).type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True,
                                                     find_unused_parameters=find_unused_parameters)


class FlatGradSync:
    """The gradient exchange of the data-parallel training step without DistributedDataParallel's autograd hooks, so that the
    forward + backward and the optimiser step stay CAPTURABLE as HIP graphs at N > 1 (DDP's reducer needs an eager backward: at
    N = 1 the update replays one graph, under DDP it was ~600 eager launches - a step-time gap that has nothing to do with RCCL).

      * every parameter's .grad is a view into ONE flat fp32 buffer (45.1 MB for ResNet-18/200, 102 MB for ResNet-50/1000);
        backward accumulates into the views, `zero_()` is one memset;
      * `all_reduce_()` sums the buffer over the ranks in `chunks` contiguous pieces (async, on RCCL's stream; the caller's
        stream waits on the device, the host does not block) - pieces >= 8 MB keep every xGMI peer link busy;
        the 1 / world scaling is ONE pass over the buffer, done where the caller wants it (`scale_()`, captured with the SGD step);
      * parameters are broadcast from rank 0 once, buffers (BatchNorm statistics) on request - DistributedDataParallel
        broadcasts rank 0's buffers before every forward, which amounts to "rank 0's running statistics are the model's";
        `broadcast_buffers()` before a validation pass gives the same statistics.
    ImageNet/experiments_imagenet.py:128-129, free_imagenet/AT_free_imagenet_ddp.py:151-152 (DDP(model)), README.md:21."""

    def __init__(self, model, chunks=3, broadcast=True):
        self.model = model
        self.params = [p for p in model.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("FlatGradSync: fp32 parameters only")
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.world = world() if dist.is_initialized() else 1
        # EEADV_FORCE_COLLECTIVES=1: issue the collectives at world size 1 as well (scripts/ddp_same_gpu.py with RCCL on a
        # one-GPU box: communicator set-up, the async all-reduce between the two captured graphs, the watchdog next to a capture)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("EEADV_FORCE_COLLECTIVES", "0") == "1")
        # contiguous pieces, element counts rounded to 64 (256 B)
        k = max(1, min(int(chunks), (n + (1 << 21) - 1) >> 21))  # never below ~8 MB a piece
        step = ((n + k - 1) // k + 63) // 64 * 64
        self.pieces = [self.flat[i:min(i + step, n)] for i in range(0, n, step)]
        self.attach()
        if broadcast and self.active:
            self.broadcast_parameters()

    def attach(self):
        """(Re-)install the views as the parameters' .grad: anything that set them to None (optimizer.zero_grad(), the .loss()
        methods of ALP / TRADES, attacks.py:265-266) is undone; gradient VALUES are not touched."""
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v

    def zero_(self):
        self.attach()
        self.flat.zero_()

    def all_reduce_(self):
        if not self.active:
            return
        works = [dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True) for piece in reversed(self.pieces)]  # deepest layers first
        for w in works:
            w.wait()

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
