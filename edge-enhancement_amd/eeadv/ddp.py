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
