"""SyncBatchNorm for the data-parallel ImageNet scripts (reference: `torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)` at
ImageNet/experiments_imagenet.py:125 and ImageNet/free_imagenet/AT_free_imagenet_ddp.py:149) on the hand-written BatchNorm kernels.

The reference's SyncBatchNorm runs, per layer, a statistics kernel, an all_gather of (mean, invstd, count), an element-wise kernel,
then ReLU / residual add as launches of their own; backward: a reduce kernel, an all_reduce of (sum_dy, sum_dy_xmu), an element-wise
kernel.  Here the local halves are ee_bn.hip's split kernels with the residual add and the ReLU fused (ee_syncbn_*_f32) and the
exchange is ONE small collective per layer and direction:

    forward   moments [C, 3] = this rank's (mean, M2, count)      -> all_gather_into_tensor -> [W, C, 3]
              every rank merges the W entries in rank order (Chan's update: the same bits everywhere) -> y, saved statistics, running stats
    backward  sums [C, 2] = this rank's (sum dz, sum dz * xhat)   -> all_reduce(SUM)
              dx from the global sums and the global count; dgamma / dbeta = the LOCAL sums (the gradient exchange of the training step
              averages them with every other parameter gradient, as with torch's SyncBatchNorm under DistributedDataParallel)

State-dict keys, constructor arguments and the eval-mode behaviour (running statistics, no collective) are nn.SyncBatchNorm's.
With one rank (or no process group) the exchange is the identity and the result is that of BatchNorm2d on the same batch.
CPU tensors (the gloo tests, `runtime.allow_cpu_plumbing`) go through a torch restatement of the same formulas.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import ops, runtime


def _group_size(group):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _forced():
    """EEADV_FORCE_COLLECTIVES=1 with a process group of ONE rank: the collectives are issued anyway (the rehearsal of the N > 1 path on a
    one-GPU box: scripts/freeat_graph_collectives.py captures them into the repeat's graph over RCCL)"""
    return os.environ.get("EEADV_FORCE_COLLECTIVES", "0") == "1" and dist.is_available() and dist.is_initialized()


def _all_gather(t, group):
    """[...] -> [W, ...] in rank order"""
    W = _group_size(group)
    if W == 1 and not _forced():
        return t.unsqueeze(0)
    if t.is_cuda and dist.get_backend(group) == "gloo":  # two ranks time-sharing one GPU in the tests: stage through the host
        host = [torch.empty(t.shape, dtype=t.dtype) for _ in range(W)]
        dist.all_gather(host, t.cpu(), group=group)
        return torch.stack(host).to(t.device)
    out = torch.empty((W * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)  # concatenated along dim 0, rank by rank
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out.view((W,) + tuple(t.shape))


def _all_reduce_sum(t, group):
    if _group_size(group) == 1 and not _forced():
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        return h.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


# ---- torch restatement of the kernels' formulas (CPU plumbing only) -----------------------------------------------------------------
def _host_stats(x):
    dims = [0] + list(range(2, x.dim()))
    n = x.numel() // x.shape[1]
    mean = x.mean(dims)
    m2 = ((x - mean.view(1, -1, *([1] * (x.dim() - 2)))) ** 2).sum(dims)
    return torch.stack([mean, m2, torch.full_like(mean, float(n))], 1)


def merge_moments(all_moments):
    """Chan's update over the ranks, in rank order: [W, C, 3] -> (mean, M2, n), each [C]"""
    mean, m2, n = all_moments[0, :, 0].clone(), all_moments[0, :, 1].clone(), all_moments[0, :, 2].clone()
    for r in range(1, all_moments.shape[0]):
        bm, b2, bn_ = all_moments[r, :, 0], all_moments[r, :, 1], all_moments[r, :, 2]
        tot = n + bn_
        d = bm - mean
        mean = mean + d * (bn_ / tot)
        m2 = (m2 + b2) + d * d * (n * bn_ / tot)
        n = tot
    return mean, m2, n


class SyncBnActFn(torch.autograd.Function):
    """[relu]( sync_batch_norm(x) [+ residual] ), training mode: two launches + one collective each way."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu, group):
        shape = (1, -1) + (1,) * (x.dim() - 2)
        if x.is_cuda:
            all_m = _all_gather(ops.syncbn_stats(x), group)
            y, sm, si = ops.syncbn_apply(x, residual, gamma, beta, all_m, running_mean, running_var, momentum, eps, relu)
        else:
            all_m = _all_gather(_host_stats(x), group)
            mean, m2, n = merge_moments(all_m)
            var = m2 / n
            sm, si = mean, 1.0 / torch.sqrt(var + eps)
            if running_mean is not None:
                with torch.no_grad():
                    running_mean.mul_(1 - momentum).add_(momentum * mean)
                    running_var.mul_(1 - momentum).add_(momentum * var * (n / (n - 1.0)))
            y = (x - sm.view(shape)) * (si * gamma).view(shape) + beta.view(shape)
            if residual is not None:
                y = y + residual
            if relu:
                y = F.relu(y)
        # the global element count per channel is host arithmetic: every rank holds the same per-rank batch shape in these drivers
        # (DistributedSampler pads, experiments_imagenet.py:154-161); ranks with different shapes exchange their counts once per call
        W = _group_size(group)
        local_n = x.numel() // x.shape[1]
        if _EQUAL_SHARDS and W > 1:
            _check_equal_counts(all_m, x, group)
        ctx.n_global = float(local_n * W) if _EQUAL_SHARDS else float(all_m[:, 0, 2].sum().item())
        ctx.save_for_backward(x, y if relu else None, gamma, beta, sm, si)
        ctx.cfg = (relu, residual is not None, group)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, beta, sm, si = ctx.saved_tensors
        relu, has_res, group = ctx.cfg
        need = ctx.needs_input_grad
        from . import functional as EF
        want_params = (need[2] or need[3]) and not EF._INPUT_GRAD_ONLY
        dy = dy.contiguous()
        if x.is_cuda:
            local = ops.syncbn_bwd_sums(dy, None, y, x, gamma, beta, sm, si, relu)
            dparams = local.clone() if want_params else None
            total = _all_reduce_sum(local, group)
            dx, dres = ops.syncbn_bwd_apply(dy, None, y, x, gamma, beta, sm, si, total, ctx.n_global, relu, need[0], has_res and need[1] and relu)
            if has_res and need[1] and not relu:
                dres = dy
            dg = dparams[:, 1].contiguous() if want_params and need[2] else None
            db = dparams[:, 0].contiguous() if want_params and need[3] else None
            return dx, dres, dg, db, None, None, None, None, None, None
        shape = (1, -1) + (1,) * (x.dim() - 2)
        dims = [0] + list(range(2, x.dim()))
        dz = dy * (y > 0) if relu else dy
        xhat = (x - sm.view(shape)) * si.view(shape)
        local = torch.stack([dz.sum(dims), (dz * xhat).sum(dims)], 1)
        total = _all_reduce_sum(local.clone(), group)
        m1, m2 = (total[:, 0] / ctx.n_global).view(shape), (total[:, 1] / ctx.n_global).view(shape)
        dx = (gamma * si).view(shape) * ((dz - m1) - xhat * m2) if need[0] else None
        return dx, (dz if has_res and need[1] else None), (local[:, 1] if want_params and need[2] else None), (local[:, 0] if want_params and need[3] else None), None, None, None, None, None, None


_COUNTS_CHECKED = set()


def _check_equal_counts(all_m, x, group):
    """the backward takes n_global = local count * W.  The forward merge already holds every rank's count on the device
    (all_m[:, 0, 2]): compared ONCE per (tensor shape, group) - one host read, never inside a graph capture - and a difference is an error
    instead of a silently mis-scaled input gradient (set eeadv.syncbn._EQUAL_SHARDS = False for samplers that do not pad)."""
    key = (tuple(x.shape), id(group))
    if key in _COUNTS_CHECKED or (x.is_cuda and torch.cuda.is_current_stream_capturing()):
        return
    _COUNTS_CHECKED.add(key)
    counts = all_m[:, 0, 2]
    if not bool((counts == counts[0]).all().item()):
        raise RuntimeError("eeadv.syncbn: the ranks hold different per-rank batch sizes %s; set eeadv.syncbn._EQUAL_SHARDS = False (the global count "
                           "is then read from the exchanged moments every step)" % counts.tolist())


_EQUAL_SHARDS = True  # every rank's batch has the same shape (what the reference's DistributedSampler guarantees); False: exchange the counts


class SyncBatchNorm2d(nn.SyncBatchNorm):
    """nn.SyncBatchNorm (same constructor, parameters, buffers, state_dict) whose training forward / backward run on ee_bn.hip around
    one collective each way; anything the kernels do not take (non-affine, no running statistics, H*W % 4 != 0, other dtypes) and the
    eval mode go to the parent class."""

    def _fast(self, x):
        return (self.training and self.affine and self.track_running_stats and self.momentum is not None and x.dtype == torch.float32 and x.dim() == 4
                and x.is_contiguous() and (ops.syncbn_supported(x) if x.is_cuda else runtime.cpu_plumbing_allowed()))

    def forward(self, input):
        return sync_bn_act(self, input, relu=False)


def sync_bn_act(bn, x, residual=None, relu=False):
    """[relu]( bn(x) [+ residual] ) for a SyncBatchNorm2d: fused on the kernels in training mode, the parent's path otherwise."""
    if isinstance(bn, SyncBatchNorm2d) and bn._fast(x) and (residual is None or (residual.is_contiguous() and residual.dtype == torch.float32)):
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        return SyncBnActFn.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, relu, bn.process_group)
    # everything the kernels do not take: torch's own SyncBatchNorm forward (one rank, eval mode, momentum = None with its cumulative average,
    # track_running_stats = False, num_batches_tracked - ADVICE r3); it refuses host tensors when ranks would have to exchange statistics
    out = nn.SyncBatchNorm.forward(bn, x)
    if residual is not None:
        out = out + residual
    return F.relu(out) if relu else out


def convert_sync_batchnorm(module, process_group=None):
    """torch.nn.SyncBatchNorm.convert_sync_batchnorm for this package's models: every BatchNorm (eeadv.models.BatchNorm2d included) becomes a
    SyncBatchNorm2d carrying the same parameters and buffers; state_dict keys do not change."""
    out = module
    if isinstance(module, nn.modules.batchnorm._BatchNorm) and not isinstance(module, nn.SyncBatchNorm):
        out = SyncBatchNorm2d(module.num_features, module.eps, module.momentum, module.affine, module.track_running_stats, process_group)
        if module.affine:
            with torch.no_grad():
                out.weight = module.weight
                out.bias = module.bias
        out.running_mean, out.running_var, out.num_batches_tracked = module.running_mean, module.running_var, module.num_batches_tracked
        out.training = module.training
    for name, child in module.named_children():
        out.add_module(name, convert_sync_batchnorm(child, process_group))
    return out
