"""The per-batch body of the reference drivers' train() / validate() loops, shared by the drivers and bench.py.

train_batch follows Tiny_ImageNet/experiments_tinyimagenet.py:245-306 (MNIST twin: MNIST/experiments_mnist.py:
215-251): `method_name` selects the attack that builds data_adv and the loss on top of it, then
zero_grad / backward / optimizer.step.  Everything stays on the device; nothing calls .item() here (the
reference syncs every batch at :299 - the drivers sync only when they print).
"""
import os
import weakref

import torch
import torch.nn as nn

from eeadv import runtime
from utils import attacks as A
from utils.helper import accuracy


class Criterion:
    """nn.CrossEntropyLoss() of the drivers (experiments_tinyimagenet.py:127) on the HIP loss kernel for ROCm logits."""

    def __init__(self):
        self._host = nn.CrossEntropyLoss()

    def __call__(self, output, target):
        if output.is_cuda:
            from eeadv import functional as EF
            return EF.cross_entropy(output, target, "mean")
        return self._host(output, target)


_FUSED_SGD = True  # False: torch's default (foreach) SGD, five multi-tensor launches per step


class _FusedSGD(torch.optim.SGD):
    """torch.optim.SGD(fused=True) that tells the weight-derived caches about its update.  torch's fused kernel (`_fused_sgd_`) moves the
    parameters WITHOUT advancing their version counters (measured: `_version` 0 -> 0 across a fused step, 0 -> 1 across the foreach one),
    and functional._dense_weight decides by version counter whether the rearranged / Winograd-transformed copy of a filter is current:
    after an eager fused step the hand-written convolutions would keep multiplying by the OLD filters (found in round 3 by comparing
    two models with equal parameters).  step() therefore marks this optimiser's parameters' cache entries stale; inside a captured
    update the graph rebuilds them itself (functional.rebuild_dense_weights) and replays mark nothing."""

    def step(self, closure=None):
        out = super().step(closure)
        from eeadv.functional import invalidate_params
        ids = getattr(self, "_param_ids", None)
        if ids is None:
            ids = self._param_ids = {id(p) for g in self.param_groups for p in g["params"]}
        invalidate_params(ids)
        return out


def make_sgd(params, lr, momentum=0.0, weight_decay=0.0):
    """optim.SGD(params, lr, momentum, weight_decay) of the drivers (experiments_tinyimagenet.py:128-129).  On the device the update runs as
    torch's FUSED implementation - weight decay, momentum and the parameter update of all tensors in one launch instead of five
    multi-tensor launches (130 us -> 45 us per step at ResNet-18's 11 M parameters); the same arithmetic, the same state_dict."""
    params = list(params)
    if _FUSED_SGD and params and all(p.is_cuda and p.dtype == torch.float32 for p in params):
        try:
            return _FusedSGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay, fused=True)
        except (TypeError, RuntimeError):
            pass
    return torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay)


def make_criterion(args):
    """experiments_tinyimagenet.py:120-127."""
    n_class = getattr(args, "num_classes", 200)
    if args.method_name == 'ALP':
        return A.ALP(args.step_size_1, args.epsilon, args.num_steps_1, args.beta)
    if args.method_name == 'tarALP':
        return A.targeted_ALP(args.step_size_1, args.epsilon, args.num_steps_1, args.beta, n_class)
    if args.method_name == 'TRADES':
        return A.Trades(args.step_size_1, args.epsilon, args.num_steps_1, args.beta)
    return Criterion()


def attack_for_training(model, criterion, args, input, target, device, avmixup=None):
    """experiments_tinyimagenet.py:245-282: returns (data_adv, preds_or_None, new_target_or_None)."""
    m = args.method_name
    n_class = getattr(args, "num_classes", 200)
    if m == 'ST':
        return input, None, None
    if m == 'ALP':
        preds = model(input)
        return criterion.PGD_Linf(model, input, target), preds, None
    if m == 'tarALP':
        preds = model(input)
        return criterion.tarPGD_Linf(model, input, target, device), preds, None
    if m == 'TRADES':
        preds = model(input)
        return criterion.PGD_Linf(model, input, preds), preds, None
    if m in ('AVmixup', 'tarAVmixup'):
        onehot = torch.eye(n_class, device=input.device)[target]
        fn = avmixup.perturb if m == 'AVmixup' else avmixup.tar_perturb
        data_adv, new_target = fn(model, input, onehot)
        return data_adv, None, new_target
    if m.startswith('tar'):  # tarAT, tarEE, tarEE_BPDA3_AT_square
        data_adv, _ = A.targeted_PGD(model, args, input, target, args.num_steps_1, args.step_size_1, n_class, device)
        return data_adv, None, None
    return A.PGD(model, args, input, target, args.num_steps_1, args.step_size_1), None, None  # AT and every EE_* method


# ---- the parameter update as ONE HIP graph ---------------------------------------------------------------------------------
# forward on data_adv, cross-entropy, zero_grad, backward and the SGD update are ~600 launches issued eagerly from Python;
# behind a captured attack loop that is where the GPU waits for the host.  With EEADV_GRAPH=1 the plain-criterion update
# (ST / AT / every EE_* and tar* method: experiments_tinyimagenet.py:283-306) of a single-process model with a torch SGD
# optimiser is captured once per (model, optimiser hyper-parameters, batch shape) and replayed; the first two updates of a
# configuration run eagerly (they create the momentum buffers and let MIOpen pick its algorithms).  A changed learning rate
# (adjust_learning_rate, once per epoch) is a new configuration: the old graph is dropped.
_UPDATES = {}
EAGER_UPDATES_BEFORE_CAPTURE = 2


def clear_update_graphs():
    _UPDATES.clear()


def _sgd_signature(optimizer):
    return tuple((float(g['lr']), float(g['momentum']), float(g['dampening']), float(g['weight_decay']), bool(g['nesterov']),
                  bool(g.get('maximize', False))) for g in optimizer.param_groups)


class _GraphedUpdate:
    """sync = None: forward + loss + backward + SGD as ONE graph.  sync = ddp.FlatGradSync (N > 1): the gradient exchange sits between
    captured graphs - the collective itself is never captured:

      model without segments:  [forward, loss, memset of the flat gradient, backward] | all-reduce | [1 / world, SGD step]
      model with segment_fns() (the ResNets: stem..layer2 | layer3 | layer4 + head), round 3:
        [forward, loss, memset, backward of the LAST segment]   -> all-reduce of its piece starts (76 % of ResNet-18's gradient bytes)
        [backward of the middle segment]                        -> its piece starts; the first piece is on the wire meanwhile
        [backward of the first segment]                         -> its piece starts
        wait for the three pieces (on the device) | [1 / world, SGD step, rebuild of the weight-derived buffers]
      The forward is cut by detaching at the segment boundaries; the backward of a segment is torch.autograd.backward from its
      outputs with the gradients the later segment left on the detached copies - the same arithmetic as one backward pass."""

    def __init__(self, model, criterion, optimizer, data_adv, target, sync=None):
        self.model, self.criterion, self.optimizer = weakref.ref(model), criterion, weakref.ref(optimizer)
        self.x = torch.empty_like(data_adv)
        self.y = torch.empty_like(target)
        self.sync = sync
        self.graph = self.graph2 = None
        self.eager_left = EAGER_UPDATES_BEFORE_CAPTURE
        self.segmented = (sync is not None and getattr(sync, "segmented", False) and hasattr(model, "segment_fns") and _SEGMENTED
                          and len(sync.pieces) == len(model.segment_fns()))
        self.seg_graphs = []

    def _fwd_bwd(self):
        model, optimizer = self.model(), self.optimizer()
        output = model(self.x)
        loss = self.criterion(output, self.y)
        if self.sync is None:
            optimizer.zero_grad(set_to_none=True)
        else:
            self.sync.zero_()
        loss.backward()
        return loss.detach(), output.detach()

    # ---- segmented form -------------------------------------------------------------------------------------------------
    def _seg_first(self):
        """forward of every segment (detached at the boundaries), loss, memset, backward of the last segment"""
        fns = self.model().segment_fns()
        h, self._cuts = self.x, []
        for fn in fns[:-1]:
            out = fn(h)
            outs = out if isinstance(out, tuple) else (out,)
            det = tuple(o.detach().requires_grad_(True) for o in outs)
            self._cuts.append((outs, det))
            h = det if isinstance(out, tuple) else det[0]
        output = fns[-1](h)
        loss = self.criterion(output, self.y)
        self.sync.zero_()
        loss.backward()
        return loss.detach(), output.detach()

    def _seg_back(self, i):
        """backward of segment i from the gradients segment i + 1 left on the detached copies of its outputs"""
        outs, det = self._cuts[i]
        pairs = [(o, d.grad) for o, d in zip(outs, det) if d.grad is not None]  # a forked output whose second copy nobody read has none
        torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])

    def _seg_body(self):
        n = len(self.sync.pieces)
        out = self._seg_first()
        self.sync.start(0)
        for k in range(1, n):
            self._seg_back(n - 1 - k)
            self.sync.start(k)
        self.sync.finish()
        self._step()
        self._cuts = []
        return out

    def _step(self):
        if self.sync is not None:
            self.sync.scale_()
        self.optimizer().step()
        if self.x.is_cuda and torch.cuda.is_current_stream_capturing():
            # a replayed update moves the weights but not their version counters: the weight-derived buffers of
            # functional.Conv3x3Map2Fn are rebuilt by the graph itself, right behind the update
            from eeadv.functional import rebuild_dense_weights
            self._rebuilt = rebuild_dense_weights(self.model())
            self._param_ids = {id(p) for p in self.model().parameters()}

    def _body(self):
        if self.segmented:
            return self._seg_body()
        out = self._fwd_bwd()
        if self.sync is not None:
            self.sync.all_reduce_()
        self._step()
        return out

    def _capture(self):
        runtime.draw_state(self.x.device)  # device-side draws inside the captured forward need their state to exist
        torch.cuda.synchronize()
        mode = runtime.capture_mode()
        self.graph = torch.cuda.CUDAGraph()
        if self.sync is None:
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                self.loss, self.output = self._body()
            return
        if self.segmented:
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                self.loss, self.output = self._seg_first()
            for k in range(1, len(self.sync.pieces)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.graph.pool(), capture_error_mode=mode):
                    self._seg_back(len(self.sync.pieces) - 1 - k)
                self.seg_graphs.append(g)
        else:
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                self.loss, self.output = self._fwd_bwd()
        self.sync.all_reduce_()  # the captured backward did not run: this reduces the warm-up's gradients, harmlessly,
        self.graph2 = torch.cuda.CUDAGraph()  # and keeps every rank's collective count equal
        with torch.cuda.graph(self.graph2, pool=self.graph.pool(), capture_error_mode=mode):
            self._step()

    def __call__(self, data_adv, target):
        self.x.copy_(data_adv)
        self.y.copy_(target)
        if self.eager_left > 0:
            self.eager_left -= 1
            return self._body()
        from eeadv.functional import refresh_dense_weights
        refresh_dense_weights()  # eager updates since the last forward (version counters moved): before capture AND replay
        if self.graph is None:
            self._capture()
        self.graph.replay()
        if self.sync is not None:
            if self.segmented:
                self.sync.start(0)
                for k, g in enumerate(self.seg_graphs, 1):
                    g.replay()
                    self.sync.start(k)
            else:
                self.sync.start()
            if PHASE_EVENTS is not None:
                PHASE_EVENTS.mark("backward")
            self.sync.finish()
            if PHASE_EVENTS is not None:
                PHASE_EVENTS.mark("all_reduce")  # what of the exchange the backward did NOT hide
            self.graph2.replay()
            if PHASE_EVENTS is not None:
                PHASE_EVENTS.mark("sgd")
        _mark_stale(self)
        return self.loss.clone(), self.output.clone()


def _mark_stale(update):
    """filter copies that did not exist when `update`'s graph was captured are not rebuilt by its replay (functional.invalidate_dense_except)"""
    from eeadv.functional import invalidate_dense_except
    invalidate_dense_except(update._param_ids, update._rebuilt)


_SEGMENTED = os.environ.get("EEADV_SEGMENTED_SYNC", "1") == "1"  # 0: one all-reduce after the whole captured backward (round 2's form)


def two_branch_backward(loss, logits_nat, logits_adv, params):
    """loss.backward() for a loss over TWO forward passes through the same parameters (Trades.loss, ALP.loss: utils/attacks.py:264-272,
    :421-429) without autograd's per-parameter accumulation: with both passes in one backward every parameter receives two gradients and
    AccumulateGrad adds the second one in a launch of its own (62 `+=` kernels per ResNet-18 step, 0.3 ms).  Here the loss is differentiated
    down to the two logits tensors, each forward pass is then backpropagated on its own into fresh .grad tensors (the adversarial pass
    first, as the engine orders them) and the two sets meet in ONE multi-tensor add: the same sums, g_adv + g_nat.  Falls back to
    loss.backward() when the two branches are not both there."""
    if logits_nat is None or logits_adv is None or not logits_nat.requires_grad or not logits_adv.requires_grad or logits_nat is logits_adv:
        loss.backward()
        return
    d_nat, d_adv = torch.autograd.grad(loss, [logits_nat, logits_adv], allow_unused=True)
    if d_nat is None or d_adv is None:
        torch.autograd.backward([t for t, d in ((logits_nat, d_nat), (logits_adv, d_adv)) if d is not None], [d for d in (d_nat, d_adv) if d is not None])
        return
    params = [p for p in params if p.requires_grad]
    for p in params:
        p.grad = None
    torch.autograd.backward([logits_adv], [d_adv])
    first = [p.grad for p in params]
    for p in params:
        p.grad = None
    torch.autograd.backward([logits_nat], [d_nat])
    both = [(a, p.grad) for p, a in zip(params, first) if a is not None and p.grad is not None]
    if both:
        torch._foreach_add_([a for a, _ in both], [b for _, b in both])
    for p, a in zip(params, first):
        if a is not None:
            p.grad = a


class _GraphedPredsUpdate:
    """TRADES / ALP / tarALP (experiments_tinyimagenet.py:250-262, :286-291): the step is  preds = model(input)  ->  attack  ->
    output = model(data_adv); loss = criterion.loss(model, preds, ...); zero_grad; backward; step  - and `preds` stays attached across
    the attack (the backward runs through that first forward too).  The attack cannot be part of a capture (it is a graph of its
    own and draws its random start eagerly), so the step is TWO captured graphs around it that share one memory pool: the first forward
    (its saved activations stay where the second graph's backward reads them), and everything after the attack.  Eager, these were ~1200
    launches from Python per step behind a replayed attack - 9 ms of host time against 18 ms on the device, close enough for the
    device to wait on the host between small kernels.  With a ddp.FlatGradSync (N > 1, round 3) the second graph ends behind the
    backward, the flat gradient is all-reduced, and a THIRD graph holds the 1 / world scaling and the SGD step."""

    def __init__(self, model, criterion, optimizer, args, input, target, device, sync=None):
        self.model, self.criterion, self.optimizer = weakref.ref(model), criterion, weakref.ref(optimizer)
        self.args, self.device, self.sync = args, device, sync
        self.x, self.y, self.adv = torch.empty_like(input), torch.empty_like(target), torch.empty_like(input)
        self.g1 = self.g2 = self.g3 = None
        self.eager_left = EAGER_UPDATES_BEFORE_CAPTURE

    def _attack(self, preds):
        model, m = self.model(), self.args.method_name
        if m == 'TRADES':
            return self.criterion.PGD_Linf(model, self.x, preds)
        if m == 'ALP':
            return self.criterion.PGD_Linf(model, self.x, self.y)
        return self.criterion.tarPGD_Linf(model, self.x, self.y, self.device)

    def _loss_backward(self, preds):
        model, optimizer = self.model(), self.optimizer()
        output = model(self.adv)
        if self.args.method_name == 'TRADES':
            loss = self.criterion.loss(model, preds, self.adv, self.y, optimizer)
        else:
            loss = self.criterion.loss(model, preds, output, self.y, optimizer)
        if self.sync is None:
            optimizer.zero_grad(set_to_none=True)
            # the loss spans two forward passes: one backward per pass and one multi-tensor add instead of 62 AccumulateGrad launches
            logits_adv = getattr(self.criterion, "last_logits_adv", None) if self.args.method_name == 'TRADES' else output
            two_branch_backward(loss, preds, logits_adv, [p for g in optimizer.param_groups for p in g['params']])
        else:
            self.sync.zero_()  # .loss() called optimizer.zero_grad() (attacks.py:265-266, :422-423): the views go back in
            loss.backward()
        if self.args.method_name == 'TRADES':
            self.criterion.last_logits_adv = None  # it holds the second forward's autograd graph (inside a capture: tensors of the graph's pool)
        return loss.detach(), output.detach()

    def _step(self):
        model, optimizer = self.model(), self.optimizer()
        if self.sync is not None:
            self.sync.scale_()
        optimizer.step()
        if self.x.is_cuda and torch.cuda.is_current_stream_capturing():
            from eeadv.functional import rebuild_dense_weights
            self._rebuilt = rebuild_dense_weights(model)
            self._param_ids = {id(p) for p in model.parameters()}

    def _after_attack(self, preds):
        out = self._loss_backward(preds)
        if self.sync is not None:
            self.sync.all_reduce_()
        self._step()
        return out

    def __call__(self, input, target):
        model = self.model()
        self.x.copy_(input)
        self.y.copy_(target)
        if self.eager_left > 0:
            self.eager_left -= 1
            preds = model(self.x)
            self.adv.copy_(self._attack(preds))
            return self._after_attack(preds)
        from eeadv.functional import refresh_dense_weights
        refresh_dense_weights()
        if self.g1 is None:
            if not model.training:
                raise RuntimeError("TRADES / ALP step: the model must be in train mode when the step starts (the .loss() methods leave it there)")
            runtime.draw_state(self.x.device)
            torch.cuda.synchronize()
            mode = runtime.capture_mode()
            self.g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g1, capture_error_mode=mode):
                self.preds = model(self.x)
            self.adv.copy_(self._attack(self.preds))  # leaves the model in eval mode, as in the eager step: the second capture starts there
            self.g2 = torch.cuda.CUDAGraph()
            if self.sync is None:
                with torch.cuda.graph(self.g2, pool=self.g1.pool(), capture_error_mode=mode):
                    self.loss, self.output = self._after_attack(self.preds)
            else:
                with torch.cuda.graph(self.g2, pool=self.g1.pool(), capture_error_mode=mode):
                    self.loss, self.output = self._loss_backward(self.preds)
                self.sync.all_reduce_()  # keeps every rank's collective count equal (the captured backward did not run)
                self.g3 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g3, pool=self.g1.pool(), capture_error_mode=mode):
                    self._step()
            # the captures recorded the kernels without running them: run the step they stand for
        self.g1.replay()
        self.adv.copy_(self._attack(self.preds))
        if PHASE_EVENTS is not None:
            PHASE_EVENTS.mark("attack")
        self.g2.replay()
        if self.sync is not None:
            if PHASE_EVENTS is not None:
                PHASE_EVENTS.mark("backward")
            self.sync.all_reduce_()
            if PHASE_EVENTS is not None:
                PHASE_EVENTS.mark("all_reduce")
            self.g3.replay()
            if PHASE_EVENTS is not None:
                PHASE_EVENTS.mark("sgd")
        model.train()  # where criterion.loss() leaves it
        _mark_stale(self)
        return self.loss.clone(), self.output.clone()


class PhaseEvents:
    """Per-rank device time of the phases of a data-parallel training step (bench.py, N > 1): HIP events on the current stream at
    the phase boundaries, read after the timed region."""

    def __init__(self):
        self.marks = []

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.marks.append(("start", e))

    def mark(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.marks.append((name, e))

    def summary(self):
        tot, cnt = {}, {}
        for (_, a), (name, b) in zip(self.marks, self.marks[1:]):
            if name == "start":
                continue
            tot[name] = tot.get(name, 0.0) + a.elapsed_time(b)
            cnt[name] = cnt.get(name, 0) + 1
        return {k: round(tot[k] / cnt[k], 4) for k in tot}


PHASE_EVENTS = None  # bench.py installs a PhaseEvents for the timed steps of an N > 1 run


def _graphable_update(model, criterion, optimizer, args, data_adv):
    from eeadv import engine
    return (engine.graphs_enabled() and data_adv.is_cuda and isinstance(criterion, Criterion) and isinstance(optimizer, torch.optim.SGD)
            and not isinstance(model, (nn.parallel.DistributedDataParallel, nn.DataParallel)) and model.training
            and args.method_name not in ('ALP', 'tarALP', 'TRADES', 'AVmixup', 'tarAVmixup'))


def _graphable_preds_update(model, criterion, optimizer, args, input, sync):
    from eeadv import engine
    return (engine.graphs_enabled() and input.is_cuda and isinstance(optimizer, torch.optim.SGD) and model.training
            and not isinstance(model, (nn.parallel.DistributedDataParallel, nn.DataParallel)) and _GRAPH_PREDS
            and args.method_name in ('TRADES', 'ALP', 'tarALP'))


_GRAPH_PREDS = True  # False: the TRADES / ALP step around the attack stays eager


def backward_and_step(loss, optimizer, sync=None):
    """zero_grad / backward / step of the drivers (experiments_tinyimagenet.py:304-306), with the gradient all-reduce of a
    data-parallel run in between when `sync` (ddp.FlatGradSync) is given."""
    if sync is None:
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        return
    sync.zero_()
    loss.backward()
    sync.all_reduce_()
    sync.scale_()
    optimizer.step()


def train_batch(model, criterion, optimizer, args, input, target, device, avmixup=None, sync=None):
    """One optimisation step; returns (loss, output) detached, both still on the device.  sync: ddp.FlatGradSync of a
    data-parallel run (the model is then NOT wrapped in DistributedDataParallel)."""
    if _graphable_preds_update(model, criterion, optimizer, args, input, sync):
        owner = (id(model), id(optimizer))
        sig = _sgd_signature(optimizer)
        slot = _UPDATES.get(owner)
        if slot is None or slot[0] != sig or slot[1]() is not model or slot[2]() is not optimizer:
            slot = (sig, weakref.ref(model), weakref.ref(optimizer), {})
            _UPDATES[owner] = slot
        key = ("preds", args.method_name, tuple(input.shape), tuple(target.shape), target.dtype, input.device.index)
        update = slot[3].get(key)
        if update is None:
            update = slot[3][key] = _GraphedPredsUpdate(model, criterion, optimizer, args, input, target, device, sync)
        return update(input.detach(), target)
    data_adv, preds, new_target = attack_for_training(model, criterion, args, input, target, device, avmixup)
    if PHASE_EVENTS is not None:
        PHASE_EVENTS.mark("attack")
    if _graphable_update(model, criterion, optimizer, args, data_adv):
        owner = (id(model), id(optimizer))
        sig = _sgd_signature(optimizer)
        slot = _UPDATES.get(owner)
        if slot is None or slot[0] != sig or slot[1]() is not model or slot[2]() is not optimizer:
            slot = (sig, weakref.ref(model), weakref.ref(optimizer), {})  # hyper-parameters changed: drop the old graphs
            _UPDATES[owner] = slot
        key = (tuple(data_adv.shape), tuple(target.shape), target.dtype, data_adv.device.index)
        update = slot[3].get(key)
        if update is None:
            update = slot[3][key] = _GraphedUpdate(model, criterion, optimizer, data_adv, target, sync)
        return update(data_adv.detach(), target)
    output = model(data_adv)
    m = args.method_name
    if m in ('ALP', 'tarALP'):
        loss = criterion.loss(model, preds, output, target, optimizer)
    elif m == 'TRADES':
        loss = criterion.loss(model, preds, data_adv, target, optimizer)
    elif m in ('AVmixup', 'tarAVmixup'):
        if output.is_cuda:
            from eeadv import functional as EF
            loss = EF.soft_cross_entropy(output, new_target.contiguous(), 1.0 / input.shape[0])
        else:
            loss = -torch.sum(nn.functional.log_softmax(output, dim=1) * new_target) / input.shape[0]
    else:
        loss = criterion(output, target)
    if sync is None and m in ('ALP', 'tarALP', 'TRADES') and output.is_cuda:
        optimizer.zero_grad()
        two_branch_backward(loss, preds, getattr(criterion, "last_logits_adv", None) if m == 'TRADES' else output,
                            [p for g in optimizer.param_groups for p in g['params']])
        optimizer.step()
    else:
        backward_and_step(loss, optimizer, sync)
    if m == 'TRADES':
        criterion.last_logits_adv = None  # (it holds the forward pass's autograd graph)
    return loss.detach(), output.detach()


def free_at_repeat(model, criterion, optimizer, input, target, noise, fgsm_step, clip_eps, return_input_grad=False, sync=None):
    """One repeat of "free" adversarial training (ImageNet/free_imagenet/AT_free_imagenet_ddp.py:287-309): a single
    forward/backward gives the weight gradient AND the input gradient; `noise` is the persistent buffer, its first
    len(input) rows are read and updated in place.  Returns (loss, output), detached, still on the device.

        in1 = clamp(input + noise[:n], 0, 1)                          ee_add_clamp_f32             (:289-290)
        loss.backward()                                               DDP's all-reduce overlaps this backward
        noise[:n] = clamp(noise[:n] + step * sign(dL/dnoise), +-eps)  ee_freeat_update_masked_f32  (:305-307)
        optimizer.step()                                                                           (:309)
    dL/dnoise of the reference is dL/din1 masked by the in-place clamp of :290; the kernel applies that mask itself."""
    from eeadv import ops
    n = input.size(0)
    x = input.contiguous()
    in1 = ops.add_clamp(x, noise[0:n], 0.0, 1.0).requires_grad_(True)
    output = model(in1)
    loss = criterion(output, target)
    if sync is None:
        optimizer.zero_grad()
    else:
        sync.zero_()
    loss.backward()
    if sync is not None:
        sync.start()  # asynchronous, on RCCL's stream; the noise update below runs on the compute stream meanwhile
    ops.freeat_update_masked_(noise, in1.grad.contiguous(), x, float(fgsm_step), float(clip_eps))
    if sync is not None:
        sync.finish()  # the compute stream waits for the reduction here (on the device; the host does not block)
        sync.scale_()
    optimizer.step()
    if return_input_grad:  # dL/din1, BEFORE the clamp mask (tests)
        return loss.detach(), output.detach(), in1.grad.detach()
    return loss.detach(), output.detach()


def awp_train_batch(model, awp_adversary, criterion, optimizer, args, input, target, epoch, device, l1=0.0):
    """One step of the AWP train loop (AWP/Tiny_imagenet/experiments_tiny_awp.py:256-286; `awp_adversary`: AdvWeightPerturb of
    AWP/Tiny_imagenet/models_tiny_awp/utils_awp.py): PGD through the hot path, then - from epoch `args.awp_warmup` on - the weight
    perturbation computed on a proxy copy is added to the model for the robust forward / backward / SGD step and removed again.
    Returns (loss, output) detached."""
    if args.method_name not in ('AT_AWP', 'EE_AT_AWP'):
        raise NotImplementedError('Wrong method name!')
    data_adv = A.PGD(model, args, input, target, args.num_steps_1, args.step_size_1)
    awp = None
    if epoch >= args.awp_warmup:
        awp = awp_adversary.calc_awp(inputs_adv=data_adv, targets=target)
        awp_adversary.perturb(awp)
    robust_output = model(data_adv)
    robust_loss = criterion(robust_output, target)
    if l1:
        for name, param in model.named_parameters():
            if 'bn' not in name and 'bias' not in name:
                robust_loss = robust_loss + l1 * param.abs().sum()
    optimizer.zero_grad()
    robust_loss.backward()
    optimizer.step()
    if awp is not None:
        awp_adversary.restore(awp)
    return robust_loss.detach(), robust_output.detach()


def graph_collectives_enabled():
    """EEADV_GRAPH_COLLECTIVES=1 and a process group on RCCL ("nccl"): collectives may sit inside a captured graph"""
    import torch.distributed as dist
    return (os.environ.get("EEADV_GRAPH_COLLECTIVES", "0") == "1" and dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl")


class FreeAtStep:
    """One batch through the `n_repeats` repeats of free adversarial training (AT_free_imagenet_ddp.py:286-309) - the "step" of
    BASELINE config 5.  With HIP graphs enabled and no collective inside the repeat (one rank: plain BatchNorm, no gradient exchange)
    ONE repeat - add_clamp, forward, cross-entropy, zero_grad, backward (weight and input gradients), the noise update, the SGD step
    and the rebuild of the weight-derived filter buffers - is captured after two eager repeats and replayed `n_repeats` times per
    batch: ResNet-50 at batch 32 is ~1100 launches per repeat, which the host cannot issue as fast as the device runs them.
    With a gradient exchange (`sync`, N > 1) or SyncBatchNorm the repeats run eagerly through free_at_repeat."""

    def __init__(self, model, criterion, optimizer, noise, fgsm_step, clip_eps, n_repeats, sync=None):
        self.model, self.criterion, self.optimizer, self.noise = model, criterion, optimizer, noise
        self.fgsm_step, self.clip_eps, self.n_repeats, self.sync = float(fgsm_step), float(clip_eps), int(n_repeats), sync
        self.graph, self.sig, self.eager_left = None, None, EAGER_UPDATES_BEFORE_CAPTURE

    def _graphable(self, x):
        from eeadv import engine
        m = self.model
        if not (engine.graphs_enabled() and x.is_cuda and isinstance(self.optimizer, torch.optim.SGD) and m.training
                and not isinstance(m, (nn.parallel.DistributedDataParallel, nn.DataParallel))):
            return False
        collectives = self.sync is not None or any(isinstance(k, nn.SyncBatchNorm) for k in m.modules())
        # round 4: the repeat WITH its collectives (SyncBatchNorm's two per layer, the gradient pieces' all-reduce) as one captured graph - opt-in
        # (EEADV_GRAPH_COLLECTIVES=1, RCCL only) until it has run on a multi-GPU node; scripts/freeat_graph_collectives.py rehearses it on one rank
        return not collectives or graph_collectives_enabled()

    def _repeat(self, x, y):
        return free_at_repeat(self.model, self.criterion, self.optimizer, x, y, self.noise, self.fgsm_step, self.clip_eps, sync=self.sync)

    def __call__(self, x, y):
        if not self._graphable(x):
            for _ in range(self.n_repeats):
                out = self._repeat(x, y)
            return out
        sig = (_sgd_signature(self.optimizer), tuple(x.shape), tuple(y.shape))
        if sig != self.sig:  # a new learning rate (adjust_learning_rate_free, once per epoch) or batch shape: a new graph
            self.sig, self.graph, self.eager_left = sig, None, max(self.eager_left, 1 if self.graph is not None else 0)
            self.x, self.y = torch.empty_like(x), torch.empty_like(y)
        self.x.copy_(x)
        self.y.copy_(y)
        done = 0
        while self.eager_left > 0 and done < self.n_repeats:
            self.eager_left -= 1
            done += 1
            out = self._repeat(self.x, self.y)
        if done == self.n_repeats:
            return out
        from eeadv.functional import rebuild_dense_weights, refresh_dense_weights
        refresh_dense_weights()
        if self.graph is None:
            runtime.draw_state(self.x.device)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode=runtime.capture_mode()):
                self.loss, self.output = self._repeat(self.x, self.y)
                self._rebuilt = rebuild_dense_weights(self.model)
                self._param_ids = {id(p) for p in self.model.parameters()}
        for _ in range(self.n_repeats - done):
            self.graph.replay()
        _mark_stale(self)
        return self.loss.clone(), self.output.clone()


def attack_for_validation(model, args, input, target, device, num_steps, step_size, n_class):
    """experiments_tinyimagenet.py:354-374."""
    targeted = "tar" in args.method_name
    if args.attack_method == 'PGD':
        if targeted:
            return A.targeted_PGD(model, args, input, target, num_steps, step_size, n_class, device)[0]
        return A.PGD(model, args, input, target, num_steps, step_size)
    if args.attack_method == 'FGSM':
        if targeted:
            off = torch.randint(low=1, high=n_class, size=target.shape).to(device)
            return A.FGSM(model, input, torch.fmod(target + off, n_class), targeted=True, step_size=step_size)
        return A.FGSM(model, input, target, targeted=False, step_size=step_size)
    if args.attack_method == 'CW':
        tl = None
        if targeted:
            off = torch.randint(low=1, high=n_class, size=target.shape).to(device)
            tl = torch.fmod(target + off, n_class)
        return A.CWLinfAttack(x=input, y=target, model=model, magnitude=args.epsilon, previous_p=None, max_eps=args.epsilon,
                              max_iters=20, target=tl, n_class=n_class, cur_device=device)[0]
    raise NotImplementedError


def validate_batch(model, criterion, args, input, target, device, num_steps, step_size, n_class):
    """experiments_tinyimagenet.py:354-397: attack, then clean and adversarial forward under no_grad.
    Returns device tensors (loss_clean, loss_adv, prec1_cle, prec5_cle, prec1_adv, prec5_adv)."""
    data_adv = attack_for_validation(model, args, input, target, device, num_steps, step_size, n_class)
    with torch.no_grad():
        output_clean = model(input)
        output_adv = model(data_adv)
        if "ALP" in args.method_name or "TRADES" in args.method_name:
            loss_clean = torch.zeros((), device=input.device)
            loss_adv = torch.zeros((), device=input.device)
        else:
            loss_clean = criterion(output_clean, target)
            loss_adv = criterion(output_adv, target)
        p1c, p5c = accuracy(output_clean.data, target, topk=(1, min(5, n_class)))
        p1a, p5a = accuracy(output_adv.data, target, topk=(1, min(5, n_class)))
    return loss_clean, loss_adv, p1c, p5c, p1a, p5a
