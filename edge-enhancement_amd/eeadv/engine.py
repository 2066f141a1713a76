"""The attack inner loop on the device (the K-step loop of utils/attacks.py:19-27 and its nine copies).

One PGD step = model forward, loss-gradient kernel on the logits, model backward to the input
(torch.autograd.grad with grad_outputs - the scalar loss itself is never formed inside the loop), then
ONE in-place update kernel on persistent buffers (x, x0): zero temporaries, where the reference allocates
six per step.  The K steps may run from a HIP graph captured on the first call for a given (model, shape,
loss, mode) - the loop body is launch-bound at B = 50..100.

Semantics kept from the reference: the model's train/eval mode is NOT touched here (PGD() runs in whatever
mode the caller left, BatchNorm statistics and dropout included); callers are entered under
torch.enable_grad() so an outer no_grad() does not matter; the result is a new detached tensor.
"""
import os
import weakref

import torch

from . import ops, runtime
from .functional import attack_forward, input_grad_only, refresh_dense_weights

CE_SUM, CE_MEAN, KL, SOFTCE = "ce_sum", "ce_mean", "kl", "softce"
# the input-gradient pass runs on the calling thread: handing it to autograd's device thread puts cross-thread stream
# synchronisation into the captured graph (a 40 us idle gap in front of the first backward kernel of every iteration)
_MT_BACKWARD = False


class LossSpec:
    """Which d(loss)/d(logits) the loop needs.  payload: labels (CE), natural logits (KL), float64 soft
    targets (SOFTCE).  attacks.py:23 (sum CE), :255 (mean CE), :412 (KL batchmean), :462-463 (soft CE)."""

    def __init__(self, kind, payload):
        self.kind, self.payload = kind, payload

    def dlogits(self, logits):
        z = logits.detach()
        if self.kind == CE_SUM:
            return ops.ce(z, self.payload, "sum", 0.0, False, True)[1]
        if self.kind == CE_MEAN:
            return ops.ce(z, self.payload, "mean", 0.0, False, True)[1]
        if self.kind == KL:
            return ops.kl_batchmean(z, self.payload, False, True, False)[1]
        if self.kind == SOFTCE:
            return ops.softce(z, self.payload, 1.0, False, True)[1].to(torch.float32)
        raise ValueError(self.kind)


_FC_HEAD = True  # models that offer head_grad (Net_2) get head + loss + way back as one launch; False: separate launches


def _body_input_grad(model, x_in, spec, through_body):
    """d loss / d x_in through model.body (through_body) or the whole model: logits -> loss gradient -> autograd.  Models that expose
    `body_pre` / `head_grad` / `head_from_pre` and whose head_grad answers (models.Net_2: from fc1's output on) get the rest of the classifier,
    the cross-entropy and the way back as ONE launch (ops.fc_ce_grad) instead of five; the ResNets': the head's forward, then the loss gradient inside the
    head's backward launch (ops.ce_pool_linear_bwd) - two launches instead of three, the same bits."""
    pre = getattr(model, "body_pre", None) if _FC_HEAD and (through_body or not hasattr(model, "front_chain")) else None
    if pre is not None and spec.kind in (CE_SUM, CE_MEAN) and x_in.is_cuda and hasattr(model, "head_grad"):
        with torch.enable_grad(), attack_forward():
            z = pre(x_in)
        dz = model.head_grad(z.detach(), spec.payload, "mean" if spec.kind == CE_MEAN else "sum")
        if dz is not None:
            with input_grad_only(), torch.autograd.set_multithreading_enabled(_MT_BACKWARD):
                (g,) = torch.autograd.grad(z, [x_in], grad_outputs=dz)
            return g
        with torch.enable_grad():
            logits = model.head_from_pre(z)
    else:
        with torch.enable_grad(), attack_forward():
            logits = model.body(x_in) if through_body else model(x_in)
    d = spec.dlogits(logits.contiguous())
    with input_grad_only(), torch.autograd.set_multithreading_enabled(_MT_BACKWARD):
        (g,) = torch.autograd.grad(logits, [x_in], grad_outputs=d)
    return g


def _unwrap(model):
    """DDP / DataParallel wrappers add nothing to an input-gradient step (no parameter gradients are
    produced, so there is nothing to all-reduce): run the wrapped module directly."""
    inner = getattr(model, "module", None)
    return inner if isinstance(inner, torch.nn.Module) and type(model).__name__ in (
        "DistributedDataParallel", "DataParallel") else model


def input_gradient(model, x, spec):
    """g = d loss / d x for the current x (x: leaf ROCm tensor), through autograd end to end."""
    x.requires_grad_(True)
    return _body_input_grad(model, x, spec, False)


def attack_step_(model, x, x0, spec, step_size, eps, direction, lo, hi):
    """One PGD iteration in place on x (attacks.py:20-27).  Edge-enhanced models that expose their front end
    (eeadv.models._EEFrontMixin) run it as explicit kernel calls around an autograd pass over the CNN body only:
    the input gradient then never exists as one tensor - the update kernel adds its two parts in registers."""
    if getattr(model, "chain_ok", None) is not None and model.chain_ok(x):
        # two launches around the CNN body: ee_chain_fwd_f32 (draws, Add_Square, low-pass, edge filter, combine) and
        # ee_chain_bwd_f32 (gate, edge adjoint, low-pass, d Add_Square, update) - nothing of the front end's gradient reaches HBM
        with torch.no_grad():
            x_in, ctx = model.front_chain(x.detach())
        x_in.requires_grad_(True)
        g_in = _body_input_grad(model, x_in, spec, True)
        with torch.no_grad():
            model.front_chain_update_(x.detach(), g_in.contiguous(), ctx, x0, step_size, eps, lo, hi, direction)
        return
    if getattr(model, "manual_ok", None) is not None and model.manual_ok(x):
        with torch.no_grad():
            x_in, ctx = model.front_manual(x.detach())
        x_in.requires_grad_(True)
        g_in = _body_input_grad(model, x_in, spec, True)
        with torch.no_grad():
            g_lp, g_edge = model.front_manual_backward(g_in.contiguous(), ctx)
            ops.pgd_step_bcast_(x.detach(), g_lp, g_edge, x0, step_size, eps, lo, hi, direction)
        return
    g = input_gradient(model, x, spec)
    ops.pgd_step_(x.detach(), g.contiguous(), x0, step_size, eps, lo, hi, direction)


class _GraphedStep:
    """`iters` consecutive PGD steps captured into one graph, bound to static buffers (one replay per attack: consecutive
    replays of a one-step graph leave a ~12 us bubble between them)."""

    def __init__(self, model, x0, spec, step_size, eps, direction, lo, hi, iters=1):
        self.iters = iters
        self.x = torch.empty_like(x0).requires_grad_(True)
        self.x0 = torch.empty_like(x0)
        self.payload = torch.empty_like(spec.payload)
        self.spec = LossSpec(spec.kind, self.payload)
        self.cfg = (step_size, eps, direction, lo, hi)
        self.model = weakref.ref(model)
        self.graph = None

    def _body(self, model):
        step_size, eps, direction, lo, hi = self.cfg
        attack_step_(model, self.x, self.x0, self.spec, step_size, eps, direction, lo, hi)

    def capture(self, model, x_init, x0, payload):
        runtime.draw_state(x0.device)  # the device-side draws of a captured iteration (Add_Square, Net_2's dropout) need their state to exist
        self.load(x_init, x0, payload)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up outside capture: MIOpen algorithm search, allocator, autograd
                self._body(model)
        torch.cuda.current_stream().wait_stream(side)
        self.load(x_init, x0, payload)
        self.graph = torch.cuda.CUDAGraph()
        from .models import deferred_bn_counters
        with torch.cuda.graph(self.graph, capture_error_mode=runtime.capture_mode()):
            with deferred_bn_counters():  # the BatchNorm counters of the `iters` forwards: one launch at the end of the graph
                for _ in range(self.iters):
                    self._body(model)

    def load(self, x_init, x0, payload):
        with torch.no_grad():
            self.x.detach().copy_(x_init)
            self.x0.copy_(x0)
            self.payload.copy_(payload)


_GRAPHS = {}
# PGD iterations per attack that run OUTSIDE the captured graph even in graph mode, so that the library's
# HIP-event hooks (ee_prof_*) can time their kernels live; bench.py sets it (DESIGN.md "Measurement").
PROBE_ITERS = 0
# a replay covers up to this many consecutive iterations (the largest divisor of the iteration count below it): one-step
# graphs leave a ~12 us bubble between replays, very long graphs cost capture time and node memory for nothing
MAX_ITERS_PER_GRAPH = 16


def graphs_enabled():
    return os.environ.get("EEADV_GRAPH", "0") == "1"


def clear_graphs():
    _GRAPHS.clear()


def pgd_loop(model, x0, x_init, spec, num_steps, step_size, eps, direction=1, lo=0.0, hi=1.0, use_graph=None):
    """Runs num_steps updates starting from x_init (a fresh tensor the loop may update in place); returns a
    detached tensor.  attacks.py:19-27: x = clamp(min(max(x + dir*alpha*sign(g), x0-eps), x0+eps), 0, 1)."""
    model = _unwrap(model)
    x0 = x0.detach().contiguous()
    if use_graph is None:
        use_graph = graphs_enabled()
    if use_graph and num_steps > min(PROBE_ITERS, num_steps):
        probe = min(PROBE_ITERS, num_steps)
        n_graph = num_steps - probe
        chunk = max(c for c in range(1, min(n_graph, MAX_ITERS_PER_GRAPH) + 1) if n_graph % c == 0)  # iterations per replay
        key = (id(model), model.training, tuple(x0.shape), spec.kind, tuple(spec.payload.shape), spec.payload.dtype,
               float(step_size), float(eps), direction, lo, hi, x0.device.index, chunk)
        gs = _GRAPHS.get(key)
        if gs is not None and gs.model() is not model:
            gs = None
        if gs is None:
            gs = _GraphedStep(model, x0, spec, step_size, eps, direction, lo, hi, iters=chunk)
            # the two warm-up executions before capture are extra train-mode forwards: shield the BatchNorm statistics
            # from them.  Restored through .data so that autograd graphs the caller still holds (TRADES / ALP keep
            # `preds = model(x)` alive across the attack) do not see a version bump on the saved running statistics.
            saved = {}
            if model.training:
                saved = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
            gs.capture(model, x_init, x0, spec.payload)
            if saved:
                live = model.state_dict()
                for k, v in saved.items():
                    live[k].data.copy_(v)
            _GRAPHS[key] = gs
        gs.load(x_init.detach().contiguous(), x0, spec.payload)
        refresh_dense_weights()  # weight-derived buffers the captured kernels read (functional.Conv3x3Map2Fn)
        for _ in range(n_graph // chunk):
            gs.graph.replay()
        x = gs.x.detach().clone()
        if probe:
            # the probed iterations come LAST: their first kernel then follows an iteration's last one, caches as warm as inside the
            # graph (as the attack's first iteration, right behind the parameter update, the front-end kernel read its tables cold and
            # measured 23 ... 45 us from run to run against rocprofv3's 24 over the in-graph launches)
            for _ in range(probe):  # eager passes launch exactly what the graph replays
                attack_step_(model, x, x0, spec, step_size, eps, direction, lo, hi)
                x = x.detach()
        return x

    x = x_init.detach().contiguous()
    for _ in range(num_steps):
        attack_step_(model, x, x0, spec, step_size, eps, direction, lo, hi)
        x = x.detach()
    return x.requires_grad_(False)
