#!/usr/bin/env python3
"""bench.py - adversarial images / second of the hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torchrun environment: this process stays off the GPU and starts `python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a child (one rank per GPU over RCCL), relays its output and exits with its code.
Started by torch.distributed.run itself (RANK / WORLD_SIZE set) it is a rank.

Default workload = BASELINE config "Tiny-ImageNet ResNet-18 AT + edge-enhance" (SURVEY.md 8(d) item 4,
Tiny_ImageNet/configs_tinyimagenet/ee_at_bpda3_square.yml): per-rank batch [100,3,64,64] synthetic U[0,1),
labels randint(200), model resnet18_EE_square (CannyFilter_step125_1, r=8, w=1, alpha=0, sigma=1,
high=76/255, Add_Square n_queries=1), train mode.  One "step" = one training step of the reference's
train() loop (Tiny_ImageNet/experiments_tinyimagenet.py:234-306): PGD-10 attack (eps 16/255, alpha 2/255,
random start) + forward on the adversarial batch + cross-entropy + backward + SGD(momentum, wd) update
(+ the gradient all-reduce over RCCL when N > 1).  value = N * B * K / time (weak scaling), fp32 throughout.

Printed JSON also carries
  roofline     : the dominant hand-written kernel = the family with the largest (launches x mean duration) among those
                 timed: algorithmic bytes per launch / its mean duration measured with HIP events on the launch
                 stream inside the timed region (one PGD iteration per attack runs outside the HIP graph so
                 that its kernels can be bracketed by events - see DESIGN.md "Measurement"); the rocprofv3 figure of
                 the same command (profiles/) is carried next to it.
  kernels      : the same measurement for every hand-written kernel family.
  cpu_baseline : the CPU oracle (oracle/ref_path.py, pinned to the reference) running the SAME step with
                 plain PyTorch CPU ops on the host cores, rank 0 at N = 1 only, bounded sample.
End-to-end throughput is bounded by the CNN's convolutions (MIOpen, fp32 MFMA), not by these kernels.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak (same guide: v_mfma_f32_32x32x2_f32 at the f32 vector rate)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 GB/s achievable

WORKLOADS = {
    # name: (arch, method, batch, shape, classes, eps, alpha, K)
    "tiny_ee_at": dict(arch="resnet18_EE_square", method="EE_BPDA3_AT_square", batch=100, shape=(3, 64, 64), classes=200,
                       eps=0.062745098039216, alpha=0.007843137254902, steps=10, lr=0.1, momentum=0.9, wd=2e-4),
    "tiny_at": dict(arch="resnet18", method="AT", batch=100, shape=(3, 64, 64), classes=200,
                    eps=0.062745098039216, alpha=0.007843137254902, steps=10, lr=0.1, momentum=0.9, wd=2e-4),
    "tiny_trades": dict(arch="resnet18", method="TRADES", batch=100, shape=(3, 64, 64), classes=200,
                        eps=0.062745098039216, alpha=0.003921568627451, steps=10, lr=0.1, momentum=0.9, wd=2e-4, beta=6.0),
    "mnist_ee_at": dict(arch="Net2_EE_square", method="EE_BPDA3_AT_square", batch=50, shape=(1, 28, 28), classes=10,
                        eps=0.3, alpha=0.01, steps=40, lr=0.1, momentum=0.3, wd=1e-4),
    # BASELINE config 5: ImageNet/free_imagenet/AT_free_imagenet_ddp.py at its defaults (:41-99) with `-a resnet50`: global batch 256
    # over 8 ranks = 32 per rank, clip_eps 4/255, fgsm_step 4/255, n_repeats 4, lr 0.1, momentum 0.9, wd 1e-4; SyncBatchNorm at N > 1
    # (:149).  One step = one batch through its 4 repeats (:286-309): each repeat is forward + CE + backward (weight AND input
    # gradient) + noise update + SGD step.
    "imagenet_free_at": dict(arch="resnet50", method="free_AT", batch=32, shape=(3, 224, 224), classes=1000,
                             eps=4.0 / 255, alpha=4.0 / 255, steps=4, lr=0.1, momentum=0.9, wd=1e-4),
    # One batch of validate() (Tiny_ImageNet/experiments_tinyimagenet.py:326-397) for the headline model: model.eval() (:337), PGD with
    # num_steps_2 = 50 / step_size_2 = 1/255 (ee_at_bpda3_square.yml), then the clean and the adversarial forward under no_grad and the
    # two cross-entropies / top-k (:376-395).  The reference's log: 3.076 s per batch of 100 (LOG_B3:1310).  No optimiser.
    "tiny_ee_eval_pgd50": dict(arch="resnet18_EE_square", method="EE_BPDA3_AT_square", batch=100, shape=(3, 64, 64), classes=200,
                               eps=0.062745098039216, alpha=0.003921568627451, steps=50, lr=0.1, momentum=0.9, wd=2e-4, eval_only=True),
}
OTHER_WORKLOADS = ("tiny_trades", "mnist_ee_at", "imagenet_free_at", "tiny_ee_eval_pgd50")  # the other BASELINE configs, timed briefly behind the headline one


class Args:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def build_model(cfg, oracle=False):
    if oracle:
        from oracle import ref_path as R
        if cfg["arch"] == "resnet18":
            return R.resnet18()
        if cfg["arch"] == "resnet50":
            return R.resnet50(num_classes=cfg["classes"], imagenet_pool=True)
        if cfg["arch"] == "resnet18_EE_square":
            front = R.EEFront(64, 3, 8, 1.0, 38.0, 76.0, 0.0, 1.0, "CannyFilter_step125_1", False, True, cfg["eps"], 1)
            return R.EEModel(front, R.resnet18())
        if cfg["arch"] == "Net2_EE_square":
            front = R.EEFront(28, 1, 4, 1.0, 25.0, 51.0, 0.3, 1.0, "CannyFilter_step125_1", False, True, cfg["eps"], 1)
            return R.EEModel(front, R.Net_2())
        raise ValueError(cfg["arch"])
    from eeadv import models as M
    if cfg["arch"] == "resnet18":
        return M.make_resnet(18, "tiny")
    if cfg["arch"] == "resnet50":
        return M.make_resnet(50, "imagenet")
    if cfg["arch"] == "resnet18_EE_square":
        return M.make_resnet_ee(18, "tiny", True, cize=64, r=8, w=1.0, with_gf=False, low=38.0, high=76.0, alpha=0, sigma=1.0,
                                type_canny="CannyFilter_step125_1", epsilon=cfg["eps"], n_queries=1)
    if cfg["arch"] == "Net2_EE_square":
        return M.Net2_EE_square(r=4, w=1.0, with_gf=False, low=25.0, high=51.0, alpha=0.3, sigma=1.0,
                                type_canny="CannyFilter_step125_1", epsilon=cfg["eps"], n_queries=1)
    raise ValueError(cfg["arch"])


def driver_args(cfg):
    """The EasyDict the reference drivers build from YAML + CLI, reduced to what the step reads."""
    return Args(method_name=cfg["method"], random=True, epsilon=cfg["eps"], num_steps_1=cfg["steps"], step_size_1=cfg["alpha"],
                beta=cfg.get("beta", 1.0), num_classes=cfg["classes"], attack_method="PGD")


def host_cores():
    """Cores this process may actually use: the cgroup CPU quota when there is one (the GPU box shows 256
    CPUs but grants 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cfg, seconds_target=25.0, threads=None):
    """The oracle's step on the host cores (cpu_baseline.kind = 'port'): oracle/ref_path.py, pinned to the reference, the same
    step with torch CPU ops.  threads: default = the cores this process may use (cgroup quota)."""
    import torch.nn.functional as F
    from oracle import ref_path as R
    threads = threads or host_cores()
    torch.set_num_threads(threads)
    torch.manual_seed(1)
    model = build_model(cfg, oracle=True).train()
    opt = torch.optim.SGD(model.parameters(), lr=cfg["lr"], momentum=cfg["momentum"], weight_decay=cfg["wd"])
    B = cfg["batch"]
    x = torch.rand(B, *cfg["shape"])
    y = torch.randint(0, cfg["classes"], (B,))
    args = Args(random=True, epsilon=cfg["eps"])
    noise = torch.zeros(B, *cfg["shape"]) if cfg["method"] == "free_AT" else None

    if cfg.get("eval_only"):
        model.eval()

    def step():
        if cfg.get("eval_only"):  # one batch of validate(): attack, clean and adversarial forward, the two losses
            adv = R.PGD(model, args, x, y, cfg["steps"], cfg["alpha"])
            with torch.no_grad():
                F.cross_entropy(model(x), y), F.cross_entropy(model(adv), y)
            return
        if cfg["method"] == "free_AT":
            for _ in range(cfg["steps"]):
                R.free_at_repeat(model, F.cross_entropy, opt, x, y, noise, cfg["alpha"], cfg["eps"])
            return
        if cfg["method"] == "TRADES":
            tr = R.Trades(cfg["alpha"], cfg["eps"], cfg["steps"], cfg["beta"])
            preds = model(x)
            adv = tr.PGD_Linf(model, x, preds)
            model(adv)
            loss = tr.loss(model, preds, adv, y, opt)
        else:
            adv = R.PGD(model, args, x, y, cfg["steps"], cfg["alpha"])
            loss = F.cross_entropy(model(adv), y)
        opt.zero_grad()
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    if cfg["method"] == "free_AT":  # a whole step is ~20 s on 16 cores: one repeat warms up
        R.free_at_repeat(model, F.cross_entropy, opt, x, y, noise, cfg["alpha"], cfg["eps"])
        warm = (time.perf_counter() - t0) * cfg["steps"]
    else:
        step()  # warm-up (also sizes the sample)
        warm = time.perf_counter() - t0
    n = max(1, min(40, int(seconds_target / max(warm, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 2), "unit": "adversarial images/s", "cores": threads, "kind": "port",
            "sample": "%d training steps of batch %d (after a warm-up), %.1f s, torch CPU ops, %d threads" % (n, B, dt, threads)}


class Job:
    """One workload on this rank: model, optimiser, gradient exchange, resident synthetic batches, and `step(i)` = one step of the
    reference's train() loop for that config (trainer.train_batch; free-AT: trainer.free_at_repeat x n_repeats)."""

    def __init__(self, name, cfg, dev, world, rank, channels_last=False):
        from utils.helper import set_seed
        from eeadv import ddp, trainer
        self.name, self.cfg, self.dev, self.world = name, cfg, dev, world
        set_seed(1 + rank)  # experiments_imagenet.py:61: seed + rank
        model = build_model(cfg).to(dev).train()
        if channels_last:
            model = model.to(memory_format=torch.channels_last)
        self.free_at = cfg["method"] == "free_AT"
        self.eval_only = bool(cfg.get("eval_only"))
        if self.eval_only:
            model.eval()  # experiments_tinyimagenet.py:337
        if self.free_at and world > 1:  # AT_free_imagenet_ddp.py:149
            model = ddp.convert_sync_batchnorm(model)
        self.model = model
        self.optimizer = trainer.make_sgd(model.parameters(), lr=cfg["lr"], momentum=cfg["momentum"], weight_decay=cfg["wd"])
        # N > 1: one flat gradient buffer all-reduced over RCCL between the captured halves of the update (eeadv.ddp.FlatGradSync);
        # the attack needs no collective at all.  EEADV_GRAD_SYNC=ddp: DistributedDataParallel + the eager update instead.
        self.sync, self.run_model = None, model
        if world > 1:
            if ddp.grad_sync_mode() == "ddp":
                self.run_model = ddp.wrap(model, dev)
            else:
                self.sync = ddp.FlatGradSync(model)
        self.dargs = driver_args(cfg)
        self.criterion = trainer.Criterion() if self.free_at else trainer.make_criterion(self.dargs)
        B = cfg["batch"]
        self.batches = [(torch.rand(B, *cfg["shape"], device=dev), torch.randint(0, cfg["classes"], (B,), device=dev)) for _ in range(4)]
        # :261: the persistent perturbation has GLOBAL-batch rows on every rank; a rank reads and updates its first B rows
        self.noise = torch.zeros(B * world, *cfg["shape"], device=dev) if self.free_at else None
        self.free_step = trainer.FreeAtStep(self.run_model, self.criterion, self.optimizer, self.noise, cfg["alpha"], cfg["eps"],
                                            cfg["steps"], sync=self.sync) if self.free_at else None

    def step(self, i):
        from eeadv import trainer
        x, y = self.batches[i % len(self.batches)]
        if self.free_at:
            return self.free_step(x, y)
        if self.eval_only:  # one batch of validate(): attack in eval mode + clean and adversarial forward + losses + top-k
            return trainer.validate_batch(self.run_model, self.criterion, self.dargs, x, y, self.dev, self.cfg["steps"], self.cfg["alpha"], self.cfg["classes"])
        return trainer.train_batch(self.run_model, self.criterion, self.optimizer, self.dargs, x, y, self.dev, sync=self.sync)

    def describe(self):
        cfg = self.cfg
        if self.free_at:
            return "%s: %s free-AT, per-rank batch %d x %s, %d repeats per batch, clip_eps %.4f fgsm_step %.4f, each repeat = fwd + bwd (weights and input) + noise update + SGD%s" % (
                self.name, cfg["arch"], cfg["batch"], "x".join(map(str, cfg["shape"])), cfg["steps"], cfg["eps"], cfg["alpha"],
                ", SyncBatchNorm + gradient all-reduce (RCCL)" if self.world > 1 else "")
        if self.eval_only:
            return "%s: %s %s, one batch of validate(): model.eval(), per-rank batch %d x %s, PGD-%d eps %.4f alpha %.4f + clean and adversarial forward, CE, top-k" % (
                self.name, cfg["arch"], cfg["method"], cfg["batch"], "x".join(map(str, cfg["shape"])), cfg["steps"], cfg["eps"], cfg["alpha"])
        return "%s: %s %s, per-rank batch %d x %s, PGD-%d eps %.4f alpha %.4f, train step incl. SGD%s" % (
            self.name, cfg["arch"], cfg["method"], cfg["batch"], "x".join(map(str, cfg["shape"])), cfg["steps"], cfg["eps"], cfg["alpha"],
            ", DDP all-reduce (RCCL)" if self.world > 1 else "")

    def grad_sync(self):
        if self.world == 1:
            return None
        if self.sync is None:
            return "DistributedDataParallel, eager update (EEADV_GRAD_SYNC=ddp)"
        return "flat %.1f MB in %d pieces, %s" % (self.sync.flat.numel() * 4 / 1e6, len(self.sync.pieces), self.sync.describe())


def fence(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()


def time_other_workload(name, dev, world, rank, steps, warmup):
    """A short timed run of another BASELINE config behind the headline one (same contract: barrier + synchronize on both sides, max
    over ranks); returns {"value", "ms_per_step", "steps", "config"}."""
    from eeadv import engine, trainer
    cfg = dict(WORKLOADS[name])
    engine.clear_graphs()
    trainer.clear_update_graphs()
    engine.PROBE_ITERS = 0
    # MIOpen's solver search over ResNet-50's ~50 convolution shapes x 3 directions compiles kernels for four and a half minutes on a fresh
    # box (measured: this leg 283 s with the search, ~40 s without); the search buys 7 % (553 vs 518 img/s).  With the recorded find-db
    # (eeadv.runtime.use_shipped_miopen_db) the search is a lookup; without it (--no-miopen-db) this extra leg of the default run takes MIOpen's
    # immediate-mode solvers, and `--workload imagenet_free_at` on its own searches.
    find_before = torch.backends.cudnn.benchmark
    if name == "imagenet_free_at" and MIOPEN_DB is None:
        torch.backends.cudnn.benchmark = False
    job = Job(name, cfg, dev, world, rank)
    for i in range(SETUP_STEPS + warmup):
        job.step(i)
    fence(world)
    t0 = time.perf_counter()
    for i in range(steps):
        last = job.step(i)
    fence(world)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res = {"value": round(world * cfg["batch"] * steps / dt, 2), "unit": "adversarial images/s", "ms_per_step": round(1e3 * dt / steps, 3),
           "steps": steps, "warmup": warmup, "final_loss": round(float(last[0].item()), 5),
           "config": {"workload": job.describe(), "global_batch": world * cfg["batch"], "grad_sync": job.grad_sync(),
                      "miopen_find": bool(torch.backends.cudnn.benchmark), "miopen_db": "recorded" if MIOPEN_DB else None,
                      "miopen_db_matched": _db_matched(), "fallback_layers": _fallback_layers(job.model)}}
    torch.backends.cudnn.benchmark = find_before
    del job, last
    engine.clear_graphs()
    trainer.clear_update_graphs()
    torch.cuda.empty_cache()
    return res


def _db_matched():
    """config.miopen_db_matched: did MIOpen actually take the shipped find-db (eeadv.runtime.shipped_miopen_db_matched)"""
    from eeadv import runtime
    return runtime.shipped_miopen_db_matched(MIOPEN_DB)


def _fallback_layers(model):
    """config.fallback_layers: the convolutions of `model` that ran on MIOpen / Tensile instead of a hand-written kernel in the steps
    just timed ({"count", "of", "layers": [...]}; eeadv.models records the route every convolution takes)."""
    from eeadv import models as M
    return M.fallback_report(model)


SETUP_STEPS = 3
MIOPEN_DB = None  # the private copy of the recorded MIOpen find-db this process uses (main sets it)


def calibrate_bracket(ops, N, cfg, dev, n=200):
    """Microseconds a HIP-event bracket adds to one launch, measured before the timed region on the update kernel at the bench
    shape: (mean bracket around single launches) - (spacing of the same launches issued back to back, which is what
    rocprofv3's per-kernel duration plus the inter-kernel gap amounts to)."""
    C, H, W = cfg["shape"]
    x0 = torch.rand(cfg["batch"], C, H, W, device=dev)
    x, g = x0.clone(), torch.randn_like(x0)
    for _ in range(10):
        ops.pgd_step_(x, g, x0, cfg["alpha"], cfg["eps"])
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()  # back to back on the device: eager launches would be spaced by the host, not by the GPU
    with torch.cuda.graph(graph):
        for _ in range(50):
            ops.pgd_step_(x, g, x0, cfg["alpha"], cfg["eps"])
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n // 50):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    spacing_us = 1e3 * e0.elapsed_time(e1) / (n // 50 * 50)
    ops.prof_reset()
    ops.prof_enable(True)
    for _ in range(n):
        ops.pgd_step_(x, g, x0, cfg["alpha"], cfg["eps"])
    torch.cuda.synchronize()
    ms, cnt = ops.prof_read(N.K_PGD_STEP)
    ops.prof_enable(False)
    ops.prof_reset()
    return max(1e3 * ms / cnt - spacing_us, 0.0) if cnt else 0.0


def large_batch_kernels(cfg, dev, mult=16, iters=50):
    """The section-8 kernels alone at `mult` x the per-rank batch (outside the timed region; one second in total): what they
    reach once a launch is no longer latency-bound - the counterpart of `roofline`, which is quoted at the reference batch.
    Torch events around `iters` back-to-back launches on the launch stream (launch overhead amortised)."""
    from eeadv import ops
    C, H, W = cfg["shape"]
    B = cfg["batch"] * mult
    x, xh, g = (torch.rand(B, C, H, W, device=dev) for _ in range(3))
    wts = ops.EdgeWeights(1.0)
    x_in, gate, _, sgx, sgy = ops.frontend_fwd_save(x, xh, wts, 0.0, 0.2, 0.5)
    x0 = x.clone()
    runs = {
        "ee_pgd_step": (lambda: ops.pgd_step_(x0, g, x, cfg["alpha"], cfg["eps"]), 16 * C),
        "ee_frontend_fwd": (lambda: ops.frontend_fwd_save(x, xh, wts, 0.0, 0.2, 0.5), 12 * C),
        "ee_frontend_bwd": (lambda: ops.frontend_bwd_saved(g, gate, sgx, sgy, wts, 0.0, 0.2, 0.5), 16 * C),
    }
    res = {"batch": B}
    for name, (fn, bytes_px) in runs.items():
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / iters
        gbs = bytes_px * B * H * W / us / 1e3
        res[name] = {"avg_us": round(us, 2), "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    return res


def self_launch(n, argv):
    """Parent of an N-rank run.  Must not touch the GPU (no torch.cuda.* / HIP call happened in this process: torch is only
    imported) and must not exec: the ranks are a CHILD process tree; stdout / stderr pass through, the exit code is theirs."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    sys.stderr.write("[bench] starting %d ranks: %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.run(cmd, env=env).returncode


def dry_launch(world, rank):
    """--dry-launch: the multi-rank plumbing without a GPU (gloo): rendezvous, barrier, max-over-ranks, rank 0 prints one line."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    seen = torch.ones(1, dtype=torch.int64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "config": {"ranks": int(seen.item()), "backend": "gloo" if world > 1 else None},
                          "max_over_ranks": float(t.item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="tiny_ee_at", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-rank batch (default: the reference config's)")
    ap.add_argument("--no-graph", action="store_true", help="run every PGD iteration eagerly (no HIP graph)")
    ap.add_argument("--probe-iters", type=int, default=1,
                    help="PGD iterations per attack that run outside the HIP graph so their kernels can be event-timed")
    ap.add_argument("--probe-every", type=int, default=10,
                    help="steps between two probed attacks (the eager probe iteration costs ~1 %% of a step; its kernels are the "
                         "roofline samples, so at least one step of the timed region is always probed)")
    ap.add_argument("--large-batch", action="store_true",
                    help="also time the section-8 kernels alone at 16x the batch (off by default: those launches would enter the "
                         "rocprofv3 per-kernel averages of this command, which must agree with `roofline.avg_launch_us`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="time only the headline workload")
    ap.add_argument("--other-steps", type=int, default=10, help="timed steps of each of the other BASELINE configs")
    ap.add_argument("--extras-timeout", type=int, default=420, help="seconds after which the headline line is printed without the extra legs")
    ap.add_argument("--dry-launch", action="store_true", help="only prove that --gpus N starts N ranks (gloo, no GPU needed)")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--no-miopen-db", action="store_true",
                    help="do not use the recorded MIOpen find-db (edge-enhancement_amd/miopen_db): MIOpen then searches its solvers itself")
    ap.add_argument("--no-miopen-benchmark", action="store_true",
                    help="leave torch.backends.cudnn.benchmark off (default: on, MIOpen searches its solvers once per shape)")
    a = ap.parse_args()

    cfg = dict(WORKLOADS[a.workload])
    if a.batch:
        cfg["batch"] = a.batch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a.gpus, sys.argv[1:]))  # before anything touches the GPU
    if world != a.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (a.gpus, world))
    if a.dry_launch:
        return dry_launch(world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the HIP path has no CPU fallback)")
    share = os.environ.get("EEADV_SHARE_GPU", "0") == "1"  # rehearsal of the N > 1 path on a one-GPU box: every rank on cuda:0, gloo between them
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    torch.backends.cudnn.benchmark = not a.no_miopen_benchmark
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(os.environ.get("EEADV_DIST_BACKEND") or "gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from eeadv import engine, ops, runtime, trainer, _native as N

    global MIOPEN_DB
    MIOPEN_DB = None if a.no_miopen_db else runtime.use_shipped_miopen_db()  # before the first convolution
    os.environ["EEADV_GRAPH"] = "0" if a.no_graph else "1"
    engine.PROBE_ITERS = 0 if a.no_graph else a.probe_iters
    job = Job(a.workload, cfg, dev, world, rank, channels_last=a.channels_last)
    B = cfg["batch"]
    probe_iters = engine.PROBE_ITERS

    def run(n):
        last = None
        for i in range(n):
            engine.PROBE_ITERS = probe_iters if i % max(1, a.probe_every) == 0 else 0
            if trainer.PHASE_EVENTS is not None:
                trainer.PHASE_EVENTS.start()
            last = job.step(i)
            ops.prof_mark_empty()  # one empty event bracket per step: the bracket's own cost, measured live
        return last

    bracket_cost_us = calibrate_bracket(ops, N, cfg, dev)
    run(SETUP_STEPS)  # one-off setup, not warm-up: MIOpen algorithm search and the two HIP-graph captures (the update graph is
    fence(world)      # captured on the third step of a configuration) must not land in the timed region when W < 3
    run(a.warmup)
    fence(world)
    ops.prof_reset()
    ops.prof_enable(True)
    if world > 1:
        trainer.PHASE_EVENTS = trainer.PhaseEvents()  # per-rank device time of attack / backward / all-reduce / SGD
    t0 = time.perf_counter()
    last = run(a.steps)
    fence(world)
    dt = time.perf_counter() - t0
    ops.prof_enable(False)
    phases = trainer.PHASE_EVENTS.summary() if trainer.PHASE_EVENTS is not None else None
    trainer.PHASE_EVENTS = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(last[0].item())
    workload_text, grad_sync_text = job.describe(), job.grad_sync()
    non_default_switches = runtime.non_default_switches()

    if rank == 0:
        C, H, W = cfg["shape"]
        px = B * H * W
        # algorithmic bytes per launch: what the kernel must move once, whatever it re-reads
        per_launch = {
            "ee_chain_bwd": (N.K_CHAIN_BWD, (17 * C + 8) * px), "ee_chain_fwd": (N.K_CHAIN_FWD, (9 * C + 8) * px),
            "ee_frontend_bwd": (N.K_FRONTEND_BWD, 16 * C * px), "ee_frontend_fwd": (N.K_FRONTEND_FWD, 12 * C * px),
            "ee_hfs": (N.K_HFS, 8 * C * px), "ee_hfs_square_fwd": (N.K_HFS_SQ_FWD, 8 * C * px), "ee_hfs_square_bwd": (N.K_HFS_SQ_BWD, 12 * C * px),
            "ee_pgd_step": (N.K_PGD_STEP, 16 * C * px), "ee_pgd_step_bcast": (N.K_PGD_STEP_BCAST, (16 * C + 4) * px),
            "ee_square_draw": (N.K_SQUARE_DRAW, 4 * (B * C * W + 1 + C)), "ee_ce": (N.K_CE, 3 * B * cfg["classes"] * 4),
        }
        ems, ecnt = ops.prof_read(N.K_EMPTY)
        empty_us = 1e3 * ems / ecnt if ecnt else 0.0
        # what a bracket adds to a launch: calibrated against back-to-back launch spacing (calibrate_bracket); the empty
        # bracket of the same region is reported next to it (it over-estimates: two markers with nothing between them are
        # not what surrounds a kernel, and subtracting it put every kernel ~2.8 us under its rocprofv3 duration)
        overhead_us = bracket_cost_us
        kernels = {}
        for name, (kid, nbytes) in per_launch.items():
            ms, cnt = ops.prof_read(kid)
            if cnt:
                raw = 1e3 * ms / cnt
                us = max(raw - overhead_us, 0.5)  # event pair cost removed
                kernels[name] = {"launches_timed": cnt, "avg_us": round(us, 3), "avg_bracket_us": round(raw, 3), "bytes": nbytes,
                                 "GBps": round(nbytes / us / 1e3, 1), "share_of_timed_us": None}
        # matrix-core families (the residual blocks' 3x3 convolutions, ee_wino.hip / ee_s2.hip): the library sums the floating-point
        # operations its timed launches declared (2 * 9 * Cin * Cout * B * H * W each), so mixed shapes average correctly
        # (ee_wino3x3_fused: the same Winograd products with a BatchNorm's work folded into the launch - eval-mode fold, train-mode exchange
        # across the kernel boundary; its flops count the convolution only, so its fraction is NOT comparable with the plain kernels')
        mfma_fams = {"ee_wino3x3": N.K_WINO, "ee_wino3x3_fused": N.K_WINO_FUSED, "ee_conv3x3s2_small_fwd": N.K_CONV3S2_FWD,
                     "ee_conv3x3s2_small_bwd_data": N.K_CONV3S2_BWD}
        # Winograd F(2x2,3x3) executes 16 multiplies per 2x2 output tile where the convolution has 36: `flops` stays the convolution's
        # algorithmic count, `executed_flops` = 4/9 of it is what the matrix cores actually do
        executed_share = {"ee_wino3x3": 4.0 / 9.0, "ee_wino3x3_fused": 4.0 / 9.0}
        for name, kid in mfma_fams.items():
            ms, cnt = ops.prof_read(kid)
            if cnt:
                raw = 1e3 * ms / cnt
                us = max(raw - overhead_us, 0.5)
                flops = ops.prof_read_work(kid) / cnt
                kernels[name] = {"launches_timed": cnt, "avg_us": round(us, 3), "avg_bracket_us": round(raw, 3), "flops": flops,
                                 "TFLOPs": round(flops / us / 1e6, 2), "share_of_timed_us": None}
                if name in executed_share:
                    kernels[name]["executed_flops"] = flops * executed_share[name]
                    kernels[name]["executed_TFLOPs"] = round(flops * executed_share[name] / us / 1e6, 2)
        # the dominant hand-written kernel of the path = the family with the largest launches x duration among the timed probes
        # (a probe iteration launches every family as often as a graph replay does, so the probe counts are proportional to
        # the real ones)
        tot = sum(k["launches_timed"] * k["avg_us"] for k in kernels.values()) or 1.0
        for k in kernels.values():
            k["share_of_timed_us"] = round(k["launches_timed"] * k["avg_us"] / tot, 4)
        weight = lambda k: kernels[k]["launches_timed"] * kernels[k]["avg_us"]
        # forward and backward-data of one convolution weigh the same to within the run-to-run noise: shares compared at 1 %, the
        # forward family named on a tie, so that the line names the same kernel every run
        order = lambda k: (round(weight(k) / tot, 2), k.endswith("_fwd"), weight(k))
        dom = max(kernels, key=order) if kernels else None
        hbm_fams = [k for k in kernels if "bytes" in kernels[k]]
        dom_hbm = max(hbm_fams, key=weight) if hbm_fams else None

        def committed(fname, key, sub=None):
            try:
                d = json.load(open(os.path.join(ROOT, "profiles", fname)))
                return (d.get(sub, {}) if sub else d).get(key)
            except (OSError, ValueError):
                return None

        def roofline_of(name):
            """HBM-bound families: algorithmic bytes / launch time against 8 TB/s; matrix-core families: declared flops /
            launch time against the dense f32 MFMA peak.  `traffic` = HBM bytes per launch from the TCC counters and
            `rocprofv3_avg_us` = the rocprofv3 --kernel-trace --stats average, both measured with the same command at this
            exact shape and committed under profiles/ (per round)."""
            if name is None:
                return None
            k = kernels[name]
            rp = committed("rocprof_kernel_us.json", name)
            traffic = committed("pmc_traffic.json", name, "%dx%dx%dx%d" % (B, C, H, W))
            common = {"traffic": traffic, "avg_launch_us": k["avg_us"], "avg_bracket_us": k["avg_bracket_us"],
                      "event_pair_overhead_us": round(overhead_us, 3), "empty_bracket_us": round(empty_us, 3), "rocprofv3_avg_us": rp}
            if "bytes" in k:
                return dict({"kernel": name, "bound": "hbm", "achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(k["GBps"] / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": k["bytes"],
                             "frac_from_rocprofv3": round(k["bytes"] / rp / 1e3 / HBM_PEAK_GBS, 4) if rp else None}, **common)
            extra = {}
            if "executed_flops" in k:  # Winograd: the matrix cores run fewer flops than the convolution has
                extra = {"executed_flops_per_launch": k["executed_flops"], "executed_TFLOPs": k["executed_TFLOPs"],
                         "frac_executed": round(k["executed_TFLOPs"] / F32_MFMA_PEAK_TFLOPS, 4)}
            return dict({"kernel": name, "bound": "mfma", "achieved": k["TFLOPs"], "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(k["TFLOPs"] / F32_MFMA_PEAK_TFLOPS, 4), "algorithmic_flops_per_launch": k["flops"],
                         "frac_from_rocprofv3": round(k["flops"] / rp / 1e6 / F32_MFMA_PEAK_TFLOPS, 4) if rp else None}, **extra, **common)

        roofline = roofline_of(dom)
        # the largest HBM-bound family as well (the fused front end), when a matrix-core family dominates
        roofline_hbm = roofline_of(dom_hbm) if dom_hbm != dom else None
        out = {
            "metric": "adversarial images/sec (%s, %s)" % ("free-AT x%d" % cfg["steps"] if cfg["method"] == "free_AT" else "PGD-%d" % cfg["steps"], cfg["arch"]),
            "value": round(world * B * a.steps / dt, 2), "unit": "adversarial images/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_text,
                "global_batch": world * B, "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                "backend": dist.get_backend() if world > 1 else None, "grad_sync": grad_sync_text,
                "switches": non_default_switches,
                "rank0_phase_ms": phases, "setup_steps": SETUP_STEPS, "hip_graph": not a.no_graph, "miopen_db": "recorded" if MIOPEN_DB else None, "miopen_db_matched": _db_matched(),
                "fallback_layers": _fallback_layers(job.model), "probe_iters": probe_iters, "probe_every": a.probe_every,
                "device": (N.lib.ee_device_name() or b"?").decode()},
            "roofline": roofline, "roofline_front_end": roofline_hbm, "kernels": kernels, "final_loss": round(loss_val, 5),
            "note": "throughput is bounded by the classifier's fp32 convolutions (hand-written MFMA kernels + MIOpen) and BatchNorm launches, "
                    "not by the front-end / update / loss kernels (3 % of an iteration); reference log (unrecorded GPU): ~143 img/s for "
                    "this config (BASELINE.md)",
        }
        if world == 1 and a.large_batch:
            out["kernels_large_batch"] = large_batch_kernels(cfg, dev)
    else:
        out = None

    # ---- everything below is outside the timed region of the headline workload; the ONE JSON line is printed at the very end, or by
    # the watchdog if an extra leg does not come back (a collective of a config that has never run on this node must not cost the line)
    printed = []

    def emit():
        if rank == 0 and not printed:
            printed.append(1)
            print(json.dumps(out), flush=True)

    def watchdog():
        sys.stderr.write("[bench] the extra workloads did not finish within %d s: printing the headline line without them\n" % a.extras_timeout)
        if rank == 0:
            out["other_workloads_error"] = "timed out after %d s" % a.extras_timeout
        emit()
        os._exit(3)  # non-zero: a stuck extra leg (a collective that never returns) must not read as a clean run

    t_start = time.perf_counter()

    def leg(name):
        sys.stderr.write("[bench] %7.1f s after the headline workload: %s\n" % (time.perf_counter() - t_start, name))
        sys.stderr.flush()

    import threading
    timer = threading.Timer(a.extras_timeout, watchdog)
    timer.daemon = True
    timer.start()
    del job, last
    others = {}
    if not a.no_other_workloads and a.workload == "tiny_ee_at":
        # the other BASELINE configs, ~10 steps each (the free-AT one at every N: it is the DDP config 5; the single-GPU ones at N = 1)
        for name in OTHER_WORKLOADS:
            if world > 1 and name != "imagenet_free_at":
                continue
            try:
                leg("other workload " + name)
                others[name] = time_other_workload(name, dev, world, rank, a.other_steps, 2)
            except Exception as exc:  # noqa: BLE001 - reported on the line, the headline number stands
                others[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
                if world > 1:
                    break  # ranks may have diverged: no further collectives
    if rank == 0:
        if others:
            out["other_workloads"] = others
        if world == 1 and not a.no_cpu_baseline:
            # all the cores this process may use AND n = 8 (the survey container's count), same step
            leg("cpu baselines")
            out["cpu_baseline"] = cpu_baseline(cfg, 14.0)
            if out["cpu_baseline"]["cores"] != 8:
                out["cpu_baseline_8_threads"] = cpu_baseline(cfg, 10.0, threads=8)
            if "imagenet_free_at" in others and "error" not in others["imagenet_free_at"]:
                others["imagenet_free_at"]["cpu_baseline"] = cpu_baseline(dict(WORKLOADS["imagenet_free_at"]), 12.0)
    timer.cancel()
    leg("done")
    emit()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
