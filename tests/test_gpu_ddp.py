"""The data-parallel update's segmented form on the device (trainer._GraphedUpdate with a segmented ddp.FlatGradSync): eager and from the
captured graphs it must be the one-piece update's arithmetic.  No process group is needed for that: at world size 1 the collectives are
skipped, everything else - the forward detached at the segment boundaries, one backward (graph) per segment, the flat gradient buffer
laid out in backward order, the SGD graph - runs as at N > 1."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model():
    from eeadv.models import make_resnet_ee
    torch.manual_seed(11)
    return make_resnet_ee(18, "tiny", square=True, cize=64, r=8, w=1.0, low=38.0, high=76.0, alpha=0.0, sigma=1.0,
                          type_canny="CannyFilter_step125_1", epsilon=16 / 255, n_queries=1).to(DEV).train()


def _flat(m):
    return torch.cat([p.detach().flatten() for p in m.parameters() if p.requires_grad])


@pytest.mark.parametrize("graphs", [False, True])
def test_segmented_update_equals_the_one_piece_update(monkeypatch, graphs):
    from eeadv import ddp, engine, runtime, trainer
    monkeypatch.setenv("EEADV_GRAPH", "1" if graphs else "0")
    engine.clear_graphs()
    trainer.clear_update_graphs()
    base = _model()
    g = torch.Generator().manual_seed(3)
    batches = [(torch.rand(16, 3, 64, 64, generator=g).to(DEV), torch.randint(0, 200, (16,), generator=g).to(DEV)) for _ in range(4)]
    runs = {}
    for segmented in (False, True):
        monkeypatch.setattr(trainer, "_SEGMENTED", segmented)
        model = _model()
        model.load_state_dict(base.state_dict())
        sync = ddp.FlatGradSync(model)
        assert sync.segmented and [round(p.numel() * 4 / 1e6, 1) for p in sync.pieces] == [34.0, 8.4, 2.7]
        opt = trainer.make_sgd(model.parameters(), lr=0.01, momentum=0.9, weight_decay=2e-4)
        upd = trainer._GraphedUpdate(model, trainer.Criterion(), opt, *batches[0], sync)
        assert upd.segmented == segmented
        if not graphs:
            upd.eager_left = 10 ** 6
        runs[segmented] = (model, sync, opt, upd)
    (ma, sa, oa, ua), (mb, sb, ob, ub) = runs[False], runs[True]
    for k, (x, y) in enumerate(batches):
        # both forms start every step from the SAME state (the one-piece run's parameters, BatchNorm buffers and momentum): a deep ReLU
        # network amplifies rounding-level differences from step to step, which is not what is under test
        with torch.no_grad():
            for pb, pa in zip(mb.parameters(), ma.parameters()):
                pb.copy_(pa)
            for bb, ba in zip(mb.buffers(), ma.buffers()):
                bb.copy_(ba)
            for pb, pa in zip(mb.parameters(), ma.parameters()):
                if pa in oa.state and "momentum_buffer" in oa.state[pa] and oa.state[pa]["momentum_buffer"] is not None and pb in ob.state:
                    ob.state[pb]["momentum_buffer"].copy_(oa.state[pa]["momentum_buffer"])
        res = []
        for model, sync, upd in ((ma, sa, ua), (mb, sb, ub)):
            runtime.reseed()
            torch.manual_seed(5 + k)  # the Add_Square draws inside the forward
            runtime.draw_state(torch.device(DEV))  # re-read now: a capture cannot
            loss, out = upd(x, y)
            res.append((float(loss), _flat(model).clone(), sync.flat.clone(), out.clone()))
        (la, pa, ga, outa), (lb, pb, gb, outb) = res
        assert abs(la - lb) <= 1e-5 * max(1.0, abs(la)), (k, la, lb)
        assert float((outa - outb).abs().max()) <= 1e-4, k
        # the same kernels on the same operands in the same order; what may differ is MIOpen's weight-gradient kernels run to run
        # the two flat gradient buffers have the same layout (one model class): compare them whole
        assert float((ga - gb).abs().max()) <= 1e-4 * float(ga.abs().max()), (k, float((ga - gb).abs().max()), float(ga.abs().max()))
        assert float((pa - pb).abs().max()) <= 1e-5 * float(pa.abs().max()), k
    if graphs:
        assert ua.graph is not None and len(ua.seg_graphs) == 0 and ua.graph2 is not None
        assert ub.graph is not None and len(ub.seg_graphs) == 2 and ub.graph2 is not None


def test_filter_caches_follow_an_eager_fused_sgd_step():
    """torch's fused SGD moves the weights without advancing their version counters; the Winograd / stride-2 / dense filter copies
    (functional._dense_weight) are keyed by version.  After an EAGER step of trainer.make_sgd's optimiser the next forward must use the
    new filters: equal to a forward with the caches dropped (round 3: it used the old ones - the eager warm-up updates before the
    update graph is captured, and every eager TRADES / ALP step at N > 1, multiplied by stale filters)."""
    from eeadv import functional as EF, runtime, trainer
    model = _model()
    opt = trainer.make_sgd(model.parameters(), lr=0.05, momentum=0.9, weight_decay=2e-4)
    g = torch.Generator().manual_seed(9)
    x, y = torch.rand(8, 3, 64, 64, generator=g).to(DEV), torch.randint(0, 200, (8,), generator=g).to(DEV)

    def fwd():
        runtime.reseed()
        torch.manual_seed(4)
        runtime.draw_state(torch.device(DEV))
        with torch.no_grad():
            return model(x)
    before = fwd()
    assert len(EF._DENSE_W) > 0
    for _ in range(2):
        loss = trainer.Criterion()(model(x), y)
        opt.zero_grad()
        loss.backward()
        opt.step()
    after = fwd()
    EF._DENSE_W.clear()
    fresh = fwd()
    assert float((after - before).abs().max()) > 1e-3  # the step moved the logits ...
    assert torch.equal(after, fresh)                   # ... and the cached filter copies moved with it


@pytest.mark.gpu
def test_free_at_repeat_with_its_collectives_in_one_graph():
    """config 5's multi-rank repeat (SyncBatchNorm's exchanges + the gradient pieces' all-reduce) captured into ONE HIP graph
    (EEADV_GRAPH_COLLECTIVES=1), rehearsed with one rank over RCCL and every collective forced on: on the 64 x 64 ResNet-18 - whose whole step
    is bit-reproducible - the graphed repeats reproduce the eager ones BIT FOR BIT (parameters, running statistics, noise), and the host needs
    less time per repeat than the device (scripts/freeat_graph_collectives.py asserts both)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(29700 + os.getpid() % 200), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "freeat_graph_collectives.py"), "18", "16", "64", "tiny"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "eager vs graphed: max |parameter difference| 0.000e+00" in r.stdout, r.stdout[-1500:]


@pytest.mark.gpu
def test_multi_rank_updates_rehearsed_with_one_rccl_rank():
    """The N > 1 forms of the training step - AT with the one-piece and the segmented gradient exchange, TRADES as three graphs around its
    all-reduce (experiments_tinyimagenet.py:250-306) - with ONE rank over RCCL and every collective forced on (scripts/ddp_same_gpu.py asserts
    equal parameters, captured graphs, finite statistics).  Regression: round 4's `Trades.last_logits_adv` stayed alive behind the captured
    backward of the multi-rank path (it holds tensors of the graph's memory pool) and the process died with SIGSEGV at the next capture."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 200),
               DDP_BACKEND="nccl", EEADV_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "ddp_same_gpu.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "TRADES segmented=True" in r.stdout and "graphs g1 g2 g3: [True, True, True]" in r.stdout, r.stdout[-1500:]
