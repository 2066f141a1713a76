#!/usr/bin/env python3
"""Un-profiled cost of the stride-2 3x3 convolutions of ResNet-18's layer3.0 / layer4.0 (B = 100, 64x64 inputs): MIOpen (NHWC implicit GEMM
between transposes) against ee_s2.hip, forward and backward-data; graph-replayed back-to-back launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from eeadv import functional as EF, ops  # noqa: E402

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda:0"


def timeit(fn, iters=30, reps=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("PROBE_EAGER"):  # plain launches, for counter collection
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        return 0.0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / (iters * reps)


FLUSH = torch.zeros(int(os.environ.get("PROBE_FLUSH_MB", "0")) * (1 << 18), device=dev) if os.environ.get("PROBE_FLUSH_MB") else None
if FLUSH is not None:  # cold caches: every call preceded by a sweep over PROBE_FLUSH_MB megabytes, whose own time is subtracted
    _timeit = timeit
    t_flush = _timeit(lambda: FLUSH.add_(1.0))
    print("flush of %d MB: %.1f us" % (FLUSH.numel() >> 18, t_flush))

    def timeit(fn, iters=30, reps=3):
        def both():
            FLUSH.add_(1.0)
            fn()
        return _timeit(both, iters, reps) - t_flush

for cin, cout, hw in ((64, 128, 16), (128, 256, 8), (256, 512, 4)):
    x = torch.randn(B, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    dy = torch.randn(B, cout, hw // 2, hw // 2, device=dev)
    wf, wb = EF._rearranged(w, "s2m_f").contiguous(), EF._rearranged(w, "s2m_b").contiguous()
    if os.environ.get("PROBE_EAGER"):
        timeit(lambda: ops.conv3x3s2_small_fwd(x, wf, cout))
        timeit(lambda: ops.conv3x3s2_small_bwd_data(dy, wb, cin))
        continue
    for mt in ("222222", "111111"):
        os.environ["EEADV_S2_MT"] = mt
        print("   EEADV_S2_MT=%s: forward %6.1f us  backward-data %6.1f us" % (mt, timeit(lambda: ops.conv3x3s2_small_fwd(x, wf, cout)),
                                                                               timeit(lambda: ops.conv3x3s2_small_bwd_data(dy, wb, cin))), flush=True)
    os.environ.pop("EEADV_S2_MT")
    mi_f = timeit(lambda: F.conv2d(x, w, None, 2, 1))
    mi_b = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    print("%3d -> %3d ch %dx%d:  forward MIOpen %6.1f us  ee_s2 %6.1f us   backward-data MIOpen %6.1f us  ee_s2 %6.1f us" % (
        cin, cout, hw, hw, mi_f, timeit(lambda: ops.conv3x3s2_small_fwd(x, wf, cout)), mi_b,
        timeit(lambda: ops.conv3x3s2_small_bwd_data(dy, wb, cin))), flush=True)
