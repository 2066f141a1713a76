#!/usr/bin/env python3
"""Where the one-launch head backward of Net_2 (ee_dense.hip, PRE 2) spends its time at batch 50: private builds with -DEE_DENSE_SKIP=<bits>
(1: no gradient forming while staging, 2: no logits products, 4: no main loop), graph-replayed back to back.  Never the product library."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402

from eeadv import _native as N  # noqa: E402

src = os.path.join(ROOT, "edge-enhancement_amd", "csrc")
dev = "cuda:0"
for B in (50,):
    z1, w1, w2 = torch.randn(B, 1024, device=dev), torch.randn(1024, 1024, device=dev) / 32, torch.randn(10, 1024, device=dev) / 32
    b2, y, dx = torch.randn(10, device=dev), torch.randint(0, 10, (B,), device=dev), torch.empty(B, 1024, device=dev)
    for skip in (0, 1, 2, 3, 4, 7):
        so = "/tmp/libee_dense_skip%d.so" % skip
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc",
                               "-DEE_DENSE_SKIP=%d" % skip, "-I" + os.path.join(ROOT, "include"), "-I" + src, "-shared", os.path.join(src, "ee_dense.hip"),
                               os.path.join(src, "ee_prof.hip"), "-o", so])
        lib = ctypes.CDLL(so)
        fn = lib.ee_net2_head_bwd_f32
        fn.argtypes = N.SIGNATURES["ee_net2_head_bwd_f32"]
        fn.restype = ctypes.c_int

        def run():
            assert fn(z1.data_ptr(), w2.data_ptr(), b2.data_ptr(), y.data_ptr(), 1.0, w1.data_ptr(), dx.data_ptr(), None, B, 1024, 10,
                      torch.cuda.current_stream().cuda_stream) == 0
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50):
                run()
        g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        print("B %d EE_DENSE_SKIP=%d (%s): %.2f us per launch" % (B, skip, ", ".join(n for bit, n in ((1, "no gradient forming"), (2, "no logits products"), (4, "no main loop")) if skip & bit) or "full kernel",
                                                                 1e3 * a.elapsed_time(b) / 200))
