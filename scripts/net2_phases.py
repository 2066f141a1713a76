#!/usr/bin/env python3
"""Where net2_conv2_bwd_mfma_kernel spends its time (batch 50): private builds of ee_net2.hip with -DEE_NET2_SKIP=<bits> (1: no products,
2: no gather, 4: no P store), graph-replayed back to back.  Never the product library."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402

from eeadv import _native as N  # noqa: E402

src = os.path.join(ROOT, "edge-enhancement_amd", "csrc")
B, dev = 50, "cuda:0"
da2, a2 = torch.randn(B, 64, 4, 4, device=dev), torch.rand(B, 64, 4, 4, device=dev)
c2 = torch.randint(0, 4, (B, 64, 4, 4), dtype=torch.uint8, device=dev)
w2, w1 = torch.randn(64, 32, 5, 5, device=dev) * 0.05, torch.randn(32, 1, 5, 5, device=dev)
a1, c1 = torch.rand(B, 32, 12, 12, device=dev), torch.randint(0, 4, (B, 32, 12, 12), dtype=torch.uint8, device=dev)
da1 = torch.empty(B, 32, 12, 12, device=dev)
for skip in (0, 1, 2, 4, 7):
    so = "/tmp/libee_net2_skip%d.so" % skip
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc",
                           "-DEE_NET2_SKIP=%d" % skip, "-I" + os.path.join(ROOT, "include"), "-I" + src, "-shared", os.path.join(src, "ee_net2.hip"),
                           os.path.join(src, "ee_prof.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    fn = lib.ee_net2_conv_bwd_f32
    fn.argtypes = N.SIGNATURES["ee_net2_conv_bwd_f32"]
    fn.restype = ctypes.c_int

    def run():
        assert fn(da2.data_ptr(), a2.data_ptr(), c2.data_ptr(), None, 1.0, w2.data_ptr(), a1.data_ptr(), c1.data_ptr(), w1.data_ptr(), da1.data_ptr(), None, B,
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
    print("EE_NET2_SKIP=%d (%s): %.2f us per launch" % (skip, ", ".join(n for bit, n in ((1, "no products"), (2, "no gather"), (4, "no P store")) if skip & bit) or "full kernel",
                                                       1e3 * a.elapsed_time(b) / 200))
