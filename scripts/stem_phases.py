#!/usr/bin/env python3
"""Where stem_fwd_mfma_kernel spends its time at the bench shape ([100,3,64,64] -> [100,64,32,32]): private builds of ee_conv.hip with
-DEE_STEM_SKIP=<bits> (1: no products, 2: no result stores, 8: products without their LDS operand reads), graph-replayed back to back.
Never the product library."""
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
B = 100
img, w, y = torch.rand(B, 3, 64, 64, device=dev), torch.randn(64, 3, 7, 7, device=dev) / 12, torch.empty(B, 64, 32, 32, device=dev)
for rows in (2,):  # output rows per unit (fixed in the kernel)
    for skip in (0, 1, 2, 3, 8, 10):
        so = "/tmp/libee_conv_r%d_s%d.so" % (rows, skip)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc",
                               "-DEE_STEM_SKIP=%d" % skip, "-I" + os.path.join(ROOT, "include"), "-I" + src, "-shared",
                               os.path.join(src, "ee_conv.hip"), os.path.join(src, "ee_prof.hip"), "-o", so])
        lib = ctypes.CDLL(so)
        fn = lib.ee_stem7x7s2_fwd_f32
        fn.argtypes = N.SIGNATURES["ee_stem7x7s2_fwd_f32"]
        fn.restype = ctypes.c_int

        def run():
            assert fn(img.data_ptr(), w.data_ptr(), y.data_ptr(), B, 64, 64, 64, torch.cuda.current_stream().cuda_stream) == 0
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(30):
                run()
        g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        print("rows %d EE_STEM_SKIP=%d (%s): %.2f us per launch" % (rows, skip, ", ".join(n for bit, n in ((1, "no products"), (2, "no stores"), (8, "products without LDS operand reads")) if skip & bit) or "full kernel",
                                                                   1e3 * a.elapsed_time(b) / 120), flush=True)
