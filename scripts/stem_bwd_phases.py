#!/usr/bin/env python3
"""Where stem_bn_pool_bwd_data_kernel (ee_stem.hip) spends its time at the bench shape (eval mode, batch 100): private builds with
-DEE_STEMB_SKIP=<bits> (1: no products, 2: no frame arithmetic, 4: no global loads in the loop, 8: no weight staging), graph-replayed back to
back.  Never the product library."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import torch  # noqa: E402

from eeadv import _native as N, ops  # noqa: E402

src = os.path.join(ROOT, "edge-enhancement_amd", "csrc")
dev = "cuda:0"
B = 100
img, w = torch.rand(B, 3, 64, 64, device=dev), torch.randn(64, 3, 7, 7, device=dev) / 12
x = ops.stem7x7s2_fwd(img, w)
gamma, beta, rm, rv = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev), torch.zeros(64, device=dev), torch.ones(64, device=dev)
y, code, _, _ = ops.bn_relu_pool_fwd(x, gamma, beta, rm, rv, 0.1, 1e-5, False)
dyp, dx = torch.randn_like(y), torch.empty(B, 3, 64, 64, device=dev)
for skip in (0, 1, 2, 4, 8, 3, 6, 7, 15):
    so = "/tmp/libee_stem_s%d.so" % skip
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc",
                           "-DEE_STEMB_SKIP=%d" % skip, "-I" + os.path.join(ROOT, "include"), "-I" + src, "-shared",
                           os.path.join(src, "ee_stem.hip"), os.path.join(src, "ee_prof.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    fn = lib.ee_stem_bn_pool_bwd_data_f32
    fn.argtypes = N.SIGNATURES["ee_stem_bn_pool_bwd_data_f32"]
    fn.restype = ctypes.c_int

    def run():
        assert fn(dyp.data_ptr(), None, code.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, None, rm.data_ptr(), rv.data_ptr(), 1e-5, 0,
                  None, 0, None, None, w.data_ptr(), dx.data_ptr(), B, 64, 64, 64, torch.cuda.current_stream().cuda_stream) == 0
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            run()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    names = ((1, "no products"), (2, "no frame arithmetic"), (4, "no loads in the loop"), (8, "no weight staging"))
    print("EE_STEMB_SKIP=%2d (%s): %.2f us per launch" % (skip, ", ".join(n for bit, n in names if skip & bit) or "full kernel", 1e3 * a.elapsed_time(b) / 80), flush=True)
