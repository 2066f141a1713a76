#!/usr/bin/env python3
"""Where ee_chain_fwd_f32 spends its time at the bench shape: private builds of ee_chain.hip with -DEE_CHAIN_SKIP=<bits> (1: no edge filter,
2: no low-pass products, 4: no combine / store), graph-replayed back to back.  Never the product library."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from eeadv import _native as N, hfs as HF, ops  # noqa: E402

src = os.path.join(ROOT, "edge-enhancement_amd", "csrc")
B, C, H, W = 100, 3, 64, 64
dev = "cuda:0"
op = HF.HFSOperator(H, W, 8, dev)
wts = ops.EdgeWeights(1.0)
x = torch.rand(B, C, H, W, device=dev)
state = torch.tensor([7, 0, 0, 0], dtype=torch.int64, device=dev)
s = max(int(round((0.8 * H * H) ** 0.5)), 1)
x_in, gate, gx, gy = torch.empty_like(x), torch.empty(B, C, H, W, dtype=torch.uint8, device=dev), torch.empty(B, 1, H, W, device=dev), torch.empty(B, 1, H, W, device=dev)
for skip in (0, 1, 2, 3, 4, 7):
    so = "/tmp/libee_chain_skip%d.so" % skip
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc",
                           "-DEE_CHAIN_SKIP=%d" % skip, "-I" + os.path.join(ROOT, "include"), "-I" + src, "-shared", os.path.join(src, "ee_chain.hip"),
                           os.path.join(src, "ee_prof.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    fn = lib.ee_chain_fwd_f32
    fn.argtypes = N.SIGNATURES["ee_chain_fwd_f32"]
    fn.restype = ctypes.c_int

    def run():
        rc = fn(x.data_ptr(), B, C, H, W, op.chain.data_ptr(), wts.ptr, 0.0, 76 / 255, 1.0, 1, 16 / 255, s, state.data_ptr(), None, None, None,
                x_in.data_ptr(), gate.data_ptr(), gx.data_ptr(), gy.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
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
    print("EE_CHAIN_SKIP=%d (%s): %.2f us per launch" % (skip, ", ".join(n for bit, n in ((1, "no edge filter"), (2, "no low-pass"), (4, "no combine / store")) if skip & bit) or "full kernel",
                                                       1e3 * a.elapsed_time(b) / 200))
