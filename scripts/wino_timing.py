#!/usr/bin/env python3
"""Where a Winograd launch spends its time, from shader-clock stamps inside the kernel (a private build of ee_wino.hip with
-DEE_WINO_TIMING, never the product library): per workgroup, for one multiplying and one producing lane - entry -> loop start (prologue),
loop (and how much of it inside the round bodies, the rest being barrier waits), loop end -> exit (output transform)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from eeadv import functional as EF  # noqa: E402  (loads the product library; its transformed filters are reused)

so = "/tmp/libee_wino_timing.so"
src = os.path.join(ROOT, "edge-enhancement_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc",
                       "-DEE_WINO_TIMING", "-I" + os.path.join(ROOT, "include"), "-I" + src, "-shared", os.path.join(src, "ee_wino.hip"),
                       os.path.join(src, "ee_prof.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.ee_wino3x3_f32.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
lib.ee_wino_timing_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda:0"
for c, hw in ((64, 16), (128, 8)):
    x = torch.randn(B, c, hw, hw, device=dev)
    w = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
    u = EF._rearranged(w, "wino_f").contiguous()
    y = torch.empty_like(x)
    for _ in range(5):
        assert lib.ee_wino3x3_f32(x.data_ptr(), u.data_ptr(), y.data_ptr(), B, c, c, hw, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    nwg = (B if hw == 16 else (B + 1) // 2) * (c // 32)
    st = np.zeros(8 * 2048, np.uint64)
    assert lib.ee_wino_timing_read(st.ctypes.data, st.size) == 0
    st = st.reshape(2048, 8)[:nwg].astype(np.int64)
    for who, o in (("multiplying lane", 0), ("producing lane  ", 4)):
        t0, t1, t2, body = st[:, o], st[:, o + 1], st[:, o + 2], st[:, o + 3]
        print("%3d ch %2dx%-2d %s: prologue %6.0f  loop %6.0f (round bodies %6.0f, barrier waits %6.0f) cycles, median over %d workgroups; rounds %d" % (
            c, hw, hw, who, np.median(t1 - t0), np.median(t2 - t1), np.median(body), np.median(t2 - t1 - body), nwg, c // 8))
    lib.ee_wino_timing_sub(1)  # second pass: the producing lane's even rounds split into (U store | transform | pixel store + loads issued)
    lib.ee_wino3x3_f32(x.data_ptr(), u.data_ptr(), y.data_ptr(), B, c, c, hw, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lib.ee_wino_timing_sub(0)
    st = np.zeros(8 * 2048, np.uint64)
    lib.ee_wino_timing_read(st.ctypes.data, st.size)
    st = st.reshape(2048, 8)[:nwg].astype(np.int64)
    half = c // 16
    print("          producing lane, per EVEN round: U store %5.0f  transform %5.0f  pixel store + load issue %5.0f cycles (each stamp waits for the lane's outstanding LDS ops)" % (
        np.median(st[:, 0]) / half, np.median(st[:, 1]) / half, np.median(st[:, 2]) / half))
