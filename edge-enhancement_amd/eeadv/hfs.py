"""HighFreqSuppress (utils/core.py:15-55) as a fixed REAL linear operator.

The reference computes  y = irfft(rfft(x, 2, onesided=False) * mask, 2, onesided=False)  with a 0/1 mask
that keeps the frequencies -r .. r-1 on both axes (core.py:23-42).  The mask is not Hermitian-symmetric,
and the old C2R transform reads only the first W//2+1 columns of its input, so the
operator equals  irfft2((fft2(x) * mask)[..., :W//2+1], s=(H, W)).  Expanding that expression gives

    y = Ar @ x @ B1 + Ai @ x @ B2                                                   (per image plane)

    Ar[h,h'] = 1/H sum_{u in keep_H} cos(2 pi u (h-h')/H)     Ai[h,h'] = 1/H sum_u sin(2 pi u (h-h')/H)
    B1[w',w] = 1/W sum_{v<=W/2, v in keep_W} k_v cos(2 pi v (w'-w)/W)      B2 likewise with sin
    k_0 = 1, k_{W/2} = 1 (even W), otherwise 2

so the whole filter is two skinny real contractions per plane instead of a complex FFT round trip with a
[B,C,H,W] complex intermediate.  The matrices are built once per (H, W, r) in float64 and cast to fp32.
PARITY UNPINNED against the reference (torch.rfft no longer exists); pinned to the fft restatement in
oracle/ref_path.py by tests/test_host_cpu.py.
"""
import numpy as np
import torch


def keep_set(n, r):
    """Kept frequencies along an axis of length n (core.py:23-39): -r .. r-1 (even n) or -r .. r (odd n)."""
    hi = r if n % 2 == 0 else r + 1
    return sorted({u % n for u in range(-r, hi)})


def hfs_matrices(H, W, r):
    """float64 (Ar, Ai, B1, B2)."""
    # signed frequency of each kept row index (indices >= H/2 are negative frequencies)
    ku = np.array([u if u < H / 2 else u - H for u in keep_set(H, r)], dtype=np.float64)
    dh = np.arange(H)[:, None] - np.arange(H)[None, :]
    Ar = np.zeros((H, H))
    Ai = np.zeros((H, H))
    for u in ku:
        Ar += np.cos(2 * np.pi * u * dh / H)
        Ai += np.sin(2 * np.pi * u * dh / H)
    Ar /= H
    Ai /= H
    kept_w = [v for v in keep_set(W, r) if v <= W // 2]
    dw = np.arange(W)[:, None] - np.arange(W)[None, :]  # w' - w
    B1 = np.zeros((W, W))
    B2 = np.zeros((W, W))
    for v in kept_w:
        k = 1.0 if (v == 0 or (W % 2 == 0 and v == W // 2)) else 2.0
        B1 += k * np.cos(2 * np.pi * v * dw / W)
        B2 += k * np.sin(2 * np.pi * v * dw / W)
    B1 /= W
    B2 /= W
    return Ar, Ai, B1, B2


def lowrank_tables(H, W, r):
    """The device table of ee_hfs_f32 (layout documented in include/eeadv.h): cos / sin factors of the operator's
    exact low-rank form  y = U Cw^T + V Sw^T  (csrc/ee_hfs.hip).  Returns (float32 array, NU, NV)."""
    us = np.array([u if u < H / 2 else u - H for u in keep_set(H, r)], dtype=np.float64)
    vs = np.array([v for v in keep_set(W, r) if v <= W // 2], dtype=np.float64)
    kap = np.array([1.0 if (v == 0 or (W % 2 == 0 and v == W // 2)) else 2.0 for v in vs])
    NU, NV = len(us), len(vs)
    up, vp, Hp, Wp = (NU + 3) & ~3, (NV + 3) & ~3, (H + 3) & ~3, (W + 3) & ~3
    hs = Hp + 4
    h, w = np.arange(H)[:, None], np.arange(W)[:, None]
    Ch, Sh = np.cos(2 * np.pi * h * us[None, :] / H), np.sin(2 * np.pi * h * us[None, :] / H)  # [H, NU]
    Cw, Sw = np.cos(2 * np.pi * w * vs[None, :] / W), np.sin(2 * np.pi * w * vs[None, :] / W)  # [W, NV]
    cwT, swT = np.zeros((vp, Wp)), np.zeros((vp, Wp))
    cwT[:NV, :W], swT[:NV, :W] = Cw.T, Sw.T
    chT, shT = np.zeros((up, hs)), np.zeros((up, hs))
    chT[:NU, :H], shT[:NU, :H] = Ch.T, Sh.T
    chN, shN = np.zeros((H, up)), np.zeros((H, up))
    chN[:, :NU], shN[:, :NU] = Ch, Sh
    dv = np.zeros(vp)
    dv[:NV] = kap / W
    flat = np.concatenate([a.reshape(-1) for a in (cwT, swT, chT, shT, chN, shN, dv)]).astype(np.float32)
    return flat, NU, NV


def chain_tables(H, W, r):
    """The constant operands of the low-pass operator in v_mfma_f32_16x16x4_f32 fragment order, for the fused front-end
    kernels (csrc/ee_chain.hip).  Layout = scripts/chain_emulate.py (which checks the index algebra against hfs_matrices):
    t1 [Wp/4][64] | t2 [2][Hp/16][4][64] | t3 [Hp/16][2][4][64] | t4 [Wp/16][4][64], lane l <-> (l & 15, l >> 4).
    cos / sin column blocks are padded to 8 (v) and 16 (u) entries with zeros, Hp / Wp = H / W rounded up to 16."""
    us = np.array([u if u < H / 2 else u - H for u in keep_set(H, r)], dtype=np.float64)
    vs = np.array([v for v in keep_set(W, r) if v <= W // 2], dtype=np.float64)
    kap = np.array([1.0 if (v == 0 or (W % 2 == 0 and v == W // 2)) else 2.0 for v in vs])
    NU, NV = len(us), len(vs)
    if NU > 16 or NV > 8 or H > 64 or W > 64:
        raise ValueError("chain_tables: %dx%d r=%d is outside the fused kernels' shape class" % (H, W, r))
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    lanes = np.arange(64)
    li, lg = lanes & 15, lanes >> 4
    h, w = np.arange(H)[:, None], np.arange(W)[:, None]
    Ch, Sh, Cw, Sw = np.zeros((Hp, 16)), np.zeros((Hp, 16)), np.zeros((Wp, 8)), np.zeros((Wp, 8))
    Ch[:H, :NU], Sh[:H, :NU] = np.cos(2 * np.pi * h * us[None, :] / H), np.sin(2 * np.pi * h * us[None, :] / H)
    Cw[:W, :NV], Sw[:W, :NV] = np.cos(2 * np.pi * w * vs[None, :] / W), np.sin(2 * np.pi * w * vs[None, :] / W)
    dv = np.zeros(8)
    dv[:NV] = kap / W
    T1 = np.concatenate([Cw * dv, Sw * dv], 1)   # [Wp, 16]: [P | Q] = X T1
    CS = np.concatenate([Ch, Sh], 1)             # [Hp, 32]: R = CS^T [P | Q];  [U | V] = CS EF / H
    T4 = np.concatenate([Cw.T, Sw.T], 0)         # [16, Wp]: y = [U | V] T4
    t1 = [T1[4 * s + lg, li] for s in range(Wp // 4)]
    t2 = [CS[16 * t + 4 * lg + rr, 16 * mt + li] for mt in range(2) for t in range(Hp // 16) for rr in range(4)]
    t3 = [CS[16 * ht + li, 16 * t + 4 * lg + rr] / H for ht in range(Hp // 16) for t in range(2) for rr in range(4)]
    t4 = [T4[4 * lg + rr, 16 * wt + li] for wt in range(Wp // 16) for rr in range(4)]
    return np.ascontiguousarray(np.concatenate(t1 + t2 + t3 + t4), dtype=np.float32)


def band_tables(H, W, r):
    """Fragment-ordered constant operands of ee_hfs_mfma_f32 (planes up to 256 x 256; csrc/ee_hfs_mfma.hip).  Layout =
    scripts/chain_emulate.py big_tables: t1 [Wp/4][2][64] | t2 [Hp/16][MT2][4][64] | t3 [Hp/16][MT2][4][64] | t4 [Wp/16][2][4][64];
    cos / sin blocks padded to 16 column frequencies and NUp = 16 or 32 row frequencies.  Returns (float32 array, NUp)."""
    us = np.array([u if u < H / 2 else u - H for u in keep_set(H, r)], dtype=np.float64)
    vs = np.array([v for v in keep_set(W, r) if v <= W // 2], dtype=np.float64)
    kap = np.array([1.0 if (v == 0 or (W % 2 == 0 and v == W // 2)) else 2.0 for v in vs])
    NU, NV = len(us), len(vs)
    if NU > 32 or NV > 16 or H > 256 or W > 256:
        raise ValueError("band_tables: %dx%d r=%d is outside ee_hfs_mfma_f32's shape class" % (H, W, r))
    NUp = 16 if NU <= 16 else 32
    MT2 = 2 * NUp // 16
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    lanes = np.arange(64)
    li, lg = lanes & 15, lanes >> 4
    h, w = np.arange(H)[:, None], np.arange(W)[:, None]
    Ch, Sh, Cw, Sw = np.zeros((Hp, NUp)), np.zeros((Hp, NUp)), np.zeros((Wp, 16)), np.zeros((Wp, 16))
    Ch[:H, :NU], Sh[:H, :NU] = np.cos(2 * np.pi * h * us[None, :] / H), np.sin(2 * np.pi * h * us[None, :] / H)
    Cw[:W, :NV], Sw[:W, :NV] = np.cos(2 * np.pi * w * vs[None, :] / W), np.sin(2 * np.pi * w * vs[None, :] / W)
    dv = np.zeros(16)
    dv[:NV] = kap / W
    T1 = np.concatenate([Cw * dv, Sw * dv], 1)   # [Wp, 32]
    CS = np.concatenate([Ch, Sh], 1)             # [Hp, 2 NUp]
    T4 = np.concatenate([Cw.T, Sw.T], 0)         # [32, Wp]
    t1 = [T1[4 * s + lg, 16 * nt + li] for s in range(Wp // 4) for nt in range(2)]
    t2 = [CS[16 * t + 4 * lg + rr, 16 * mt + li] for t in range(Hp // 16) for mt in range(MT2) for rr in range(4)]
    t3 = [CS[16 * ht + li, 16 * kt + 4 * lg + rr] / H for ht in range(Hp // 16) for kt in range(MT2) for rr in range(4)]
    t4 = [T4[16 * nt + 4 * lg + rr, 16 * wt + li] for wt in range(Wp // 16) for nt in range(2) for rr in range(4)]
    return np.ascontiguousarray(np.concatenate(t1 + t2 + t3 + t4), dtype=np.float32), NUp


class HFSOperator:
    """Device-resident factors of the operator (which equals its own adjoint).  ROCm planes up to 64x64 go through the
    single-launch LDS kernel ee_hfs_f32, planes up to 256x256 (ImageNet 224x224, r = 16) through the matrix-core kernel
    ee_hfs_mfma_f32; anything larger and the CPU plumbing path use the dense form."""

    def __init__(self, H, W, r, device):
        Ar, Ai, B1, B2 = hfs_matrices(H, W, r)
        self.kernel = None
        if torch.device(device).type == "cuda":
            flat, NU, NV = lowrank_tables(H, W, r)
            if H <= 64 and W <= 64 and NU <= 16 and NV <= 8:
                self.kernel = (torch.from_numpy(flat).to(device), NU, NV)
        self.chain = None  # fragment-ordered factors of the fused front-end kernels (ee_chain.hip), where the shape allows
        if self.kernel is not None:
            self.chain = torch.from_numpy(chain_tables(H, W, r)).to(device)
        self.mfma = None   # ... and of the band kernel for larger planes (ee_hfs_mfma.hip)
        if torch.device(device).type == "cuda" and H <= 256 and W <= 256:
            try:
                flat, nu_pad = band_tables(H, W, r)
                self.mfma = (torch.from_numpy(flat).to(device), nu_pad)
            except ValueError:
                pass
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=torch.float32)
        self.H, self.W = H, W
        self.Bcat = f(np.concatenate([B1, B2], axis=1))      # [W, 2W]   x @ Bcat = [x B1 | x B2]
        self.Ar, self.Ai = f(Ar), f(Ai)                      # [H, H]
        # adjoint: g_x = Ar^T g B1^T + Ai^T g B2^T
        self.BcatT = f(np.concatenate([B1.T, B2.T], axis=1))  # [W, 2W]
        self.ArT, self.AiT = f(Ar.T), f(Ai.T)

    def _apply(self, x, Bcat, Al, Ar_):
        B, C, H, W = x.shape
        if (H, W) != (self.H, self.W):
            raise ValueError("HighFreqSuppress built for %dx%d, got %dx%d" % (self.H, self.W, H, W))
        n = B * C
        t = torch.mm(x.reshape(n * H, W), Bcat).view(n, H, 2 * W)
        y = torch.bmm(Al.unsqueeze(0).expand(n, H, H), t[:, :, :W])
        y = torch.baddbmm(y, Ar_.unsqueeze(0).expand(n, H, H), t[:, :, W:])
        return y.view(B, C, H, W)

    @property
    def _use_lds_kernel(self):
        """The VALU / LDS kernel for small planes (28 x 28: 3.7 us against 5.8), the matrix-core band kernel from 48 x 48 up
        (64 x 64, B = 100: 8.4 us against 10.4; B = 1600: 54 against 71; scripts/hfs_bench.py)."""
        return self.kernel is not None and (self.mfma is None or self.H * self.W < 48 * 48)

    @property
    def fused_square(self):
        """True when Add_Square can be fused into the low-pass kernel's load / store (one launch each way)."""
        return self.kernel is not None or self.mfma is not None

    def forward(self, x):
        if x.is_cuda and self._use_lds_kernel:
            from . import ops
            return ops.hfs(x.contiguous(), *self.kernel)
        if x.is_cuda and self.mfma is not None:
            from . import ops
            return ops.hfs_mfma(x.contiguous(), *self.mfma)
        return self._apply(x, self.Bcat, self.Ar, self.Ai)

    def adjoint(self, g):
        if g.is_cuda and self._use_lds_kernel:
            from . import ops
            return ops.hfs(g.contiguous(), *self.kernel)  # self-adjoint
        if g.is_cuda and self.mfma is not None:
            from . import ops
            return ops.hfs_mfma(g.contiguous(), *self.mfma)
        return self._apply(g, self.BcatT, self.ArT, self.AiT)

    def forward_square(self, x, eps, draws):
        """F(add_square(x)) in one launch."""
        from . import ops
        if self._use_lds_kernel:
            t, NU, NV = self.kernel
            return ops.hfs(x.contiguous(), t, NU, NV, 1, None, eps, draws["stripe"], draws["sq_sign"], draws["sq_pos"], draws["sq_size"])
        t, nu_pad = self.mfma
        return ops.hfs_mfma(x.contiguous(), t, nu_pad, 1, None, eps, draws["stripe"], draws["sq_sign"], draws["sq_pos"], draws["sq_size"])

    def backward_square(self, g, x, eps, draws):
        """F(g) * d add_square/dx (x): the backward of forward_square, one launch."""
        from . import ops
        if self._use_lds_kernel:
            t, NU, NV = self.kernel
            return ops.hfs(g.contiguous(), t, NU, NV, 2, x, eps, draws["stripe"], draws["sq_sign"], draws["sq_pos"], draws["sq_size"])
        t, nu_pad = self.mfma
        return ops.hfs_mfma(g.contiguous(), t, nu_pad, 2, x, eps, draws["stripe"], draws["sq_sign"], draws["sq_pos"], draws["sq_size"])


class _HFSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        ctx.op = op
        return op.forward(x)

    @staticmethod
    def backward(ctx, g):
        return ctx.op.adjoint(g.contiguous()), None


def hfs_apply(x, op):
    return _HFSFn.apply(x, op)


class _SquareHFSFn(torch.autograd.Function):
    """hfs(add_square(x)) with the Add_Square arithmetic fused into the low-pass kernel's load / store stages."""

    @staticmethod
    def forward(ctx, x, op, eps, stripe, sq_sign, sq_pos, sq_size):
        x = x.contiguous()
        draws = {"stripe": stripe, "sq_sign": sq_sign, "sq_pos": sq_pos, "sq_size": sq_size}
        ctx.save_for_backward(x, stripe, sq_sign, sq_pos, sq_size)
        ctx.op, ctx.eps = op, eps
        return op.forward_square(x, eps, draws)

    @staticmethod
    def backward(ctx, g):
        x, stripe, sq_sign, sq_pos, sq_size = ctx.saved_tensors
        draws = {"stripe": stripe, "sq_sign": sq_sign, "sq_pos": sq_pos, "sq_size": sq_size}
        return ctx.op.backward_square(g.contiguous(), x, ctx.eps, draws), None, None, None, None, None, None


def square_hfs_apply(x, op, eps, draws):
    return _SquareHFSFn.apply(x, op, float(eps), draws["stripe"], draws["sq_sign"], draws["sq_pos"], draws["sq_size"])
