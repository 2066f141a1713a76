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


class HFSOperator:
    """Device-resident factors of the operator and of its adjoint."""

    def __init__(self, H, W, r, device):
        Ar, Ai, B1, B2 = hfs_matrices(H, W, r)
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

    def forward(self, x):
        return self._apply(x, self.Bcat, self.Ar, self.Ai)

    def adjoint(self, g):
        return self._apply(g, self.BcatT, self.ArT, self.AiT)


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
