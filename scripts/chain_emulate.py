"""Lane-level numpy emulation of the MFMA formulation of HighFreqSuppress used by csrc/ee_chain.hip.

v_mfma_f32_16x16x4_f32 (guides/cdna_hip_programming.md): lane l holds A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15];
the 16x16 result sits in 4 registers per lane: D[i = 4 * (l >> 4) + r][j = l & 15].

The four contractions of the low-rank form (csrc/ee_hfs.hip header) are chained THROUGH THE ACCUMULATORS: a result tile is
used as the next product's B (or A) operand as it lies in the registers, and the constant factor on the other side is stored
in the K-permuted order that makes that legal.  This script checks the index algebra against the dense operator in float64
(eeadv/hfs.py: hfs_matrices) before any of it runs on a GPU, and is also the specification of the table layout
(`chain_tables`, mirrored by eeadv.hfs.chain_tables).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "edge-enhancement_amd"))

LANES = np.arange(64)
LI, LG = LANES & 15, LANES >> 4  # (i or j) index, k / row group


def mfma(a, b, acc):
    """a, b: [64] one value per lane; acc: [64, 4].  Returns acc + A @ B in the D layout."""
    A = np.zeros((16, 4), a.dtype)
    Bm = np.zeros((4, 16), a.dtype)
    A[LI, LG] = a
    Bm[LG, LI] = b
    D = A @ Bm
    out = acc.copy()
    for r in range(4):
        out[:, r] += D[4 * LG + r, LI]
    return out


def chain_tables(H, W, r, dtype=np.float64):
    """Fragment-ordered constant operands.  Returns dict of arrays [n_frag, 64] plus dims."""
    from eeadv.hfs import keep_set
    us = np.array([u if u < H / 2 else u - H for u in keep_set(H, r)], dtype=np.float64)
    vs = np.array([v for v in keep_set(W, r) if v <= W // 2], dtype=np.float64)
    kap = np.array([1.0 if (v == 0 or (W % 2 == 0 and v == W // 2)) else 2.0 for v in vs])
    NU, NV = len(us), len(vs)
    assert NU <= 16 and NV <= 8
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    h, w = np.arange(Hp)[:, None], np.arange(Wp)[:, None]
    Ch = np.zeros((Hp, 16)); Sh = np.zeros((Hp, 16)); Cw = np.zeros((Wp, 8)); Sw = np.zeros((Wp, 8))
    Ch[:H, :NU] = np.cos(2 * np.pi * h[:H] * us[None, :] / H)
    Sh[:H, :NU] = np.sin(2 * np.pi * h[:H] * us[None, :] / H)
    Cw[:W, :NV] = np.cos(2 * np.pi * w[:W] * vs[None, :] / W)
    Sw[:W, :NV] = np.sin(2 * np.pi * w[:W] * vs[None, :] / W)
    dv = np.zeros(8); dv[:NV] = kap / W
    T1m = np.concatenate([Cw * dv, Sw * dv], 1)      # [Wp, 16]   PQ = X @ T1m
    CS = np.concatenate([Ch, Sh], 1)                 # [Hp, 32]   R = CS^T @ PQ ;  UV = CS @ EF / H
    T4m = np.concatenate([Cw.T, Sw.T], 0)            # [16, Wp]   y = UV @ T4m
    t1 = np.stack([T1m[4 * s + LG, LI] for s in range(Wp // 4)])
    t2 = np.stack([CS[16 * t + 4 * LG + rr, 16 * mt + LI] for mt in range(2) for t in range(Hp // 16) for rr in range(4)])
    t3 = np.stack([CS[16 * ht + LI, 16 * t + 4 * LG + rr] / H for ht in range(Hp // 16) for t in range(2) for rr in range(4)])
    t4 = np.stack([T4m[4 * LG + rr, 16 * wt + LI] for wt in range(Wp // 16) for rr in range(4)])
    return {"t1": t1.astype(dtype), "t2": t2.astype(dtype), "t3": t3.astype(dtype), "t4": t4.astype(dtype), "Hp": Hp, "Wp": Wp}


def chain_apply(x, T):
    """One plane x [H, W] through the four chained products; returns y [H, W]."""
    H, W = x.shape
    Hp, Wp = T["Hp"], T["Wp"]
    dt = T["t1"].dtype
    xp = np.zeros((Hp, Wp), dt)
    xp[:H, :W] = x
    HT, WT = Hp // 16, Wp // 16
    # S1: PQ tile per h-tile; A = x[16 mt + i][4 s + g]
    pq = [np.zeros((64, 4), dt) for _ in range(HT)]
    for mt in range(HT):
        for s in range(Wp // 4):
            pq[mt] = mfma(xp[16 * mt + LI, 4 * s + LG], T["t1"][s], pq[mt])
    # S2: R (cos rows, sin rows); B = PQ accumulators as they lie: k-step (t, r) <-> h = 16 t + 4 g + r
    rc = [np.zeros((64, 4), dt) for _ in range(2)]
    for mt in range(2):
        for t in range(HT):
            for rr in range(4):
                rc[mt] = mfma(T["t2"][(mt * HT + t) * 4 + rr], pq[t][:, rr], rc[mt])
    Rc, Rs = rc
    part = LANES ^ 8
    lo = (LI < 8)[:, None]
    EFc = np.where(lo, Rc - Rs[part], Rc + Rs[part])
    EFs = np.where(lo, Rs + Rc[part], Rs - Rc[part])
    EF = [EFc, EFs]
    # S3': UV^T tile per h-tile; A = EF registers as they lie: k-step (t, r) <-> k = 16 t + 4 g + r
    uvT = [np.zeros((64, 4), dt) for _ in range(HT)]
    for ht in range(HT):
        for t in range(2):
            for rr in range(4):
                uvT[ht] = mfma(EF[t][:, rr], T["t3"][(ht * 2 + t) * 4 + rr], uvT[ht])
    # S4: y tile (ht, wt); A = UV^T registers as they lie: k-step r <-> n = 4 g + r
    y = np.zeros((Hp, Wp), dt)
    for ht in range(HT):
        for wt in range(WT):
            acc = np.zeros((64, 4), dt)
            for rr in range(4):
                acc = mfma(uvT[ht][:, rr], T["t4"][wt * 4 + rr], acc)
            for rr in range(4):
                y[16 * ht + 4 * LG + rr, 16 * wt + LI] = acc[:, rr]
    return y[:H, :W]


if __name__ == "__main__":
    from eeadv.hfs import hfs_matrices
    rng = np.random.RandomState(0)
    for (H, W, r) in [(64, 64, 8), (28, 28, 4), (32, 32, 4), (16, 16, 2), (64, 48, 8)]:
        x = rng.rand(H, W)
        Ar, Ai, B1, B2 = hfs_matrices(H, W, r)
        want = Ar @ x @ B1 + Ai @ x @ B2
        got = chain_apply(x, chain_tables(H, W, r))
        got32 = chain_apply(x.astype(np.float32), chain_tables(H, W, r, np.float32))
        print("%dx%d r=%d: f64 err %.2e, f32 err %.2e" % (H, W, r, np.abs(got - want).max(), np.abs(got32 - want).max()))
        assert np.abs(got - want).max() < 1e-12


# ---- the band-parallel variant for planes that do not fit one wavefront's registers / the LDS (csrc/ee_hfs_mfma.hip) --------------
def big_tables(H, W, r, dtype=np.float64):
    from eeadv.hfs import keep_set
    us = np.array([u if u < H / 2 else u - H for u in keep_set(H, r)], dtype=np.float64)
    vs = np.array([v for v in keep_set(W, r) if v <= W // 2], dtype=np.float64)
    kap = np.array([1.0 if (v == 0 or (W % 2 == 0 and v == W // 2)) else 2.0 for v in vs])
    NU, NV = len(us), len(vs)
    assert NU <= 32 and NV <= 16
    NUp = 16 if NU <= 16 else 32
    MT2 = 2 * NUp // 16
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    h, w = np.arange(H)[:, None], np.arange(W)[:, None]
    Ch = np.zeros((Hp, NUp)); Sh = np.zeros((Hp, NUp)); Cw = np.zeros((Wp, 16)); Sw = np.zeros((Wp, 16))
    Ch[:H, :NU] = np.cos(2 * np.pi * h * us[None, :] / H)
    Sh[:H, :NU] = np.sin(2 * np.pi * h * us[None, :] / H)
    Cw[:W, :NV] = np.cos(2 * np.pi * w * vs[None, :] / W)
    Sw[:W, :NV] = np.sin(2 * np.pi * w * vs[None, :] / W)
    dv = np.zeros(16); dv[:NV] = kap / W
    T1m = np.concatenate([Cw * dv, Sw * dv], 1)   # [Wp, 32]
    CS = np.concatenate([Ch, Sh], 1)              # [Hp, 2 NUp]
    T4m = np.concatenate([Cw.T, Sw.T], 0)         # [32, Wp]
    t1 = np.stack([T1m[4 * s + LG, 16 * nt + LI] for s in range(Wp // 4) for nt in range(2)])
    t2 = np.stack([CS[16 * t + 4 * LG + rr, 16 * mt + LI] for t in range(Hp // 16) for mt in range(MT2) for rr in range(4)])
    t3 = np.stack([CS[16 * ht + LI, 16 * kt + 4 * LG + rr] / H for ht in range(Hp // 16) for kt in range(MT2) for rr in range(4)])
    t4 = np.stack([T4m[16 * nt + 4 * LG + rr, 16 * wt + LI] for wt in range(Wp // 16) for nt in range(2) for rr in range(4)])
    return {"t1": t1.astype(dtype), "t2": t2.astype(dtype), "t3": t3.astype(dtype), "t4": t4.astype(dtype), "Hp": Hp, "Wp": Wp, "MT2": MT2}


def big_apply(x, T):
    H, W = x.shape
    Hp, Wp, MT2 = T["Hp"], T["Wp"], T["MT2"]
    dt = T["t1"].dtype
    xp = np.zeros((Hp, Wp), dt)
    xp[:H, :W] = x
    NB, WT, half = Hp // 16, Wp // 16, MT2 // 2
    Rsum = [[np.zeros((64, 4), dt) for _ in range(2)] for _ in range(MT2)]
    for b in range(NB):  # one wavefront per 16-row band
        pq = [np.zeros((64, 4), dt) for _ in range(2)]
        for s in range(Wp // 4):
            for nt in range(2):
                pq[nt] = mfma(xp[16 * b + LI, 4 * s + LG], T["t1"][s * 2 + nt], pq[nt])
        for mt in range(MT2):
            for nt in range(2):
                acc = np.zeros((64, 4), dt)
                for rr in range(4):
                    acc = mfma(T["t2"][(b * MT2 + mt) * 4 + rr], pq[nt][:, rr], acc)
                Rsum[mt][nt] += acc  # the cross-band reduction (through LDS on the device)
    EF = [[None, None] for _ in range(MT2)]
    for j in range(half):
        a, c, bb, d = Rsum[j][0], Rsum[j][1], Rsum[half + j][0], Rsum[half + j][1]
        EF[j][0], EF[j][1] = a - d, c + bb
        EF[half + j][0], EF[half + j][1] = bb + c, d - a
    y = np.zeros((Hp, Wp), dt)
    for b in range(NB):
        uvT = [np.zeros((64, 4), dt) for _ in range(2)]
        for nt in range(2):
            for kt in range(MT2):
                for rr in range(4):
                    uvT[nt] = mfma(EF[kt][nt][:, rr], T["t3"][(b * MT2 + kt) * 4 + rr], uvT[nt])
        for wt in range(WT):
            acc = np.zeros((64, 4), dt)
            for nt in range(2):
                for rr in range(4):
                    acc = mfma(uvT[nt][:, rr], T["t4"][(wt * 2 + nt) * 4 + rr], acc)
            for rr in range(4):
                y[16 * b + 4 * LG + rr, 16 * wt + LI] = acc[:, rr]
    return y[:H, :W]


if __name__ == "__main__":
    from eeadv.hfs import hfs_matrices
    rng = np.random.RandomState(1)
    for (H, W, r) in [(224, 224, 16), (64, 64, 8), (7, 9, 2), (96, 128, 12), (28, 28, 4)]:
        x = rng.rand(H, W)
        Ar, Ai, B1, B2 = hfs_matrices(H, W, r)
        want = Ar @ x @ B1 + Ai @ x @ B2
        got = big_apply(x, big_tables(H, W, r))
        print("band variant %dx%d r=%d: f64 err %.2e" % (H, W, r, np.abs(got - want).max()))
        assert np.abs(got - want).max() < 1e-11
