// ee_hfs.hip - HighFreqSuppress (utils/core.py:15-55) as ONE kernel per direction, optionally fused with Add_Square.
//
// The reference's filter, irfft(rfft(x, 2, onesided=False) * mask, 2, onesided=False) with a mask that keeps the
// frequencies -r..r-1 on both axes, is a fixed REAL, SELF-ADJOINT linear operator (eeadv/hfs.py derives it):
//     y = Ar x B1 + Ai x B2,     Ar/Ai built from the NU kept row frequencies, B1/B2 from the NV kept half-spectrum
// column frequencies.  Ar, Ai have rank NU (16 for 64x64, r=8) and B1, B2 rank 2 NV, so the dense form (2 MFLOP per
// 64x64 plane, three rocBLAS launches and a [B,C,H,2W] intermediate in HBM) collapses to five skinny contractions
// that live entirely in LDS (0.4 MFLOP per plane):
//     P = x Cw, Q = x Sw                     [H x NV]   cos / sin projections along W, scaled by kappa_v / W
//     a = Ch^T P, b = Sh^T P, c = Ch^T Q, d = Sh^T Q          [NU x NV]
//     U = (Ch (a - d) + Sh (b + c)) / H,  V = (Ch (b + c) + Sh (d - a)) / H         [H x NV]
//     y = U Cw^T + V Sw^T                                                              [H x W]
// One 256-thread workgroup owns one [H,W] plane (H, W <= 64): HBM traffic is the algorithmic minimum, read x + write y
// = 8 B per element.  Because the operator is self-adjoint the same kernel is its own backward.
// Fusions: `sq_mode 1` applies Add_Square (core.py:636-655) to the input while it is loaded (forward of the *_square
// models: hfs(add_square(x))); `sq_mode 2` multiplies the result by d add_square / dx evaluated at sq_x (their backward).
//
// Parity: UNPINNED against the reference (torch.rfft no longer exists); agrees with the FFT restatement in
// oracle/ref_path.py to fp32 rounding (tests/test_gpu_path.py).  The contraction order differs from rocBLAS's, so this
// kernel is tolerance-level (1e-6), not bit-level, like every floating-point result that passes through an FFT.
#include "ee_common.hpp"
#include "ee_square.hpp"

namespace {

using namespace ee;

struct HfsDims {
    int H, W, Hp, Wp;  // Hp / Wp: H / W rounded up to a multiple of 4 (float4 loop bounds)
    int NU, NV;        // kept row frequencies / kept half-spectrum column frequencies (both padded to x4 with zeros)
    float inv_h;
};

// LDS carve-up (floats)
struct HfsLds {
    int x, cwT, swT, chT, shT, chN, shN, dv, m1T, m2T, e, u, v, total;
    int xs, hs;  // row strides of the planes read row-per-lane with 16-B LDS reads: +4 floats so that consecutive rows
                 // start 4 banks apart (a 256-B stride would put all 16 lanes of a ds_read_b128 group on the same banks)
    __host__ __device__ explicit HfsLds(const HfsDims &d) {
        xs = d.Wp + 4;
        hs = d.Hp + 4;
        int o = 0;
        x = o;   o += d.H * xs;
        cwT = o; o += d.NV * d.Wp;
        swT = o; o += d.NV * d.Wp;
        chT = o; o += d.NU * hs;
        shT = o; o += d.NU * hs;
        chN = o; o += d.H * d.NU;
        shN = o; o += d.H * d.NU;
        dv = o;  o += d.NV;
        m1T = o; o += d.NV * hs;
        m2T = o; o += d.NV * hs;
        e = o;   o += 4 * d.NV * d.NU;  // E1T, E2T, F1T, F2T as [NV][NU]
        u = o;   o += d.H * d.NV;
        v = o;   o += d.H * d.NV;
        total = o;
    }
};

__device__ __forceinline__ float dot4(const float4 a, const float4 b, float acc) {
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    return fmaf(a.w, b.w, acc);
}

// SQ 0: plain, 1: Add_Square on load, 2: times d(Add_Square)/dx on store.  HS_/WS_/NUS_/NVS_: compile-time plane and
// table sizes (0 = take them from `d` at run time); the two shapes of the reference configs (64x64 r=8, 28x28 r=4) are
// instantiated so that every contraction loop has a constant trip count and is fully unrolled / software-pipelined.
template <int SQ, int HS_, int WS_, int NUS_, int NVS_>
__global__ __launch_bounds__(kBlock) void hfs_kernel(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ tables,
                                                     HfsDims d, const float *__restrict__ sq_x, SquareArgs sq) {
    extern __shared__ __align__(16) float lds[];
    if (HS_) {
        d.H = HS_; d.W = WS_; d.Hp = (HS_ + 3) & ~3; d.Wp = (WS_ + 3) & ~3; d.NU = NUS_; d.NV = NVS_;
    }
    const HfsLds L(d);
    const int H = d.H, W = d.W, Hp = d.Hp, Wp = d.Wp, NU = d.NU, NV = d.NV;
    (void)Hp;
    const int plane = blockIdx.x;
    const int c = plane % sq.C;  // sq.C always carries the channel count
    const float *src = in + static_cast<size_t>(plane) * H * W;
    float *sx = lds + L.x;
    SquarePlane pl{};
    const float *stripe_row = nullptr;
    if (SQ != 0) {
        pl = square_plane(sq, c);
        stripe_row = sq.stripe + static_cast<size_t>(plane) * W;  // stripe[b][c][0][w]
    }

    // ---- load: tables (contiguous block of the device buffer, same layout as LDS from cwT to dv) and the plane -----
    const int ntab = L.m1T - L.cwT;  // multiple of 4; hipMalloc'd tables and the LDS offset are 16-B aligned
    const int XS = L.xs, HS = L.hs;
    const bool vec_in = ((W & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    // compile-time shapes: every global load of the kernel (tables, plane, stripe; for SQ == 2 also the operands of the
    // output stage) is issued here, unconditionally and back to back, so the whole kernel pays ONE memory round trip; with
    // loops of predicated loads each iteration waited for its own (rocprofv3: 17.8 / 19.7 us per launch before)
    constexpr bool FAST = HS_ > 0 && (WS_ % 4) == 0;
    constexpr int cW4 = FAST ? WS_ / 4 : 1, cH = FAST ? HS_ : 1;
    constexpr int cWp = (WS_ + 3) & ~3, cHp = (HS_ + 3) & ~3;
    constexpr int cNTAB4 = FAST ? (2 * NVS_ * cWp + 2 * NUS_ * (cHp + 4) + 2 * HS_ * NUS_ + NVS_) / 4 : 1;
    constexpr int PER_T = (cNTAB4 + kBlock - 1) / kBlock, PER_P = (cH * cW4 + kBlock - 1) / kBlock;
    const bool fast = FAST && vec_in && (SQ == 0 || ((reinterpret_cast<uintptr_t>(stripe_row) & 15u) == 0)) &&
                      (SQ != 2 || ((reinterpret_cast<uintptr_t>(sq_x) & 15u) == 0));
    float4 o_x[SQ == 2 ? PER_P : 1], o_s[SQ == 2 ? PER_P : 1];  // SQ == 2: x and stripe at this lane's output positions
    if (fast) {
        float4 tv[PER_T], pv[PER_P], sv[SQ == 1 ? PER_P : 1];
#pragma unroll
        for (int q = 0; q < PER_T; ++q) {
            const int i0 = threadIdx.x + q * kBlock;
            tv[q] = reinterpret_cast<const float4 *>(tables)[i0 < cNTAB4 ? i0 : cNTAB4 - 1];
        }
#pragma unroll
        for (int q = 0; q < PER_P; ++q) {
            const int i0 = threadIdx.x + q * kBlock;
            const int i = i0 < cH * cW4 ? i0 : cH * cW4 - 1;
            const int h = i / cW4, w0 = 4 * (i - h * cW4);
            pv[q] = *reinterpret_cast<const float4 *>(src + h * W + w0);
            if (SQ == 1) sv[q] = *reinterpret_cast<const float4 *>(stripe_row + w0);
            if (SQ == 2) {
                o_x[q] = *reinterpret_cast<const float4 *>(sq_x + static_cast<size_t>(plane) * H * W + h * W + w0);
                o_s[q] = *reinterpret_cast<const float4 *>(stripe_row + w0);
            }
        }
#pragma unroll
        for (int q = 0; q < PER_T; ++q) {
            const int i0 = threadIdx.x + q * kBlock;
            if (i0 < cNTAB4) reinterpret_cast<float4 *>(lds + L.cwT)[i0] = tv[q];
        }
#pragma unroll
        for (int q = 0; q < PER_P; ++q) {
            const int i0 = threadIdx.x + q * kBlock;
            if (i0 < cH * cW4) {
                const int h = i0 / cW4, w0 = 4 * (i0 - h * cW4);
                float v[4] = {pv[q].x, pv[q].y, pv[q].z, pv[q].w};
                if (SQ == 1) {
                    const float st[4] = {sv[q].x, sv[q].y, sv[q].z, sv[q].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float dd;
                        v[k] = square_elem<false>(sq, pl, v[k], st[k], c, h, w0 + k, dd);
                    }
                }
                *reinterpret_cast<float4 *>(sx + h * XS + w0) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        // padding columns [W, XS) of every row read as zeros
        for (int i = threadIdx.x; i < H * ((XS - W) / 4); i += kBlock) {
            const int h = i / ((XS - W) / 4), w0 = W + 4 * (i - h * ((XS - W) / 4));
            *reinterpret_cast<float4 *>(sx + h * XS + w0) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    } else {
    for (int i = threadIdx.x; i < ntab / 4; i += kBlock)
        reinterpret_cast<float4 *>(lds + L.cwT)[i] = reinterpret_cast<const float4 *>(tables)[i];
    for (int i = threadIdx.x; i < H * (XS / 4); i += kBlock) {
        const int h = i / (XS / 4), w0 = 4 * (i - h * (XS / 4));
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (vec_in && w0 + 3 < W) {
            const float4 t = *reinterpret_cast<const float4 *>(src + h * W + w0);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (w0 + k < W) v[k] = src[h * W + w0 + k];
        }
        if (SQ == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (w0 + k < W) {
                    float dd;
                    v[k] = square_elem<false>(sq, pl, v[k], stripe_row[w0 + k], c, h, w0 + k, dd);
                }
        }
        *reinterpret_cast<float4 *>(sx + h * XS + w0) = make_float4(v[0], v[1], v[2], v[3]);
    }
    }
    __syncthreads();

    // ---- P, Q (stored transposed and pre-scaled: M1T[v][h] = kappa_v/W * sum_w x[h][w] cos, M2T likewise with sin) -----
    for (int i = threadIdx.x; i < H * NV; i += kBlock) {
        const int v = i / H, h = i - v * H;  // consecutive lanes: consecutive h (conflict-free row reads), same v (broadcast tables)
        const float4 *xr = reinterpret_cast<const float4 *>(sx + h * XS);
        const float4 *cr = reinterpret_cast<const float4 *>(lds + L.cwT + v * Wp);
        const float4 *sr = reinterpret_cast<const float4 *>(lds + L.swT + v * Wp);
        float p = 0.0f, q = 0.0f;
#pragma unroll 4
        for (int k = 0; k < Wp / 4; ++k) {
            const float4 xv = xr[k];
            p = dot4(xv, cr[k], p);
            q = dot4(xv, sr[k], q);
        }
        const float s = lds[L.dv + v];
        lds[L.m1T + v * HS + h] = p * s;
        lds[L.m2T + v * HS + h] = q * s;
    }
    // zero the padding columns h in [H, Hp) of M1T / M2T (read as float4 below)
    for (int i = threadIdx.x; i < NV * (Hp - H); i += kBlock) {
        const int v = i / (Hp - H), h = H + i % (Hp - H);
        lds[L.m1T + v * HS + h] = 0.0f;
        lds[L.m2T + v * HS + h] = 0.0f;
    }
    __syncthreads();

    // ---- a, b, c, d  ->  E1 = a - d, E2 = b + c, F1 = c + b, F2 = d - a, stored as [v][u] --------------------------------
    for (int i = threadIdx.x; i < NU * NV; i += kBlock) {
        const int v = i / NU, u = i - v * NU;
        const float4 *ch = reinterpret_cast<const float4 *>(lds + L.chT + u * HS);
        const float4 *sh = reinterpret_cast<const float4 *>(lds + L.shT + u * HS);
        const float4 *m1 = reinterpret_cast<const float4 *>(lds + L.m1T + v * HS);
        const float4 *m2 = reinterpret_cast<const float4 *>(lds + L.m2T + v * HS);
        float a = 0.0f, bb = 0.0f, cc = 0.0f, dd = 0.0f;
#pragma unroll 4
        for (int k = 0; k < Hp / 4; ++k) {
            const float4 c4 = ch[k], s4 = sh[k], p4 = m1[k], q4 = m2[k];
            a = dot4(c4, p4, a);
            bb = dot4(s4, p4, bb);
            cc = dot4(c4, q4, cc);
            dd = dot4(s4, q4, dd);
        }
        float *e = lds + L.e;
        e[0 * NV * NU + v * NU + u] = a - dd;
        e[1 * NV * NU + v * NU + u] = bb + cc;
        e[2 * NV * NU + v * NU + u] = cc + bb;
        e[3 * NV * NU + v * NU + u] = dd - a;
    }
    __syncthreads();

    // ---- U, V [H][NV] ------------------------------------------------------------------------------------------------------
    for (int i = threadIdx.x; i < H * NV; i += kBlock) {
        const int h = i / NV, v = i - h * NV;
        const float4 *ch = reinterpret_cast<const float4 *>(lds + L.chN + h * NU);
        const float4 *sh = reinterpret_cast<const float4 *>(lds + L.shN + h * NU);
        const float *e = lds + L.e;
        const float4 *e1 = reinterpret_cast<const float4 *>(e + 0 * NV * NU + v * NU);
        const float4 *e2 = reinterpret_cast<const float4 *>(e + 1 * NV * NU + v * NU);
        const float4 *f1 = reinterpret_cast<const float4 *>(e + 2 * NV * NU + v * NU);
        const float4 *f2 = reinterpret_cast<const float4 *>(e + 3 * NV * NU + v * NU);
        float uu = 0.0f, vv = 0.0f;
#pragma unroll 4
        for (int k = 0; k < NU / 4; ++k) {
            const float4 c4 = ch[k], s4 = sh[k];
            uu = dot4(c4, e1[k], uu);
            uu = dot4(s4, e2[k], uu);
            vv = dot4(c4, f1[k], vv);
            vv = dot4(s4, f2[k], vv);
        }
        lds[L.u + h * NV + v] = uu * d.inv_h;
        lds[L.v + h * NV + v] = vv * d.inv_h;
    }
    __syncthreads();

    // ---- y[h][w] = sum_v U[h][v] Cw[w][v] + V[h][v] Sw[w][v], four consecutive w per work item ---------------------------
    float *dst = out + static_cast<size_t>(plane) * H * W;
    const float *xo = (SQ == 2) ? sq_x + static_cast<size_t>(plane) * H * W : nullptr;
    const int W4 = Wp / 4;
    int oq = 0;  // fast path: W4 == W / 4, so this loop walks exactly the positions the prologue preloaded
    for (int i = threadIdx.x; i < H * W4; i += kBlock, ++oq) {
        const int h = i / W4, w0 = 4 * (i - h * W4);
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll 8
        for (int v = 0; v < NV; ++v) {
            const float uu = lds[L.u + h * NV + v], vv = lds[L.v + h * NV + v];
            const float4 c4 = *reinterpret_cast<const float4 *>(lds + L.cwT + v * Wp + w0);
            const float4 s4 = *reinterpret_cast<const float4 *>(lds + L.swT + v * Wp + w0);
            acc.x = fmaf(uu, c4.x, acc.x); acc.y = fmaf(uu, c4.y, acc.y); acc.z = fmaf(uu, c4.z, acc.z); acc.w = fmaf(uu, c4.w, acc.w);
            acc.x = fmaf(vv, s4.x, acc.x); acc.y = fmaf(vv, s4.y, acc.y); acc.z = fmaf(vv, s4.z, acc.z); acc.w = fmaf(vv, s4.w, acc.w);
        }
        float r[4] = {acc.x, acc.y, acc.z, acc.w};
        if (SQ == 2 && fast) {
            float4 ox = o_x[0], os = o_s[0];
#pragma unroll
            for (int q = 1; q < PER_P; ++q)
                if (oq == q) {
                    ox = o_x[q];
                    os = o_s[q];
                }
            const float xv[4] = {ox.x, ox.y, ox.z, ox.w}, st[4] = {os.x, os.y, os.z, os.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float dd = 0.0f;
                (void)square_elem<true>(sq, pl, xv[k], st[k], c, h, w0 + k, dd);
                r[k] = r[k] * dd;
            }
        } else if (SQ == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (w0 + k < W) {
                    float dd = 0.0f;
                    (void)square_elem<true>(sq, pl, xo[h * W + w0 + k], stripe_row[w0 + k], c, h, w0 + k, dd);
                    r[k] = r[k] * dd;
                }
        }
        if ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
            *reinterpret_cast<float4 *>(dst + h * W + w0) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
            for (int k = 0; k < 4 && w0 + k < W; ++k) dst[h * W + w0 + k] = r[k];
        }
    }
}

}  // namespace

EE_API int ee_hfs_table_floats(int H, int W, int NU, int NV) {
    if (H < 1 || W < 1 || NU < 1 || NV < 1) return EE_ERR_SHAPE;
    HfsDims d{H, W, (H + 3) & ~3, (W + 3) & ~3, (NU + 3) & ~3, (NV + 3) & ~3, 0.0f};
    const HfsLds L(d);
    return L.m1T - L.cwT;
}

EE_API int ee_hfs_f32(const float *in, float *out, int B, int C, int H, int W, const float *tables, int NU, int NV, float inv_h,
                      int sq_mode, const float *sq_x, float eps, const float *stripe, const float *sq_sign, const int64_t *sq_pos,
                      const int32_t *sq_size, int nq, void *stream) {
    if (B < 0 || C < 1 || H < 1 || W < 1 || NU < 1 || NV < 1 || sq_mode < 0 || sq_mode > 2 || nq < 0) return EE_ERR_SHAPE;
    if (H > 64 || W > 64 || NU > 16 || NV > 8) return EE_ERR_UNSUPPORTED;  // larger planes: the host layer uses the dense rocBLAS form
    if (B == 0) return EE_OK;
    if (!in || !out || !tables) return EE_ERR_NULL;
    if (sq_mode != 0 && (!stripe || (nq > 0 && (!sq_sign || !sq_pos || !sq_size)))) return EE_ERR_NULL;
    if (sq_mode == 2 && !sq_x) return EE_ERR_NULL;
    HfsDims d{H, W, (H + 3) & ~3, (W + 3) & ~3, (NU + 3) & ~3, (NV + 3) & ~3, inv_h};
    const HfsLds L(d);
    const size_t lds_bytes = sizeof(float) * static_cast<size_t>(L.total);
    SquareArgs sq{stripe, sq_sign, sq_pos, sq_size, nq, C, H, W, eps, static_cast<float>(2.0 * static_cast<double>(eps))};
    const unsigned grid = static_cast<unsigned>(static_cast<int64_t>(B) * C);
    ProfScope prof(sq_mode == 0 ? EE_K_HFS : (sq_mode == 1 ? EE_K_HFS_SQ_FWD : EE_K_HFS_SQ_BWD), as_stream(stream));
    hipStream_t st = as_stream(stream);
#define EE_HFS_LAUNCH(H_, W_, NU_, NV_)                                                                                      \
    do {                                                                                                                    \
        if (sq_mode == 0) EE_LAUNCH((hfs_kernel<0, H_, W_, NU_, NV_>), dim3(grid), dim3(kBlock), lds_bytes, st, in, out, tables, d, sq_x, sq); \
        else if (sq_mode == 1) EE_LAUNCH((hfs_kernel<1, H_, W_, NU_, NV_>), dim3(grid), dim3(kBlock), lds_bytes, st, in, out, tables, d, sq_x, sq); \
        else EE_LAUNCH((hfs_kernel<2, H_, W_, NU_, NV_>), dim3(grid), dim3(kBlock), lds_bytes, st, in, out, tables, d, sq_x, sq); \
    } while (0)
    if (H == 64 && W == 64 && d.NU == 16 && d.NV == 8)
        EE_HFS_LAUNCH(64, 64, 16, 8);  // Tiny-ImageNet, r = 8
    else if (H == 28 && W == 28 && d.NU == 8 && d.NV == 4)
        EE_HFS_LAUNCH(28, 28, 8, 4);  // MNIST, r = 4
    else
        EE_HFS_LAUNCH(0, 0, 0, 0);
#undef EE_HFS_LAUNCH
    return launch_status();
}
