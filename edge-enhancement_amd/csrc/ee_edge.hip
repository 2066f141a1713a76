// ee_edge.hip - CannyFilter_step125_1 (utils/core.py:509-585) forward / backward and the fused EE front
// end, as LDS-tiled, register-blocked stencil kernels for gfx950.
//
// Tiling.  One 256-thread workgroup (4 waves) owns a TH x TW tile of one image (16x64 when W > 32, else
// 32x32); grid = B * tilesY * tilesX, e.g. 400 workgroups for a batch of 100 64x64 images.  The only data
// staged in LDS are the clamped input frame (tile + halo, filled with 16-B global loads / 16-B LDS
// stores; replicate padding = clamped coordinates, never a padded tensor) and, in the backward, the
// per-pixel magnitude gradients.  Each lane owns 4 consecutive pixels of a row: it pulls a 5x8 window per
// channel out of LDS (ds_read_b64), keeps the 3x6 blurred neighbourhood of all channels in registers and
// runs the Sobel / magnitude / threshold chain there, so the forward needs ONE barrier and the backward
// three.  All global traffic is 16 B per lane (1 KiB per wave-instruction) and is issued before the first
// barrier, so its latency hides under the stencil arithmetic.
//
// Arithmetic is identical, operation for operation, to oracle/ee_oracle.c (fmaf chains: blur row-major over
// taps, Sobel over (kh, kw, c) with c innermost; IEEE divide / sqrt; -ffp-contract=off): edge bits, gate
// bits and gradients (NaNs included) are bit-exact against the oracle.
#include <math.h>

#include "ee_common.hpp"
#include "ee_stencil.hpp"

namespace {

using namespace ee;

// =====================================================================================================
// forward:  edge map, optionally fused with  x_in = clamp(x_hfs + w*edge, 0, 1)  and the clamp gate
// =====================================================================================================
template <int C, int TH, int TW, bool FUSED>
__global__ __launch_bounds__(kBlock) void edge_fwd_kernel(EdgeParams p, Weights wt) {
    constexpr int FH = TH + 4, FW = TW + 2 * kColHalo, LX = TW / 4;
    __shared__ __align__(16) float xs[C * FH * FW];
    const int H = p.H, W = p.W;
    int n, i0, j0;
    tile_origin(p, TH, TW, n, i0, j0);
    const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
    const int i = i0 + ly, jb = j0 + 4 * lx;
    const bool live = (i < H) && (jb < W);
    const bool vec = p.vec != 0;

    // low-pass branch values for this lane's pixels: issued before the barrier so the latency overlaps the stencil
    float4 xh[C];
    if (FUSED && vec) {  // unconditional on clamped addresses (dead lanes never store): no branch, the loads batch
#pragma unroll
        for (int c = 0; c < C; ++c)
            xh[c] = *reinterpret_cast<const float4 *>(p.x_hfs + ((static_cast<size_t>(n) * C + c) * H + (i < H ? i : H - 1)) * W + clamp_col4(jb, W));
    } else if (FUSED && live) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float *src = p.x_hfs + ((static_cast<size_t>(n) * C + c) * H + i) * W + jb;
            {
                xh[c].x = src[0];
                xh[c].y = (jb + 1 < W) ? src[1] : 0.0f;
                xh[c].z = (jb + 2 < W) ? src[2] : 0.0f;
                xh[c].w = (jb + 3 < W) ? src[3] : 0.0f;
            }
        }
    }
    load_frame<FH, FW, kColHalo, C>(xs, p.x + static_cast<size_t>(n) * C * H * W, C, H, W, i0, j0, 2, vec);
    __syncthreads();
    if (!live) return;

    float b[C][3][6];
    blur_group<C, FH, FW>(xs, wt, ly, 4 * lx + kColHalo - 2, i, jb, H, W, b);
    float e[4], m[4], gxs[4], gys[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float ax, ay, s2, mag_a;
        sobel_px<C>(b, wt, k, ax, ay);
        edge_from_sums<C>(ax, ay, p.alpha, p.high, gxs[k], gys[k], s2, m[k], mag_a, e[k]);
    }
    const size_t pix = (static_cast<size_t>(n) * H + i) * W + jb;
    if (vec) {
        if (p.edge) *reinterpret_cast<float4 *>(p.edge + pix) = make_float4(e[0], e[1], e[2], e[3]);
        if (p.mag) *reinterpret_cast<float4 *>(p.mag + pix) = make_float4(m[0], m[1], m[2], m[3]);
        if (p.gx_out) {
            *reinterpret_cast<float4 *>(p.gx_out + pix) = make_float4(gxs[0], gxs[1], gxs[2], gxs[3]);
            *reinterpret_cast<float4 *>(p.gy_out + pix) = make_float4(gys[0], gys[1], gys[2], gys[3]);
        }
    } else {
        for (int k = 0; k < 4 && jb + k < W; ++k) {
            if (p.edge) p.edge[pix + k] = e[k];
            if (p.mag) p.mag[pix + k] = m[k];
            if (p.gx_out) {
                p.gx_out[pix + k] = gxs[k];
                p.gy_out[pix + k] = gys[k];
            }
        }
    }
    if (FUSED) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const size_t o = ((static_cast<size_t>(n) * C + c) * H + i) * W + jb;
            const float s0 = xh[c].x + p.w * e[0], s1 = xh[c].y + p.w * e[1], s2_ = xh[c].z + p.w * e[2], s3 = xh[c].w + p.w * e[3];
            const float sv[4] = {s0, s1, s2_, s3};
            if (vec) {
                *reinterpret_cast<float4 *>(p.x_in + o) =
                    make_float4(tclamp(s0, 0.0f, 1.0f), tclamp(s1, 0.0f, 1.0f), tclamp(s2_, 0.0f, 1.0f), tclamp(s3, 0.0f, 1.0f));
                if (p.gate) {
                    uchar4 gt;
                    gt.x = (s0 >= 0.0f && s0 <= 1.0f);
                    gt.y = (s1 >= 0.0f && s1 <= 1.0f);
                    gt.z = (s2_ >= 0.0f && s2_ <= 1.0f);
                    gt.w = (s3 >= 0.0f && s3 <= 1.0f);
                    *reinterpret_cast<uchar4 *>(p.gate + o) = gt;
                }
            } else {
                for (int k = 0; k < 4 && jb + k < W; ++k) {
                    p.x_in[o + k] = tclamp(sv[k], 0.0f, 1.0f);
                    if (p.gate) p.gate[o + k] = (sv[k] >= 0.0f && sv[k] <= 1.0f);
                }
            }
        }
    }
}

// =====================================================================================================
// backward:  g_img = adjoint of the edge filter applied to u (NaN-faithful); the fused variant forms
//            g_hfs = g_in * gate and u = w * sum_c g_hfs_c itself.
//
// Frame = (TH+8) x (TW+8) positions with origin at image pixel (i0-4, j0-4).
//   stage 1 (global -> LDS/regs): clamped x frame; u on the frame; this lane's g_in / gate values
//   stage 2: gg = d(loss)/d(gx1, gy1) per pixel, rows/cols [i0-2, i0+TH+2) x [j0-2, j0+TW+2), 0 outside the image
//   stage 3: gb = ReplicationPad^T( Sx^T ggx + Sy^T ggy ), rows/cols [i0-1, i0+TH+1) x [j0-1, j0+TW+1)
//   stage 4: out = ReplicationPad^T( G^T gb ) on the tile
// Transposed correlations visit taps in row-major order as one fmaf chain and DO multiply zero-weight
// taps (0 * NaN = NaN, like a GEMM-based dgrad); the pad adjoint adds its cells in raster order.
// =====================================================================================================
template <int C, int TH, int TW, bool FUSED>
__global__ __launch_bounds__(kBlock) void edge_bwd_kernel(EdgeParams p, Weights wt) {
    constexpr int FH = TH + 8, FW = TW + 2 * kColHalo, PL = FH * FW, LX = TW / 4;
    constexpr int GG_GX = (TW + 4 + 3) / 4, GG_ROWS = TH + 4;  // 4-pixel groups of the gg region
    __shared__ __align__(16) float lds[(C + 4) * PL];
    float *xs = lds;           // [C] planes: clamped input frame
    float *us = lds + C * PL;  // upstream gradient u on the frame
    float *ggx = us + PL, *ggy = ggx + PL, *gb = ggy + PL;  // zero wherever nothing is written (see stage 1)
    const int H = p.H, W = p.W;
    int n, i0, j0;
    tile_origin(p, TH, TW, n, i0, j0);
    const int oi = i0 - 4, oj = j0 - kColHalo;  // image coordinates of frame (0, 0)
    const bool vec = p.vec != 0;
    const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
    const int ti = i0 + ly, tjb = j0 + 4 * lx;
    const bool live = (ly < TH) && (ti < H) && (tjb < W);

    // ---- stage 1 -------------------------------------------------------------------------------------
    float4 gin[FUSED ? C : 1];
    uchar4 gtv[FUSED ? C : 1];
    if (FUSED && vec) {  // unconditional on clamped (always valid) addresses: the loads batch with the frame's; dead lanes never store
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const size_t o = ((static_cast<size_t>(n) * C + c) * H + (ti < H ? ti : H - 1)) * W + clamp_col4(tjb, W);
            gin[c] = *reinterpret_cast<const float4 *>(p.g_in + o);
            gtv[c] = *reinterpret_cast<const uchar4 *>(p.gate_in + o);
        }
    } else if (FUSED && live) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const size_t o = ((static_cast<size_t>(n) * C + c) * H + ti) * W + tjb;
            {
                float gv[4] = {0, 0, 0, 0};
                unsigned char tv[4] = {0, 0, 0, 0};
                for (int k = 0; k < 4 && tjb + k < W; ++k) {
                    gv[k] = p.g_in[o + k];
                    tv[k] = p.gate_in[o + k];
                }
                gin[c] = make_float4(gv[0], gv[1], gv[2], gv[3]);
                gtv[c] = make_uchar4(tv[0], tv[1], tv[2], tv[3]);
            }
        }
    }
    load_frame<FH, FW, kColHalo, C>(xs, p.x + static_cast<size_t>(n) * C * H * W, C, H, W, i0, j0, 4, vec);
    if (FUSED) {
        load_u_fused<C, FH, FW, 2>(us, p.g_in, p.gate_in, n, H, W, oi, oj, p.w, vec);  // u is read on frame rows [2, FH-2) only
    } else {
        load_frame<FH, FW, kColHalo, 1>(us, p.u + static_cast<size_t>(n) * H * W, 1, H, W, i0, j0, 4, vec);
    }
    // gg / gb planes start as zeros: a cell outside the image, or outside the range a stage fills, then reads as
    // "contributes nothing" and the transposed correlations below need no per-tap bounds test
    for (int idx = threadIdx.x; idx < 3 * PL / 4; idx += kBlock) reinterpret_cast<float4 *>(ggx)[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();

    // ---- stage 2: magnitude-stage gradient on 4-pixel groups ---------------------------------------------
    for (int gidx = threadIdx.x; gidx < GG_ROWS * GG_GX; gidx += kBlock) {
        const int r = gidx / GG_GX, g = gidx - r * GG_GX;
        const int fr = r + 2, fc = 2 + 4 * g;  // frame coordinates of the group's first pixel
        const int i = oi + fr, jb = oj + fc;
        if (i >= 0 && i < H && jb < W && jb + 3 >= 0) {
            float b[C][3][6];
            blur_group<C, FH, FW>(xs, wt, fr - 2, fc - 2, i, jb, H, W, b);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = jb + k;
                if (j >= 0 && j < W && fc + k < FW) {
                    float ax, ay, gx1, gy1, s2, mag, mag_a, e;
                    sobel_px<C>(b, wt, k, ax, ay);
                    edge_from_sums<C>(ax, ay, p.alpha, p.high, gx1, gy1, s2, mag, mag_a, e);
                    float gm = us[fr * FW + fc + k];
                    if (mag_a <= p.high) gm = 0.0f;     // To_compare.backward core.py:356
                    if (mag_a > 1.001f) gm = 0.0f;      // core.py:357
                    if (mag < p.alpha) gm = 0.0f;       // where() backward core.py:575
                    const float rs = 1.0f / sqrtf(s2);  // pow(s2, -0.5): 0 -> inf
                    const float gs = gm * (0.5f * rs);  // 0*inf = NaN kept
                    ggx[fr * FW + fc + k] = (gs * (2.0f * gx1)) / static_cast<float>(C);
                    ggy[fr * FW + fc + k] = (gs * (2.0f * gy1)) / static_cast<float>(C);
                }
            }
        }
    }
    __syncthreads();

    // ---- stages 3 + 4: gb = pad^T(Sx^T ggx + Sy^T ggy), barrier, out = pad^T(G^T gb) on this lane's 4 pixels ------------
    float o4[4];
    adjoint_tail<TH, TW, FW>(ggx, ggy, gb, wt, H, W, i0, j0, oi, oj, live, ti, tjb, o4);
    if (!live) return;
    const size_t pix = (static_cast<size_t>(n) * H + ti) * W + tjb;
    if (vec) {
        *reinterpret_cast<float4 *>(p.g_img + pix) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    } else {
        for (int k = 0; k < 4 && tjb + k < W; ++k) p.g_img[pix + k] = o4[k];
    }
    if (FUSED) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const size_t o = ((static_cast<size_t>(n) * C + c) * H + ti) * W + tjb;
            const float4 g = gin[c];
            const uchar4 gt = gtv[c];
            const float r4[4] = {gt.x ? g.x : 0.0f, gt.y ? g.y : 0.0f, gt.z ? g.z : 0.0f, gt.w ? g.w : 0.0f};
            if (vec) {
                *reinterpret_cast<float4 *>(p.g_hfs + o) = make_float4(r4[0], r4[1], r4[2], r4[3]);
            } else {
                for (int k = 0; k < 4 && tjb + k < W; ++k) p.g_hfs[o + k] = r4[k];
            }
        }
    }
}

// =====================================================================================================
// backward from SAVED Sobel responses: the forward keeps gx1, gy1 (8 B per pixel), so the backward neither
// re-reads x nor recomputes blur + Sobel (~60 % of the recomputing kernel's instructions).  Same stages 2-4,
// same arithmetic on the same values: results are bit-identical to edge_bwd_kernel.
//   stage 1: u on the frame (LDS); this lane's stage-2 group of gx1 / gy1 straight into registers (issued first)
//   stage 2: gg per pixel -> LDS;   stages 3 + 4: adjoint_tail
// =====================================================================================================
template <int C, int TH, int TW>
__global__ __launch_bounds__(kBlock) void edge_bwd_saved_kernel(EdgeParams p, Weights wt) {
    constexpr int FH = TH + 8, FW = TW + 2 * kColHalo, PL = FH * FW, LX = TW / 4;
    constexpr int GG_GX = (TW + 4 + 3) / 4, GG_ROWS = TH + 4;
    static_assert(GG_ROWS * GG_GX <= kBlock, "stage 2 must be one round: its operands are preloaded per lane");
    __shared__ __align__(16) float lds[4 * PL];
    float *us = lds, *ggx = us + PL, *ggy = ggx + PL, *gb = ggy + PL;
    const int H = p.H, W = p.W;
    int n, i0, j0;
    tile_origin(p, TH, TW, n, i0, j0);
    const int oi = i0 - 4, oj = j0 - kColHalo;
    const bool vec = p.vec != 0;
    const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
    const int ti = i0 + ly, tjb = j0 + 4 * lx;
    const bool live = (ly < TH) && (ti < H) && (tjb < W);

    // ---- every global load of the kernel, issued up front on clamped addresses -----------------------------------------
    // stage-2 role of this lane: 4-pixel group (r, g) of the (TH+4) x (TW+4) region, pixels (i, jb .. jb+3)
    const int r2 = threadIdx.x / GG_GX, g2 = threadIdx.x - r2 * GG_GX;
    const int fr2 = r2 + 2, fc2 = 2 + 4 * g2;
    const int i2 = oi + fr2, jb2 = oj + fc2;
    const bool grp = r2 < GG_ROWS && i2 >= 0 && i2 < H && jb2 < W && jb2 + 3 >= 0;
    float gxv[4], gyv[4];
    {
        const int ic = i2 < 0 ? 0 : (i2 >= H ? H - 1 : i2);
        const size_t row = (static_cast<size_t>(n) * H + ic) * W;
        if (vec) {  // jb2 = j0 - 2 + 4 g2 is even: two 8-byte loads per plane, columns clamped into the row
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                int jc = jb2 + 2 * hlf;
                jc = jc < 0 ? 0 : (jc > W - 2 ? W - 2 : jc);
                const float2 a = *reinterpret_cast<const float2 *>(p.gx_in + row + jc);
                const float2 b = *reinterpret_cast<const float2 *>(p.gy_in + row + jc);
                gxv[2 * hlf] = a.x; gxv[2 * hlf + 1] = a.y;
                gyv[2 * hlf] = b.x; gyv[2 * hlf + 1] = b.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int jc = clampi(jb2 + k, 0, W - 1);
                gxv[k] = p.gx_in[row + jc];
                gyv[k] = p.gy_in[row + jc];
            }
        }
    }
    float4 gin[C];
    uchar4 gtv[C];
    if (vec) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const size_t o = ((static_cast<size_t>(n) * C + c) * H + (ti < H ? ti : H - 1)) * W + clamp_col4(tjb, W);
            gin[c] = *reinterpret_cast<const float4 *>(p.g_in + o);
            gtv[c] = *reinterpret_cast<const uchar4 *>(p.gate_in + o);
        }
    } else if (live) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const size_t o = ((static_cast<size_t>(n) * C + c) * H + ti) * W + tjb;
            float gv[4] = {0, 0, 0, 0};
            unsigned char tv[4] = {0, 0, 0, 0};
            for (int k = 0; k < 4 && tjb + k < W; ++k) {
                gv[k] = p.g_in[o + k];
                tv[k] = p.gate_in[o + k];
            }
            gin[c] = make_float4(gv[0], gv[1], gv[2], gv[3]);
            gtv[c] = make_uchar4(tv[0], tv[1], tv[2], tv[3]);
        }
    }
    load_u_fused<C, FH, FW, 2>(us, p.g_in, p.gate_in, n, H, W, oi, oj, p.w, vec);  // u is read on frame rows [2, FH-2) only
    for (int idx = threadIdx.x; idx < 3 * PL / 4; idx += kBlock) reinterpret_cast<float4 *>(ggx)[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();

    // ---- stage 2 (one round): the magnitude-stage gradient from the saved responses -------------------------------------
    if (grp) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = jb2 + k;
            if (j >= 0 && j < W && fc2 + k < FW) {
                const float gx1 = gxv[k], gy1 = gyv[k];
                const float s2 = gx1 * gx1 + gy1 * gy1;
                const float mag = sqrtf(s2);
                const float mag_a = (mag < p.alpha) ? 0.0f : mag;
                float gm = us[fr2 * FW + fc2 + k];
                if (mag_a <= p.high) gm = 0.0f;     // To_compare.backward core.py:356
                if (mag_a > 1.001f) gm = 0.0f;      // core.py:357
                if (mag < p.alpha) gm = 0.0f;       // where() backward core.py:575
                const float rs = 1.0f / sqrtf(s2);  // pow(s2, -0.5): 0 -> inf
                const float gs = gm * (0.5f * rs);  // 0*inf = NaN kept
                ggx[fr2 * FW + fc2 + k] = (gs * (2.0f * gx1)) / static_cast<float>(C);
                ggy[fr2 * FW + fc2 + k] = (gs * (2.0f * gy1)) / static_cast<float>(C);
            }
        }
    }
    __syncthreads();

    float o4[4];
    adjoint_tail<TH, TW, FW>(ggx, ggy, gb, wt, H, W, i0, j0, oi, oj, live, ti, tjb, o4);
    if (!live) return;
    const size_t pix = (static_cast<size_t>(n) * H + ti) * W + tjb;
    if (vec) {
        *reinterpret_cast<float4 *>(p.g_img + pix) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    } else {
        for (int k = 0; k < 4 && tjb + k < W; ++k) p.g_img[pix + k] = o4[k];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const size_t o = ((static_cast<size_t>(n) * C + c) * H + ti) * W + tjb;
        const float4 g = gin[c];
        const uchar4 gt = gtv[c];
        const float r4[4] = {gt.x ? g.x : 0.0f, gt.y ? g.y : 0.0f, gt.z ? g.z : 0.0f, gt.w ? g.w : 0.0f};
        if (vec) {
            *reinterpret_cast<float4 *>(p.g_hfs + o) = make_float4(r4[0], r4[1], r4[2], r4[3]);
        } else {
            for (int k = 0; k < 4 && tjb + k < W; ++k) p.g_hfs[o + k] = r4[k];
        }
    }
}

// ---- host side --------------------------------------------------------------------------------------
struct Launch {
    int th, tw, tiles_x, tiles_y;
    unsigned grid;
};

// forward: 16x64 / 32x32 tiles = exactly 256 four-pixel groups.  backward: 11x64 / 24x32, because its heaviest
// stage works on the (TH+4) x (TW+4) halo region: 15 x 17 = 255 (resp. 28 x 9 = 252) groups = one round of 256 lanes.
constexpr int kFwdTH64 = 16, kFwdTH32 = 32, kBwdTH64 = 11, kBwdTH32 = 24;

Launch plan(int B, int H, int W, bool bwd) {
    Launch l;
    int th = (W > 32) ? (bwd ? kBwdTH64 : kFwdTH64) : (bwd ? kBwdTH32 : kFwdTH32);
    l.tw = (W > 32) ? 64 : 32;
    // plenty of workgroups (> 4 per CU): total work, not the critical path of one workgroup, sets the time, and the
    // 16-row tile recomputes less halo (two rounds of stage 2, but 20/16 instead of 15/11 rows per output row)
    if (bwd && W > 32 && static_cast<int64_t>(B) * ((H + kFwdTH64 - 1) / kFwdTH64) * ((W + 63) / 64) >= 1024) th = kFwdTH64;
    l.th = th;
    l.tiles_x = (W + l.tw - 1) / l.tw;
    l.tiles_y = (H + th - 1) / th;
    l.grid = static_cast<unsigned>(static_cast<int64_t>(B) * l.tiles_x * l.tiles_y);
    return l;
}

int check_dims(int B, int C, int H, int W) {
    if (B < 0 || C < 1 || H < 1 || W < 1) return EE_ERR_SHAPE;
    if (C > 4) return EE_ERR_UNSUPPORTED;
    if (static_cast<int64_t>(B) * ((H + 10) / 11) * ((W + 31) / 32) > 0x7fffffffLL) return EE_ERR_SHAPE;
    return EE_OK;
}

Weights load_weights(const float *w27) {
    Weights wt;
    for (int k = 0; k < 9; ++k) {
        wt.g[k] = w27[k];
        wt.sx[k] = w27[9 + k];
        wt.sy[k] = w27[18 + k];
    }
    return wt;
}

template <int C, bool FUSED>
void launch_fwd_c(const EdgeParams &p, const Weights &wt, const Launch &l, hipStream_t s) {
    if (l.tw == 64)
        EE_LAUNCH((edge_fwd_kernel<C, kFwdTH64, 64, FUSED>), dim3(l.grid), dim3(kBlock), 0, s, p, wt);
    else
        EE_LAUNCH((edge_fwd_kernel<C, kFwdTH32, 32, FUSED>), dim3(l.grid), dim3(kBlock), 0, s, p, wt);
}

template <int C, bool FUSED>
void launch_bwd_c(const EdgeParams &p, const Weights &wt, const Launch &l, hipStream_t s) {
    if (l.tw == 64 && l.th == kBwdTH64)
        EE_LAUNCH((edge_bwd_kernel<C, kBwdTH64, 64, FUSED>), dim3(l.grid), dim3(kBlock), 0, s, p, wt);
    else if (l.tw == 64)
        EE_LAUNCH((edge_bwd_kernel<C, kFwdTH64, 64, FUSED>), dim3(l.grid), dim3(kBlock), 0, s, p, wt);
    else
        EE_LAUNCH((edge_bwd_kernel<C, kBwdTH32, 32, FUSED>), dim3(l.grid), dim3(kBlock), 0, s, p, wt);
}

template <bool FUSED, bool BWD>
int launch(const EdgeParams &p0, const Weights &wt, int B, int C, hipStream_t s) {
    EdgeParams p = p0;
    const Launch l = plan(B, p.H, p.W, BWD);
    p.tiles_x = l.tiles_x;
    p.tiles_y = l.tiles_y;
    if (l.grid == 0) return EE_OK;
    switch (C) {
        case 1: BWD ? launch_bwd_c<1, FUSED>(p, wt, l, s) : launch_fwd_c<1, FUSED>(p, wt, l, s); break;
        case 2: BWD ? launch_bwd_c<2, FUSED>(p, wt, l, s) : launch_fwd_c<2, FUSED>(p, wt, l, s); break;
        case 3: BWD ? launch_bwd_c<3, FUSED>(p, wt, l, s) : launch_fwd_c<3, FUSED>(p, wt, l, s); break;
        default: BWD ? launch_bwd_c<4, FUSED>(p, wt, l, s) : launch_fwd_c<4, FUSED>(p, wt, l, s); break;
    }
    return launch_status();
}

inline bool al16(const void *q) { return !q || aligned16(q); }
inline bool al4(const void *q) { return !q || (reinterpret_cast<uintptr_t>(q) & 3u) == 0; }

}  // namespace

EE_API int ee_edge125_fwd_f32(const float *x, int B, int C, int H, int W, const float *weights27, float alpha, float high,
                              float *edge, float *mag, void *stream) {
    if (int rc = check_dims(B, C, H, W)) return rc;
    if (B == 0) return EE_OK;
    if (!x || !weights27 || !edge) return EE_ERR_NULL;
    EdgeParams p{};
    p.x = x;
    p.edge = edge;
    p.mag = mag;
    p.H = H; p.W = W;
    p.alpha = alpha; p.high = high; p.w = 0.0f;
    p.vec = (W % 4 == 0) && al16(x) && al16(edge) && al16(mag);
    ProfScope prof(EE_K_EDGE_FWD, as_stream(stream));
    return launch<false, false>(p, load_weights(weights27), B, C, as_stream(stream));
}

EE_API int ee_edge125_bwd_f32(const float *x, const float *u, int B, int C, int H, int W, const float *weights27, float alpha,
                              float high, float *g_img, void *stream) {
    if (int rc = check_dims(B, C, H, W)) return rc;
    if (B == 0) return EE_OK;
    if (!x || !u || !weights27 || !g_img) return EE_ERR_NULL;
    EdgeParams p{};
    p.x = x;
    p.u = u;
    p.g_img = g_img;
    p.H = H; p.W = W;
    p.alpha = alpha; p.high = high; p.w = 0.0f;
    p.vec = (W % 4 == 0) && al16(x) && al16(u) && al16(g_img);
    ProfScope prof(EE_K_EDGE_BWD, as_stream(stream));
    return launch<false, true>(p, load_weights(weights27), B, C, as_stream(stream));
}

EE_API int ee_frontend_fwd_f32(const float *x, const float *x_hfs, int B, int C, int H, int W, const float *weights27,
                               float alpha, float high, float w, float *x_in, uint8_t *gate, float *edge, void *stream) {
    if (int rc = check_dims(B, C, H, W)) return rc;
    if (B == 0) return EE_OK;
    if (!x || !x_hfs || !weights27 || !x_in) return EE_ERR_NULL;
    EdgeParams p{};
    p.x = x;
    p.x_hfs = x_hfs;
    p.x_in = x_in;
    p.gate = gate;
    p.edge = edge;
    p.H = H; p.W = W;
    p.alpha = alpha; p.high = high; p.w = w;
    p.vec = (W % 4 == 0) && al16(x) && al16(x_hfs) && al16(x_in) && al16(edge) && al4(gate);
    ProfScope prof(EE_K_FRONTEND_FWD, as_stream(stream));
    return launch<true, false>(p, load_weights(weights27), B, C, as_stream(stream));
}

EE_API int ee_frontend_bwd_f32(const float *g_in, const uint8_t *gate, const float *x, int B, int C, int H, int W,
                               const float *weights27, float alpha, float high, float w, float *g_hfs, float *g_edge,
                               void *stream) {
    if (int rc = check_dims(B, C, H, W)) return rc;
    if (B == 0) return EE_OK;
    if (!g_in || !gate || !x || !weights27 || !g_hfs || !g_edge) return EE_ERR_NULL;
    EdgeParams p{};
    p.x = x;
    p.g_in = g_in;
    p.gate_in = gate;
    p.g_hfs = g_hfs;
    p.g_img = g_edge;
    p.H = H; p.W = W;
    p.alpha = alpha; p.high = high; p.w = w;
    p.vec = (W % 4 == 0) && al16(x) && al16(g_in) && al16(g_hfs) && al16(g_edge) && al4(gate);
    ProfScope prof(EE_K_FRONTEND_BWD, as_stream(stream));
    return launch<true, true>(p, load_weights(weights27), B, C, as_stream(stream));
}

// ---- the pair that keeps the Sobel responses between forward and backward --------------------------------------------------
namespace {

template <int C>
void launch_bwd_saved_c(const EdgeParams &p, const Weights &wt, int B, hipStream_t st) {
    EdgeParams q = p;
    const bool wide = p.W > 32;
    const int th = wide ? kBwdTH64 : kBwdTH32, tw = wide ? 64 : 32;
    q.tiles_x = (p.W + tw - 1) / tw;
    q.tiles_y = (p.H + th - 1) / th;
    const unsigned grid = static_cast<unsigned>(static_cast<int64_t>(B) * q.tiles_x * q.tiles_y);
    if (wide)
        EE_LAUNCH((edge_bwd_saved_kernel<C, kBwdTH64, 64>), dim3(grid), dim3(kBlock), 0, st, q, wt);
    else
        EE_LAUNCH((edge_bwd_saved_kernel<C, kBwdTH32, 32>), dim3(grid), dim3(kBlock), 0, st, q, wt);
}

}  // namespace

EE_API int ee_frontend_fwd_save_f32(const float *x, const float *x_hfs, int B, int C, int H, int W, const float *weights27, float alpha,
                                    float high, float w, float *x_in, uint8_t *gate, float *edge, float *gx, float *gy, void *stream) {
    if (int rc = check_dims(B, C, H, W)) return rc;
    if (B == 0) return EE_OK;
    if (!x || !x_hfs || !weights27 || !x_in || !gx || !gy) return EE_ERR_NULL;
    EdgeParams p{};
    p.x = x;
    p.x_hfs = x_hfs;
    p.x_in = x_in;
    p.gate = gate;
    p.edge = edge;
    p.gx_out = gx;
    p.gy_out = gy;
    p.H = H; p.W = W;
    p.alpha = alpha; p.high = high; p.w = w;
    p.vec = (W % 4 == 0) && al16(x) && al16(x_hfs) && al16(x_in) && al16(edge) && al4(gate) && al16(gx) && al16(gy);
    ProfScope prof(EE_K_FRONTEND_FWD, as_stream(stream));
    return launch<true, false>(p, load_weights(weights27), B, C, as_stream(stream));
}

EE_API int ee_frontend_bwd_saved_f32(const float *g_in, const uint8_t *gate, const float *gx, const float *gy, int B, int C, int H, int W,
                                     const float *weights27, float alpha, float high, float w, float *g_hfs, float *g_edge, void *stream) {
    if (int rc = check_dims(B, C, H, W)) return rc;
    if (B == 0) return EE_OK;
    if (!g_in || !gate || !gx || !gy || !weights27 || !g_hfs || !g_edge) return EE_ERR_NULL;
    EdgeParams p{};
    p.g_in = g_in;
    p.gate_in = gate;
    p.gx_in = gx;
    p.gy_in = gy;
    p.g_hfs = g_hfs;
    p.g_img = g_edge;
    p.H = H; p.W = W;
    p.alpha = alpha; p.high = high; p.w = w;
    p.vec = (W % 4 == 0) && al16(g_in) && al16(g_hfs) && al16(g_edge) && al4(gate) && al16(gx) && al16(gy);
    const Weights wt = load_weights(weights27);
    hipStream_t st = as_stream(stream);
    ProfScope prof(EE_K_FRONTEND_BWD, st);
    switch (C) {
        case 1: launch_bwd_saved_c<1>(p, wt, B, st); break;
        case 2: launch_bwd_saved_c<2>(p, wt, B, st); break;
        case 3: launch_bwd_saved_c<3>(p, wt, B, st); break;
        default: launch_bwd_saved_c<4>(p, wt, B, st); break;
    }
    return launch_status();
}
