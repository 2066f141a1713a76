// ee_canny.hip - the full CannyFilter (utils/core.py:148-326): Gaussian -> Sobel -> magnitude -> alpha mask ->
// orientation-quantised non-maximum suppression -> straight-through double threshold -> hysteresis, forward and backward,
// optionally fused with the EE front-end combine  x_in = clamp(x_hfs + w*edge, 0, 1)  (Net2_EE.py:36-49, resnet_EE.py:176-191).
//
// Same machinery as ee_edge.hip (one 256-lane workgroup per tile, clamped input frame in LDS, 4-pixel register blocking,
// bit-for-bit the operation order of oracle/ee_oracle.c).  Extra LDS stages carry the masked magnitude, the quantised
// direction index and the {0, 0.5, 1} threshold map so that the 3x3 neighbourhood logic (NMS, hysteresis) never leaves
// the CU.  Only the path every model takes is implemented: low_threshold and high_threshold given, hysteresis=True.
//
// Backward: only `high` carries gradient (low / hysteresis enter through comparisons):  g_mag = (u/2) * 1[|t - high| <= 1.001]
// * 1[not suppressed] * 1[mag >= alpha], followed by exactly the adjoint chain of the step125 filter (0 * inf = NaN kept).
//
// PARITY UNPINNED: the 8 directional kernels of the reference are built with cv2 (core.py:87-112), which is not available;
// the direction table is passed in by the host (derived table, eeadv/ops.py) and both oracle and kernel use it.
#include "ee_common.hpp"
#include "ee_stencil.hpp"

namespace {

using namespace ee;

struct CannyDirs {
    int dr[8], dc[8];
};

struct CannyParams {
    EdgeParams e;  // pointers / sizes exactly as in ee_edge.hip
    float low;
    // CannyFilter_BPDA (core.py:386-505): the forward also stores the thinned magnitude and the {0, .5, 1} threshold map;
    // the backward then receives d loss / d thin (bpda_threshold_bwd_kernel) instead of d loss / d edge
    float *thin_out, *t2_out;
    int thin_grad;
};

// quantised orientation index of core.py:258-260, 270 (fp32 throughout): NaN (gx = gy = 0) -> -1
__device__ __forceinline__ float dir_index(float gx1, float gy1) {
    const float ori = atanf(gy1 / gx1) * static_cast<float>(360.0 / 3.141592653589793) + 180.0f;
    const float ori2 = rintf(ori / 45.0f) * 45.0f;
    const float q = ori2 / 45.0f;
    const float pidx = q - 8.0f * floorf(q / 8.0f);
    return (pidx == pidx) ? pidx : -1.0f;
}

// suppressed-by-NMS test for the pixel at frame position (fr, fc) (core.py:268-290); magA is 0 outside the image (zero padding)
template <int FW>
__device__ __forceinline__ bool nms_removed(const float *magA, const float *kdir, const int *dirs, int fr, int fc) {
    const float kf = kdir[fr * FW + fc];
    if (kf < 0.0f) return false;
    const int kk = static_cast<int>(kf) & 3;
    const float m = magA[fr * FW + fc];
    const float d1 = m - magA[(fr + dirs[kk]) * FW + fc + dirs[8 + kk]];
    const float d2 = m - magA[(fr + dirs[kk + 4]) * FW + fc + dirs[8 + kk + 4]];
    const float mn = (d1 != d1 || d2 != d2) ? (d1 + d2) : (d1 < d2 ? d1 : d2);
    return !(mn > 0.0f);
}

// magnitude stage for the 4-pixel group at frame (fr, fc): writes gx1 / gy1 (optional), the alpha-masked magnitude and the
// direction index of every in-image pixel of the group; everything else keeps its zero initialisation.
template <int C, int FH, int FW>
__device__ __forceinline__ void magnitude_group(const float *xs, const Weights &wt, int fr, int fc, int i, int jb, int H, int W, float alpha,
                                                float *gx1s, float *gy1s, float *magAs, float *kdirs) {
    float b[C][3][6];
    blur_group<C, FH, FW>(xs, wt, fr - 2, fc - 2, i, jb, H, W, b);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = jb + k;
        if (j >= 0 && j < W && fc + k < FW) {
            float ax, ay, gx1, gy1, s2, mag, mag_a, e;
            sobel_px<C>(b, wt, k, ax, ay);
            edge_from_sums<C>(ax, ay, alpha, 0.0f, gx1, gy1, s2, mag, mag_a, e);
            const int o = fr * FW + fc + k;
            if (gx1s) {
                gx1s[o] = gx1;
                gy1s[o] = gy1;
            }
            magAs[o] = mag_a;
            kdirs[o] = dir_index(gx1, gy1);
        }
    }
}

// =====================================================================================================
// forward
// =====================================================================================================
template <int C, int TH, int TW, bool FUSED>
__global__ __launch_bounds__(kBlock) void canny_fwd_kernel(CannyParams cp, Weights wt, CannyDirs cd) {
    constexpr int FH = TH + 8, FW = TW + 8, PL = FH * FW, LX = TW / 4;
    constexpr int MG_GX = (TW + 4 + 3) / 4, MG_ROWS = TH + 4;
    __shared__ __align__(16) float lds[(C + 3) * PL];
    __shared__ int dirs[16];
    const EdgeParams &p = cp.e;
    float *xs = lds, *magAs = lds + C * PL, *kdirs = magAs + PL, *t2s = kdirs + PL;
    const int H = p.H, W = p.W;
    int n, i0, j0;
    tile_origin(p, TH, TW, n, i0, j0);
    const int oi = i0 - 4, oj = j0 - 4;
    const bool vec = p.vec != 0;
    const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
    const int ti = i0 + ly, tjb = j0 + 4 * lx;
    const bool live = (ly < TH) && (ti < H) && (tjb < W);

    float4 xh[C];
    if (FUSED && live) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float *src = p.x_hfs + ((static_cast<size_t>(n) * C + c) * H + ti) * W + tjb;
            if (vec) {
                xh[c] = *reinterpret_cast<const float4 *>(src);
            } else {
                xh[c].x = src[0];
                xh[c].y = (tjb + 1 < W) ? src[1] : 0.0f;
                xh[c].z = (tjb + 2 < W) ? src[2] : 0.0f;
                xh[c].w = (tjb + 3 < W) ? src[3] : 0.0f;
            }
        }
    }
    if (threadIdx.x < 8) {
        dirs[threadIdx.x] = cd.dr[threadIdx.x];
        dirs[8 + threadIdx.x] = cd.dc[threadIdx.x];
    }
    load_frame<FH, FW, 4, C>(xs, p.x + static_cast<size_t>(n) * C * H * W, C, H, W, i0, j0, 4, vec);
    for (int idx = threadIdx.x; idx < 3 * PL / 4; idx += kBlock) reinterpret_cast<float4 *>(magAs)[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();

    // ---- magnitude / direction on rows [i0-2, i0+TH+2) x cols [j0-2, j0+TW+2) ------------------------------------------
    for (int gidx = threadIdx.x; gidx < MG_ROWS * MG_GX; gidx += kBlock) {
        const int r = gidx / MG_GX, g = gidx - r * MG_GX;
        const int fr = r + 2, fc = 2 + 4 * g;
        const int i = oi + fr, jb = oj + fc;
        if (i >= 0 && i < H && jb < W && jb + 3 >= 0)
            magnitude_group<C, FH, FW>(xs, wt, fr, fc, i, jb, H, W, p.alpha, nullptr, nullptr, magAs, kdirs);
    }
    __syncthreads();

    // ---- NMS + double threshold -> t2 in {0, .5, 1} on rows [i0-1, i0+TH+1) x cols [j0-1, j0+TW+1) ---------------------
    {
        constexpr int RW = TW + 2, RH = TH + 2;
        for (int idx = threadIdx.x; idx < RH * RW; idx += kBlock) {
            const int r = idx / RW, s = idx - r * RW;
            const int i = i0 - 1 + r, j = j0 - 1 + s;
            if (i >= 0 && i < H && j >= 0 && j < W) {
                const int fr = i - oi, fc = j - oj;
                const float t = nms_removed<FW>(magAs, kdirs, dirs, fr, fc) ? 0.0f : magAs[fr * FW + fc];
                const float lowb = ((t - cp.low) > 0.0f) ? 1.0f : 0.0f, highb = ((t - p.high) > 0.0f) ? 1.0f : 0.0f;
                const float t2 = lowb * 0.5f + highb * 0.5f;
                t2s[fr * FW + fc] = t2;
                if (cp.thin_out && r >= 1 && r <= TH && s >= 1 && s <= TW) {  // this tile's own pixels
                    const size_t o = (static_cast<size_t>(n) * H + i) * W + j;
                    cp.thin_out[o] = t;
                    cp.t2_out[o] = t2;
                }
            }
        }
    }
    __syncthreads();
    if (!live) return;

    // ---- hysteresis (core.py:317-321) on this lane's 4 pixels -----------------------------------------------------------
    float e[4];
    {
        float win[3][8];  // t2 rows ti-1..ti+1, cols tjb-2..tjb+5
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float *q = t2s + (ti - 1 + a - oi) * FW + (tjb - 2 - oj);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float2 v = *reinterpret_cast<const float2 *>(q + 2 * t);
                win[a][2 * t] = v.x;
                win[a][2 * t + 1] = v.y;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float acc = 0.0f;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) acc = fmaf(1.25f, win[a][k + 1 + dj], acc);
            const float t2 = win[1][k + 2];
            const float highb = (t2 == 1.0f) ? 1.0f : 0.0f;
            e[k] = highb + ((acc > 1.0f && t2 == 0.5f) ? 1.0f : 0.0f);
        }
    }
    const size_t pix = (static_cast<size_t>(n) * H + ti) * W + tjb;
    if (p.edge) {
        if (vec) {
            *reinterpret_cast<float4 *>(p.edge + pix) = make_float4(e[0], e[1], e[2], e[3]);
        } else {
            for (int k = 0; k < 4 && tjb + k < W; ++k) p.edge[pix + k] = e[k];
        }
    }
    if (FUSED) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const size_t o = ((static_cast<size_t>(n) * C + c) * H + ti) * W + tjb;
            const float sv[4] = {xh[c].x + p.w * e[0], xh[c].y + p.w * e[1], xh[c].z + p.w * e[2], xh[c].w + p.w * e[3]};
            if (vec) {
                *reinterpret_cast<float4 *>(p.x_in + o) =
                    make_float4(tclamp(sv[0], 0.0f, 1.0f), tclamp(sv[1], 0.0f, 1.0f), tclamp(sv[2], 0.0f, 1.0f), tclamp(sv[3], 0.0f, 1.0f));
                if (p.gate) {
                    uchar4 gt;
                    gt.x = (sv[0] >= 0.0f && sv[0] <= 1.0f);
                    gt.y = (sv[1] >= 0.0f && sv[1] <= 1.0f);
                    gt.z = (sv[2] >= 0.0f && sv[2] <= 1.0f);
                    gt.w = (sv[3] >= 0.0f && sv[3] <= 1.0f);
                    *reinterpret_cast<uchar4 *>(p.gate + o) = gt;
                }
            } else {
                for (int k = 0; k < 4 && tjb + k < W; ++k) {
                    p.x_in[o + k] = tclamp(sv[k], 0.0f, 1.0f);
                    if (p.gate) p.gate[o + k] = (sv[k] >= 0.0f && sv[k] <= 1.0f);
                }
            }
        }
    }
}

// =====================================================================================================
// backward
// =====================================================================================================
template <int C, int TH, int TW, bool FUSED>
__global__ __launch_bounds__(kBlock) void canny_bwd_kernel(CannyParams cp, Weights wt, CannyDirs cd) {
    constexpr int RH_ = 5, CH_ = 8;  // row halo 5, column halo 8 (keeps every frame row 16-B aligned)
    constexpr int FH = TH + 2 * RH_, FW = TW + 2 * CH_, PL = FH * FW, LX = TW / 4;
    constexpr int MG_GX = (TW + 8 + 3) / 4, MG_ROWS = TH + 6;  // magnitude region: rows [i0-3, i0+TH+3), cols [j0-4, j0+TW+4)
    constexpr int NA = C > 3 ? C : 3;
    __shared__ __align__(16) float lds[(NA + 5) * PL];
    __shared__ int dirs[16];
    const EdgeParams &p = cp.e;
    float *xs = lds;  // [NA] planes: input frame, later ggx (0), ggy (1), gb (2)
    float *us = lds + NA * PL, *gx1s = us + PL, *gy1s = gx1s + PL, *magAs = gy1s + PL, *kdirs = magAs + PL;
    float *ggx = xs, *ggy = xs + PL, *gb = xs + 2 * PL;
    const int H = p.H, W = p.W;
    int n, i0, j0;
    tile_origin(p, TH, TW, n, i0, j0);
    const int oi = i0 - RH_, oj = j0 - CH_;
    const bool vec = p.vec != 0;
    const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
    const int ti = i0 + ly, tjb = j0 + 4 * lx;
    const bool live = (ly < TH) && (ti < H) && (tjb < W);

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
    if (threadIdx.x < 8) {
        dirs[threadIdx.x] = cd.dr[threadIdx.x];
        dirs[8 + threadIdx.x] = cd.dc[threadIdx.x];
    }
    load_frame<FH, FW, CH_, C>(xs, p.x + static_cast<size_t>(n) * C * H * W, C, H, W, i0, j0, RH_, vec);
    if (FUSED)
        load_u_fused<C, FH, FW>(us, p.g_in, p.gate_in, n, H, W, oi, oj, p.w, vec);
    else
        load_frame<FH, FW, CH_, 1>(us, p.u + static_cast<size_t>(n) * H * W, 1, H, W, i0, j0, RH_, vec);
    for (int idx = threadIdx.x; idx < 4 * PL / 4; idx += kBlock) reinterpret_cast<float4 *>(gx1s)[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();

    // ---- magnitude maps on rows [i0-3, i0+TH+3) x cols [j0-4, j0+TW+4) ---------------------------------------------------
    for (int gidx = threadIdx.x; gidx < MG_ROWS * MG_GX; gidx += kBlock) {
        const int r = gidx / MG_GX, g = gidx - r * MG_GX;
        const int fr = r + 2, fc = 4 + 4 * g;
        const int i = oi + fr, jb = oj + fc;
        if (i >= 0 && i < H && jb < W && jb + 3 >= 0)
            magnitude_group<C, FH, FW>(xs, wt, fr, fc, i, jb, H, W, p.alpha, gx1s, gy1s, magAs, kdirs);
    }
    __syncthreads();  // the input frame is dead from here on: its planes become ggx / ggy / gb
    for (int idx = threadIdx.x; idx < 3 * PL / 4; idx += kBlock) reinterpret_cast<float4 *>(xs)[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();

    // ---- gg = d loss / d (gx1, gy1) on rows [i0-2, i0+TH+2) x cols [j0-2, j0+TW+2) -----------------------------------------
    {
        constexpr int RW = TW + 4, RHh = TH + 4;
        for (int idx = threadIdx.x; idx < RHh * RW; idx += kBlock) {
            const int r = idx / RW, s = idx - r * RW;
            const int i = i0 - 2 + r, j = j0 - 2 + s;
            if (i >= 0 && i < H && j >= 0 && j < W) {
                const int fr = i - oi, fc = j - oj, o = fr * FW + fc;
                const bool removed = nms_removed<FW>(magAs, kdirs, dirs, fr, fc);
                const float gx1 = gx1s[o], gy1 = gy1s[o];
                const float s2 = gx1 * gx1 + gy1 * gy1;
                const float mag = sqrtf(s2);
                const float t = removed ? 0.0f : magAs[o];
                float gm;
                if (cp.thin_grad) {                                   // CannyFilter_BPDA: us already holds d loss / d thin
                    gm = us[o];
                    if (removed) gm = gm * 0.0f;                      // thin * ~to_remove, core.py:480 (keeps the sign of zero)
                } else {
                    gm = us[o] / 2.0f;                                // (sign + 1) / 2
                    if (fabsf(t - p.high) > 1.001f) gm = 0.0f;        // BinaryConnectDeterministic.backward core.py:138-145
                    if (removed) gm = 0.0f;                           // thin_edges[to_remove] = 0.0, core.py:290
                    if (mag < p.alpha) gm = 0.0f;                     // where() backward core.py:264
                }
                const float rs = 1.0f / sqrtf(s2);
                const float gs = gm * (0.5f * rs);
                ggx[o] = (gs * (2.0f * gx1)) / static_cast<float>(C);
                ggy[o] = (gs * (2.0f * gy1)) / static_cast<float>(C);
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

// ---- host side --------------------------------------------------------------------------------------
constexpr int kFwdTH64 = 11, kFwdTH32 = 24, kBwdTH64 = 8, kBwdTH32 = 16;

int check_dims(int B, int C, int H, int W) {
    if (B < 0 || C < 1 || H < 1 || W < 1) return EE_ERR_SHAPE;
    if (C > 4) return EE_ERR_UNSUPPORTED;
    if (static_cast<int64_t>(B) * ((H + 7) / 8) * ((W + 31) / 32) > 0x7fffffffLL) return EE_ERR_SHAPE;
    return EE_OK;
}

template <int C, bool FUSED, bool BWD>
void launch_c(CannyParams &cp, const Weights &wt, const CannyDirs &cd, int B, hipStream_t s) {
    EdgeParams &p = cp.e;
    const bool wide = p.W > 32;
    const int tw = wide ? 64 : 32;
    const int th = wide ? (BWD ? kBwdTH64 : kFwdTH64) : (BWD ? kBwdTH32 : kFwdTH32);
    p.tiles_x = (p.W + tw - 1) / tw;
    p.tiles_y = (p.H + th - 1) / th;
    const unsigned grid = static_cast<unsigned>(static_cast<int64_t>(B) * p.tiles_x * p.tiles_y);
    if (BWD) {
        if (wide)
            EE_LAUNCH((canny_bwd_kernel<C, kBwdTH64, 64, FUSED>), dim3(grid), dim3(kBlock), 0, s, cp, wt, cd);
        else
            EE_LAUNCH((canny_bwd_kernel<C, kBwdTH32, 32, FUSED>), dim3(grid), dim3(kBlock), 0, s, cp, wt, cd);
    } else {
        if (wide)
            EE_LAUNCH((canny_fwd_kernel<C, kFwdTH64, 64, FUSED>), dim3(grid), dim3(kBlock), 0, s, cp, wt, cd);
        else
            EE_LAUNCH((canny_fwd_kernel<C, kFwdTH32, 32, FUSED>), dim3(grid), dim3(kBlock), 0, s, cp, wt, cd);
    }
}

template <bool FUSED, bool BWD>
int launch(CannyParams cp, const float *w27, const int *dirs16, int B, int C, hipStream_t s) {
    Weights wt;
    for (int k = 0; k < 9; ++k) {
        wt.g[k] = w27[k];
        wt.sx[k] = w27[9 + k];
        wt.sy[k] = w27[18 + k];
    }
    CannyDirs cd;
    for (int k = 0; k < 8; ++k) {
        cd.dr[k] = dirs16[2 * k];
        cd.dc[k] = dirs16[2 * k + 1];
        if (cd.dr[k] < -1 || cd.dr[k] > 1 || cd.dc[k] < -1 || cd.dc[k] > 1) return EE_ERR_SHAPE;  // 3x3 neighbourhood only
    }
    switch (C) {
        case 1: launch_c<1, FUSED, BWD>(cp, wt, cd, B, s); break;
        case 2: launch_c<2, FUSED, BWD>(cp, wt, cd, B, s); break;
        case 3: launch_c<3, FUSED, BWD>(cp, wt, cd, B, s); break;
        default: launch_c<4, FUSED, BWD>(cp, wt, cd, B, s); break;
    }
    return launch_status();
}

// CannyFilter_BPDA backward, threshold / hysteresis part (core.py:482-504): d loss / d thin from u = d loss / d edge,
// the stored thin and t2 planes.  out = high + To_compare(conv(t2), 1) * To_eq(t2);  t2 = .5 low + .5 high;
// low / high = To_compare(thin, thr).  To_compare passes the gradient where thr < x <= 1.001, To_eq where x == 0.5.
__global__ __launch_bounds__(256) void bpda_threshold_bwd_kernel(const float *__restrict__ u, const float *__restrict__ thin,
                                                                 const float *__restrict__ t2, float *__restrict__ g_thin, int H, int W,
                                                                 float low, float high, int64_t total) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = static_cast<int>(idx % W);
    const int64_t r = idx / W;
    const int i = static_cast<int>(r % H);
    const float *t2p = t2 + (r / H) * H * W, *up = u + (r / H) * H * W;
    // weak_0 and its To_compare(., 1) gradient mask at (a, b); the hysteresis weights are all 1.25 (core.py:422)
    auto weak0 = [&](int a, int b) {
        float acc = 0.0f;
        for (int di = -1; di <= 1; ++di)
            for (int dj = -1; dj <= 1; ++dj) {
                const int rr = a + di, ss = b + dj;
                acc = fmaf(1.25f, (rr < 0 || rr >= H || ss < 0 || ss >= W) ? 0.0f : t2p[rr * W + ss], acc);
            }
        return acc;
    };
    const float w0 = weak0(i, j);
    const float t2c = t2p[i * W + j], uc = up[i * W + j];
    const float weak1 = (w0 <= 1.0f) ? 0.0f : ((w0 > 1.0f) ? 1.0f : w0);
    float g_eq = uc * weak1;            // To_eq.backward
    if (t2c != 0.5f) g_eq = 0.0f;
    float g_cv = 0.0f;                  // transposed hysteresis convolution of the To_compare(., 1) gradient
    for (int di = -1; di <= 1; ++di)
        for (int dj = -1; dj <= 1; ++dj) {
            const int rr = i - di, ss = j - dj;
            float v = 0.0f;
            if (rr >= 0 && rr < H && ss >= 0 && ss < W) {
                const float wq = weak0(rr, ss);
                v = up[rr * W + ss] * ((t2p[rr * W + ss] == 0.5f) ? 1.0f : 0.0f);
                if (wq <= 1.0f) v = 0.0f;
                if (wq > 1.001f) v = 0.0f;
            }
            g_cv = fmaf(1.25f, v, g_cv);
        }
    const float g_t2 = g_eq + g_cv;
    const float th = thin[idx];
    float g_low = g_t2 * 0.5f, g_high = uc * 1.0f + g_t2 * 0.5f;
    if (th <= low) g_low = 0.0f;
    if (th > 1.001f) g_low = 0.0f;
    if (th <= high) g_high = 0.0f;
    if (th > 1.001f) g_high = 0.0f;
    g_thin[idx] = g_low + g_high;
}

inline bool al16(const void *q) { return !q || aligned16(q); }
inline bool al4(const void *q) { return !q || (reinterpret_cast<uintptr_t>(q) & 3u) == 0; }

}  // namespace

EE_API int ee_canny_fwd_f32(const float *x, const float *x_hfs, int B, int C, int H, int W, const float *weights27, const int *dirs16,
                            float alpha, float low, float high, float w, float *edge, float *x_in, uint8_t *gate, void *stream) {
    if (int rc = check_dims(B, C, H, W)) return rc;
    if (B == 0) return EE_OK;
    if (!x || !weights27 || !dirs16 || (!edge && !x_in)) return EE_ERR_NULL;
    if (x_in && !x_hfs) return EE_ERR_NULL;
    CannyParams cp{};
    cp.e.x = x;
    cp.e.x_hfs = x_hfs;
    cp.e.edge = edge;
    cp.e.x_in = x_in;
    cp.e.gate = gate;
    cp.e.H = H; cp.e.W = W;
    cp.e.alpha = alpha; cp.e.high = high; cp.e.w = w;
    cp.low = low;
    cp.e.vec = (W % 4 == 0) && al16(x) && al16(x_hfs) && al16(edge) && al16(x_in) && al4(gate);
    return x_in ? launch<true, false>(cp, weights27, dirs16, B, C, as_stream(stream))
                : launch<false, false>(cp, weights27, dirs16, B, C, as_stream(stream));
}

EE_API int ee_canny_bwd_f32(const float *x, const float *u, const float *g_in, const uint8_t *gate, int B, int C, int H, int W,
                            const float *weights27, const int *dirs16, float alpha, float low, float high, float w, float *g_img,
                            float *g_hfs, void *stream) {
    if (int rc = check_dims(B, C, H, W)) return rc;
    if (B == 0) return EE_OK;
    const bool fused = g_in != nullptr;
    if (!x || !weights27 || !dirs16 || !g_img || (!fused && !u) || (fused && (!gate || !g_hfs))) return EE_ERR_NULL;
    CannyParams cp{};
    cp.e.x = x;
    cp.e.u = u;
    cp.e.g_in = g_in;
    cp.e.gate_in = gate;
    cp.e.g_img = g_img;
    cp.e.g_hfs = g_hfs;
    cp.e.H = H; cp.e.W = W;
    cp.e.alpha = alpha; cp.e.high = high; cp.e.w = w;
    cp.low = low;
    cp.e.vec = (W % 4 == 0) && al16(x) && al16(u) && al16(g_in) && al16(g_img) && al16(g_hfs) && al4(gate);
    return fused ? launch<true, true>(cp, weights27, dirs16, B, C, as_stream(stream))
                 : launch<false, true>(cp, weights27, dirs16, B, C, as_stream(stream));
}

// CannyFilter_BPDA (utils/core.py:386-505): the forward is the CannyFilter pipeline without the alpha mask; thin / t2 are kept
// for the backward.  NaN pixels (NaN input only) are not reproduced: To_compare keeps them, the sign-based forward does not.
EE_API int ee_canny_bpda_fwd_f32(const float *x, int B, int C, int H, int W, const float *weights27, const int *dirs16, float low, float high,
                                 float *edge, float *thin, float *t2, void *stream) {
    if (int rc = check_dims(B, C, H, W)) return rc;
    if (B == 0) return EE_OK;
    if (!x || !weights27 || !dirs16 || !edge || !thin || !t2) return EE_ERR_NULL;
    CannyParams cp{};
    cp.e.x = x;
    cp.e.edge = edge;
    cp.e.H = H; cp.e.W = W;
    cp.e.alpha = 0.0f; cp.e.high = high; cp.e.w = 0.0f;
    cp.low = low;
    cp.thin_out = thin;
    cp.t2_out = t2;
    cp.e.vec = (W % 4 == 0) && al16(x) && al16(edge);
    return launch<false, false>(cp, weights27, dirs16, B, C, as_stream(stream));
}

EE_API int ee_canny_bpda_bwd_f32(const float *x, const float *u, const float *thin, const float *t2, int B, int C, int H, int W,
                                 const float *weights27, const int *dirs16, float low, float high, float *g_thin, float *g_img,
                                 void *stream) {
    if (int rc = check_dims(B, C, H, W)) return rc;
    if (B == 0) return EE_OK;
    if (!x || !u || !thin || !t2 || !weights27 || !dirs16 || !g_thin || !g_img) return EE_ERR_NULL;
    const int64_t total = static_cast<int64_t>(B) * H * W;
    if ((total + 255) / 256 > 0x7fffffffLL) return EE_ERR_SHAPE;
    EE_LAUNCH(bpda_threshold_bwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, as_stream(stream), u, thin, t2, g_thin, H,
              W, low, high, total);
    if (int rc = launch_status()) return rc;
    CannyParams cp{};
    cp.e.x = x;
    cp.e.u = g_thin;
    cp.e.g_img = g_img;
    cp.e.H = H; cp.e.W = W;
    cp.e.alpha = 0.0f; cp.e.high = high; cp.e.w = 0.0f;
    cp.low = low;
    cp.thin_grad = 1;
    cp.e.vec = (W % 4 == 0) && al16(x) && al16(g_thin) && al16(g_img);
    return launch<false, true>(cp, weights27, dirs16, B, C, as_stream(stream));
}
