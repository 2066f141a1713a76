// ee_edge.hip - CannyFilter_step125_1 (utils/core.py:509-585) forward / backward and the fused EE front
// end, as LDS-tiled stencil kernels for gfx950.
//
// Tiling.  One 256-thread workgroup owns a TH x TW tile of one image (16x64 when W > 32, else 32x32);
// grid = B * tilesY * tilesX workgroups, so a batch of 100 64x64 images launches 400 workgroups over
// the 256 CUs.  Every stage works in ONE local frame of (TH+2*HALO) x (TW+2*HALO) positions whose
// origin is image pixel (i0-HALO, j0-HALO); replicate padding is expressed by clamping coordinates,
// never by materialising a padded tensor.  Each lane produces 4 consecutive pixels of a row, so global
// stores (and the x_hfs / g_in loads) are 16 B per lane, 1 KiB per wave-instruction.
//
// Arithmetic: identical, operation for operation, to oracle/ee_oracle.c (fmaf chains: blur row-major
// over taps, Sobel over (kh, kw, c) with c innermost; IEEE divide / sqrt; -ffp-contract=off), so edge
// bits, gate bits and gradients (NaNs included) are bit-exact against the oracle.
#include <math.h>

#include "ee_common.hpp"

namespace {

using namespace ee;

struct Weights {
    float g[9], sx[9], sy[9];
};

struct EdgeParams {
    const float *x;      // [B,C,H,W]
    const float *x_hfs;  // [B,C,H,W] (fused forward)
    const float *u;      // [B,1,H,W] (plain backward)
    const float *g_in;   // [B,C,H,W] (fused backward)
    const uint8_t *gate_in;
    float *edge;   // [B,1,H,W]
    float *mag;    // [B,1,H,W]
    float *x_in;   // [B,C,H,W]
    uint8_t *gate; // [B,C,H,W]
    float *g_hfs;  // [B,C,H,W]
    float *g_img;  // [B,1,H,W]
    int C, H, W, tiles_x, tiles_y;
    float alpha, high, w;
};

template <int TH, int TW, int HALO>
struct Frame {
    static constexpr int FH = TH + 2 * HALO, FW = TW + 2 * HALO;
    static constexpr int kPlane = FH * FW;
};

// edge decision for one pixel from the channel-summed Sobel responses (core.py:570-583, To_compare.forward)
__device__ __forceinline__ void edge_from_sums(float ax, float ay, int C, float alpha, float high, float &gx1, float &gy1,
                                               float &s2, float &mag, float &mag_a, float &e) {
    gx1 = ax / static_cast<float>(C);
    gy1 = ay / static_cast<float>(C);
    s2 = gx1 * gx1 + gy1 * gy1;
    mag = sqrtf(s2);
    mag_a = (mag < alpha) ? 0.0f : mag;
    e = (mag_a > high) ? 1.0f : ((mag_a <= high) ? 0.0f : mag_a);
}

// ---- stage helpers on the shared local frame ------------------------------------------------------
// xs[c][r][s] = x(n, c, clamp(i0-HALO+r), clamp(j0-HALO+s))
template <int CT, int FH, int FW>
__device__ __forceinline__ void load_x_frame(float *xs, const float *__restrict__ xn, int C, int H, int W, int i0, int j0,
                                             int halo) {
    const int total = C * FH * FW;
    for (int idx = threadIdx.x; idx < total; idx += kBlock) {
        const int c = idx / (FH * FW), rem = idx - c * (FH * FW);
        const int r = rem / FW, s = rem - r * FW;
        const int gi = clampi(i0 - halo + r, 0, H - 1), gj = clampi(j0 - halo + s, 0, W - 1);
        xs[idx] = xn[(static_cast<size_t>(c) * H + gi) * W + gj];
    }
}

// bs[c][r][s] = blur(c, clamp(i0-HALO+r), clamp(j0-HALO+s)) for r in [1, FH-1), s in [1, FW-1)
// (core.py:560-563: replicate-pad 1 then 3x3 cross-correlation, fmaf chain row-major from 0)
template <int FH, int FW>
__device__ __forceinline__ void blur_frame(float *bs, const float *xs, const Weights &wt, int C, int H, int W, int i0, int j0,
                                           int halo) {
    constexpr int IH = FH - 2, IW = FW - 2;
    const int total = IH * IW;
    for (int idx = threadIdx.x; idx < total; idx += kBlock) {
        const int r = 1 + idx / IW, s = 1 + idx % IW;
        const int ci = clampi(i0 - halo + r, 0, H - 1), cj = clampi(j0 - halo + s, 0, W - 1);
        int rr[3], ss[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            rr[d] = clampi(ci + d - 1, 0, H - 1) - (i0 - halo);
            ss[d] = clampi(cj + d - 1, 0, W - 1) - (j0 - halo);
        }
        for (int c = 0; c < C; ++c) {
            const float *xc = xs + c * (FH * FW);
            float acc = 0.0f;
#pragma unroll
            for (int di = 0; di < 3; ++di)
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) acc = fmaf(wt.g[di * 3 + dj], xc[rr[di] * FW + ss[dj]], acc);
            bs[c * (FH * FW) + r * FW + s] = acc;
        }
    }
}

// channel-summed Sobel responses at IMAGE pixel (i, j) from the blurred frame (core.py:565-567)
template <int FH, int FW>
__device__ __forceinline__ void sobel_at(const float *bs, const Weights &wt, int C, int H, int W, int i0, int j0, int halo,
                                         int i, int j, float &ax, float &ay) {
    int rr[3], ss[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        rr[d] = clampi(i + d - 1, 0, H - 1) - (i0 - halo);
        ss[d] = clampi(j + d - 1, 0, W - 1) - (j0 - halo);
    }
    ax = 0.0f;
    ay = 0.0f;
#pragma unroll
    for (int di = 0; di < 3; ++di)
#pragma unroll
        for (int dj = 0; dj < 3; ++dj)
            for (int c = 0; c < C; ++c) {
                const float bv = bs[c * (FH * FW) + rr[di] * FW + ss[dj]];
                ax = fmaf(wt.sx[di * 3 + dj], bv, ax);
                ay = fmaf(wt.sy[di * 3 + dj], bv, ay);
            }
}

// =====================================================================================================
// forward:  edge map, optionally fused with  x_in = clamp(x_hfs + w*edge, 0, 1)  and the clamp gate
// =====================================================================================================
template <int TH, int TW, bool FUSED, bool VEC>
__global__ __launch_bounds__(kBlock) void edge_fwd_kernel(EdgeParams p, Weights wt) {
    constexpr int HALO = 2;
    using F = Frame<TH, TW, HALO>;
    extern __shared__ __align__(16) float lds[];
    const int C = p.C, H = p.H, W = p.W;
    float *xs = lds;                 // [C][FH][FW]
    float *bs = lds + C * F::kPlane; // [C][FH][FW]

    int t = blockIdx.x;
    const int tx_ = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty_ = t % p.tiles_y;
    const int n = t / p.tiles_y;
    const int i0 = ty_ * TH, j0 = tx_ * TW;
    const float *xn = p.x + static_cast<size_t>(n) * C * H * W;

    load_x_frame<0, F::FH, F::FW>(xs, xn, C, H, W, i0, j0, HALO);
    __syncthreads();
    blur_frame<F::FH, F::FW>(bs, xs, wt, C, H, W, i0, j0, HALO);
    __syncthreads();

    constexpr int LANES_X = TW / 4;
    const int lx = threadIdx.x % LANES_X, ly = threadIdx.x / LANES_X;
    const int i = i0 + ly, jb = j0 + 4 * lx;
    if (i >= H || jb >= W) return;

    float e[4], m[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = jb + k;
        e[k] = 0.0f;
        m[k] = 0.0f;
        if (j < W) {
            float ax, ay, gx1, gy1, s2, mag_a;
            sobel_at<F::FH, F::FW>(bs, wt, C, H, W, i0, j0, HALO, i, j, ax, ay);
            edge_from_sums(ax, ay, C, p.alpha, p.high, gx1, gy1, s2, m[k], mag_a, e[k]);
        }
    }
    const size_t pix = (static_cast<size_t>(n) * H + i) * W + jb;
    if (VEC) {
        if (p.edge) *reinterpret_cast<float4 *>(p.edge + pix) = make_float4(e[0], e[1], e[2], e[3]);
        if (p.mag) *reinterpret_cast<float4 *>(p.mag + pix) = make_float4(m[0], m[1], m[2], m[3]);
    } else {
        for (int k = 0; k < 4 && jb + k < W; ++k) {
            if (p.edge) p.edge[pix + k] = e[k];
            if (p.mag) p.mag[pix + k] = m[k];
        }
    }
    if (FUSED) {
        for (int c = 0; c < C; ++c) {
            const size_t o = ((static_cast<size_t>(n) * C + c) * H + i) * W + jb;
            if (VEC) {
                const float4 xh = *reinterpret_cast<const float4 *>(p.x_hfs + o);
                const float s0 = xh.x + p.w * e[0], s1 = xh.y + p.w * e[1], s2_ = xh.z + p.w * e[2], s3 = xh.w + p.w * e[3];
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
                    const float s = p.x_hfs[o + k] + p.w * e[k];
                    p.x_in[o + k] = tclamp(s, 0.0f, 1.0f);
                    if (p.gate) p.gate[o + k] = (s >= 0.0f && s <= 1.0f);
                }
            }
        }
    }
}

// =====================================================================================================
// backward:  g_img = adjoint of the edge filter applied to u (NaN-faithful); fused variant also forms
//            g_hfs = g_in*gate and u = w * sum_c g_hfs_c itself.
// All stages share one (TH+8) x (TW+8) frame; a stage writes 0 wherever its quantity is outside the
// image or not needed, so later stages may read any in-frame neighbour unconditionally.
// =====================================================================================================
template <int TH, int TW, bool FUSED, bool VEC>
__global__ __launch_bounds__(kBlock) void edge_bwd_kernel(EdgeParams p, Weights wt) {
    constexpr int HALO = 4;
    using F = Frame<TH, TW, HALO>;
    constexpr int FH = F::FH, FW = F::FW, PL = F::kPlane;
    extern __shared__ __align__(16) float lds[];
    const int C = p.C, H = p.H, W = p.W;
    const int nA = C > 2 ? C : 2;
    float *poolA = lds;             // xs[C]  -> later ggx, ggy
    float *poolB = lds + nA * PL;   // bs[C]  -> later gbp, gb, dxp   (max(C,3) planes)

    int t = blockIdx.x;
    const int tx_ = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty_ = t % p.tiles_y;
    const int n = t / p.tiles_y;
    const int i0 = ty_ * TH, j0 = tx_ * TW;
    const int oi = i0 - HALO, oj = j0 - HALO;  // image coordinates of local (0,0)
    const float *xn = p.x + static_cast<size_t>(n) * C * H * W;

    // ---- A, B: x frame and blurred frame (rows/cols [1, F-1)) ------------------------------------
    load_x_frame<0, FH, FW>(poolA, xn, C, H, W, i0, j0, HALO);
    __syncthreads();
    blur_frame<FH, FW>(poolB, poolA, wt, C, H, W, i0, j0, HALO);
    __syncthreads();

    // ---- C: per-pixel gradient of the magnitude stage, local rows/cols [2, F-2), 0 elsewhere ----------
    // ggx/ggy overwrite the x planes, which are dead once the blurred frame exists
    float *ggx = poolA, *ggy = poolA + PL;
    for (int idx = threadIdx.x; idx < PL; idx += kBlock) {
        const int r = idx / FW, s = idx - r * FW;
        const int i = oi + r, j = oj + s;
        float vx = 0.0f, vy = 0.0f;
        if (r >= 2 && r < FH - 2 && s >= 2 && s < FW - 2 && i >= 0 && i < H && j >= 0 && j < W) {
            float uu;
            if (FUSED) {
                float acc = 0.0f;
                for (int c = 0; c < C; ++c) {
                    const size_t o = ((static_cast<size_t>(n) * C + c) * H + i) * W + j;
                    const float v = p.gate_in[o] ? p.g_in[o] : 0.0f;
                    acc = (c == 0) ? v : acc + v;
                }
                uu = acc * p.w;
            } else {
                uu = p.u[(static_cast<size_t>(n) * H + i) * W + j];
            }
            float ax, ay, gx1, gy1, s2, mag, mag_a, e;
            sobel_at<FH, FW>(poolB, wt, C, H, W, i0, j0, HALO, i, j, ax, ay);
            edge_from_sums(ax, ay, C, p.alpha, p.high, gx1, gy1, s2, mag, mag_a, e);
            float gm = uu;
            if (mag_a <= p.high) gm = 0.0f;     // To_compare.backward core.py:356
            if (mag_a > 1.001f) gm = 0.0f;      // core.py:357
            if (mag < p.alpha) gm = 0.0f;       // where() backward core.py:575
            const float rs = 1.0f / sqrtf(s2);  // pow(s2, -0.5): 0 -> inf
            const float gs = gm * (0.5f * rs);  // 0*inf = NaN kept
            vx = (gs * (2.0f * gx1)) / static_cast<float>(C);
            vy = (gs * (2.0f * gy1)) / static_cast<float>(C);
        }
        ggx[idx] = vx;
        ggy[idx] = vy;
    }
    __syncthreads();  // gg complete; the blurred planes (poolB) are dead from here on

    // From here on a local (r, s) of a PADDED-domain array denotes padded position (oi + r + 1, oj + s + 1),
    // i.e. the padded cell that sits on top of image pixel (oi + r, oj + s); pad row 0 <-> image row -1.
    float *gbp = poolB, *gb = poolB + PL, *dxp = poolB + 2 * PL;

    // transposed 3x3 correlation into the padded domain:  out(P,Q) = sum_{di,dj} w[di][dj] * src(P-di, Q-dj)
    // (src at image coordinates, zero outside the image; zero-weight taps still multiply: 0*NaN = NaN)
    auto corrT = [&](const float *src0, const float *w0, const float *src1, const float *w1, float *dst, int lo, int hiR,
                     int hiS) {
        for (int idx = threadIdx.x; idx < PL; idx += kBlock) {
            const int r = idx / FW, s = idx - r * FW;
            float acc = 0.0f;
            if (r >= lo && r < hiR && s >= lo && s < hiS) {
                const int P = oi + r + 1, Q = oj + s + 1;  // padded coordinates
                if (P >= 0 && P <= H + 1 && Q >= 0 && Q <= W + 1) {
                    for (int pass = 0; pass < 2; ++pass) {
                        const float *src = pass ? src1 : src0;
                        const float *wv = pass ? w1 : w0;
                        if (!src) break;
#pragma unroll
                        for (int di = 0; di < 3; ++di)
#pragma unroll
                            for (int dj = 0; dj < 3; ++dj) {
                                const int i = P - di, j = Q - dj;  // image coords of the contributing output
                                if (i < 0 || i >= H || j < 0 || j >= W) continue;
                                acc = fmaf(wv[di * 3 + dj], src[(i - oi) * FW + (j - oj)], acc);
                            }
                    }
                }
            }
            dst[idx] = acc;
        }
    };
    // adjoint of ReplicationPad2d(1): image pixel (i,j) gathers its padded cells in raster order
    auto fold = [&](const float *srcp, int i, int j) -> float {
        const int P0 = (i == 0) ? 0 : i + 1, P1 = (i == H - 1) ? H + 1 : i + 1;
        const int Q0 = (j == 0) ? 0 : j + 1, Q1 = (j == W - 1) ? W + 1 : j + 1;
        float acc = 0.0f;
        for (int P = P0; P <= P1; ++P)
            for (int Q = Q0; Q <= Q1; ++Q) acc = acc + srcp[(P - 1 - oi) * FW + (Q - 1 - oj)];
        return acc;
    };

    // ---- D: gbp = Sx^T ggx + Sy^T ggy on padded cells, local [2, F-1) ------------------------------
    corrT(ggx, wt.sx, ggy, wt.sy, gbp, 2, FH - 1, FW - 1);
    __syncthreads();
    // ---- E: gb = fold(gbp) on image pixels, local [3, F-3) ------------------------------------------
    for (int idx = threadIdx.x; idx < PL; idx += kBlock) {
        const int r = idx / FW, s = idx - r * FW;
        const int i = oi + r, j = oj + s;
        float v = 0.0f;
        if (r >= 3 && r < FH - 3 && s >= 3 && s < FW - 3 && i >= 0 && i < H && j >= 0 && j < W) v = fold(gbp, i, j);
        gb[idx] = v;
    }
    __syncthreads();
    // ---- F: dxp = G^T gb on padded cells, local [3, F-2) ---------------------------------------------
    corrT(gb, wt.g, nullptr, nullptr, dxp, 3, FH - 2, FW - 2);
    __syncthreads();

    // ---- G: fold onto the tile's own pixels and store; fused: also g_hfs = g_in * gate --------------
    constexpr int LANES_X = TW / 4;
    const int lx = threadIdx.x % LANES_X, ly = threadIdx.x / LANES_X;
    const int i = i0 + ly, jb = j0 + 4 * lx;
    if (i >= H || jb >= W) return;
    float o4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o4[k] = (jb + k < W) ? fold(dxp, i, jb + k) : 0.0f;
    const size_t pix = (static_cast<size_t>(n) * H + i) * W + jb;
    if (VEC) {
        *reinterpret_cast<float4 *>(p.g_img + pix) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    } else {
        for (int k = 0; k < 4 && jb + k < W; ++k) p.g_img[pix + k] = o4[k];
    }
    if (FUSED) {
        for (int c = 0; c < C; ++c) {
            const size_t o = ((static_cast<size_t>(n) * C + c) * H + i) * W + jb;
            if (VEC) {
                const float4 g = *reinterpret_cast<const float4 *>(p.g_in + o);
                const uchar4 gt = *reinterpret_cast<const uchar4 *>(p.gate_in + o);
                *reinterpret_cast<float4 *>(p.g_hfs + o) =
                    make_float4(gt.x ? g.x : 0.0f, gt.y ? g.y : 0.0f, gt.z ? g.z : 0.0f, gt.w ? g.w : 0.0f);
            } else {
                for (int k = 0; k < 4 && jb + k < W; ++k) p.g_hfs[o + k] = p.gate_in[o + k] ? p.g_in[o + k] : 0.0f;
            }
        }
    }
}

// ---- host side --------------------------------------------------------------------------------------
struct Launch {
    int th, tw, tiles_x, tiles_y;
    unsigned grid;
};

Launch plan(int B, int H, int W) {
    Launch l;
    if (W > 32) {
        l.th = 16;
        l.tw = 64;
    } else {
        l.th = 32;
        l.tw = 32;
    }
    l.tiles_x = (W + l.tw - 1) / l.tw;
    l.tiles_y = (H + l.th - 1) / l.th;
    l.grid = static_cast<unsigned>(static_cast<int64_t>(B) * l.tiles_x * l.tiles_y);
    return l;
}

int check_dims(int B, int C, int H, int W) {
    if (B < 0 || C < 1 || H < 1 || W < 1) return EE_ERR_SHAPE;
    if (C > 4) return EE_ERR_UNSUPPORTED;
    if (static_cast<int64_t>(B) * ((H + 15) / 16) * ((W + 31) / 32) > 0x7fffffffLL) return EE_ERR_SHAPE;
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

template <bool FUSED>
int launch_fwd(const EdgeParams &p0, const Weights &wt, int B, bool vec, hipStream_t s) {
    EdgeParams p = p0;
    const Launch l = plan(B, p.H, p.W);
    p.tiles_x = l.tiles_x;
    p.tiles_y = l.tiles_y;
    if (l.grid == 0) return EE_OK;
    if (l.tw == 64) {
        const size_t lds = sizeof(float) * 2 * p.C * Frame<16, 64, 2>::kPlane;
        if (vec)
            EE_LAUNCH((edge_fwd_kernel<16, 64, FUSED, true>), dim3(l.grid), dim3(kBlock), lds, s, p, wt);
        else
            EE_LAUNCH((edge_fwd_kernel<16, 64, FUSED, false>), dim3(l.grid), dim3(kBlock), lds, s, p, wt);
    } else {
        const size_t lds = sizeof(float) * 2 * p.C * Frame<32, 32, 2>::kPlane;
        if (vec)
            EE_LAUNCH((edge_fwd_kernel<32, 32, FUSED, true>), dim3(l.grid), dim3(kBlock), lds, s, p, wt);
        else
            EE_LAUNCH((edge_fwd_kernel<32, 32, FUSED, false>), dim3(l.grid), dim3(kBlock), lds, s, p, wt);
    }
    return launch_status();
}

template <bool FUSED>
int launch_bwd(const EdgeParams &p0, const Weights &wt, int B, bool vec, hipStream_t s) {
    EdgeParams p = p0;
    const Launch l = plan(B, p.H, p.W);
    p.tiles_x = l.tiles_x;
    p.tiles_y = l.tiles_y;
    if (l.grid == 0) return EE_OK;
    const int planes = (p.C > 2 ? p.C : 2) + (p.C > 3 ? p.C : 3);
    if (l.tw == 64) {
        const size_t lds = sizeof(float) * planes * Frame<16, 64, 4>::kPlane;
        if (vec)
            EE_LAUNCH((edge_bwd_kernel<16, 64, FUSED, true>), dim3(l.grid), dim3(kBlock), lds, s, p, wt);
        else
            EE_LAUNCH((edge_bwd_kernel<16, 64, FUSED, false>), dim3(l.grid), dim3(kBlock), lds, s, p, wt);
    } else {
        const size_t lds = sizeof(float) * planes * Frame<32, 32, 4>::kPlane;
        if (vec)
            EE_LAUNCH((edge_bwd_kernel<32, 32, FUSED, true>), dim3(l.grid), dim3(kBlock), lds, s, p, wt);
        else
            EE_LAUNCH((edge_bwd_kernel<32, 32, FUSED, false>), dim3(l.grid), dim3(kBlock), lds, s, p, wt);
    }
    return launch_status();
}

}  // namespace

EE_API int ee_edge125_fwd_f32(const float *x, int B, int C, int H, int W, const float *weights27, float alpha, float high,
                              float *edge, float *mag, void *stream) {
    if (!x || !weights27 || !edge) return EE_ERR_NULL;
    if (int rc = check_dims(B, C, H, W)) return rc;
    EdgeParams p{};
    p.x = x;
    p.edge = edge;
    p.mag = mag;
    p.C = C; p.H = H; p.W = W;
    p.alpha = alpha; p.high = high; p.w = 0.0f;
    const bool vec = (W % 4 == 0) && aligned16(edge) && (!mag || aligned16(mag));
    ProfScope prof(EE_K_EDGE_FWD, as_stream(stream));
    return launch_fwd<false>(p, load_weights(weights27), B, vec, as_stream(stream));
}

EE_API int ee_edge125_bwd_f32(const float *x, const float *u, int B, int C, int H, int W, const float *weights27, float alpha,
                              float high, float *g_img, void *stream) {
    if (!x || !u || !weights27 || !g_img) return EE_ERR_NULL;
    if (int rc = check_dims(B, C, H, W)) return rc;
    EdgeParams p{};
    p.x = x;
    p.u = u;
    p.g_img = g_img;
    p.C = C; p.H = H; p.W = W;
    p.alpha = alpha; p.high = high; p.w = 0.0f;
    const bool vec = (W % 4 == 0) && aligned16(g_img);
    ProfScope prof(EE_K_EDGE_BWD, as_stream(stream));
    return launch_bwd<false>(p, load_weights(weights27), B, vec, as_stream(stream));
}

EE_API int ee_frontend_fwd_f32(const float *x, const float *x_hfs, int B, int C, int H, int W, const float *weights27,
                               float alpha, float high, float w, float *x_in, uint8_t *gate, float *edge, void *stream) {
    if (!x || !x_hfs || !weights27 || !x_in) return EE_ERR_NULL;
    if (int rc = check_dims(B, C, H, W)) return rc;
    EdgeParams p{};
    p.x = x;
    p.x_hfs = x_hfs;
    p.x_in = x_in;
    p.gate = gate;
    p.edge = edge;
    p.C = C; p.H = H; p.W = W;
    p.alpha = alpha; p.high = high; p.w = w;
    const bool vec = (W % 4 == 0) && aligned16(x_hfs) && aligned16(x_in) && (!edge || aligned16(edge)) &&
                     (!gate || (reinterpret_cast<uintptr_t>(gate) & 3u) == 0);
    ProfScope prof(EE_K_FRONTEND_FWD, as_stream(stream));
    return launch_fwd<true>(p, load_weights(weights27), B, vec, as_stream(stream));
}

EE_API int ee_frontend_bwd_f32(const float *g_in, const uint8_t *gate, const float *x, int B, int C, int H, int W,
                               const float *weights27, float alpha, float high, float w, float *g_hfs, float *g_edge,
                               void *stream) {
    if (!g_in || !gate || !x || !weights27 || !g_hfs || !g_edge) return EE_ERR_NULL;
    if (int rc = check_dims(B, C, H, W)) return rc;
    EdgeParams p{};
    p.x = x;
    p.g_in = g_in;
    p.gate_in = gate;
    p.g_hfs = g_hfs;
    p.g_img = g_edge;
    p.C = C; p.H = H; p.W = W;
    p.alpha = alpha; p.high = high; p.w = w;
    const bool vec = (W % 4 == 0) && aligned16(g_in) && aligned16(g_hfs) && aligned16(g_edge) &&
                     (reinterpret_cast<uintptr_t>(gate) & 3u) == 0;
    ProfScope prof(EE_K_FRONTEND_BWD, as_stream(stream));
    return launch_bwd<true>(p, load_weights(weights27), B, vec, as_stream(stream));
}
