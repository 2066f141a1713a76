// ee_stencil.hpp - device helpers shared by the edge-filter kernels (ee_edge.hip: CannyFilter_step125_1, ee_canny.hip:
// full CannyFilter): clamped LDS frames, register-blocked blur / Sobel of a 4-pixel group, register-blocked transposed
// 3x3 correlation and the ReplicationPad2d adjoint.  Operation order = oracle/ee_oracle.c (see ee_edge.hip header).
#pragma once
#include <math.h>

#include "ee_common.hpp"

namespace ee {

struct Weights {
    float g[9], sx[9], sy[9];
};

struct EdgeParams {
    const float *x;      // [B,C,H,W]
    const float *x_hfs;  // [B,C,H,W] (fused forward)
    const float *u;      // [B,1,H,W] (plain backward)
    const float *g_in;   // [B,C,H,W] (fused backward)
    const uint8_t *gate_in;
    float *edge;    // [B,1,H,W]
    float *mag;     // [B,1,H,W]
    float *x_in;    // [B,C,H,W]
    uint8_t *gate;  // [B,C,H,W]
    float *g_hfs;   // [B,C,H,W]
    float *g_img;   // [B,1,H,W]
    float *gx_out, *gy_out;        // [B,1,H,W] (forward, optional): the channel-mean Sobel responses, kept for the backward
    const float *gx_in, *gy_in;    // [B,1,H,W] (backward from saved responses)
    int H, W, tiles_x, tiles_y;
    int vec;  // W % 4 == 0 and every pointer 16-B aligned: 16-B global accesses are legal
    float alpha, high, w;
};

constexpr int kColHalo = 4;  // frame columns start at j0 - 4 so that every row of the frame is 16-B aligned

// edge decision for one pixel from the channel-summed Sobel responses (core.py:570-583, To_compare.forward)
template <int C>
__device__ __forceinline__ void edge_from_sums(float ax, float ay, float alpha, float high, float &gx1, float &gy1, float &s2,
                                               float &mag, float &mag_a, float &e) {
    gx1 = ax / static_cast<float>(C);
    gy1 = ay / static_cast<float>(C);
    s2 = gx1 * gx1 + gy1 * gy1;
    mag = sqrtf(s2);
    mag_a = (mag < alpha) ? 0.0f : mag;
    e = (mag_a > high) ? 1.0f : ((mag_a <= high) ? 0.0f : mag_a);
}

// frame[c][r][s] = src(c, clamp(i0 - hr + r), clamp(j0 - COLH + s)),  r < FH, s < FW (FW % 4 == 0); planes of `src`
// are H*W apart.  16-B loads wherever a whole float4 lies inside the image.
// replicate-padded float4 at columns gj .. gj+3 of a row whose width is a multiple of 4 (gj a multiple of 4): ONE aligned,
// always in-bounds 16-B load plus selects - a whole float4 left of the image is row[0] four times, right of it row[W-1]
__device__ __forceinline__ int clamp_col4(int gj, int W) { return gj < 0 ? 0 : (gj > W - 4 ? W - 4 : gj); }
__device__ __forceinline__ float4 replicate4(float4 v, int gj, int W) {
    if (gj < 0) v = make_float4(v.x, v.x, v.x, v.x);
    if (gj >= W) v = make_float4(v.w, v.w, v.w, v.w);
    return v;
}

template <int FH, int FW, int COLH = kColHalo, int PLANES = 0>
__device__ __forceinline__ void load_frame(float *frame, const float *__restrict__ src, int planes, int H, int W, int i0, int j0,
                                           int hr, bool vec) {
    constexpr int F4 = FW / 4;
    if (PLANES > 0 && vec) {
        // every lane's loads are issued back to back on clamped addresses (no predicate: a predicated load is a branch
        // with its own wait, and the frame would cost one memory round trip per iteration instead of one in total)
        constexpr int TOTAL = PLANES * FH * F4, PER = (TOTAL + kBlock - 1) / kBlock;
        float4 v[PER];
        int gjs[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx0 = threadIdx.x + q * kBlock;
            const int idx = idx0 < TOTAL ? idx0 : TOTAL - 1;
            const int c = idx / (FH * F4), rem = idx - c * (FH * F4);
            const int r = rem / F4, f = rem - r * F4;
            const int gi = clampi(i0 - hr + r, 0, H - 1);
            gjs[q] = j0 - COLH + 4 * f;
            v[q] = *reinterpret_cast<const float4 *>(src + (static_cast<size_t>(c) * H + gi) * W + clamp_col4(gjs[q], W));
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx = threadIdx.x + q * kBlock;
            if (idx < TOTAL) *reinterpret_cast<float4 *>(frame + idx * 4) = replicate4(v[q], gjs[q], W);
        }
        return;
    }
    const int total = planes * FH * F4;
    for (int idx = threadIdx.x; idx < total; idx += kBlock) {
        const int c = idx / (FH * F4), rem = idx - c * (FH * F4);
        const int r = rem / F4, f = rem - r * F4;
        const int gi = clampi(i0 - hr + r, 0, H - 1), gj = j0 - COLH + 4 * f;
        const float *row = src + (static_cast<size_t>(c) * H + gi) * W;
        float4 v;
        if (vec && gj >= 0 && gj + 3 < W) {
            v = *reinterpret_cast<const float4 *>(row + gj);
        } else {
            v.x = row[clampi(gj, 0, W - 1)];
            v.y = row[clampi(gj + 1, 0, W - 1)];
            v.z = row[clampi(gj + 2, 0, W - 1)];
            v.w = row[clampi(gj + 3, 0, W - 1)];
        }
        *reinterpret_cast<float4 *>(frame + (c * FH + r) * FW + 4 * f) = v;
    }
}

// us[r][s] = w * ((g_hfs_0 + g_hfs_1) + ...)(clamp(oi + r), clamp(oj + s)), g_hfs_c = gate_c ? g_in_c : 0: the gradient that
// reaches the edge map through the broadcast add and the clamp of the front end (clamped coordinates; cells outside the
// image are never used).  oj must be a multiple of 4.
// SKIP: frame rows [0, SKIP) and [FH - SKIP, FH) are not filled (callers that read u only on an inner band of the frame).
template <int C, int FH, int FW, int SKIP = 0>
__device__ __forceinline__ void load_u_fused(float *us, const float *__restrict__ g_in, const uint8_t *__restrict__ gate, int n, int H, int W,
                                             int oi, int oj, float w, bool vec) {
    constexpr int F4 = FW / 4;
    if (vec) {
        constexpr int TOTAL = (FH - 2 * SKIP) * F4, PER = (TOTAL + kBlock - 1) / kBlock;
        us += SKIP * FW;
        oi += SKIP;
        float4 g4[PER][C];
        uchar4 t4[PER][C];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx0 = threadIdx.x + q * kBlock;
            const int idx = idx0 < TOTAL ? idx0 : TOTAL - 1;
            const int r = idx / F4, f = idx - r * F4;
            const int gi = clampi(oi + r, 0, H - 1), gj = clamp_col4(oj + 4 * f, W);  // columns outside the image are never used
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const size_t o = ((static_cast<size_t>(n) * C + c) * H + gi) * W + gj;
                g4[q][c] = *reinterpret_cast<const float4 *>(g_in + o);
                t4[q][c] = *reinterpret_cast<const uchar4 *>(gate + o);
            }
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx = threadIdx.x + q * kBlock;
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float v[4] = {t4[q][c].x ? g4[q][c].x : 0.0f, t4[q][c].y ? g4[q][c].y : 0.0f, t4[q][c].z ? g4[q][c].z : 0.0f,
                                    t4[q][c].w ? g4[q][c].w : 0.0f};
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = (c == 0) ? v[k] : acc[k] + v[k];
            }
            if (idx < TOTAL) *reinterpret_cast<float4 *>(us + idx * 4) = make_float4(acc[0] * w, acc[1] * w, acc[2] * w, acc[3] * w);
        }
        return;
    }
    for (int idx = threadIdx.x; idx < FH * F4; idx += kBlock) {
        const int r = idx / F4, f = idx - r * F4;
        const int gi = clampi(oi + r, 0, H - 1), gj = oj + 4 * f;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const size_t o = ((static_cast<size_t>(n) * C + c) * H + gi) * W;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cj = clampi(gj + k, 0, W - 1);
                v[k] = gate[o + cj] ? g_in[o + cj] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = (c == 0) ? v[k] : acc[k] + v[k];
        }
        *reinterpret_cast<float4 *>(us + r * FW + 4 * f) = make_float4(acc[0] * w, acc[1] * w, acc[2] * w, acc[3] * w);
    }
}

// Blurred 3x6 neighbourhood (rows i-1..i+1, cols jb-1..jb+4) of a 4-pixel group whose pixels are (i, jb..jb+3),
// for all C channels, from the clamped frame.  (row0, col0) = frame coordinates of image pixel (i-2, jb-2);
// col0 must be even.  core.py:560-563: replicate-pad 1, 3x3 cross-correlation, fmaf chain row-major from 0.
// Positions outside the image take the value of the clamped position (replicate padding of the BLURRED
// plane, core.py:565), which is why the border fix-up copies registers instead of re-blurring.
template <int C, int FH, int FW>
__device__ __forceinline__ void blur_group(const float *frame, const Weights &wt, int row0, int col0, int i, int jb, int H, int W,
                                           float (&b)[C][3][6]) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float xv[5][8];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const float *p = frame + (c * FH + row0 + r) * FW + col0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float2 t = *reinterpret_cast<const float2 *>(p + 2 * q);
                xv[r][2 * q] = t.x;
                xv[r][2 * q + 1] = t.y;
            }
        }
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                float acc = 0.0f;
#pragma unroll
                for (int di = 0; di < 3; ++di)
#pragma unroll
                    for (int dj = 0; dj < 3; ++dj) acc = fmaf(wt.g[di * 3 + dj], xv[rb + di][k + dj], acc);
                b[c][rb][k] = acc;
            }
        // replicate padding of the blurred plane at the image border
        if (i - 1 < 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) b[c][0][k] = b[c][1][k];
        }
        if (i + 1 > H - 1) {
#pragma unroll
            for (int k = 0; k < 6; ++k) b[c][2][k] = b[c][1][k];
        }
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
#pragma unroll
            for (int k = 4; k >= 0; --k)  // columns left of the image take column 0 (groups may start at jb = -2)
                if (jb - 1 + k < 0) b[c][rb][k] = b[c][rb][k + 1];
#pragma unroll
            for (int k = 1; k < 6; ++k)  // columns right of the image take column W-1
                if (jb - 1 + k > W - 1) b[c][rb][k] = b[c][rb][k - 1];
        }
    }
}

// channel-summed Sobel responses of pixel k (0..3) of the group (core.py:565-567): fmaf chain over (kh, kw, c)
template <int C>
__device__ __forceinline__ void sobel_px(const float (&b)[C][3][6], const Weights &wt, int k, float &ax, float &ay) {
    ax = 0.0f;
    ay = 0.0f;
#pragma unroll
    for (int di = 0; di < 3; ++di)
#pragma unroll
        for (int dj = 0; dj < 3; ++dj)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float bv = b[c][di][k + dj];
                ax = fmaf(wt.sx[di * 3 + dj], bv, ax);
                ay = fmaf(wt.sy[di * 3 + dj], bv, ay);
            }
}

__device__ __forceinline__ void tile_origin(const EdgeParams &p, int TH, int TW, int &n, int &i0, int &j0) {
    int t = blockIdx.x;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    n = t / p.tiles_y;
    i0 = ty * TH;
    j0 = tx * TW;
}


// Transposed 3x3 correlation, register-blocked.  cell[q] += fmaf-chain over (di, dj) row-major of
// w[di][dj] * src(P - di, Q0 + q - dj), q < 6, where src(i, j) is the plane value at IMAGE pixel (i, j) read from a frame
// whose (0, 0) is image pixel (oi, oj) and whose cells outside the image / outside the filled range hold 0 ("contributes
// nothing"; zero-weight taps still multiply, 0 * NaN = NaN like a GEMM-based dgrad).  Q0 - 2 - oj must be even.
template <int FW>
__device__ __forceinline__ void cells_row(const float *src, const float *w9, int P, int Q0, int oi, int oj, float (&cell)[6]) {
    float win[3][8];
#pragma unroll
    for (int di = 0; di < 3; ++di) {
        const float *q = src + (P - di - oi) * FW + (Q0 - 2 - oj);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float2 v = *reinterpret_cast<const float2 *>(q + 2 * t);
            win[di][2 * t] = v.x;
            win[di][2 * t + 1] = v.y;
        }
    }
#pragma unroll
    for (int qq = 0; qq < 6; ++qq) {
        float acc = cell[qq];
#pragma unroll
        for (int di = 0; di < 3; ++di)
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) acc = fmaf(w9[di * 3 + dj], win[di][qq + 2 - dj], acc);
        cell[qq] = acc;
    }
}

// ReplicationPad2d(1) adjoint for the 4 pixels (i, jb..jb+3): cells[a][q] holds padded cell (i + a, jb + q); raster-order
// sum of the cells that replicate padding maps onto each pixel.
__device__ __forceinline__ void fold4(const float (&cells)[3][6], int i, int jb, int H, int W, float (&out)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = jb + k;
        float acc = 0.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const bool row_in = (a == 1) || (a == 0 && i == 0) || (a == 2 && i == H - 1);
#pragma unroll
            for (int qd = 0; qd < 3; ++qd) {
                const bool col_in = (qd == 1) || (qd == 0 && j == 0) || (qd == 2 && j == W - 1);
                if (row_in && col_in) acc = acc + cells[a][k + qd];
            }
        }
        out[k] = acc;
    }
}

// The two tail stages every backward shares: gb = pad^T(Sx^T ggx + Sy^T ggy) on rows [i0-1, i0+TH+1) x cols [j0-2, j0+TW+2)
// (4-pixel groups; column j0-2 is built from an incomplete window and only ever feeds a cell the fold discards), then, after
// a barrier, out = pad^T(G^T gb) for the 4 pixels (ti, tjb..tjb+3) of this lane.  gb must be zero-initialised.
template <int TH, int TW, int FW>
__device__ __forceinline__ void adjoint_tail(const float *ggx, const float *ggy, float *gb, const Weights &wt, int H, int W, int i0, int j0,
                                             int oi, int oj, bool live, int ti, int tjb, float (&o4)[4]) {
    constexpr int GB_GX = (TW + 4 + 3) / 4, GB_ROWS = TH + 2;
    for (int gidx = threadIdx.x; gidx < GB_ROWS * GB_GX; gidx += kBlock) {
        const int r = gidx / GB_GX, g = gidx - r * GB_GX;
        const int i = i0 - 1 + r, jb = j0 - 2 + 4 * g;
        if (i >= 0 && i < H && jb < W && jb + 3 >= 0) {
            float cells[3][6];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const bool need = (a == 1) || (a == 0 && i == 0) || (a == 2 && i == H - 1);
#pragma unroll
                for (int q = 0; q < 6; ++q) cells[a][q] = 0.0f;
                if (need) {
                    cells_row<FW>(ggx, wt.sx, i + a, jb, oi, oj, cells[a]);
                    cells_row<FW>(ggy, wt.sy, i + a, jb, oi, oj, cells[a]);
                }
            }
            float o[4];
            fold4(cells, i, jb, H, W, o);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (jb + k >= 0 && jb + k < W) gb[(i - oi) * FW + (jb + k - oj)] = o[k];
        }
    }
    __syncthreads();
    if (!live) return;
    float cells[3][6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const bool need = (a == 1) || (a == 0 && ti == 0) || (a == 2 && ti == H - 1);
#pragma unroll
        for (int q = 0; q < 6; ++q) cells[a][q] = 0.0f;
        if (need) cells_row<FW>(gb, wt.g, ti + a, tjb, oi, oj, cells[a]);
    }
    fold4(cells, ti, tjb, H, W, o4);
}

}  // namespace ee
