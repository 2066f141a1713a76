// ee_conv.hip - the ResNet shortcut convolution: Conv2d(Cin, Cout, kernel_size=1, stride=2, bias=False)
// (Tiny_ImageNet/models_tinyimagenet/resnet.py:137-142), forward and backward-data, on the exact-f32 matrix cores.
//
// At the reference batch these are 0.1 GFLOP products ([100*8*8, 64] x [64, 128] and two smaller ones); MIOpen answers
// them with an NHWC implicit-GEMM solver wrapped in layout transposes - 25 us each way per layer on MI355X (torch events,
// scripts/conv_layers.py), six such launches sequences per PGD iteration.  Here: one wavefront per 32 x 32 output tile,
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: same numerics class as an FMA loop), operands read straight from
// global memory (everything is L2-resident), output channels on the accumulator rows so that stores run along pixels.
//   forward : Y[co][p] = sum_ci W[co][ci] * X[ci][p@stride2]
//   backward: dX[ci][p@stride2] = sum_co W[co][ci] * dY[co][p];  the other three pixels of every 2x2 cell are zero.
// The weight gradient (once per training step) stays on MIOpen.
//
// CNN-body glue, not a row of SURVEY.md section 8: parity is "logits within 1e-4" through the model tests.
#include <stdlib.h>

#include "ee_common.hpp"

namespace {

using namespace ee;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvDims {
    int B, Cin, Cout, H, W, OH, OW;
};

// accumulator register r of lane l holds D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// One product for both directions, D[c][p] = sum_k A[c][k] * Bm[k][p] on 32 x 32 tiles:
//   forward : c = output channel, k = input channel,  A = W[c][k],  Bm = x[n, k, 2*oh, 2*ow]
//   backward: c = input channel,  k = output channel, A = W[k][c],  Bm = dy[n, k, oh, ow]
// A workgroup is 4 wavefronts = (4 / KS channel tiles) x (KS slices of k); every wavefront issues ALL its loads for 16 k-steps
// ahead of the matching MFMAs (the products are tiny, the kernel is bound by load latency, not by the matrix cores), and the
// KS partial tiles meet in LDS and are added in slice order.  grid (ceil(P / 32), ceil(C / (32 * 4 / KS))).
template <int KS, bool FWD>
__global__ __launch_bounds__(256) void conv1x1s2_kernel(const float *__restrict__ act, const float *__restrict__ w, float *__restrict__ out,
                                                        ConvDims d) {
    constexpr int CT = 4 / KS;  // channel tiles per workgroup
    __shared__ float part[KS > 1 ? 4 * 32 * 33 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int ct = wave / KS, ks = wave - ct * KS;
    const int OHW = d.OH * d.OW, P = d.B * OHW;
    const int Cdim = FWD ? d.Cout : d.Cin, Kdim = FWD ? d.Cin : d.Cout;
    const int p = static_cast<int>(blockIdx.x) * 32 + i;
    const int c0 = (static_cast<int>(blockIdx.y) * CT + ct) * 32;
    const int c = c0 + i;
    const bool pv = p < P, cv = c < Cdim;
    const int n = pv ? p / OHW : 0, rem = pv ? p - n * OHW : 0;  // out-of-range lanes read pixel 0 / channel 0 and are masked:
    const int oh = rem / d.OW, ow = rem - oh * d.OW;             // unconditional loads batch, predicated ones serialise
    const int steps = Kdim / (2 * KS);                 // k-steps of this wavefront (2 values of k per step)
    const int k0 = ks * (Kdim / KS) + kk * steps;      // this lane's first k
    const size_t plane = static_cast<size_t>(d.H) * d.W;
    const float *bp;
    size_t bstride;
    const float *ap;
    size_t astride;
    if (FWD) {
        bp = act + (static_cast<size_t>(n) * d.Cin + k0) * plane + static_cast<size_t>(2 * oh) * d.W + 2 * ow;
        bstride = plane;
        ap = w + static_cast<size_t>(cv ? c : 0) * d.Cin + k0;
        astride = 1;
    } else {
        bp = act + (static_cast<size_t>(n) * d.Cout + k0) * OHW + rem;
        bstride = static_cast<size_t>(OHW);
        ap = w + static_cast<size_t>(k0) * d.Cin + (cv ? c : 0);
        astride = static_cast<size_t>(d.Cin);
    }
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < Cdim) {
        float a0[16], b0[16], a1[16], b1[16];
        auto load = [&](float (&av)[16], float (&bv)[16], int t0) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                // every load is issued (clamped index, valid address) and masked afterwards: a predicated load becomes a
                // branch around it, and sixteen branches serialise what should be sixteen loads in flight
                const int tc = (t0 + u < steps) ? t0 + u : steps - 1;
                const float keep = (t0 + u < steps) ? 1.0f : 0.0f;
                const float a = ap[static_cast<size_t>(tc) * astride], b = bp[static_cast<size_t>(tc) * bstride];
                av[u] = cv ? a * keep : 0.0f;
                bv[u] = pv ? b : 0.0f;
            }
        };
        auto fma16 = [&](const float (&av)[16], const float (&bv)[16]) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        };
        load(a0, b0, 0);
        for (int t = 0; t < steps; t += 32) {
            if (t + 16 < steps) load(a1, b1, t + 16);
            fma16(a0, b0);
            if (t + 16 < steps) {
                if (t + 32 < steps) load(a0, b0, t + 32);
                fma16(a1, b1);
            }
        }
    }
    // ---- meet the KS partial tiles in LDS (slice order: deterministic), then store along pixels ---------------------------
    if (KS > 1) {
        float *mine = part + wave * (32 * 33);
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[acc_row(r, lane) * 33 + i] = acc[r];
        __syncthreads();
        if (ks != 0) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
            for (int s2 = 1; s2 < KS; ++s2) v += part[(wave + s2) * (32 * 33) + acc_row(r, lane) * 33 + i];
            acc[r] = v;
        }
    }
    if (!pv || c0 >= Cdim) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cc = c0 + acc_row(r, lane);
        if (cc >= Cdim) continue;
        if (FWD) {
            out[(static_cast<size_t>(n) * d.Cout + cc) * OHW + rem] = acc[r];
        } else {
            float *o = out + ((static_cast<size_t>(n) * d.Cin + cc) * d.H + 2 * oh) * d.W + 2 * ow;
            *reinterpret_cast<float2 *>(o) = make_float2(acc[r], 0.0f);
            *reinterpret_cast<float2 *>(o + d.W) = make_float2(0.0f, 0.0f);
        }
    }
}

template <bool FWD>
int conv_launch(const float *act, const float *w, float *out, const ConvDims &d, hipStream_t st) {
    const int Cdim = FWD ? d.Cout : d.Cin, Kdim = FWD ? d.Cin : d.Cout;
    const int P = d.B * d.OH * d.OW;
    const unsigned gx = static_cast<unsigned>((P + 31) / 32);
    // slices of k per workgroup: keep ~32 MFMAs per wavefront so that one round of loads covers the whole product
    if (Kdim % 8 == 0 && Kdim >= 256)
        EE_LAUNCH((conv1x1s2_kernel<4, FWD>), dim3(gx, static_cast<unsigned>((Cdim + 31) / 32)), dim3(256), 0, st, act, w, out, d);
    else if (Kdim % 4 == 0 && Kdim >= 128)
        EE_LAUNCH((conv1x1s2_kernel<2, FWD>), dim3(gx, static_cast<unsigned>((Cdim + 63) / 64)), dim3(256), 0, st, act, w, out, d);
    else
        EE_LAUNCH((conv1x1s2_kernel<1, FWD>), dim3(gx, static_cast<unsigned>((Cdim + 127) / 128)), dim3(256), 0, st, act, w, out, d);
    return launch_status();
}

int conv_check(int B, int Cin, int Cout, int H, int W) {
    if (B < 0 || Cin < 2 || Cout < 2 || H < 2 || W < 2) return EE_ERR_SHAPE;
    if ((Cin & 1) || (Cout & 1) || (H & 1) || (W & 1)) return EE_ERR_UNSUPPORTED;
    if (static_cast<int64_t>(B) * (H / 2) * (W / 2) > 0x7fffffffLL) return EE_ERR_SHAPE;
    return EE_OK;
}

}  // namespace

EE_API int ee_conv1x1s2_fwd_f32(const float *x, const float *weight, float *y, int B, int Cin, int Cout, int H, int W, void *stream) {
    if (int rc = conv_check(B, Cin, Cout, H, W)) return rc;
    if (B == 0) return EE_OK;
    if (!x || !weight || !y) return EE_ERR_NULL;
    return conv_launch<true>(x, weight, y, ConvDims{B, Cin, Cout, H, W, H / 2, W / 2}, as_stream(stream));
}

EE_API int ee_conv1x1s2_bwd_f32(const float *dy, const float *weight, float *dx, int B, int Cin, int Cout, int H, int W, void *stream) {
    if (int rc = conv_check(B, Cin, Cout, H, W)) return rc;
    if (B == 0) return EE_OK;
    if (!dy || !weight || !dx) return EE_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(dx) & 7u) return EE_ERR_ALIGN;
    return conv_launch<false>(dy, weight, dx, ConvDims{B, Cin, Cout, H, W, H / 2, W / 2}, as_stream(stream));
}

// =====================================================================================================================
// Backward-data of the stem convolution Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
// (Tiny_ImageNet/models_tinyimagenet/resnet.py:112-113): the last step of every PGD iteration's backward pass, the gradient
// with respect to the IMAGE.  With three output channels the generic solvers have nothing to tile over (MIOpen: 110 us
// for [100,64,32,32] -> [100,3,64,64] plus a zero fill, rocprofv3; 172 us with torch events).
//
// dx[n,c,2a+ph,2b+pw] = sum_k sum_{u,v} dy[n,k,a-1+u,b-1+v] * W[k,c,r(ph,u),s(pw,v)],  r(0,u) = 5-2u (u < 3), r(1,u) = 6-2u (u < 4)
// i.e. a 4x4-window correlation on the dy grid producing the 2x2x3 = 12 values of one input cell: one lane per cell, 12
// accumulators - which is a GEMM with 12 of 16 columns used.  History (rocprofv3, [100,64,32,32]): VALU version with the
// 147 weights of one k as wave-uniform scalar operands 97 us; one v_fmac per weight forced 112 us (37 KB of weights miss the
// scalar cache); weights as LDS broadcast reads 130 us; the MFMA form below 45 us.
// =====================================================================================================================
namespace {

constexpr int ST_TA = 8, ST_TB = 32;  // dy-grid tile (rows x cols) of one workgroup

// D[pixel][j] with j = (ph*2 + pw)*3 + c (12 of 16 columns used), K = (channel, u, v): one v_mfma_f32_16x16x4_f32 covers the four
// v of one (channel, u) for 16 neighbouring cells of a dy row.  A workgroup (4 wavefronts) owns 8 x 32 cells = 16 M-tiles,
// four per wavefront; per round 16 channels of the dy frame and their rearranged weights W'[ch][u][v][j] (zero where the
// tap does not exist for that parity) sit in LDS; the A operand is one ds_read_b32 per MFMA, B is shared by the four M-tiles.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SM_TA = 8, SM_TB = 32, SM_KC = 16, SM_FH = SM_TA + 3, SM_FW = 36;

__global__ __launch_bounds__(256) void stem_bwd_data_mfma_kernel(const float *__restrict__ dy, const float *__restrict__ w, float *__restrict__ dx,
                                                                 int K, int OH, int OW, int tiles_a, int tiles_b) {
    __shared__ __align__(16) float fr[SM_KC * SM_FH * SM_FW];  // [ch][frame row][frame col]; reused as the output tile [3][16][64]
    __shared__ float wp[SM_KC * 4 * 4 * 16];                   // [ch][u][v][j]
    static_assert(SM_KC * SM_FH * SM_FW >= 3 * 16 * 64, "output tile must fit the frame buffer");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    int bid = blockIdx.x;
    const int tjb = bid % tiles_b;
    bid /= tiles_b;
    const int tia = bid % tiles_a;
    const int n = bid / tiles_a;
    const int a0 = tia * SM_TA, b0 = tjb * SM_TB;
    const float *dyn = dy + static_cast<size_t>(n) * K * OH * OW;
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // column j of the product -> (ph, pw, c)
    const int jc = i % 3, jq = i / 3, jph = jq >> 1, jpw = jq & 1;
    // staging roles, fixed per thread for the whole kernel (no index arithmetic inside the channel loops):
    //   weights: thread t fills wp[k][u][v][j] for (u, v, j) = (t >> 6, (t >> 4) & 3, t & 15) and every k of the round
    //   frame  : the 11 x 36 frame positions go to threads t (position t) and t < 140 (position 256 + t)
    const int su = threadIdx.x >> 6, sv = (threadIdx.x >> 4) & 3;
    const bool wvalid = i < 12 && su < 3 + jph && sv < 3 + jpw;
    const int woff = jc * 49 + (5 + jph - 2 * su) * 7 + (5 + jpw - 2 * sv);
    int foff[2];
    bool fin[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pos = q * 256 + static_cast<int>(threadIdx.x);
        const int frow = pos / SM_FW, fc = pos - frow * SM_FW;
        const int r = a0 - 1 + frow, c = b0 - 1 + fc;
        fin[q] = pos < SM_FH * SM_FW && r >= 0 && r < OH && c >= 0 && c < OW;
        foff[q] = fin[q] ? r * OW + c : 0;  // loads are unconditional (always in bounds) and masked afterwards, so that they batch
    }
    const size_t chan = static_cast<size_t>(OH) * OW;
    for (int kc = 0; kc < K; kc += SM_KC) {
        __syncthreads();
        float v0[SM_KC], v1[SM_KC], vw[SM_KC];
#pragma unroll
        for (int k = 0; k < SM_KC; ++k) {
            const size_t kch = static_cast<size_t>(kc + k < K ? kc + k : K - 1);
            v0[k] = dyn[kch * chan + foff[0]];
            v1[k] = dyn[kch * chan + foff[1]];
            vw[k] = w[kch * 147 + (wvalid ? woff : 0)];
        }
#pragma unroll
        for (int k = 0; k < SM_KC; ++k) {
            const bool kin = kc + k < K;
            fr[k * (SM_FH * SM_FW) + threadIdx.x] = (kin && fin[0]) ? v0[k] : 0.0f;
            if (threadIdx.x < SM_FH * SM_FW - 256) fr[k * (SM_FH * SM_FW) + 256 + threadIdx.x] = (kin && fin[1]) ? v1[k] : 0.0f;
            wp[k * 256 + threadIdx.x] = (kin && wvalid) ? vw[k] : 0.0f;
        }
        __syncthreads();
        for (int k = 0; k < SM_KC; ++k) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float b = wp[((k * 4 + u) * 4 + kk) * 16 + i];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int row = wave * 2 + (mt >> 1), col0 = (mt & 1) * 16;
                    const float a = fr[(k * SM_FH + row + u) * SM_FW + col0 + i + kk];
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[mt], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();  // the frame is dead: it becomes the [3][16][64] output tile
    if (i < 12) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int row = wave * 2 + (mt >> 1), col0 = (mt & 1) * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int px = col0 + kk * 4 + r;  // accumulator register r of lane (i, kk) holds D[pixel kk*4 + r][column i]
                fr[(jc * 16 + 2 * row + jph) * 64 + 2 * px + jpw] = acc[mt][r];
            }
        }
    }
    __syncthreads();
    const int H = 2 * OH, W = 2 * OW;
    for (int e = threadIdx.x; e < 3 * 16 * 16; e += 256) {
        const int q4 = e & 15, orow = (e >> 4) & 15, c = e >> 8;
        const int h = 2 * a0 + orow, wcol = 2 * b0 + 4 * q4;
        if (h < H && wcol < W) {  // W is even and wcol a multiple of 4: either the whole float4 or its first half is inside
            float *o = dx + ((static_cast<size_t>(n) * 3 + c) * H + h) * W + wcol;
            const float4 v = *reinterpret_cast<const float4 *>(fr + (c * 16 + orow) * 64 + 4 * q4);
            if (wcol + 3 < W) {
                *reinterpret_cast<float2 *>(o) = make_float2(v.x, v.y);
                *reinterpret_cast<float2 *>(o + 2) = make_float2(v.z, v.w);
            } else {
                *reinterpret_cast<float2 *>(o) = make_float2(v.x, v.y);
            }
        }
    }
}

}  // namespace

EE_API int ee_stem7x7s2_bwd_data_f32(const float *dy, const float *weight, float *dx, int B, int K, int H, int W, void *stream) {
    if (B < 0 || K < 1 || H < 2 || W < 2) return EE_ERR_SHAPE;
    if ((H & 1) || (W & 1)) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!dy || !weight || !dx) return EE_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(dx) & 7u) return EE_ERR_ALIGN;
    const int OH = H / 2, OW = W / 2;
    const int tiles_a = (OH + ST_TA - 1) / ST_TA, tiles_b = (OW + ST_TB - 1) / ST_TB;
    const int64_t grid = static_cast<int64_t>(B) * tiles_a * tiles_b;
    if (grid > 0x7fffffffLL) return EE_ERR_SHAPE;
    EE_LAUNCH(stem_bwd_data_mfma_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, as_stream(stream), dy, weight, dx, K, OH, OW, tiles_a, tiles_b);
    return launch_status();
}

// =====================================================================================================================
// Forward of the stem convolution Conv2d(3, 64k, kernel_size=7, stride=2, padding=3, bias=False) (resnet.py:112-113), round 2.
// MIOpen's best solver (Winograd f3x2_stride2) takes 52 us un-profiled for [100,3,64,64] -> [100,64,32,32], the most expensive
// single launch of a PGD iteration's forward pass.  Implicit GEMM on v_mfma_f32_32x32x2_f32:
//   D[co][p] = sum_{(ci, ky, kx)} W[co][ci][ky][kx] * x[n, ci, 2 oy + ky - 3, 2 ox + kx - 3],     K = 3 * 7 * 8 (kx padded to 8, W = 0 there)
// One MFMA step takes the taps (kx, kx + 1) of one (ci, ky) on the two lane halves: the B operand of lane (i, half) is the frame
// element at column 2 i + kx + half - even and odd columns on disjoint LDS banks, one ds_read_b32 with an immediate offset per
// MFMA and tile, no index arithmetic in the loop.  A workgroup (4 wavefronts = 2 channel halves x 2 row pairs) owns 64 output
// channels x 4 output rows x 32 columns; its 13 x 72 x 3 input frame sits in LDS (zero outside the image) and the 84 A operands
// of a lane stay in REGISTERS for both of its tiles (weights pass through LDS once, in their natural layout, 16-B global loads).
// =====================================================================================================================
namespace {

constexpr int SF_ROWS = 4, SF_FH = 2 * SF_ROWS + 5, SF_FW = 72, SF_K = 147, SF_WS = 149, SF_STEPS = 3 * 7 * 4;

__global__ __launch_bounds__(256, 3) void stem_fwd_mfma_kernel(const float *__restrict__ x, const float *__restrict__ w, float *__restrict__ y,
                                                               float *__restrict__ stats, int K, int H, int W, int tiles_r, int tiles_c) {
    __shared__ float fr[3 * SF_FH * SF_FW];
    __shared__ __align__(16) float wn[64 * SF_WS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, kk = lane >> 5;
    const int mt = wave >> 1, nt = wave & 1;
    const int OH = H / 2, OW = W / 2;
    int t = static_cast<int>(blockIdx.x);
    const int tc = t % tiles_c;
    t /= tiles_c;
    const int tr = t % tiles_r, n = t / tiles_r;
    const int oy0 = tr * SF_ROWS, ox0 = tc * 32;
    const int cb = static_cast<int>(blockIdx.y) * 64;
    // ---- all global loads first (clamped addresses, masked afterwards) ----------------------------------------------------------
    constexpr int WV4 = 64 * SF_K / 4, WPT = (WV4 + 255) / 256, FTOT = 3 * SF_FH * SF_FW, FPT = (FTOT + 255) / 256;
    float4 wv[WPT];
    const float4 *wsrc = reinterpret_cast<const float4 *>(w + static_cast<size_t>(cb) * SF_K);
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const int e = threadIdx.x + j * 256;
        wv[j] = wsrc[e < WV4 ? e : WV4 - 1];
    }
    float fv[FPT];
    bool fok[FPT];
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
#pragma unroll
    for (int j = 0; j < FPT; ++j) {
        const int idx = threadIdx.x + j * 256;
        const int c = idx % SF_FW, q = idx / SF_FW;
        const int r = q % SF_FH, ci = idx < FTOT ? q / SF_FH : 0;
        const int iy = iy0 + r, ix = ix0 + c;
        fok[j] = idx < FTOT && iy >= 0 && iy < H && ix >= 0 && ix < W;
        fv[j] = x[((static_cast<size_t>(n) * 3 + ci) * H + (fok[j] ? iy : 0)) * W + (fok[j] ? ix : 0)];
    }
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const int e = threadIdx.x + j * 256;
        if (e < WV4) {
            const float v4[4] = {wv[j].x, wv[j].y, wv[j].z, wv[j].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = 4 * e + u, co = f / SF_K, k = f - co * SF_K;
                wn[co * SF_WS + k] = v4[u];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < FPT; ++j)
        if (static_cast<int>(threadIdx.x) + j * 256 < FTOT) fr[threadIdx.x + j * 256] = fok[j] ? fv[j] : 0.0f;
    __syncthreads();
    // ---- this lane's 84 A operands: W[cb + mt*32 + i][ci][ky][2 kxp + kk], zero for the padded tap ---------------------------------
    float a[SF_STEPS];
    const float *wrow = wn + (mt * 32 + i) * SF_WS + kk;
#pragma unroll
    for (int s = 0; s < SF_STEPS; ++s) {
        const int kxp = s & 3, cy = s >> 2;  // cy = ci * 7 + ky
        const float v = wrow[cy * 7 + 2 * kxp];  // for (kxp 3, kk 1) this is the next row's first tap (or the pad column): finite, dropped
        a[s] = (kxp == 3 && kk) ? 0.0f : v;
    }
#pragma unroll
    for (int s = 0; s < SF_STEPS; ++s) asm volatile("" : "+v"(a[s]));  // keep them in registers: the compiler would re-read LDS per MFMA
    // ---- two tiles (output rows 2 nt, 2 nt + 1 of the workgroup's four), 84 MFMAs each, interleaved ---------------------------------
    f32x16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
    const float *b0 = fr + (2 * (2 * nt)) * SF_FW + 2 * i + kk;  // frame row 2 r + ky, column 2 i + 2 kxp + kk
    const float *b1 = b0 + 2 * SF_FW;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kxp = 0; kxp < 4; ++kxp) {
                const int off = (ci * SF_FH + ky) * SF_FW + 2 * kxp;
                const float av = a[(ci * 7 + ky) * 4 + kxp];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[off], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1[off], acc1, 0, 0, 0);
            }
    // ---- store: lane (i, kk) holds column ox0 + i of rows acc_row(r, lane) ------------------------------------------------------------
    const size_t plane = static_cast<size_t>(OH) * OW;
    const int oy = oy0 + 2 * nt;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = cb + mt * 32 + acc_row(r, lane);
        float *o = y + (static_cast<size_t>(n) * K + co) * plane + static_cast<size_t>(oy) * OW + ox0 + i;
        if (oy < OH) o[0] = acc0[r];
        if (oy + 1 < OH) o[OW] = acc1[r];
    }
    // ---- optional: this workgroup's share of the BatchNorm statistics that follow the stem (resnet.py:113) - per output channel
    // (sum, M2 about the tile's own mean, count) of its <= 4 x 32 values, so that bn1 does not have to read y twice for them:
    // stats[(co * S + workgroup) * 3 + {0,1,2}], S = gridDim.x; merged by ee_bn.hip (bn_stats_finalize_kernel, Chan's update) -------
    if (!stats) return;
    // Through LDS, not through lane shuffles (a butterfly over 32 lanes for 2 x 16 registers is 160 dependent ds_bpermute: +13 us):
    // the tile goes to LDS as [64 channels][4 rows x 32 columns] (channel stride 132: conflict-free), then four lanes per channel
    // take 32 values each (two passes on registers: sum, then M2 about the tile mean) and meet through two quad exchanges.
    constexpr int CS = 132;
    __syncthreads();  // the weight rows in LDS are dead (every lane holds its A operands in registers): reuse them
    float *ex = wn;
    static_assert(64 * CS <= 64 * SF_WS, "the tile fits where the weights were");
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float *e = ex + (mt * 32 + acc_row(r, lane)) * CS + (2 * nt) * 32 + i;
        e[0] = acc0[r];
        e[32] = acc1[r];
    }
    __syncthreads();
    const int c = threadIdx.x >> 2, q = threadIdx.x & 3;
    const int rows = OH - oy0 < SF_ROWS ? OH - oy0 : SF_ROWS;  // valid output rows of this tile (>= 1)
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = ex[c * CS + q + 4 * j];  // element q + 4j: row j / 8
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 32; ++j) sum += (j / 8 < rows) ? v[j] : 0.0f;
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    const float cnt = 32.0f * static_cast<float>(rows), mean = sum / cnt;
    float m2 = 0.0f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const float d = v[j] - mean;
        m2 += (j / 8 < rows) ? d * d : 0.0f;
    }
    m2 += __shfl_xor(m2, 1);
    m2 += __shfl_xor(m2, 2);
    if (q == 0) {
        float *o = stats + (static_cast<size_t>(cb + c) * gridDim.x + blockIdx.x) * 3;
        o[0] = sum;
        o[1] = m2;
        o[2] = cnt;
    }
}

}  // namespace

EE_API int ee_stem7x7s2_fwd_stats_floats(int B, int K, int H, int W) {
    if (B < 1 || K < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || (W / 2) % 32 != 0 || K % 64 != 0) return 0;
    const int64_t grid = static_cast<int64_t>(B) * ((H / 2 + SF_ROWS - 1) / SF_ROWS) * (W / 2 / 32);
    if (grid * K * 3 > 0x7fffffffLL) return 0;
    return static_cast<int>(grid * K * 3);
}

EE_API int ee_stem7x7s2_fwd_f32(const float *x, const float *weight, float *y, int B, int K, int H, int W, void *stream) {
    return ee_stem7x7s2_fwd_stats_f32(x, weight, y, nullptr, B, K, H, W, stream);
}

EE_API int ee_stem7x7s2_fwd_stats_f32(const float *x, const float *weight, float *y, float *stats, int B, int K, int H, int W, void *stream) {
    if (B < 0 || K < 1 || H < 2 || W < 2) return EE_ERR_SHAPE;
    if ((H & 1) || (W & 1) || (W / 2) % 32 != 0 || K % 64 != 0) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!x || !weight || !y) return EE_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(weight) & 15u) return EE_ERR_ALIGN;
    const int OH = H / 2, OW = W / 2;
    const int tiles_r = (OH + SF_ROWS - 1) / SF_ROWS, tiles_c = OW / 32;
    const int64_t grid = static_cast<int64_t>(B) * tiles_r * tiles_c;
    if (grid > 0x7fffffffLL || static_cast<int64_t>(B) * K * OH * OW > 0x7fffffffLL * 4LL) return EE_ERR_SHAPE;
    EE_LAUNCH(stem_fwd_mfma_kernel, dim3(static_cast<unsigned>(grid), static_cast<unsigned>(K / 64)), dim3(256), 0, as_stream(stream), x, weight, y,
              stats, K, H, W, tiles_r, tiles_c);
    return launch_status();
}

// =====================================================================================================================
// Conv2d(3x3, stride 1, padding 1, bias=False) of the residual blocks (Tiny_ImageNet/models_tinyimagenet/resnet.py:26-31),
// forward and backward-data, as an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact f32).
//
// Why: at the reference batch (100 x 64x64 images) the 64-channel 16x16 layers are 1.9 GFLOP each; MIOpen's best fp32
// solver (Winograd F(2,3)) needs 36 us for them either way, i.e. ~30 % of the f32 matrix rate, and they are 8 of the
// ~40 convolution launches of a PGD iteration.
//
//   D[rc][p] = sum_{kc, tap} A[rc][(kc, tap)] * Bm[(kc, tap)][p]
//   forward : rc = output channel, kc = input channel,  A = W[rc][kc][tap],      Bm = x [n, kc, h+kh-1, w+kw-1]
//   backward: rc = input channel,  kc = output channel, A = W[kc][rc][8 - tap],  Bm = dy[n, kc, h+kh-1, w+kw-1]
//
// A workgroup (4 wavefronts) owns 64 result channels x 64 pixels = (64 / W) whole rows of the (image, row) sequence; per
// round 16 reduction channels: their weights in LDS as [kc][tap][65] (result channel fastest: conflict-free operand reads)
// and a zero-bordered frame [kc][rows + 2][W + 2].  One MFMA step takes the two channels 2c, 2c+1 (lane halves) at one tap.
// The next round's operands are requested from global memory before this round's MFMAs (registers), so one L2 round trip
// overlaps one round of multiplies.  Measured (rocprofv3, B=100): 64ch 16x16 31 us forward / 34 us backward (MIOpen 36);
// 128ch 8x8 37 / 40 us (MIOpen 30-36: not used there).  An 8-channel round was latency-bound at every barrier (37 / 46 us);
// a prefetch distance of two rounds needs 299 VGPRs and loses the second workgroup per CU (36-43 us).
// =====================================================================================================================
namespace {

constexpr int C3_CK = 16, C3_WS = 66;  // 66: the four weight-staging sub-roles of a wavefront land on disjoint bank groups

struct Conv3Dims {
    int B, KC, RC, H, W;  // reduction channels, result channels
    int dbg;              // EEADV_CONV3_DBG (measurement only, grouped kernel): 1 no weight loads, 2 no frame loads, 4 no MFMAs, 8 no LDS staging
};

template <bool BWD, int TW>
__global__ __launch_bounds__(256) void conv3x3s1_kernel(const float *__restrict__ in, const float *__restrict__ w, float *__restrict__ out,
                                                        Conv3Dims d) {
    constexpr int PR = 64 / TW;                   // rows of the tile
    constexpr int FRW = TW + 2, FRH = PR + 2;     // frame
    constexpr int WTOT = C3_CK * 9 * 64, FTOT = C3_CK * FRH * FRW;
    constexpr int WPT = WTOT / 256, FPT = (FTOT + 255) / 256;  // elements staged per thread and round
    static_assert(WTOT % 256 == 0, "weight tile must divide over the workgroup");
    __shared__ float ws[C3_CK * 9 * C3_WS];       // [kc][tap][result channel (+1 pad)]
    __shared__ float fr[FTOT];                    // [kc][frame row][frame col]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int rt = wave >> 1, pt = wave & 1;      // result-channel tile and pixel tile of this wavefront
    const int rc_base = static_cast<int>(blockIdx.y) * 64;
    const int g0 = static_cast<int>(blockIdx.x) * PR;  // first (image, row) of the tile
    const int rows_total = d.B * d.H;
    // this lane's pixel (B-operand column)
    const int pj = pt * 32 + i;
    const int rj = pj / TW, wj = pj - rj * TW;
    const int gj = g0 + rj;
    const bool pv = gj < rows_total;
    const int nj = pv ? gj / d.H : 0, hj = pv ? gj - nj * d.H : 0;
    const bool top = hj == 0, bot = hj == d.H - 1;
    const float *bbase = fr + kk * (FRH * FRW) + rj * FRW + wj;
    const float *abase = ws + kk * (9 * C3_WS) + rt * 32 + i;
    const size_t plane = static_cast<size_t>(d.H) * TW;

    // ---- staging roles, fixed per thread: 32-bit element offsets relative to the round's first reduction channel ---------
    // weights: global reads run along (kc, tap) (forward) or (rc, tap) (backward), both contiguous.  KC % 16 == 0 and
    // RC % 64 == 0 are guaranteed by the launcher, so no load needs a predicate (a predicated load is a branch: it would
    // serialise the round trips) and the per-round address update is one add of a wave-uniform stride.
    // weights: each thread owns 36 CONTIGUOUS floats of the round (9 x 16-B loads, no index arithmetic per element):
    //   forward : result channel rl = tid / 4, reduction channels 4*(tid % 4) .. +3, all 9 taps   (W[rc][kc][tap] is contiguous in kc, tap)
    //   backward: reduction channel kc = tid / 16, result channels 4*(tid % 16) .. +3, all 9 taps (W[kc][rc][tap] is contiguous in rc, tap)
    static_assert(WPT == 36 && C3_CK == 16, "the weight roles below assume 16-channel rounds");
    const int wq = BWD ? (threadIdx.x >> 4) : (threadIdx.x >> 2), wp = BWD ? (threadIdx.x & 15) : (threadIdx.x & 3);
    const unsigned wsrc0 = static_cast<unsigned>(BWD ? (wq * d.RC + rc_base + 4 * wp) * 9 : ((rc_base + wq) * d.KC + 4 * wp) * 9);
    unsigned fsrc[FPT];
    bool fok[FPT];
#pragma unroll
    for (int j = 0; j < FPT; ++j) {
        const int idx = threadIdx.x + j * 256;
        const int fc = idx % FRW, tq = idx / FRW;
        const int frow = tq % FRH, kc = idx < FTOT ? tq / FRH : 0;
        const int g = g0 - 1 + frow, c = fc - 1;
        fok[j] = idx < FTOT && g >= 0 && g < rows_total && c >= 0 && c < TW;
        const int gc = fok[j] ? g : 0, cc = fok[j] ? c : 0;
        const int n = gc / d.H, h = gc - n * d.H;
        fsrc[j] = static_cast<unsigned>((n * d.KC + kc) * static_cast<int>(plane) + h * TW + cc);
    }
    const unsigned wstep = static_cast<unsigned>(BWD ? C3_CK * d.RC * 9 : C3_CK * 9), fstep = static_cast<unsigned>(C3_CK * plane);
    float4 wv4[WPT / 4];
    float fv[FPT];
    auto prefetch = [&](unsigned round) {
        const unsigned wo = round * wstep, fo = round * fstep;
#pragma unroll
        for (int j = 0; j < WPT / 4; ++j) wv4[j] = *reinterpret_cast<const float4 *>(w + wsrc0 + wo + 4 * j);
#pragma unroll
        for (int j = 0; j < FPT; ++j) fv[j] = in[fsrc[j] + fo];
    };
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned rounds = static_cast<unsigned>(d.KC / C3_CK);
    prefetch(0);
    for (unsigned round = 0; round < rounds; ++round) {
        __syncthreads();  // the previous round's MFMAs have read their operands
#pragma unroll
        for (int j = 0; j < WPT / 4; ++j) {
            const float v4[4] = {wv4[j].x, wv4[j].y, wv4[j].z, wv4[j].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = 4 * j + u;  // element e of the thread's 36: (sub-channel e / 9, tap e % 9), compile-time
                if (!BWD) ws[((wp * 4 + e / 9) * 9 + e % 9) * C3_WS + wq] = v4[u];
                else ws[(wq * 9 + 8 - e % 9) * C3_WS + 4 * wp + e / 9] = v4[u];
            }
        }
#pragma unroll
        for (int j = 0; j < FPT; ++j)
            if (static_cast<int>(threadIdx.x) + j * 256 < FTOT) fr[threadIdx.x + j * 256] = fok[j] ? fv[j] : 0.0f;
        __syncthreads();
        if (round + 1 < rounds) prefetch(round + 1);  // next round's operands travel while this round multiplies
        // operands of half a round into registers first (LDS reads in flight together), then the MFMAs back to back
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float av[C3_CK / 4 * 9], bv[C3_CK / 4 * 9];
#pragma unroll
            for (int c2 = 0; c2 < C3_CK / 4; ++c2) {
                const float *ap = abase + (half * (C3_CK / 4) + c2) * (2 * 9 * C3_WS);
                const float *bp = bbase + (half * (C3_CK / 4) + c2) * (2 * FRH * FRW);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        av[c2 * 9 + kh * 3 + kw] = ap[(kh * 3 + kw) * C3_WS];
                        float b = bp[kh * FRW + kw];
                        if (kh == 0 && top) b = 0.0f;  // the frame row above belongs to the previous image
                        if (kh == 2 && bot) b = 0.0f;
                        bv[c2 * 9 + kh * 3 + kw] = b;
                    }
            }
            // (two accumulator chains change nothing: the stalls counted by SQ_WAIT_INST_ANY are the MFMA issue cadence itself)
#pragma unroll
            for (int q = 0; q < C3_CK / 4 * 9; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
        }
    }
    if (!pv) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rc = rc_base + rt * 32 + acc_row(r, lane);
        out[(static_cast<size_t>(nj) * d.RC + rc) * plane + static_cast<size_t>(hj) * TW + wj] = acc[r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Software-pipelined version (round 2).  Counters on the kernel above (rocprofv3 --pmc, 128ch 8x8): matrix pipe 28 % busy, LDS
// 14 %, 5 % of wave cycles waiting on LDS - nothing is saturated; each round is a serial chain [barrier, stage, barrier, 72
// operand reads, wait, 36 multiplies, 72 reads, wait, 36 multiplies] and phase-skipping (EEADV_CONV3_DBG) shows the multiply
// time (15 us) ADDED to everything else (16 us) instead of hiding it; a second wavefront per SIMD does not change that (it
// runs the same phases at the same time).  Here one wavefront overlaps its own phases:
//   * operands come out of LDS through a ring of 4 register pairs, requested 3 multiplies ahead of their use: the ds_reads sit
//     between the (dependent, 64-cycle) MFMAs in program order and issue while the previous MFMA executes;
//   * LDS is double-buffered: the next round's weights / frame (prefetched from global memory one round earlier) are written
//     into the other buffer one element per multiply, so a round needs ONE barrier, at its end;
//   * the global prefetch for the round after next is issued as soon as the staging registers are free.
// Same tiles, same operand order (bit-identical sums) as the kernel above.  RT = 2, KG = 1: 64 result channels x 64 pixels,
// 4 wavefronts; RT = 1, KG = 2: 32 x 64 with the reduction split over two 2-wavefront groups (twice the workgroups on the
// 4x4 layers), partial tiles added in group order through LDS.
// ---------------------------------------------------------------------------------------------------------------------
// A barrier among the wavefronts of ONE reduction group (gfx950 has no named barriers): an LDS counter that only grows; every
// wavefront adds 1 and sleeps until the count reaches `target` (generation x wavefronts per group).  All wavefronts of a
// workgroup are resident together, so the wait cannot deadlock.
__device__ __forceinline__ void group_barrier(unsigned *ctr, unsigned target) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <bool BWD, int TW, int RT, int KG>
__global__ __launch_bounds__(256) void conv3x3s1_pipe_kernel(const float *__restrict__ in, const float *__restrict__ w, float *__restrict__ out,
                                                             Conv3Dims d) {
    constexpr int PR = 64 / TW, FRW = TW + 2, FRH = PR + 2;
    constexpr int RCW = 32 * RT, WSS = RCW + 2, GT = 128 * RT;
    constexpr int WTOT = C3_CK * 9 * RCW, FTOT = C3_CK * FRH * FRW;
    constexpr int WPT = WTOT / GT, FPT = (FTOT + GT - 1) / GT;
    constexpr int BUF = (C3_CK * 9 * WSS + FTOT + 3) & ~3;  // floats of one (weights, frame) buffer
    constexpr int NQ = C3_CK / 2 * 9;                        // multiplies per round and wavefront (72)
    constexpr int LOOK = 3, RING = 4;  // 3 x (2 reads + 1 write) = 9 LDS operations in flight: lgkmcnt counts to 15
    static_assert(WPT == 36 && 2 * RT * KG == 4 && WPT + FPT <= NQ - 8, "4 wavefronts; staging fits inside the multiply stream");
    extern __shared__ __align__(16) float lds[];
    __shared__ unsigned gctr[KG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kg = wave / (2 * RT), tw = wave - kg * (2 * RT), rt = tw >> 1, pt = tw & 1;
    const int gtid = static_cast<int>(threadIdx.x) - kg * GT;
    float *gbuf = lds + kg * (2 * BUF);
    if (KG > 1) {
        if (threadIdx.x < KG) gctr[threadIdx.x] = 0u;
        __syncthreads();
    }
    unsigned gen = 0;
    auto gsync = [&]() {
        if (KG == 1) __syncthreads();
        else group_barrier(&gctr[kg], ++gen * (2 * RT));
    };
    const int i = lane & 31, kk = lane >> 5;
    const int rc_base = static_cast<int>(blockIdx.y) * RCW;
    const int g0 = static_cast<int>(blockIdx.x) * PR;
    const int rows_total = d.B * d.H;
    const int pj = pt * 32 + i;
    const int rj = pj / TW, wj = pj - rj * TW;
    const int gj = g0 + rj;
    const bool pv = gj < rows_total;
    const int nj = pv ? gj / d.H : 0, hj = pv ? gj - nj * d.H : 0;
    const bool top = hj == 0, bot = hj == d.H - 1;
    const int boff = C3_CK * 9 * WSS + kk * (FRH * FRW) + rj * FRW + wj;  // this lane's frame operand, relative to a buffer
    const int aoff = kk * (9 * WSS) + rt * 32 + i;
    const size_t plane = static_cast<size_t>(d.H) * TW;
    const int wq = BWD ? gtid / (8 * RT) : (gtid >> 2), wp = BWD ? gtid % (8 * RT) : (gtid & 3);
    const unsigned wsrc0 = static_cast<unsigned>(BWD ? (wq * d.RC + rc_base + 4 * wp) * 9 : ((rc_base + wq) * d.KC + 4 * wp) * 9);
    unsigned fsrc[FPT];
    bool fok[FPT];
#pragma unroll
    for (int j = 0; j < FPT; ++j) {
        const int idx = gtid + j * GT;
        const int fc = idx % FRW, tq = idx / FRW;
        const int frow = tq % FRH, kc = idx < FTOT ? tq / FRH : 0;
        const int g = g0 - 1 + frow, c = fc - 1;
        fok[j] = idx < FTOT && g >= 0 && g < rows_total && c >= 0 && c < TW;
        const int gc = fok[j] ? g : 0, cc = fok[j] ? c : 0;
        const int n = gc / d.H, h = gc - n * d.H;
        fsrc[j] = static_cast<unsigned>((n * d.KC + kc) * static_cast<int>(plane) + h * TW + cc);
    }
    const unsigned wstep = static_cast<unsigned>(BWD ? C3_CK * d.RC * 9 : C3_CK * 9), fstep = static_cast<unsigned>(C3_CK * plane);
    float wv[WPT], fv[FPT];
    auto prefetch = [&](unsigned round) {
        const unsigned wo = round * wstep, fo = round * fstep;
#pragma unroll
        for (int j = 0; j < WPT / 4; ++j) {
            const float4 t = *reinterpret_cast<const float4 *>(w + wsrc0 + wo + 4 * j);
            wv[4 * j] = t.x; wv[4 * j + 1] = t.y; wv[4 * j + 2] = t.z; wv[4 * j + 3] = t.w;
        }
#pragma unroll
        for (int j = 0; j < FPT; ++j) fv[j] = in[fsrc[j] + fo];
    };
    // staging element e of this thread (e < WPT: weight, else frame) into buffer `b`
    auto stage = [&](float *b, int e) {
        if (e < WPT) {
            if (!BWD) b[((wp * 4 + e / 9) * 9 + e % 9) * WSS + wq] = wv[e];
            else b[(wq * 9 + 8 - e % 9) * WSS + 4 * wp + e / 9] = wv[e];
        } else {
            const int j = e - WPT;
            if (gtid + j * GT < FTOT) b[C3_CK * 9 * WSS + gtid + j * GT] = fok[j] ? fv[j] : 0.0f;
        }
    };
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned iters = static_cast<unsigned>(d.KC / C3_CK / KG);
    prefetch(kg);
#pragma unroll
    for (int e = 0; e < WPT + FPT; ++e) stage(gbuf, e);
    if (iters > 1) prefetch(kg + KG);
    gsync();
    for (unsigned it = 0; it < iters; ++it) {
        const float *cur = gbuf + (it & 1) * BUF;
        float *nxt = gbuf + ((it + 1) & 1) * BUF;
        const bool more = it + 1 < iters, more2 = it + 2 < iters;
        float ra[RING], rb[RING];
        auto fetch = [&](int q) {  // operands of multiply q: channels 2 (q / 9) + kk, tap q % 9
            const int c2 = q / 9, tap = q % 9, kh = tap / 3, kw = tap % 3;
            ra[q % RING] = cur[aoff + c2 * (2 * 9 * WSS) + tap * WSS];
            float b = cur[boff + c2 * (2 * FRH * FRW) + kh * FRW + kw];
            if (kh == 0 && top) b = 0.0f;  // the frame row above belongs to the previous image
            if (kh == 2 && bot) b = 0.0f;
            rb[q % RING] = b;
        };
#pragma unroll
        for (int q = 0; q < LOOK; ++q) fetch(q);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + LOOK < NQ) fetch(q + LOOK);
            if (more && q < WPT + FPT) stage(nxt, q);
            if (more2 && q == WPT + FPT) prefetch(kg + (it + 2) * KG);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[q % RING], rb[q % RING], acc, 0, 0, 0);
        }
        gsync();  // everybody has read `cur` and written `nxt`
    }
    if (KG > 1) {  // partial tiles meet in LDS, group 0 adds them in group order
        __syncthreads();
        float *red = lds;  // [KG - 1][2 RT][16][64]
        if (kg > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((kg - 1) * (2 * RT) + tw) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (kg != 0) return;
#pragma unroll
        for (int g = 1; g < KG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += red[(((g - 1) * (2 * RT) + tw) * 16 + r) * 64 + lane];
    }
    if (!pv) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rc = rc_base + rt * 32 + acc_row(r, lane);
        out[(static_cast<size_t>(nj) * d.RC + rc) * plane + static_cast<size_t>(hj) * TW + wj] = acc[r];
    }
}

template <bool BWD, int TW, int RT, int KG>
void conv3_pipe_launch(const float *in, const float *w, float *out, const Conv3Dims &dims, int64_t gx, hipStream_t st) {
    constexpr int PR = 64 / TW, FRW = TW + 2, FRH = PR + 2, WSS = 32 * RT + 2;
    constexpr int BUF = (C3_CK * 9 * WSS + C3_CK * FRH * FRW + 3) & ~3;
    constexpr size_t bytes = sizeof(float) * static_cast<size_t>(KG) * 2 * BUF;
    static bool opted = false;
    if (!opted && bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3s1_pipe_kernel<BWD, TW, RT, KG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(bytes)) != hipSuccess)
            (void)hipGetLastError();
        opted = true;
    }
    const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(dims.RC / (32 * RT))), block(256);
    EE_LAUNCH((conv3x3s1_pipe_kernel<BWD, TW, RT, KG>), grid, block, bytes, st, in, w, out, dims);
}

// EEADV_CONV3_KG: 0 forces the 4-wavefront kernel, 10 / 11 the pipelined kernel (64 x 64 tiles / 32 x 64 tiles with a 2-way
// reduction split) wherever its tiling exists - A/B measurements; unset: the measured choice below
int conv3_variant() {
    static const int v = [] {
        const char *e = getenv("EEADV_CONV3_KG");
        return e ? atoi(e) : -1;
    }();
    return v;
}

template <bool BWD>
int conv3_launch(const float *in, const float *w, float *out, int B, int KC, int RC, int H, int W, hipStream_t st) {
    if (B < 0 || KC < 1 || RC < 1 || H < 1 || W < 1) return EE_ERR_SHAPE;
    if (W > 64 || 64 % W != 0 || KC % C3_CK != 0 || RC % 64 != 0) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!in || !w || !out) return EE_ERR_NULL;
    const int PR = 64 / W;
    const int64_t rows = static_cast<int64_t>(B) * H;
    const int64_t gx = (rows + PR - 1) / PR;
    if (gx > 0x7fffffffLL) return EE_ERR_SHAPE;
    if (static_cast<int64_t>(B) * KC * H * W > 0x7fffffffLL || static_cast<int64_t>(KC) * RC * 9 > 0x7fffffffLL) return EE_ERR_SHAPE;  // 32-bit offsets
    const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(RC / 64)), block(256);
    static const int dbg = [] {
        const char *e = getenv("EEADV_CONV3_DBG");
        return e ? atoi(e) : 0;
    }();
    const Conv3Dims dims{B, KC, RC, H, W, dbg};
    const int rounds = KC / C3_CK, var = conv3_variant();
    const int64_t wgs64 = gx * (RC / 64);
    // Measured, un-profiled, B = 100 (fwd us: 4-wavefront kernel / pipelined 64x64 / pipelined 32x64 split / MIOpen Winograd):
    //   64ch 16x16  28.5 / 35 / -  / 32.3      128ch 8x8  31.9 / 31.6 / - / 29.4      256ch 4x4  57.8 / 59 / 34 / 37.5
    // so the pipelined kernel is the default only where the 64-channel tiling leaves most of the chip idle (<= 128 workgroups).
    int pipe = 0;
    if (var == 10 && W >= 4) pipe = 10;
    else if (var == 11 && W >= 4 && W <= 16) pipe = 11;
    else if (var == -1 && W == 4 && wgs64 <= 128) pipe = 11;
    if (pipe == 11 && rounds % 2 != 0) pipe = 0;
    const double flops = 2.0 * 9.0 * KC * RC * static_cast<double>(B) * H * W;
    ProfScope prof(pipe ? (BWD ? EE_K_CONV3P_BWD : EE_K_CONV3P_FWD) : (BWD ? EE_K_CONV3_BWD : EE_K_CONV3_FWD), st, flops);
    if (pipe == 10) {
        switch (W) {
            case 64: conv3_pipe_launch<BWD, 64, 2, 1>(in, w, out, dims, gx, st); break;
            case 32: conv3_pipe_launch<BWD, 32, 2, 1>(in, w, out, dims, gx, st); break;
            case 16: conv3_pipe_launch<BWD, 16, 2, 1>(in, w, out, dims, gx, st); break;
            case 8: conv3_pipe_launch<BWD, 8, 2, 1>(in, w, out, dims, gx, st); break;
            default: conv3_pipe_launch<BWD, 4, 2, 1>(in, w, out, dims, gx, st); break;
        }
        return launch_status();
    }
    if (pipe == 11) {
        switch (W) {
            case 16: conv3_pipe_launch<BWD, 16, 1, 2>(in, w, out, dims, gx, st); break;
            case 8: conv3_pipe_launch<BWD, 8, 1, 2>(in, w, out, dims, gx, st); break;
            default: conv3_pipe_launch<BWD, 4, 1, 2>(in, w, out, dims, gx, st); break;
        }
        return launch_status();
    }
    switch (W) {
        case 64: EE_LAUNCH((conv3x3s1_kernel<BWD, 64>), grid, block, 0, st, in, w, out, dims); break;
        case 32: EE_LAUNCH((conv3x3s1_kernel<BWD, 32>), grid, block, 0, st, in, w, out, dims); break;
        case 16: EE_LAUNCH((conv3x3s1_kernel<BWD, 16>), grid, block, 0, st, in, w, out, dims); break;
        case 8: EE_LAUNCH((conv3x3s1_kernel<BWD, 8>), grid, block, 0, st, in, w, out, dims); break;
        case 4: EE_LAUNCH((conv3x3s1_kernel<BWD, 4>), grid, block, 0, st, in, w, out, dims); break;
        case 2: EE_LAUNCH((conv3x3s1_kernel<BWD, 2>), grid, block, 0, st, in, w, out, dims); break;
        default: EE_LAUNCH((conv3x3s1_kernel<BWD, 1>), grid, block, 0, st, in, w, out, dims); break;
    }
    return launch_status();
}

}  // namespace

EE_API int ee_conv3x3s1_fwd_f32(const float *x, const float *weight, float *y, int B, int Cin, int Cout, int H, int W, void *stream) {
    return conv3_launch<false>(x, weight, y, B, Cin, Cout, H, W, as_stream(stream));
}

EE_API int ee_conv3x3s1_bwd_data_f32(const float *dy, const float *weight, float *dx, int B, int Cin, int Cout, int H, int W, void *stream) {
    return conv3_launch<true>(dy, weight, dx, B, Cout, Cin, H, W, as_stream(stream));
}

// =====================================================================================================================
// Conv2d(3x3, stride 2, padding 1, bias=False) - the first convolution of layers 2-4 (resnet.py:26-31 with stride 2),
// forward, on the same f32-MFMA implicit GEMM.  0.94 GFLOP at the reference batch, for which MIOpen's solvers need ~50 us
// (Winograd-stride2) or an NHWC implicit GEMM wrapped in three layout transposes and a zero fill.
//   y[n,co,oh,ow] = sum_{ci,kh,kw} W[co][ci][kh][kw] * x[n, ci, 2oh-1+kh, 2ow-1+kw]
// H even, so input row (n, 2oh-1+kh) is row 2g-1+kh of the flattened (image, row) sequence for output row g = n*OH + oh: a
// tile of 64 / OW consecutive output rows reads 2*PR + 1 consecutive input rows, whatever images it spans; only the row above
// an image's first row has to be masked (the column left of the image is a zero column of the frame).
// =====================================================================================================================
namespace {

template <int TOW>
__global__ __launch_bounds__(256) void conv3x3s2_fwd_kernel(const float *__restrict__ in, const float *__restrict__ w, float *__restrict__ out,
                                                            Conv3Dims d) {  // d.H, d.W: INPUT size; KC = Cin, RC = Cout
    constexpr int PR = 64 / TOW;                       // output rows of the tile
    constexpr int FRH = 2 * PR + 1, FRW = 2 * TOW + 2;  // frame: input rows 2g0-1 .. 2(g0+PR)-1, columns -1 .. W-1 (+1 pad)
    constexpr int TWI = 2 * TOW;                        // input width
    constexpr int WTOT = C3_CK * 9 * 64, FTOT = C3_CK * FRH * FRW;
    constexpr int WPT = WTOT / 256, FPT = (FTOT + 255) / 256;
    __shared__ float ws[C3_CK * 9 * C3_WS];
    __shared__ float fr[FTOT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int rt = wave >> 1, pt = wave & 1;
    const int rc_base = static_cast<int>(blockIdx.y) * 64;
    const int OH = d.H / 2;
    const int g0 = static_cast<int>(blockIdx.x) * PR;   // first output (image, row)
    const int orows_total = d.B * OH, irows_total = d.B * d.H;
    const int pj = pt * 32 + i;
    const int rj = pj / TOW, wj = pj - rj * TOW;
    const int gj = g0 + rj;
    const bool pv = gj < orows_total;
    const int nj = pv ? gj / OH : 0, ohj = pv ? gj - nj * OH : 0;
    const bool top = ohj == 0;
    const float *bbase = fr + kk * (FRH * FRW) + (2 * rj) * FRW + 2 * wj;
    const float *abase = ws + kk * (9 * C3_WS) + rt * 32 + i;
    const size_t iplane = static_cast<size_t>(d.H) * TWI, oplane = static_cast<size_t>(OH) * TOW;

    // weights: thread = (result channel tid / 4, reduction channels 4*(tid % 4) .. +3, all taps) = 36 contiguous floats, 9 x 16-B loads
    static_assert(WPT == 36 && C3_CK == 16, "the weight roles below assume 16-channel rounds");
    const int wq = threadIdx.x >> 2, wp = threadIdx.x & 3;
    const unsigned wsrc0 = static_cast<unsigned>(((rc_base + wq) * d.KC + 4 * wp) * 9);
    unsigned fsrc[FPT];
    bool fok[FPT];
#pragma unroll
    for (int j = 0; j < FPT; ++j) {
        const int idx = threadIdx.x + j * 256;
        const int fc = idx % FRW, tq = idx / FRW;
        const int frow = tq % FRH, kc = idx < FTOT ? tq / FRH : 0;
        const int g = 2 * g0 - 1 + frow, c = fc - 1;
        fok[j] = idx < FTOT && g >= 0 && g < irows_total && c >= 0 && c < TWI;
        const int gc = fok[j] ? g : 0, cc = fok[j] ? c : 0;
        const int n = gc / d.H, h = gc - n * d.H;
        fsrc[j] = static_cast<unsigned>((n * d.KC + kc) * static_cast<int>(iplane) + h * TWI + cc);
    }
    const unsigned wstep = static_cast<unsigned>(C3_CK * 9), fstep = static_cast<unsigned>(C3_CK * iplane);
    float4 wv4[WPT / 4];
    float fv[FPT];
    auto prefetch = [&](unsigned round) {
        const unsigned wo = round * wstep, fo = round * fstep;
#pragma unroll
        for (int j = 0; j < WPT / 4; ++j) wv4[j] = *reinterpret_cast<const float4 *>(w + wsrc0 + wo + 4 * j);
#pragma unroll
        for (int j = 0; j < FPT; ++j) fv[j] = in[fsrc[j] + fo];
    };
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned rounds = static_cast<unsigned>(d.KC / C3_CK);
    prefetch(0);
    for (unsigned round = 0; round < rounds; ++round) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WPT / 4; ++j) {
            const float v4[4] = {wv4[j].x, wv4[j].y, wv4[j].z, wv4[j].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = 4 * j + u;
                ws[((wp * 4 + e / 9) * 9 + e % 9) * C3_WS + wq] = v4[u];
            }
        }
#pragma unroll
        for (int j = 0; j < FPT; ++j)
            if (static_cast<int>(threadIdx.x) + j * 256 < FTOT) fr[threadIdx.x + j * 256] = fok[j] ? fv[j] : 0.0f;
        __syncthreads();
        if (round + 1 < rounds) prefetch(round + 1);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float av[C3_CK / 4 * 9], bv[C3_CK / 4 * 9];
#pragma unroll
            for (int c2 = 0; c2 < C3_CK / 4; ++c2) {
                const float *ap = abase + (half * (C3_CK / 4) + c2) * (2 * 9 * C3_WS);
                const float *bp = bbase + (half * (C3_CK / 4) + c2) * (2 * FRH * FRW);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        av[c2 * 9 + kh * 3 + kw] = ap[(kh * 3 + kw) * C3_WS];
                        float b = bp[kh * FRW + kw];
                        if (kh == 0 && top) b = 0.0f;  // the input row above an image's first row belongs to the previous image
                        bv[c2 * 9 + kh * 3 + kw] = b;
                    }
            }
#pragma unroll
            for (int q = 0; q < C3_CK / 4 * 9; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
        }
    }
    if (!pv) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rc = rc_base + rt * 32 + acc_row(r, lane);
        out[(static_cast<size_t>(nj) * d.RC + rc) * oplane + static_cast<size_t>(ohj) * TOW + wj] = acc[r];
    }
}

}  // namespace

EE_API int ee_conv3x3s2_fwd_f32(const float *x, const float *weight, float *y, int B, int Cin, int Cout, int H, int W, void *stream) {
    if (B < 0 || Cin < 1 || Cout < 1 || H < 2 || W < 2) return EE_ERR_SHAPE;
    const int OW = W / 2;
    if ((H & 1) || (W & 1) || OW > 64 || 64 % OW != 0 || Cin % C3_CK != 0 || Cout % 64 != 0) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!x || !weight || !y) return EE_ERR_NULL;
    if (static_cast<int64_t>(B) * Cin * H * W > 0x7fffffffLL || static_cast<int64_t>(Cin) * Cout * 9 > 0x7fffffffLL) return EE_ERR_SHAPE;
    const int PR = 64 / OW;
    const int64_t orows = static_cast<int64_t>(B) * (H / 2);
    const dim3 grid(static_cast<unsigned>((orows + PR - 1) / PR), static_cast<unsigned>(Cout / 64)), block(256);
    const Conv3Dims dims{B, Cin, Cout, H, W, 0};
    hipStream_t st = as_stream(stream);
    switch (OW) {
        case 64: EE_LAUNCH((conv3x3s2_fwd_kernel<64>), grid, block, 0, st, x, weight, y, dims); break;
        case 32: EE_LAUNCH((conv3x3s2_fwd_kernel<32>), grid, block, 0, st, x, weight, y, dims); break;
        case 16: EE_LAUNCH((conv3x3s2_fwd_kernel<16>), grid, block, 0, st, x, weight, y, dims); break;
        case 8: EE_LAUNCH((conv3x3s2_fwd_kernel<8>), grid, block, 0, st, x, weight, y, dims); break;
        case 4: EE_LAUNCH((conv3x3s2_fwd_kernel<4>), grid, block, 0, st, x, weight, y, dims); break;
        case 2: EE_LAUNCH((conv3x3s2_fwd_kernel<2>), grid, block, 0, st, x, weight, y, dims); break;
        default: EE_LAUNCH((conv3x3s2_fwd_kernel<1>), grid, block, 0, st, x, weight, y, dims); break;
    }
    return launch_status();
}

// =====================================================================================================================
// Backward-data of the stride-2 3x3 convolution, on the dy grid (MIOpen: Winograd-dilation 52 us, or an NHWC implicit GEMM
// plus three transposes and a zero fill, for 0.94 GFLOP).
//   dx[n,ci,2a+ph,2b+pw] = sum_{co,u,v} dy[n,co,a+u,b+v] * W[co,ci,kh(ph,u),kw(pw,v)],   (ph,u) -> kh: (0,0)->1, (1,0)->2, (1,1)->0
// i.e. every dy cell (a,b) produces the 2x2 input cell above it from the 2x2 window of dy cells at (a..a+1, b..b+1): a GEMM
// D[(ci,ph,pw)][cell] with K = (co,u,v), 9 of every 16 weight entries non-zero (the zero ones are written once).  Rows are
// ordered ci*4 + ph*2 + pw, so the four accumulator registers r..r+3 of a lane are the 2x2 cell of one input channel and
// leave as two 8-byte stores.  Workgroup: 16 input channels x 64 cells, 32 output channels per round.
// =====================================================================================================================
namespace {

constexpr int S2B_CK = 32;  // output channels (reduction) per round

template <int TOW>
__global__ __launch_bounds__(256) void conv3x3s2_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ w, float *__restrict__ dx,
                                                            Conv3Dims d) {  // d.H, d.W: size of dx; KC = Cout, RC = Cin
    constexpr int PR = 64 / TOW;
    constexpr int FRH = PR + 1, FRW = TOW + 2;  // cells a .. a+PR, b .. b+TOW (+1 pad)
    constexpr int FTOT = S2B_CK * FRH * FRW, FPT = (FTOT + 255) / 256;
    constexpr int WROW = 66;
    __shared__ float ws[S2B_CK * 4 * WROW];  // [kc][t = u*2+v][row = ci_l*4 + ph*2 + pw]
    __shared__ float fr[FTOT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int rt = wave >> 1, pt = wave & 1;
    const int ci_base = static_cast<int>(blockIdx.y) * 16;
    const int OH = d.H / 2;
    const int g0 = static_cast<int>(blockIdx.x) * PR;
    const int rows_total = d.B * OH;
    const int pj = pt * 32 + i;
    const int rj = pj / TOW, wj = pj - rj * TOW;
    const int gj = g0 + rj;
    const bool pv = gj < rows_total;
    const int nj = pv ? gj / OH : 0, aj = pv ? gj - nj * OH : 0;
    const bool last_row = aj == OH - 1;
    const float *bbase = fr + kk * (FRH * FRW) + rj * FRW + wj;
    const float *abase = ws + kk * (4 * WROW) + rt * 32 + i;
    const size_t oplane = static_cast<size_t>(OH) * TOW;

    for (int idx = threadIdx.x; idx < S2B_CK * 4 * WROW; idx += 256) ws[idx] = 0.0f;  // the structural zeros, once

    // weights: 32 co x 16 ci pairs per round, two pairs per thread, the nine taps of a pair are contiguous
    unsigned wsrc[2], wdst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pair = threadIdx.x + j * 256;
        const int kc = pair >> 4, cl = pair & 15;
        wsrc[j] = static_cast<unsigned>((kc * d.RC + ci_base + cl) * 9);
        wdst[j] = static_cast<unsigned>(kc * 4 * WROW + cl * 4);
    }
    unsigned fsrc[FPT];
    bool fok[FPT];
#pragma unroll
    for (int j = 0; j < FPT; ++j) {
        const int idx = threadIdx.x + j * 256;
        const int fc = idx % FRW, tq = idx / FRW;
        const int frow = tq % FRH, kc = idx < FTOT ? tq / FRH : 0;
        const int g = g0 + frow;
        fok[j] = idx < FTOT && g < rows_total && fc < TOW;
        const int gc = fok[j] ? g : 0, cc = fok[j] ? fc : 0;
        const int n = gc / OH, a = gc - n * OH;
        fsrc[j] = static_cast<unsigned>((n * d.KC + kc) * static_cast<int>(oplane) + a * TOW + cc);
    }
    const unsigned wstep = static_cast<unsigned>(S2B_CK * d.RC * 9), fstep = static_cast<unsigned>(S2B_CK * oplane);
    float wv[2][9], fv[FPT];
    auto prefetch = [&](unsigned round) {
        const unsigned wo = round * wstep, fo = round * fstep;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 9; ++t) wv[j][t] = w[wsrc[j] + wo + t];
#pragma unroll
        for (int j = 0; j < FPT; ++j) fv[j] = dy[fsrc[j] + fo];
    };
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned rounds = static_cast<unsigned>(d.KC / S2B_CK);
    prefetch(0);
    for (unsigned round = 0; round < rounds; ++round) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ph = kh == 1 ? 0 : 1, u = kh == 0 ? 1 : 0, pw = kw == 1 ? 0 : 1, v = kw == 0 ? 1 : 0;
                    ws[wdst[j] + (u * 2 + v) * WROW + ph * 2 + pw] = wv[j][kh * 3 + kw];
                }
#pragma unroll
        for (int j = 0; j < FPT; ++j)
            if (static_cast<int>(threadIdx.x) + j * 256 < FTOT) fr[threadIdx.x + j * 256] = fok[j] ? fv[j] : 0.0f;
        __syncthreads();
        if (round + 1 < rounds) prefetch(round + 1);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float av[S2B_CK / 4 * 4], bv[S2B_CK / 4 * 4];
#pragma unroll
            for (int c2 = 0; c2 < S2B_CK / 4; ++c2) {
                const float *ap = abase + (half * (S2B_CK / 4) + c2) * (2 * 4 * WROW);
                const float *bp = bbase + (half * (S2B_CK / 4) + c2) * (2 * FRH * FRW);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        av[c2 * 4 + u * 2 + v] = ap[(u * 2 + v) * WROW];
                        float b = bp[u * FRW + v];
                        if (u == 1 && last_row) b = 0.0f;  // the dy row below an image's last row belongs to the next image
                        bv[c2 * 4 + u * 2 + v] = b;
                    }
            }
#pragma unroll
            for (int q = 0; q < S2B_CK / 4 * 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
        }
    }
    if (!pv) return;
    // registers 4j .. 4j+3 of a lane: rows (2j + (lane >> 5)) * 4 + {0,1,2,3} = input channel 2j + (lane >> 5) of this tile, (ph, pw)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ci = ci_base + rt * 8 + 2 * j + kk;
        float *o = dx + ((static_cast<size_t>(nj) * d.RC + ci) * d.H + 2 * aj) * (2 * TOW) + 2 * wj;
        *reinterpret_cast<float2 *>(o) = make_float2(acc[4 * j + 0], acc[4 * j + 1]);
        *reinterpret_cast<float2 *>(o + 2 * TOW) = make_float2(acc[4 * j + 2], acc[4 * j + 3]);
    }
}

}  // namespace

EE_API int ee_conv3x3s2_bwd_data_f32(const float *dy, const float *weight, float *dx, int B, int Cin, int Cout, int H, int W, void *stream) {
    if (B < 0 || Cin < 1 || Cout < 1 || H < 2 || W < 2) return EE_ERR_SHAPE;
    const int OW = W / 2;
    if ((H & 1) || (W & 1) || OW > 64 || 64 % OW != 0 || Cout % S2B_CK != 0 || Cin % 16 != 0) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!dy || !weight || !dx) return EE_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(dx) & 7u) return EE_ERR_ALIGN;
    if (static_cast<int64_t>(B) * Cout * (H / 2) * OW > 0x7fffffffLL || static_cast<int64_t>(Cin) * Cout * 9 > 0x7fffffffLL) return EE_ERR_SHAPE;
    const int PR = 64 / OW;
    const int64_t rows = static_cast<int64_t>(B) * (H / 2);
    const dim3 grid(static_cast<unsigned>((rows + PR - 1) / PR), static_cast<unsigned>(Cin / 16)), block(256);
    const Conv3Dims dims{B, Cout, Cin, H, W, 0};
    hipStream_t st = as_stream(stream);
    switch (OW) {
        case 64: EE_LAUNCH((conv3x3s2_bwd_kernel<64>), grid, block, 0, st, dy, weight, dx, dims); break;
        case 32: EE_LAUNCH((conv3x3s2_bwd_kernel<32>), grid, block, 0, st, dy, weight, dx, dims); break;
        case 16: EE_LAUNCH((conv3x3s2_bwd_kernel<16>), grid, block, 0, st, dy, weight, dx, dims); break;
        case 8: EE_LAUNCH((conv3x3s2_bwd_kernel<8>), grid, block, 0, st, dy, weight, dx, dims); break;
        case 4: EE_LAUNCH((conv3x3s2_bwd_kernel<4>), grid, block, 0, st, dy, weight, dx, dims); break;
        case 2: EE_LAUNCH((conv3x3s2_bwd_kernel<2>), grid, block, 0, st, dy, weight, dx, dims); break;
        default: EE_LAUNCH((conv3x3s2_bwd_kernel<1>), grid, block, 0, st, dy, weight, dx, dims); break;
    }
    return launch_status();
}
