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
// The weight gradient: ee_wrw.hip (together with the 3x3 / stride 2 convolution of the same block; a shortcut on its own keeps ATen's).
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
// Forward of the stem convolution Conv2d(3, 64k, kernel_size=7, stride=2, padding=3, bias=False) (resnet.py:112-113).
// MIOpen's best solver (Winograd f3x2_stride2) takes 52 us un-profiled for [100,3,64,64] -> [100,64,32,32], the most expensive
// single launch of a PGD iteration's forward pass.  Implicit GEMM on v_mfma_f32_32x32x2_f32:
//   D[co][p] = sum_{(ci, ky, kx)} W[co][ci][ky][kx] * x[n, ci, 2 oy + ky - 3, 2 ox + kx - 3],     K = 3 * 7 * 7 = 147 (+ 1)
// Round 2 padded kx to 8 (84 steps, the B operand of lane (i, half) at column 2 i + kx + half: one immediate offset per step); round 4
// walks the 147 taps flat, two per step (74 steps = 12 % fewer products; the half's frame offset is a select of two constants).
// Round 4: PERSISTENT workgroups.  Round 2's kernel (one workgroup = one 4-row tile: weights through LDS into registers, frame, 168
// products, stores, moments - one after the other; 800 workgroups for 768 slots) measured 8.8 us without its products and 32.5 with
// them: the phases added up (scripts/stem_phases.py at commit 0905c3e+2).  Now a workgroup (4 wavefronts = 2 channel halves x 2 output
// rows) loads its 84 A operands per lane ONCE and walks over units of 64 channels x 2 rows x 32 columns (unit = blockIdx.x, += gridDim.x):
// the next unit's 9 x 72 x 3 input frame travels global -> registers while the current one is multiplied out of LDS (two frame buffers),
// the result stores are fire-and-forget, one barrier per unit (two with the BatchNorm moments).
// =====================================================================================================================
namespace {

#ifndef EE_STEM_SKIP
#define EE_STEM_SKIP 0  // probe builds only (scripts/stem_phases.py): 1 no products, 2 no result stores, 8 products without their LDS operand reads
#endif
constexpr int SF_ROWS = 2, SF_FH = 2 * SF_ROWS + 5, SF_FW = 72, SF_K = 147, SF_WS = 149, SF_STEPS = (SF_K + 1) / 2;
constexpr int SF_FR = 3 * SF_FH * SF_FW, SF_FPT = (SF_FR + 255) / 256;  // frame floats, per lane
constexpr int SF_CS = SF_ROWS * 32 + 4, SF_NV = SF_ROWS * 8;            // moments exchange: channel stride (conflict-free), values per lane
constexpr int SF_WG_PER_CU = 2;
static_assert(2 * SF_FR + 64 * SF_CS <= 64 * SF_WS, "both frame buffers and the moments exchange live where the weights were staged");

struct StemUnit {
    int n, oy0, ox0;
};
__device__ __forceinline__ StemUnit stem_unit(int u, int tiles_r, int tiles_c) {
    const int tc = u % tiles_c;
    u /= tiles_c;
    const int tr = u % tiles_r;
    return StemUnit{u / tiles_r, tr * SF_ROWS, tc * 32};
}
// this lane's share of a unit's input frame: loads on clamped addresses (all in flight together), validity as a bit mask
__device__ __forceinline__ void stem_frame_load(const float *__restrict__ x, const StemUnit &t, int H, int W, float (&fv)[SF_FPT], unsigned &fok) {
    const int iy0 = 2 * t.oy0 - 3, ix0 = 2 * t.ox0 - 3;
    const float *xn = x + static_cast<size_t>(t.n) * 3 * H * W;
    fok = 0;
#pragma unroll
    for (int j = 0; j < SF_FPT; ++j) {
        const int idx = static_cast<int>(threadIdx.x) + j * 256;
        const int c = idx % SF_FW, q = idx / SF_FW;
        const int r = q % SF_FH, ci = idx < SF_FR ? q / SF_FH : 0;
        const int iy = iy0 + r, ix = ix0 + c;
        const bool ok = idx < SF_FR && iy >= 0 && iy < H && ix >= 0 && ix < W;
        fok |= ok ? 1u << j : 0u;
        fv[j] = xn[ok ? static_cast<unsigned>((ci * H + iy) * W + ix) : 0u];  // uniform base + 32-bit lane offset
    }
}
__device__ __forceinline__ void stem_frame_store(float *fr, const float (&fv)[SF_FPT], unsigned fok) {
#pragma unroll
    for (int j = 0; j < SF_FPT; ++j)
        if (static_cast<int>(threadIdx.x) + j * 256 < SF_FR) fr[threadIdx.x + j * 256] = (fok >> j) & 1u ? fv[j] : 0.0f;
}

__global__ __launch_bounds__(256, SF_WG_PER_CU) void stem_fwd_mfma_kernel(const float *__restrict__ x, const float *__restrict__ w, float *__restrict__ y,
                                                                          float *__restrict__ stats, int K, int H, int W, int tiles_r, int tiles_c,
                                                                          int units) {
    __shared__ __align__(16) float lds[64 * SF_WS];
    float *wn = lds, *fr = lds, *ex = lds + 2 * SF_FR;  // the weight rows are dead once every lane holds its A operands
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, kk = lane >> 5;
    const int mt = wave >> 1, nt = wave & 1;
    const int OH = H / 2, OW = W / 2;
    const int cb = static_cast<int>(blockIdx.y) * 64;
    int u = static_cast<int>(blockIdx.x);
    StemUnit t = stem_unit(u, tiles_r, tiles_c);
    // ---- all global loads first (clamped addresses, masked afterwards): the weights and the first frame ----------------------------------
    constexpr int WV4 = 64 * SF_K / 4, WPT = (WV4 + 255) / 256;
    float4 wv[WPT];
    const float4 *wsrc = reinterpret_cast<const float4 *>(w + static_cast<size_t>(cb) * SF_K);
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const int e = threadIdx.x + j * 256;
        wv[j] = wsrc[e < WV4 ? e : WV4 - 1];
    }
    float fv[SF_FPT];
    unsigned fok;
    stem_frame_load(x, t, H, W, fv, fok);
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const int e = threadIdx.x + j * 256;
        if (e < WV4) {
            const float v4[4] = {wv[j].x, wv[j].y, wv[j].z, wv[j].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = 4 * e + q, co = f / SF_K, k = f - co * SF_K;
                wn[co * SF_WS + k] = v4[q];
            }
        }
    }
    __syncthreads();
    // ---- this lane's 74 A operands: W[cb + mt*32 + i][k] for the flat tap index k = 2 s + kk = (ci * 7 + ky) * 7 + kx, zero for k = 147 ---------
    float a[SF_STEPS];
    const float *wrow = wn + (mt * 32 + i) * SF_WS + kk;
#pragma unroll
    for (int s = 0; s < SF_STEPS; ++s) {
        const float v = wrow[2 * s];  // (k = 147: the pad column of the row - finite, dropped)
        a[s] = (2 * s + 1 >= SF_K && kk) ? 0.0f : v;
    }
    __syncthreads();  // every lane has read its weight rows (they stay in registers): the frames take their place
    stem_frame_store(fr, fv, fok);
    __syncthreads();
    const size_t plane = static_cast<size_t>(OH) * OW;
    const int stride = static_cast<int>(gridDim.x);
    int cur = 0;
    float run_sum = 0.0f, run_m2 = 0.0f, run_n = 0.0f;  // lanes 4 c: channel c's moments over this workgroup's units
    for (; u < units; u += stride) {
        const bool more = u + stride < units;  // (uniform)
        StemUnit tn = t;
        if (more) {
            tn = stem_unit(u + stride, tiles_r, tiles_c);
            stem_frame_load(x, tn, H, W, fv, fok);  // in flight behind the products
        }
        // ---- one tile per wavefront (output row nt of the unit's two), 74 MFMAs: step s takes the taps k = 2 s, 2 s + 1 on the two lane halves;
        // the B operand of lane (i, kk) is the frame element at row 2 nt + ky(k), column 2 i + kx(k) - an offset known at compile time per half
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const float *b0 = fr + cur * SF_FR + (2 * nt) * SF_FW + 2 * i;
        // the second half's tap is one column further (p1), or - after kx = 6 - at the start of the next row (pr) / of the next plane (pc);
        // the pad tap reads the first half's element (pz): four lane bases, every step an immediate offset
        const float *p1 = b0 + kk, *pr = b0 + kk * (SF_FW - 6), *pc = b0 + kk * ((SF_FH - 6) * SF_FW - 6), *pz = b0;
#pragma unroll
        for (int s2 = 0; s2 < ((EE_STEM_SKIP & 1) ? 0 : SF_STEPS); ++s2) {
            const int k0 = 2 * s2, k1 = k0 + 1;
            const int f0 = ((k0 / 49) * SF_FH + (k0 % 49) / 7) * SF_FW + k0 % 7;
            const float *pb = k1 >= SF_K ? pz : (k0 % 7 != 6 ? p1 : (k0 % 49 != 48 ? pr : pc));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], (EE_STEM_SKIP & 8) ? fv[s2 & 3] : pb[f0], acc, 0, 0, 0);
        }
        // ---- store: lane (i, kk) holds column ox0 + i of rows acc_row(r, lane) -------------------------------------------------------------
        const int oy = t.oy0 + nt;
        if (oy < OH && (!(EE_STEM_SKIP & 2) || acc[0] == 123.456f)) {
            float *yb = y + (static_cast<size_t>(t.n) * K + cb) * plane + static_cast<size_t>(t.oy0) * OW + t.ox0;  // uniform base + 32-bit lane offsets
            const unsigned lo = static_cast<unsigned>(mt * 32 + 4 * kk) * static_cast<unsigned>(plane) + static_cast<unsigned>(nt * OW + i);
#pragma unroll
            for (int r = 0; r < 16; ++r) yb[lo + static_cast<unsigned>((r & 3) + 8 * (r >> 2)) * static_cast<unsigned>(plane)] = acc[r];  // acc_row(r, lane)
        }
        // ---- optional: this unit's share of the BatchNorm statistics that follow the stem (resnet.py:113) - per output channel
        // (sum, M2 about the unit's own mean, count) of its <= 2 x 32 values, so that bn1 does not have to read y twice for them;
        // accumulated over the workgroup's units and written once at the end; merged by ee_bn.hip (bn_pool_fwd_kernel) -------------------
        // Through LDS, not through lane shuffles (a butterfly over 32 lanes for 16 registers is 80 dependent ds_bpermute): the tile goes
        // to LDS as [64 channels][2 rows x 32 columns] (channel stride 68: conflict-free), then four lanes per channel take 16 values
        // each (two passes on registers: sum, then M2 about the tile mean) and meet through two quad exchanges.
        if (stats) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ex[(mt * 32 + acc_row(r, lane)) * SF_CS + nt * 32 + i] = acc[r];
            __syncthreads();
            const int c = threadIdx.x >> 2, q = threadIdx.x & 3;
            const int rows = OH - t.oy0 < SF_ROWS ? OH - t.oy0 : SF_ROWS;  // valid output rows of this unit (>= 1)
            float v[SF_NV];
#pragma unroll
            for (int j = 0; j < SF_NV; ++j) v[j] = ex[c * SF_CS + q + 4 * j];  // element q + 4j: row j / 8
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < SF_NV; ++j) sum += (j / 8 < rows) ? v[j] : 0.0f;
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            const float cnt = 32.0f * static_cast<float>(rows), mean = sum / cnt;
            float m2 = 0.0f;
#pragma unroll
            for (int j = 0; j < SF_NV; ++j) {
                const float d = v[j] - mean;
                m2 += (j / 8 < rows) ? d * d : 0.0f;
            }
            m2 += __shfl_xor(m2, 1);
            m2 += __shfl_xor(m2, 2);
            // this workgroup's units met so far and this one (Chan's update; a workgroup walks its units in a fixed order)
            const float n2 = run_n + cnt, d = mean - (run_n > 0.0f ? run_sum / run_n : 0.0f);
            run_m2 = (run_m2 + m2) + d * d * (run_n * cnt / n2);
            run_sum += sum;
            run_n = n2;
        }
        if (more) stem_frame_store(fr + (cur ^ 1) * SF_FR, fv, fok);
        __syncthreads();  // the next frame is complete; the moments exchange has been read
        cur ^= 1;
        t = tn;
    }
    if (stats && (threadIdx.x & 3) == 0) {  // ONE slot per workgroup and channel: stats[(co * S + workgroup) * 3 + {0,1,2}], S = gridDim.x
        float *o = stats + (static_cast<size_t>(cb + (threadIdx.x >> 2)) * gridDim.x + blockIdx.x) * 3;
        o[0] = run_sum;
        o[1] = run_m2;
        o[2] = run_n;
    }
}

}  // namespace

// persistent workgroups: as many as the chip holds at once (per 64-channel slab), each walking units blockIdx.x, + gridDim.x, ...
static int64_t stem_fwd_grid(int64_t units, int K) {
    const int64_t slots = static_cast<int64_t>(device_cus()) * SF_WG_PER_CU / (K / 64);
    return units < slots ? units : (slots > 0 ? slots : 1);
}

EE_API int ee_stem7x7s2_fwd_stats_floats(int B, int K, int H, int W) {
    if (B < 1 || K < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || (W / 2) % 32 != 0 || K % 64 != 0) return 0;
    const int64_t units = static_cast<int64_t>(B) * ((H / 2 + SF_ROWS - 1) / SF_ROWS) * (W / 2 / 32);
    if (units > 0x7fffffffLL / 3 / K) return 0;
    return static_cast<int>(stem_fwd_grid(units, K) * K * 3);  // one slot per workgroup and channel
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
    const int64_t units = static_cast<int64_t>(B) * tiles_r * tiles_c;
    if (units > 0x7fffffffLL / 3 / K || static_cast<int64_t>(B) * K * OH * OW > 0x7fffffffLL * 4LL) return EE_ERR_SHAPE;
    const int64_t grid = stem_fwd_grid(units, K);
    EE_LAUNCH(stem_fwd_mfma_kernel, dim3(static_cast<unsigned>(grid), static_cast<unsigned>(K / 64)), dim3(256), 0, as_stream(stream), x, weight, y,
              stats, K, H, W, tiles_r, tiles_c, static_cast<int>(units));
    return launch_status();
}
