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
    const int n = pv ? p / OHW : 0, rem = pv ? p - n * OHW : 0;
    const int oh = rem / d.OW, ow = rem - oh * d.OW;
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
                const bool in = t0 + u < steps;
                av[u] = (in && cv) ? ap[static_cast<size_t>(t0 + u) * astride] : 0.0f;
                bv[u] = (in && pv) ? bp[static_cast<size_t>(t0 + u) * bstride] : 0.0f;
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
