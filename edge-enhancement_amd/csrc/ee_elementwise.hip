// ee_elementwise.hip - PGD / FGSM / free-AT / AVmixup element-wise kernels (HBM-bound, gfx950).
//
// Layout: flat fp32 streams.  Each lane moves 16 B per access (float4, 1 KiB per wave-instruction),
// the grid is capped at 8 workgroups per CU and grid-strides the rest.  Arithmetic follows
// oracle/ee_oracle.c operation for operation (utils/attacks.py lines cited in include/eeadv.h).
#include <math.h>

#include "ee_common.hpp"

namespace {

using namespace ee;

struct PgdStepOp {
    float a, eps, lo, hi;
    __device__ __forceinline__ float operator()(float x, float g, float x0) const {
        float t = x + a * sgn(g);
        t = tmax(t, x0 - eps);
        t = tmin(t, x0 + eps);
        return tclamp(t, lo, hi);
    }
};

// out = op(a, b, c) over n elements; VEC = 4 (all pointers 16-B aligned) or 1
template <int VEC, class Op>
__global__ __launch_bounds__(kBlock) void map3_kernel(float *out, const float *a,  // out may alias a (in-place update)
                                                      const float *__restrict__ b, const float *__restrict__ c,
                                                      int64_t n, Op op) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (VEC == 4) {
        const int64_t nv = n >> 2;
        for (int64_t v = i; v < nv; v += stride) {
            const float4 va = reinterpret_cast<const float4 *>(a)[v];
            const float4 vb = reinterpret_cast<const float4 *>(b)[v];
            const float4 vc = c ? reinterpret_cast<const float4 *>(c)[v] : make_float4(0, 0, 0, 0);
            float4 r;
            r.x = op(va.x, vb.x, vc.x);
            r.y = op(va.y, vb.y, vc.y);
            r.z = op(va.z, vb.z, vc.z);
            r.w = op(va.w, vb.w, vc.w);
            reinterpret_cast<float4 *>(out)[v] = r;
        }
        for (int64_t k = (nv << 2) + i; k < n; k += stride) out[k] = op(a[k], b[k], c ? c[k] : 0.0f);
    } else {
        for (int64_t k = i; k < n; k += stride) out[k] = op(a[k], b[k], c ? c[k] : 0.0f);
    }
}

template <class Op>
int launch_map3(float *out, const float *a, const float *b, const float *c, int64_t n, Op op, hipStream_t s) {
    if (n == 0) return EE_OK;
    const bool vec = aligned16(out) && aligned16(a) && aligned16(b) && (!c || aligned16(c));
    const int64_t work = vec ? (n + 3) / 4 : n;
    int64_t blocks = (work + kBlock - 1) / kBlock;
    if (blocks > kMaxGrid) blocks = kMaxGrid;
    if (vec)
        EE_LAUNCH((map3_kernel<4, Op>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, s, out, a, b, c, n, op);
    else
        EE_LAUNCH((map3_kernel<1, Op>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, s, out, a, b, c, n, op);
    return launch_status();
}

struct InitOp {  // a = x0, b = noise
    float lo, hi;
    __device__ __forceinline__ float operator()(float x0, float nz, float) const { return tclamp(x0 + nz, lo, hi); }
};
struct FgsmOp {  // a = x, b = g
    float a_, lo, hi;
    __device__ __forceinline__ float operator()(float x, float g, float) const { return tclamp(x + a_ * sgn(g), lo, hi); }
};
struct FreeAtOp {  // a = delta, b = g
    float alpha, eps;
    __device__ __forceinline__ float operator()(float d, float g, float) const {
        return tclamp(d + alpha * sgn(g), -eps, eps);
    }
};

struct FreeAtMaskedOp {  // a = delta, b = d loss / d clamp(x + delta), c = x
    float alpha, eps;
    __device__ __forceinline__ float operator()(float d, float g, float x) const {
        const float s = x + d;  // AT_free_imagenet_ddp.py:289-290: in1 = (x + delta).clamp_(0, 1); clamp passes the gradient on [0, 1]
        const float gd = (s >= 0.0f && s <= 1.0f) ? g : 0.0f;
        return tclamp(d + alpha * sgn(gd), -eps, eps);
    }
};

// ---- random start drawn on the device -----------------------------------------------------------
template <int DIST>
__global__ __launch_bounds__(kBlock) void init_rng_kernel(float *__restrict__ x, const float *__restrict__ x0,
                                                          int64_t n, float scale, uint64_t seed, uint64_t offset,
                                                          float lo, float hi) {
    const Philox rng(seed);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t nv = (n + 3) >> 2;
    for (int64_t v = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; v < nv; v += stride) {
        const uint4 r = rng(offset + static_cast<uint64_t>(v));
        float z[4];
        if (DIST == 0) {  // uniform_(-scale, scale): u*(to-from)+from
            const float span = scale - (-scale);
            z[0] = u01(r.x) * span + (-scale);
            z[1] = u01(r.y) * span + (-scale);
            z[2] = u01(r.z) * span + (-scale);
            z[3] = u01(r.w) * span + (-scale);
        } else {  // Box-Muller, two pairs
            const float u0 = 1.0f - u01(r.x), u1 = u01(r.y), u2 = 1.0f - u01(r.z), u3 = u01(r.w);
            const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
            z[0] = scale * (r0 * cosf(6.28318530717958647692f * u1));
            z[1] = scale * (r0 * sinf(6.28318530717958647692f * u1));
            z[2] = scale * (r1 * cosf(6.28318530717958647692f * u3));
            z[3] = scale * (r1 * sinf(6.28318530717958647692f * u3));
        }
        const int64_t base = v << 2;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (base + k < n) x[base + k] = tclamp(x0[base + k] + z[k], lo, hi);
    }
}

// ---- PGD update with the edge-branch gradient broadcast over channels ----------------------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void pgd_step_bcast_kernel(float *x, const float *__restrict__ g_lp,
                                                                const float *__restrict__ g_edge,
                                                                const float *__restrict__ x0, int C, int64_t hw,
                                                                int64_t n, PgdStepOp op) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t chw = hw * C;
    if (VEC == 4) {  // hw % 4 == 0: a float4 never straddles a plane
        for (int64_t v = i; v < (n >> 2); v += stride) {
            const int64_t e = v << 2;
            const int64_t b = e / chw, p = e % hw;
            const float4 vx = reinterpret_cast<const float4 *>(x)[v];
            const float4 vg = reinterpret_cast<const float4 *>(g_lp)[v];
            const float4 ve = *reinterpret_cast<const float4 *>(g_edge + b * hw + p);
            const float4 v0 = reinterpret_cast<const float4 *>(x0)[v];
            float4 r;
            r.x = op(vx.x, vg.x + ve.x, v0.x);
            r.y = op(vx.y, vg.y + ve.y, v0.y);
            r.z = op(vx.z, vg.z + ve.z, v0.z);
            r.w = op(vx.w, vg.w + ve.w, v0.w);
            reinterpret_cast<float4 *>(x)[v] = r;
        }
    } else {
        for (int64_t e = i; e < n; e += stride) {
            const int64_t b = e / chw, p = e % hw;
            x[e] = op(x[e], g_lp[e] + g_edge[b * hw + p], x0[e]);
        }
    }
}

// ---- AVmixup ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void avmix_kernel(float *__restrict__ out, const float *__restrict__ x,
                                                       const float *__restrict__ x0, const double *__restrict__ wgt,
                                                       int64_t per, int64_t n, float gamma) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n; k += stride) {
        const double w = wgt[k / per];
        const float p = (x[k] - x0[k]) * gamma;
        const float v = tclamp(x0[k] + p, 0.0f, 1.0f);
        out[k] = static_cast<float>(static_cast<double>(x0[k]) * w + static_cast<double>(v) * (1.0 - w));
    }
}

__global__ __launch_bounds__(kBlock) void avmix_labels_kernel(double *__restrict__ out, const int64_t *__restrict__ labels,
                                                              const double *__restrict__ wgt, int64_t B, int64_t K,
                                                              float l1, float l2) {
    const int64_t n = B * K;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    // _label_smoothing in fp32 (attacks.py:444-445): one_hot*f + (one_hot-1)*((f-1)/(K-1)), the python
    // scalar (f-1)/float(K-1) is computed in double and cast to fp32 when it meets the fp32 tensor
    const float c1 = static_cast<float>((static_cast<double>(l1) - 1.0) / static_cast<double>(K - 1));
    const float c2 = static_cast<float>((static_cast<double>(l2) - 1.0) / static_cast<double>(K - 1));
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n; k += stride) {
        const int64_t b = k / K, c = k % K;
        const float o = (labels[b] == c) ? 1.0f : 0.0f;
        const float yn = o * l1 + (o - 1.0f) * c1;
        const float yv = o * l2 + (o - 1.0f) * c2;
        const double w = wgt[b];
        out[k] = static_cast<double>(yn) * w + static_cast<double>(yv) * (1.0 - w);
    }
}

// ---- TRADES L2 step (attacks.py:389-400): per-sample RMS normalisation of the gradient, step, RMS-ball projection ----
// One workgroup per sample; three sweeps over the sample (L2-resident at every reference shape: 12 KB .. 600 KB):
//   1. ss_g = sum g^2                      -> gn = sqrt(float(ss_g / per))                         (l2_norm, :362-366: MEAN of squares)
//   2. t = x + step * (g / (gn + 1e-8)),  d = t - x0,  ss_d = sum d^2 -> dn = sqrt(float(ss_d / per))
//   3. d *= eps / dn  where dn > eps;  x = clamp(x0 + d, lo, hi)
// The two sums are accumulated in double (fixed order: lane-strided, wave butterflies, waves in order), then rounded to
// float once - ATen's fp32 pairwise sums differ from that by an ulp or so, which is why this kernel's parity test is a
// tolerance test, the only one among the update kernels.
constexpr int kL2Block = 1024;

__device__ __forceinline__ double block_sum_1024(double v, double *sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();  // sh may still be read from a previous reduction
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kL2Block / 64; ++w) t += sh[w];
    return t;
}

__global__ __launch_bounds__(kL2Block) void l2_step_kernel(float *x, const float *__restrict__ g, const float *__restrict__ x0,
                                                           int64_t per, float step, float eps, float lo, float hi) {
    __shared__ double sh[kL2Block / 64];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * per;
    x += base;
    g += base;
    x0 += base;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < per; i += kL2Block) {
        const float v = g[i];
        acc += static_cast<double>(v * v);
    }
    const float gn = sqrtf(static_cast<float>(block_sum_1024(acc, sh) / static_cast<double>(per))) + 1e-8f;
    acc = 0.0;
    for (int64_t i = threadIdx.x; i < per; i += kL2Block) {
        const float t = x[i] + step * (g[i] / gn);
        const float d = t - x0[i];
        acc += static_cast<double>(d * d);
    }
    const float dn = sqrtf(static_cast<float>(block_sum_1024(acc, sh) / static_cast<double>(per)));
    const bool shrink = dn > eps;
    const float f = eps / dn;
    for (int64_t i = threadIdx.x; i < per; i += kL2Block) {
        const float t = x[i] + step * (g[i] / gn);
        float d = t - x0[i];
        if (shrink) d *= f;
        x[i] = tclamp(x0[i] + d, lo, hi);
    }
}

int grid_for(int64_t n) {
    int64_t blocks = (n + kBlock - 1) / kBlock;
    return static_cast<int>(blocks > kMaxGrid ? kMaxGrid : (blocks < 1 ? 1 : blocks));
}

}  // namespace

EE_API int ee_pgd_init_f32(float *x, const float *x0, const float *noise, int64_t n, float lo, float hi, void *stream) {
    if (n < 0) return EE_ERR_SHAPE;
    if (n == 0) return EE_OK;  // empty batch: nothing to do, pointers may be NULL
    if (!x || !x0 || !noise) return EE_ERR_NULL;
    if (!aligned4(x) || !aligned4(x0) || !aligned4(noise)) return EE_ERR_ALIGN;
    return launch_map3(x, x0, noise, nullptr, n, InitOp{lo, hi}, as_stream(stream));
}

EE_API int ee_pgd_init_rng_f32(float *x, const float *x0, int64_t n, float scale, int dist, uint64_t seed,
                               uint64_t offset, float lo, float hi, void *stream) {
    if (n < 0 || dist < 0 || dist > 1) return EE_ERR_SHAPE;
    if (n == 0) return EE_OK;
    if (!x || !x0) return EE_ERR_NULL;
    const int g = grid_for((n + 3) / 4);
    if (dist == 0)
        EE_LAUNCH(init_rng_kernel<0>, dim3(g), dim3(kBlock), 0, as_stream(stream), x, x0, n, scale, seed, offset, lo, hi);
    else
        EE_LAUNCH(init_rng_kernel<1>, dim3(g), dim3(kBlock), 0, as_stream(stream), x, x0, n, scale, seed, offset, lo, hi);
    return launch_status();
}

EE_API int ee_pgd_step_f32(float *x, const float *g, const float *x0, int64_t n, float alpha, float eps, float lo,
                           float hi, int dir, void *stream) {
    if (n < 0 || (dir != 1 && dir != -1)) return EE_ERR_SHAPE;
    if (n == 0) return EE_OK;
    if (!x || !g || !x0) return EE_ERR_NULL;
    if (!aligned4(x) || !aligned4(g) || !aligned4(x0)) return EE_ERR_ALIGN;
    ProfScope prof(EE_K_PGD_STEP, as_stream(stream));
    return launch_map3(x, x, g, x0, n, PgdStepOp{dir > 0 ? alpha : -alpha, eps, lo, hi}, as_stream(stream));
}

EE_API int ee_fgsm_step_f32(float *out, const float *x, const float *g, int64_t n, float alpha, float lo, float hi,
                            int dir, void *stream) {
    if (n < 0 || (dir != 1 && dir != -1)) return EE_ERR_SHAPE;
    if (n == 0) return EE_OK;
    if (!out || !x || !g) return EE_ERR_NULL;
    if (!aligned4(out) || !aligned4(x) || !aligned4(g)) return EE_ERR_ALIGN;
    return launch_map3(out, x, g, nullptr, n, FgsmOp{dir > 0 ? alpha : -alpha, lo, hi}, as_stream(stream));
}

EE_API int ee_add_clamp_f32(float *out, const float *x, const float *delta, int64_t n, float lo, float hi, void *stream) {
    if (n < 0) return EE_ERR_SHAPE;
    if (n == 0) return EE_OK;
    if (!out || !x || !delta) return EE_ERR_NULL;
    if (!aligned4(out) || !aligned4(x) || !aligned4(delta)) return EE_ERR_ALIGN;
    return launch_map3(out, x, delta, nullptr, n, InitOp{lo, hi}, as_stream(stream));
}

EE_API int ee_freeat_update_f32(float *delta, const float *g, int64_t n, float alpha, float eps, void *stream) {
    if (n < 0) return EE_ERR_SHAPE;
    if (n == 0) return EE_OK;
    if (!delta || !g) return EE_ERR_NULL;
    if (!aligned4(delta) || !aligned4(g)) return EE_ERR_ALIGN;
    return launch_map3(delta, delta, g, nullptr, n, FreeAtOp{alpha, eps}, as_stream(stream));
}

EE_API int ee_freeat_update_masked_f32(float *delta, const float *g_in1, const float *x, int64_t n, float alpha, float eps, void *stream) {
    if (n < 0) return EE_ERR_SHAPE;
    if (n == 0) return EE_OK;
    if (!delta || !g_in1 || !x) return EE_ERR_NULL;
    if (!aligned4(delta) || !aligned4(g_in1) || !aligned4(x)) return EE_ERR_ALIGN;
    return launch_map3(delta, delta, g_in1, x, n, FreeAtMaskedOp{alpha, eps}, as_stream(stream));
}

EE_API int ee_pgd_step_bcast_f32(float *x, const float *g_lp, const float *g_edge, const float *x0, int B, int C,
                                 int64_t hw, float alpha, float eps, float lo, float hi, int dir, void *stream) {
    if (!x || !g_lp || !g_edge || !x0) return EE_ERR_NULL;
    if (B < 0 || C < 1 || hw < 1 || (dir != 1 && dir != -1)) return EE_ERR_SHAPE;
    if (!aligned4(x) || !aligned4(g_lp) || !aligned4(g_edge) || !aligned4(x0)) return EE_ERR_ALIGN;
    const int64_t n = static_cast<int64_t>(B) * C * hw;
    if (n == 0) return EE_OK;
    const PgdStepOp op{dir > 0 ? alpha : -alpha, eps, lo, hi};
    const bool vec = (hw % 4 == 0) && aligned16(x) && aligned16(g_lp) && aligned16(g_edge) && aligned16(x0);
    ProfScope prof(EE_K_PGD_STEP_BCAST, as_stream(stream));
    if (vec)
        EE_LAUNCH(pgd_step_bcast_kernel<4>, dim3(grid_for(n / 4)), dim3(kBlock), 0, as_stream(stream), x, g_lp,
                           g_edge, x0, C, hw, n, op);
    else
        EE_LAUNCH(pgd_step_bcast_kernel<1>, dim3(grid_for(n)), dim3(kBlock), 0, as_stream(stream), x, g_lp,
                           g_edge, x0, C, hw, n, op);
    return launch_status();
}

EE_API int ee_avmix_f32(float *out, const float *x, const float *x0, const double *wgt, int64_t B, int64_t per_sample,
                        float gamma, void *stream) {
    if (!out || !x || !x0 || !wgt) return EE_ERR_NULL;
    if (B < 0 || per_sample < 1) return EE_ERR_SHAPE;
    const int64_t n = B * per_sample;
    if (n == 0) return EE_OK;
    EE_LAUNCH(avmix_kernel, dim3(grid_for(n)), dim3(kBlock), 0, as_stream(stream), out, x, x0, wgt, per_sample, n, gamma);
    return launch_status();
}

EE_API int ee_avmix_labels_f64(double *out, const int64_t *labels, const double *wgt, int64_t B, int64_t K, float lambda1,
                               float lambda2, void *stream) {
    if (!out || !labels || !wgt) return EE_ERR_NULL;
    if (B < 0 || K < 2) return EE_ERR_SHAPE;
    if (B == 0) return EE_OK;
    EE_LAUNCH(avmix_labels_kernel, dim3(grid_for(B * K)), dim3(kBlock), 0, as_stream(stream), out, labels, wgt, B, K,
                       lambda1, lambda2);
    return launch_status();
}

EE_API int ee_l2_step_f32(float *x, const float *g, const float *x0, int64_t B, int64_t per_sample, float step, float eps,
                          float lo, float hi, void *stream) {
    if (B < 0 || per_sample < 1 || B > 0x7fffffff) return EE_ERR_SHAPE;
    if (B == 0) return EE_OK;
    if (!x || !g || !x0) return EE_ERR_NULL;
    if (!aligned4(x) || !aligned4(g) || !aligned4(x0)) return EE_ERR_ALIGN;
    EE_LAUNCH(l2_step_kernel, dim3(static_cast<unsigned>(B)), dim3(kL2Block), 0, as_stream(stream), x, g, x0, per_sample, step,
              eps, lo, hi);
    return launch_status();
}
