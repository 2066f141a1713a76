// ee_bn.hip - BatchNorm2d fused with the residual add and the ReLU that follow it in every ResNet block
// (Tiny_ImageNet/models_tinyimagenet/resnet.py:44-59: out = relu(bn(conv(x)) [+ residual])).
//
// Why it exists: at the reference batch size (100 x 64x64 images) one PGD iteration of ResNet-18 is ~170 kernel
// launches averaging ~12 us, so the attack loop is launch-bound; BatchNorm -> (+ residual) -> ReLU are three (forward)
// and two (backward) launches per layer of pure HBM-bound element-wise work.  Fused: one launch each way, the
// normalised tensor is written once, and the ReLU mask is taken from the output instead of a saved copy.
//
// Three families, chosen per shape by the launcher: (1) "cached": one workgroup per channel, the channel held in registers
// (one read pass, all loads of a lane in flight together) - every layer but the stem at the reference batch; (2) "split":
// a channel too large for that is cut into slices, one workgroup each, partial statistics through a small workspace and a
// second launch (the 64-channel stem); (3) the generic three-pass loop (any shape, also unaligned).  Training mode: mean,
// then centred variance (two-pass, fp32, fixed-order wave / workgroup reductions -> bit-reproducible), then normalise +
// add + ReLU.  Running statistics are updated in the kernel exactly as nn.BatchNorm2d does (momentum, unbiased
// variance).  NCHW fp32; 16-B accesses when H*W % 4 == 0 (every ResNet stage: 1024, 256, 64, 16, 4).
//
// This is CNN-body glue, not one of SURVEY.md section 8's rows: parity is "logits within 1e-4" through the model tests.
#include "ee_common.hpp"

namespace {

using namespace ee;

template <int NT>
__device__ __forceinline__ float block_sum(float v, float *scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();  // scratch may still be read from a previous call
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += scratch[w];
    return t;
}

// two / three sums through ONE pair of barriers (round 4: the backward kernels reduced their sums one after the other, four or six
// barriers of up to sixteen wavefronts); each sum is formed exactly as block_sum forms it: the same bits.  scratch: NV * NT / 64 floats.
template <int NT, int NV>
__device__ __forceinline__ void block_sums(float (&v)[NV], float *scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] += __shfl_xor(v[k], off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();  // scratch may still be read from a previous call
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) scratch[k * (NT / 64) + wave] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) t += scratch[k * (NT / 64) + w];
        v[k] = t;
    }
}

struct BnShape {
    int B, C, HW;
};

// the gradient may arrive in two pieces (the block output feeds the next block's convolution AND its identity branch; autograd
// would add them in a separate launch): dy + dy2, the same fp32 add, on load
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// The ReLU mask of the backward is y > 0.  Without a residual branch y = relu((x - mean) * a + b0), so the mask can be recomputed
// from x - which the backward reads anyway - with the forward's expression (the same bits) instead of reading y (one tensor less:
// the layer-1 kernels run one workgroup per channel on 64 CUs and are bound by what a CU can pull).  MaskArgs.on selects it.
struct MaskArgs {
    int on;
    float a, b0;  // invstd * gamma, beta
};
__device__ __forceinline__ float4 pre4(float4 v, float mean, const MaskArgs &m) {
    return make_float4((v.x - mean) * m.a + m.b0, (v.y - mean) * m.a + m.b0, (v.z - mean) * m.a + m.b0, (v.w - mean) * m.a + m.b0);
}
__device__ __forceinline__ MaskArgs mask_args(const float *y, const float *gamma, const float *beta, float invstd, int c) {
    MaskArgs m;
    m.on = y == nullptr;
    m.a = invstd * (gamma ? gamma[c] : 1.0f);
    m.b0 = (beta && m.on) ? beta[c] : 0.0f;
    return m;
}

// element e (0 .. B*HW) of channel c lives at ((b*C + c)*HW + p), b = e / HW, p = e % HW; VEC = 4 walks float4s (HW % 4 == 0)
template <int NT, int VEC, class F>
__device__ __forceinline__ void for_channel(const BnShape s, int c, F f) {
    const int per = s.HW / VEC;
    const int total = s.B * per;
    for (int e = threadIdx.x; e < total; e += NT) {
        const int b = e / per, q = e - b * per;
        f((static_cast<size_t>(b) * s.C + c) * s.HW + static_cast<size_t>(q) * VEC);
    }
}

template <int NT, int VEC, bool RELU, bool RES>
__global__ __launch_bounds__(NT) void bn_fwd_kernel(const float *__restrict__ x, const float *__restrict__ res, const float *__restrict__ gamma,
                                                    const float *__restrict__ beta, float *running_mean, float *running_var, float momentum,
                                                    float eps, int training, float *__restrict__ y, float *__restrict__ save_mean,
                                                    float *__restrict__ save_invstd, BnShape s) {
    __shared__ float scratch[3 * (NT / 64)];
    const int c = blockIdx.x;
    const float n = static_cast<float>(s.B) * static_cast<float>(s.HW);
    float mean, invstd;
    if (training) {
        float acc = 0.0f;
        for_channel<NT, VEC>(s, c, [&](size_t o) {
            if (VEC == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(x + o);
                acc += (v.x + v.y) + (v.z + v.w);
            } else {
                acc += x[o];
            }
        });
        mean = block_sum<NT>(acc, scratch) / n;
        float var = 0.0f;
        for_channel<NT, VEC>(s, c, [&](size_t o) {
            if (VEC == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(x + o);
                const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
                var += (a * a + b * b) + (cc * cc + d * d);
            } else {
                const float a = x[o] - mean;
                var += a * a;
            }
        });
        var = block_sum<NT>(var, scratch) / n;
        invstd = 1.0f / sqrtf(var + eps);
        if (threadIdx.x == 0) {
            save_mean[c] = mean;
            save_invstd[c] = invstd;
            if (running_mean) {
                const float unbiased = (n > 1.0f) ? var * (n / (n - 1.0f)) : var;
                running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
            }
        }
    } else {
        mean = running_mean[c];
        invstd = 1.0f / sqrtf(running_var[c] + eps);
    }
    const float a = invstd * (gamma ? gamma[c] : 1.0f), b0 = beta ? beta[c] : 0.0f;
    for_channel<NT, VEC>(s, c, [&](size_t o) {
        if (VEC == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + o);
            float4 r = make_float4((v.x - mean) * a + b0, (v.y - mean) * a + b0, (v.z - mean) * a + b0, (v.w - mean) * a + b0);
            if (RES) {
                const float4 q = *reinterpret_cast<const float4 *>(res + o);
                r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
            }
            if (RELU) {
                r.x = r.x > 0.0f ? r.x : (r.x != r.x ? r.x : 0.0f);
                r.y = r.y > 0.0f ? r.y : (r.y != r.y ? r.y : 0.0f);
                r.z = r.z > 0.0f ? r.z : (r.z != r.z ? r.z : 0.0f);
                r.w = r.w > 0.0f ? r.w : (r.w != r.w ? r.w : 0.0f);
            }
            *reinterpret_cast<float4 *>(y + o) = r;
        } else {
            float r = (x[o] - mean) * a + b0;
            if (RES) r += res[o];
            if (RELU) r = r > 0.0f ? r : (r != r ? r : 0.0f);
            y[o] = r;
        }
    });
}

// backward.  dz = RELU ? dy * (y > 0) : dy  (threshold_backward); dres = dz (the residual branch's gradient);
// training: dx = gamma*invstd * (dz - mean(dz) - xhat * mean(dz*xhat));  eval: dx = gamma*invstd_running * dz.
template <int NT, int VEC, bool RELU>
__global__ __launch_bounds__(NT) void bn_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ dy2, const float *__restrict__ y,
                                                    const float *__restrict__ x, const float *__restrict__ beta,
                                                    const float *__restrict__ gamma, const float *__restrict__ save_mean,
                                                    const float *__restrict__ save_invstd, const float *__restrict__ running_mean,
                                                    const float *__restrict__ running_var, float eps, int training, float *__restrict__ dx,
                                                    float *__restrict__ dres, float *__restrict__ dgamma, float *__restrict__ dbeta, BnShape s) {
    __shared__ float scratch[3 * (NT / 64)];
    const int c = blockIdx.x;
    const float n = static_cast<float>(s.B) * static_cast<float>(s.HW);
    const float mean = training ? save_mean[c] : running_mean[c];
    const float invstd = training ? save_invstd[c] : 1.0f / sqrtf(running_var[c] + eps);
    const MaskArgs mk = mask_args(y, gamma, beta, invstd, c);
    auto masked4 = [&](size_t o, float4 v) {
        float4 g = *reinterpret_cast<const float4 *>(dy + o);
        if (dy2) g = add4(g, *reinterpret_cast<const float4 *>(dy2 + o));
        if (RELU) {
            const float4 yy = mk.on ? pre4(v, mean, mk) : *reinterpret_cast<const float4 *>(y + o);
            g.x = yy.x > 0.0f ? g.x : 0.0f;
            g.y = yy.y > 0.0f ? g.y : 0.0f;
            g.z = yy.z > 0.0f ? g.z : 0.0f;
            g.w = yy.w > 0.0f ? g.w : 0.0f;
        }
        return g;
    };
    auto masked1 = [&](size_t o, float v) {
        float g = dy2 ? dy[o] + dy2[o] : dy[o];
        if (RELU) g = (mk.on ? (v - mean) * mk.a + mk.b0 : y[o]) > 0.0f ? g : 0.0f;
        return g;
    };
    float sdz = 0.0f, sdzx = 0.0f;
    for_channel<NT, VEC>(s, c, [&](size_t o) {
        if (VEC == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + o);
            const float4 g = masked4(o, v);
            sdz += (g.x + g.y) + (g.z + g.w);
            sdzx += (g.x * ((v.x - mean) * invstd) + g.y * ((v.y - mean) * invstd)) + (g.z * ((v.z - mean) * invstd) + g.w * ((v.w - mean) * invstd));
        } else {
            const float v = x[o];
            const float g = masked1(o, v);
            sdz += g;
            sdzx += g * ((v - mean) * invstd);
        }
    });
    {
        float two[2] = {sdz, sdzx};
        block_sums<NT, 2>(two, scratch);
        sdz = two[0], sdzx = two[1];
    }
    if (threadIdx.x == 0) {
        if (dgamma) dgamma[c] = sdzx;
        if (dbeta) dbeta[c] = sdz;
    }
    const float w = (gamma ? gamma[c] : 1.0f) * invstd;
    const float m1 = training ? sdz / n : 0.0f, m2 = training ? sdzx / n : 0.0f;
    for_channel<NT, VEC>(s, c, [&](size_t o) {
        if (VEC == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + o);
            const float4 g = masked4(o, v);
            if (dres) *reinterpret_cast<float4 *>(dres + o) = g;
            if (dx) {
                float4 r;
                r.x = w * ((g.x - m1) - ((v.x - mean) * invstd) * m2);
                r.y = w * ((g.y - m1) - ((v.y - mean) * invstd) * m2);
                r.z = w * ((g.z - m1) - ((v.z - mean) * invstd) * m2);
                r.w = w * ((g.w - m1) - ((v.w - mean) * invstd) * m2);
                *reinterpret_cast<float4 *>(dx + o) = r;
            }
        } else {
            const float v = x[o];
            const float g = masked1(o, v);
            if (dres) dres[o] = g;
            if (dx) dx[o] = w * ((g - m1) - ((v - mean) * invstd) * m2);
        }
    });
}


// ---- register-cached variants: a channel that fits MAXV float4s per lane is read from memory ONCE (the two-pass
// statistics and the normalisation run on registers), which removes two of the three dependent L2 round trips -------------
__device__ __forceinline__ float relu_nan(float r) { return r > 0.0f ? r : (r != r ? r : 0.0f); }

template <int NT, int MAXV, bool RELU, bool RES>
__global__ __launch_bounds__(NT) void bn_fwd_cached_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           float *running_mean, float *running_var, float momentum, float eps, int training,
                                                           float *__restrict__ y, float *__restrict__ save_mean,
                                                           float *__restrict__ save_invstd, BnShape s) {
    __shared__ float scratch[3 * (NT / 64)];
    const int c = blockIdx.x;
    // gridDim.y = P workgroups per channel (round 4; few-channel layers: 64 channels fill 64 of 256 CUs).  Every one of them reads the whole
    // channel and forms the SAME statistics in the same order (the same bits); part p then normalises, adds the residual for and stores only
    // the register slots j with j * P / MAXV == p - a quarter of the residual reads and of the stores per CU; part 0 owns the statistics' side
    // effects
    const int P = static_cast<int>(gridDim.y), part = static_cast<int>(blockIdx.y);
    const int per = s.HW / 4, total = s.B * per;
    const float n = static_cast<float>(s.B) * static_cast<float>(s.HW);
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const float4 *r4 = reinterpret_cast<const float4 *>(res);
    float4 xv[MAXV], rv[RES ? MAXV : 1];
    unsigned off[MAXV];
    float acc = 0.0f;
    // every load is issued unconditionally on a clamped (always valid) element and masked afterwards: a load under a
    // predicate becomes a branch with its own s_waitcnt, i.e. MAXV serialised memory round trips instead of MAXV in flight
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int e0 = threadIdx.x + j * NT;
        const int e = e0 < total ? e0 : total - 1;
        const int b = e / per, q = e - b * per;
        off[j] = static_cast<unsigned>(b * s.C + c) * static_cast<unsigned>(per) + static_cast<unsigned>(q);
        xv[j] = x4[off[j]];
    }
    if (RES) {  // needed only by the last pass: its latency hides behind the statistics.  Slots of other parts: a load of this part's first slot
        const int jf = (part * MAXV + P - 1) / P;  // (smallest j with j * P / MAXV == part)
        unsigned of = off[0];
#pragma unroll
        for (int j = 1; j < MAXV; ++j) of = j == jf ? off[j] : of;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) rv[j] = r4[(j * P / MAXV == part) ? off[j] : of];
    }
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        if (static_cast<int>(threadIdx.x) + j * NT >= total) xv[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        acc += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);
    }
    float mean, invstd;
    if (training) {
        mean = block_sum<NT>(acc, scratch) / n;
        float var = 0.0f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j)
            if (static_cast<int>(threadIdx.x) + j * NT < total) {
                const float a = xv[j].x - mean, b = xv[j].y - mean, cc = xv[j].z - mean, d = xv[j].w - mean;
                var += (a * a + b * b) + (cc * cc + d * d);
            }
        var = block_sum<NT>(var, scratch) / n;
        invstd = 1.0f / sqrtf(var + eps);
        if (threadIdx.x == 0 && part == 0) {
            save_mean[c] = mean;
            save_invstd[c] = invstd;
            if (running_mean) {
                const float unbiased = (n > 1.0f) ? var * (n / (n - 1.0f)) : var;
                running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
            }
        }
    } else {
        mean = running_mean[c];
        invstd = 1.0f / sqrtf(running_var[c] + eps);
    }
    const float a = invstd * (gamma ? gamma[c] : 1.0f), b0 = beta ? beta[c] : 0.0f;
    float4 *y4 = reinterpret_cast<float4 *>(y);
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
        if (static_cast<int>(threadIdx.x) + j * NT < total && j * P / MAXV == part) {
            const float4 v = xv[j];
            float4 r = make_float4((v.x - mean) * a + b0, (v.y - mean) * a + b0, (v.z - mean) * a + b0, (v.w - mean) * a + b0);
            if (RES) {
                const float4 q = rv[j];
                r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
            }
            if (RELU) {
                r.x = relu_nan(r.x); r.y = relu_nan(r.y); r.z = relu_nan(r.z); r.w = relu_nan(r.w);
            }
            y4[off[j]] = r;
        }
}

template <int NT, int MAXV, bool RELU>
__global__ __launch_bounds__(NT) void bn_bwd_cached_kernel(const float *__restrict__ dy, const float *__restrict__ dy2, const float *__restrict__ y,
                                                           const float *__restrict__ x, const float *__restrict__ beta,
                                                           const float *__restrict__ gamma, const float *__restrict__ save_mean,
                                                           const float *__restrict__ save_invstd, const float *__restrict__ running_mean,
                                                           const float *__restrict__ running_var, float eps, int training,
                                                           float *__restrict__ dx, float *__restrict__ dres, float *__restrict__ dgamma,
                                                           float *__restrict__ dbeta, BnShape s) {
    __shared__ float scratch[3 * (NT / 64)];
    const int c = blockIdx.x;
    const int P = static_cast<int>(gridDim.y), part = static_cast<int>(blockIdx.y);  // as in bn_fwd_cached_kernel: the sums by every part, the stores split
    const int per = s.HW / 4, total = s.B * per;
    const float n = static_cast<float>(s.B) * static_cast<float>(s.HW);
    const float mean = training ? save_mean[c] : running_mean[c];
    const float invstd = training ? save_invstd[c] : 1.0f / sqrtf(running_var[c] + eps);
    const float4 *dy4 = reinterpret_cast<const float4 *>(dy), *y4 = reinterpret_cast<const float4 *>(y), *x4 = reinterpret_cast<const float4 *>(x);
    const float4 *dy24 = reinterpret_cast<const float4 *>(dy2);
    float4 gv[MAXV], hv[MAXV];  // masked gradient dz and xhat
    unsigned off[MAXV];
    float sdz = 0.0f, sdzx = 0.0f;
    // unconditional clamped loads first (all in flight together), masking second - see bn_fwd_cached_kernel
    float4 yv[RELU ? MAXV : 1];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int e0 = threadIdx.x + j * NT;
        const int e = e0 < total ? e0 : total - 1;
        const int b = e / per, q = e - b * per;
        off[j] = static_cast<unsigned>(b * s.C + c) * static_cast<unsigned>(per) + static_cast<unsigned>(q);
        gv[j] = dy4[off[j]];
        hv[j] = x4[off[j]];
        if (RELU && y4) yv[j] = y4[off[j]];
    }
    const MaskArgs mk = mask_args(y, gamma, beta, invstd, c);
    if (dy24) {
#pragma unroll
        for (int j = 0; j < MAXV; ++j) gv[j] = add4(gv[j], dy24[off[j]]);
    }
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const bool in = static_cast<int>(threadIdx.x) + j * NT < total;
        float4 g = gv[j];
        const float4 v = hv[j];
        if (RELU) {
            const float4 yy = mk.on ? pre4(v, mean, mk) : yv[j];
            g.x = yy.x > 0.0f ? g.x : 0.0f;
            g.y = yy.y > 0.0f ? g.y : 0.0f;
            g.z = yy.z > 0.0f ? g.z : 0.0f;
            g.w = yy.w > 0.0f ? g.w : 0.0f;
        }
        float4 h = make_float4((v.x - mean) * invstd, (v.y - mean) * invstd, (v.z - mean) * invstd, (v.w - mean) * invstd);
        if (!in) g = h = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        sdz += (g.x + g.y) + (g.z + g.w);
        sdzx += (g.x * h.x + g.y * h.y) + (g.z * h.z + g.w * h.w);
        gv[j] = g;
        hv[j] = h;
    }
    {
        float two[2] = {sdz, sdzx};
        block_sums<NT, 2>(two, scratch);
        sdz = two[0], sdzx = two[1];
    }
    if (threadIdx.x == 0 && part == 0) {
        if (dgamma) dgamma[c] = sdzx;
        if (dbeta) dbeta[c] = sdz;
    }
    const float w = (gamma ? gamma[c] : 1.0f) * invstd;
    const float m1 = training ? sdz / n : 0.0f, m2 = training ? sdzx / n : 0.0f;
    float4 *dx4 = reinterpret_cast<float4 *>(dx), *dr4 = reinterpret_cast<float4 *>(dres);
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
        if (static_cast<int>(threadIdx.x) + j * NT < total && j * P / MAXV == part) {
            const float4 g = gv[j], h = hv[j];
            if (dres) dr4[off[j]] = g;
            if (dx) dx4[off[j]] = make_float4(w * ((g.x - m1) - h.x * m2), w * ((g.y - m1) - h.y * m2), w * ((g.z - m1) - h.z * m2),
                                              w * ((g.w - m1) - h.w * m2));
        }
}

// ---- a residual block with a down-sampling shortcut ends in TWO BatchNorms: relu(bn2(conv2 out) + bn_ds(conv1x1 out)) (resnet.py:54-59,
// :137-142).  Both are per-channel, so one workgroup does both for its channel - one launch each way instead of two (at these sizes
// a BatchNorm launch is 5-6 us of launch floor).  Same statistics, same expressions, same order as bn_fwd/bwd_cached_kernel run one
// after the other: bit-identical results. ---------------------------------------------------------------------------------------------
struct BnParams {
    const float *gamma, *beta;
    float *running_mean, *running_var;
    float momentum, eps;
    float *save_mean, *save_invstd;
};

template <int NT, int MAXV>
__device__ __forceinline__ void cached_stats(const float4 (&xv)[MAXV], int total, float n, const BnParams &p, int training, int c, float *scratch,
                                             float &mean, float &invstd) {
    if (training) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) acc += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);  // lanes past the end hold zeros
        mean = block_sum<NT>(acc, scratch) / n;
        float var = 0.0f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j)
            if (static_cast<int>(threadIdx.x) + j * NT < total) {
                const float a = xv[j].x - mean, b = xv[j].y - mean, cc = xv[j].z - mean, d = xv[j].w - mean;
                var += (a * a + b * b) + (cc * cc + d * d);
            }
        var = block_sum<NT>(var, scratch) / n;
        invstd = 1.0f / sqrtf(var + p.eps);
        if (threadIdx.x == 0) {
            p.save_mean[c] = mean;
            p.save_invstd[c] = invstd;
            if (p.running_mean) {
                const float unbiased = (n > 1.0f) ? var * (n / (n - 1.0f)) : var;
                p.running_mean[c] = (1.0f - p.momentum) * p.running_mean[c] + p.momentum * mean;
                p.running_var[c] = (1.0f - p.momentum) * p.running_var[c] + p.momentum * unbiased;
            }
        }
    } else {
        mean = p.running_mean[c];
        invstd = 1.0f / sqrtf(p.running_var[c] + p.eps);
    }
}

// y = relu( bn_a(xa) + bn_b(xb) ): bn_b first (it is the `residual` of the unfused call), then bn_a with the residual added
template <int NT, int MAXV>
__global__ __launch_bounds__(NT) void bn_dual_fwd_cached_kernel(const float *__restrict__ xa, const float *__restrict__ xb, BnParams pa, BnParams pb,
                                                                int training, float *__restrict__ y, BnShape s) {
    __shared__ float scratch[3 * (NT / 64)];
    const int c = blockIdx.x;
    const int per = s.HW / 4, total = s.B * per;
    const float n = static_cast<float>(s.B) * static_cast<float>(s.HW);
    const float4 *a4 = reinterpret_cast<const float4 *>(xa), *b4 = reinterpret_cast<const float4 *>(xb);
    float4 av[MAXV], bv[MAXV];
    unsigned off[MAXV];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int e0 = threadIdx.x + j * NT;
        const int e = e0 < total ? e0 : total - 1;
        const int b = e / per, q = e - b * per;
        off[j] = static_cast<unsigned>(b * s.C + c) * static_cast<unsigned>(per) + static_cast<unsigned>(q);
        av[j] = a4[off[j]];
        bv[j] = b4[off[j]];
    }
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
        if (static_cast<int>(threadIdx.x) + j * NT >= total) av[j] = bv[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float mean_b, inv_b, mean_a, inv_a;
    cached_stats<NT, MAXV>(bv, total, n, pb, training, c, scratch, mean_b, inv_b);
    cached_stats<NT, MAXV>(av, total, n, pa, training, c, scratch, mean_a, inv_a);
    const float sb = inv_b * (pb.gamma ? pb.gamma[c] : 1.0f), tb = pb.beta ? pb.beta[c] : 0.0f;
    const float sa = inv_a * (pa.gamma ? pa.gamma[c] : 1.0f), ta = pa.beta ? pa.beta[c] : 0.0f;
    float4 *y4 = reinterpret_cast<float4 *>(y);
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
        if (static_cast<int>(threadIdx.x) + j * NT < total) {
            const float4 u = bv[j], v = av[j];
            const float4 q = make_float4((u.x - mean_b) * sb + tb, (u.y - mean_b) * sb + tb, (u.z - mean_b) * sb + tb, (u.w - mean_b) * sb + tb);
            float4 r = make_float4((v.x - mean_a) * sa + ta, (v.y - mean_a) * sa + ta, (v.z - mean_a) * sa + ta, (v.w - mean_a) * sa + ta);
            r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
            y4[off[j]] = make_float4(relu_nan(r.x), relu_nan(r.y), relu_nan(r.z), relu_nan(r.w));
        }
}

// dz = (y > 0) * (dy + dy2); bn_a's backward on dz -> dxa, then bn_b's (no ReLU of its own) on the same dz -> dxb
template <int NT, int MAXV>
__global__ __launch_bounds__(NT) void bn_dual_bwd_cached_kernel(const float *__restrict__ dy, const float *__restrict__ dy2, const float *__restrict__ y,
                                                                const float *__restrict__ xa, const float *__restrict__ xb, BnParams pa, BnParams pb,
                                                                int training, float *__restrict__ dxa, float *__restrict__ dxb,
                                                                float *__restrict__ dgamma_a, float *__restrict__ dbeta_a,
                                                                float *__restrict__ dgamma_b, float *__restrict__ dbeta_b, BnShape s) {
    __shared__ float scratch[3 * (NT / 64)];
    const int c = blockIdx.x;
    const int per = s.HW / 4, total = s.B * per;
    const float n = static_cast<float>(s.B) * static_cast<float>(s.HW);
    const float mean_a = training ? pa.save_mean[c] : pa.running_mean[c], mean_b = training ? pb.save_mean[c] : pb.running_mean[c];
    const float inv_a = training ? pa.save_invstd[c] : 1.0f / sqrtf(pa.running_var[c] + pa.eps);
    const float inv_b = training ? pb.save_invstd[c] : 1.0f / sqrtf(pb.running_var[c] + pb.eps);
    const float4 *dy4 = reinterpret_cast<const float4 *>(dy), *dy24 = reinterpret_cast<const float4 *>(dy2), *y4 = reinterpret_cast<const float4 *>(y);
    const float4 *a4 = reinterpret_cast<const float4 *>(xa), *b4 = reinterpret_cast<const float4 *>(xb);
    float4 gv[MAXV], ha[MAXV], hb[MAXV], yv[MAXV];
    unsigned off[MAXV];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int e0 = threadIdx.x + j * NT;
        const int e = e0 < total ? e0 : total - 1;
        const int b = e / per, q = e - b * per;
        off[j] = static_cast<unsigned>(b * s.C + c) * static_cast<unsigned>(per) + static_cast<unsigned>(q);
        gv[j] = dy4[off[j]];
        ha[j] = a4[off[j]];
        hb[j] = b4[off[j]];
        yv[j] = y4[off[j]];
    }
    if (dy24) {
#pragma unroll
        for (int j = 0; j < MAXV; ++j) gv[j] = add4(gv[j], dy24[off[j]]);
    }
    float sdz = 0.0f, sa = 0.0f, sb = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const bool in = static_cast<int>(threadIdx.x) + j * NT < total;
        float4 g = gv[j];
        const float4 yy = yv[j];
        g.x = yy.x > 0.0f ? g.x : 0.0f;
        g.y = yy.y > 0.0f ? g.y : 0.0f;
        g.z = yy.z > 0.0f ? g.z : 0.0f;
        g.w = yy.w > 0.0f ? g.w : 0.0f;
        const float4 va = ha[j], vb = hb[j];
        float4 xa_ = make_float4((va.x - mean_a) * inv_a, (va.y - mean_a) * inv_a, (va.z - mean_a) * inv_a, (va.w - mean_a) * inv_a);
        float4 xb_ = make_float4((vb.x - mean_b) * inv_b, (vb.y - mean_b) * inv_b, (vb.z - mean_b) * inv_b, (vb.w - mean_b) * inv_b);
        if (!in) g = xa_ = xb_ = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        sdz += (g.x + g.y) + (g.z + g.w);
        sa += (g.x * xa_.x + g.y * xa_.y) + (g.z * xa_.z + g.w * xa_.w);
        sb += (g.x * xb_.x + g.y * xb_.y) + (g.z * xb_.z + g.w * xb_.w);
        gv[j] = g;
        ha[j] = xa_;
        hb[j] = xb_;
    }
    {
        float three[3] = {sdz, sa, sb};
        block_sums<NT, 3>(three, scratch);
        sdz = three[0], sa = three[1], sb = three[2];
    }
    if (threadIdx.x == 0) {
        if (dgamma_a) dgamma_a[c] = sa;
        if (dbeta_a) dbeta_a[c] = sdz;
        if (dgamma_b) dgamma_b[c] = sb;
        if (dbeta_b) dbeta_b[c] = sdz;
    }
    const float wa = (pa.gamma ? pa.gamma[c] : 1.0f) * inv_a, wb = (pb.gamma ? pb.gamma[c] : 1.0f) * inv_b;
    const float m1 = training ? sdz / n : 0.0f, ma = training ? sa / n : 0.0f, mb = training ? sb / n : 0.0f;
    float4 *da4 = reinterpret_cast<float4 *>(dxa), *db4 = reinterpret_cast<float4 *>(dxb);
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
        if (static_cast<int>(threadIdx.x) + j * NT < total) {
            const float4 g = gv[j], p = ha[j], q = hb[j];
            if (dxa) da4[off[j]] = make_float4(wa * ((g.x - m1) - p.x * ma), wa * ((g.y - m1) - p.y * ma), wa * ((g.z - m1) - p.z * ma), wa * ((g.w - m1) - p.w * ma));
            if (dxb) db4[off[j]] = make_float4(wb * ((g.x - m1) - q.x * mb), wb * ((g.y - m1) - q.y * mb), wb * ((g.z - m1) - q.z * mb), wb * ((g.w - m1) - q.w * mb));
        }
}

// workgroups per channel of the register-cached kernels: as many as keep the grid within the chip (layer1's 64 channels: 4; 128: 2), at most one
// per register slot
inline int cached_parts(int C, int maxv) {
    static const int off = [] { const char *e = getenv("EEADV_BN_PARTS"); return (e && e[0] == '0') ? 1 : 0; }();
    if (off) return 1;
    int p = device_cus() / (C > 0 ? C : 1);
    p = p > 4 ? 4 : p;
    p = p > maxv ? maxv : p;
    return p < 1 ? 1 : p;
}

template <int NT, int MAXV>
void launch_fwd_cached(bool relu, bool has_res, hipStream_t st, const float *x, const float *res, const float *gamma, const float *beta, float *rm,
                       float *rv, float momentum, float eps, int training, float *y, float *sm, float *si, BnShape s) {
    const dim3 grid(static_cast<unsigned>(s.C), static_cast<unsigned>(cached_parts(s.C, MAXV))), block(NT);
    if (relu && has_res)
        EE_LAUNCH((bn_fwd_cached_kernel<NT, MAXV, true, true>), grid, block, 0, st, x, res, gamma, beta, rm, rv, momentum, eps, training, y, sm, si, s);
    else if (relu)
        EE_LAUNCH((bn_fwd_cached_kernel<NT, MAXV, true, false>), grid, block, 0, st, x, res, gamma, beta, rm, rv, momentum, eps, training, y, sm, si, s);
    else if (has_res)
        EE_LAUNCH((bn_fwd_cached_kernel<NT, MAXV, false, true>), grid, block, 0, st, x, res, gamma, beta, rm, rv, momentum, eps, training, y, sm, si, s);
    else
        EE_LAUNCH((bn_fwd_cached_kernel<NT, MAXV, false, false>), grid, block, 0, st, x, res, gamma, beta, rm, rv, momentum, eps, training, y, sm, si, s);
}

template <int NT, int MAXV>
void launch_bwd_cached(bool relu, hipStream_t st, const float *dy, const float *dy2, const float *y, const float *x, const float *beta, const float *gamma, const float *sm,
                       const float *si, const float *rm, const float *rv, float eps, int training, float *dx, float *dres, float *dgamma,
                       float *dbeta, BnShape s) {
    const dim3 grid(static_cast<unsigned>(s.C), static_cast<unsigned>(cached_parts(s.C, MAXV))), block(NT);
    if (relu)
        EE_LAUNCH((bn_bwd_cached_kernel<NT, MAXV, true>), grid, block, 0, st, dy, dy2, y, x, beta, gamma, sm, si, rm, rv, eps, training, dx, dres, dgamma, dbeta, s);
    else
        EE_LAUNCH((bn_bwd_cached_kernel<NT, MAXV, false>), grid, block, 0, st, dy, dy2, y, x, beta, gamma, sm, si, rm, rv, eps, training, dx, dres, dgamma, dbeta, s);
}

// which cached variant holds a channel of `quads` float4s: 0 = none
inline int cached_variant(int64_t quads, int64_t numel) {
    if (numel / 4 > 0xffffffffLL) return 0;  // 32-bit float4 offsets
    if (quads <= 2 * 256) return 1;          // <256, 2>
    if (quads <= 7 * 256) return 2;          // <256, 7>
    if (quads <= 7 * 1024) return 3;         // <1024, 7>
    return 0;
}


// ---- split variants: a channel too large for one workgroup's registers (the ResNet stem: 102 400 values per channel at the
// reference batch, only 64 channels) is cut into S slices, one workgroup each, in two launches: partial statistics into a
// small workspace, then every workgroup recombines the S partials in the same fixed order (Chan's parallel variance
// update, so the two-pass accuracy is kept) and handles its slice.  One workgroup per channel left 3/4 of the chip idle
// and ran at the per-CU L2 bandwidth (35 us forward / 50 us backward for the stem, rocprofv3). --------------------------
constexpr int SPLIT_NT = 256;
constexpr int SPLIT_MAX = 64;

inline int split_slices(int64_t quads) {
    const int64_t s = (quads + 2047) / 2048;
    return static_cast<int>(s < 1 ? 1 : (s > SPLIT_MAX ? SPLIT_MAX : s));
}

struct Slice {
    int begin, end;  // float4 index range inside the channel
};
__device__ __forceinline__ Slice my_slice(int total, int S, int s) {
    const int chunk = (total + S - 1) / S;
    const int b = s * chunk, e = b + chunk;
    return {b < total ? b : total, e < total ? e : total};
}

// walks the slice four elements per lane at a time: `load` (memory only, no side effects) runs unconditionally on clamped
// offsets for all four - four round trips in flight - and `use` then runs on the valid ones
template <class L, class U>
__device__ __forceinline__ void for_slice(const BnShape s, int c, Slice sl, L load, U use) {
    const int per = s.HW / 4;
    if (sl.end <= sl.begin) return;
    for (int e0 = sl.begin + threadIdx.x; e0 < sl.end; e0 += 4 * SPLIT_NT) {
        size_t off[4];
        decltype(load(size_t{})) val[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e1 = e0 + u * SPLIT_NT;
            const int e = e1 < sl.end ? e1 : sl.end - 1;
            const int b = e / per, q = e - b * per;
            off[u] = (static_cast<size_t>(b) * s.C + c) * per + q;
            val[u] = load(off[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + u * SPLIT_NT < sl.end) use(val[u], off[u]);
    }
}

// Chan's recombination of the S per-slice (sum, M2) partials of bn_split_stats_kernel, in slice order (as bn_split_apply_kernel).
// The partials come through LDS in ONE parallel load: a loop of dependent global loads (S up to 64) cost 10 us per workgroup.
__device__ __forceinline__ void combine_slices(const float *__restrict__ ws, int c, int S, int total, float n, float *sh, float &mean, float &var) {
    for (int i = threadIdx.x; i < 2 * S; i += SPLIT_NT) sh[i] = ws[static_cast<size_t>(c) * S * 2 + i];
    __syncthreads();
    float sum = 0.0f;
#pragma unroll 4
    for (int i = 0; i < S; ++i) sum += sh[2 * i];
    mean = sum / n;
    float m2 = 0.0f;
#pragma unroll 4
    for (int i = 0; i < S; ++i) {
        const Slice sl = my_slice(total, S, i);
        const float cnt = 4.0f * static_cast<float>(sl.end - sl.begin);
        if (cnt > 0.0f) {
            const float d = sh[2 * i] / cnt - mean;
            m2 += sh[2 * i + 1] + cnt * (d * d);
        }
    }
    var = m2 / n;
}

// ws[(c*S + s)*2 + {0,1}] = (sum, M2 about the slice's own mean)
__global__ __launch_bounds__(SPLIT_NT) void bn_split_stats_kernel(const float *__restrict__ x, float *__restrict__ ws, BnShape s, int S) {
    __shared__ float scratch[3 * (SPLIT_NT / 64)];
    const int c = blockIdx.x, sl_i = blockIdx.y;
    const Slice sl = my_slice(s.B * (s.HW / 4), S, sl_i);
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    float acc = 0.0f;
    for_slice(s, c, sl, [&](size_t o) { return x4[o]; }, [&](float4 v, size_t) { acc += (v.x + v.y) + (v.z + v.w); });
    const float sum = block_sum<SPLIT_NT>(acc, scratch);
    const float cnt = 4.0f * static_cast<float>(sl.end - sl.begin);
    const float mean = cnt > 0.0f ? sum / cnt : 0.0f;
    float m2 = 0.0f;
    for_slice(s, c, sl, [&](size_t o) { return x4[o]; }, [&](float4 v, size_t) {
        const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
        m2 += (a * a + b * b) + (cc * cc + d * d);
    });
    m2 = block_sum<SPLIT_NT>(m2, scratch);
    if (threadIdx.x == 0) {
        ws[(static_cast<size_t>(c) * S + sl_i) * 2 + 0] = sum;
        ws[(static_cast<size_t>(c) * S + sl_i) * 2 + 1] = m2;
    }
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(SPLIT_NT) void bn_split_apply_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                  float *running_mean, float *running_var, float momentum, float eps,
                                                                  int training, float *__restrict__ y, float *__restrict__ save_mean,
                                                                  float *__restrict__ save_invstd, const float *__restrict__ ws, BnShape s, int S) {
    const int c = blockIdx.x, sl_i = blockIdx.y;
    const int total = s.B * (s.HW / 4);
    const float n = static_cast<float>(s.B) * static_cast<float>(s.HW);
    float mean, invstd;
    if (training) {  // every lane recombines the S partials itself, in slice order: identical in all workgroups
        __shared__ float parts[2 * SPLIT_MAX];
        float var;
        combine_slices(ws, c, S, total, n, parts, mean, var);
        invstd = 1.0f / sqrtf(var + eps);
        if (sl_i == 0 && threadIdx.x == 0) {
            save_mean[c] = mean;
            save_invstd[c] = invstd;
            if (running_mean) {
                const float unbiased = (n > 1.0f) ? var * (n / (n - 1.0f)) : var;
                running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
            }
        }
    } else {
        mean = running_mean[c];
        invstd = 1.0f / sqrtf(running_var[c] + eps);
    }
    const float a = invstd * (gamma ? gamma[c] : 1.0f), b0 = beta ? beta[c] : 0.0f;
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *r4 = reinterpret_cast<const float4 *>(res);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    struct XR {
        float4 x, r;
    };
    for_slice(s, c, my_slice(total, S, sl_i), [&](size_t o) { return XR{x4[o], RES ? r4[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f)}; },
              [&](XR in, size_t o) {
        const float4 v = in.x;
        float4 r = make_float4((v.x - mean) * a + b0, (v.y - mean) * a + b0, (v.z - mean) * a + b0, (v.w - mean) * a + b0);
        if (RES) {
            const float4 q = in.r;
            r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
        }
        if (RELU) {
            r.x = relu_nan(r.x); r.y = relu_nan(r.y); r.z = relu_nan(r.z); r.w = relu_nan(r.w);
        }
        y4[o] = r;
    });
}

struct BwdIn {
    float4 dy, y, x;
};
__device__ __forceinline__ BwdIn bwd_load(const float4 *dy4, const float4 *dy24, const float4 *y4, const float4 *x4, size_t o, bool relu, bool want_x,
                                          float mean, const MaskArgs &mk) {
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 xv = (want_x || (relu && mk.on)) ? x4[o] : z;
    return BwdIn{dy24 ? add4(dy4[o], dy24[o]) : dy4[o], relu ? (mk.on ? pre4(xv, mean, mk) : y4[o]) : z, xv};
}
__device__ __forceinline__ float4 masked_dz(const BwdIn &in, bool relu) {
    float4 g = in.dy;
    if (relu) {
        g.x = in.y.x > 0.0f ? g.x : 0.0f;
        g.y = in.y.y > 0.0f ? g.y : 0.0f;
        g.z = in.y.z > 0.0f ? g.z : 0.0f;
        g.w = in.y.w > 0.0f ? g.w : 0.0f;
    }
    return g;
}

// ws[(c*S + s)*2 + {0,1}] = (sum dz, sum dz*xhat) of the slice
template <bool RELU>
__global__ __launch_bounds__(SPLIT_NT) void bn_split_bwd_partial_kernel(const float *__restrict__ dy, const float *__restrict__ dy2, const float *__restrict__ y,
                                                                        const float *__restrict__ x, const float *__restrict__ gamma,
                                                                        const float *__restrict__ beta, const float *__restrict__ save_mean,
                                                                        const float *__restrict__ save_invstd,
                                                                        const float *__restrict__ running_mean,
                                                                        const float *__restrict__ running_var, float eps, int training,
                                                                        float *__restrict__ ws, BnShape s, int S) {
    __shared__ float scratch[3 * (SPLIT_NT / 64)];
    const int c = blockIdx.x, sl_i = blockIdx.y;
    const float mean = training ? save_mean[c] : running_mean[c];
    const float invstd = training ? save_invstd[c] : 1.0f / sqrtf(running_var[c] + eps);
    const float4 *dy4 = reinterpret_cast<const float4 *>(dy), *y4 = reinterpret_cast<const float4 *>(y), *x4 = reinterpret_cast<const float4 *>(x);
    const MaskArgs mk = mask_args(y, gamma, beta, invstd, c);
    float sdz = 0.0f, sdzx = 0.0f;
    for_slice(s, c, my_slice(s.B * (s.HW / 4), S, sl_i), [&](size_t o) { return bwd_load(dy4, reinterpret_cast<const float4 *>(dy2), y4, x4, o, RELU, true, mean, mk); }, [&](BwdIn in, size_t) {
        const float4 g = masked_dz(in, RELU);
        const float4 v = in.x;
        sdz += (g.x + g.y) + (g.z + g.w);
        sdzx += (g.x * ((v.x - mean) * invstd) + g.y * ((v.y - mean) * invstd)) + (g.z * ((v.z - mean) * invstd) + g.w * ((v.w - mean) * invstd));
    });
    {
        float two[2] = {sdz, sdzx};
        block_sums<SPLIT_NT, 2>(two, scratch);
        sdz = two[0], sdzx = two[1];
    }
    if (threadIdx.x == 0) {
        ws[(static_cast<size_t>(c) * S + sl_i) * 2 + 0] = sdz;
        ws[(static_cast<size_t>(c) * S + sl_i) * 2 + 1] = sdzx;
    }
}

template <bool RELU>
__global__ __launch_bounds__(SPLIT_NT) void bn_split_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ dy2, const float *__restrict__ y,
                                                                      const float *__restrict__ x, const float *__restrict__ beta, const float *__restrict__ gamma,
                                                                      const float *__restrict__ save_mean, const float *__restrict__ save_invstd,
                                                                      const float *__restrict__ running_mean,
                                                                      const float *__restrict__ running_var, float eps, int training,
                                                                      float *__restrict__ dx, float *__restrict__ dres, float *__restrict__ dgamma,
                                                                      float *__restrict__ dbeta, const float *__restrict__ ws, BnShape s, int S) {
    const int c = blockIdx.x, sl_i = blockIdx.y;
    const float n = static_cast<float>(s.B) * static_cast<float>(s.HW);
    const float mean = training ? save_mean[c] : running_mean[c];
    const float invstd = training ? save_invstd[c] : 1.0f / sqrtf(running_var[c] + eps);
    __shared__ float parts[2 * SPLIT_MAX];
    for (int i = threadIdx.x; i < 2 * S; i += SPLIT_NT) parts[i] = ws[static_cast<size_t>(c) * S * 2 + i];
    __syncthreads();
    float sdz = 0.0f, sdzx = 0.0f;
#pragma unroll 4
    for (int i = 0; i < S; ++i) {
        sdz += parts[2 * i];
        sdzx += parts[2 * i + 1];
    }
    if (sl_i == 0 && threadIdx.x == 0) {
        if (dgamma) dgamma[c] = sdzx;
        if (dbeta) dbeta[c] = sdz;
    }
    if (!dx && !dres) return;
    const float w = (gamma ? gamma[c] : 1.0f) * invstd;
    const float m1 = training ? sdz / n : 0.0f, m2 = training ? sdzx / n : 0.0f;
    const float4 *dy4 = reinterpret_cast<const float4 *>(dy), *y4 = reinterpret_cast<const float4 *>(y), *x4 = reinterpret_cast<const float4 *>(x);
    float4 *dx4 = reinterpret_cast<float4 *>(dx), *dr4 = reinterpret_cast<float4 *>(dres);
    const bool want_x = dx != nullptr;
    const MaskArgs mk = mask_args(y, gamma, beta, invstd, c);
    for_slice(s, c, my_slice(s.B * (s.HW / 4), S, sl_i), [&](size_t o) { return bwd_load(dy4, reinterpret_cast<const float4 *>(dy2), y4, x4, o, RELU, want_x, mean, mk); }, [&](BwdIn in, size_t o) {
        const float4 g = masked_dz(in, RELU);
        if (dres) dr4[o] = g;
        if (dx) {
            const float4 v = in.x;
            dx4[o] = make_float4(w * ((g.x - m1) - ((v.x - mean) * invstd) * m2), w * ((g.y - m1) - ((v.y - mean) * invstd) * m2),
                                 w * ((g.z - m1) - ((v.z - mean) * invstd) * m2), w * ((g.w - m1) - ((v.w - mean) * invstd) * m2));
        }
    });
}

// ---- the stem: relu(bn(x)) followed by MaxPool2d(3, 2, 1) (resnet.py:113-117) without the full-resolution activation -----------------
// The stem's [100,64,32,32] map is the largest tensor of the network (26 MB): writing relu(bn(x)) and reading it back for the pool,
// then - backward - writing the pool's input gradient and reading it together with y and x twice, made the stem's BatchNorm + pool
// 36 us forward and 49 us backward (242 MB of traffic).  Here the forward reads x once and writes the pooled map and its one-byte
// argmax codes (34 MB); the backward never sees a full-resolution gradient or activation: both of its passes gather the pooled
// gradient through the codes (ATen's accumulation order) and recompute the ReLU mask from x (the same expression, the same bits).
// One workgroup = one channel x a few images; a plane is staged in LDS as y = relu(bn(x)) and pooled from there with
// ee_pool.hip's scan (first maximum wins, a NaN always wins): values, codes and gradients equal the unfused sequence's bit for bit
// (the gradient sums are taken over a different partition, so dgamma / dbeta / dx agree to rounding).
struct PoolShape {
    int B, C, H, W, OH, OW, IPW, G;  // IPW images per workgroup, G = ceil(B / IPW) groups
    int nosums;                      // backward, eval mode, no parameter gradients wanted: pass 1 did not run, pass 2 needs no sums
};

// Statistics handed over by the producing convolution (ee_conv.hip: the stem forward writes, per output channel and workgroup,
// (sum, M2 about the tile's own mean, count)): one workgroup per channel merges the S partials - each lane its share in index
// order, then a fixed butterfly (Chan's update) - and writes mean / invstd and the running statistics.  3.5 us instead of the two
// passes over the 26 MB map that bn_split_stats_kernel needs.
struct Moments {
    float n, mean, m2;
};
__device__ __forceinline__ Moments merge(Moments a, Moments b) {
    if (b.n <= 0.0f) return a;
    if (a.n <= 0.0f) return b;
    const float n = a.n + b.n, d = b.mean - a.mean;
    return Moments{n, a.mean + d * (b.n / n), (a.m2 + b.m2) + d * d * (a.n * b.n / n)};
}
__global__ __launch_bounds__(SPLIT_NT) void bn_stats_finalize_kernel(const float *__restrict__ parts, int S, float eps, float momentum,
                                                                     float *running_mean, float *running_var, float *__restrict__ save_mean,
                                                                     float *__restrict__ save_invstd) {
    __shared__ Moments sh[SPLIT_NT / 64];
    const int c = blockIdx.x;
    const float *p = parts + static_cast<size_t>(c) * S * 3;
    Moments m{0.0f, 0.0f, 0.0f};
    for (int i = threadIdx.x; i < S; i += SPLIT_NT) {
        const float n = p[3 * i + 2];
        m = merge(m, Moments{n, n > 0.0f ? p[3 * i] / n : 0.0f, p[3 * i + 1]});
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {  // lower lane first at every level: lane 0 ends with the merge in a fixed order
        Moments o{__shfl_xor(m.n, off), __shfl_xor(m.mean, off), __shfl_xor(m.m2, off)};
        m = (threadIdx.x & off) ? merge(o, m) : merge(m, o);
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        Moments t = sh[0];
#pragma unroll
        for (int w = 1; w < SPLIT_NT / 64; ++w) t = merge(t, sh[w]);
        const float var = t.n > 0.0f ? t.m2 / t.n : 0.0f;
        save_mean[c] = t.mean;
        save_invstd[c] = 1.0f / sqrtf(var + eps);
        if (running_mean) {
            const float unbiased = (t.n > 1.0f) ? var * (t.n / (t.n - 1.0f)) : var;
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * t.mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
        }
    }
}

constexpr int POOL_MERGE_MAX = 7;  // units per lane that bn_pool_fwd_kernel merges itself (1792 units: batch 112 of the 64 x 64 stem)

__global__ __launch_bounds__(SPLIT_NT) void bn_pool_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta, float *running_mean, float *running_var,
                                                               float momentum, float eps, int training, float *__restrict__ yp,
                                                               uint8_t *__restrict__ code, float *__restrict__ save_mean,
                                                               float *__restrict__ save_invstd, const float *__restrict__ ws, PoolShape p, int S,
                                                               float *__restrict__ xa) {
    extern __shared__ __align__(16) float plane[];  // y = relu(bn(x)) of one image plane
    __shared__ float parts[2 * SPLIT_MAX];
    const int c = blockIdx.x, g = blockIdx.y;
    const int HW = p.H * p.W, HWq = HW / 4, OHW = p.OH * p.OW;
    const float n = static_cast<float>(p.B) * static_cast<float>(HW);
    float mean, invstd;
    if (training == 2) {  // statistics already final (bn_stats_finalize_kernel)
        mean = save_mean[c];
        invstd = save_invstd[c];
    } else if (training == 3) {
        // the producing convolution's per-unit moments (ws = [C][S][3]: sum, M2 about the unit's mean, count; S <= POOL_MERGE_MAX * SPLIT_NT) merged
        // HERE, by every workgroup of the channel the same way (the same bits): mean = sum sum_i / sum n_i, M2 = sum (M2_i + n_i (mean_i - mean)^2)
        // - two parallel sums, no chain of dependent divisions (Chan's sequential update, which made this merge 15 us in round 3) - instead of
        // a launch of its own (bn_stats_finalize_kernel: 8.5 us for 1600 units)
        const float *q = ws + static_cast<size_t>(c) * S * 3;
        float ps[POOL_MERGE_MAX], pm[POOL_MERGE_MAX], pn[POOL_MERGE_MAX];
#pragma unroll
        for (int j = 0; j < POOL_MERGE_MAX; ++j) {
            const int i = static_cast<int>(threadIdx.x) + j * SPLIT_NT, ic = i < S ? i : S - 1;
            ps[j] = q[3 * ic], pm[j] = q[3 * ic + 1], pn[j] = q[3 * ic + 2];
            if (i >= S) ps[j] = pm[j] = pn[j] = 0.0f;
        }
        float two[2] = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < POOL_MERGE_MAX; ++j) two[0] += ps[j], two[1] += pn[j];
        block_sums<SPLIT_NT, 2>(two, parts);
        const float cnt = two[1];
        mean = cnt > 0.0f ? two[0] / cnt : 0.0f;
        float m2 = 0.0f;
#pragma unroll
        for (int j = 0; j < POOL_MERGE_MAX; ++j) {
            const float d = (pn[j] > 0.0f ? ps[j] / pn[j] : mean) - mean;
            m2 += pm[j] + pn[j] * d * d;
        }
        __syncthreads();  // (the scratch of the first pair of sums is read)
        m2 = block_sum<SPLIT_NT>(m2, parts);
        const float var = cnt > 0.0f ? m2 / cnt : 0.0f;
        invstd = 1.0f / sqrtf(var + eps);
        if (g == 0 && threadIdx.x == 0) {
            save_mean[c] = mean;
            save_invstd[c] = invstd;
            if (running_mean) {
                const float unbiased = (cnt > 1.0f) ? var * (cnt / (cnt - 1.0f)) : var;
                running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
            }
        }
    } else if (training) {
        float var;
        combine_slices(ws, c, S, p.B * HWq, n, parts, mean, var);
        invstd = 1.0f / sqrtf(var + eps);
        if (g == 0 && threadIdx.x == 0) {
            save_mean[c] = mean;
            save_invstd[c] = invstd;
            if (running_mean) {
                const float unbiased = (n > 1.0f) ? var * (n / (n - 1.0f)) : var;
                running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
            }
        }
    } else {
        mean = running_mean[c];
        invstd = 1.0f / sqrtf(running_var[c] + eps);
    }
    const float a = invstd * (gamma ? gamma[c] : 1.0f), b0 = beta ? beta[c] : 0.0f;
    const int b_end = (g + 1) * p.IPW < p.B ? (g + 1) * p.IPW : p.B;
    for (int b = g * p.IPW; b < b_end; ++b) {
        const size_t pl = static_cast<size_t>(b) * p.C + c;
        const float4 *x4 = reinterpret_cast<const float4 *>(x) + pl * HWq;
        for (int q = threadIdx.x; q < HWq; q += SPLIT_NT) {
            const float4 v = x4[q];
            reinterpret_cast<float4 *>(plane)[q] = make_float4(relu_nan((v.x - mean) * a + b0), relu_nan((v.y - mean) * a + b0),
                                                               relu_nan((v.z - mean) * a + b0), relu_nan((v.w - mean) * a + b0));
        }
        __syncthreads();
        for (int o = threadIdx.x; o < OHW; o += SPLIT_NT) {
            const int oh = o / p.OW, ow = o - oh * p.OW;
            const int h0 = oh * 2 - 1, w0 = ow * 2 - 1;
            const int hs = h0 < 0 ? 0 : h0, wsx = w0 < 0 ? 0 : w0;
            const int he = h0 + 3 > p.H ? p.H : h0 + 3, we = w0 + 3 > p.W ? p.W : w0 + 3;
            float best = -INFINITY;
            int bc = (hs - h0) * 3 + (wsx - w0);
            for (int h = hs; h < he; ++h)
                for (int w = wsx; w < we; ++w) {
                    const float v = plane[h * p.W + w];
                    if (v > best || v != v) {  // ATen's max_pool2d: first maximum wins, a NaN always wins (ee_pool.hip)
                        best = v;
                        bc = (h - h0) * 3 + (w - w0);
                    }
                }
            yp[pl * OHW + o] = best;
            code[pl * OHW + o] = static_cast<uint8_t>(bc);
            // x at the window's argmax (optional): the backward's batch sums then need the pooled tensors only (bn_pool_bwd_sums_pooled_kernel)
            if (xa) xa[pl * OHW + o] = x[pl * HW + (h0 + bc / 3) * p.W + (w0 + bc % 3)];
        }
        __syncthreads();
    }
}

// pass 1 of the backward from POOLED tensors (round 4): a position's masked gradient dz is the sum of the pooled gradients of the windows
// whose argmax it is, so  sum dz = sum over windows (pre > 0 ? g : 0)  and  sum dz * xhat = sum over windows (pre > 0 ? g * xhat : 0)  with
// pre / xhat taken from x at the window's argmax (xa, written by the forward): 13-20 MB read instead of the 26 MB map + pooled gradient +
// codes.  Same partition and workspace as bn_pool_bwd_kernel<false>; the sums are taken in another order (rounding-level difference).
__global__ __launch_bounds__(SPLIT_NT) void bn_pool_bwd_sums_pooled_kernel(const float *__restrict__ dyp, const float *__restrict__ dyp2, const float *__restrict__ xa,
                                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                           const float *__restrict__ save_mean, const float *__restrict__ save_invstd,
                                                                           float *__restrict__ ws, PoolShape p) {
    __shared__ float scratch[3 * (SPLIT_NT / 64)];
    const int c = blockIdx.x, g = blockIdx.y;
    const int OHW = p.OH * p.OW;
    const float mean = save_mean[c], invstd = save_invstd[c];
    const float a = invstd * (gamma ? gamma[c] : 1.0f), b0 = beta ? beta[c] : 0.0f;
    const int b_end = (g + 1) * p.IPW < p.B ? (g + 1) * p.IPW : p.B;
    float sdz = 0.0f, sdzx = 0.0f;
    for (int b = g * p.IPW; b < b_end; ++b) {
        const size_t pl = (static_cast<size_t>(b) * p.C + c) * OHW;
        for (int o0 = threadIdx.x; o0 < OHW; o0 += 4 * SPLIT_NT) {
            float gv[4], g2[4], xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // all loads first (clamped), then the arithmetic
                const int o = o0 + u * SPLIT_NT < OHW ? o0 + u * SPLIT_NT : OHW - 1;
                gv[u] = dyp[pl + o];
                g2[u] = dyp2 ? dyp2[pl + o] : 0.0f;
                xv[u] = xa[pl + o];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (o0 + u * SPLIT_NT < OHW) {
                    const float gg = dyp2 ? gv[u] + g2[u] : gv[u];
                    const float pre = (xv[u] - mean) * a + b0;
                    const float dz = pre > 0.0f ? gg : 0.0f;
                    sdz += dz;
                    sdzx += dz * ((xv[u] - mean) * invstd);
                }
        }
    }
    float two[2] = {sdz, sdzx};
    block_sums<SPLIT_NT, 2>(two, scratch);
    if (threadIdx.x == 0) {
        ws[(static_cast<size_t>(c) * p.G + g) * 2 + 0] = two[0];
        ws[(static_cast<size_t>(c) * p.G + g) * 2 + 1] = two[1];
    }
}

// pass 1 (APPLY = false): ws[(c*G + g)*2 + {0,1}] = (sum dz, sum dz*xhat) of the group's images; pass 2 (APPLY = true): dx
template <bool APPLY>
__global__ __launch_bounds__(SPLIT_NT) void bn_pool_bwd_kernel(const float *__restrict__ dyp, const float *__restrict__ dyp2, const uint8_t *__restrict__ code,
                                                               const float *__restrict__ x, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta, const float *__restrict__ save_mean,
                                                               const float *__restrict__ save_invstd, const float *__restrict__ running_mean,
                                                               const float *__restrict__ running_var, float eps, int training,
                                                               float *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                               float *__restrict__ ws, PoolShape p) {
    extern __shared__ __align__(16) float lds[];
    __shared__ float scratch[3 * (SPLIT_NT / 64)];
    const int c = blockIdx.x, g = blockIdx.y;
    const int HW = p.H * p.W, HWq = HW / 4, OHW = p.OH * p.OW, Wq = p.W / 4;
    float *gp = lds;                                            // pooled gradient of one plane
    uint8_t *cd = reinterpret_cast<uint8_t *>(lds + OHW);       // its argmax codes
    const float mean = training ? save_mean[c] : running_mean[c];
    const float invstd = training ? save_invstd[c] : 1.0f / sqrtf(running_var[c] + eps);
    const float a = invstd * (gamma ? gamma[c] : 1.0f), b0 = beta ? beta[c] : 0.0f;
    float m1 = 0.0f, m2 = 0.0f;
    if (APPLY && !p.nosums) {
        __shared__ float parts[2 * SPLIT_MAX];  // the G partial sums, one parallel load (a loop of dependent global loads cost 8 us)
        for (int i = threadIdx.x; i < 2 * p.G; i += SPLIT_NT) parts[i] = ws[static_cast<size_t>(c) * p.G * 2 + i];
        __syncthreads();
        float sdz = 0.0f, sdzx = 0.0f;
#pragma unroll 4
        for (int i = 0; i < p.G; ++i) {
            sdz += parts[2 * i];
            sdzx += parts[2 * i + 1];
        }
        if (g == 0 && threadIdx.x == 0) {
            if (dgamma) dgamma[c] = sdzx;
            if (dbeta) dbeta[c] = sdz;
        }
        if (!dx) return;
        const float n = static_cast<float>(p.B) * static_cast<float>(HW);
        m1 = training ? sdz / n : 0.0f;
        m2 = training ? sdzx / n : 0.0f;
    }
    float sdz = 0.0f, sdzx = 0.0f;
    const int b_end = (g + 1) * p.IPW < p.B ? (g + 1) * p.IPW : p.B;
    for (int b = g * p.IPW; b < b_end; ++b) {
        const size_t pl = static_cast<size_t>(b) * p.C + c;
        for (int o = threadIdx.x; o < OHW; o += SPLIT_NT) {
            gp[o] = dyp2 ? dyp[pl * OHW + o] + dyp2[pl * OHW + o] : dyp[pl * OHW + o];
            cd[o] = code[pl * OHW + o];
        }
        __syncthreads();
        const float4 *x4 = reinterpret_cast<const float4 *>(x) + pl * HWq;
        for (int q = threadIdx.x; q < HWq; q += SPLIT_NT) {
            const float4 v4 = x4[q];
            const int h = q / Wq, w_base = 4 * (q - h * Wq);
            // the 2 x 3 candidate windows of these four pixels (ee_pool.hip: maxpool_bwd_kernel<4>), from LDS
            const int oh0 = h >> 1, ow0 = w_base >> 1;
            const bool row2 = (h & 1) && oh0 + 1 < p.OH;
            float gv[2][3];
            int cv[2][3];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const bool ok = (r == 0 || row2) && ow0 + k < p.OW;
                    const int idx = ok ? (oh0 + r) * p.OW + ow0 + k : 0;
                    gv[r][k] = ok ? gp[idx] : 0.0f;
                    cv[r][k] = ok ? cd[idx] : 255;
                }
            const float xv[4] = {v4.x, v4.y, v4.z, v4.w};
            float out[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int w = w_base + v, k0 = v >> 1;
                float acc = 0.0f;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        if (kb == 1 && !(v & 1)) continue;  // even pixels belong to one window column only
                        const int k = k0 + kb;
                        const int want = (h - ((oh0 + r) * 2 - 1)) * 3 + (w - ((ow0 + k) * 2 - 1));
                        if (cv[r][k] == want) acc += gv[r][k];
                    }
                const float pre = (xv[v] - mean) * a + b0;  // the forward's expression: the same bits, hence the same mask as y > 0
                const float dz = pre > 0.0f ? acc : 0.0f;
                const float xhat = (xv[v] - mean) * invstd;
                if (!APPLY) {
                    sdz += dz;
                    sdzx += dz * xhat;
                } else {
                    out[v] = a * ((dz - m1) - xhat * m2);
                }
            }
            if (APPLY) reinterpret_cast<float4 *>(dx)[pl * HWq + q] = make_float4(out[0], out[1], out[2], out[3]);
        }
        __syncthreads();
    }
    if (!APPLY) {
        {
            float two[2] = {sdz, sdzx};
            block_sums<SPLIT_NT, 2>(two, scratch);
            sdz = two[0], sdzx = two[1];
        }
        if (threadIdx.x == 0) {
            ws[(static_cast<size_t>(c) * p.G + g) * 2 + 0] = sdz;
            ws[(static_cast<size_t>(c) * p.G + g) * 2 + 1] = sdzx;
        }
    }
}

inline bool pool_shape(int B, int C, int H, int W, PoolShape &p) {
    if (B < 1 || C < 1 || H < 1 || W < 4 || W % 4 || static_cast<int64_t>(H) * W > 16000) return false;  // one plane in 64 KB of LDS (ImageNet: 112 x 112)
    if (static_cast<int64_t>(B) * C * H * W / 4 > 0x7fffffffLL) return false;
    const int ipw = (B + SPLIT_MAX - 1) / SPLIT_MAX;
    p = PoolShape{B, C, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1, ipw, (B + ipw - 1) / ipw, 0};
    return true;
}

template <int NT, int VEC>
void launch_fwd(bool relu, bool has_res, hipStream_t st, const float *x, const float *res, const float *gamma, const float *beta, float *rm,
                float *rv, float momentum, float eps, int training, float *y, float *sm, float *si, BnShape s) {
    const dim3 grid(static_cast<unsigned>(s.C)), block(NT);
    if (relu && has_res)
        EE_LAUNCH((bn_fwd_kernel<NT, VEC, true, true>), grid, block, 0, st, x, res, gamma, beta, rm, rv, momentum, eps, training, y, sm, si, s);
    else if (relu)
        EE_LAUNCH((bn_fwd_kernel<NT, VEC, true, false>), grid, block, 0, st, x, res, gamma, beta, rm, rv, momentum, eps, training, y, sm, si, s);
    else if (has_res)
        EE_LAUNCH((bn_fwd_kernel<NT, VEC, false, true>), grid, block, 0, st, x, res, gamma, beta, rm, rv, momentum, eps, training, y, sm, si, s);
    else
        EE_LAUNCH((bn_fwd_kernel<NT, VEC, false, false>), grid, block, 0, st, x, res, gamma, beta, rm, rv, momentum, eps, training, y, sm, si, s);
}

template <int NT, int VEC>
void launch_bwd(bool relu, hipStream_t st, const float *dy, const float *dy2, const float *y, const float *x, const float *beta, const float *gamma, const float *sm, const float *si,
                const float *rm, const float *rv, float eps, int training, float *dx, float *dres, float *dgamma, float *dbeta, BnShape s) {
    const dim3 grid(static_cast<unsigned>(s.C)), block(NT);
    if (relu)
        EE_LAUNCH((bn_bwd_kernel<NT, VEC, true>), grid, block, 0, st, dy, dy2, y, x, beta, gamma, sm, si, rm, rv, eps, training, dx, dres, dgamma, dbeta, s);
    else
        EE_LAUNCH((bn_bwd_kernel<NT, VEC, false>), grid, block, 0, st, dy, dy2, y, x, beta, gamma, sm, si, rm, rv, eps, training, dx, dres, dgamma, dbeta, s);
}


// ---- SyncBatchNorm (ImageNet/experiments_imagenet.py:125, free_imagenet/AT_free_imagenet_ddp.py:149): the batch statistics of a
// data-parallel run are those of the GLOBAL batch.  Per layer and direction the ranks exchange one small tensor (host side,
// torch.distributed over RCCL: eeadv/syncbn.py); the kernels here are the local halves around that exchange, built on the split
// kernels above (any channel size, C x S workgroups):
//   forward   bn_split_stats_kernel -> bn_moments_kernel: this rank's (mean, M2, count) per channel          -> all_gather
//             bn_sync_apply_kernel: Chan's merge of the W ranks' moments in RANK ORDER (the same bits on every rank), running
//             statistics with the global count, y = [relu]((x - mean) * invstd * gamma + beta [+ residual])
//   backward  bn_split_bwd_partial_kernel -> bn_sums_kernel: this rank's (sum dz, sum dz * xhat) per channel -> all_reduce
//             bn_sync_bwd_apply_kernel: dx = gamma * invstd * (dz - SUM dz / N - xhat * SUM dz xhat / N) with the global sums and N;
//             dgamma / dbeta stay this rank's sums (the gradient exchange of the training step averages them like every other
//             parameter gradient - torch's SyncBatchNorm does the same)
__global__ __launch_bounds__(64) void bn_moments_kernel(const float *__restrict__ ws, float *__restrict__ moments, BnShape s, int S) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= s.C) return;
    const int total = s.B * (s.HW / 4);
    const float n = static_cast<float>(s.B) * static_cast<float>(s.HW);
    float sum = 0.0f;
    for (int i = 0; i < S; ++i) sum += ws[(static_cast<size_t>(c) * S + i) * 2];
    const float mean = sum / n;
    float m2 = 0.0f;
    for (int i = 0; i < S; ++i) {  // the same recombination, in the same order, as combine_slices
        const Slice sl = my_slice(total, S, i);
        const float cnt = 4.0f * static_cast<float>(sl.end - sl.begin);
        if (cnt > 0.0f) {
            const float d = ws[(static_cast<size_t>(c) * S + i) * 2] / cnt - mean;
            m2 += ws[(static_cast<size_t>(c) * S + i) * 2 + 1] + cnt * (d * d);
        }
    }
    moments[3 * c + 0] = mean;
    moments[3 * c + 1] = m2;
    moments[3 * c + 2] = n;
}

__global__ __launch_bounds__(64) void bn_sums_kernel(const float *__restrict__ ws, float *__restrict__ sums, int C, int S) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    float a = 0.0f, b = 0.0f;
    for (int i = 0; i < S; ++i) {
        a += ws[(static_cast<size_t>(c) * S + i) * 2];
        b += ws[(static_cast<size_t>(c) * S + i) * 2 + 1];
    }
    sums[2 * c] = a;
    sums[2 * c + 1] = b;
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(SPLIT_NT) void bn_sync_apply_kernel(const float *__restrict__ x, const float *__restrict__ res, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, const float *__restrict__ all_moments, int W,
                                                                 float *running_mean, float *running_var, float momentum, float eps,
                                                                 float *__restrict__ y, float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                                 BnShape s, int S) {
    const int c = blockIdx.x, sl_i = blockIdx.y;
    Moments t{0.0f, 0.0f, 0.0f};
    for (int r = 0; r < W; ++r) {  // every lane of every workgroup on every rank: the same merge in the same order
        const float *m = all_moments + (static_cast<size_t>(r) * s.C + c) * 3;
        t = merge(t, Moments{m[2], m[0], m[1]});
    }
    const float var = t.n > 0.0f ? t.m2 / t.n : 0.0f;
    const float mean = t.mean, invstd = 1.0f / sqrtf(var + eps);
    if (sl_i == 0 && threadIdx.x == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        if (running_mean) {
            const float unbiased = (t.n > 1.0f) ? var * (t.n / (t.n - 1.0f)) : var;
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
        }
    }
    const float a = invstd * (gamma ? gamma[c] : 1.0f), b0 = beta ? beta[c] : 0.0f;
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *r4 = reinterpret_cast<const float4 *>(res);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    struct XR {
        float4 x, r;
    };
    for_slice(s, c, my_slice(s.B * (s.HW / 4), S, sl_i), [&](size_t o) { return XR{x4[o], RES ? r4[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f)}; },
              [&](XR in, size_t o) {
        const float4 v = in.x;
        float4 r = make_float4((v.x - mean) * a + b0, (v.y - mean) * a + b0, (v.z - mean) * a + b0, (v.w - mean) * a + b0);
        if (RES) {
            const float4 q = in.r;
            r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
        }
        if (RELU) {
            r.x = relu_nan(r.x); r.y = relu_nan(r.y); r.z = relu_nan(r.z); r.w = relu_nan(r.w);
        }
        y4[o] = r;
    });
}

template <bool RELU>
__global__ __launch_bounds__(SPLIT_NT) void bn_sync_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ dy2, const float *__restrict__ y,
                                                                     const float *__restrict__ x, const float *__restrict__ beta, const float *__restrict__ gamma,
                                                                     const float *__restrict__ save_mean, const float *__restrict__ save_invstd,
                                                                     const float *__restrict__ global_sums, float n_global, float *__restrict__ dx,
                                                                     float *__restrict__ dres, BnShape s, int S) {
    const int c = blockIdx.x, sl_i = blockIdx.y;
    const float mean = save_mean[c], invstd = save_invstd[c];
    const float w = (gamma ? gamma[c] : 1.0f) * invstd;
    const float m1 = global_sums[2 * c] / n_global, m2 = global_sums[2 * c + 1] / n_global;
    const float4 *dy4 = reinterpret_cast<const float4 *>(dy), *y4 = reinterpret_cast<const float4 *>(y), *x4 = reinterpret_cast<const float4 *>(x);
    float4 *dx4 = reinterpret_cast<float4 *>(dx), *dr4 = reinterpret_cast<float4 *>(dres);
    const bool want_x = dx != nullptr;
    const MaskArgs mk = mask_args(y, gamma, beta, invstd, c);
    for_slice(s, c, my_slice(s.B * (s.HW / 4), S, sl_i), [&](size_t o) { return bwd_load(dy4, reinterpret_cast<const float4 *>(dy2), y4, x4, o, RELU, want_x, mean, mk); }, [&](BwdIn in, size_t o) {
        const float4 g = masked_dz(in, RELU);
        if (dres) dr4[o] = g;
        if (dx) {
            const float4 v = in.x;
            dx4[o] = make_float4(w * ((g.x - m1) - ((v.x - mean) * invstd) * m2), w * ((g.y - m1) - ((v.y - mean) * invstd) * m2),
                                 w * ((g.z - m1) - ((v.z - mean) * invstd) * m2), w * ((g.w - m1) - ((v.w - mean) * invstd) * m2));
        }
    });
}

inline bool al16(const void *q) { return !q || aligned16(q); }

}  // namespace

// floats of workspace the split path wants for this shape; 0 = the shape runs in one workgroup per channel
EE_API int ee_bn_workspace_floats(int B, int C, int HW) {
    if (B < 1 || C < 1 || HW < 1 || HW % 4) return 0;
    const int64_t quads = static_cast<int64_t>(B) * (HW / 4);
    if (cached_variant(quads, static_cast<int64_t>(B) * C * HW) || static_cast<int64_t>(B) * C * HW / 4 > 0x7fffffffLL) return 0;
    return C * split_slices(quads) * 2;
}

EE_API int ee_bn_act_fwd_f32(const float *x, const float *residual, const float *gamma, const float *beta, float *running_mean,
                             float *running_var, float momentum, float eps, int training, int relu, float *y, float *save_mean,
                             float *save_invstd, float *workspace, int B, int C, int HW, void *stream) {
    if (B < 0 || C < 1 || HW < 1) return EE_ERR_SHAPE;
    if (static_cast<int64_t>(B) * HW > 0x7fffffffLL) return EE_ERR_SHAPE;
    if (B == 0) return EE_OK;
    if (!x || !y) return EE_ERR_NULL;
    if (training && (!save_mean || !save_invstd)) return EE_ERR_NULL;
    if (!training && (!running_mean || !running_var)) return EE_ERR_NULL;
    const BnShape s{B, C, HW};
    const bool vec = (HW % 4 == 0) && al16(x) && al16(y) && al16(residual);
    const bool big = static_cast<int64_t>(B) * HW >= 16384;
    hipStream_t st = as_stream(stream);
    const int cv = vec ? cached_variant(static_cast<int64_t>(B) * (HW / 4), static_cast<int64_t>(B) * C * HW) : 0;
    if (cv == 1) launch_fwd_cached<256, 2>(relu != 0, residual != nullptr, st, x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, y, save_mean, save_invstd, s);
    else if (cv == 2) launch_fwd_cached<256, 7>(relu != 0, residual != nullptr, st, x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, y, save_mean, save_invstd, s);
    else if (cv == 3) launch_fwd_cached<1024, 7>(relu != 0, residual != nullptr, st, x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, y, save_mean, save_invstd, s);
    else if (vec && workspace && static_cast<int64_t>(B) * C * HW / 4 <= 0x7fffffffLL) {
        const int S = split_slices(static_cast<int64_t>(B) * (HW / 4));
        const dim3 grid(static_cast<unsigned>(C), static_cast<unsigned>(S)), block(SPLIT_NT);
        if (training) EE_LAUNCH(bn_split_stats_kernel, grid, block, 0, st, x, workspace, s, S);
        if (relu && residual)
            EE_LAUNCH((bn_split_apply_kernel<true, true>), grid, block, 0, st, x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, y, save_mean, save_invstd, workspace, s, S);
        else if (relu)
            EE_LAUNCH((bn_split_apply_kernel<true, false>), grid, block, 0, st, x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, y, save_mean, save_invstd, workspace, s, S);
        else if (residual)
            EE_LAUNCH((bn_split_apply_kernel<false, true>), grid, block, 0, st, x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, y, save_mean, save_invstd, workspace, s, S);
        else
            EE_LAUNCH((bn_split_apply_kernel<false, false>), grid, block, 0, st, x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, y, save_mean, save_invstd, workspace, s, S);
    } else if (vec && big) launch_fwd<1024, 4>(relu != 0, residual != nullptr, st, x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, y, save_mean, save_invstd, s);
    else if (vec) launch_fwd<256, 4>(relu != 0, residual != nullptr, st, x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, y, save_mean, save_invstd, s);
    else if (big) launch_fwd<1024, 1>(relu != 0, residual != nullptr, st, x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, y, save_mean, save_invstd, s);
    else launch_fwd<256, 1>(relu != 0, residual != nullptr, st, x, residual, gamma, beta, running_mean, running_var, momentum, eps, training, y, save_mean, save_invstd, s);
    return launch_status();
}

EE_API int ee_bn_act_bwd2_f32(const float *dy, const float *dy2, const float *y, const float *x, const float *gamma, const float *beta, const float *save_mean,
                             const float *save_invstd, const float *running_mean, const float *running_var, float eps, int training, int relu,
                             float *dx, float *dresidual, float *dgamma, float *dbeta, float *workspace, int B, int C, int HW, void *stream) {
    if (B < 0 || C < 1 || HW < 1) return EE_ERR_SHAPE;
    if (static_cast<int64_t>(B) * HW > 0x7fffffffLL) return EE_ERR_SHAPE;
    if (B == 0) return EE_OK;
    if (!dy || !x) return EE_ERR_NULL;  // y == NULL with relu: the mask is recomputed from x, gamma, beta (no residual branch: the caller's promise)
    if (training && (!save_mean || !save_invstd)) return EE_ERR_NULL;
    if (!training && (!running_mean || !running_var)) return EE_ERR_NULL;
    const BnShape s{B, C, HW};
    const bool vec = (HW % 4 == 0) && al16(dy) && al16(dy2) && al16(y) && al16(x) && al16(dx) && al16(dresidual);
    const bool big = static_cast<int64_t>(B) * HW >= 16384;
    hipStream_t st = as_stream(stream);
    const int cv = vec ? cached_variant(static_cast<int64_t>(B) * (HW / 4), static_cast<int64_t>(B) * C * HW) : 0;
    if (cv == 1) launch_bwd_cached<256, 2>(relu != 0, st, dy, dy2, y, x, beta, gamma, save_mean, save_invstd, running_mean, running_var, eps, training, dx, dresidual, dgamma, dbeta, s);
    else if (cv == 2) launch_bwd_cached<256, 7>(relu != 0, st, dy, dy2, y, x, beta, gamma, save_mean, save_invstd, running_mean, running_var, eps, training, dx, dresidual, dgamma, dbeta, s);
    else if (cv == 3) launch_bwd_cached<1024, 7>(relu != 0, st, dy, dy2, y, x, beta, gamma, save_mean, save_invstd, running_mean, running_var, eps, training, dx, dresidual, dgamma, dbeta, s);
    else if (vec && workspace && static_cast<int64_t>(B) * C * HW / 4 <= 0x7fffffffLL) {
        const int S = split_slices(static_cast<int64_t>(B) * (HW / 4));
        const dim3 grid(static_cast<unsigned>(C), static_cast<unsigned>(S)), block(SPLIT_NT);
        if (relu) {
            EE_LAUNCH((bn_split_bwd_partial_kernel<true>), grid, block, 0, st, dy, dy2, y, x, gamma, beta, save_mean, save_invstd, running_mean, running_var, eps, training, workspace, s, S);
            EE_LAUNCH((bn_split_bwd_apply_kernel<true>), grid, block, 0, st, dy, dy2, y, x, beta, gamma, save_mean, save_invstd, running_mean, running_var, eps, training, dx, dresidual, dgamma, dbeta, workspace, s, S);
        } else {
            EE_LAUNCH((bn_split_bwd_partial_kernel<false>), grid, block, 0, st, dy, dy2, y, x, gamma, beta, save_mean, save_invstd, running_mean, running_var, eps, training, workspace, s, S);
            EE_LAUNCH((bn_split_bwd_apply_kernel<false>), grid, block, 0, st, dy, dy2, y, x, beta, gamma, save_mean, save_invstd, running_mean, running_var, eps, training, dx, dresidual, dgamma, dbeta, workspace, s, S);
        }
    } else if (vec && big) launch_bwd<1024, 4>(relu != 0, st, dy, dy2, y, x, beta, gamma, save_mean, save_invstd, running_mean, running_var, eps, training, dx, dresidual, dgamma, dbeta, s);
    else if (vec) launch_bwd<256, 4>(relu != 0, st, dy, dy2, y, x, beta, gamma, save_mean, save_invstd, running_mean, running_var, eps, training, dx, dresidual, dgamma, dbeta, s);
    else if (big) launch_bwd<1024, 1>(relu != 0, st, dy, dy2, y, x, beta, gamma, save_mean, save_invstd, running_mean, running_var, eps, training, dx, dresidual, dgamma, dbeta, s);
    else launch_bwd<256, 1>(relu != 0, st, dy, dy2, y, x, beta, gamma, save_mean, save_invstd, running_mean, running_var, eps, training, dx, dresidual, dgamma, dbeta, s);
    return launch_status();
}

EE_API int ee_bn_act_bwd_f32(const float *dy, const float *y, const float *x, const float *gamma, const float *save_mean,
                             const float *save_invstd, const float *running_mean, const float *running_var, float eps, int training, int relu,
                             float *dx, float *dresidual, float *dgamma, float *dbeta, float *workspace, int B, int C, int HW, void *stream) {
    if (relu && !y) return EE_ERR_NULL;
    return ee_bn_act_bwd2_f32(dy, nullptr, y, x, gamma, nullptr, save_mean, save_invstd, running_mean, running_var, eps, training, relu, dx, dresidual, dgamma,
                              dbeta, workspace, B, C, HW, stream);
}

// floats of workspace ee_bn_relu_pool_fwd/bwd_f32 want (statistics partials forward, gradient-sum partials backward); 0 = unsupported shape
EE_API int ee_bn_relu_pool_workspace_floats(int B, int C, int H, int W) {
    PoolShape p;
    if (!pool_shape(B, C, H, W, p)) return 0;
    return C * SPLIT_MAX * 2;
}

static int bn_relu_pool_fwd_impl(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                                 float momentum, float eps, int training, float *y_pool, uint8_t *code, float *x_argmax, float *save_mean,
                                 float *save_invstd, float *workspace, const float *conv_stats, int conv_stats_slices, int B, int C, int H,
                                 int W, void *stream) {
    if (B < 0 || C < 1 || H < 1 || W < 1) return EE_ERR_SHAPE;
    if (B == 0) return EE_OK;
    PoolShape p;
    if (!pool_shape(B, C, H, W, p)) return EE_ERR_UNSUPPORTED;
    if (!x || !y_pool || !code || !workspace) return EE_ERR_NULL;
    if (training && (!save_mean || !save_invstd)) return EE_ERR_NULL;
    if (!training && (!running_mean || !running_var)) return EE_ERR_NULL;
    if (!aligned16(x)) return EE_ERR_ALIGN;
    hipStream_t st = as_stream(stream);
    const BnShape s{B, C, H * W};
    const int S = split_slices(static_cast<int64_t>(B) * (H * W / 4));
    int mode = training ? 1 : 0;
    const float *wsk = workspace;
    int Sk = S;
    if (training && conv_stats && conv_stats_slices > 0 && conv_stats_slices <= POOL_MERGE_MAX * SPLIT_NT) {
        mode = 3, wsk = conv_stats, Sk = conv_stats_slices;  // merged inside bn_pool_fwd_kernel
    } else if (training && conv_stats && conv_stats_slices > 0) {
        EE_LAUNCH(bn_stats_finalize_kernel, dim3(static_cast<unsigned>(C)), dim3(SPLIT_NT), 0, st, conv_stats, conv_stats_slices, eps, momentum, running_mean,
                  running_var, save_mean, save_invstd);
        mode = 2;
    } else if (training) {
        EE_LAUNCH(bn_split_stats_kernel, dim3(static_cast<unsigned>(C), static_cast<unsigned>(S)), dim3(SPLIT_NT), 0, st, x, workspace, s, S);
    }
    EE_LAUNCH(bn_pool_fwd_kernel, dim3(static_cast<unsigned>(C), static_cast<unsigned>(p.G)), dim3(SPLIT_NT), static_cast<size_t>(H) * W * sizeof(float), st,
              x, gamma, beta, running_mean, running_var, momentum, eps, mode, y_pool, code, save_mean, save_invstd, wsk, p, Sk, x_argmax);
    return launch_status();
}

EE_API int ee_bn_relu_pool_fwd_f32(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                                   float momentum, float eps, int training, float *y_pool, uint8_t *code, float *save_mean,
                                   float *save_invstd, float *workspace, const float *conv_stats, int conv_stats_slices, int B, int C, int H,
                                   int W, void *stream) {
    return bn_relu_pool_fwd_impl(x, gamma, beta, running_mean, running_var, momentum, eps, training, y_pool, code, nullptr, save_mean, save_invstd, workspace,
                                 conv_stats, conv_stats_slices, B, C, H, W, stream);
}

// ... also writing x_argmax [B,C,OH,OW] = x at every window's argmax: ee_bn_relu_pool_bwd_xa_f32 then takes its batch sums from the pooled
// tensors alone (training mode: one pass over the 26 MB map less)
EE_API int ee_bn_relu_pool_fwd_xa_f32(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                                      float momentum, float eps, int training, float *y_pool, uint8_t *code, float *x_argmax, float *save_mean,
                                      float *save_invstd, float *workspace, const float *conv_stats, int conv_stats_slices, int B, int C, int H,
                                      int W, void *stream) {
    if (B > 0 && !x_argmax) return EE_ERR_NULL;
    return bn_relu_pool_fwd_impl(x, gamma, beta, running_mean, running_var, momentum, eps, training, y_pool, code, x_argmax, save_mean, save_invstd, workspace,
                                 conv_stats, conv_stats_slices, B, C, H, W, stream);
}

static int bn_relu_pool_bwd_impl(const float *dy_pool, const float *dy_pool2, const uint8_t *code, const float *x, const float *x_argmax, const float *gamma,
                                 const float *beta, const float *save_mean, const float *save_invstd, const float *running_mean, const float *running_var,
                                 float eps, int training, float *dx, float *dgamma, float *dbeta, float *workspace, int B, int C, int H,
                                 int W, void *stream) {
    if (B < 0 || C < 1 || H < 1 || W < 1) return EE_ERR_SHAPE;
    if (B == 0) return EE_OK;
    PoolShape p;
    if (!pool_shape(B, C, H, W, p)) return EE_ERR_UNSUPPORTED;
    if (!dy_pool || !code || !x || !workspace) return EE_ERR_NULL;
    if (training && (!save_mean || !save_invstd)) return EE_ERR_NULL;
    if (!training && (!running_mean || !running_var)) return EE_ERR_NULL;
    if (!aligned16(x) || (dx && !aligned16(dx))) return EE_ERR_ALIGN;
    hipStream_t st = as_stream(stream);
    const dim3 grid(static_cast<unsigned>(C), static_cast<unsigned>(p.G)), block(SPLIT_NT);
    const size_t lds = (static_cast<size_t>(p.OH) * p.OW * 5 + 15) / 16 * 16;  // floats + bytes
    // eval mode (running statistics) and no parameter gradients wanted - every iteration of an eval-mode attack: dx = a * dz needs no sums
    p.nosums = (!training && !dgamma && !dbeta) ? 1 : 0;
    if (p.nosums && !dx) return EE_OK;
    if (!p.nosums && training && x_argmax)
        EE_LAUNCH(bn_pool_bwd_sums_pooled_kernel, grid, block, 0, st, dy_pool, dy_pool2, x_argmax, gamma, beta, save_mean, save_invstd, workspace, p);
    else if (!p.nosums)
        EE_LAUNCH((bn_pool_bwd_kernel<false>), grid, block, lds, st, dy_pool, dy_pool2, code, x, gamma, beta, save_mean, save_invstd, running_mean, running_var, eps,
                  training, dx, dgamma, dbeta, workspace, p);
    EE_LAUNCH((bn_pool_bwd_kernel<true>), grid, block, lds, st, dy_pool, dy_pool2, code, x, gamma, beta, save_mean, save_invstd, running_mean, running_var, eps,
              training, dx, dgamma, dbeta, workspace, p);
    return launch_status();
}

EE_API int ee_bn_relu_pool_bwd_f32(const float *dy_pool, const float *dy_pool2, const uint8_t *code, const float *x, const float *gamma, const float *beta,
                                   const float *save_mean, const float *save_invstd, const float *running_mean, const float *running_var,
                                   float eps, int training, float *dx, float *dgamma, float *dbeta, float *workspace, int B, int C, int H,
                                   int W, void *stream) {
    return bn_relu_pool_bwd_impl(dy_pool, dy_pool2, code, x, nullptr, gamma, beta, save_mean, save_invstd, running_mean, running_var, eps, training, dx, dgamma,
                                 dbeta, workspace, B, C, H, W, stream);
}

// ... with x_argmax (ee_bn_relu_pool_fwd_xa_f32; nullable = the form above): training mode takes the batch sums from dy_pool and x_argmax
EE_API int ee_bn_relu_pool_bwd_xa_f32(const float *dy_pool, const float *dy_pool2, const uint8_t *code, const float *x, const float *x_argmax, const float *gamma,
                                      const float *beta, const float *save_mean, const float *save_invstd, const float *running_mean,
                                      const float *running_var, float eps, int training, float *dx, float *dgamma, float *dbeta, float *workspace, int B,
                                      int C, int H, int W, void *stream) {
    return bn_relu_pool_bwd_impl(dy_pool, dy_pool2, code, x, x_argmax, gamma, beta, save_mean, save_invstd, running_mean, running_var, eps, training, dx, dgamma,
                                 dbeta, workspace, B, C, H, W, stream);
}

// 1 when relu(bn_a(xa) + bn_b(xb)) runs as one launch each way (the register-cached variants with 256-lane workgroups), else 0
EE_API int ee_bn_dual_supported(int B, int C, int HW) {
    if (B < 1 || C < 1 || HW < 1 || HW % 4) return 0;
    const int cv = cached_variant(static_cast<int64_t>(B) * (HW / 4), static_cast<int64_t>(B) * C * HW);
    return (cv == 1 || cv == 2) ? 1 : 0;
}

EE_API int ee_bn_dual_fwd_f32(const float *xa, const float *xb, const float *gamma_a, const float *beta_a, float *running_mean_a, float *running_var_a,
                              float momentum_a, float eps_a, float *save_mean_a, float *save_invstd_a, const float *gamma_b, const float *beta_b,
                              float *running_mean_b, float *running_var_b, float momentum_b, float eps_b, float *save_mean_b, float *save_invstd_b,
                              int training, float *y, int B, int C, int HW, void *stream) {
    if (B < 0 || C < 1 || HW < 1) return EE_ERR_SHAPE;
    if (B == 0) return EE_OK;
    if (!ee_bn_dual_supported(B, C, HW)) return EE_ERR_UNSUPPORTED;
    if (!xa || !xb || !y) return EE_ERR_NULL;
    if (training && (!save_mean_a || !save_invstd_a || !save_mean_b || !save_invstd_b)) return EE_ERR_NULL;
    if (!training && (!running_mean_a || !running_var_a || !running_mean_b || !running_var_b)) return EE_ERR_NULL;
    if (!aligned16(xa) || !aligned16(xb) || !aligned16(y)) return EE_ERR_ALIGN;
    const BnShape s{B, C, HW};
    const BnParams pa{gamma_a, beta_a, running_mean_a, running_var_a, momentum_a, eps_a, save_mean_a, save_invstd_a};
    const BnParams pb{gamma_b, beta_b, running_mean_b, running_var_b, momentum_b, eps_b, save_mean_b, save_invstd_b};
    const dim3 grid(static_cast<unsigned>(C)), block(256);
    if (cached_variant(static_cast<int64_t>(B) * (HW / 4), static_cast<int64_t>(B) * C * HW) == 1)
        EE_LAUNCH((bn_dual_fwd_cached_kernel<256, 2>), grid, block, 0, as_stream(stream), xa, xb, pa, pb, training, y, s);
    else
        EE_LAUNCH((bn_dual_fwd_cached_kernel<256, 7>), grid, block, 0, as_stream(stream), xa, xb, pa, pb, training, y, s);
    return launch_status();
}

EE_API int ee_bn_dual_bwd_f32(const float *dy, const float *dy2, const float *y, const float *xa, const float *xb, const float *gamma_a,
                              const float *save_mean_a, const float *save_invstd_a, const float *running_mean_a, const float *running_var_a,
                              float eps_a, const float *gamma_b, const float *save_mean_b, const float *save_invstd_b,
                              const float *running_mean_b, const float *running_var_b, float eps_b, int training, float *dxa, float *dxb,
                              float *dgamma_a, float *dbeta_a, float *dgamma_b, float *dbeta_b, int B, int C, int HW, void *stream) {
    if (B < 0 || C < 1 || HW < 1) return EE_ERR_SHAPE;
    if (B == 0) return EE_OK;
    if (!ee_bn_dual_supported(B, C, HW)) return EE_ERR_UNSUPPORTED;
    if (!dy || !y || !xa || !xb) return EE_ERR_NULL;
    if (training && (!save_mean_a || !save_invstd_a || !save_mean_b || !save_invstd_b)) return EE_ERR_NULL;
    if (!training && (!running_mean_a || !running_var_a || !running_mean_b || !running_var_b)) return EE_ERR_NULL;
    if (!aligned16(dy) || !al16(dy2) || !aligned16(y) || !aligned16(xa) || !aligned16(xb) || !al16(dxa) || !al16(dxb)) return EE_ERR_ALIGN;
    const BnShape s{B, C, HW};
    const BnParams pa{gamma_a, nullptr, const_cast<float *>(running_mean_a), const_cast<float *>(running_var_a), 0.0f, eps_a, const_cast<float *>(save_mean_a),
                      const_cast<float *>(save_invstd_a)};
    const BnParams pb{gamma_b, nullptr, const_cast<float *>(running_mean_b), const_cast<float *>(running_var_b), 0.0f, eps_b, const_cast<float *>(save_mean_b),
                      const_cast<float *>(save_invstd_b)};
    const dim3 grid(static_cast<unsigned>(C)), block(256);
    if (cached_variant(static_cast<int64_t>(B) * (HW / 4), static_cast<int64_t>(B) * C * HW) == 1)
        EE_LAUNCH((bn_dual_bwd_cached_kernel<256, 2>), grid, block, 0, as_stream(stream), dy, dy2, y, xa, xb, pa, pb, training, dxa, dxb, dgamma_a, dbeta_a,
                  dgamma_b, dbeta_b, s);
    else
        EE_LAUNCH((bn_dual_bwd_cached_kernel<256, 7>), grid, block, 0, as_stream(stream), dy, dy2, y, xa, xb, pa, pb, training, dxa, dxb, dgamma_a, dbeta_a,
                  dgamma_b, dbeta_b, s);
    return launch_status();
}

// ---- SyncBatchNorm: the local halves around the two exchanges (see the kernels above; eeadv/syncbn.py does the collectives) ----------
// workspace: ee_syncbn_workspace_floats(B, C, HW) floats.  moments [C][3] = (mean, M2, count) of this rank's batch.
EE_API int ee_syncbn_workspace_floats(int B, int C, int HW) {
    if (B < 1 || C < 1 || HW < 1 || HW % 4) return 0;
    return C * split_slices(static_cast<int64_t>(B) * (HW / 4)) * 2;
}

static int syncbn_check(int B, int C, int HW) {
    if (B < 1 || C < 1 || HW < 1) return EE_ERR_SHAPE;
    if (HW % 4) return EE_ERR_UNSUPPORTED;
    if (static_cast<int64_t>(B) * C * HW / 4 > 0x7fffffffLL) return EE_ERR_SHAPE;
    return EE_OK;
}

EE_API int ee_syncbn_stats_f32(const float *x, float *workspace, float *moments, int B, int C, int HW, void *stream) {
    if (const int e = syncbn_check(B, C, HW)) return e;
    if (!x || !workspace || !moments) return EE_ERR_NULL;
    if (!aligned16(x)) return EE_ERR_ALIGN;
    const BnShape s{B, C, HW};
    const int S = split_slices(static_cast<int64_t>(B) * (HW / 4));
    hipStream_t st = as_stream(stream);
    EE_LAUNCH(bn_split_stats_kernel, dim3(static_cast<unsigned>(C), static_cast<unsigned>(S)), dim3(SPLIT_NT), 0, st, x, workspace, s, S);
    EE_LAUNCH(bn_moments_kernel, dim3(static_cast<unsigned>((C + 63) / 64)), dim3(64), 0, st, workspace, moments, s, S);
    return launch_status();
}

// all_moments [W][C][3]: every rank's moments in rank order (all_gather).  Writes y, save_mean / save_invstd (the GLOBAL batch's) and
// updates running_* (nullable) with the global count.
EE_API int ee_syncbn_apply_f32(const float *x, const float *residual, const float *gamma, const float *beta, const float *all_moments, int W,
                               float *running_mean, float *running_var, float momentum, float eps, int relu, float *y, float *save_mean,
                               float *save_invstd, int B, int C, int HW, void *stream) {
    if (const int e = syncbn_check(B, C, HW)) return e;
    if (W < 1) return EE_ERR_SHAPE;
    if (!x || !y || !all_moments || !save_mean || !save_invstd) return EE_ERR_NULL;
    if (!aligned16(x) || !aligned16(y) || !al16(residual)) return EE_ERR_ALIGN;
    const BnShape s{B, C, HW};
    const int S = split_slices(static_cast<int64_t>(B) * (HW / 4));
    const dim3 grid(static_cast<unsigned>(C), static_cast<unsigned>(S)), block(SPLIT_NT);
    hipStream_t st = as_stream(stream);
    if (relu && residual)
        EE_LAUNCH((bn_sync_apply_kernel<true, true>), grid, block, 0, st, x, residual, gamma, beta, all_moments, W, running_mean, running_var, momentum, eps, y, save_mean, save_invstd, s, S);
    else if (relu)
        EE_LAUNCH((bn_sync_apply_kernel<true, false>), grid, block, 0, st, x, residual, gamma, beta, all_moments, W, running_mean, running_var, momentum, eps, y, save_mean, save_invstd, s, S);
    else if (residual)
        EE_LAUNCH((bn_sync_apply_kernel<false, true>), grid, block, 0, st, x, residual, gamma, beta, all_moments, W, running_mean, running_var, momentum, eps, y, save_mean, save_invstd, s, S);
    else
        EE_LAUNCH((bn_sync_apply_kernel<false, false>), grid, block, 0, st, x, residual, gamma, beta, all_moments, W, running_mean, running_var, momentum, eps, y, save_mean, save_invstd, s, S);
    return launch_status();
}

// sums [C][2] = this rank's (sum dz, sum dz * xhat), dz = relu ? (dy + dy2) * (y > 0) : dy + dy2 (dy2, y nullable as in ee_bn_act_bwd2_f32)
EE_API int ee_syncbn_bwd_sums_f32(const float *dy, const float *dy2, const float *y, const float *x, const float *gamma, const float *beta,
                                  const float *save_mean, const float *save_invstd, int relu, float *workspace, float *sums, int B, int C, int HW,
                                  void *stream) {
    if (const int e = syncbn_check(B, C, HW)) return e;
    if (!dy || !x || !save_mean || !save_invstd || !workspace || !sums) return EE_ERR_NULL;
    if (!aligned16(dy) || !aligned16(x) || !al16(dy2) || !al16(y)) return EE_ERR_ALIGN;
    const BnShape s{B, C, HW};
    const int S = split_slices(static_cast<int64_t>(B) * (HW / 4));
    const dim3 grid(static_cast<unsigned>(C), static_cast<unsigned>(S)), block(SPLIT_NT);
    hipStream_t st = as_stream(stream);
    if (relu)
        EE_LAUNCH((bn_split_bwd_partial_kernel<true>), grid, block, 0, st, dy, dy2, y, x, gamma, beta, save_mean, save_invstd, nullptr, nullptr, 0.0f, 1, workspace, s, S);
    else
        EE_LAUNCH((bn_split_bwd_partial_kernel<false>), grid, block, 0, st, dy, dy2, y, x, gamma, beta, save_mean, save_invstd, nullptr, nullptr, 0.0f, 1, workspace, s, S);
    EE_LAUNCH(bn_sums_kernel, dim3(static_cast<unsigned>((C + 63) / 64)), dim3(64), 0, st, workspace, sums, C, S);
    return launch_status();
}

// global_sums [C][2]: the all-reduced sums; n_global: elements per channel over all ranks.  dx and / or dresidual (= dz), nullable.
EE_API int ee_syncbn_bwd_apply_f32(const float *dy, const float *dy2, const float *y, const float *x, const float *gamma, const float *beta,
                                   const float *save_mean, const float *save_invstd, const float *global_sums, double n_global, int relu, float *dx,
                                   float *dresidual, int B, int C, int HW, void *stream) {
    if (const int e = syncbn_check(B, C, HW)) return e;
    if (!dy || !x || !save_mean || !save_invstd || !global_sums) return EE_ERR_NULL;
    if (!(n_global >= 1.0)) return EE_ERR_SHAPE;
    if (!aligned16(dy) || !aligned16(x) || !al16(dy2) || !al16(y) || !al16(dx) || !al16(dresidual)) return EE_ERR_ALIGN;
    if (!dx && !dresidual) return EE_OK;
    const BnShape s{B, C, HW};
    const int S = split_slices(static_cast<int64_t>(B) * (HW / 4));
    const dim3 grid(static_cast<unsigned>(C), static_cast<unsigned>(S)), block(SPLIT_NT);
    hipStream_t st = as_stream(stream);
    if (relu)
        EE_LAUNCH((bn_sync_bwd_apply_kernel<true>), grid, block, 0, st, dy, dy2, y, x, beta, gamma, save_mean, save_invstd, global_sums, static_cast<float>(n_global), dx, dresidual, s, S);
    else
        EE_LAUNCH((bn_sync_bwd_apply_kernel<false>), grid, block, 0, st, dy, dy2, y, x, beta, gamma, save_mean, save_invstd, global_sums, static_cast<float>(n_global), dx, dresidual, s, S);
    return launch_status();
}
