// ee_fuse.hpp - eval-mode BatchNorm folded into the convolution kernels (ee_wino.hip, ee_s2.hip, ee_conv.hip, ee_dense.hip).
//
// Every validate() pass (PGD-10/50/100, experiments_tinyimagenet.py:337,354-358) and the inner loops of TRADES / ALP (utils/attacks.py:249,
// :405) run the classifier with model.eval(): BatchNorm then reads its RUNNING statistics, i.e. it is the per-channel constant map
// y = (x - mean) * (invstd * gamma) + beta.  Nothing crosses workgroups, so it needs no launch of its own:
//   forward   the convolution's output transform applies it, adds the block's residual and the ReLU (FusePost) before the only store;
//   backward  the backward-data convolution applies dz = (y > 0) * (dy [+ dy2]),  x = gamma * invstd * dz  while it stages its input
//             (FusePre), and one workgroup per image writes dz out as the residual branch's gradient.
// The expressions and their order are those of ee_bn.hip's kernels in eval mode (bn_fwd_cached_kernel / bn_bwd_cached_kernel with
// training = 0; -ffp-contract=off), so the fused path gives the bits of the unfused one.
//
// CNN-body glue, not a row of SURVEY.md section 8.
#pragma once
#include "ee_common.hpp"

namespace ee {

// what happens to a convolution's input while it is staged: v = in [+ add];  v = mask > 0 ? v : 0;  store <- v;  v = (gamma / sqrt(var + eps))[c] * v
struct FusePre {
    const float *add;    // second piece of the gradient of a forked block output (same shape as the input), or null
    const float *mask;   // the forward pass's ReLU output (same shape), or null: no mask
    float *store;        // receives the masked sum (the gradient of the residual branch), or null
    const float *var, *gamma;  // [reduction channels]; gamma null: 1
    float eps;
};

// what happens to a convolution's output before it is stored: r = (c - mean) * (invstd * gamma) + beta;  r += res;  r = relu(r)
struct FusePost {
    const float *mean, *var, *gamma, *beta;  // [result channels]; gamma / beta null: 1 / 0;  mean null: no affine map (r = c)
    float eps;
    const float *res;    // added to the result (same shape as the output), or null.  Forward: the block's residual; backward-data: the
                         // gradient that reaches the block's input through its identity branch (one summed gradient leaves the block)
    int relu;
};

// invstd exactly as ee_bn.hip computes it in eval mode
__device__ __forceinline__ float bn_invstd(float var, float eps) { return 1.0f / sqrtf(var + eps); }
// bn_fwd_cached_kernel: a = invstd * gamma;  bn_bwd_cached_kernel: w = gamma * invstd - one product, commutative: the same bits
__device__ __forceinline__ float bn_scale(const float *var, const float *gamma, float eps, int c) {
    return bn_invstd(var[c], eps) * (gamma ? gamma[c] : 1.0f);
}
// ReLU that keeps NaN (ee_bn.hip: relu_nan)
__device__ __forceinline__ float relu_keep_nan(float r) { return r > 0.0f ? r : (r != r ? r : 0.0f); }

struct PostConst {  // a lane's constants for its result channel
    float mean, a, b;
};
__device__ __forceinline__ PostConst post_const(const FusePost &p, int c) {
    if (!p.mean) return PostConst{0.0f, 1.0f, 0.0f};
    return PostConst{p.mean[c], bn_scale(p.var, p.gamma, p.eps, c), p.beta ? p.beta[c] : 0.0f};
}
__device__ __forceinline__ float post_apply(float c, const PostConst &k) { return (c - k.mean) * k.a + k.b; }

__device__ __forceinline__ float4 mask4(float4 v, float4 m) {
    return make_float4(m.x > 0.0f ? v.x : 0.0f, m.y > 0.0f ? v.y : 0.0f, m.z > 0.0f ? v.z : 0.0f, m.w > 0.0f ? v.w : 0.0f);
}
__device__ __forceinline__ float4 sum4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 scale4(float w, float4 v) { return make_float4(w * v.x, w * v.y, w * v.z, w * v.w); }

// ---- TRAIN-mode BatchNorm across the kernel boundary (round 4) ------------------------------------------------------------------------
// The convolution in front of a mid-block BatchNorm (resnet.py:44-49: conv1 -> bn1 -> relu -> conv2) writes, next to its raw output, the
// moments of what each workgroup holds: stats_out [channels][S][2] = (mean, M2) of `cnt` values each (equal counts).  The convolution
// behind it merges the S partials of every reduction channel in its prologue - each of 16 lanes its share in index order, then a fixed
// butterfly: mean = sum mean_s / S, M2 = sum (M2_s + cnt * (mean_s - mean)^2), no division inside the sums, the same bits in every
// workgroup and every run - and stages relu((x - mean) * invstd * gamma + beta).  Workgroup 0 also writes save_mean / save_invstd (the
// backward pass reads them) and moves the running statistics exactly as ee_bn.hip's forward does.
struct TrainBn {
    float *stats_out;            // producer: [result channels][S][2]; null: no statistics epilogue
    const float *part;           // consumer: the producer's stats_out, [reduction channels][S][2]
    int S;                       // partials per channel
    float cnt;                   // values per partial
    const float *gamma, *beta;   // [reduction channels]; null: 1 / 0
    float eps, momentum;
    float *running_mean, *running_var, *save_mean, *save_invstd;  // written by workgroup 0 (running_*: may be null)
};

// table [3 c + {0, 1, 2}] = (mean, invstd * gamma, beta) of channel c, for c < KC; call with all NT threads of the workgroup, then barrier
template <int NT>
__device__ __forceinline__ void train_bn_merge(const TrainBn &tb, int KC, float *table, bool writer) {
    const float n = static_cast<float>(tb.S) * tb.cnt;
    for (int c = threadIdx.x >> 4; c < KC; c += NT / 16) {
        const int j = threadIdx.x & 15;
        const float2 *pc = reinterpret_cast<const float2 *>(tb.part) + static_cast<size_t>(c) * tb.S;
        float s1 = 0.0f;
        for (int i = j; i < tb.S; i += 16) s1 += pc[i].x;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) s1 += __shfl_xor(s1, off, 16);
        const float mean = s1 / static_cast<float>(tb.S);
        float s2 = 0.0f;
        for (int i = j; i < tb.S; i += 16) {
            const float2 pn = pc[i];
            const float dm = pn.x - mean;
            s2 += pn.y + tb.cnt * (dm * dm);
        }
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) s2 += __shfl_xor(s2, off, 16);
        if (j == 0) {
            const float var = s2 / n, invstd = bn_invstd(var, tb.eps);
            table[3 * c] = mean;
            table[3 * c + 1] = invstd * (tb.gamma ? tb.gamma[c] : 1.0f);
            table[3 * c + 2] = tb.beta ? tb.beta[c] : 0.0f;
            if (writer) {
                tb.save_mean[c] = mean;
                tb.save_invstd[c] = invstd;
                if (tb.running_mean) {  // ee_bn.hip: bn_fwd_cached_kernel
                    const float unbiased = (n > 1.0f) ? var * (n / (n - 1.0f)) : var;
                    tb.running_mean[c] = (1.0f - tb.momentum) * tb.running_mean[c] + tb.momentum * mean;
                    tb.running_var[c] = (1.0f - tb.momentum) * tb.running_var[c] + tb.momentum * unbiased;
                }
            }
        }
    }
}
// ... and the BACKWARD direction (input gradient only): dx = gamma * invstd * ((dz - mean(dz)) - xhat * mean(dz * xhat)), dz = (bn(x) > 0) * dy.
// The convolution that PRODUCES dy (the backward-data of the layer behind the BatchNorm) writes, next to dy, per (channel, image) the sums
// (sum dz, sum dz * xhat) of the plane it holds - it reads x (the raw output of the layer in front) and the saved statistics for the mask and
// xhat; the convolution that CONSUMES dx merges the S partials per reduction channel (16 lanes, index order, fixed butterfly) and applies the
// expression while it stages dy and x.  table [7 c + ..] = (mean, invstd, invstd * gamma, beta, gamma * invstd, m1, m2)
template <int NT>
__device__ __forceinline__ void train_bn_bwd_merge(const TrainBn &tb, int KC, float *table) {
    const float n = static_cast<float>(tb.S) * tb.cnt;
    for (int c = threadIdx.x >> 4; c < KC; c += NT / 16) {
        const int j = threadIdx.x & 15;
        const float2 *pc = reinterpret_cast<const float2 *>(tb.part) + static_cast<size_t>(c) * tb.S;
        float s1 = 0.0f, s2 = 0.0f;
        for (int i = j; i < tb.S; i += 16) {
            const float2 pn = pc[i];
            s1 += pn.x, s2 += pn.y;
        }
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) s1 += __shfl_xor(s1, off, 16), s2 += __shfl_xor(s2, off, 16);
        if (j == 0) {
            const float invstd = tb.save_invstd[c], g = tb.gamma ? tb.gamma[c] : 1.0f;
            float *t = table + 7 * c;
            t[0] = tb.save_mean[c], t[1] = invstd, t[2] = invstd * g, t[3] = tb.beta ? tb.beta[c] : 0.0f, t[4] = g * invstd, t[5] = s1 / n, t[6] = s2 / n;
        }
    }
}
// ee_bn.hip: bn_bwd_cached_kernel with the mask recomputed from x (MaskArgs.on) - the same expressions
__device__ __forceinline__ float train_bn_bwd_apply(float d, float x, const float *t) {
    const float dz = ((x - t[0]) * t[2] + t[3]) > 0.0f ? d : 0.0f;
    return t[4] * ((dz - t[5]) - ((x - t[0]) * t[1]) * t[6]);
}
__device__ __forceinline__ float4 train_bn_bwd_apply4(float4 d, float4 x, const float *t) {
    return make_float4(train_bn_bwd_apply(d.x, x.x, t), train_bn_bwd_apply(d.y, x.y, t), train_bn_bwd_apply(d.z, x.z, t), train_bn_bwd_apply(d.w, x.w, t));
}

__device__ __forceinline__ float train_bn_apply(float v, const float *t) { return relu_keep_nan((v - t[0]) * t[1] + t[2]); }
__device__ __forceinline__ float4 train_bn_apply4(float4 v, const float *t) {
    return make_float4(train_bn_apply(v.x, t), train_bn_apply(v.y, t), train_bn_apply(v.z, t), train_bn_apply(v.w, t));
}
// (mean, M2) of a group of LANES consecutive lanes' values, NV per lane; every lane of the group returns the pair
template <int LANES, int NV>
__device__ __forceinline__ float2 group_moments(const float (&v)[NV]) {
    float s1 = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s1 += v[i];
#pragma unroll
    for (int off = 1; off < LANES; off <<= 1) s1 += __shfl_xor(s1, off, LANES);
    const float mean = s1 / static_cast<float>(LANES * NV);
    float s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i] - mean;
        s2 += a * a;
    }
#pragma unroll
    for (int off = 1; off < LANES; off <<= 1) s2 += __shfl_xor(s2, off, LANES);
    return make_float2(mean, s2);
}

static inline int check_post(const FusePost &p) {
    if (p.mean && !p.var) return EE_ERR_NULL;
    if (p.res && !aligned16(p.res)) return EE_ERR_ALIGN;
    return EE_OK;
}
static inline int check_pre(const FusePre &p) {
    if (!p.var) return EE_ERR_NULL;
    if ((p.add && !aligned16(p.add)) || (p.mask && !aligned16(p.mask)) || (p.store && !aligned16(p.store))) return EE_ERR_ALIGN;
    return EE_OK;
}

}  // namespace ee
