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
