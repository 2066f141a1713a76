// ee_square.hpp - Add_Square (utils/core.py:589-655) per-element value and derivative, shared by the stand-alone
// kernels (ee_square.hip) and by the low-pass kernel that fuses them into its load / store stage (ee_hfs.hip).
#pragma once
#include "ee_common.hpp"

namespace ee {

struct SquareArgs {
    const float *stripe;     // [B,C,W]  sign(2*rand-1), core.py:637
    const float *sq_sign;    // [nq,C]   sign draws, core.py:648
    const int64_t *sq_pos;   // [nq]     vh, core.py:645
    const int32_t *sq_size;  // [nq]     s, core.py:644
    int nq, C, H, W;
    float eps, two_eps;
};

// The first kSquareCached queries of one (b, c) plane, pulled into registers once per lane so that the per-element
// loop does not re-load them from global memory (the reference configs use n_queries = 1).
constexpr int kSquareCached = 2;
struct SquarePlane {
    int vh[kSquareCached], s[kSquareCached];
    float delta[kSquareCached];  // 2*eps*sign of the query for this plane's channel
};

__device__ __forceinline__ SquarePlane square_plane(const SquareArgs &a, int c) {
    SquarePlane p;
#pragma unroll
    for (int q = 0; q < kSquareCached; ++q) {
        const bool on = q < a.nq;
        p.vh[q] = on ? static_cast<int>(a.sq_pos[q]) : 0;
        p.s[q] = on ? a.sq_size[q] : 0;
        p.delta[q] = on ? a.two_eps * a.sq_sign[q * a.C + c] : 0.0f;
    }
    return p;
}

// value and d(out)/d(x) for one element, following core.py:637-653 and autograd's rules:
// clamp passes the gradient on the closed interval; max/min split a tie 1/2 : 1/2 (both operands
// depend on x with slope 1, except the clamped stripe start whose slope is `d`).
// `stripe_v` = the stripe sign of this element's (b, c, w) (core.py:637).
template <bool WANT_D>
__device__ __forceinline__ float square_elem(const SquareArgs &a, const SquarePlane &pl, float x, float stripe_v, int c, int h, int w,
                                             float &d) {
    // x finite (an image): hardware min / max (one instruction each) give torch.min / max / clamp exactly; a NaN x - which torch
    // propagates through every step - is patched in at the end (value NaN, derivative 0: every comparison on the way is false)
    const float t0 = x + a.eps * stripe_v;
    float y = fminf(fmaxf(t0, 0.0f), 1.0f);
    if (WANT_D) d = (t0 >= 0.0f && t0 <= 1.0f) ? 1.0f : 0.0f;
    const float lb = x - a.eps, ub = x + a.eps;
    auto query = [&](int vh, int s, float dq) {
        const bool inside = (h >= vh && h < vh + s && w >= vh && w < vh + s);
        const float y1 = y + (inside ? dq : 0.0f);
        const float m = fmaxf(y1, lb);
        const float y2 = fminf(m, ub);
        if (WANT_D) {
            const float dm = (y1 > lb) ? d : ((y1 < lb) ? 1.0f : 0.5f * d + 0.5f);
            const float d2 = (m < ub) ? dm : ((m > ub) ? 1.0f : 0.5f * dm + 0.5f);
            d = (y2 >= 0.0f && y2 <= 1.0f) ? d2 : 0.0f;
        }
        y = fminf(fmaxf(y2, 0.0f), 1.0f);
    };
#pragma unroll
    for (int q = 0; q < kSquareCached; ++q)
        if (q < a.nq) query(pl.vh[q], pl.s[q], pl.delta[q]);
    for (int q = kSquareCached; q < a.nq; ++q)
        query(static_cast<int>(a.sq_pos[q]), a.sq_size[q], a.two_eps * a.sq_sign[q * a.C + c]);
    if (x != x) {
        y = x;
        if (WANT_D) d = 0.0f;
    }
    return y;
}

}  // namespace ee
