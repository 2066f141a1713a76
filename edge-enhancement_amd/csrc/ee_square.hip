// ee_square.hip - Add_Square (utils/core.py:589-655) as ONE element-wise kernel per direction.
//
// The reference builds a dense `new_deltas` tensor per query and runs ~12 launches plus a device->host
// sync (the random offset is used as a Python slice bound, core.py:645-647).  Given its random draws the
// op is element-wise, so here every element walks the nq queries in registers: 8 B of HBM traffic per
// element forward (read x, write out) and 12 B backward (read g_out and x, write g_x), whatever nq is.
// The draws stay on the device (stripe / sq_sign / sq_pos are device arrays), so nothing synchronises.
#include "ee_common.hpp"

namespace {

using namespace ee;

struct SquareArgs {
    const float *stripe;     // [B,C,W]  sign(2*rand-1), core.py:637
    const float *sq_sign;    // [nq,C]   sign draws, core.py:648
    const int64_t *sq_pos;   // [nq]     vh, core.py:645
    const int32_t *sq_size;  // [nq]     s, core.py:644
    int nq, C, H, W;
    float eps, two_eps;
};

// value and d(out)/d(x) for one element, following core.py:637-653 and autograd's rules:
// clamp passes the gradient on the closed interval; max/min split a tie 1/2 : 1/2 (both operands
// depend on x with slope 1, except the clamped stripe start whose slope is `d`).
template <bool WANT_D>
__device__ __forceinline__ float square_elem(const SquareArgs &a, float x, int b, int c, int h, int w, float &d) {
    const float t0 = x + a.eps * a.stripe[(static_cast<size_t>(b) * a.C + c) * a.W + w];
    float y = tclamp(t0, 0.0f, 1.0f);
    if (WANT_D) d = (t0 >= 0.0f && t0 <= 1.0f) ? 1.0f : 0.0f;
    const float lb = x - a.eps, ub = x + a.eps;
    for (int q = 0; q < a.nq; ++q) {
        const int vh = static_cast<int>(a.sq_pos[q]), s = a.sq_size[q];
        const bool inside = (h >= vh && h < vh + s && w >= vh && w < vh + s);
        const float delta = inside ? a.two_eps * a.sq_sign[q * a.C + c] : 0.0f;
        const float y1 = y + delta;
        const float m = tmax(y1, lb);
        const float y2 = tmin(m, ub);
        if (WANT_D) {
            const float dm = (y1 > lb) ? d : ((y1 < lb) ? 1.0f : 0.5f * d + 0.5f);
            const float d2 = (m < ub) ? dm : ((m > ub) ? 1.0f : 0.5f * dm + 0.5f);
            d = (y2 >= 0.0f && y2 <= 1.0f) ? d2 : 0.0f;
        }
        y = tclamp(y2, 0.0f, 1.0f);
    }
    return y;
}

template <bool BWD>
__global__ __launch_bounds__(kBlock) void square_kernel(const float *__restrict__ x, const float *__restrict__ g_out,
                                                        float *__restrict__ out, int64_t n, SquareArgs a) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t hw = static_cast<int64_t>(a.H) * a.W;
    for (int64_t v = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; v < (n + 3) / 4; v += stride) {
        const int64_t e0 = v * 4;
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t e = e0 + k;
            r[k] = 0.0f;
            if (e < n) {
                const int64_t plane = e / hw, pix = e - plane * hw;
                const int b = static_cast<int>(plane / a.C), c = static_cast<int>(plane - static_cast<int64_t>(b) * a.C);
                const int h = static_cast<int>(pix / a.W), w = static_cast<int>(pix - static_cast<int64_t>(h) * a.W);
                float d = 0.0f;
                const float y = square_elem<BWD>(a, x[e], b, c, h, w, d);
                r[k] = BWD ? g_out[e] * d : y;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (e0 + k < n) out[e0 + k] = r[k];
    }
}

int launch(bool bwd, const float *x, const float *g_out, float *out, int B, int C, int H, int W, float eps, const float *stripe,
           const float *sq_sign, const int64_t *sq_pos, const int32_t *sq_size, int nq, hipStream_t s) {
    if (!x || !out || !stripe || (bwd && !g_out)) return EE_ERR_NULL;
    if (nq > 0 && (!sq_sign || !sq_pos || !sq_size)) return EE_ERR_NULL;
    if (B < 0 || C < 1 || H < 1 || W < 1 || nq < 0) return EE_ERR_SHAPE;
    const int64_t n = static_cast<int64_t>(B) * C * H * W;
    if (n == 0) return EE_OK;
    SquareArgs a{stripe, sq_sign, sq_pos, sq_size, nq, C, H, W, eps, static_cast<float>(2.0 * static_cast<double>(eps))};
    int64_t blocks = ((n + 3) / 4 + kBlock - 1) / kBlock;
    if (blocks > kMaxGrid) blocks = kMaxGrid;
    if (bwd)
        EE_LAUNCH(square_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, s, x, g_out, out, n, a);
    else
        EE_LAUNCH(square_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, s, x, g_out, out, n, a);
    return launch_status();
}

}  // namespace

EE_API int ee_add_square_fwd_f32(const float *x, int B, int C, int H, int W, float eps, const float *stripe, const float *sq_sign,
                                 const int64_t *sq_pos, const int32_t *sq_size, int nq, float *out, void *stream) {
    return launch(false, x, nullptr, out, B, C, H, W, eps, stripe, sq_sign, sq_pos, sq_size, nq, as_stream(stream));
}

EE_API int ee_add_square_bwd_f32(const float *g_out, const float *x, int B, int C, int H, int W, float eps, const float *stripe,
                                 const float *sq_sign, const int64_t *sq_pos, const int32_t *sq_size, int nq, float *g_x,
                                 void *stream) {
    return launch(true, x, g_out, g_x, B, C, H, W, eps, stripe, sq_sign, sq_pos, sq_size, nq, as_stream(stream));
}
