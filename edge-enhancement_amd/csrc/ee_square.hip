// ee_square.hip - Add_Square (utils/core.py:589-655) as ONE element-wise kernel per direction.
//
// The reference builds a dense `new_deltas` tensor per query and runs ~12 launches plus a device->host
// sync (the random offset is used as a Python slice bound, core.py:645-647).  Given its random draws the
// op is element-wise, so here every element walks the nq queries in registers: 8 B of HBM traffic per
// element forward (read x, write out) and 12 B backward (read g_out and x, write g_x), whatever nq is.
// The draws stay on the device (stripe / sq_sign / sq_pos are device arrays), so nothing synchronises.
#include "ee_common.hpp"
#include "ee_square.hpp"

namespace {

using namespace ee;

template <bool BWD>
__global__ __launch_bounds__(kBlock) void square_kernel(const float *__restrict__ x, const float *__restrict__ g_out,
                                                        float *__restrict__ out, int64_t n, SquareArgs a) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t hw = static_cast<int64_t>(a.H) * a.W;
    const bool vec = ((a.W & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0) &&
                     (!BWD || (reinterpret_cast<uintptr_t>(g_out) & 15u) == 0);
    for (int64_t v = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; v < (n + 3) / 4; v += stride) {
        const int64_t e0 = v * 4;
        // W % 4 == 0: the four elements share (b, c, h); otherwise decode each element
        const int64_t plane0 = e0 / hw;
        const int b0 = static_cast<int>(plane0 / a.C), c0 = static_cast<int>(plane0 - static_cast<int64_t>(b0) * a.C);
        const SquarePlane pl0 = square_plane(a, c0);
        float xv[4] = {0, 0, 0, 0}, gv[4] = {0, 0, 0, 0}, r[4];
        if (vec) {
            const float4 t = *reinterpret_cast<const float4 *>(x + e0);
            xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            if (BWD) {
                const float4 g4 = *reinterpret_cast<const float4 *>(g_out + e0);
                gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
            }
        } else {
            for (int k = 0; k < 4 && e0 + k < n; ++k) {
                xv[k] = x[e0 + k];
                if (BWD) gv[k] = g_out[e0 + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t e = e0 + k;
            r[k] = 0.0f;
            if (e < n) {
                const int64_t plane = e / hw, pix = e - plane * hw;
                const int b = static_cast<int>(plane / a.C), c = static_cast<int>(plane - static_cast<int64_t>(b) * a.C);
                const int h = static_cast<int>(pix / a.W), w = static_cast<int>(pix - static_cast<int64_t>(h) * a.W);
                const float sv = a.stripe[(static_cast<size_t>(b) * a.C + c) * a.W + w];
                float d = 0.0f;
                const float y = (c == c0 && b == b0) ? square_elem<BWD>(a, pl0, xv[k], sv, c, h, w, d)
                                                     : square_elem<BWD>(a, square_plane(a, c), xv[k], sv, c, h, w, d);
                r[k] = BWD ? gv[k] * d : y;
            }
        }
        if (vec) {
            *reinterpret_cast<float4 *>(out + e0) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
            for (int k = 0; k < 4 && e0 + k < n; ++k) out[e0 + k] = r[k];
        }
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

// ---- the random draws of one Add_Square forward (core.py:637, :645, :648) in ONE launch ----------------------------------
// The reference spells them as ~15 tiny torch ops (rand, 2*t-1, sign, rand, (h-s)*u, .long(), rand, ...), i.e. ~15 launches per
// model forward inside a launch-bound attack loop.  Here a single workgroup draws everything from Philox4x32-10 keyed by a
// (seed, offset) pair that lives in DEVICE memory and is advanced by the kernel itself, so a captured HIP graph draws fresh
// numbers at every replay.  Same expressions as the reference on the uniforms: sign(2u - 1) and long(0 + (h - s) * u).
constexpr int DRAW_NT = 1024;

__global__ __launch_bounds__(DRAW_NT) void square_draw_kernel(float *__restrict__ stripe, int64_t n_stripe, int64_t *__restrict__ sq_pos,
                                                              float *__restrict__ sq_sign, const int32_t *__restrict__ sq_size, int nq, int C,
                                                              int h, unsigned long long *state) {
    const unsigned long long seed = state[0], base = state[1];
    __syncthreads();  // everybody has read the state before lane 0 advances it
    const Philox rng(seed);
    const int64_t total = n_stripe + static_cast<int64_t>(nq) * (1 + C);
    const int64_t nv = (total + 3) >> 2;
    for (int64_t v = threadIdx.x; v < nv; v += DRAW_NT) {
        const uint4 r = rng(base + static_cast<unsigned long long>(v));
        const float uu[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t e = (v << 2) + k;
            if (e >= total) break;
            if (e < n_stripe) {
                stripe[e] = sgn(2.0f * uu[k] - 1.0f);
            } else if (e < n_stripe + nq) {
                const int64_t q = e - n_stripe;
                const float span = static_cast<float>(h) - static_cast<float>(sq_size[q]);
                sq_pos[q] = static_cast<int64_t>(0.0f + (span - 0.0f) * uu[k]);
            } else {
                sq_sign[e - n_stripe - nq] = sgn(2.0f * uu[k] - 1.0f);
            }
        }
    }
    if (threadIdx.x == 0) state[1] = base + static_cast<unsigned long long>(nv);
}

}  // namespace

EE_API int ee_square_draw_f32(float *stripe, int64_t n_stripe, int64_t *sq_pos, float *sq_sign, const int32_t *sq_size, int nq, int C, int h,
                              uint64_t *state, void *stream) {
    if (n_stripe < 0 || nq < 0 || C < 1 || h < 1) return EE_ERR_SHAPE;
    if (!state) return EE_ERR_NULL;
    if ((n_stripe && !stripe) || (nq && (!sq_pos || !sq_sign || !sq_size))) return EE_ERR_NULL;
    ProfScope prof(EE_K_SQUARE_DRAW, as_stream(stream));
    EE_LAUNCH(square_draw_kernel, dim3(1), dim3(DRAW_NT), 0, as_stream(stream), stripe, n_stripe, sq_pos, sq_sign, sq_size, nq, C, h,
              reinterpret_cast<unsigned long long *>(state));
    return launch_status();
}

EE_API int ee_add_square_fwd_f32(const float *x, int B, int C, int H, int W, float eps, const float *stripe, const float *sq_sign,
                                 const int64_t *sq_pos, const int32_t *sq_size, int nq, float *out, void *stream) {
    return launch(false, x, nullptr, out, B, C, H, W, eps, stripe, sq_sign, sq_pos, sq_size, nq, as_stream(stream));
}

EE_API int ee_add_square_bwd_f32(const float *g_out, const float *x, int B, int C, int H, int W, float eps, const float *stripe,
                                 const float *sq_sign, const int64_t *sq_pos, const int32_t *sq_size, int nq, float *g_x,
                                 void *stream) {
    return launch(true, x, g_out, g_x, B, C, H, W, eps, stripe, sq_sign, sq_pos, sq_size, nq, as_stream(stream));
}
