// ee_pool.hip - the ResNet stem's MaxPool2d(kernel_size=3, stride=2, padding=1) (Tiny_ImageNet/models_tinyimagenet/resnet.py:117)
// forward + backward.  The stock backward gathers through an int64 index plane and takes 48 us for the stem's
// [100,64,32,32] map on MI355X (rocprofv3); here the argmax is a one-byte window code and the backward writes four input
// pixels per lane: both directions are plain HBM streams (fwd 26 MB read + 8 MB written, bwd 8 MB read + 26 MB written).
// Scan order, strict '>' and NaN capture follow ATen's max_pool2d (first maximum wins; a NaN always wins), so values,
// argmax and gradients are bit-identical to torch's.
//
// CNN-body glue, not a row of SURVEY.md section 8.
#include "ee_common.hpp"

#include <math.h>

namespace {

using namespace ee;

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, uint8_t *__restrict__ code,
                                                          int H, int W, int OH, int OW, int64_t total) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= total) return;
    const int ow = static_cast<int>(i % OW);
    const int64_t t = i / OW;
    const int oh = static_cast<int>(t % OH);
    const int64_t plane = t / OH;
    const float *p = x + plane * H * W;
    const int h0 = oh * 2 - 1, w0 = ow * 2 - 1;
    const int hs = h0 < 0 ? 0 : h0, ws = w0 < 0 ? 0 : w0;
    const int he = h0 + 3 > H ? H : h0 + 3, we = w0 + 3 > W ? W : w0 + 3;
    float best = -INFINITY;
    int bc = (hs - h0) * 3 + (ws - w0);
    for (int h = hs; h < he; ++h)
        for (int w = ws; w < we; ++w) {
            const float v = p[h * W + w];
            if (v > best || v != v) {
                best = v;
                bc = (h - h0) * 3 + (w - w0);
            }
        }
    y[i] = best;
    code[i] = static_cast<uint8_t>(bc);
}

// An input pixel (h, w) lies in the windows oh in {h/2} (h even) or {(h-1)/2, (h+1)/2} (h odd), same along w; ATen's
// backward adds the matching windows' gradients in (oh, ow) ascending order, and so does this.
__device__ __forceinline__ float pool_gather(const float *__restrict__ g, const uint8_t *__restrict__ cd, int h, int w, int OH, int OW) {
    const int oh0 = h >> 1, ow0 = w >> 1;
    const int noh = (h & 1) && oh0 + 1 < OH ? 2 : 1, now = (w & 1) && ow0 + 1 < OW ? 2 : 1;
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (a < noh && b < now) {
                const int oh = oh0 + a, ow = ow0 + b;
                const int want = (h - (oh * 2 - 1)) * 3 + (w - (ow * 2 - 1));
                if (cd[oh * OW + ow] == want) acc += g[oh * OW + ow];
            }
        }
    return acc;
}

// one lane = 4 consecutive input pixels of one row (W % 4 == 0): the 2 x 3 candidate windows are loaded once
template <int VEC>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float *__restrict__ dy, const uint8_t *__restrict__ code, float *__restrict__ dx,
                                                          int H, int W, int OH, int OW, int64_t total) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= total) return;
    const int wq = W / VEC;
    const int w_base = static_cast<int>(i % wq) * VEC;
    const int64_t t = i / wq;
    const int h = static_cast<int>(t % H);
    const int64_t plane = t / H;
    const float *g = dy + plane * OH * OW;
    const uint8_t *cd = code + plane * OH * OW;
    float *o = dx + (plane * H + h) * W + w_base;
    if (VEC == 1) {
        o[0] = pool_gather(g, cd, h, w_base, OH, OW);
        return;
    }
    // windows rows oh0 (+1 when h is odd), columns ow0 .. ow0+2 (w_base is even: pixels are even, odd, even, odd)
    const int oh0 = h >> 1, ow0 = w_base >> 1;
    const bool row2 = (h & 1) && oh0 + 1 < OH;
    float gv[2][3];
    int cv[2][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const bool ok = (a == 0 || row2) && ow0 + b < OW;
            const int idx = ok ? (oh0 + a) * OW + ow0 + b : 0;
            gv[a][b] = ok ? g[idx] : 0.0f;
            cv[a][b] = ok ? cd[idx] : 255;
        }
    float out[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int w = w_base + v;
        const int b0 = v >> 1;  // first candidate column relative to ow0
        float acc = 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                if (bb == 1 && !(v & 1)) continue;  // even pixels belong to one window column only
                const int b = b0 + bb;
                const int want = (h - ((oh0 + a) * 2 - 1)) * 3 + (w - ((ow0 + b) * 2 - 1));
                if (cv[a][b] == want) acc += gv[a][b];
            }
        out[v] = acc;
    }
    *reinterpret_cast<float4 *>(o) = make_float4(out[0], out[1], out[2], out[3]);
}

}  // namespace

EE_API int ee_maxpool3s2_fwd_f32(const float *x, float *y, uint8_t *code, int planes, int H, int W, void *stream) {
    if (planes < 0 || H < 1 || W < 1) return EE_ERR_SHAPE;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;  // floor((H + 2*1 - 3) / 2) + 1
    const int64_t total = static_cast<int64_t>(planes) * OH * OW;
    if (total == 0) return EE_OK;
    if (!x || !y || !code) return EE_ERR_NULL;
    if ((total + 255) / 256 > 0x7fffffffLL) return EE_ERR_SHAPE;
    EE_LAUNCH(maxpool_fwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, as_stream(stream), x, y, code, H, W, OH, OW, total);
    return launch_status();
}

EE_API int ee_maxpool3s2_bwd_f32(const float *dy, const uint8_t *code, float *dx, int planes, int H, int W, void *stream) {
    if (planes < 0 || H < 1 || W < 1) return EE_ERR_SHAPE;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    if (planes == 0) return EE_OK;
    if (!dy || !code || !dx) return EE_ERR_NULL;
    const bool vec = (W % 4 == 0) && aligned16(dx);
    const int64_t total = static_cast<int64_t>(planes) * H * (vec ? W / 4 : W);
    if ((total + 255) / 256 > 0x7fffffffLL) return EE_ERR_SHAPE;
    const dim3 grid(static_cast<unsigned>((total + 255) / 256)), block(256);
    if (vec)
        EE_LAUNCH(maxpool_bwd_kernel<4>, grid, block, 0, as_stream(stream), dy, code, dx, H, W, OH, OW, total);
    else
        EE_LAUNCH(maxpool_bwd_kernel<1>, grid, block, 0, as_stream(stream), dy, code, dx, H, W, OH, OW, total);
    return launch_status();
}
