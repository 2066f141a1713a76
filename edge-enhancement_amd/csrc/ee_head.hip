// ee_head.hip - the classifier head of the reference's ResNets in one launch each way:
//   x = avgpool(x); x = x.view(B, -1); x = fc(x)        (Tiny_ImageNet/models_tinyimagenet/resnet.py:157-160)
// At the reference batch size this is a [100 x 512] x [512 x 200] product: the BLAS heuristics pick a 224x128 macro tile
// for it (ONE workgroup, 63 us on MI355X, measured with rocprofv3), and the pooling / its backward are two more launches.
// Here: workgroups of (image, 64 logits); the pooled feature row is staged in LDS, each wavefront walks four rows of W at a
// time with coalesced 256-B reads (a first version with one row in flight was latency-bound at 150 us).  The weight / bias gradients are needed once per training step only and stay on the BLAS.
//
// CNN-body glue, not a row of SURVEY.md section 8: parity is "logits within 1e-4" through the model tests.
#include "ee_common.hpp"

namespace {

using namespace ee;

constexpr int HEAD_NT = 256;
constexpr int HEAD_MAXC = 4096;  // pooled row staged in LDS (16 KB)

constexpr int HEAD_KCHUNK = 64;  // logits per workgroup: 4 wavefronts x 16, four rows of W in flight per wavefront

// grid (B, ceil(K / 64)): every workgroup pools its image's feature row into LDS (C*HW reads, L2-resident), then each
// wavefront produces 16 logits, 4 at a time so that 4 x C/64 independent loads are in flight per lane.
__global__ __launch_bounds__(HEAD_NT) void pool_linear_fwd_kernel(const float *__restrict__ feat, const float *__restrict__ w,
                                                                  const float *__restrict__ bias, float *__restrict__ pooled,
                                                                  float *__restrict__ logits, int C, int HW, int K, int vec4) {
    __shared__ __align__(16) float p[HEAD_MAXC];
    const int b = blockIdx.x;
    const float inv = 1.0f / static_cast<float>(HW);
    const float *f = feat + static_cast<size_t>(b) * C * HW;
    const bool quad = HW == 4 && (reinterpret_cast<uintptr_t>(f) & 15u) == 0;  // a 2x2 map (ResNet-18 at 64x64 inputs): one 16-B load per channel, same summation order
    for (int c = threadIdx.x; c < C; c += HEAD_NT) {
        float s = 0.0f;
        if (quad) {
            const float4 q = reinterpret_cast<const float4 *>(f)[c];
            s = ((q.x + q.y) + q.z) + q.w;
        } else {
            for (int i = 0; i < HW; ++i) s += f[static_cast<size_t>(c) * HW + i];
        }
        s *= inv;
        p[c] = s;
        if (blockIdx.y == 0) pooled[static_cast<size_t>(b) * C + c] = s;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k_base = blockIdx.y * HEAD_KCHUNK + wave * 16;
    if (vec4) {
        // C % 4 == 0, 16-byte aligned rows: all 16 rows of the wavefront in flight as 16-B loads (ResNet-18: 2 per row and lane),
        // one memory round trip for the whole product (the 4-rows-at-a-time loop below took four)
        const int C4 = C / 4;
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        for (int c0 = 0; c0 < C4; c0 += 128) {
            float4 wv[16][2];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = k_base + i < K ? k_base + i : K - 1;
                const float4 *row = reinterpret_cast<const float4 *>(w + static_cast<size_t>(k) * C);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c4 = c0 + lane + 64 * u;
                    wv[i][u] = row[c4 < C4 ? c4 : C4 - 1];
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c4 = c0 + lane + 64 * u;
                if (c4 < C4) {
                    const float4 pv = reinterpret_cast<const float4 *>(p)[c4];
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        acc[i] = fmaf(pv.w, wv[i][u].w, fmaf(pv.z, wv[i][u].z, fmaf(pv.y, wv[i][u].y, fmaf(pv.x, wv[i][u].x, acc[i]))));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off);
        }
        float mine = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) mine = lane == i ? acc[i] : mine;
        if (lane < 16 && k_base + lane < K) logits[static_cast<size_t>(b) * K + k_base + lane] = mine + (bias ? bias[k_base + lane] : 0.0f);
        return;
    }
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
        const int k0 = k_base + g * 4;
        if (k0 >= K) break;
        const float *wr[4];
        float acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + i < K ? k0 + i : K - 1;
            wr[i] = w + static_cast<size_t>(k) * C;
            acc[i] = 0.0f;
        }
        for (int c = lane; c < C; c += 64) {
            const float pv = p[c];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(pv, wr[i][c], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off);
        }
        if (lane < 4 && k0 + lane < K) {
            const float v = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
            logits[static_cast<size_t>(b) * K + k0 + lane] = v + (bias ? bias[k0 + lane] : 0.0f);
        }
    }
}

// dfeat[b,c,:] = (sum_k dlogits[b,k] * W[k,c]) / HW        grid (B, ceil(C / 256)), one lane per channel
__global__ __launch_bounds__(HEAD_NT) void pool_linear_bwd_kernel(const float *__restrict__ dlogits, const float *__restrict__ w,
                                                                  float *__restrict__ dfeat, int C, int HW, int K) {
    extern __shared__ float dl[];
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += HEAD_NT) dl[k] = dlogits[static_cast<size_t>(b) * K + k];
    __syncthreads();
    const int c = blockIdx.y * HEAD_NT + threadIdx.x;
    if (c >= C) return;
    const float inv = 1.0f / static_cast<float>(HW);
    float acc = 0.0f;
    int k = 0;
    for (; k + 8 <= K; k += 8) {
        float wv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wv[i] = w[static_cast<size_t>(k + i) * C + c];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = fmaf(dl[k + i], wv[i], acc);
    }
    for (; k < K; ++k) acc = fmaf(dl[k], w[static_cast<size_t>(k) * C + c], acc);
    acc *= inv;
    float *d = dfeat + (static_cast<size_t>(b) * C + c) * HW;
    for (int i = 0; i < HW; ++i) d[i] = acc;
}

// The cross-entropy gradient AND the head's backward in one launch (round 4): d CrossEntropyLoss(logits, labels) / d feat.  Every workgroup of
// an image (ceil(C / 256) of them) forms the row's softmax itself - wavefront 0 with ee_loss.hip's row_stats expressions (row maximum, the
// exponentials' sum carried in float64 through the same butterfly, its logarithm), then dl[k] = (exp((z_k - mx) - lse) - [k == y]) * gscale as
// ce_kernel writes it - and goes on as pool_linear_bwd_kernel does: the same bits as the two launches, one launch (5 us) less per iteration.
__global__ __launch_bounds__(HEAD_NT) void ce_pool_linear_bwd_kernel(const float *__restrict__ logits, const int64_t *__restrict__ labels, float gscale,
                                                                     const float *__restrict__ w, float *__restrict__ dfeat, int C, int HW, int K) {
    extern __shared__ float dl[];
    __shared__ float stats[2];
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const float *z = logits + static_cast<size_t>(b) * K;
    if (threadIdx.x < 64) {
        float m = -INFINITY;
        for (int k = lane; k < K; k += 64) m = fmaxf(m, z[k]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        double sum = 0.0;
        for (int k = lane; k < K; k += 64) sum += static_cast<double>(expf(z[k] - m));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        if (lane == 0) stats[0] = m, stats[1] = logf(static_cast<float>(sum));
    }
    __syncthreads();
    const float mx = stats[0], lse = stats[1];
    const int y = static_cast<int>(labels[b]);
    for (int k = threadIdx.x; k < K; k += HEAD_NT) dl[k] = (expf((z[k] - mx) - lse) - (k == y ? 1.0f : 0.0f)) * gscale;
    __syncthreads();
    const int c = blockIdx.y * HEAD_NT + threadIdx.x;
    if (c >= C) return;
    const float inv = 1.0f / static_cast<float>(HW);
    float acc = 0.0f;
    int k = 0;
    for (; k + 8 <= K; k += 8) {
        float wv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wv[i] = w[static_cast<size_t>(k + i) * C + c];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = fmaf(dl[k + i], wv[i], acc);
    }
    for (; k < K; ++k) acc = fmaf(dl[k], w[static_cast<size_t>(k) * C + c], acc);
    acc *= inv;
    float *d = dfeat + (static_cast<size_t>(b) * C + c) * HW;
    for (int i = 0; i < HW; ++i) d[i] = acc;
}

}  // namespace

EE_API int ee_pool_linear_fwd_f32(const float *feat, const float *weight, const float *bias, float *pooled, float *logits, int B, int C,
                                  int HW, int K, void *stream) {
    if (B < 0 || C < 1 || HW < 1 || K < 1) return EE_ERR_SHAPE;
    if (C > HEAD_MAXC) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!feat || !weight || !pooled || !logits) return EE_ERR_NULL;
    EE_LAUNCH(pool_linear_fwd_kernel, dim3(static_cast<unsigned>(B), static_cast<unsigned>((K + HEAD_KCHUNK - 1) / HEAD_KCHUNK)), dim3(HEAD_NT), 0, as_stream(stream), feat, weight, bias, pooled, logits, C,
              HW, K, (C % 4 == 0 && aligned16(weight)) ? 1 : 0);
    return launch_status();
}

EE_API int ee_pool_linear_bwd_f32(const float *dlogits, const float *weight, float *dfeat, int B, int C, int HW, int K, void *stream) {
    if (B < 0 || C < 1 || HW < 1 || K < 1) return EE_ERR_SHAPE;
    if (K > 8192) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!dlogits || !weight || !dfeat) return EE_ERR_NULL;
    EE_LAUNCH(pool_linear_bwd_kernel, dim3(static_cast<unsigned>(B), static_cast<unsigned>((C + HEAD_NT - 1) / HEAD_NT)), dim3(HEAD_NT), static_cast<size_t>(K) * sizeof(float), as_stream(stream),
              dlogits, weight, dfeat, C, HW, K);
    return launch_status();
}

// d CrossEntropyLoss(logits, labels) / d feat for logits = fc(global_avgpool(feat)) (resnet.py:157-160 + attacks.py:23 / :255), the loss
// gradient formed inside the head's backward launch: logits [B,K] (ee_pool_linear_fwd_f32), labels [B], gscale = 1 ("sum") or 1 / B ("mean"),
// weight [K,C] -> dfeat [B,C,HW].  The same bits as ee_ce_f32 (smoothing 0) followed by ee_pool_linear_bwd_f32.
EE_API int ee_ce_pool_linear_bwd_f32(const float *logits, const int64_t *labels, float gscale, const float *weight, float *dfeat, int B, int C, int HW,
                                     int K, void *stream) {
    if (B < 0 || C < 1 || HW < 1 || K < 1) return EE_ERR_SHAPE;
    if (K > 8192) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!logits || !labels || !weight || !dfeat) return EE_ERR_NULL;
    EE_LAUNCH(ce_pool_linear_bwd_kernel, dim3(static_cast<unsigned>(B), static_cast<unsigned>((C + HEAD_NT - 1) / HEAD_NT)), dim3(HEAD_NT),
              static_cast<size_t>(K) * sizeof(float), as_stream(stream), logits, labels, gscale, weight, dfeat, C, HW, K);
    return launch_status();
}
