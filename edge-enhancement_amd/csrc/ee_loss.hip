// ee_loss.hip - per-row loss kernels on logits [B,K] (utils/attacks.py losses, utils/helper.py accuracy).
//
// One 64-lane wavefront owns one row (4 rows per 256-thread workgroup): row max, sum-exp and the
// weighted sums are wave-shuffle butterflies, so a row never touches LDS and the result does not
// depend on launch geometry.  Row arithmetic is fp32 like the reference's log_softmax; sums that feed
// a scalar loss are carried in float64 and folded in a fixed order (ee_reduce_rows_f64), which makes
// every scalar bit-reproducible run to run.  These kernels are latency-bound (B*K*4 B <= 400 KB).
#include <math.h>

#include "ee_common.hpp"

namespace {

using namespace ee;

constexpr int kRowsPerBlock = kBlock / kWave;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// max and log-sum-exp of one row, as log_softmax needs them:  logp_k = (z_k - mx) - lse
__device__ __forceinline__ void row_stats(const float *__restrict__ z, int K, int lane, float &mx, float &lse) {
    float m = -INFINITY;
    for (int k = lane; k < K; k += kWave) m = fmaxf(m, z[k]);
    mx = wave_max(m);
    double s = 0.0;
    for (int k = lane; k < K; k += kWave) s += static_cast<double>(expf(z[k] - mx));
    s = wave_sum(s);
    lse = logf(static_cast<float>(s));
}

__global__ __launch_bounds__(kBlock) void ce_kernel(const float *__restrict__ logits, const int64_t *__restrict__ labels, int B,
                                                    int K, float smoothing, float gscale, double *__restrict__ row_loss,
                                                    float *__restrict__ dlogits) {
    const int lane = threadIdx.x & (kWave - 1);
    const int row = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= B) return;
    const float *z = logits + static_cast<size_t>(row) * K;
    const int y = static_cast<int>(labels[row]);
    float mx, lse;
    row_stats(z, K, lane, mx, lse);
    const float w_off = smoothing / (static_cast<float>(K) - 1.0f), w_on = 1.0f - smoothing;
    if (row_loss) {
        double v;
        if (smoothing == 0.0f) {
            v = static_cast<double>(lse - (z[y] - mx));
        } else {
            double acc = 0.0;
            for (int k = lane; k < K; k += kWave) {
                const float lp = (z[k] - mx) - lse;
                acc += static_cast<double>(-(k == y ? w_on : w_off) * lp);
            }
            v = wave_sum(acc);
        }
        if (lane == 0) row_loss[row] = v;
    }
    if (dlogits) {
        float *d = dlogits + static_cast<size_t>(row) * K;
        for (int k = lane; k < K; k += kWave) {
            const float pk = expf((z[k] - mx) - lse);
            const float wk = (smoothing == 0.0f) ? (k == y ? 1.0f : 0.0f) : (k == y ? w_on : w_off);
            d[k] = (pk - wk) * gscale;
        }
    }
}

__global__ __launch_bounds__(kBlock) void kl_kernel(const float *__restrict__ zq, const float *__restrict__ zp, int B, int K,
                                                    float gscale, double *__restrict__ row_loss, float *__restrict__ dzq,
                                                    float *__restrict__ dzp) {
    const int lane = threadIdx.x & (kWave - 1);
    const int row = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= B) return;
    const float *q = zq + static_cast<size_t>(row) * K, *p = zp + static_cast<size_t>(row) * K;
    float mq, lq, mp, lp;
    row_stats(q, K, lane, mq, lq);
    row_stats(p, K, lane, mp, lp);
    double acc = 0.0;
    for (int k = lane; k < K; k += kWave) {
        const float logp = (p[k] - mp) - lp, logq = (q[k] - mq) - lq;
        const float pk = expf(logp);
        if (pk > 0.0f) acc += static_cast<double>(pk * (logp - logq));  // KLDivLoss: 0 where target == 0
    }
    const double klb = wave_sum(acc);
    if (row_loss && lane == 0) row_loss[row] = klb;
    if (dzq || dzp) {
        const float klf = static_cast<float>(klb);
        for (int k = lane; k < K; k += kWave) {
            const float logp = (p[k] - mp) - lp, logq = (q[k] - mq) - lq;
            const float pk = expf(logp), qk = expf(logq);
            if (dzq) dzq[static_cast<size_t>(row) * K + k] = (qk - pk) * gscale;
            if (dzp) dzp[static_cast<size_t>(row) * K + k] = pk * ((logp - logq) - klf) * gscale;
        }
    }
}

__global__ __launch_bounds__(kBlock) void softce_kernel(const float *__restrict__ z_, const double *__restrict__ t_, int B, int K,
                                                        double gscale, double *__restrict__ row_loss, double *__restrict__ dz) {
    const int lane = threadIdx.x & (kWave - 1);
    const int row = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= B) return;
    const float *z = z_ + static_cast<size_t>(row) * K;
    const double *t = t_ + static_cast<size_t>(row) * K;
    float mx, lse;
    row_stats(z, K, lane, mx, lse);
    double acc = 0.0, ts = 0.0;
    for (int k = lane; k < K; k += kWave) {
        const float lp = (z[k] - mx) - lse;
        acc -= static_cast<double>(lp) * t[k];
        ts += t[k];
    }
    acc = wave_sum(acc);
    ts = wave_sum(ts);
    if (row_loss && lane == 0) row_loss[row] = acc;
    if (dz)
        for (int k = lane; k < K; k += kWave) {
            const float lp = (z[k] - mx) - lse;
            dz[static_cast<size_t>(row) * K + k] = gscale * (static_cast<double>(expf(lp)) * ts - t[k]);
        }
}

constexpr int kMseChunk = 4096;  // elements per workgroup -> one float64 partial
__global__ __launch_bounds__(kBlock) void mse_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t n,
                                                     float gscale, double *__restrict__ partial, float *__restrict__ da) {
    __shared__ double wsum[kRowsPerBlock];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kMseChunk;
    double acc = 0.0;
    for (int k = threadIdx.x; k < kMseChunk; k += kBlock) {
        const int64_t i = base + k;
        if (i < n) {
            const float d = a[i] - b[i];
            acc += static_cast<double>(d * d);
            if (da) da[i] = gscale * d;
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && partial) partial[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}

// out[0] = scale * sum(rows) with a fixed association: lane-strided partials, butterfly, wave order
__global__ __launch_bounds__(kBlock) void reduce_rows_kernel(const double *__restrict__ rows, int64_t n, double scale,
                                                             double *__restrict__ out) {
    __shared__ double wsum[kRowsPerBlock];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += kBlock) acc += rows[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = scale * (((wsum[0] + wsum[1]) + wsum[2]) + wsum[3]);
}

// top-k by repeated wave arg-max; ties -> lower index; NaN ranks above everything (as torch.topk)
__device__ __forceinline__ bool better(float va, int ia, float vb, int ib) {
    const bool na = va != va, nb = vb != vb;
    if (na != nb) return na;
    if (!na && va != vb) return va > vb;
    return ia < ib;
}

constexpr int kMaxTopK = 16;
__global__ __launch_bounds__(kBlock) void topk_kernel(const float *__restrict__ logits, const int64_t *__restrict__ labels, int B,
                                                      int K, int k, int64_t *__restrict__ idx,
                                                      unsigned long long *__restrict__ correct) {
    const int lane = threadIdx.x & (kWave - 1);
    const int row = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= B) return;
    const float *z = logits + static_cast<size_t>(row) * K;
    int chosen[kMaxTopK];
#pragma unroll
    for (int j = 0; j < kMaxTopK; ++j) chosen[j] = -1;
    int hit = -1;
    const int y = labels ? static_cast<int>(labels[row]) : -1;
#pragma unroll
    for (int j = 0; j < kMaxTopK; ++j) {
        if (j < k) {
            float bv = 0.0f;
            int bi = 0x7fffffff;
            for (int c = lane; c < K; c += kWave) {
                bool taken = false;
#pragma unroll
                for (int jj = 0; jj < kMaxTopK; ++jj) taken |= (jj < j && chosen[jj] == c);
                if (taken) continue;
                const float v = z[c];
                if (bi == 0x7fffffff || better(v, c, bv, bi)) {
                    bv = v;
                    bi = c;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(bv, off);
                const int oi = __shfl_xor(bi, off);
                if (oi != 0x7fffffff && (bi == 0x7fffffff || better(ov, oi, bv, bi))) {
                    bv = ov;
                    bi = oi;
                }
            }
            chosen[j] = bi;
            if (lane == 0) idx[static_cast<size_t>(row) * k + j] = bi;
            if (bi == y && hit < 0) hit = j;
        }
    }
    if (lane == 0 && correct && hit >= 0)
        for (int j = hit; j < k; ++j) atomicAdd(&correct[j], 1ULL);
}

inline unsigned row_grid(int B) { return static_cast<unsigned>((B + kRowsPerBlock - 1) / kRowsPerBlock); }

// ---- the last two layers of a LeNet-style classifier and the cross-entropy behind them, gradient only ----------------------------------
// `return self.fc2(F.relu(self.fc1(x)))` (MNIST/models_mnist/Net2.py:27-28) followed by CrossEntropyLoss (attacks.py:23): given the
// pre-activation z1 = fc1(x) [B][Hd] this computes h = relu(z1), logits = h W2^T + b2 [K <= 64], the loss gradient with ce_kernel's own
// arithmetic, and d loss / d z1 = (dlogits W2) * (z1 > 0) - five launches of the stock sequence (ReLU, GEMM, CE, GEMM, ReLU mask: 27 us
// of a 124 us PGD iteration on the MNIST config) as one.  One workgroup per image.
constexpr int kFcMaxK = 64, kFcMaxH = 8192;
__global__ __launch_bounds__(kBlock) void fc_ce_grad_kernel(const float *__restrict__ z1, const float *__restrict__ w2, const float *__restrict__ b2,
                                                            const int64_t *__restrict__ labels, int Hd, int K, float gscale, float *__restrict__ dz1,
                                                            float *__restrict__ logits_out) {
    extern __shared__ float h[];  // [Hd] relu(z1), then [K] logits, [K] dlogits
    float *lg = h + Hd, *dl = lg + kFcMaxK;
    const int row = blockIdx.x, lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const float *z = z1 + static_cast<size_t>(row) * Hd;
    for (int j = threadIdx.x; j < Hd; j += kBlock) {
        const float v = z[j];
        h[j] = v > 0.0f ? v : (v != v ? v : 0.0f);  // relu keeps a NaN
    }
    __syncthreads();
    for (int k = wave; k < K; k += kRowsPerBlock) {  // one wavefront per class: lane-strided partial sums in index order, then the butterfly
        const float *wr = w2 + static_cast<size_t>(k) * Hd;
        float acc = 0.0f;
        for (int j = lane; j < Hd; j += kWave) acc = fmaf(wr[j], h[j], acc);
        acc = wave_sum(acc);
        if (lane == 0) lg[k] = acc + (b2 ? b2[k] : 0.0f);
    }
    __syncthreads();
    if (wave == 0) {
        float mx, lse;
        row_stats(lg, K, lane, mx, lse);
        const int y = static_cast<int>(labels[row]);
        for (int k = lane; k < K; k += kWave) {
            dl[k] = (expf((lg[k] - mx) - lse) - (k == y ? 1.0f : 0.0f)) * gscale;
            if (logits_out) logits_out[static_cast<size_t>(row) * K + k] = lg[k];
        }
    }
    __syncthreads();
    float *d = dz1 + static_cast<size_t>(row) * Hd;
    for (int j = threadIdx.x; j < Hd; j += kBlock) {
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc = fmaf(dl[k], w2[static_cast<size_t>(k) * Hd + j], acc);
        d[j] = (z[j] <= 0.0f) ? 0.0f : acc;  // threshold_backward: zero where the input was <= 0, so a NaN input passes the gradient
    }
}

// The same for K <= 16 classes and Hd <= 1024 (the MNIST shape: 10 x 1024): a lane owns four hidden units and keeps their fc2 weights in
// registers for both directions (one round of independent loads instead of per-class dependent rounds and a second pass over W2): 10.9 -> 6 us
__global__ __launch_bounds__(kBlock) void fc_ce_grad_small_kernel(const float *__restrict__ z1, const float *__restrict__ w2, const float *__restrict__ b2,
                                                                  const int64_t *__restrict__ labels, int Hd, int K, float gscale, float *__restrict__ dz1,
                                                                  float *__restrict__ logits_out) {
    constexpr int KM = 16, JP = 4;
    __shared__ float part[kRowsPerBlock][KM], lg[KM], dl[KM];
    const int row = blockIdx.x, lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const float *z = z1 + static_cast<size_t>(row) * Hd;
    float zv[JP], w[KM][JP];
    bool in[JP];
#pragma unroll
    for (int i = 0; i < JP; ++i) {
        const int j = threadIdx.x + kBlock * i;
        in[i] = j < Hd;
        zv[i] = z[in[i] ? j : 0];  // clamped address, masked below
    }
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int i = 0; i < JP; ++i) {
            const int j = threadIdx.x + kBlock * i;
            w[k][i] = w2[static_cast<size_t>(k < K ? k : K - 1) * Hd + (in[i] ? j : 0)];
        }
    float hv[JP];
#pragma unroll
    for (int i = 0; i < JP; ++i) hv[i] = !in[i] ? 0.0f : (zv[i] > 0.0f ? zv[i] : (zv[i] != zv[i] ? zv[i] : 0.0f));
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < JP; ++i) acc = fmaf(in[i] ? w[k][i] : 0.0f, hv[i], acc);
        acc = wave_sum(acc);
        if (lane == 0) part[wave][k] = acc;
    }
    __syncthreads();
    const int tk = static_cast<int>(threadIdx.x);
    if (tk < KM) lg[tk] = tk < K ? (((part[0][tk] + part[1][tk]) + part[2][tk]) + part[3][tk]) + (b2 ? b2[tk] : 0.0f) : 0.0f;
    __syncthreads();
    if (wave == 0) {
        float mx, lse;
        row_stats(lg, K, lane, mx, lse);
        const int y = static_cast<int>(labels[row]);
        if (lane < KM) dl[lane] = lane < K ? (expf((lg[lane] - mx) - lse) - (lane == y ? 1.0f : 0.0f)) * gscale : 0.0f;
        if (logits_out && lane < K) logits_out[static_cast<size_t>(row) * K + lane] = lg[lane];
    }
    __syncthreads();
    float *d = dz1 + static_cast<size_t>(row) * Hd;
#pragma unroll
    for (int i = 0; i < JP; ++i) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < KM; ++k) acc = fmaf(dl[k], k < K ? w[k][i] : 0.0f, acc);
        if (in[i]) d[threadIdx.x + kBlock * i] = (zv[i] <= 0.0f) ? 0.0f : acc;  // as above: !(z <= 0), NaN included
    }
}

}  // namespace

EE_API int ee_ce_f32(const float *logits, const int64_t *labels, int B, int K, float smoothing, float gscale, double *row_loss,
                     float *dlogits, void *stream) {
    if (B < 0 || K < 1 || K > 65536 || (smoothing != 0.0f && K < 2)) return EE_ERR_SHAPE;
    if (B == 0) return EE_OK;  // an empty batch has no storage: its NULL pointers are not an error
    if (!logits || !labels) return EE_ERR_NULL;
    ProfScope prof(EE_K_CE, as_stream(stream));
    EE_LAUNCH(ce_kernel, dim3(row_grid(B)), dim3(kBlock), 0, as_stream(stream), logits, labels, B, K, smoothing, gscale,
                       row_loss, dlogits);
    return launch_status();
}

EE_API int ee_kl_f32(const float *zq, const float *zp, int B, int K, float gscale, double *row_loss, float *dzq, float *dzp,
                     void *stream) {
    if (B < 0 || K < 1 || K > 65536) return EE_ERR_SHAPE;
    if (B == 0) return EE_OK;
    if (!zq || !zp) return EE_ERR_NULL;
    EE_LAUNCH(kl_kernel, dim3(row_grid(B)), dim3(kBlock), 0, as_stream(stream), zq, zp, B, K, gscale, row_loss, dzq, dzp);
    return launch_status();
}

EE_API int ee_softce_f64(const float *z, const double *t, int B, int K, double gscale, double *row_loss, double *dz,
                         void *stream) {
    if (B < 0 || K < 1 || K > 65536) return EE_ERR_SHAPE;
    if (B == 0) return EE_OK;
    if (!z || !t) return EE_ERR_NULL;
    EE_LAUNCH(softce_kernel, dim3(row_grid(B)), dim3(kBlock), 0, as_stream(stream), z, t, B, K, gscale, row_loss, dz);
    return launch_status();
}

EE_API int64_t ee_mse_num_partials(int64_t n) { return n <= 0 ? 0 : (n + kMseChunk - 1) / kMseChunk; }

EE_API int ee_mse_f32(const float *a, const float *b, int64_t n, float gscale, double *partial, float *da, void *stream) {
    if (n < 0 || ee_mse_num_partials(n) > 0x7fffffffLL) return EE_ERR_SHAPE;
    if (n == 0) return EE_OK;
    if (!a || !b) return EE_ERR_NULL;
    EE_LAUNCH(mse_kernel, dim3(static_cast<unsigned>(ee_mse_num_partials(n))), dim3(kBlock), 0, as_stream(stream), a, b, n,
                       gscale, partial, da);
    return launch_status();
}

EE_API int ee_reduce_rows_f64(const double *rows, int64_t n, double scale, double *out, void *stream) {
    if (n < 0) return EE_ERR_SHAPE;
    if (!out || (n > 0 && !rows)) return EE_ERR_NULL;
    EE_LAUNCH(reduce_rows_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), rows, n, scale, out);
    return launch_status();
}

EE_API int ee_topk_i64(const float *logits, const int64_t *labels, int B, int K, int k, int64_t *idx, int64_t *correct,
                       void *stream) {
    if (B < 0 || K < 1 || k < 1 || k > kMaxTopK || k > K) return EE_ERR_SHAPE;
    if (B > 0 && (!logits || !idx)) return EE_ERR_NULL;
    if (correct) {
        hipError_t e = hipMemsetAsync(correct, 0, sizeof(int64_t) * k, as_stream(stream));
        if (e != hipSuccess) return static_cast<int>(e);
    }
    if (B == 0) return EE_OK;
    EE_LAUNCH(topk_kernel, dim3(row_grid(B)), dim3(kBlock), 0, as_stream(stream), logits, labels, B, K, k, idx,
                       reinterpret_cast<unsigned long long *>(correct));
    return launch_status();
}

// d loss / d z1 of CrossEntropyLoss(fc2(relu(z1)), labels) in one launch (see fc_ce_grad_kernel): z1 [B][Hd], w2 [K][Hd], b2 [K] or NULL,
// labels [B] -> dz1 [B][Hd] (and the logits [B][K] if logits_out is given); gscale = 1 (reduction "sum") or 1/B ("mean").
// K <= 64, Hd <= 8192 (else EE_ERR_UNSUPPORTED).
EE_API int ee_fc_ce_grad_f32(const float *z1, const float *w2, const float *b2, const int64_t *labels, float *dz1, float *logits_out, int B, int Hd,
                             int K, float gscale, void *stream) {
    if (B < 0 || Hd < 1 || K < 1) return EE_ERR_SHAPE;
    if (K > kFcMaxK || Hd > kFcMaxH) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!z1 || !w2 || !labels || !dz1) return EE_ERR_NULL;
    if (K <= 16 && Hd <= 4 * kBlock)
        EE_LAUNCH(fc_ce_grad_small_kernel, dim3(static_cast<unsigned>(B)), dim3(kBlock), 0, as_stream(stream), z1, w2, b2, labels, Hd, K, gscale, dz1, logits_out);
    else
        EE_LAUNCH(fc_ce_grad_kernel, dim3(static_cast<unsigned>(B)), dim3(kBlock), (static_cast<size_t>(Hd) + 2 * kFcMaxK) * sizeof(float), as_stream(stream), z1, w2,
                  b2, labels, Hd, K, gscale, dz1, logits_out);
    return launch_status();
}

