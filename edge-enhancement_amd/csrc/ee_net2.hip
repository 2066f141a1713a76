// ee_net2.hip - the convolutional half of the MNIST classifier Net_2 (MNIST/models_mnist/Net2.py:13-16), forward and input gradient:
//     x = relu(max_pool2d(conv1(x), 2))                         conv1 = Conv2d(1, 32, 5)      [B,1,28,28]  -> [B,32,12,12]
//     x = relu(max_pool2d(conv2_drop(conv2(x)), 2))             conv2 = Conv2d(32, 64, 5)     [B,32,12,12] -> [B,64,4,4]
// At the reference batch (50 images) one PGD iteration of the MNIST configs spends 256 us in 35 launches of ~5 us - convolutions of
// 0.02 / 0.16 GFLOP through MIOpen (Winograd + NHWC implicit GEMM with three layout transposes and a zero fill each), bias adds, pools,
// ReLUs, dropout multiplies and their backward twins - around a 15 us hand-written front end.  Here each half is ONE launch each way:
// convolution, bias, dropout scale, 2x2 pool (ATen's scan: first maximum wins, a NaN always wins) and ReLU fused, the pooled map and a
// 2-bit argmax code the only things written; the backward gathers through the codes, applies ATen's threshold rule (the gradient passes
// unless the output is <= 0, so a NaN output passes it) and runs the transposed convolution.  Exact-f32 fma chains in a fixed order;
// parity is "logits within 1e-4" through the model tests plus per-kernel tests against ATen.  Only the attack loop (input gradient)
// runs through those; the parameter gradients of a training step: net2_conv*_wrw_kernel below (round 3).
//
// CNN-body glue, not a row of SURVEY.md section 8.
#include "ee_common.hpp"

#include <math.h>
#include <stdlib.h>

namespace {

using namespace ee;

constexpr int N2_NT = 256;
constexpr int N2_H0 = 28, N2_C1 = 32, N2_H1 = 12, N2_C2 = 64, N2_H2 = 4;

__device__ __forceinline__ float relu_keep_nan(float v) { return v > 0.0f ? v : (v != v ? v : 0.0f); }

// max over a 2x2 window in ATen's order (0,0) (0,1) (1,0) (1,1); code = index of the winner
__device__ __forceinline__ float pool4(const float v[4], int &code) {
    float best = -INFINITY;
    code = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (v[k] > best || v[k] != v[k]) {
            best = v[k];
            code = k;
        }
    return best;
}

// ---- conv1 + bias -> pool -> relu.  grid (B, 4): 8 output channels of one image per workgroup ---------------------------------------
__global__ __launch_bounds__(N2_NT) void net2_conv1_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                               float *__restrict__ a1, uint8_t *__restrict__ code1) {
    __shared__ __align__(16) float xs[N2_H0 * N2_H0];
    __shared__ float ws[8 * 25];
    __shared__ float bs[8];
    const int b = blockIdx.x, c0 = blockIdx.y * 8;
    if (threadIdx.x < N2_H0 * N2_H0 / 4)
        reinterpret_cast<float4 *>(xs)[threadIdx.x] = reinterpret_cast<const float4 *>(x + static_cast<size_t>(b) * N2_H0 * N2_H0)[threadIdx.x];
    if (threadIdx.x < 200) ws[threadIdx.x] = w[c0 * 25 + threadIdx.x];
    if (threadIdx.x < 8) bs[threadIdx.x] = bias ? bias[c0 + threadIdx.x] : 0.0f;
    __syncthreads();
    for (int o = threadIdx.x; o < 8 * N2_H1 * N2_H1; o += N2_NT) {
        const int c = o / (N2_H1 * N2_H1), r = o - c * (N2_H1 * N2_H1);
        const int py = r / N2_H1, px = r - py * N2_H1;
        float patch[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float2 *row = reinterpret_cast<const float2 *>(xs + (2 * py + i) * N2_H0 + 2 * px);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float2 t = row[j];
                patch[i][2 * j] = t.x;
                patch[i][2 * j + 1] = t.y;
            }
        }
        float v[4] = {bs[c], bs[c], bs[c], bs[c]};
#pragma unroll
        for (int ky = 0; ky < 5; ++ky)
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const float wv = ws[c * 25 + ky * 5 + kx];
                v[0] = fmaf(wv, patch[ky][kx], v[0]);
                v[1] = fmaf(wv, patch[ky][kx + 1], v[1]);
                v[2] = fmaf(wv, patch[ky + 1][kx], v[2]);
                v[3] = fmaf(wv, patch[ky + 1][kx + 1], v[3]);
            }
        int code;
        const float best = pool4(v, code);
        const size_t dst = (static_cast<size_t>(b) * N2_C1 + c0 + c) * (N2_H1 * N2_H1) + r;
        a1[dst] = relu_keep_nan(best);
        code1[dst] = static_cast<uint8_t>(code);
    }
}

// ---- conv2 + bias -> dropout scale -> pool -> relu.  grid (B, 8): 8 output channels per workgroup; the 32 input channels are split
// over the two halves of the workgroup (threads 0-127 / 128-255), partial sums meet in LDS and are added in that order ----------------
constexpr int N2_WP = 28;  // weights of one (co, ci) padded 25 -> 28: 16-byte rows
// RNG: Dropout2d's Bernoulli(keep) draw per (image, channel) is made HERE (Philox4x32-10 on the device-resident {seed, offset, ticket} state
// that Add_Square's draws use, stream id 7): element e = b * 64 + co <-> component e % 4 of counter offset + e / 4; the mask (0 / 1) is also
// written to drop_out [B,64] for the backward kernel.  The workgroup holding the last ticket advances the offset, so a replayed HIP graph
// draws fresh masks - and torch's bernoulli_ launch (5 us, once per forward of the attack loop) disappears from the captured iteration.
template <bool RNG>
__global__ __launch_bounds__(N2_NT) void net2_conv2_fwd_kernel(const float *__restrict__ a1, const float *__restrict__ w, const float *__restrict__ bias,
                                                               const float *__restrict__ drop, float keep, float *__restrict__ a2,
                                                               uint8_t *__restrict__ code2, unsigned long long *state, float *__restrict__ drop_out, int B) {
    __shared__ __align__(16) float as[N2_C1 * N2_H1 * N2_H1];      // 18 KB
    constexpr int CS = N2_C1 * N2_WP + 4;  // per-channel stride: the 8 channels of a wavefront on disjoint banks (32 * 28 floats apart they collide 8-fold)
    __shared__ __align__(16) float ws[8 * CS];                     // 28 KB
    __shared__ float part[128 * 4];
    const int b = blockIdx.x, c0 = blockIdx.y * 8;
    unsigned long long seed = 0ull, base = 0ull;
    if (RNG) seed = state[0], base = state[1];  // every lane reads the state ...
    const float4 *src = reinterpret_cast<const float4 *>(a1 + static_cast<size_t>(b) * N2_C1 * N2_H1 * N2_H1);
    for (int i = threadIdx.x; i < N2_C1 * N2_H1 * N2_H1 / 4; i += N2_NT) reinterpret_cast<float4 *>(as)[i] = src[i];
    for (int i = threadIdx.x; i < 8 * N2_C1 * 25; i += N2_NT) {  // w[c0 + co][ci][25] is one contiguous block of 8 * 32 * 25 floats
        const int pair = i / 25, k = i - pair * 25;
        ws[(pair >> 5) * CS + (pair & 31) * N2_WP + k] = w[static_cast<size_t>(c0) * N2_C1 * 25 + i];
    }
    __syncthreads();  // (waits for every outstanding load of the workgroup, the state's included)
    if (RNG && threadIdx.x == 0) {
        // ... and only then takes the workgroup's ticket: the last ticket of the grid advances the offset.  No fence: every read of the state
        // precedes its workgroup's ticket, the tickets are device-scope atomics, and the new offset only has to be visible to the NEXT launch
        // (an agent-scope fence here flushed the XCD's L2 in all 400 workgroups: +22 us per launch)
        const unsigned long long done = atomicAdd(state + 2, 1ull);
        if (done + 1ull == static_cast<unsigned long long>(B) * 8ull) {
            state[1] = base + static_cast<unsigned long long>((static_cast<long long>(B) * N2_C2 + 3) >> 2);
            state[2] = 0ull;
        }
    }
    const int half = threadIdx.x >> 7, idx = threadIdx.x & 127;
    const int co = idx >> 4, p = idx & 15, py = p >> 2, px = p & 3;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 2
    for (int cc = 0; cc < 16; ++cc) {
        const int ci = half * 16 + cc;
        float patch[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float2 *row = reinterpret_cast<const float2 *>(as + ci * (N2_H1 * N2_H1) + (2 * py + i) * N2_H1 + 2 * px);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float2 t = row[j];
                patch[i][2 * j] = t.x;
                patch[i][2 * j + 1] = t.y;
            }
        }
        float wv[N2_WP];
        const float4 *wr = reinterpret_cast<const float4 *>(ws + co * CS + ci * N2_WP);
#pragma unroll
        for (int j = 0; j < N2_WP / 4; ++j) {
            const float4 t = wr[j];
            wv[4 * j] = t.x; wv[4 * j + 1] = t.y; wv[4 * j + 2] = t.z; wv[4 * j + 3] = t.w;
        }
#pragma unroll
        for (int ky = 0; ky < 5; ++ky)
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const float t = wv[ky * 5 + kx];
                v[0] = fmaf(t, patch[ky][kx], v[0]);
                v[1] = fmaf(t, patch[ky][kx + 1], v[1]);
                v[2] = fmaf(t, patch[ky + 1][kx], v[2]);
                v[3] = fmaf(t, patch[ky + 1][kx + 1], v[3]);
            }
    }
    if (half == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) part[idx * 4 + k] = v[k];
    }
    __syncthreads();
    if (half == 0) {
        const float bv = bias ? bias[c0 + co] : 0.0f;
        float mask = 1.0f;
        if (RNG) {
            const long long e = static_cast<long long>(b) * N2_C2 + c0 + co;
            const uint4 r = Philox(seed)(base + static_cast<unsigned long long>(e >> 2), 7u);
            const unsigned rr[4] = {r.x, r.y, r.z, r.w};
            mask = u01(rr[e & 3]) < keep ? 1.0f : 0.0f;  // bernoulli_(keep): 1 with probability keep
            if (p == 0) drop_out[e] = mask;
        } else if (drop) {
            mask = drop[static_cast<size_t>(b) * N2_C2 + c0 + co];
        }
        const bool dropping = RNG || drop != nullptr;
        const float dm = mask / keep;  // Bernoulli(keep) draw (0 / 1) -> 0 or 1 / keep, as noise.div_(keep)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float s = (v[k] + part[idx * 4 + k]) + bv;
            v[k] = dropping ? s * dm : s;
        }
        int code;
        const float best = pool4(v, code);
        const size_t dst = (static_cast<size_t>(b) * N2_C2 + c0 + co) * (N2_H2 * N2_H2) + p;
        a2[dst] = relu_keep_nan(best);
        code2[dst] = static_cast<uint8_t>(code);
    }
}

// ---- conv2 on the matrix cores (round 4) -------------------------------------------------------------------------------------
// The scalar kernel above runs 64 x 800 fused multiply-adds per output pixel on the vector pipes (15.8 us per launch, bound by their issue rate).
// As a product per image: D[co][pixel] = sum_k W[co][k] * P[k][pixel], K = 32 channels x 25 taps = 800, 64 x 64 results.  A workgroup owns one
// image x 16 output channels (grid (B, 4): 200 workgroups at batch 50); its four wavefronts take 16 pixels each (two rows of the 8 x 8 result)
// and walk K in steps of four input channels at one tap: 200 v_mfma_f32_16x16x4_f32 per wavefront.  Operands straight from LDS: the image's
// a1 planes as they lie ([32][144]: the im2col operand is one ds_read_b32 at lane offset ci * 144 + oy * 12 + ox plus a compile-time tap
// offset), the 16 filter rows as they lie in memory ([co][800], row pitch 836: bank = (4 co + 25 lq) % 64 over the 16 x 4 lanes of an operand
// read - conflict-free).  Epilogue as the scalar kernel's: bias, dropout scale, 2 x 2 pool in ATen's scan order (the window's four pixels sit
// in lanes l, l + 1, l + 8, l + 9 of the accumulator tile), ReLU that keeps NaN, the argmax code.  The products are summed in another order
// than the scalar chain's: rounding-level difference (tests: 1e-5 against ATen), the same NaN footprint (a column of D depends on its own
// pixel's window only).
constexpr int N2_WPITCH = 836;  // 800 + 36

template <bool RNG>
__global__ __launch_bounds__(N2_NT) void net2_conv2_fwd_mfma_kernel(const float *__restrict__ a1, const float *__restrict__ w, const float *__restrict__ bias,
                                                                    const float *__restrict__ drop, float keep, float *__restrict__ a2,
                                                                    uint8_t *__restrict__ code2, unsigned long long *state, float *__restrict__ drop_out, int B) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __align__(16) float lds[];
    float *as = lds;                      // [32][144]
    float *ws = lds + N2_C1 * 144;        // [16][836]
    const int b = blockIdx.x, c0 = blockIdx.y * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    unsigned long long seed = 0ull, base = 0ull;
    if (RNG) seed = state[0], base = state[1];
    // every global load first, on clamped indices (a load -> store loop with a run-time trip count is one memory round trip PER ITERATION:
    // 12 of them for the filters), the LDS stores afterwards
    const float4 *src = reinterpret_cast<const float4 *>(a1 + static_cast<size_t>(b) * N2_C1 * 144);
    const float4 *wsrc = reinterpret_cast<const float4 *>(w + static_cast<size_t>(c0) * 800);  // 16 rows of 800 floats, contiguous
    constexpr int NA = (N2_C1 * 144 / 4 + N2_NT - 1) / N2_NT, NW = (16 * 200 + N2_NT - 1) / N2_NT;  // 5, 13
    float4 va[NA], vw[NW];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int i = threadIdx.x + j * N2_NT;
        va[j] = src[i < N2_C1 * 144 / 4 ? i : N2_C1 * 144 / 4 - 1];
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int i = threadIdx.x + j * N2_NT;
        vw[j] = wsrc[i < 16 * 200 ? i : 16 * 200 - 1];
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int i = threadIdx.x + j * N2_NT;
        if (i < N2_C1 * 144 / 4) reinterpret_cast<float4 *>(as)[i] = va[j];
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int i = threadIdx.x + j * N2_NT;
        if (i < 16 * 200) {
            const int r = i / 200, f = i - r * 200;
            *reinterpret_cast<float4 *>(ws + r * N2_WPITCH + 4 * f) = vw[j];
        }
    }
    __syncthreads();  // (waits for every outstanding load of the workgroup, the state's included)
    unsigned long long ticket = 0ull;
    if (RNG && threadIdx.x == 0) ticket = atomicAdd(state + 2, 1ull);  // read at the end: the round trip hides behind the products (see ee_chain.hip)
    // pixel of this lane: rows 2 wave, 2 wave + 1 of the 8 x 8 result, column l15 & 7
    const float *bp = as + lq * 144 + (2 * wave + (l15 >> 3)) * N2_H1 + (l15 & 7);
    const float *ap = ws + l15 * N2_WPITCH + lq * 25;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx)
#pragma unroll
            for (int s = 0; s < 8; ++s)  // input channels 4 s + lq
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(4 * s) * 25 + ky * 5 + kx], bp[(4 * s) * 144 + ky * N2_H1 + kx], acc, 0, 0, 0);
    // D[row = 4 lq + r = output channel][column = l15 = pixel]: bias, dropout scale, then the 2 x 2 pool across lanes l, l + 1, l + 8, l + 9
    float mask4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    const bool dropping = RNG || drop != nullptr;
    if (RNG) {
        const long long e = static_cast<long long>(b) * N2_C2 + c0 + 4 * lq;  // elements e .. e + 3: one counter
        const uint4 r = Philox(seed)(base + static_cast<unsigned long long>(e >> 2), 7u);
        const unsigned rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) mask4[k] = u01(rr[k]) < keep ? 1.0f : 0.0f;
        if (wave == 0 && l15 == 0) *reinterpret_cast<float4 *>(drop_out + e) = make_float4(mask4[0], mask4[1], mask4[2], mask4[3]);
    } else if (drop) {
        const float4 m = *reinterpret_cast<const float4 *>(drop + static_cast<size_t>(b) * N2_C2 + c0 + 4 * lq);
        mask4[0] = m.x, mask4[1] = m.y, mask4[2] = m.z, mask4[3] = m.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = c0 + 4 * lq + r;
        const float s = acc[r] + (bias ? bias[co] : 0.0f);
        const float v0 = dropping ? s * (mask4[r] / keep) : s;
        float v[4];
        v[0] = v0;
        v[1] = __shfl_down(v0, 1, 16);
        v[2] = __shfl_down(v0, 8, 16);
        v[3] = __shfl_down(v0, 9, 16);
        if (l15 < 8 && !(l15 & 1)) {
            int code;
            const float best = pool4(v, code);
            const size_t dst = (static_cast<size_t>(b) * N2_C2 + co) * (N2_H2 * N2_H2) + wave * N2_H2 + (l15 >> 1);
            a2[dst] = relu_keep_nan(best);
            code2[dst] = static_cast<uint8_t>(code);
        }
    }
    if (RNG && threadIdx.x == 0 && ticket + 1ull == static_cast<unsigned long long>(B) * 4ull) {  // the last ticket of the grid advances the offset
        state[1] = base + static_cast<unsigned long long>((static_cast<long long>(B) * N2_C2 + 3) >> 2);
        state[2] = 0ull;
    }
}

// ---- backward of the second half: d a2 -> d a1.  G[co][8][8] = the un-pooled gradient (threshold rule, argmax position, dropout scale),
// then the transposed convolution d a1[ci][y][x] = sum_co sum_{ky,kx} G[co][y - ky][x - kx] * w[co][ci][ky][kx].
// grid (B, 8): 4 input channels per workgroup; a thread owns one row y of one channel for an eighth of the output channels in flight
// (384 of 512 threads work; 32 output channels at a time in LDS), the eight partial rows meet in LDS and are added in group order --
constexpr int N2_GW = 20;  // G rows: 8 -> 16 columns (4 zeros either side) + 4 of padding (a wavefront's 12 rows on disjoint banks); 16 rows: [co][16][20]
// Round 3: 512 lanes, EIGHT groups of 48 (channel, row) lanes, each taking an eighth of the output channels (round 2: 256 lanes, four
// quarters).  The kernel is bound by the issue rate of its scalar FMAs, and a wavefront alone on a SIMD issues one vector instruction
// per 4 cycles where two issue one per 2: with 3 active wavefronts per workgroup the SIMDs ran at half rate.
constexpr int N2_BT = 512, N2_BG = 8;
__global__ __launch_bounds__(N2_BT) void net2_conv2_bwd_kernel(const float *__restrict__ da2, const float *__restrict__ a2, const uint8_t *__restrict__ code2,
                                                               const float *__restrict__ drop, float keep, const float *__restrict__ w,
                                                               float *__restrict__ da1) {
    constexpr int GP = 16 * N2_GW;                             // one channel's frame
    __shared__ __align__(16) float G[32 * GP];                 // 40 KB
    __shared__ __align__(16) float ws[N2_C2 * 4 * N2_WP];      // w[co][ci0 .. ci0+3][25 -> 28]: 28 KB
    const int b = blockIdx.x, ci0 = blockIdx.y * 4;
    for (int i = threadIdx.x; i < N2_C2 * 4 * 25; i += N2_BT) {
        const int co = i / 100, rem = i - co * 100, cl = rem / 25, k = rem - cl * 25;
        ws[(co * 4 + cl) * N2_WP + k] = w[(static_cast<size_t>(co) * N2_C1 + ci0 + cl) * 25 + k];
    }
    float acc[N2_H1];
#pragma unroll
    for (int xx = 0; xx < N2_H1; ++xx) acc[xx] = 0.0f;
    const int q = threadIdx.x / 48, rem = threadIdx.x - q * 48, cl = rem / N2_H1, y = rem - cl * N2_H1;  // threads 0..383: (group, channel, row)
    constexpr int CPG = 32 / N2_BG;  // output channels per group and half
    for (int h = 0; h < 2; ++h) {
        __syncthreads();  // the previous half's rows have been read (and, first time round, nothing is pending)
        for (int i = threadIdx.x; i < 32 * GP / 4; i += N2_BT) reinterpret_cast<float4 *>(G)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        __syncthreads();
        for (int i = threadIdx.x; i < 32 * 16; i += N2_BT) {
            const int cc = i >> 4, p = i & 15, wy = p >> 2, wx = p & 3, co = 32 * h + cc;
            const size_t src = (static_cast<size_t>(b) * N2_C2 + co) * 16 + p;
            float g = da2[src];
            if (a2[src] <= 0.0f) g = 0.0f;  // ATen's threshold_backward: the gradient passes unless the output is <= 0
            if (drop) g *= drop[static_cast<size_t>(b) * N2_C2 + co] / keep;
            const int cd = code2[src];
            G[cc * GP + (4 + 2 * wy + (cd >> 1)) * N2_GW + 4 + 2 * wx + (cd & 1)] = g;
        }
        __syncthreads();
        if (threadIdx.x < 48 * N2_BG) {
            for (int cc = 0; cc < CPG; ++cc) {
                const int gl = q * CPG + cc, co = 32 * h + gl;
                float wv[N2_WP];
                const float4 *wr = reinterpret_cast<const float4 *>(ws + (co * 4 + cl) * N2_WP);
#pragma unroll
                for (int j = 0; j < N2_WP / 4; ++j) {
                    const float4 t = wr[j];
                    wv[4 * j] = t.x; wv[4 * j + 1] = t.y; wv[4 * j + 2] = t.z; wv[4 * j + 3] = t.w;
                }
#pragma unroll
                for (int ky = 0; ky < 5; ++ky) {
                    // G row y - ky of the 8x8 map = padded row 4 + y - ky; its padded columns 0 .. 15 cover x - kx for x in 0..11, kx in 0..4
                    const float4 *gr = reinterpret_cast<const float4 *>(G + gl * GP + (4 + y - ky) * N2_GW);
                    float row[16];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 t = gr[j];
                        row[4 * j] = t.x; row[4 * j + 1] = t.y; row[4 * j + 2] = t.z; row[4 * j + 3] = t.w;
                    }
#pragma unroll
                    for (int kx = 0; kx < 5; ++kx) {
                        const float t = wv[ky * 5 + kx];
#pragma unroll
                        for (int xx = 0; xx < N2_H1; ++xx) acc[xx] = fmaf(t, row[4 + xx - kx], acc[xx]);
                    }
                }
            }
        }
    }
    __syncthreads();  // G is dead: the partial rows of groups 1 .. 7 go there
    float *part = G;
    if (threadIdx.x < 48 * N2_BG && q > 0) {
#pragma unroll
        for (int xx = 0; xx < N2_H1; ++xx) part[((q - 1) * 48 + rem) * N2_H1 + xx] = acc[xx];
    }
    __syncthreads();
    if (threadIdx.x < 48) {  // added in group order: a fixed summation order, reproducible run to run
        float *dst = da1 + ((static_cast<size_t>(b) * N2_C1 + ci0 + cl) * N2_H1 + y) * N2_H1;
#pragma unroll
        for (int xx = 0; xx < N2_H1; ++xx) {
            float sacc = acc[xx];
#pragma unroll
            for (int g = 0; g < N2_BG - 1; ++g) sacc += part[(g * 48 + rem) * N2_H1 + xx];
            dst[xx] = sacc;
        }
    }
}

// ---- conv2's backward-data on the matrix cores (round 4) ------------------------------------------------------------------------------------------
// The un-pooled gradient G[co][8][8] has ONE non-zero per 2 x 2 window (the argmax the forward recorded): the scalar kernel above multiplies the
// zeros too (64 x 25 fused multiply-adds per output element, 22 us).  Group the non-zeros by their position inside the window, d = (dy, dx):
//     P_d[w][ci][tap] = sum_co  W[co][ci][tap] * G_d[co][w],      G_d[co][w] = g[co][w] if the window's argmax sits at d, else 0
// is a product on the matrix cores - M = (ci, tap), K = 64 output channels, N = 4 classes x 16 windows = 64 columns - and
//     d a1[ci][y][x] = sum over (ky, kx) of  P_d[w][ci][ky][kx]   with  (2 wy + dy, 2 wx + dx) = (y - ky, x - kx)
// is a gather of at most 25 of its entries per output element, in a fixed order (bit-reproducible).  A workgroup owns one image x 8 input
// channels (grid (B, 4)): M = 200 rows in 13 tiles, each wavefront one column tile: 13 x 16 = 208 MFMAs per wavefront against 1600 scalar
// fused multiply-adds per output element before.  Filters as they lie in memory ([co][8 ci x 25 taps], row pitch 208: conflict-free A reads),
// G_d as [co][80]; P overlays the filters once they have been consumed.  NaN / inf footprint: a column of the product depends on its own window
// only, and the zeros of the other classes meet finite filters - what the scalar kernel's zeros do.
// -DEE_NET2_SKIP=<bits> (scripts/net2_phases.py builds its own copy; never the product): 1: no products, 2: no gather (zeros stored), 4: no P store
#ifndef EE_NET2_SKIP
#define EE_NET2_SKIP 0
#endif
constexpr int N2_BWP = 208, N2_BGP = 80, N2_BPP = 201;  // P's column pitch: odd, so that the gather's 64 lanes (64 different columns) hit 64 banks

__global__ __launch_bounds__(N2_NT) void net2_conv2_bwd_mfma_kernel(const float *__restrict__ da2, const float *__restrict__ a2, const uint8_t *__restrict__ code2,
                                                                    const float *__restrict__ drop, float keep, const float *__restrict__ w,
                                                                    float *__restrict__ da1) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __align__(16) float lds[];
    float *ws = lds;                       // [64 co][208]: w[co][ci0 .. ci0 + 7][25]; later P [64 columns][208]
    float *gs = lds + N2_C2 * N2_BWP;      // [64 co][80]: column n = 16 d + window
    const int b = blockIdx.x, ci0 = blockIdx.y * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    // every global load first (see the forward kernel): 64 filter rows of 200 contiguous floats = 3200 float4, 13 per lane; the gradient,
    // activation, code and dropout draw of the image's 1024 pooled elements, 4 per lane
    constexpr int NW = (N2_C2 * 50 + N2_NT - 1) / N2_NT, NG = N2_C2 * 16 / N2_NT;  // 13, 4
    float4 vw[NW];
    float gv[NG], av[NG], dv[NG];
    int cv[NG];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int i0 = threadIdx.x + j * N2_NT, i = i0 < N2_C2 * 50 ? i0 : N2_C2 * 50 - 1;
        const int co = i / 50, f = i - co * 50;
        vw[j] = *reinterpret_cast<const float4 *>(w + (static_cast<size_t>(co) * N2_C1 + ci0) * 25 + 4 * f);
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int i = threadIdx.x + j * N2_NT;
        const size_t src = static_cast<size_t>(b) * N2_C2 * 16 + i;
        gv[j] = da2[src];
        av[j] = a2[src];
        cv[j] = code2[src];
        dv[j] = drop ? drop[static_cast<size_t>(b) * N2_C2 + (i >> 4)] : 1.0f;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int i = threadIdx.x + j * N2_NT;
        if (i < N2_C2 * 50) {
            const int co = i / 50, f = i - co * 50;
            *reinterpret_cast<float4 *>(ws + co * N2_BWP + 4 * f) = vw[j];
        }
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int i = threadIdx.x + j * N2_NT, co = i >> 4, p = i & 15;
        float g = gv[j];
        if (av[j] <= 0.0f) g = 0.0f;  // ATen's threshold_backward: the gradient passes unless the output is <= 0
        if (drop) g *= dv[j] / keep;
#pragma unroll
        for (int dcl = 0; dcl < 4; ++dcl) gs[co * N2_BGP + 16 * dcl + p] = dcl == cv[j] ? g : 0.0f;
    }
    __syncthreads();
    f32x4 acc[13];
#pragma unroll
    for (int mt = 0; mt < 13; ++mt) acc[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    {
        const float *ap = ws + lq * N2_BWP + l15, *bp = gs + lq * N2_BGP + 16 * wave + l15;
#pragma unroll 4
        for (int s_ = 0; s_ < ((EE_NET2_SKIP & 1) ? 0 : 16); ++s_) {  // output channels 4 s + lq
            const float bv = bp[4 * s_ * N2_BGP];
#pragma unroll
            for (int mt = 0; mt < 13; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s_ * N2_BWP + 16 * mt], bv, acc[mt], 0, 0, 0);
        }
    }
    __syncthreads();  // every filter has been consumed: P takes their place
    float *ps = ws;   // [64 columns][pitch 201: 200 rows used] (at the filters' pitch of 208 the gather below was a 16-way bank conflict)
#pragma unroll
    for (int mt = 0; mt < 13; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (!(EE_NET2_SKIP & 4) && 16 * mt + 4 * lq + r < 200) ps[(16 * wave + l15) * N2_BPP + 16 * mt + 4 * lq + r] = acc[mt][r];
    __syncthreads();
    for (int o = threadIdx.x; o < 8 * N2_H1 * N2_H1; o += N2_NT) {
        const int cl = o / (N2_H1 * N2_H1), px = o - cl * (N2_H1 * N2_H1), y = px / N2_H1, x = px - y * N2_H1;
        // column n = 16 (2 (Y & 1) + (X & 1)) + 4 (Y >> 1) + (X >> 1) of P, row cl * 25 + 5 ky + kx: the index splits into a part per ky and a
        // part per kx; reads are unconditional on clamped positions and SELECTED (a branch per tap made the gather 7.7 us of the kernel's 16)
        int colp[5], rowp[5];
        bool cok[5], rok[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int X = x - k, Y = y - k, Xc = clampi(X, 0, 7), Yc = clampi(Y, 0, 7);
            cok[k] = X >= 0 && X <= 7;
            rok[k] = Y >= 0 && Y <= 7;
            colp[k] = (16 * (Xc & 1) + (Xc >> 1)) * N2_BPP + k;
            rowp[k] = (32 * (Yc & 1) + 4 * (Yc >> 1)) * N2_BPP + cl * 25 + 5 * k;
        }
        float sum = 0.0f;
#pragma unroll
        for (int ky = 0; ky < ((EE_NET2_SKIP & 2) ? 0 : 5); ++ky)
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const float v = ps[rowp[ky] + colp[kx]];
                if (rok[ky] && cok[kx]) sum += v;
            }
        da1[(static_cast<size_t>(b) * N2_C1 + ci0 + cl) * (N2_H1 * N2_H1) + px] = sum;
    }
}

// ---- backward of the first half: d a1 -> d x.  G1[c][24][24] un-pooled (threshold rule, argmax), d x[y][x] = sum_c sum_k G1[c][y-ky][x-kx] w1[c][ky][kx].
// grid (B, 4): 7 rows of one image per workgroup; a thread owns 4 neighbouring pixels of a row for a quarter of the channels.  Round 3:
// ALL 32 channels' zero-padded frames [c][11][36] sit in LDS at once (50 KB) - one zero pass, one scatter whose three global loads per element
// are unconditional and all in flight together, one barrier - where round 2 ran four rounds of 8 channels with three barriers each and
// the gradient / activation loads behind a predicate that depended on the code load (eight serialised memory round trips: 19.6 us for
// 31 MFLOP).  The same fma chains in the same order per thread (channels c0 + 2 hc + cc for c0 = 0, 8, 16, 24): bit-identical results;
// the four quarters meet in LDS and are added in quarter order ------------------------------------------------------------------------
__global__ __launch_bounds__(N2_NT) void net2_conv1_bwd_kernel(const float *__restrict__ da1, const float *__restrict__ a1, const uint8_t *__restrict__ code1,
                                                               const float *__restrict__ w, float *__restrict__ dx) {
    constexpr int ROWS = 7, FR = ROWS + 4, FW = 36, NQ = ROWS * 7;  // 49 pixel quads
    constexpr int NE = N2_C1 * N2_H1 * N2_H1, EPT = NE / N2_NT;     // 4608 pooled elements of the image, 18 per thread
    static_assert(NE % N2_NT == 0, "elements per thread");
    __shared__ __align__(16) float G[N2_C1 * FR * FW];  // 50.7 KB: [c][4 + (y - y0) - ky][4 + x - kx]
    __shared__ float ws[N2_C1 * 25];
    __shared__ __align__(16) float part[3 * NQ * 4];
    const int b = blockIdx.x, y0 = blockIdx.y * ROWS;
    // every load of the scatter first (registers), the frame zeroed while they travel
    float gv[EPT], av[EPT];
    int cv[EPT];
    const size_t base = static_cast<size_t>(b) * NE;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const size_t src = base + threadIdx.x + N2_NT * k;
        cv[k] = code1[src];
        gv[k] = da1[src];
        av[k] = a1[src];
    }
    for (int i = threadIdx.x; i < N2_C1 * 25; i += N2_NT) ws[i] = w[i];
    for (int i = threadIdx.x; i < N2_C1 * FR * FW / 4; i += N2_NT) reinterpret_cast<float4 *>(G)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int i = threadIdx.x + N2_NT * k;
        const int c = i / (N2_H1 * N2_H1), r = i - c * (N2_H1 * N2_H1), py = r / N2_H1, px = r - py * N2_H1;
        const int lr = 2 * py + (cv[k] >> 1) - y0 + 4;  // frame row of the un-pooled position
        if (lr >= 0 && lr < FR) G[(c * FR + lr) * FW + 4 + 2 * px + (cv[k] & 1)] = av[k] <= 0.0f ? 0.0f : gv[k];
    }
    __syncthreads();
    const int hc = threadIdx.x / NQ, quad = threadIdx.x - hc * NQ, ry = quad / 7, x0 = 4 * (quad - ry * 7);  // threads 0..195: (channel quarter, quad)
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (threadIdx.x < 4 * NQ) {
        for (int c0 = 0; c0 < N2_C1; c0 += 8) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int c = c0 + hc * 2 + cc;
                const float *wc = ws + c * 25;
#pragma unroll
                for (int ky = 0; ky < 5; ++ky) {
                    const float4 *gr = reinterpret_cast<const float4 *>(G + (c * FR + 4 + ry - ky) * FW + x0);
                    const float4 t0 = gr[0], t1 = gr[1];  // frame columns x0 .. x0 + 7 = pixels x0 - 4 .. x0 + 3
                    const float row[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                    for (int kx = 0; kx < 5; ++kx) {
                        const float t = wc[ky * 5 + kx];
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] = fmaf(t, row[4 + k - kx], acc[k]);
                    }
                }
            }
        }
    }
    if (threadIdx.x >= NQ && threadIdx.x < 4 * NQ) reinterpret_cast<float4 *>(part)[(hc - 1) * NQ + quad] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (threadIdx.x < NQ) {
        const float4 *pp = reinterpret_cast<const float4 *>(part);
        const float4 o1 = pp[quad], o2 = pp[NQ + quad], o3 = pp[2 * NQ + quad];
        *reinterpret_cast<float4 *>(dx + (static_cast<size_t>(b) * N2_H0 + y0 + ry) * N2_H0 + x0) =
            make_float4(((acc[0] + o1.x) + o2.x) + o3.x, ((acc[1] + o1.y) + o2.y) + o3.y, ((acc[2] + o1.z) + o2.z) + o3.z, ((acc[3] + o1.w) + o2.w) + o3.w);
    }
}

// ---- parameter gradients (the training step's backward: once per 40 attack iterations) -----------------------------------------------------
// Through the pools only one position of every 2x2 window carries a gradient (the argmax the forward recorded), so
//     dW2[co][ci][ky][kx] = sum over images and the 16 windows of  g2 * a1[ci][oy + ky][ox + kx],   g2 = da2 (0 where a2 <= 0) * drop / keep,
//     (oy, ox) = the window's argmax position in conv2's 8x8 output;       db2[co] = sum g2;        dW1 / db1 likewise from da1, a1's mask, code1, x.
// The gather position depends on the OUTPUT channel, so this is no GEMM: plain fma chains.  A workgroup takes ten images at once (one round of
// loads, one barrier) for one output channel (and 8 input channels of conv2) and adds them in order; the groups of ten meet in a partial buffer
// that net2_sum_kernel adds in order: a fixed summation order, bit-reproducible.  (Before round 3 a pass that needed these gradients re-ran the
// stock ATen / MIOpen sequence: 25 launches and 0.27 ms of the MNIST training step.)
constexpr int N2_WI = 10;                                   // images per workgroup
constexpr int N2_PG = 64 * 32 * 25 + 32 * 25 + 64 + 32;     // one group's partials / the result: [dw2 | dw1 | db2 | db1] = 52096 floats

__global__ __launch_bounds__(N2_NT) void net2_conv2_wrw_kernel(const float *__restrict__ da2, const float *__restrict__ a2, const uint8_t *__restrict__ code2,
                                                               const float *__restrict__ drop, float keep, const float *__restrict__ a1,
                                                               float *__restrict__ out, int B) {
    __shared__ __align__(16) float A[N2_WI * 8 * 144];  // 46 KB
    __shared__ float gs[N2_WI * 16];
    __shared__ int off[N2_WI * 16];
    const int co = blockIdx.x, ci0 = blockIdx.y * 8, b0 = blockIdx.z * N2_WI;
    const int nb = B - b0 < N2_WI ? B - b0 : N2_WI;
    for (int i = threadIdx.x; i < nb * 288; i += N2_NT) {
        const int img = i / 288, f = i - img * 288;
        reinterpret_cast<float4 *>(A)[i] = reinterpret_cast<const float4 *>(a1 + (static_cast<size_t>(b0 + img) * N2_C1 + ci0) * 144)[f];
    }
    if (static_cast<int>(threadIdx.x) < nb * 16) {
        const int img = threadIdx.x >> 4, p = threadIdx.x & 15;
        const size_t src = (static_cast<size_t>(b0 + img) * N2_C2 + co) * 16 + p;
        float g = da2[src];
        if (a2[src] <= 0.0f) g = 0.0f;  // ATen's threshold_backward
        if (drop) g *= drop[static_cast<size_t>(b0 + img) * N2_C2 + co] / keep;
        const int cd = code2[src];
        gs[threadIdx.x] = g;
        off[threadIdx.x] = img * (8 * 144) + (2 * (p >> 2) + (cd >> 1)) * 12 + 2 * (p & 3) + (cd & 1);
    }
    __syncthreads();
    float *o = out + static_cast<size_t>(blockIdx.z) * N2_PG;
    if (threadIdx.x < 200) {
        const int cl = threadIdx.x / 25, k = threadIdx.x - 25 * cl;
        const float *ap = A + cl * 144 + (k / 5) * 12 + k % 5;
        float acc = 0.0f;
        for (int i = 0; i < nb * 16; ++i) acc = fmaf(gs[i], ap[off[i]], acc);
        o[(static_cast<size_t>(co) * N2_C1 + ci0 + cl) * 25 + k] = acc;
    } else if (threadIdx.x == 200 && blockIdx.y == 0) {
        float bsum = 0.0f;
        for (int i = 0; i < nb * 16; ++i) bsum += gs[i];
        o[64 * 32 * 25 + 32 * 25 + co] = bsum;
    }
}

// (output channel of conv1, group of ten images); lane (tap k, slice j of the 144 windows: j, j + 10, ...); the ten slices meet in LDS in order
__global__ __launch_bounds__(N2_NT) void net2_conv1_wrw_kernel(const float *__restrict__ da1, const float *__restrict__ a1, const uint8_t *__restrict__ code1,
                                                               const float *__restrict__ x, float *__restrict__ out, int B) {
    __shared__ __align__(16) float xs[N2_WI * N2_H0 * N2_H0];  // 31 KB
    __shared__ float gs[N2_WI * 144];
    __shared__ int off[N2_WI * 144];
    __shared__ float red[10][26];
    const int co = blockIdx.x, b0 = blockIdx.y * N2_WI;
    const int nb = B - b0 < N2_WI ? B - b0 : N2_WI;
    for (int i = threadIdx.x; i < nb * 196; i += N2_NT) reinterpret_cast<float4 *>(xs)[i] = reinterpret_cast<const float4 *>(x + static_cast<size_t>(b0) * 784)[i];
    for (int i = threadIdx.x; i < nb * 144; i += N2_NT) {
        const int img = i / 144, p = i - img * 144, py = p / N2_H1, px = p - py * N2_H1;
        const size_t src = (static_cast<size_t>(b0 + img) * N2_C1 + co) * 144 + p;
        const int cd = code1[src];
        gs[i] = a1[src] <= 0.0f ? 0.0f : da1[src];
        off[i] = img * 784 + (2 * py + (cd >> 1)) * N2_H0 + 2 * px + (cd & 1);
    }
    __syncthreads();
    const int j = threadIdx.x / 25, k = threadIdx.x - 25 * j;
    if (threadIdx.x < 250) {
        const float *xp = xs + (k / 5) * N2_H0 + k % 5;
        float acc = 0.0f;
        for (int img = 0; img < nb; ++img)
            for (int p = j; p < 144; p += 10) acc = fmaf(gs[img * 144 + p], xp[off[img * 144 + p]], acc);
        red[j][k] = acc;
    }
    __syncthreads();
    float *o = out + static_cast<size_t>(blockIdx.y) * N2_PG + 64 * 32 * 25;
    if (threadIdx.x < 25) {
        float t = red[0][threadIdx.x];
#pragma unroll
        for (int jj = 1; jj < 10; ++jj) t += red[jj][threadIdx.x];
        o[co * 25 + threadIdx.x] = t;
    } else if (threadIdx.x == 250) {
        float bsum = 0.0f;
        for (int i = 0; i < nb * 144; ++i) bsum += gs[i];
        o[32 * 25 + 64 + co] = bsum;
    }
}

__global__ __launch_bounds__(N2_NT) void net2_sum_kernel(const float4 *__restrict__ part, float4 *__restrict__ out, int G) {
    const int i = blockIdx.x * N2_NT + threadIdx.x;
    if (i >= N2_PG / 4) return;
    float4 a = part[i];
    for (int g = 1; g < G; ++g) {
        const float4 v = part[static_cast<size_t>(g) * (N2_PG / 4) + i];
        a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
    }
    out[i] = a;
}

// EEADV_NET2_SCALAR=1: conv2 forward / backward-data on the scalar kernels of rounds 1-3 (A/B, and the yardstick of the matrix-core kernels'
// tests); read per call
inline bool net2_mfma_on() {
    const char *e = getenv("EEADV_NET2_SCALAR");
    return !(e && e[0] == '1');
}

}  // namespace

EE_API int ee_net2_conv_fwd_f32(const float *x, const float *w1, const float *b1, const float *w2, const float *b2, const float *drop, float keep,
                                uint64_t *draw_state, float *drop_out, float *a1, uint8_t *code1, float *a2, uint8_t *code2, int B, void *stream) {
    if (B < 0) return EE_ERR_SHAPE;
    if (B == 0) return EE_OK;
    if (!x || !w1 || !w2 || !a1 || !code1 || !a2 || !code2) return EE_ERR_NULL;
    const bool rng = !drop && draw_state;
    if ((drop || rng) && !(keep > 0.0f)) return EE_ERR_SHAPE;
    if (rng && !drop_out) return EE_ERR_NULL;
    if (!aligned16(x) || !aligned16(a1)) return EE_ERR_ALIGN;
    hipStream_t st = as_stream(stream);
    EE_LAUNCH(net2_conv1_fwd_kernel, dim3(static_cast<unsigned>(B), 4), dim3(N2_NT), 0, st, x, w1, b1, a1, code1);
    if (net2_mfma_on()) {
        constexpr size_t bytes = (N2_C1 * 144 + 16 * N2_WPITCH) * sizeof(float);  // 72 KB: above the static limit
        static int ok = (hipFuncSetAttribute(reinterpret_cast<const void *>(net2_conv2_fwd_mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)) == hipSuccess) &&
                        (hipFuncSetAttribute(reinterpret_cast<const void *>(net2_conv2_fwd_mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)) == hipSuccess);
        if (!ok) return EE_ERR_UNSUPPORTED;
        if (!aligned16(w2) || (drop && !aligned16(drop)) || (drop_out && !aligned16(drop_out))) return EE_ERR_ALIGN;
        if (rng)
            EE_LAUNCH(net2_conv2_fwd_mfma_kernel<true>, dim3(static_cast<unsigned>(B), 4), dim3(N2_NT), bytes, st, a1, w2, b2, drop, keep, a2, code2,
                      reinterpret_cast<unsigned long long *>(draw_state), drop_out, B);
        else
            EE_LAUNCH(net2_conv2_fwd_mfma_kernel<false>, dim3(static_cast<unsigned>(B), 4), dim3(N2_NT), bytes, st, a1, w2, b2, drop, keep, a2, code2,
                      static_cast<unsigned long long *>(nullptr), static_cast<float *>(nullptr), B);
        return launch_status();
    }
    if (rng)
        EE_LAUNCH(net2_conv2_fwd_kernel<true>, dim3(static_cast<unsigned>(B), 8), dim3(N2_NT), 0, st, a1, w2, b2, drop, keep, a2, code2,
                  reinterpret_cast<unsigned long long *>(draw_state), drop_out, B);
    else
        EE_LAUNCH(net2_conv2_fwd_kernel<false>, dim3(static_cast<unsigned>(B), 8), dim3(N2_NT), 0, st, a1, w2, b2, drop, keep, a2, code2,
                  static_cast<unsigned long long *>(nullptr), static_cast<float *>(nullptr), B);
    return launch_status();
}

EE_API int ee_net2_conv_bwd_f32(const float *da2, const float *a2, const uint8_t *code2, const float *drop, float keep, const float *w2,
                                const float *a1, const uint8_t *code1, const float *w1, float *da1, float *dx, int B, void *stream) {
    if (B < 0) return EE_ERR_SHAPE;
    if (B == 0) return EE_OK;
    if (!da2 || !a2 || !code2 || !w2 || !a1 || !code1 || !w1 || !da1) return EE_ERR_NULL;
    if (dx && !aligned16(dx)) return EE_ERR_ALIGN;
    hipStream_t st = as_stream(stream);
    if (net2_mfma_on() && aligned16(w2)) {
        constexpr size_t bytes = (N2_C2 * N2_BWP + N2_C2 * N2_BGP) * sizeof(float);  // 73.7 KB: above the static limit
        static int ok = hipFuncSetAttribute(reinterpret_cast<const void *>(net2_conv2_bwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            static_cast<int>(bytes)) == hipSuccess;
        if (!ok) return EE_ERR_UNSUPPORTED;
        EE_LAUNCH(net2_conv2_bwd_mfma_kernel, dim3(static_cast<unsigned>(B), 4), dim3(N2_NT), bytes, st, da2, a2, code2, drop, keep, w2, da1);
    } else {
        EE_LAUNCH(net2_conv2_bwd_kernel, dim3(static_cast<unsigned>(B), 8), dim3(N2_BT), 0, st, da2, a2, code2, drop, keep, w2, da1);
    }
    if (dx)  // NULL: only da1 is wanted (a training step whose input needs no gradient; ee_net2_conv_wrw_f32 reads da1)
        EE_LAUNCH(net2_conv1_bwd_kernel, dim3(static_cast<unsigned>(B), 4), dim3(N2_NT), 0, st, da1, a1, code1, w1, dx);
    return launch_status();
}

// Parameter gradients of the same two halves: da2 [B,64,4,4] (the gradient of a2) and da1 [B,32,12,12] (the gradient of a1: what
// ee_net2_conv_bwd_f32 left in its scratch argument) -> out = [ dw2 [64,32,5,5] | dw1 [32,1,5,5] | db2 [64] | db1 [32] ] = 52096 floats, overwritten;
// workspace: ee_net2_conv_wrw_workspace_floats(B) floats (0: none needed).  Images are added in order: bit-reproducible.
EE_API int64_t ee_net2_conv_wrw_workspace_floats(int B) {
    const int G = (B + N2_WI - 1) / N2_WI;
    return G > 1 ? static_cast<int64_t>(G) * N2_PG : 0;
}

EE_API int ee_net2_conv_wrw_f32(const float *x, const float *a1, const uint8_t *code1, const float *da1, const float *a2, const uint8_t *code2,
                                const float *da2, const float *drop, float keep, float *out, float *workspace, int B, void *stream) {
    if (B < 0) return EE_ERR_SHAPE;
    if (!out) return EE_ERR_NULL;
    hipStream_t st = as_stream(stream);
    if (B == 0) return static_cast<int>(hipMemsetAsync(out, 0, sizeof(float) * N2_PG, st));
    if (!x || !a1 || !code1 || !da1 || !a2 || !code2 || !da2) return EE_ERR_NULL;
    if (drop && !(keep > 0.0f)) return EE_ERR_SHAPE;
    const int G = (B + N2_WI - 1) / N2_WI;
    if (G > 1 && !workspace) return EE_ERR_NULL;
    if (!aligned16(x) || !aligned16(a1) || !aligned16(out) || (G > 1 && !aligned16(workspace))) return EE_ERR_ALIGN;
    float *dst = G > 1 ? workspace : out;
    EE_LAUNCH(net2_conv2_wrw_kernel, dim3(N2_C2, N2_C1 / 8, static_cast<unsigned>(G)), dim3(N2_NT), 0, st, da2, a2, code2, drop, keep, a1, dst, B);
    EE_LAUNCH(net2_conv1_wrw_kernel, dim3(N2_C1, static_cast<unsigned>(G)), dim3(N2_NT), 0, st, da1, a1, code1, x, dst, B);
    if (G > 1) EE_LAUNCH(net2_sum_kernel, dim3((N2_PG / 4 + N2_NT - 1) / N2_NT), dim3(N2_NT), 0, st, reinterpret_cast<const float4 *>(workspace), reinterpret_cast<float4 *>(out), G);
    return launch_status();
}
