// ee_s2.hip - Conv2d(3x3, stride 2, padding 1, bias=False) on small maps (the first convolution of ResNet-18's layer3 / layer4 at 64x64
// inputs: 8x8 -> 4x4 and 4x4 -> 2x2, resnet.py:26-31, :132-137), forward and backward-data, on the f32 matrix cores.
//
// Why a kernel of its own: these are GEMMs with few pixels (1600 / 400 columns) and long reductions (1152 / 2304), which MIOpen serves
// with an NHWC implicit GEMM between two layout transposes and a zero fill (5 launches, 28-39 us un-profiled), and which the direct
// kernels of ee_conv.hip tile badly.  Here a workgroup owns 32 result channels x 32 pixels (two v_mfma_f32_16x16x4
// column blocks: two images of a 4x4 result, eight of a 2x2 one) and its four wavefronts SPLIT THE REDUCTION: per round of 16
// reduction channels, wavefront w multiplies channel quad w through all nine taps (2 x 2 accumulator tiles, 36 MFMAs, operands read
// once per two products), and the four partial sums meet in LDS at the end.
//   forward       the round's inputs are scattered into LDS tap by tap (an im2col of 9 x 16 pixels per channel, zero where a tap leaves
//                 the map), so a B operand is one conflict-free ds_read_b32 at a compile-time offset;
//   backward-data the four parities of dx are four small stride-1 correlations of dy (1, 2, 2 and 4 taps: no multiplication by the
//                 zeros a transposed stride-2 convolution inserts): dy goes to LDS as its four one-pixel shifts, the accumulators are
//                 kept per parity class and interleaved on the way out.
// The filters arrive rearranged (`w9`, built once per weight version by the host: functional._rearranged kinds "s2m_f" / "s2m_b") as
// [result block of 32][round][tap][quad][half of 16][m][k] - the exact order the A operands are read in, so staging them is a straight
// 16-byte copy.  Exact f32 products; the summation order differs from MIOpen's (tolerance-level parity, like every convolution here).
//
// CNN-body glue, not a row of SURVEY.md section 8.
#include <cstdlib>
#include <cstring>

#include "ee_common.hpp"

namespace {

using namespace ee;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int S2_NT = 256, S2_CK = 16, S2_RB = 32;
constexpr int S2_WF = 9 * 4 * 2 * 64;  // one round's filter slab: [tap][quad][half][16 m][4 k] = 4608 floats
constexpr int S2_RS = 36;              // row stride of the partial-sum exchange: the four k of a wavefront on disjoint banks

struct S2Dims {
    int B, KC, RC;  // reduction channels (Cin forward, Cout backward), result channels
};

// H = the LARGE map's side (x forward, dx backward); the small map is OH x OH
template <int H>
struct S2Geo {
    static constexpr int OH = H / 2, PX = OH * OH;  // small-map pixels per image: 16 or 4
    static constexpr int IPT = 16 / PX;             // images per column block
    static constexpr int IMG = 2 * IPT;             // images per workgroup
};

// filter staging: a round's slab is 1152 float4 (both 16-channel halves) or 576 (MT = 1: half `h_` of it); float4 number p = tid + 256 j of
// the workgroup's part sits at source index p (MT = 2) or (p >> 4) * 32 + 16 h + (p & 15).  Registers are NAMED: as an array they went to scratch.
#define S2_W_SETUP()                                                                                             \
    const int wh_ = MT == 2 ? 0 : (blockIdx.y & 1);                                                              \
    const float4 *wsrc = reinterpret_cast<const float4 *>(w9) +                                                  \
                         static_cast<size_t>(MT == 2 ? blockIdx.y : blockIdx.y >> 1) * rounds * (S2_WF / 4);     \
    const int wp2_ = MT == 2 ? threadIdx.x + 512 : (threadIdx.x < 64 ? threadIdx.x + 512 : threadIdx.x + 256);   \
    const int wp4_ = threadIdx.x < 128 ? threadIdx.x + 1024 : threadIdx.x + 768;                                 \
    const int ws0_ = MT == 2 ? threadIdx.x : (threadIdx.x >> 4) * 32 + 16 * wh_ + (threadIdx.x & 15);            \
    const int ws1_ = MT == 2 ? threadIdx.x + 256 : ws0_ + 512;                                                   \
    const int ws2_ = MT == 2 ? wp2_ : (wp2_ >> 4) * 32 + 16 * wh_ + (wp2_ & 15);                                 \
    float4 w0, w1, w2, w3, w4;                                                                                   \
    w3 = w4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f)

#define S2_LOAD_W(round_)                                                                     \
    do {                                                                                      \
        const float4 *wp_ = wsrc + static_cast<size_t>(round_) * (S2_WF / 4);                 \
        w0 = wp_[ws0_];                                                                       \
        w1 = wp_[ws1_];                                                                       \
        w2 = wp_[ws2_];                                                                       \
        if (MT == 2) {                                                                        \
            w3 = wp_[threadIdx.x + 768];                                                      \
            w4 = wp_[wp4_];                                                                   \
        }                                                                                     \
    } while (0)

#define S2_STORE_W(dst_)                                                                      \
    do {                                                                                      \
        float4 *wd_ = reinterpret_cast<float4 *>(dst_);                                       \
        wd_[threadIdx.x] = w0;                                                                \
        wd_[threadIdx.x + 256] = w1;                                                          \
        if (MT == 2) {                                                                        \
            wd_[threadIdx.x + 512] = w2;                                                      \
            wd_[threadIdx.x + 768] = w3;                                                      \
            if (threadIdx.x < 128) wd_[threadIdx.x + 1024] = w4;                              \
        } else if (threadIdx.x < 64) {                                                        \
            wd_[threadIdx.x + 512] = w2;                                                      \
        }                                                                                     \
    } while (0)

constexpr int S2_XS = 9 * 4 * 2 * 64;  // forward: a round's inputs [tap][quad][column block][16 n][4 k] = 4608 floats

// ---- forward: x [B][KC][H][H] -> y [B][RC][H/2][H/2].  grid (ceil(B / IMG), RC / (16 MT)) -------------------------------------------
template <int H, int MT>
__global__ __launch_bounds__(S2_NT) void conv3s2_fwd_mfma_kernel(const float *__restrict__ x, const float *__restrict__ w9, float *__restrict__ y, S2Dims d) {
    using G = S2Geo<H>;
    constexpr int OH = G::OH, PX = G::PX, IPT = G::IPT, IMG = G::IMG;
    constexpr int WFM = 9 * 4 * MT * 64, BUF = WFM + S2_XS, RB = 16 * MT;
    // two buffers of {filters of a round, inputs}: a round's products run while the next round is written; after the rounds the first
    // one holds the four wavefronts' partial sums [4][RB][36]
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int b0 = blockIdx.x * IMG, co0 = blockIdx.y * RB;
    for (int i = threadIdx.x; i < S2_XS; i += S2_NT) lds[WFM + i] = lds[BUF + WFM + i] = 0.0f;  // the taps that leave the map stay zero: every round rewrites the same other slots
    const int rounds = d.KC / S2_CK;
    S2_W_SETUP();
    // x of a round: IMG images x 16 channels x H^2 floats = 512 float4, two per thread.  H = 8: float4 tid of image j = (ci, row, half);
    // H = 4: float4 tid + 256 j = (image, ci, row)
    int img0, img1, ci_s, row_s, x0_s;
    if (H == 8) {
        img0 = 0, img1 = 1, ci_s = threadIdx.x >> 4, row_s = (threadIdx.x & 15) >> 1, x0_s = 4 * (threadIdx.x & 1);
    } else {
        img0 = threadIdx.x >> 6, img1 = img0 + 4, ci_s = (threadIdx.x & 63) >> 2, row_s = threadIdx.x & 3, x0_s = 0;
    }
    const int bi0 = b0 + img0 < d.B ? b0 + img0 : d.B - 1, bi1 = b0 + img1 < d.B ? b0 + img1 : d.B - 1;  // past the batch: a valid image, never stored
    const float *xsrc0 = x + (static_cast<size_t>(bi0) * d.KC + ci_s) * (H * H) + row_s * H + x0_s;
    const float *xsrc1 = x + (static_cast<size_t>(bi1) * d.KC + ci_s) * (H * H) + row_s * H + x0_s;
    float4 xa, xb;
    // scatter one float4 (row y, columns x0 .. x0+3 of channel ci_s, image slot img) to the tap planes it feeds
    auto put = [&](float *xs, float4 v, int img) {
        const int nt = img / IPT, il = img - nt * IPT;
        float *base = xs + ((ci_s >> 2) * 2 + nt) * 64 + (ci_s & 3) + 4 * (il * PX);
        const int o = x0_s >> 1, yh = row_s >> 1;
        // row taps: even row -> ky 1 at oy = y/2; odd row -> ky 2 at oy = (y-1)/2 and ky 0 at oy = (y+1)/2 (if inside)
        const bool odd = row_s & 1;
        const int kyA = odd ? 2 : 1;
        const bool okB = odd && yh + 1 < OH;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            if (rt == 1 && !okB) break;
            const int ky = rt == 0 ? kyA : 0, oy = rt == 0 ? yh : yh + 1;
            float *p = base + (ky * 3) * 512 + 4 * (oy * OH);
            p[1 * 512 + 4 * o] = v.x;        // column x0 (even): kx 1
            p[2 * 512 + 4 * o] = v.y;        // x0+1 (odd): kx 2 at ox = o, kx 0 at ox = o+1
            p[0 * 512 + 4 * (o + 1)] = v.y;
            p[1 * 512 + 4 * (o + 1)] = v.z;  // x0+2 (even): kx 1
            p[2 * 512 + 4 * (o + 1)] = v.w;  // x0+3 (odd): kx 2 at o+1, kx 0 at o+2 (if inside)
            if (o + 2 < OH) p[0 * 512 + 4 * (o + 2)] = v.w;
        }
    };
    f32x4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const size_t xstep = static_cast<size_t>(S2_CK) * (H * H);
    S2_LOAD_W(0);
    xa = *reinterpret_cast<const float4 *>(xsrc0);
    xb = *reinterpret_cast<const float4 *>(xsrc1);
    __syncthreads();  // the zero fill
    S2_STORE_W(lds);
    put(lds + WFM, xa, img0);
    put(lds + WFM, xb, img1);
    {
        const int nr = rounds > 1 ? 1 : 0;
        S2_LOAD_W(nr);
        xa = *reinterpret_cast<const float4 *>(xsrc0 + nr * xstep);
        xb = *reinterpret_cast<const float4 *>(xsrc1 + nr * xstep);
    }
    __syncthreads();
    const int aofs = wave * (MT * 64) + 4 * l15 + lq, bofs = WFM + wave * 128 + 4 * l15 + lq;
    for (int round = 0; round < rounds; ++round) {
        float *cur = lds + (round & 1) * BUF, *nxt = lds + ((round + 1) & 1) * BUF;
        if (round + 1 < rounds) {  // the other buffer was last read before the barrier that ended the previous round
            S2_STORE_W(nxt);
            put(nxt + WFM, xa, img0);
            put(nxt + WFM, xb, img1);
        }
        {
            const int nr = round + 2 < rounds ? round + 2 : rounds - 1;  // always issued: a load under a condition costs its own round trip
            S2_LOAD_W(nr);
            xa = *reinterpret_cast<const float4 *>(xsrc0 + nr * xstep);
            xb = *reinterpret_cast<const float4 *>(xsrc1 + nr * xstep);
        }
        const float *ap = cur + aofs, *bp = cur + bofs;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v0 = bp[t * 512], v1 = bp[t * 512 + 64];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float a = ap[t * (256 * MT) + m * 64];
                acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v0, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v1, acc[m][1], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- the four partial sums meet: red[wave][co RB][36], D row = 4 (lane >> 4) + reg, column = lane & 15 ---------------------------------
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[wave * (RB * S2_RS) + (m * 16 + 4 * lq + r) * S2_RS + n * 16 + l15] = acc[m][n][r];
    __syncthreads();
    if (threadIdx.x < 8 * RB) {
        const int co = threadIdx.x >> 3, nq = threadIdx.x & 7;  // four consecutive columns of one result channel
        const float *rp = lds + co * S2_RS + 4 * nq;
        float4 s = *reinterpret_cast<const float4 *>(rp);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 v = *reinterpret_cast<const float4 *>(rp + w * (RB * S2_RS));
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
        // H = 8: columns = pixels 4 (nq & 3) .. of image nq >> 2;  H = 4: the 2x2 plane of image nq
        const int img = H == 8 ? nq >> 2 : nq, off = H == 8 ? 4 * (nq & 3) : 0;
        if (b0 + img < d.B) *reinterpret_cast<float4 *>(y + (static_cast<size_t>(b0 + img) * d.RC + co0 + co) * PX + off) = s;
    }
}

// ---- backward-data: dy [B][KC][H/2][H/2] -> dx [B][RC][H][H].  grid (ceil(B / IMG), RC / (16 MT)) -----------------------------------
// dx[2i+py][2j+px] = sum over the taps of parity class (py, px): row taps py = 0: ky 1 (dy row i); py = 1: ky 0 (row i+1) and ky 2 (row i).
constexpr int S2_DS = 4 * 4 * 2 * 64;  // dy shifts of a round: [shift sy*2+sx][quad][column block][16 n][4 k] = 2048 floats
constexpr int S2_OS = 20;              // row stride of the interleave exchange [class][column block][RB ci][20]

template <int H, int MT>
__global__ __launch_bounds__(S2_NT) void conv3s2_bwd_mfma_kernel(const float *__restrict__ dy, const float *__restrict__ w9, float *__restrict__ dx, S2Dims d) {
    using G = S2Geo<H>;
    constexpr int OH = G::OH, PX = G::PX, IPT = G::IPT, IMG = G::IMG;
    constexpr int WFM = 9 * 4 * MT * 64, BUF = WFM + S2_DS, RB = 16 * MT, OW = 4 * 2 * RB * S2_OS;  // OW: one accumulator set in the exchange
    __shared__ __align__(16) float lds[2 * BUF];  // rounds: two buffers of {filters, dy shifts}; afterwards: two accumulator sets
    static_assert(BUF >= OW, "the exchange fits the staging buffers");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int b0 = blockIdx.x * IMG, ci0 = blockIdx.y * RB;
    for (int i = threadIdx.x; i < S2_DS; i += S2_NT) lds[WFM + i] = lds[BUF + WFM + i] = 0.0f;  // shifted-out slots stay zero
    const int rounds = d.KC / S2_CK;
    S2_W_SETUP();
    // dy of a round: IMG images x 16 channels x PX floats = 128 float4: threads 0..127 (the others repeat a valid address and skip the write).
    // H = 8 (4x4 dy): float4 = (image tid >> 6, co (tid & 63) >> 2, row tid & 3);  H = 4 (2x2 dy): float4 = (image tid >> 4, co tid & 15) plane
    const int dt = threadIdx.x & 127;
    const int img_s = H == 8 ? dt >> 6 : dt >> 4, co_s = H == 8 ? (dt & 63) >> 2 : dt & 15, row_s = H == 8 ? dt & 3 : 0;
    const int bi = b0 + img_s < d.B ? b0 + img_s : d.B - 1;
    const float *dsrc = dy + (static_cast<size_t>(bi) * d.KC + co_s) * PX + 4 * row_s;
    float4 da;
    auto put = [&](float *ds, float4 v) {
        const int nt = img_s / IPT, il = img_s - nt * IPT;
        float *base = ds + ((co_s >> 2) * 2 + nt) * 64 + (co_s & 3) + 4 * (il * PX);  // shift stride 512 floats
        if (H == 8) {  // row oy = row_s, columns 0..3: shift (sy, sx) holds dy[i + sy][j + sx] at (i, j)
            const int oy = row_s;
            float *p0 = base + 4 * (oy * OH);
            p0[0] = v.x, p0[4] = v.y, p0[8] = v.z, p0[12] = v.w;                    // (0, 0)
            p0[512 + 0] = v.y, p0[512 + 4] = v.z, p0[512 + 8] = v.w;                // (0, 1): j = ox - 1
            if (oy > 0) {
                float *p1 = base + 4 * ((oy - 1) * OH);
                p1[1024 + 0] = v.x, p1[1024 + 4] = v.y, p1[1024 + 8] = v.z, p1[1024 + 12] = v.w;  // (1, 0)
                p1[1536 + 0] = v.y, p1[1536 + 4] = v.z, p1[1536 + 8] = v.w;                          // (1, 1)
            }
        } else {  // the 2x2 plane: v = dy[0][0], [0][1], [1][0], [1][1]
            base[0] = v.x, base[4] = v.y, base[8] = v.z, base[12] = v.w;  // (0, 0)
            base[512 + 0] = v.y, base[512 + 8] = v.w;                     // (0, 1)
            base[1024 + 0] = v.z, base[1024 + 4] = v.w;                   // (1, 0)
            base[1536 + 0] = v.w;                                         // (1, 1)
        }
    };
    f32x4 acc[4][2][MT];  // [parity class py*2+px][column block][channel half]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[c][n][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const size_t dstep = static_cast<size_t>(S2_CK) * PX;
    S2_LOAD_W(0);
    da = *reinterpret_cast<const float4 *>(dsrc);
    __syncthreads();  // the zero fill
    S2_STORE_W(lds);
    if (threadIdx.x < 128) put(lds + WFM, da);
    {
        const int nr = rounds > 1 ? 1 : 0;
        S2_LOAD_W(nr);
        da = *reinterpret_cast<const float4 *>(dsrc + nr * dstep);
    }
    __syncthreads();
    const int aofs = wave * (MT * 64) + 4 * l15 + lq, bofs = WFM + wave * 128 + 4 * l15 + lq;
    for (int round = 0; round < rounds; ++round) {
        float *cur = lds + (round & 1) * BUF, *nxt = lds + ((round + 1) & 1) * BUF;
        if (round + 1 < rounds) {
            S2_STORE_W(nxt);
            if (threadIdx.x < 128) put(nxt + WFM, da);
        }
        {
            const int nr = round + 2 < rounds ? round + 2 : rounds - 1;
            S2_LOAD_W(nr);
            da = *reinterpret_cast<const float4 *>(dsrc + nr * dstep);
        }
        const float *ap = cur + aofs, *bp = cur + bofs;
        float bv[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int n = 0; n < 2; ++n) bv[s][n] = bp[s * 512 + n * 64];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - 3 * ky;
            const int cls = (ky != 1 ? 2 : 0) + (kx != 1 ? 1 : 0), sh = (ky == 0 ? 2 : 0) + (kx == 0 ? 1 : 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float a = ap[t * (256 * MT) + m * 64];
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[cls][n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[sh][n], acc[cls][n][m], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- pairwise sum of the four wavefronts' accumulators through LDS, then the parity interleave ------------------------------------
    // exchange layout [class][column block][ci RB][20]: lane (lq, l15) of tile (m, r) -> ci = 16 m + 4 lq + r, column l15
    auto spill = [&](float *dst) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[((c * 2 + n) * RB + m * 16 + 4 * lq + r) * S2_OS + l15] = acc[c][n][m][r];
    };
    if (wave >= 2) spill(lds + (wave - 2) * OW);
    __syncthreads();
    if (wave < 2) {
        const float *src = lds + wave * OW;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[c][n][m][r] += src[((c * 2 + n) * RB + m * 16 + 4 * lq + r) * S2_OS + l15];
    }
    __syncthreads();
    if (wave < 2) spill(lds + wave * OW);
    __syncthreads();
    // IMG x RB channels x H^2 floats = 512 MT float4 (4 consecutive x of one row): 2 MT per thread
#pragma unroll
    for (int k = 0; k < 2 * MT; ++k) {
        const int f = threadIdx.x + S2_NT * k;
        constexpr int F4R = H / 4, F4I = RB * H * F4R;  // float4 per row / per image
        const int img = f / F4I, g = f - img * F4I, ci = g / (H * F4R), q = g - ci * (H * F4R), yy = q / F4R, xq = q - yy * F4R;
        const int nt = img / IPT, il = img - nt * IPT, i = yy >> 1, py = yy & 1;
        float out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int xx = 4 * xq + e, j = xx >> 1, c = py * 2 + (xx & 1), n = il * PX + i * OH + j;
            const int o = ((c * 2 + nt) * RB + ci) * S2_OS + n;
            out[e] = lds[o] + lds[OW + o];
        }
        if (b0 + img < d.B)
            *reinterpret_cast<float4 *>(dx + (static_cast<size_t>(b0 + img) * d.RC + ci0 + ci) * (H * H) + yy * H + 4 * xq) =
                make_float4(out[0], out[1], out[2], out[3]);
    }
}

// 16 MT result channels per workgroup.  B = 100, un-profiled us, MT = 2 / 1: forward 8x8 18.6 / 23.9, 4x4 21.6 / 22.0; backward-data 8x8
// 19.6 / 18.0, 4x4 35.2 / 27.4 (104 workgroups at MT = 2).  EEADV_S2_MT = four digits (forward 8x8, forward 4x4, backward 8x8, backward
// 4x4) overrides, for A/B runs.
int s2_mt(bool bwd, int H) {
    const char *env = std::getenv("EEADV_S2_MT");  // read per call (tests flip it)
    const int slot = (bwd ? 2 : 0) + (H == 4 ? 1 : 0);
    if (env && std::strlen(env) == 4 && (env[slot] == '1' || env[slot] == '2')) return env[slot] - '0';
    return bwd ? 1 : 2;
}

template <int H, int MT>
int s2_launch(bool bwd, const float *in, const float *w9, float *out, const S2Dims &d, hipStream_t st) {
    const dim3 grid(static_cast<unsigned>((d.B + S2Geo<H>::IMG - 1) / S2Geo<H>::IMG), static_cast<unsigned>(d.RC / (16 * MT)));
    if (bwd) {
        EE_LAUNCH((conv3s2_bwd_mfma_kernel<H, MT>), grid, dim3(S2_NT), 0, st, in, w9, out, d);
    } else {
        constexpr size_t bytes = 2 * (9 * 4 * MT * 64 + S2_XS) * sizeof(float);  // 72 / 54 KB: above the static limit
        static int ok = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3s2_fwd_mfma_kernel<H, MT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            static_cast<int>(bytes)) == hipSuccess;
        if (!ok) return EE_ERR_UNSUPPORTED;
        EE_LAUNCH((conv3s2_fwd_mfma_kernel<H, MT>), grid, dim3(S2_NT), bytes, st, in, w9, out, d);
    }
    return launch_status();
}

int s2_dispatch(bool bwd, const float *in, const float *w9, float *out, const S2Dims &d, int H, hipStream_t st) {
    const int mt = s2_mt(bwd, H);
    if (H == 8) return mt == 2 ? s2_launch<8, 2>(bwd, in, w9, out, d, st) : s2_launch<8, 1>(bwd, in, w9, out, d, st);
    return mt == 2 ? s2_launch<4, 2>(bwd, in, w9, out, d, st) : s2_launch<4, 1>(bwd, in, w9, out, d, st);
}

int s2_check(const void *a, const void *b, const void *c, int B, int KC, int RC, int H) {
    if (B < 0 || KC < 1 || RC < 1) return EE_ERR_SHAPE;
    if (KC % S2_CK != 0 || RC % S2_RB != 0 || (H != 4 && H != 8)) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!a || !b || !c) return EE_ERR_NULL;
    if (!aligned16(a) || !aligned16(b) || !aligned16(c)) return EE_ERR_ALIGN;
    if (static_cast<int64_t>(B) * (KC > RC ? KC : RC) * H * H > 0x7fffffffLL) return EE_ERR_SHAPE;
    return EE_OK;
}

}  // namespace

// x [B][Cin][H][H] (H = 8 or 4), w9 = the filters as [Cout/32][Cin/16][9][4][2][16][4] (w9[..][t][q][h][m][k] = weight[32 cb + 16 h + m]
// [16 round + 4 q + k][t]) -> y [B][Cout][H/2][H/2]
EE_API int ee_conv3x3s2_small_fwd_f32(const float *x, const float *w9, float *y, int B, int Cin, int Cout, int H, void *stream) {
    const int rc = s2_check(x, w9, y, B, Cin, Cout, H);
    if (rc != EE_OK || B == 0) return rc;
    const S2Dims d{B, Cin, Cout};
    ProfScope prof(EE_K_CONV3S2_FWD, as_stream(stream), 2.0 * 9.0 * Cin * Cout * static_cast<double>(B) * (H / 2) * (H / 2));
    return s2_dispatch(false, x, w9, y, d, H, as_stream(stream));
}

// dy [B][Cout][H/2][H/2], w9 = the filters as [Cin/32][Cout/16][9][4][2][16][4] (w9[..][t][q][h][m][k] = weight[16 round + 4 q + k]
// [32 cb + 16 h + m][t]) -> dx [B][Cin][H][H] (H = 8 or 4; all of it is written)
EE_API int ee_conv3x3s2_small_bwd_data_f32(const float *dy, const float *w9, float *dx, int B, int Cin, int Cout, int H, void *stream) {
    const int rc = s2_check(dy, w9, dx, B, Cout, Cin, H);
    if (rc != EE_OK || B == 0) return rc;
    const S2Dims d{B, Cout, Cin};
    ProfScope prof(EE_K_CONV3S2_BWD, as_stream(stream), 2.0 * 9.0 * Cin * Cout * static_cast<double>(B) * (H / 2) * (H / 2));
    return s2_dispatch(true, dy, w9, dx, d, H, as_stream(stream));
}
