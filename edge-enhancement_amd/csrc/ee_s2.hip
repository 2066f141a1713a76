// ee_s2.hip - Conv2d(3x3, stride 2, padding 1, bias=False) on small maps (the first convolution of ResNet-18's layer2 / layer3 / layer4 at
// 64x64 inputs: 16x16 -> 8x8, 8x8 -> 4x4 and 4x4 -> 2x2, resnet.py:26-31, :132-137), forward and backward-data, on the f32 matrix cores.
//
// Why a kernel of its own: these are GEMMs with few pixels (6400 / 1600 / 400 columns) and long reductions (576 / 1152 / 2304), which MIOpen
// serves with an NHWC implicit GEMM between two layout transposes and a zero fill (5 launches, 28-39 us un-profiled), and which the direct
// kernels of ee_conv.hip tile badly.  Here a workgroup owns 32 (or 16) result channels x 32 pixels (two
// v_mfma_f32_16x16x4 column blocks: half an image of an 8x8 result, two images of a 4x4 one, eight of a 2x2 one) and its four wavefronts
// SPLIT THE REDUCTION: per round of 16 reduction channels, wavefront w multiplies channel quad w through all nine taps (2 x 2
// accumulator tiles, 36 MFMAs, operands read once per two products), and the four partial sums meet in LDS at the end.  The workgroup has
// 512 lanes: wavefronts 0-3 only multiply, wavefronts 4-7 only PRODUCE (keep two rounds of loads in flight in two named register sets and
// write the next round into the other LDS buffer while the current one is multiplied) - a wavefront issues in order, so with four
// do-everything wavefronts (one per SIMD where the grid gives one workgroup per CU) load latency, staging and products just added up
// (same finding and same cure as ee_wino.hip).  One barrier per round.
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
#include "ee_fuse.hpp"

namespace {

using namespace ee;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int S2_NT = 512, S2_CK = 16, S2_RB = 32;  // 256 consumer + 256 producer lanes
constexpr int S2_RS = 36;              // row stride of the partial-sum exchange: the four k of a wavefront on disjoint banks
constexpr int S2_MAX_KC = 512;         // backward-data with PRE: reduction channels whose gamma * invstd table fits (2 x 2 KB of LDS)

struct S2Dims {
    int B, KC, RC;  // reduction channels (Cin forward, Cout backward), result channels
    int wl;         // XCD numbering with the channel block fastest (ee_common.hpp: xcd_decode)
};

// H = the LARGE map's side (x forward, dx backward); the small map is OH x OH.  A workgroup's 32 small-map pixels: H = 4: eight images
// (four per column block); H = 8: two images; H = 16: four rows of one image (grid.x = 2 B: blockIdx.x & 1 = which half).
template <int H>
struct S2Geo {
    static constexpr int OH = H / 2, PX = OH * OH;
    static constexpr int IMG = H == 16 ? 1 : 32 / PX;  // images per workgroup
};

// float offset (inside a [column block][16 n][4 k] pair of blocks) of small-map pixel (row r, column 0) - r = the row inside the image
// (H = 4, 8) or inside the workgroup's four rows (H = 16); columns advance by 4 floats
template <int H>
__device__ __forceinline__ int s2_slot(int img, int r) {
    if (H == 16) return (r >> 1) * 64 + 32 * (r & 1);
    if (H == 8) return img * 64 + 16 * r;
    return (img >> 2) * 64 + 4 * ((img & 3) * 4 + 2 * r);
}

// filter staging: a round's slab holds TAPS x 128 float4 (both 16-channel halves; TAPS = 9, or 10 with the shortcut's 1x1 filters as the last
// tap); a workgroup stages all of it (MT = 2) or half `wh_` of it (MT = 1): WF4 float4, number p = tid + 256 j sits at source index p or
// (p >> 4) * 32 + 16 wh_ + (p & 15).  The last, partial batch is loaded by everybody (a valid duplicate address) and written by the first
// LV threads.  Registers are NAMED: as an array they went to scratch.
#define S2_W_SETUP()                                                                                             \
    constexpr int TAPS_F4 = TAPS * 128, WF4 = TAPS * 64 * MT, NJ = (WF4 + 255) / 256, LV = WF4 - 256 * (NJ - 1); \
    static_assert(NJ == (MT == 2 ? 5 : 3), "five / three float4 per thread");                                    \
    const int wh_ = MT == 2 ? 0 : (by & 1);                                                                      \
    const float4 *wsrc = reinterpret_cast<const float4 *>(w9) +                                                  \
                         static_cast<size_t>(MT == 2 ? by : by >> 1) * rounds * TAPS_F4;                         \
    const int wpl_ = pt < LV ? pt + 256 * (NJ - 1) : pt + 256 * (NJ - 2);                                        \
    const int ws0_ = MT == 2 ? pt : (pt >> 4) * 32 + 16 * wh_ + (pt & 15);                                       \
    const int ws1_ = MT == 2 ? pt + 256 : ws0_ + 512;                                                            \
    const int wsl_ = MT == 2 ? wpl_ : (wpl_ >> 4) * 32 + 16 * wh_ + (wpl_ & 15);                                 \
    float4 Aw0, Aw1, Aw2, Aw3, Aw4, Bw0, Bw1, Bw2, Bw3, Bw4;                                                     \
    Aw3 = Aw4 = Bw3 = Bw4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f)

#define S2_LOAD_W(S, round_)                                                                  \
    do {                                                                                      \
        const int r_ = (round_) < rounds ? (round_) : rounds - 1; /* always issued: a load under a condition costs its own round trip */ \
        const float4 *wp_ = wsrc + static_cast<size_t>(r_) * TAPS_F4;                         \
        S##w0 = wp_[ws0_];                                                                    \
        S##w1 = wp_[ws1_];                                                                    \
        if (MT == 2) {                                                                        \
            S##w2 = wp_[pt + 512];                                                            \
            S##w3 = wp_[pt + 768];                                                            \
            S##w4 = wp_[wsl_];                                                                \
        } else {                                                                              \
            S##w2 = wp_[wsl_];                                                                \
        }                                                                                     \
    } while (0)

#define S2_STORE_W(dst_, S)                                                                   \
    do {                                                                                      \
        float4 *wd_ = reinterpret_cast<float4 *>(dst_);                                       \
        wd_[pt] = S##w0;                                                                      \
        wd_[pt + 256] = S##w1;                                                                \
        if (MT == 2) {                                                                        \
            wd_[pt + 512] = S##w2;                                                            \
            wd_[pt + 768] = S##w3;                                                            \
            if (pt < LV) wd_[pt + 1024] = S##w4;                                              \
        } else if (pt < LV) {                                                                 \
            wd_[pt + 512] = S##w2;                                                            \
        }                                                                                     \
    } while (0)

constexpr int S2_XS = 9 * 4 * 2 * 64;  // forward: a round's inputs [tap][quad][column block][16 n][4 k] = 4608 floats

// ---- forward: x [B][KC][H][H] -> y [B][RC][H/2][H/2].  1-D grid of (ceil(B / IMG) or 2 B) x RC / (16 MT) workgroups ------------------------------------
// DS: the block's shortcut Conv2d(1x1, stride 2) of the SAME input (resnet.py:137-142) rides along as a tenth tap - its B operand is the centre
// tap's plane - into accumulators of its own -> y1
// POST (eval-mode BatchNorm folded in, ee_fuse.hpp): y = relu(bn(conv3x3s2(x))) by `post`, y1 = bn_ds(conv1x1s2(x)) by `post1`
// STATS (train-mode BatchNorm across the kernel boundary, ee_fuse.hpp: TrainBn): next to the raw y, per result channel the (mean, M2) of the
// workgroup's pixels of each image -> stats_out [RC][S][2]: H = 16: S = 2 B half images of 32 values; H = 8: S = B images of 16 values
template <int H, int MT, bool DS, bool POST, bool STATS = false>
__global__ __launch_bounds__(S2_NT, 4) void conv3s2_fwd_mfma_kernel(const float *__restrict__ x, const float *__restrict__ w9, float *__restrict__ y,
                                                                 float *__restrict__ y1, S2Dims d, FusePost post, FusePost post1, float *__restrict__ stats_out = nullptr) {
    using G = S2Geo<H>;
    constexpr int OH = G::OH, PX = G::PX, IMG = G::IMG, TAPS = DS ? 10 : 9;
    constexpr int WFM = TAPS * 4 * MT * 64, BUF = WFM + S2_XS, RB = 16 * MT;
    // two buffers of {filters of a round, inputs}; after the rounds the first one holds the four wavefronts' partial sums [4][RB][36]
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const bool producer = wave >= 4;
    const int pt = threadIdx.x & 255;  // lane number inside its group of four wavefronts
    // 1-D grid numbered for the XCDs' L2s (ee_common.hpp: xcd_decode): channel block fastest where the filters are the bigger footprint (an
    // XCD then fetches only the slabs of the blocks congruent to it: 4.7 MB on the 4x4 layer, otherwise eight times), image group otherwise
    int bx, by;
    if (!xcd_decode(blockIdx.x, d.RC / RB, H == 16 ? 2 * d.B : (d.B + IMG - 1) / IMG, d.wl, bx, by)) return;
    const int b0 = H == 16 ? bx >> 1 : bx * IMG, hh = H == 16 ? bx & 1 : 0, co0 = by * RB;
    for (int i = threadIdx.x; i < S2_XS; i += S2_NT) lds[WFM + i] = lds[BUF + WFM + i] = 0.0f;  // the taps that leave the map stay zero: every round rewrites the same other slots
    const int rounds = d.KC / S2_CK;
    S2_W_SETUP();
    // x of a round, float4 number j of a thread (two; three for H = 16, the third by one wavefront):
    //   H = 4: 8 images x 16 channels x 4 rows:     (image (tid >> 6) + 4 j, ci (tid & 63) >> 2, row tid & 3)
    //   H = 8: 2 images x 16 channels x 8 rows x 2: (image j, ci tid >> 4, row (tid & 15) >> 1, half tid & 1)
    //   H = 16: 9 rows (8 hh - 1 ...) x 16 channels x 4: (row (tid >> 6) + 4 j, ci (tid & 63) >> 2, quarter tid & 3)
    const int ci_s = H == 8 ? pt >> 4 : (pt & 63) >> 2;
    const int x0_s = H == 8 ? 4 * (pt & 1) : (H == 16 ? 4 * (pt & 3) : 0), o_s = x0_s >> 1;
    // per float4: the two row taps it feeds.  Even input row -> ky 1 at oy = y/2; odd row -> ky 2 at oy = (y-1)/2 and ky 0 at oy = (y+1)/2
    int offA[3], offB[3];  // LDS float offsets of (tap row, small-map row); < 0: none.  Compile-time indexed after unrolling.
    const float *src[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int img = 0, yy, oyA, kyA, oyB;
        bool okA = true, okB, live = true;
        if (H == 16) {
            const int r = (pt >> 6) + 4 * j;  // input row 8 hh - 1 + r, r = 0..8
            live = j < 2 || pt < 64;
            yy = 8 * hh - 1 + r;
            const bool odd_r = r & 1;  // odd r = even input row
            kyA = odd_r ? 1 : 2, oyA = odd_r ? (r - 1) >> 1 : (r >> 1) - 1, okA = live && (odd_r || r >= 2);
            oyB = r >> 1, okB = live && !odd_r && r <= 6 && yy >= 0;
            if (yy < 0) yy = 0;  // the padding row: a valid address, never scattered
            if (!live) yy = 8 * hh;
        } else {
            img = H == 8 ? j : (pt >> 6) + 4 * j;
            live = j < 2;
            yy = H == 8 ? (pt & 15) >> 1 : pt & 3;
            const bool odd = yy & 1;
            kyA = odd ? 2 : 1, oyA = yy >> 1, okA = live;
            oyB = (yy >> 1) + 1, okB = live && odd && oyB < OH;
        }
        offA[j] = okA ? kyA * 3 * 512 + s2_slot<H>(img, oyA) : -1;
        offB[j] = okB ? s2_slot<H>(img, oyB) : -1;
        const int bi = b0 + img < d.B ? b0 + img : d.B - 1;  // past the batch: a valid image, never stored
        src[j] = x + (static_cast<size_t>(bi) * d.KC + ci_s) * (H * H) + yy * H + x0_s;
    }
    const float *src0 = src[0], *src1 = src[1], *src2 = src[H == 16 ? 2 : 1];
    const int oA0 = offA[0], oA1 = offA[1], oA2 = offA[2], oB0 = offB[0], oB1 = offB[1], oB2 = offB[2];
    float4 Axa, Axb, Axc, Bxa, Bxb, Bxc;  // two NAMED register sets (A holds even rounds, B odd ones)
    Axc = Bxc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // scatter one float4 (columns x0 .. x0+3 of channel ci_s) to the tap planes of one (tap row, small-map row)
    auto put_cols = [&](float *p, float4 v) {
        p[1 * 512 + 4 * o_s] = v.x;        // column x0 (even): kx 1
        p[2 * 512 + 4 * o_s] = v.y;        // x0+1 (odd): kx 2 at ox = o, kx 0 at ox = o+1
        p[0 * 512 + 4 * (o_s + 1)] = v.y;
        p[1 * 512 + 4 * (o_s + 1)] = v.z;  // x0+2 (even): kx 1
        p[2 * 512 + 4 * (o_s + 1)] = v.w;  // x0+3 (odd): kx 2 at o+1, kx 0 at o+2 (if inside)
        if (o_s + 2 < OH) p[0 * 512 + 4 * (o_s + 2)] = v.w;
    };
    auto put = [&](float *xs, float4 v, int oA, int oB) {
        float *base = xs + (ci_s >> 2) * 128 + (ci_s & 3);
        if (oA >= 0) put_cols(base + oA, v);
        if (oB >= 0) put_cols(base + oB, v);
    };
#define S2_LOAD_X(S, round_)                                                                   \
    do {                                                                                       \
        const size_t xo_ = static_cast<size_t>((round_) < rounds ? (round_) : rounds - 1) * xstep; \
        S##xa = *reinterpret_cast<const float4 *>(src0 + xo_);                                 \
        S##xb = *reinterpret_cast<const float4 *>(src1 + xo_);                                 \
        if (H == 16) S##xc = *reinterpret_cast<const float4 *>(src2 + xo_);                    \
    } while (0)
#define S2_PUT_X(dst_, S)                                                                      \
    do {                                                                                       \
        put(dst_, S##xa, oA0, oB0);                                                            \
        put(dst_, S##xb, oA1, oB1);                                                            \
        if (H == 16) put(dst_, S##xc, oA2, oB2);                                               \
    } while (0)
    f32x4 acc[MT][2], acc1[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = acc1[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const size_t xstep = static_cast<size_t>(S2_CK) * (H * H);
    auto multiply = [&](const float *cur) {
        const float *ap = cur + wave * (MT * 64) + 4 * l15 + lq, *bp = cur + WFM + wave * 128 + 4 * l15 + lq;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v0 = bp[t * 512], v1 = bp[t * 512 + 64];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float a = ap[t * (256 * MT) + m * 64];
                acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v0, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v1, acc[m][1], 0, 0, 0);
                if (DS && t == 4) {  // x[2 oy][2 ox]: the centre tap's plane under the 1x1 filters
                    const float a1 = ap[9 * (256 * MT) + m * 64];
                    acc1[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, v0, acc1[m][0], 0, 0, 0);
                    acc1[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, v1, acc1[m][1], 0, 0, 0);
                }
            }
        }
    };
    float *buf0 = lds, *buf1 = lds + BUF;
    if (producer) {
        S2_LOAD_W(A, 0);
        S2_LOAD_X(A, 0);
    }
    __syncthreads();  // the zero fill
    if (producer) {
        S2_STORE_W(buf0, A);
        S2_PUT_X(buf0 + WFM, A);
        S2_LOAD_W(B, 1);
        S2_LOAD_X(B, 1);
        S2_LOAD_W(A, 2);
        S2_LOAD_X(A, 2);
    }
    __syncthreads();
    for (int round = 0; round < rounds; round += 2) {
        // iteration `round` (even): multiply buffer 0 while round + 1 (set B) goes to buffer 1 (past the end: the last round again, never read)
        if (producer) {
            S2_STORE_W(buf1, B);
            S2_PUT_X(buf1 + WFM, B);
            S2_LOAD_W(B, round + 3);
            S2_LOAD_X(B, round + 3);
        } else {
            multiply(buf0);
        }
        __syncthreads();
        if (round + 1 < rounds) {  // iteration round + 1: multiply buffer 1 while round + 2 (set A) goes to buffer 0
            if (producer) {
                S2_STORE_W(buf0, A);
                S2_PUT_X(buf0 + WFM, A);
                S2_LOAD_W(A, round + 4);
                S2_LOAD_X(A, round + 4);
            } else {
                multiply(buf1);
            }
            __syncthreads();
        }
    }
#undef S2_LOAD_X
#undef S2_PUT_X
    // ---- the four partial sums meet: red[set][wave][co RB][36], D row = 4 (lane >> 4) + reg, column = lane & 15 -----------------------------
    constexpr int RED = 4 * RB * S2_RS;
    static_assert((DS ? 2 : 1) * RED <= 2 * BUF, "the exchange fits the staging buffers");
    if (!producer) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = wave * (RB * S2_RS) + (m * 16 + 4 * lq + r) * S2_RS + n * 16 + l15;
                    lds[o] = acc[m][n][r];
                    if (DS) lds[RED + o] = acc1[m][n][r];
                }
    }
    __syncthreads();
    if (threadIdx.x < 8 * RB) {
        const int co = threadIdx.x >> 3, nq = threadIdx.x & 7;  // four consecutive columns of one result channel
        // H = 16: columns 4 nq .. = half a row of the workgroup's four; H = 8: pixels 4 (nq & 3) .. of image nq >> 2; H = 4: the 2x2 plane of image nq
        const int img = H == 16 ? 0 : (H == 8 ? nq >> 2 : nq), off = H == 16 ? (4 * hh + (nq >> 1)) * 8 + 4 * (nq & 1) : (H == 8 ? 4 * (nq & 3) : 0);
        const size_t dst = (static_cast<size_t>(b0 + img) * d.RC + co0 + co) * PX + off;
#pragma unroll
        for (int set = 0; set < (DS ? 2 : 1); ++set) {
            const float *rp = lds + set * RED + co * S2_RS + 4 * nq;
            float4 s = *reinterpret_cast<const float4 *>(rp);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 v = *reinterpret_cast<const float4 *>(rp + w * (RB * S2_RS));
                s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
            }
            if constexpr (POST) {
                const FusePost &pp = set == 0 ? post : post1;
                if (pp.mean) {
                    const PostConst k = post_const(pp, co0 + co);
                    s.x = post_apply(s.x, k), s.y = post_apply(s.y, k), s.z = post_apply(s.z, k), s.w = post_apply(s.w, k);
                }
                if (pp.relu) s.x = relu_keep_nan(s.x), s.y = relu_keep_nan(s.y), s.z = relu_keep_nan(s.z), s.w = relu_keep_nan(s.w);
            }
            if (b0 + img < d.B) *reinterpret_cast<float4 *>((set == 0 ? y : y1) + dst) = s;
            if constexpr (STATS) {
                static_assert(!STATS || H == 16 || H == 8, "statistics epilogue: 8x8 or 4x4 results");
                if (set == 0) {
                    constexpr int LANES = H == 16 ? 8 : 4;  // the lanes holding one image's pixels of this workgroup
                    const float vals[4] = {s.x, s.y, s.z, s.w};
                    const float2 m = group_moments<LANES, 4>(vals);
                    const int part = H == 16 ? 2 * b0 + hh : b0 + img, S = H == 16 ? 2 * d.B : d.B;
                    if ((nq & (LANES - 1)) == 0 && b0 + img < d.B) *reinterpret_cast<float2 *>(stats_out + (static_cast<size_t>(co0 + co) * S + part) * 2) = m;
                }
            }
        }
    }
}

// ---- backward-data: dy [B][KC][H/2][H/2] -> dx [B][RC][H][H].  1-D grid of (ceil(B / IMG) or 2 B) x RC / (16 MT) workgroups ----------------------------
// dx[2i+py][2j+px] = sum over the taps of parity class (py, px): row taps py = 0: ky 1 (dy row i); py = 1: ky 0 (row i+1) and ky 2 (row i).
constexpr int S2_DS = 4 * 4 * 2 * 64;  // dy shifts of a round: [shift sy*2+sx][quad][column block][16 n][4 k] = 2048 floats
constexpr int S2_OS = 20;              // row stride of the interleave exchange [class][column block][RB ci][20]

// DS: plus the backward-data of the block's shortcut Conv2d(1x1, stride 2): dy1 (same shape as dy) under the transposed 1x1 filters (the
// tenth tap) lands on the even-even parity class only
// PRE (eval-mode BatchNorm backward folded in, ee_fuse.hpp): the staged dy is (pre.mask > 0 ? dy : 0) * gamma / sqrt(var + eps) of `pre`
// (the block's first BatchNorm + ReLU), the staged dy1 is dy1 * gamma / sqrt(var + eps) of `pre1` (the shortcut's BatchNorm)
template <int H, int MT, bool DS, bool PRE>
__global__ __launch_bounds__(S2_NT) void conv3s2_bwd_mfma_kernel(const float *__restrict__ dy, const float *__restrict__ dy1, const float *__restrict__ w9,
                                                                 float *__restrict__ dx, S2Dims d, FusePre pre, FusePre pre1) {
    using G = S2Geo<H>;
    constexpr int OH = G::OH, PX = G::PX, IMG = G::IMG, TAPS = DS ? 10 : 9;
    constexpr int WFM = TAPS * 4 * MT * 64, D1 = WFM + S2_DS, BUF = D1 + (DS ? 512 : 0), RB = 16 * MT, OW = 4 * 2 * RB * S2_OS;  // OW: one accumulator set in the exchange
    __shared__ __align__(16) float lds[2 * BUF];  // rounds: two buffers of {filters, dy shifts}; afterwards: two accumulator sets
    __shared__ float wtab[PRE ? 2 * S2_MAX_KC : 1];  // PRE: gamma * invstd per reduction channel, of `pre` and of `pre1`
    static_assert(BUF >= OW, "the exchange fits the staging buffers");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const bool producer = wave >= 4;
    const int pt = threadIdx.x & 255;  // lane number inside its group of four wavefronts
    int bx, by;  // XCD-aware numbering (see the forward kernel)
    if (!xcd_decode(blockIdx.x, d.RC / RB, H == 16 ? 2 * d.B : (d.B + IMG - 1) / IMG, d.wl, bx, by)) return;
    const int b0 = H == 16 ? bx >> 1 : bx * IMG, hh = H == 16 ? bx & 1 : 0, ci0 = by * RB;
    for (int i = threadIdx.x; i < S2_DS; i += S2_NT) lds[WFM + i] = lds[BUF + WFM + i] = 0.0f;  // shifted-out slots stay zero
    if constexpr (PRE)
        for (int c = threadIdx.x; c < d.KC; c += S2_NT) {
            wtab[c] = bn_scale(pre.var, pre.gamma, pre.eps, c);
            if (DS) wtab[S2_MAX_KC + c] = bn_scale(pre1.var, pre1.gamma, pre1.eps, c);
        }
    const int rounds = d.KC / S2_CK;
    S2_W_SETUP();
    // dy of a round, one float4 per thread (the idle threads repeat a valid address and skip the write):
    //   H = 4 (2x2 dy): 8 images x 16 channels: 128 planes (image tid >> 4, co tid & 15)
    //   H = 8 (4x4 dy): 2 images x 16 channels x 4 rows: 128 (image tid >> 6, co (tid & 63) >> 2, row tid & 3)
    //   H = 16 (8x8 dy): rows 4 hh .. 4 hh + 4 (the last one for the shift; row 8 does not exist) x 16 channels x 2 halves: 160
    //           (row tid >> 5, co (tid & 31) >> 1, half tid & 1)
    const int dt = H == 16 ? (pt < 160 ? pt : pt - 128) : pt & 127;
    const int img_s = H == 16 ? 0 : (H == 8 ? dt >> 6 : dt >> 4), co_s = H == 16 ? (dt & 31) >> 1 : (H == 8 ? (dt & 63) >> 2 : dt & 15);
    const int row_s = H == 16 ? dt >> 5 : (H == 8 ? dt & 3 : 0), half_s = H == 16 ? dt & 1 : 0;
    const int oy_s = H == 16 ? 4 * hh + row_s : row_s;
    const bool live = (H == 16 ? pt < 160 : pt < 128) && oy_s < OH;
    const int bi = b0 + img_s < d.B ? b0 + img_s : d.B - 1;
    const size_t doff = (static_cast<size_t>(bi) * d.KC + co_s) * PX + (H == 4 ? 0 : (oy_s < OH ? oy_s : OH - 1) * OH + 4 * half_s);
    const float *dsrc = dy + doff, *dsrc1 = DS ? dy1 + doff : dy + doff, *dsrcm = PRE ? pre.mask + doff : dy + doff;
    const bool live1 = DS && live && row_s < 4;  // the shortcut's gradient needs no shifted row
    float4 Ada, Adb, Bda, Bdb, Adm, Bdm;  // two NAMED register sets (A holds even rounds, B odd ones); dm: the mask of da (PRE)
    Adb = Bdb = Adm = Bdm = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto put1 = [&](float *d1, float4 v) {  // dy1 unshifted: [quad][column block][16 n][4 k]
        float *p = d1 + (co_s >> 2) * 128 + (co_s & 3) + s2_slot<H>(img_s, row_s) + 16 * half_s;
        p[0] = v.x, p[4] = v.y, p[8] = v.z, p[12] = v.w;
    };
    auto put = [&](float *ds, float4 v) {  // shift (sy, sx) holds dy[i + sy][j + sx] at (i, j); shift stride 512 floats
        float *base = ds + (co_s >> 2) * 128 + (co_s & 3);
        if (H == 4) {  // the 2x2 plane: v = dy[0][0], [0][1], [1][0], [1][1]
            float *p = base + s2_slot<H>(img_s, 0);
            p[0] = v.x, p[4] = v.y, p[8] = v.z, p[12] = v.w;  // (0, 0)
            p[512 + 0] = v.y, p[512 + 8] = v.w;               // (0, 1)
            p[1024 + 0] = v.z, p[1024 + 4] = v.w;             // (1, 0)
            p[1536 + 0] = v.w;                                // (1, 1)
        } else {  // four columns 4 half .. of dy row oy: (sy = 0) at small row i = oy, (sy = 1) at i = oy - 1 - inside the workgroup's rows
            const int r0 = row_s, c0 = 4 * half_s;
#pragma unroll
            for (int sy = 0; sy < 2; ++sy) {
                const int il = r0 - sy;
                if (il < 0 || il >= (H == 16 ? 4 : OH)) continue;
                float *p = base + sy * 1024 + s2_slot<H>(img_s, il) + 4 * c0;
                p[0] = v.x, p[4] = v.y, p[8] = v.z, p[12] = v.w;           // sx = 0: j = ox
                if (c0 > 0) p[512 - 4] = v.x;                              // sx = 1: j = ox - 1
                p[512 + 0] = v.y, p[512 + 4] = v.z, p[512 + 8] = v.w;
            }
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
#define S2_LOAD_D(S, round_)                                                                   \
    do {                                                                                       \
        const size_t do_ = static_cast<size_t>((round_) < rounds ? (round_) : rounds - 1) * dstep; \
        S##da = *reinterpret_cast<const float4 *>(dsrc + do_);                                 \
        if (DS) S##db = *reinterpret_cast<const float4 *>(dsrc1 + do_);                        \
        if constexpr (PRE) S##dm = *reinterpret_cast<const float4 *>(dsrcm + do_);             \
    } while (0)
#define S2_PUT_D(buf_, S, round_)                                                              \
    do {                                                                                       \
        if constexpr (PRE) {                                                                   \
            const int c_ = ((round_) < rounds ? (round_) : rounds - 1) * S2_CK + co_s;         \
            S##da = scale4(wtab[c_], mask4(S##da, S##dm));                                     \
            if (DS) S##db = scale4(wtab[S2_MAX_KC + c_], S##db);                               \
        }                                                                                      \
        if (live) put((buf_) + WFM, S##da);                                                    \
        if (live1) put1((buf_) + D1, S##db);                                                   \
    } while (0)
    auto multiply = [&](const float *cur) {
        const float *ap = cur + wave * (MT * 64) + 4 * l15 + lq, *bp = cur + WFM + wave * 128 + 4 * l15 + lq;
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
        if (DS) {
            const float *b1 = cur + D1 + wave * 128 + 4 * l15 + lq;
            const float u0 = b1[0], u1 = b1[64];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float a = ap[9 * (256 * MT) + m * 64];
                acc[0][0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, u0, acc[0][0][m], 0, 0, 0);
                acc[0][1][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, u1, acc[0][1][m], 0, 0, 0);
            }
        }
    };
    float *buf0 = lds, *buf1 = lds + BUF;
    if (producer) {
        S2_LOAD_W(A, 0);
        S2_LOAD_D(A, 0);
    }
    __syncthreads();  // the zero fill
    if (producer) {
        S2_STORE_W(buf0, A);
        S2_PUT_D(buf0, A, 0);
        S2_LOAD_W(B, 1);
        S2_LOAD_D(B, 1);
        S2_LOAD_W(A, 2);
        S2_LOAD_D(A, 2);
    }
    __syncthreads();
    for (int round = 0; round < rounds; round += 2) {
        if (producer) {  // round + 1 (set B) -> buffer 1 while buffer 0 is multiplied
            S2_STORE_W(buf1, B);
            S2_PUT_D(buf1, B, round + 1);
            S2_LOAD_W(B, round + 3);
            S2_LOAD_D(B, round + 3);
        } else {
            multiply(buf0);
        }
        __syncthreads();
        if (round + 1 < rounds) {
            if (producer) {  // round + 2 (set A) -> buffer 0 while buffer 1 is multiplied
                S2_STORE_W(buf0, A);
                S2_PUT_D(buf0, A, round + 2);
                S2_LOAD_W(A, round + 4);
                S2_LOAD_D(A, round + 4);
            } else {
                multiply(buf1);
            }
            __syncthreads();
        }
    }
#undef S2_LOAD_D
#undef S2_PUT_D
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
    if (wave == 2 || wave == 3) spill(lds + (wave - 2) * OW);
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
    // the workgroup's RB channels x 128 large-map pixels = 512 MT float4 (4 consecutive x of one row): MT per thread
#pragma unroll
    for (int k = 0; k < MT; ++k) {
        const int f = threadIdx.x + S2_NT * k;
        // (image, channel, large-map row yy inside the workgroup's part, float4 xq of the row)
        constexpr int F4R = H / 4, ROWS = H == 16 ? 8 : H, F4I = RB * ROWS * F4R;
        const int img = f / F4I, g = f - img * F4I, ci = g / (ROWS * F4R), q = g - ci * (ROWS * F4R), yy = q / F4R, xq = q - yy * F4R;
        const int i = yy >> 1, py = yy & 1, sl = s2_slot<H>(img, i);  // block * 64 + 4 * (pixel of (i, 0))
        float out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int xx = 4 * xq + e, j = xx >> 1, c = py * 2 + (xx & 1), nt = sl >> 6, n = ((sl & 63) >> 2) + j;
            const int o = ((c * 2 + nt) * RB + ci) * S2_OS + n;
            out[e] = lds[o] + lds[OW + o];
        }
        if (b0 + img < d.B)
            *reinterpret_cast<float4 *>(dx + (static_cast<size_t>(b0 + img) * d.RC + ci0 + ci) * (H * H) + (8 * hh + yy) * H + 4 * xq) =
                make_float4(out[0], out[1], out[2], out[3]);
    }
}

// 16 MT result channels per workgroup.  B = 100, un-profiled us (scripts/s2_probe.py), MT = 2 / 1: forward 16x16 20.2 / 22.7, 8x8 17.2 / 22.3,
// 4x4 18.1 / 19.7; backward-data 16x16 20.2 / 18.6, 8x8 16.2 / 15.5, 4x4 27.7 / 18.1 (104 workgroups at MT = 2) -> 2 forward, 1 backward.
// EEADV_S2_MT = six digits (forward 16x16, 8x8, 4x4, backward 16x16, 8x8, 4x4) overrides, for A/B runs.
int s2_mt(bool bwd, int H) {
    const char *env = std::getenv("EEADV_S2_MT");  // read per call (tests flip it)
    const int slot = (bwd ? 3 : 0) + (H == 16 ? 0 : (H == 8 ? 1 : 2));
    if (env && std::strlen(env) == 6 && (env[slot] == '1' || env[slot] == '2')) return env[slot] - '0';
    return bwd ? 1 : 2;
}

template <int H, int MT, bool DS, bool FUSED>
int s2_launch(bool bwd, const float *in, const float *in1, const float *w9, float *out, float *out1, const S2Dims &d, hipStream_t st, const void *f0, const void *f1) {
    const dim3 grid(xcd_grid(H == 16 ? 2 * d.B : (d.B + S2Geo<H>::IMG - 1) / S2Geo<H>::IMG, d.RC / (16 * MT), d.wl));
    if (bwd) {
        const FusePre pre = FUSED ? *static_cast<const FusePre *>(f0) : FusePre{}, pre1 = FUSED && f1 ? *static_cast<const FusePre *>(f1) : FusePre{};
        EE_LAUNCH((conv3s2_bwd_mfma_kernel<H, MT, DS, FUSED>), grid, dim3(S2_NT), 0, st, in, in1, w9, out, d, pre, pre1);
    } else {
        const FusePost post = FUSED ? *static_cast<const FusePost *>(f0) : FusePost{}, post1 = FUSED && f1 ? *static_cast<const FusePost *>(f1) : FusePost{};
        constexpr size_t bytes = 2 * ((DS ? 10 : 9) * 4 * MT * 64 + S2_XS) * sizeof(float);  // 54-76 KB: above the static limit
        static int ok = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3s2_fwd_mfma_kernel<H, MT, DS, FUSED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            static_cast<int>(bytes)) == hipSuccess;
        if (!ok) return EE_ERR_UNSUPPORTED;
        EE_LAUNCH((conv3s2_fwd_mfma_kernel<H, MT, DS, FUSED>), grid, dim3(S2_NT), bytes, st, in, w9, out, out1, d, post, post1);
    }
    return launch_status();
}

// the pair's forward with the statistics epilogue on y3 (MT as the plain forward)
template <int H, int MT>
int s2_launch_stats(const float *x, const float *w10, float *y3, float *y1, float *stats, const S2Dims &d, hipStream_t st) {
    const dim3 grid(xcd_grid(H == 16 ? 2 * d.B : (d.B + S2Geo<H>::IMG - 1) / S2Geo<H>::IMG, d.RC / (16 * MT), d.wl));
    constexpr size_t bytes = 2 * (10 * 4 * MT * 64 + S2_XS) * sizeof(float);
    static int ok = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3s2_fwd_mfma_kernel<H, MT, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(bytes)) == hipSuccess;
    if (!ok) return EE_ERR_UNSUPPORTED;
    EE_LAUNCH((conv3s2_fwd_mfma_kernel<H, MT, true, false, true>), grid, dim3(S2_NT), bytes, st, x, w10, y3, y1, d, FusePost{}, FusePost{}, stats);
    return launch_status();
}

// f0 / f1: null (the plain convolutions) or the two FusePost (forward) / FusePre (backward-data) of the eval-mode BatchNorms folded in
template <bool DS, bool FUSED>
int s2_dispatch_f(bool bwd, const float *in, const float *in1, const float *w9, float *out, float *out1, const S2Dims &d, int H, hipStream_t st, const void *f0, const void *f1) {
    const int mt = s2_mt(bwd, H);
    if (H == 16) return mt == 2 ? s2_launch<16, 2, DS, FUSED>(bwd, in, in1, w9, out, out1, d, st, f0, f1) : s2_launch<16, 1, DS, FUSED>(bwd, in, in1, w9, out, out1, d, st, f0, f1);
    if (H == 8) return mt == 2 ? s2_launch<8, 2, DS, FUSED>(bwd, in, in1, w9, out, out1, d, st, f0, f1) : s2_launch<8, 1, DS, FUSED>(bwd, in, in1, w9, out, out1, d, st, f0, f1);
    return mt == 2 ? s2_launch<4, 2, DS, FUSED>(bwd, in, in1, w9, out, out1, d, st, f0, f1) : s2_launch<4, 1, DS, FUSED>(bwd, in, in1, w9, out, out1, d, st, f0, f1);
}

template <bool DS>
int s2_dispatch(bool bwd, const float *in, const float *in1, const float *w9, float *out, float *out1, const S2Dims &d, int H, hipStream_t st) {
    return s2_dispatch_f<DS, false>(bwd, in, in1, w9, out, out1, d, H, st, nullptr, nullptr);
}

int s2_check(const void *a, const void *b, const void *c, int B, int KC, int RC, int H) {
    if (B < 0 || KC < 1 || RC < 1) return EE_ERR_SHAPE;
    if (KC % S2_CK != 0 || RC % S2_RB != 0 || (H != 4 && H != 8 && H != 16)) return EE_ERR_UNSUPPORTED;
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
    const S2Dims d{B, Cin, Cout, xcd_weights_local(4.0 * B * Cin * H * H, 36.0 * Cin * Cout, Cout / 32) ? 1 : 0};
    ProfScope prof(EE_K_CONV3S2_FWD, as_stream(stream), 2.0 * 9.0 * Cin * Cout * static_cast<double>(B) * (H / 2) * (H / 2));
    return s2_dispatch<false>(false, x, nullptr, w9, y, nullptr, d, H, as_stream(stream));
}

// dy [B][Cout][H/2][H/2], w9 = the filters as [Cin/32][Cout/16][9][4][2][16][4] (w9[..][t][q][h][m][k] = weight[16 round + 4 q + k]
// [32 cb + 16 h + m][t]) -> dx [B][Cin][H][H] (H = 8 or 4; all of it is written)
EE_API int ee_conv3x3s2_small_bwd_data_f32(const float *dy, const float *w9, float *dx, int B, int Cin, int Cout, int H, void *stream) {
    const int rc = s2_check(dy, w9, dx, B, Cout, Cin, H);
    if (rc != EE_OK || B == 0) return rc;
    const S2Dims d{B, Cout, Cin, xcd_weights_local(1.0 * B * Cout * H * H, 36.0 * Cin * Cout, Cin / 32) ? 1 : 0};
    ProfScope prof(EE_K_CONV3S2_BWD, as_stream(stream), 2.0 * 9.0 * Cin * Cout * static_cast<double>(B) * (H / 2) * (H / 2));
    return s2_dispatch<false>(true, dy, nullptr, w9, dx, nullptr, d, H, as_stream(stream));
}

// The first convolution of a down-sampling BasicBlock TOGETHER with the block's shortcut convolution (resnet.py:26-31, :137-142): both read
// the same x, and the 1x1 / stride 2 filter sees exactly the 3x3's centre tap.  w10 = the filters as [Cout/32][Cin/16][10][4][2][16][4], taps 0..8
// the 3x3's (as w9 above), tap 9 the 1x1's (forward: weight1[32 cb + 16 h + m][16 rd + 4 q + k]).
//   x [B][Cin][H][H] -> y3 = conv3x3s2(x) and y1 = conv1x1s2(x), both [B][Cout][H/2][H/2]
EE_API int ee_conv3x3s2_pair_fwd_f32(const float *x, const float *w10, float *y3, float *y1, int B, int Cin, int Cout, int H, void *stream) {
    const int rc = s2_check(x, w10, y3, B, Cin, Cout, H);
    if (rc != EE_OK || B == 0) return rc;
    if (!y1) return EE_ERR_NULL;
    if (!aligned16(y1)) return EE_ERR_ALIGN;
    const S2Dims d{B, Cin, Cout, xcd_weights_local(4.0 * B * Cin * H * H, 40.0 * Cin * Cout, Cout / 32) ? 1 : 0};
    ProfScope prof(EE_K_CONV3S2_FWD, as_stream(stream), 2.0 * 10.0 * Cin * Cout * static_cast<double>(B) * (H / 2) * (H / 2));
    return s2_dispatch<true>(false, x, nullptr, w10, y3, y1, d, H, as_stream(stream));
}

// their backward-data in one pass: dx = conv3x3s2^T(dy3) + conv1x1s2^T(dy1).  w10 [Cin/32][Cout/16][10][4][2][16][4] (backward order: tap 9 =
// weight1[16 rd + 4 q + k][32 cb + 16 h + m])
EE_API int ee_conv3x3s2_pair_bwd_data_f32(const float *dy3, const float *dy1, const float *w10, float *dx, int B, int Cin, int Cout, int H, void *stream) {
    const int rc = s2_check(dy3, w10, dx, B, Cout, Cin, H);
    if (rc != EE_OK || B == 0) return rc;
    if (!dy1) return EE_ERR_NULL;
    if (!aligned16(dy1)) return EE_ERR_ALIGN;
    const S2Dims d{B, Cout, Cin, xcd_weights_local(2.0 * B * Cout * H * H, 40.0 * Cin * Cout, Cin / 32) ? 1 : 0};
    ProfScope prof(EE_K_CONV3S2_BWD, as_stream(stream), 2.0 * 10.0 * Cin * Cout * static_cast<double>(B) * (H / 2) * (H / 2));
    return s2_dispatch<true>(true, dy3, dy1, w10, dx, nullptr, d, H, as_stream(stream));
}

// The pair above with the eval-mode BatchNorms behind the two convolutions (running statistics) and the ReLU folded into the output stage:
//     y3 = relu( bn1(conv3x3s2(x)) ),   y1 = bn_ds(conv1x1s2(x))                               resnet.py:50-59, :137-142 under model.eval()
// - ee_conv3x3s2_pair_fwd_f32 followed by ee_bn_act_fwd_f32(training = 0) on each output, bit for bit, in one launch.  mean / var / gamma / beta: [Cout].
EE_API int ee_conv3x3s2_pair_bn_eval_fwd_f32(const float *x, const float *w10, const float *mean3, const float *var3, const float *gamma3, const float *beta3,
                                             float eps3, const float *mean1, const float *var1, const float *gamma1, const float *beta1, float eps1, float *y3,
                                             float *y1, int B, int Cin, int Cout, int H, void *stream) {
    const int rc = s2_check(x, w10, y3, B, Cin, Cout, H);
    if (rc != EE_OK || B == 0) return rc;
    if (!y1 || !mean3 || !var3 || !mean1 || !var1) return EE_ERR_NULL;
    if (!aligned16(y1)) return EE_ERR_ALIGN;
    const FusePost p3{mean3, var3, gamma3, beta3, eps3, nullptr, 1}, p1{mean1, var1, gamma1, beta1, eps1, nullptr, 0};
    const S2Dims d{B, Cin, Cout, xcd_weights_local(4.0 * B * Cin * H * H, 40.0 * Cin * Cout, Cout / 32) ? 1 : 0};
    ProfScope prof(EE_K_CONV3S2_FWD, as_stream(stream), 2.0 * 10.0 * Cin * Cout * static_cast<double>(B) * (H / 2) * (H / 2));
    return s2_dispatch_f<true, true>(false, x, nullptr, w10, y3, y1, d, H, as_stream(stream), &p3, &p1);
}

// ... and its backward-data, given the gradients of y3 and y1 and y3 itself (the ReLU mask):
//     dx = conv3x3s2^T( gamma3 / sqrt(var3 + eps3) * (y3 > 0) * dy3 ) + conv1x1s2^T( gamma1 / sqrt(var1 + eps1) * dy1 )
// - two ee_bn_act_bwd_f32(training = 0) and ee_conv3x3s2_pair_bwd_data_f32 in one launch.  var / gamma: [Cout] (<= 512).
EE_API int ee_conv3x3s2_pair_bn_eval_bwd_f32(const float *dy3, const float *y3, const float *dy1, const float *w10, const float *var3, const float *gamma3,
                                             float eps3, const float *var1, const float *gamma1, float eps1, float *dx, int B, int Cin, int Cout, int H,
                                             void *stream) {
    const int rc = s2_check(dy3, w10, dx, B, Cout, Cin, H);
    if (rc != EE_OK || B == 0) return rc;
    if (!dy1 || !y3 || !var3 || !var1) return EE_ERR_NULL;
    if (!aligned16(dy1) || !aligned16(y3)) return EE_ERR_ALIGN;
    if (Cout > S2_MAX_KC) return EE_ERR_UNSUPPORTED;
    const FusePre p3{nullptr, y3, nullptr, var3, gamma3, eps3}, p1{nullptr, nullptr, nullptr, var1, gamma1, eps1};
    const S2Dims d{B, Cout, Cin, xcd_weights_local(2.0 * B * Cout * H * H, 40.0 * Cin * Cout, Cin / 32) ? 1 : 0};
    ProfScope prof(EE_K_CONV3S2_BWD, as_stream(stream), 2.0 * 10.0 * Cin * Cout * static_cast<double>(B) * (H / 2) * (H / 2));
    return s2_dispatch_f<true, true>(true, dy3, dy1, w10, dx, nullptr, d, H, as_stream(stream), &p3, &p1);
}

// The pair's forward with the statistics of y3 for the TRAIN-mode BatchNorm behind it (ee_wino3x3_bn_train_pre_f32 consumes them): next to the raw
// y3 / y1, stats [Cout][S][2] = (mean, M2) per result channel and partial - H = 16: S = 2 B (half images, 32 values each); H = 8: S = B (16 values).
EE_API int ee_conv3x3s2_pair_stats_fwd_f32(const float *x, const float *w10, float *y3, float *y1, float *stats, int B, int Cin, int Cout, int H, void *stream) {
    const int rc = s2_check(x, w10, y3, B, Cin, Cout, H);
    if (rc != EE_OK || B == 0) return rc;
    if (H != 16 && H != 8) return EE_ERR_UNSUPPORTED;
    if (!y1 || !stats) return EE_ERR_NULL;
    if (!aligned16(y1) || (reinterpret_cast<uintptr_t>(stats) & 7u)) return EE_ERR_ALIGN;
    const S2Dims d{B, Cin, Cout, xcd_weights_local(4.0 * B * Cin * H * H, 40.0 * Cin * Cout, Cout / 32) ? 1 : 0};
    ProfScope prof(EE_K_CONV3S2_FWD, as_stream(stream), 2.0 * 10.0 * Cin * Cout * static_cast<double>(B) * (H / 2) * (H / 2));
    const int mt = s2_mt(false, H);
    if (H == 16) return mt == 2 ? s2_launch_stats<16, 2>(x, w10, y3, y1, stats, d, as_stream(stream)) : s2_launch_stats<16, 1>(x, w10, y3, y1, stats, d, as_stream(stream));
    return mt == 2 ? s2_launch_stats<8, 2>(x, w10, y3, y1, stats, d, as_stream(stream)) : s2_launch_stats<8, 1>(x, w10, y3, y1, stats, d, as_stream(stream));
}
