// ee_dense.hip - Conv2d(3x3, stride 1, padding 1, bias=False) on a 2x2 map (ResNet-18's layer4 at 64x64 inputs, resnet.py:26-31) as ONE dense
// product on the f32 matrix cores, with the eval-mode BatchNorm / residual / ReLU of the block folded in (ee_fuse.hpp).
//
// On a 2x2 map every input pixel reaches every output pixel: y[n][(co, p)] = sum_{(ci, q)} x[n][(ci, q)] * W2[(ci, q)][(co, p)] with
// W2 = the weights rearranged once per optimiser step (functional._rearranged kind "s1": [4 Cin][4 Cout]; the backward-data product reads
// its transpose, kind "s1t").  Rounds 2-3 ran these three layers as torch.mm (a Tensile kernel, 14.5 us for 100 x 2048 x 2048) between
// BatchNorm launches; here the product is hand-written so that the BatchNorm of an eval-mode pass costs nothing extra:
//   forward   y = [relu]( (x W2 - mean) * gamma / sqrt(var + eps) + beta [+ res] )                                  one launch
//   backward  dz = (y > 0) * (dy [+ dy2]);  dres <- dz;  dx = (gamma / sqrt(var + eps) * dz) W2^T [+ dx_add]        one launch
// A workgroup owns a 32 x 32 tile of the result (images x result columns: 4 x 64 = 256 workgroups at batch 100, 2048 columns) and walks
// the reduction in rounds of 64: the A tile [32][64] and the B tile [64][32] of the NEXT round travel from L2 into registers while the
// current one is multiplied out of LDS (two buffers, one barrier per round); the four wavefronts split the round's 64 reduction indices
// (16 each: four k-steps x 2 x 2 accumulator tiles = 16 MFMAs, one LDS read per MFMA) and their partial tiles meet in LDS, added in
// wavefront order (bit-reproducible).  LDS strides 68 / 48 put the 64 lanes of an operand read on 64 distinct banks.
// A float4 of A or of the result lies inside ONE channel (4 pixels of a 2x2 plane), which is what makes the per-channel pre / post maps cheap.
//
// Round 4 also tried this kernel as the ONE-launch head backward of MNIST's Net_2 (logits of a tile's rows on MFMA in a prologue, the cross-entropy
// gradient through fc2^T and the ReLU gate formed while fc1's output is staged, then the product with fc1's weight): correct, and 19.8 us against
// 16.0 us for ee_fc_ce_grad_f32 + the Tensile product it would replace - each of the 32 column tiles repeats its rows' logits (6.7 us) and forms
// the gradient while staging (3.7 us).  Not kept: commit 0905c3e holds it, profiles/round4_q_net2_head_one_launch.txt the phase timings.
//
// CNN-body glue, not a row of SURVEY.md section 8.
#include "ee_common.hpp"
#include "ee_fuse.hpp"

namespace {

using namespace ee;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DN_NT = 512, DN_PT = 256, DN_TM = 32, DN_TN = 32, DN_KU = 128, DN_DEPTH = 4;  // 4 multiplying + 4 producing wavefronts
constexpr int DN_SA = DN_KU + 4;   // A tile row stride [m][k]: bank = (4 m + k) % 64 over the 16 x 4 lanes of an operand read
constexpr int DN_SB = DN_TN + 16;  // B tile row stride [k][n]: the four k of an operand read on disjoint banks
constexpr int DN_ABUF = DN_TM * DN_SA, DN_BBUF = DN_KU * DN_SB;
constexpr int DN_RS = 36;          // row stride of the partial-tile exchange
constexpr int DN_MAX_C = 512;      // PRE: channels whose gamma * invstd table fits (2 KB)
constexpr int DN_F4 = DN_TM * DN_KU / 4 / DN_PT;  // float4 of the A tile (and of the B tile) per producing lane and round: 4

struct DenseDims {
    int M, K, N;  // images, reduction (4 x channels in), result columns (4 x channels out)
};

// PRE: the staged A is  gamma / sqrt(var + eps) [k >> 2] * (mask > 0 ? a [+ add] : 0), the masked sum also written to pre.store by the
// workgroups of the first column block;  POST: the stored result is [relu]((c - mean) * invstd * gamma + beta [+ res]) (post.mean) or
// c + res (post.mean null).
// Rounds of 128 reduction indices; the tiles of the next FOUR rounds are in flight in four register sets (a round's products take
// ~0.4 us, the weights come from the Infinity Cache at ~1 us), written to the other LDS buffer one round ahead; one barrier per round.
template <bool PRE, bool POST>
__global__ __launch_bounds__(DN_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void dense_mfma_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ c, DenseDims d,
                                                           FusePre pre, FusePost post) {
    extern __shared__ __align__(16) float lds[];  // two buffers of {A tile, B tile}; afterwards the partial-tile exchange
    __shared__ float wtab[PRE ? DN_MAX_C : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const bool producer = wave >= 4;   // wavefronts 0-3 multiply, 4-7 keep four rounds of loads in flight and write the next round's tiles to LDS:
    const int tid = threadIdx.x & 255; // a wavefront issues in order, so with do-everything wavefronts load latency, staging and products add up
    // blockIdx.x -> (n tile, m tile): the few m tiles of one n tile are neighbours in the numbering
    const int mtiles = (d.M + DN_TM - 1) / DN_TM;
    const int nt = blockIdx.x / mtiles, mt = blockIdx.x - nt * mtiles;
    const int m0 = mt * DN_TM, n0 = nt * DN_TN;
    const int rounds = d.K / DN_KU;  // a multiple of DN_DEPTH (dense_check)
    if constexpr (PRE) {
        for (int ch = threadIdx.x; ch < d.K / 4; ch += DN_NT) wtab[ch] = bn_scale(pre.var, pre.gamma, pre.eps, ch);
    }
    // staging roles: A tile 32 rows x 32 float4: thread t takes (row (t >> 5) + 8 j, float4 t & 31); B tile 128 k-rows x 8 float4: (k row (t >> 3) + 32 j, float4 t & 7)
    const int ar = tid >> 5, af = tid & 31, br = tid >> 3, bf = tid & 7;
    size_t ao[DN_F4];
    bool ast[DN_F4];
#pragma unroll
    for (int j = 0; j < DN_F4; ++j) {
        const int m = m0 + ar + 8 * j;
        ao[j] = static_cast<size_t>(m < d.M ? m : d.M - 1) * d.K + 4 * af;  // past the batch: a valid row, never stored
        ast[j] = PRE && pre.store != nullptr && nt == 0 && m < d.M;
    }
    const float *bsrc = b + static_cast<size_t>(br) * d.N + n0 + 4 * bf;
    const size_t bstep32 = static_cast<size_t>(32) * d.N, bround = static_cast<size_t>(DN_KU) * d.N;
    // NAMED registers (as arrays indexed through a lambda they went to scratch memory): the weights (B: from the Infinity Cache, ~1 us away) travel
    // FOUR rounds ahead in four sets; the activations and their mask (A, M: written by the previous kernel, L2-resident) two rounds ahead in two
    static_assert(DN_F4 == 4, "four float4 of each tile per producing lane");
    float4 RA0_0, RA0_1, RA0_2, RA0_3, RA1_0, RA1_1, RA1_2, RA1_3;
    float4 RM0_0, RM0_1, RM0_2, RM0_3, RM1_0, RM1_1, RM1_2, RM1_3;
    float4 RB0_0, RB0_1, RB0_2, RB0_3, RB1_0, RB1_1, RB1_2, RB1_3, RB2_0, RB2_1, RB2_2, RB2_3, RB3_0, RB3_1, RB3_2, RB3_3;
#define DN_FETCH_A1(AS, J, ko_)                                                                \
    do {                                                                                       \
        RA##AS##_##J = *reinterpret_cast<const float4 *>(a + ao[J] + (ko_));                   \
        if constexpr (PRE) RM##AS##_##J = *reinterpret_cast<const float4 *>(pre.mask + ao[J] + (ko_)); \
    } while (0)
#define DN_FETCH_A(AS, round_)                                                                 \
    do {                                                                                       \
        const size_t ko_ = static_cast<size_t>((round_) < rounds ? (round_) : rounds - 1) * DN_KU; /* always issued */ \
        DN_FETCH_A1(AS, 0, ko_);                                                               \
        DN_FETCH_A1(AS, 1, ko_);                                                               \
        DN_FETCH_A1(AS, 2, ko_);                                                               \
        DN_FETCH_A1(AS, 3, ko_);                                                               \
    } while (0)
#define DN_FETCH_B(S, round_)                                                                  \
    do {                                                                                       \
        const float *bp_ = bsrc + ((round_) < rounds ? (round_) : rounds - 1) * bround;        \
        RB##S##_0 = *reinterpret_cast<const float4 *>(bp_);                                    \
        RB##S##_1 = *reinterpret_cast<const float4 *>(bp_ + bstep32);                          \
        RB##S##_2 = *reinterpret_cast<const float4 *>(bp_ + 2 * bstep32);                      \
        RB##S##_3 = *reinterpret_cast<const float4 *>(bp_ + 3 * bstep32);                      \
    } while (0)
#define DN_STAGE1(buf_, AS, BS, J, round_)                                                     \
    do {                                                                                       \
        float4 v_ = RA##AS##_##J;                                                              \
        if constexpr (PRE) {                                                                   \
            const size_t ko_ = static_cast<size_t>(round_) * DN_KU;                            \
            /* two gradient pieces: rare, fetched here */                                      \
            if (pre.add) v_ = sum4(v_, *reinterpret_cast<const float4 *>(pre.add + ao[J] + ko_)); \
            v_ = mask4(v_, RM##AS##_##J);                                                      \
            if (ast[J]) *reinterpret_cast<float4 *>(pre.store + ao[J] + ko_) = v_;             \
            v_ = scale4(wtab[((round_) * DN_KU + 4 * af) >> 2], v_);                           \
        }                                                                                      \
        *reinterpret_cast<float4 *>((buf_) + (ar + 8 * J) * DN_SA + 4 * af) = v_;              \
        *reinterpret_cast<float4 *>((buf_) + DN_ABUF + (br + 32 * J) * DN_SB + 4 * bf) = RB##BS##_##J; \
    } while (0)
#define DN_STAGE(buf_, AS, BS, round_)                                                         \
    do {                                                                                       \
        DN_STAGE1(buf_, AS, BS, 0, round_);                                                    \
        DN_STAGE1(buf_, AS, BS, 1, round_);                                                    \
        DN_STAGE1(buf_, AS, BS, 2, round_);                                                    \
        DN_STAGE1(buf_, AS, BS, 3, round_);                                                    \
    } while (0)
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    auto multiply = [&](const float *buf) {
        // this wavefront's 32 reduction indices of the round: k = 32 wave + 4 s + lq
        const float *ap = buf + l15 * DN_SA + 32 * wave + lq, *bp = buf + DN_ABUF + (32 * wave + lq) * DN_SB + l15;
        float av[8][2], bv[8][2];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            av[s][0] = ap[4 * s], av[s][1] = ap[16 * DN_SA + 4 * s];
            bv[s][0] = bp[4 * s * DN_SB], bv[s][1] = bp[4 * s * DN_SB + 16];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][i], bv[s][j], acc[i][j], 0, 0, 0);
    };
    constexpr int BUF = DN_ABUF + DN_BBUF;
    static_assert(DN_DEPTH == 4, "four named register sets");
    if (producer) {
        DN_FETCH_A(0, 0);
        DN_FETCH_A(1, 1);
        DN_FETCH_B(0, 0);
        DN_FETCH_B(1, 1);
        DN_FETCH_B(2, 2);
        DN_FETCH_B(3, 3);
    }
    if constexpr (PRE) __syncthreads();  // the scale table
    if (producer) {
        DN_STAGE(lds, 0, 0, 0);
        DN_FETCH_A(0, 2);
    }
    __syncthreads();
    // round r0 + S multiplies buffer S & 1 while the producers write round r0 + S + 1 (A set AS = (S + 1) & 1, B set SN = (S + 1) & 3) into the
    // other buffer and refill the sets that became free: A set AS with round + 3, B set S with round + 4
#define DN_STEP(S, SN, AS)                                                                     \
    do {                                                                                       \
        const int r_s = r0 + S;                                                                \
        if (producer) {                                                                        \
            if (r_s + 1 < rounds) DN_STAGE(lds + ((S + 1) & 1) * BUF, AS, SN, r_s + 1);        \
            DN_FETCH_A(AS, r_s + 3);                                                           \
            DN_FETCH_B(S, r_s + 4);                                                            \
        } else {                                                                               \
            multiply(lds + (S & 1) * BUF);                                                     \
        }                                                                                      \
        __syncthreads();                                                                       \
    } while (0)
    for (int r0 = 0; r0 < rounds; r0 += 4) {
        DN_STEP(0, 1, 1);
        DN_STEP(1, 2, 0);
        DN_STEP(2, 3, 1);
        DN_STEP(3, 0, 0);
    }
#undef DN_STEP
#undef DN_STAGE
#undef DN_STAGE1
#undef DN_FETCH_B
#undef DN_FETCH_A
#undef DN_FETCH_A1
    // ---- the four wavefronts' partial tiles meet: red[wave][32][36]; D row = 4 lq + reg, column = l15 --------------------------------------
    float *red = lds;
    static_assert(4 * DN_TM * DN_RS <= 2 * BUF, "the exchange fits the staging buffers");
    if (!producer) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(wave * DN_TM + 16 * i + 4 * lq + r) * DN_RS + 16 * j + l15] = acc[i][j][r];
    }
    __syncthreads();
    if (!producer) {
        const int row = tid >> 3, f = tid & 7;  // one float4 of the 32 x 32 tile per thread
        const float *rp = red + row * DN_RS + 4 * f;
        float4 s = *reinterpret_cast<const float4 *>(rp);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 v = *reinterpret_cast<const float4 *>(rp + w * (DN_TM * DN_RS));
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
        if (m0 + row < d.M) {
            const size_t o = static_cast<size_t>(m0 + row) * d.N + n0 + 4 * f;
            if constexpr (POST) {
                if (post.mean) {
                    const PostConst k = post_const(post, (n0 + 4 * f) >> 2);
                    s.x = post_apply(s.x, k), s.y = post_apply(s.y, k), s.z = post_apply(s.z, k), s.w = post_apply(s.w, k);
                }
                if (post.res) {
                    const float4 q = *reinterpret_cast<const float4 *>(post.res + o);
                    s.x += q.x, s.y += q.y, s.z += q.z, s.w += q.w;
                }
                if (post.relu) s.x = relu_keep_nan(s.x), s.y = relu_keep_nan(s.y), s.z = relu_keep_nan(s.z), s.w = relu_keep_nan(s.w);
            }
            *reinterpret_cast<float4 *>(c + o) = s;
        }
    }
}

int dense_check(const void *a, const void *b, const void *c, int B, int Cin, int Cout) {
    if (B < 0 || Cin < 1 || Cout < 1) return EE_ERR_SHAPE;
    if ((4 * Cin) % (DN_KU * DN_DEPTH) != 0 || (4 * Cout) % DN_TN != 0) return EE_ERR_UNSUPPORTED;  // Cin % 128, Cout % 8
    if (B == 0) return EE_OK;
    if (!a || !b || !c) return EE_ERR_NULL;
    if (!aligned16(a) || !aligned16(b) || !aligned16(c)) return EE_ERR_ALIGN;
    if (static_cast<int64_t>(B) * 4 * (Cin > Cout ? Cin : Cout) > 0x7fffffffLL) return EE_ERR_SHAPE;
    return EE_OK;
}

template <bool PRE, bool POST>
int dense_launch(const float *a, const float *b, float *c, const DenseDims &d, const FusePre &pre, const FusePost &post, hipStream_t st) {
    const unsigned grid = static_cast<unsigned>((d.M + DN_TM - 1) / DN_TM) * static_cast<unsigned>(d.N / DN_TN);
    constexpr size_t bytes = 2 * (DN_ABUF + DN_BBUF) * sizeof(float);  // 83 KB: above the static limit
    static int ok = hipFuncSetAttribute(reinterpret_cast<const void *>(dense_mfma_kernel<PRE, POST>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(bytes)) == hipSuccess;
    if (!ok) return EE_ERR_UNSUPPORTED;
    EE_LAUNCH((dense_mfma_kernel<PRE, POST>), dim3(grid), dim3(DN_NT), bytes, st, a, b, c, d, pre, post);
    return launch_status();
}

}  // namespace

// y [B][Cout][2][2] = conv3x3(x [B][Cin][2][2]) as x[B][4 Cin] . w2[4 Cin][4 Cout] (w2: EE_WPREP_DENSE_S1, or its transpose for the
// backward-data product with Cin / Cout exchanged).  Cin a multiple of 128, Cout of 8.
EE_API int ee_dense2x2_f32(const float *x, const float *w2, float *y, int B, int Cin, int Cout, void *stream) {
    const int rc = dense_check(x, w2, y, B, Cin, Cout);
    if (rc != EE_OK || B == 0) return rc;
    return dense_launch<false, false>(x, w2, y, DenseDims{B, 4 * Cin, 4 * Cout}, FusePre{}, FusePost{}, as_stream(stream));
}

// ... with the eval-mode BatchNorm (running statistics), the block's residual and the ReLU in the epilogue (see ee_wino3x3_bn_eval_fwd_f32)
EE_API int ee_dense2x2_bn_eval_fwd_f32(const float *x, const float *w2, const float *mean, const float *var, const float *gamma, const float *beta, float eps,
                                       const float *res, int relu, float *y, int B, int Cin, int Cout, void *stream) {
    const int rc = dense_check(x, w2, y, B, Cin, Cout);
    if (rc != EE_OK || B == 0) return rc;
    const FusePost post{mean, var, gamma, beta, eps, res, relu};
    if (!mean || !var) return EE_ERR_NULL;
    const int pc = check_post(post);
    if (pc != EE_OK) return pc;
    return dense_launch<false, true>(x, w2, y, DenseDims{B, 4 * Cin, 4 * Cout}, FusePre{}, post, as_stream(stream));
}

// ... and its backward-data: dz = (y > 0) * (dy [+ dy2]); dres <- dz; dx = (gamma / sqrt(var + eps) * dz) . w2t [+ dx_add]
// (w2t [4 Cout][4 Cin] = the transpose of the forward's w2; see ee_wino3x3_bn_eval_bwd_f32).  Cout <= 512.
EE_API int ee_dense2x2_bn_eval_bwd_f32(const float *dy, const float *dy2, const float *y, const float *w2t, const float *var, const float *gamma, float eps,
                                       float *dres, const float *dx_add, float *dx, int B, int Cin, int Cout, void *stream) {
    const int rc = dense_check(dy, w2t, dx, B, Cout, Cin);
    if (rc != EE_OK || B == 0) return rc;
    if (!y) return EE_ERR_NULL;
    if (Cout > DN_MAX_C) return EE_ERR_UNSUPPORTED;
    const FusePre pre{dy2, y, dres, var, gamma, eps};
    const int pc = check_pre(pre);
    if (pc != EE_OK) return pc;
    const DenseDims d{B, 4 * Cout, 4 * Cin};
    if (dx_add) {
        if (!aligned16(dx_add)) return EE_ERR_ALIGN;
        return dense_launch<true, true>(dy, w2t, dx, d, pre, FusePost{nullptr, nullptr, nullptr, nullptr, 0.0f, dx_add, 0}, as_stream(stream));
    }
    return dense_launch<true, false>(dy, w2t, dx, d, pre, FusePost{}, as_stream(stream));
}
