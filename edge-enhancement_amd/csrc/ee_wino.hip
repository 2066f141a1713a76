// ee_wino.hip - Conv2d(3x3, stride 1, padding 1, bias=False) on 8x8 and 16x16 maps (ResNet-18 layer2 / layer1 at 64x64 inputs,
// resnet.py:26-31) as Winograd F(2x2, 3x3) around the f32 matrix cores.
//
// Why: the direct implicit-GEMM kernels of ee_conv.hip sit at the matrix pipe's effective rate for these occupancies;
// on the 128-channel 8x8 layers they only tie MIOpen's Winograd (31.9 vs 29.4 us for 1.9 GFLOP), which does 2.25x fewer multiplies on the
// vector ALUs.  Here the 16 element-wise products of the transform domain are 16 small GEMMs on v_mfma_f32_16x16x4_f32:
//     V = B^T d B  (4x4 input patch d of tile t, channel ci)        U = G g G^T  (3x3 filter g of (co, ci); precomputed, `u`)
//     M_xi[co][t] = sum_ci U_xi[co][ci] * V_xi[ci][t]   xi = 0..15     Y = A^T M A  (2x2 outputs of tile t, channel co)
// An 8x8 map is 4x4 = 16 tiles - exactly the N of one MFMA - so a workgroup owns ONE image x 32 result channels: per 16-channel round
// it stages the input patch plane (zero ring) and the U slice [16 xi][16 ci][32 co] in LDS, every lane transforms one (ci, tile) patch
// into V, and each of the 4 wavefronts multiplies its four xi (2 channel tiles x 4 k-steps = 32 MFMAs); the next round's global loads
// travel meanwhile.  The accumulators meet in LDS for the output transform.  Backward-data is the same kernel on U built from the
// flipped, transposed filter.  Exact f32 products; the transforms add a few 1e-7 of relative error (as MIOpen's solver does).
//
// CNN-body glue, not a row of SURVEY.md section 8.
#include "ee_common.hpp"
#include "ee_fuse.hpp"

namespace {

using namespace ee;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// -DEE_WINO_TIMING (scripts/wino_timing.py builds its own copy of this file with it; never the product): shader-clock stamps of one
// multiplying and one producing lane per workgroup - entry, loop start, loop end, and the cycles spent inside the round bodies (the rest
// of the loop being barrier waits); a second mode splits the producing lane's even rounds into U store | transform | pixel store + loads
#ifdef EE_WINO_TIMING
__device__ unsigned long long g_wino_stamps[8 * 2048];
__device__ int g_wino_sub;
#define WT_DECL unsigned long long wt_body = 0, wt_t = 0, wt_p[3] = {0, 0, 0}, wt_q = 0
#define WT_STAMP(k) do { if ((threadIdx.x & 63) == 0 && (wave == 0 || wave == PC_CW) && blockIdx.x < 2048) g_wino_stamps[blockIdx.x * 8 + (wave ? 4 : 0) + (k)] = (k) == 3 ? wt_body : clock64(); } while (0)
#define WT_BEGIN wt_t = clock64()
#define WT_END wt_body += clock64() - wt_t
#define WT_P0 wt_q = clock64()
#define WT_P(i) do { const unsigned long long n_ = clock64(); wt_p[i] += n_ - wt_q; wt_q = n_; } while (0)
#define WT_DUMP do { if ((g_wino_sub & 1) && (threadIdx.x & 63) == 0 && wave == PC_CW && blockIdx.x < 2048) { g_wino_stamps[blockIdx.x * 8 + 0] = wt_p[0]; g_wino_stamps[blockIdx.x * 8 + 1] = wt_p[1]; g_wino_stamps[blockIdx.x * 8 + 2] = wt_p[2]; } } while (0)
#else
#define WT_DECL
#define WT_STAMP(k)
#define WT_BEGIN
#define WT_END
#define WT_P0
#define WT_P(i)
#define WT_DUMP
#endif

// the float4 held by another lane of the same quad: DPP quad_perm CTRL = p0 | p1 << 2 | p2 << 4 | p3 << 6, lane i reads lane p_i
template <int CTRL>
__device__ __forceinline__ float4 quad_rot(float4 v) {
    auto mv = [](float f) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), CTRL, 0xF, 0xF, true)); };
    return make_float4(mv(v.x), mv(v.y), mv(v.z), mv(v.w));
}

constexpr int WN_CK = 16, WN_CO = 32;
constexpr int WN_CP = 48;                      // U row stride in LDS: the four k of a wavefront on disjoint banks (32 + 16)
constexpr int WN_US = WN_CK * WN_CP;           // one xi's [16 ci][48]

struct WinoDims {
    int B, KC, RC;  // reduction channels (input channels forward), result channels
    int wl;         // producer / consumer kernels: XCD numbering with the channel block fastest (ee_common.hpp: xcd_decode)
};

// ---- 4x4 maps (layer3 at 64x64 inputs), second version: producer and consumer wavefronts ------------------------------------------------
// 200 workgroups of four images = one per CU and, with 256 lanes, ONE wavefront per SIMD: a wavefront issues in order, so its loads' latency,
// its staging / transform work and its MFMAs simply add up (phase skipping on the generic kernel and on a double-buffered variant of it:
// 8.4 us of products + 6.6 staging + 6.4 waiting for loads + 10 of launch / prologue / epilogue, nothing hidden behind anything).  Here the
// workgroup has 512 lanes: wavefronts 0-3 only multiply (the generic kernel's roles: four xi each), wavefronts 4-7 only produce - they keep
// TWO rounds of U / input loads in flight in two named register sets, transform the input patches from registers (a lane owns one
// (channel, image, tile) patch and takes the three rows it needs of that 4x4 plane: no pixel staging in LDS) and write U and V of the
// NEXT round into the other LDS buffer while the consumers multiply the current one.  One barrier per round; each SIMD hosts one wavefront
// of either kind, so the hardware overlaps the two streams.  Same arithmetic in the same order as the generic kernel.
constexpr int W4_VS = WN_CK * 16;                 // one xi's V [16 ci][16 tiles]
constexpr int W4_BUF = 16 * WN_US + 16 * W4_VS;   // U [16][16][48] + V [16][16][16] = 16384 floats
constexpr size_t W4_LDS = 2 * W4_BUF * sizeof(float);
constexpr int W4_NT = 512;

// PRE / POST: eval-mode BatchNorm folded in (ee_fuse.hpp).  PRE 1: the staged input is mask > 0 ? x : 0 times the channel's gamma * invstd
// (BatchNorm + ReLU backward with running statistics), PRE 2: with a second input piece added first; both write the masked sum to pre.store
// (one workgroup per image) when it is given.  POST: the output transform applies (c - mean) * invstd * gamma + beta, the residual, the ReLU.
// PRE = 0, POST = false is the plain convolution, instruction for instruction what it was.
// PRE 3 / STATS: the train-mode pair (see wino3x3_pc_kernel; four images per workgroup: four lanes hold a 4x4 plane).
template <int PRE, bool POST, bool STATS = false>
__global__ __launch_bounds__(W4_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino3x3_map4_kernel(const float *__restrict__ x, const float *__restrict__ u, float *__restrict__ y, WinoDims d,
                                                                                                        FusePre pre, FusePost post, TrainBn tb = TrainBn{}) {
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool producer = wave >= 4;
    const int pt = threadIdx.x & 255;  // lane number inside its group of four wavefronts
    // 1-D grid numbered for the XCDs' L2s (ee_common.hpp: xcd_decode): with the channel block fastest an XCD fetches only the U slices of the
    // blocks congruent to it (4.2 MB of transformed filters were fetched by all eight: 28 MB of the 32 MB a launch pulled in)
    int bx, by;
    if (!xcd_decode(blockIdx.x, d.RC / WN_CO, (d.B + 3) / 4, d.wl, bx, by)) return;
    const int b = bx * 4, co0 = by * WN_CO;
    const int l15 = lane & 15, lq = lane >> 4;
    const int urow0 = pt >> 3, uq = pt & 7;
    const int ci_s = urow0 & 15, xi_s = urow0 >> 4;
    const float *ubase = u + (static_cast<size_t>(xi_s) * d.KC + ci_s) * d.RC + co0 + 4 * uq;
    const size_t uxi2 = 2 * static_cast<size_t>(d.KC) * d.RC;
    // patch of this producer lane: channel ci_x of the round, tile t_x = (image, ty, tx) of the workgroup's 16
    const int ci_x = pt >> 4, t_x = pt & 15, ty = (t_x >> 1) & 1, tx = t_x & 1;
    const int xb = b + (t_x >> 2) < d.B ? b + (t_x >> 2) : d.B - 1;  // past the batch: a valid image, never stored
    const size_t xo = (static_cast<size_t>(xb) * d.KC + ci_x) * 16 + 4 * ty;
    const float *xsrc = x + xo;  // rows ty .. ty+2 of the plane (patch rows 2ty-1 .. 2ty+2 minus the padding one)
    const size_t ustep = static_cast<size_t>(WN_CK) * d.RC, xstep = static_cast<size_t>(WN_CK) * 16;
    const int rounds = d.KC / WN_CK;
    float *wtab = lds + 2 * W4_BUF;  // PRE: gamma * invstd of every reduction channel
    if constexpr (PRE == 1 || PRE == 2) {
        for (int c = threadIdx.x; c < d.KC; c += W4_NT) wtab[c] = bn_scale(pre.var, pre.gamma, pre.eps, c);
        __syncthreads();
    }
    if constexpr (PRE == 3) {
        train_bn_merge<W4_NT>(tb, d.KC, wtab, blockIdx.x == 0);
        __syncthreads();
    }
    const unsigned xq = static_cast<unsigned>(xb * d.KC + ci_x) * 16u + 4u * (t_x & 3);  // PRE: row q = t_x & 3 of the lane's plane
    const bool pre_store = (PRE == 1 || PRE == 2) && pre.store != nullptr && by == 0 && b + (t_x >> 2) < d.B;
    // one round's prefetch of a producer lane: 8 float4 of U, 3 rows of its input plane - two sets (A, B) of NAMED registers filled by
    // straight-line code (as a struct handed to a lambda they lived in scratch memory)
    float4 Au0, Au1, Au2, Au3, Au4, Au5, Au6, Au7, Ax0, Ax1, Ax2, Bu0, Bu1, Bu2, Bu3, Bu4, Bu5, Bu6, Bu7, Bx0, Bx1, Bx2;
    float4 Am0, Bm0, Ay0, By0;  // PRE: the lane's mask row and its row of the second piece
    Am0 = Bm0 = Ay0 = By0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#define W4_FETCH(S, round_)                                                                        \
    do {                                                                                           \
        const int r_ = (round_) < rounds ? (round_) : rounds - 1; /* always issued */              \
        const float *up_ = ubase + r_ * ustep;                                                     \
        const float *xp_ = xsrc + r_ * xstep;                                                      \
        S##u0 = *reinterpret_cast<const float4 *>(up_);                                            \
        S##u1 = *reinterpret_cast<const float4 *>(up_ + uxi2);                                     \
        S##u2 = *reinterpret_cast<const float4 *>(up_ + 2 * uxi2);                                 \
        S##u3 = *reinterpret_cast<const float4 *>(up_ + 3 * uxi2);                                 \
        S##u4 = *reinterpret_cast<const float4 *>(up_ + 4 * uxi2);                                 \
        S##u5 = *reinterpret_cast<const float4 *>(up_ + 5 * uxi2);                                 \
        S##u6 = *reinterpret_cast<const float4 *>(up_ + 6 * uxi2);                                 \
        S##u7 = *reinterpret_cast<const float4 *>(up_ + 7 * uxi2);                                 \
        S##x0 = *reinterpret_cast<const float4 *>(xp_);                                            \
        S##x1 = *reinterpret_cast<const float4 *>(xp_ + 4);                                        \
        S##x2 = *reinterpret_cast<const float4 *>(xp_ + 8);                                        \
    } while (0)
    // PRE: a lane fetches ONE row of its plane - row q = its index in the quad of the plane's four tiles - of the gradient, the mask and
    // the second piece (instead of three rows of each: 2-3 float4 per round, not 6-9), applies the BatchNorm / ReLU backward to it and the
    // quad passes the rows around with three DPP rotations
#define W4_FETCH_ROW(S, round_)                                                                    \
    do {                                                                                           \
        const int r_ = (round_) < rounds ? (round_) : rounds - 1;                                  \
        const float *up_ = ubase + r_ * ustep;                                                     \
        S##u0 = *reinterpret_cast<const float4 *>(up_);                                            \
        S##u1 = *reinterpret_cast<const float4 *>(up_ + uxi2);                                     \
        S##u2 = *reinterpret_cast<const float4 *>(up_ + 2 * uxi2);                                 \
        S##u3 = *reinterpret_cast<const float4 *>(up_ + 3 * uxi2);                                 \
        S##u4 = *reinterpret_cast<const float4 *>(up_ + 4 * uxi2);                                 \
        S##u5 = *reinterpret_cast<const float4 *>(up_ + 5 * uxi2);                                 \
        S##u6 = *reinterpret_cast<const float4 *>(up_ + 6 * uxi2);                                 \
        S##u7 = *reinterpret_cast<const float4 *>(up_ + 7 * uxi2);                                 \
        S##x0 = *reinterpret_cast<const float4 *>((x + r_ * xstep) + xq);                          \
        if constexpr (PRE != 3) S##m0 = *reinterpret_cast<const float4 *>((pre.mask + r_ * xstep) + xq); \
        if constexpr (PRE == 2) S##y0 = *reinterpret_cast<const float4 *>((pre.add + r_ * xstep) + xq); \
    } while (0)
    // PRE: dz = mask > 0 ? (x + add) : 0 of the three rows, written out by the lanes that own them (tx = 0: rows 0-2 with ty = 0, row 3
    // with ty = 1), then times the channel's gamma * invstd
    auto pre_rows = [&](float4 &p0, float4 &p1, float4 &p2, float4 y0, float4 m0, int round_) {
        float4 own = p0;  // row q of the plane
        if constexpr (PRE == 3) {
            own = train_bn_apply4(own, wtab + 3 * ((round_ < rounds ? round_ : rounds - 1) * WN_CK + ci_x));
        } else {
            if constexpr (PRE == 2) own = sum4(own, y0);
            own = mask4(own, m0);
            if (pre_store && round_ < rounds) *reinterpret_cast<float4 *>((pre.store + round_ * xstep) + xq) = own;
            own = scale4(wtab[(round_ < rounds ? round_ : rounds - 1) * WN_CK + ci_x], own);
        }
        // quad rotations: rot k = the row held by quad lane (q + k) & 3
        const float4 r1 = quad_rot<0x39>(own), r2 = quad_rot<0x4E>(own), r3 = quad_rot<0x93>(own);
        // rows ty .. ty + 2:  q = 0: (own, r1, r2)   q = 1: (r3, own, r1)   q = 2: (r3, own, r1)   q = 3: (r2, r3, own)
        const bool q0 = (t_x & 3) == 0, q3 = (t_x & 3) == 3;
        p0 = q0 ? own : (q3 ? r2 : r3);
        p1 = q0 ? r1 : (q3 ? r3 : own);
        p2 = q0 ? r2 : (q3 ? own : r1);
    };
    // V = B^T d B of this lane's patch from the three rows p0..p2 (rows ty .. ty+2 of the plane) -> buffer `buf`.  Patch rows: ty = 0:
    // (pad, row 0, 1, 2); ty = 1: (row 1, 2, 3, pad); columns likewise with tx
    auto transform = [&](float *buf, float4 p0, float4 p1, float4 p2) {
        const float q0[4] = {p0.x, p0.y, p0.z, p0.w}, q1[4] = {p1.x, p1.y, p1.z, p1.w}, q2[4] = {p2.x, p2.y, p2.z, p2.w};
        float rw[4][4], dd[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            rw[0][c] = ty ? q0[c] : 0.0f;
            rw[1][c] = ty ? q1[c] : q0[c];
            rw[2][c] = ty ? q2[c] : q1[c];
            rw[3][c] = ty ? 0.0f : q2[c];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dd[i][0] = tx ? rw[i][1] : 0.0f;
            dd[i][1] = tx ? rw[i][2] : rw[i][0];
            dd[i][2] = tx ? rw[i][3] : rw[i][1];
            dd[i][3] = tx ? 0.0f : rw[i][2];
        }
        float tt[4][4];  // B^T d
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tt[0][j] = dd[0][j] - dd[2][j];
            tt[1][j] = dd[1][j] + dd[2][j];
            tt[2][j] = dd[2][j] - dd[1][j];
            tt[3][j] = dd[1][j] - dd[3][j];
        }
        float *vp = buf + 16 * WN_US + ci_x * 16 + t_x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vp[(4 * i + 0) * W4_VS] = tt[i][0] - tt[i][2];
            vp[(4 * i + 1) * W4_VS] = tt[i][1] + tt[i][2];
            vp[(4 * i + 2) * W4_VS] = tt[i][2] - tt[i][1];
            vp[(4 * i + 3) * W4_VS] = tt[i][1] - tt[i][3];
        }
    };
    // U slice and V of a fetched round -> buffer `buf_`
#define W4_FETCH_ANY(S, round_)                                                                    \
    do {                                                                                           \
        if constexpr (PRE != 0) W4_FETCH_ROW(S, round_);                                           \
        else W4_FETCH(S, round_);                                                                  \
    } while (0)
#define W4_STAGE(buf_, S, round_)                                                                  \
    do {                                                                                           \
        float *ud_ = (buf_) + xi_s * WN_US + ci_s * WN_CP + 4 * uq; /* row urow0 + 32 j -> xi = xi_s + 2 j */ \
        *reinterpret_cast<float4 *>(ud_) = S##u0;                                                  \
        *reinterpret_cast<float4 *>(ud_ + 2 * WN_US) = S##u1;                                      \
        *reinterpret_cast<float4 *>(ud_ + 4 * WN_US) = S##u2;                                      \
        *reinterpret_cast<float4 *>(ud_ + 6 * WN_US) = S##u3;                                      \
        *reinterpret_cast<float4 *>(ud_ + 8 * WN_US) = S##u4;                                      \
        *reinterpret_cast<float4 *>(ud_ + 10 * WN_US) = S##u5;                                     \
        *reinterpret_cast<float4 *>(ud_ + 12 * WN_US) = S##u6;                                     \
        *reinterpret_cast<float4 *>(ud_ + 14 * WN_US) = S##u7;                                     \
        if constexpr (PRE != 0) pre_rows(S##x0, S##x1, S##x2, S##y0, S##m0, round_);               \
        transform(buf_, S##x0, S##x1, S##x2);                                                      \
    } while (0)
    f32x4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[a][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // consumer: M_xi += U_xi V_xi for this wavefront's four xi, all 48 operands of the round first, then the 32 products
    auto multiply = [&](const float *cur) {
        float a0[4][4], a1[4][4], bv[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int xi = 4 * wave + a;
            const float *up = cur + xi * WN_US + lq * WN_CP + l15;
            const float *vp = cur + 16 * WN_US + xi * W4_VS + lq * 16 + l15;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) a0[a][kq] = up[kq * 4 * WN_CP], a1[a][kq] = up[kq * 4 * WN_CP + 16], bv[a][kq] = vp[kq * 4 * 16];
        }
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[a][kq], bv[a][kq], acc[a][0], 0, 0, 0);
                acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[a][kq], bv[a][kq], acc[a][1], 0, 0, 0);
            }
    };
    // producers: set B holds round r + 1 when iteration r starts (r even), set A round r + 2; the set just staged is refilled with round r + 3
    if (producer) {
        W4_FETCH_ANY(A, 0);
        W4_STAGE(lds, A, 0);
        W4_FETCH_ANY(B, 1);
        W4_FETCH_ANY(A, 2);
    }
    __syncthreads();
    for (int round = 0; round < rounds; round += 2) {
        // iteration `round` (even): current buffer 0, next buffer 1
        if (producer) {
            W4_STAGE(lds + W4_BUF, B, round + 1);  // round + 1 (past the end: the last slice again, nobody reads it)
            W4_FETCH_ANY(B, round + 3);
        } else {
            multiply(lds);
        }
        __syncthreads();
        if (round + 1 < rounds) {  // iteration round + 1: current buffer 1, next buffer 0
            if (producer) {
                W4_STAGE(lds, A, round + 2);  // round + 2
                W4_FETCH_ANY(A, round + 4);
            } else {
                multiply(lds + W4_BUF);
            }
            __syncthreads();
        }
    }
#undef W4_FETCH
#undef W4_FETCH_ROW
#undef W4_FETCH_ANY
#undef W4_STAGE
    // ---- output transform Y = A^T M A.  D[row = 4 lq + reg][col = l15] -> ms[xi][co][tile]; one (co, tile) per lane -----------------------
    float *ms = lds;
    if (!producer) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) ms[((4 * wave + a) * WN_CO + 16 * m + 4 * lq + r) * 16 + l15] = acc[a][m][r];
    }
    __syncthreads();
    {
        const int idx = threadIdx.x;  // (co, tile)
        const int co = idx >> 4, tl = idx & 15, img = tl >> 2, oy = (tl >> 1) & 1, ox = tl & 1;
        float mm[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mm[i][j] = ms[((4 * i + j) * WN_CO + co) * 16 + tl];
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t0[j] = (mm[0][j] + mm[1][j]) + mm[2][j];
            t1[j] = (mm[1][j] - mm[2][j]) - mm[3][j];
        }
        if (b + img < d.B) {
            const size_t oo = ((static_cast<size_t>(b + img) * d.RC + co0 + co) * 4 + 2 * oy) * 4 + 2 * ox;
            float2 r0 = make_float2((t0[0] + t0[1]) + t0[2], (t0[1] - t0[2]) - t0[3]), r1 = make_float2((t1[0] + t1[1]) + t1[2], (t1[1] - t1[2]) - t1[3]);
            if constexpr (POST) {
                if (post.mean) {
                    const PostConst k = post_const(post, co0 + co);
                    r0.x = post_apply(r0.x, k), r0.y = post_apply(r0.y, k), r1.x = post_apply(r1.x, k), r1.y = post_apply(r1.y, k);
                }
                if (post.res) {
                    const float2 q0 = *reinterpret_cast<const float2 *>(post.res + oo), q1 = *reinterpret_cast<const float2 *>(post.res + oo + 4);
                    r0.x += q0.x, r0.y += q0.y, r1.x += q1.x, r1.y += q1.y;
                }
                if (post.relu) r0.x = relu_keep_nan(r0.x), r0.y = relu_keep_nan(r0.y), r1.x = relu_keep_nan(r1.x), r1.y = relu_keep_nan(r1.y);
            }
            *reinterpret_cast<float2 *>(y + oo) = r0;
            *reinterpret_cast<float2 *>(y + oo + 4) = r1;
        }
        if constexpr (STATS) {  // the four lanes (tiles) of an image hold its 4x4 plane of channel co
            const float vals[4] = {(t0[0] + t0[1]) + t0[2], (t0[1] - t0[2]) - t0[3], (t1[0] + t1[1]) + t1[2], (t1[1] - t1[2]) - t1[3]};
            const float2 m = group_moments<4, 4>(vals);
            if ((tl & 3) == 0 && b + img < d.B) *reinterpret_cast<float2 *>(tb.stats_out + (static_cast<size_t>(co0 + co) * d.B + b + img) * 2) = m;
        }
    }
}

// ---- 8x8 and 16x16 maps, second version: the same producer / consumer split --------------------------------------------------------------
// The generic kernel above runs these maps with one wavefront per SIMD too (16x16: 151 KB of LDS per workgroup; 8x8: two workgroups per CU
// on little more than half of the CUs): 24.5 us = 10.7 of products + 3.8 transform + 0.9 staging + 9.6 launch / prologue / epilogue on 16x16,
// added up, not overlapped.  Here: 512 lanes, rounds of EIGHT reduction channels so that two buffers of {U slice, V, pixel frames} fit
// (8x8: two images per workgroup, 111 KB; 16x16: one image, 151 KB), wavefronts 0-3 multiply the current buffer while wavefronts 4-7, per
// round r: write U of round r+1 (loaded two rounds earlier into one of two named register sets), transform the pixel frames of round r+1
// (written during round r-1) into V, write the pixels of round r+2.  One barrier per round.  Same arithmetic in the same order as the
// generic kernel.
template <int MAP>
struct PcGeo {
    static constexpr int TX = MAP / 2, TI = TX * TX, IMG = MAP == 8 ? 2 : 1, T = TI * IMG, NB = T / 16;
    static constexpr int CK = 8;
    static constexpr int US = CK * WN_CP;                             // one xi's U [8 ci][48]
    static constexpr int VT = T + 16, VS = CK * VT;                   // V row stride (the four k of a wavefront on disjoint banks), one xi's V
    // pixel frames (zero ring).  Row stride and image stride are chosen so that a wavefront's patch reads - one ds_read_b64 per half row,
    // lanes = tiles - fall on 64 distinct banks: 16x16: rows of 24 floats; 8x8: rows of 12, the second image 160 floats behind the first
    static constexpr int XW = MAP == 16 ? 24 : 12, XI = MAP == 16 ? (MAP + 2) * XW : 160, XP = IMG * XI, XS = CK * XP;
    static_assert(XI >= (MAP + 2) * XW && XW >= MAP + 2 && XW % 2 == 0, "frame fits");
    static constexpr int BUF = 16 * US + 16 * VS + XS;
    // producing wavefronts: the 16x16 kernel is bound by its producers (clock stamps, scripts/wino_timing.py: 3056 cycles of producer work per
    // round against 1770 of the multiplying side with four of them) -> eight there, four on 8x8 (one patch per lane already)
    static constexpr int PW = MAP == 16 ? 8 : 4, PL = PW * 64, NT = (8 + PW) * 64, WPE = NT / 256;
    static constexpr int UPL = 1024 / PL;                             // float4 of the U slice per producer lane and round: 4 or 2
    static constexpr int XF4 = CK * IMG * MAP * MAP / 4 / PL;         // float4 of pixels per producer lane and round: 1
    static constexpr int PPT = CK * T / PL;                           // patches per producer lane and round: 1
    static constexpr size_t lds_bytes = 2 * BUF * sizeof(float);
};

// EIGHT multiplying wavefronts (two xi each) and four (8x8) or eight (16x16) producing ones: three or four wavefronts per SIMD (round 3;
// round 2: four multiplying wavefronts with four xi each + four producing ones, two per SIMD).  Shader-clock stamps inside the round-2
// kernel (scripts/wino_timing.py) showed the rounds bound by the PRODUCERS - 3056 cycles of producer work per 8-channel round on the 16x16
// layer (2314 of them the two input-patch transforms of a lane, i.e. LDS latency), 1770 on the multiplying side, whose wavefronts spent
// half of the loop waiting at the round barrier.  Twice the producing lanes halve a lane's share (one patch, two U rows, one pixel float4
// per round); the 64 accumulator registers per multiplying wavefront (instead of 128) make room for the extra wavefronts.  Same products
// in the same order per accumulator: bit-identical results.
constexpr int PC_CW = 8, PC_XPW = 16 / PC_CW;

// PRE / POST: see wino3x3_map4_kernel.  The train-mode pair (round 4; ee_fuse.hpp: TrainBn - BatchNorm with BATCH statistics exchanged
// across the kernel boundary instead of inside a BatchNorm launch): STATS - the output transform also writes, per (channel, image), the
// plane's (mean, M2) to tb.stats_out [RC][B][2]; PRE 3 - the prologue merges the partials of every reduction channel in a fixed order and
// the staged input is relu((x - mean) * invstd * gamma + beta).  The backward pair (input gradient): STATS 2 - this launch is the backward-data
// of the layer BEHIND the BatchNorm: next to dy it writes per (channel, image) (sum dz, sum dz * xhat), reading the BatchNorm's input
// tb.part_x and its saved statistics; PRE 4 - the prologue merges those sums, the staged input is the BatchNorm's dx formed from dy
// (x) and the BatchNorm's input (pre.mask).
template <int MAP, int PRE, bool POST, int STATS = 0>
__global__ __launch_bounds__(PcGeo<MAP>::NT) __attribute__((amdgpu_waves_per_eu(PcGeo<MAP>::WPE, PcGeo<MAP>::WPE))) void wino3x3_pc_kernel(const float *__restrict__ x, const float *__restrict__ u,
                                                                                                     float *__restrict__ y, WinoDims d, FusePre pre, FusePost post,
                                                                                                     TrainBn tb = TrainBn{}) {
    using G = PcGeo<MAP>;
    static_assert(G::XF4 == 1 && G::PPT == 1 && (G::UPL == 2 || G::UPL == 4), "8x8 or 16x16 maps");
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool producer = wave >= PC_CW;
    const int pt = (threadIdx.x - PC_CW * 64) & (G::PL - 1);  // producers: lane number inside their group of wavefronts
    WT_DECL;
    WT_STAMP(0);
    int bx, by;  // XCD-aware numbering (see the 4x4 kernel)
    if (!xcd_decode(blockIdx.x, d.RC / WN_CO, (d.B + G::IMG - 1) / G::IMG, d.wl, bx, by)) return;
    const int b = bx * G::IMG, co0 = by * WN_CO;
    const int l15 = lane & 15, lq = lane >> 4;
    const int rounds = d.KC / G::CK;
    // U slice of a round: 16 xi x 8 ci rows of 32 floats = 1024 float4, UPL per producer lane: rows (xi0 + PW j, ci_u)
    const int uq = pt & 7, ci_u = (pt >> 3) & 7, xi0 = pt >> 6;
    const float *ubase = u + (static_cast<size_t>(xi0) * d.KC + ci_u) * d.RC + co0 + 4 * uq;
    const size_t uxi4 = G::PW * static_cast<size_t>(d.KC) * d.RC, ustep = static_cast<size_t>(G::CK) * d.RC;
    // pixels of a round: 16x16: the 8 planes are contiguous, float4 number pt; 8x8: image pt >> 7, float4 pt & 127 of its 8 planes
    const int ximg = MAP == 8 ? pt >> 7 : 0, xbi = b + ximg < d.B ? b + ximg : d.B - 1;  // past the batch: a valid image, never stored
    // (element offsets fit 31 bits: wino_check) a 32-bit lane offset under a per-round uniform base = the scalar-base form of the global load
    const unsigned xo = static_cast<unsigned>(xbi) * static_cast<unsigned>(d.KC * (MAP * MAP)) + 4u * (MAP == 8 ? pt & 127 : pt);
    const float *xsrc = x + xo;
    const size_t xstep = static_cast<size_t>(G::CK) * (MAP * MAP);
    float4 UA0, UA1, UA2, UA3, UB0, UB1, UB2, UB3, XA0, XB0;  // NAMED register sets (see the 4x4 kernel)
    UA2 = UA3 = UB2 = UB3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 XA1, XB1, XA2, XB2;  // PRE: the second piece (1) and the mask (2) of the same pixels
    XA1 = XB1 = XA2 = XB2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float *wtab = lds + 2 * G::BUF;  // PRE: gamma * invstd of every reduction channel (filled before the prologue's first barrier)
    const int ci_p = MAP == 8 ? (pt & 127) >> 4 : pt >> 6;  // this producer lane's channel inside a round
    // 16x16 with PRE: the plain kernel sits at the 128 registers four wavefronts per SIMD leave; three pixel streams in two prefetch sets
    // spilled (54 registers).  There pixels and filters travel ONE round ahead in one set (a round is ~1.3 us: enough for an L2 hit) instead of two
    constexpr bool ONE_X = MAP == 16 && (PRE == 1 || PRE == 2 || PRE == 4);
    const bool pre_store = (PRE == 1 || PRE == 2) && pre.store != nullptr && by == 0 && b + ximg < d.B;
#define PC_FETCH_U(S, round_)                                                                      \
    do {                                                                                           \
        const int r_ = (round_) < rounds ? (round_) : rounds - 1; /* always issued */              \
        const float *up_ = ubase + r_ * ustep;                                                     \
        S##0 = *reinterpret_cast<const float4 *>(up_);                                             \
        S##1 = *reinterpret_cast<const float4 *>(up_ + uxi4);                                      \
        if (G::UPL > 2) {                                                                          \
            S##2 = *reinterpret_cast<const float4 *>(up_ + 2 * uxi4);                              \
            S##3 = *reinterpret_cast<const float4 *>(up_ + 3 * uxi4);                              \
        }                                                                                          \
    } while (0)
#define PC_FETCH_X(S, round_)                                                                      \
    do {                                                                                           \
        const int r_ = (round_) < rounds ? (round_) : rounds - 1;                                  \
        const float *xp_ = xsrc + r_ * xstep;                                                      \
        S##0 = *reinterpret_cast<const float4 *>(xp_);                                             \
        if constexpr (PRE == 1 || PRE == 2 || PRE == 4) S##2 = *reinterpret_cast<const float4 *>((pre.mask + r_ * xstep) + xo); \
        if constexpr (PRE == 2) S##1 = *reinterpret_cast<const float4 *>((pre.add + r_ * xstep) + xo);  \
    } while (0)
#define PC_STORE_U(buf_, S)                                                                        \
    do {                                                                                           \
        float *ud_ = (buf_) + xi0 * G::US + ci_u * WN_CP + 4 * uq;                                 \
        *reinterpret_cast<float4 *>(ud_) = S##0;                                                   \
        *reinterpret_cast<float4 *>(ud_ + G::PW * G::US) = S##1;                                   \
        if (G::UPL > 2) {                                                                          \
            *reinterpret_cast<float4 *>(ud_ + 2 * G::PW * G::US) = S##2;                           \
            *reinterpret_cast<float4 *>(ud_ + 3 * G::PW * G::US) = S##3;                           \
        }                                                                                          \
    } while (0)
    auto put_x = [&](float *xs, float4 v) {  // this lane's float4 -> frame interior
        int ci, img, row, c4;
        if (MAP == 8) {
            const int w = pt & 127;
            img = pt >> 7, ci = w >> 4, row = (w & 15) >> 1, c4 = 4 * (w & 1);
        } else {
            const int f = pt, q = f & 63;
            img = 0, ci = f >> 6, row = q >> 2, c4 = 4 * (q & 3);
        }
        float *dst = xs + ci * G::XP + img * G::XI + (1 + row) * G::XW + 1 + c4;
        dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
    };
    // PRE: dz = mask > 0 ? (x + add) : 0, written out by the by = 0 workgroup of the image, then times the channel's gamma * invstd
    auto pre_px = [&](float4 v, float4 a, float4 m, int round_) {
        if constexpr (PRE == 3)  // train-mode BatchNorm + ReLU of the producing layer, statistics from the table the prologue merged
            return train_bn_apply4(v, wtab + 3 * ((round_ < rounds ? round_ : rounds - 1) * G::CK + ci_p));
        if constexpr (PRE == 4)  // train-mode BatchNorm + ReLU BACKWARD: v = dy, m = the BatchNorm's input
            return train_bn_bwd_apply4(v, m, wtab + 7 * ((round_ < rounds ? round_ : rounds - 1) * G::CK + ci_p));
        if constexpr (PRE == 2) v = sum4(v, a);
        v = mask4(v, m);
        if (pre_store && round_ < rounds) *reinterpret_cast<float4 *>((pre.store + round_ * xstep) + xo) = v;
        return scale4(wtab[(round_ < rounds ? round_ : rounds - 1) * G::CK + ci_p], v);
    };
#define PC_PUT_X(buf_, S, round_)                                                                  \
    do {                                                                                           \
        if constexpr (PRE != 0) S##0 = pre_px(S##0, S##1, S##2, round_);                           \
        put_x((buf_) + 16 * G::US + 16 * G::VS, S##0);                                             \
    } while (0)
    // V = B^T d B of this lane's patches, from the frames of `buf` into its V
    auto transform = [&](float *buf) {
        const float *xs = buf + 16 * G::US + 16 * G::VS;
        float *vs = buf + 16 * G::US;
        {
            const int pr = pt, ci = pr / G::T, t = pr - ci * G::T, img = t / G::TI, ti = t - img * G::TI;
            const int ty = ti / G::TX, tx = ti - ty * G::TX;
            const float *p = xs + ci * G::XP + img * G::XI + (2 * ty) * G::XW + 2 * tx;  // patch rows 2ty-1 .. 2ty+2 = frame rows 2ty .. 2ty+3
            float dd[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // an even column offset: two 8-byte reads per row (bank-conflict-free, see PcGeo)
                const float2 lo = *reinterpret_cast<const float2 *>(p + i * G::XW), hi = *reinterpret_cast<const float2 *>(p + i * G::XW + 2);
                dd[i][0] = lo.x, dd[i][1] = lo.y, dd[i][2] = hi.x, dd[i][3] = hi.y;
            }
            float tt[4][4];  // B^T d
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tt[0][j] = dd[0][j] - dd[2][j];
                tt[1][j] = dd[1][j] + dd[2][j];
                tt[2][j] = dd[2][j] - dd[1][j];
                tt[3][j] = dd[1][j] - dd[3][j];
            }
            // V as [8 xi pairs][8 ci][VT tiles][2]: a multiplying wavefront owns the pair (2 w, 2 w + 1) and reads both values of a (ci, tile)
            // with one ds_read_b64; eight 8-byte stores here instead of sixteen 4-byte ones
            float2 *vp = reinterpret_cast<float2 *>(vs) + ci * G::VT + t;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vp[(2 * i + 0) * G::VS] = make_float2(tt[i][0] - tt[i][2], tt[i][1] + tt[i][2]);
                vp[(2 * i + 1) * G::VS] = make_float2(tt[i][2] - tt[i][1], tt[i][1] - tt[i][3]);
            }
        }
    };
    f32x4 acc[PC_XPW][G::NB][2];
#pragma unroll
    for (int a = 0; a < PC_XPW; ++a)
#pragma unroll
        for (int nb = 0; nb < G::NB; ++nb)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[a][nb][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // consumer: M_xi += U_xi V_xi for this wavefront's four xi; a round's operands first, then its products
    auto multiply = [&](const float *cur) {
        float a0[PC_XPW][2], a1[PC_XPW][2], bv[PC_XPW][2][G::NB];
        static_assert(PC_XPW == 2, "one xi pair per multiplying wavefront");
        const float2 *vp = reinterpret_cast<const float2 *>(cur + 16 * G::US) + (wave * G::CK + lq) * G::VT + l15;  // V[pair = wave][ci = 4 kq + lq][tile]
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
#pragma unroll
            for (int nb = 0; nb < G::NB; ++nb) {
                const float2 v2 = vp[kq * 4 * G::VT + 16 * nb];
                bv[0][kq][nb] = v2.x, bv[1][kq][nb] = v2.y;
            }
#pragma unroll
        for (int a = 0; a < PC_XPW; ++a) {
            const int xi = PC_XPW * wave + a;
            const float *up = cur + xi * G::US + lq * WN_CP + l15;
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) a0[a][kq] = up[kq * 4 * WN_CP], a1[a][kq] = up[kq * 4 * WN_CP + 16];
        }
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
#pragma unroll
            for (int a = 0; a < PC_XPW; ++a)
#pragma unroll
                for (int nb = 0; nb < G::NB; ++nb) {
                    acc[a][nb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[a][kq], bv[a][kq][nb], acc[a][nb][0], 0, 0, 0);
                    acc[a][nb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[a][kq], bv[a][kq][nb], acc[a][nb][1], 0, 0, 0);
                }
    };
    float *buf0 = lds, *buf1 = lds + G::BUF;
    // prologue: pixels of rounds 0 and 1 into their frames, U of round 0, V of round 0; sets: UA / XA hold even rounds, UB / XB odd ones
    if (producer) {  // the first loads are in flight while the rings are zeroed
        PC_FETCH_U(UA, 0);
        PC_FETCH_X(XA, 0);
        PC_FETCH_X(XB, 1);
    }
    for (int i = threadIdx.x; i < G::XS; i += G::NT) lds[16 * G::US + 16 * G::VS + i] = lds[G::BUF + 16 * G::US + 16 * G::VS + i] = 0.0f;  // zero rings
    if constexpr (PRE == 1 || PRE == 2)
        for (int c = threadIdx.x; c < d.KC; c += G::NT) wtab[c] = bn_scale(pre.var, pre.gamma, pre.eps, c);
    if constexpr (PRE == 3) train_bn_merge<G::NT>(tb, d.KC, wtab, blockIdx.x == 0);
    if constexpr (PRE == 4) train_bn_bwd_merge<G::NT>(tb, d.KC, wtab);
    __syncthreads();  // the zero rings
    if (producer) {
        PC_PUT_X(buf0, XA, 0);
        PC_PUT_X(buf1, XB, 1);
        PC_STORE_U(buf0, UA);
        if constexpr (ONE_X) {
            PC_FETCH_U(UA, 1);
        } else {
            PC_FETCH_U(UB, 1);
            PC_FETCH_U(UA, 2);
        }
        PC_FETCH_X(XA, 2);
        if constexpr (!ONE_X) PC_FETCH_X(XB, 3);
    }
    __syncthreads();
    if (producer) transform(buf0);
    __syncthreads();
    WT_STAMP(1);
    for (int round = 0; round < rounds; round += 2) {
        // iteration `round` (even): multiply buffer 0; U and V of round + 1 -> buffer 1, pixels of round + 2 -> frames of buffer 0
        WT_BEGIN;
        if (producer) {
            WT_P0;
            if constexpr (ONE_X) PC_STORE_U(buf1, UA);
            else PC_STORE_U(buf1, UB);
            WT_P(0);
            transform(buf1);
            WT_P(1);
            PC_PUT_X(buf0, XA, round + 2);
            if constexpr (ONE_X) {
                PC_FETCH_U(UA, round + 2);
                PC_FETCH_X(XA, round + 3);
            } else {
                PC_FETCH_U(UB, round + 3);
                PC_FETCH_X(XA, round + 4);
            }
            WT_P(2);
        } else {
            multiply(buf0);
        }
        WT_END;
        __syncthreads();
        if (round + 1 < rounds) {  // iteration round + 1: multiply buffer 1; U and V of round + 2 -> buffer 0, pixels of round + 3 -> frames of buffer 1
            WT_BEGIN;
            if (producer) {
                PC_STORE_U(buf0, UA);
                transform(buf0);
                if constexpr (ONE_X) {
                    PC_PUT_X(buf1, XA, round + 3);
                    PC_FETCH_U(UA, round + 3);
                    PC_FETCH_X(XA, round + 4);
                } else {
                    PC_PUT_X(buf1, XB, round + 3);
                    PC_FETCH_U(UA, round + 4);
                    PC_FETCH_X(XB, round + 5);
                }
            } else {
                multiply(buf1);
            }
            WT_END;
            __syncthreads();
        }
    }
    WT_STAMP(2);
#undef PC_FETCH_U
#undef PC_FETCH_X
#undef PC_STORE_U
#undef PC_PUT_X
    // ---- output transform Y = A^T M A.  D[row = 4 lq + reg][col = l15] -> ms[nb][xi][co][tile]: the accumulators of ALL N blocks go to
    // LDS at once (NB x 32 KB <= the two round buffers), ONE barrier, then every lane transforms its NB (co, tile) items back to back -
    // 16 LDS reads and two 8-byte stores each, all independent (round 2 ran the blocks one at a time through one 32 KB area: two barriers
    // and a serialised read -> store chain per block)
    float *ms = lds;
    static_assert(static_cast<size_t>(G::NB) * 16 * WN_CO * 16 * sizeof(float) <= G::lds_bytes, "the accumulators of all N blocks fit the round buffers");
    if (!producer) {
#pragma unroll
        for (int nb = 0; nb < G::NB; ++nb)
#pragma unroll
            for (int a = 0; a < PC_XPW; ++a)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ms[((nb * 16 + PC_XPW * wave + a) * WN_CO + 16 * m + 4 * lq + r) * 16 + l15] = acc[a][nb][m][r];
    }
    // POST: the lane's channel constants and its residual values are fetched while the accumulators settle in LDS
    PostConst pk{0.0f, 1.0f, 0.0f};
    float2 q0[G::NB], q1[G::NB];
    float bs_mean = 0.0f, bs_inv = 0.0f, bs_a = 0.0f, bs_b = 0.0f;
    if constexpr (STATS == 2) {  // the BatchNorm's input at this lane's outputs, and its channel's saved statistics
        if (threadIdx.x < 512) {
            const int co = threadIdx.x >> 4, tl = threadIdx.x & 15, c = co0 + co;
            bs_mean = tb.save_mean[c], bs_inv = tb.save_invstd[c], bs_a = bs_inv * (tb.gamma ? tb.gamma[c] : 1.0f), bs_b = tb.beta ? tb.beta[c] : 0.0f;
#pragma unroll
            for (int nb = 0; nb < G::NB; ++nb) {
                const int t = 16 * nb + tl, img = t / G::TI, ti = t - img * G::TI, ty = ti / G::TX, tx = ti - ty * G::TX;
                const int bi = b + img < d.B ? b + img : d.B - 1;
                const float *rp = tb.part + ((static_cast<size_t>(bi) * d.RC + c) * MAP + 2 * ty) * MAP + 2 * tx;
                q0[nb] = *reinterpret_cast<const float2 *>(rp), q1[nb] = *reinterpret_cast<const float2 *>(rp + MAP);
            }
        }
    }
    if constexpr (POST) {
        if (threadIdx.x < 512) {
            const int co = threadIdx.x >> 4, tl = threadIdx.x & 15;
            pk = post_const(post, co0 + co);
            if (post.res) {
#pragma unroll
                for (int nb = 0; nb < G::NB; ++nb) {
                    const int t = 16 * nb + tl, img = t / G::TI, ti = t - img * G::TI, ty = ti / G::TX, tx = ti - ty * G::TX;
                    const int bi = b + img < d.B ? b + img : d.B - 1;
                    const float *rp = post.res + ((static_cast<size_t>(bi) * d.RC + co0 + co) * MAP + 2 * ty) * MAP + 2 * tx;
                    q0[nb] = *reinterpret_cast<const float2 *>(rp), q1[nb] = *reinterpret_cast<const float2 *>(rp + MAP);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 512) {
        const int idx = threadIdx.x;  // (co, tile of a block)
        const int co = idx >> 4, tl = idx & 15;
        float sv[STATS ? G::NB : 1][4];
        float bsum1 = 0.0f, bsum2 = 0.0f;
#pragma unroll
        for (int nb = 0; nb < G::NB; ++nb) {
            const int t = 16 * nb + tl, img = t / G::TI, ti = t - img * G::TI, ty = ti / G::TX, tx = ti - ty * G::TX;
            float mm[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mm[i][j] = ms[((nb * 16 + 4 * i + j) * WN_CO + co) * 16 + tl];
            float t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t0[j] = (mm[0][j] + mm[1][j]) + mm[2][j];
                t1[j] = (mm[1][j] - mm[2][j]) - mm[3][j];
            }
            float2 r0 = make_float2((t0[0] + t0[1]) + t0[2], (t0[1] - t0[2]) - t0[3]), r1 = make_float2((t1[0] + t1[1]) + t1[2], (t1[1] - t1[2]) - t1[3]);
            if (b + img < d.B) {
                const size_t oo = ((static_cast<size_t>(b + img) * d.RC + co0 + co) * MAP + 2 * ty) * MAP + 2 * tx;
                if constexpr (POST) {
                    if (post.mean) r0.x = post_apply(r0.x, pk), r0.y = post_apply(r0.y, pk), r1.x = post_apply(r1.x, pk), r1.y = post_apply(r1.y, pk);
                    if (post.res) r0.x += q0[nb].x, r0.y += q0[nb].y, r1.x += q1[nb].x, r1.y += q1[nb].y;
                    if (post.relu) r0.x = relu_keep_nan(r0.x), r0.y = relu_keep_nan(r0.y), r1.x = relu_keep_nan(r1.x), r1.y = relu_keep_nan(r1.y);
                }
                *reinterpret_cast<float2 *>(y + oo) = r0;
                *reinterpret_cast<float2 *>(y + oo + MAP) = r1;
            }
            if constexpr (STATS == 1) sv[nb][0] = r0.x, sv[nb][1] = r0.y, sv[nb][2] = r1.x, sv[nb][3] = r1.y;  // (past the batch: never written out)
            if constexpr (STATS == 2) {
                const float dd[4] = {r0.x, r0.y, r1.x, r1.y}, xx[4] = {q0[nb].x, q0[nb].y, q1[nb].x, q1[nb].y};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dz = ((xx[e] - bs_mean) * bs_a + bs_b) > 0.0f ? dd[e] : 0.0f;
                    bsum1 += dz;
                    bsum2 += dz * ((xx[e] - bs_mean) * bs_inv);
                }
            }
        }
        if constexpr (STATS == 2) {
            static_assert(STATS != 2 || G::IMG == 1, "backward sums epilogue: one image per workgroup (16x16 maps)");
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) bsum1 += __shfl_xor(bsum1, off, 16), bsum2 += __shfl_xor(bsum2, off, 16);
            if (tl == 0 && b < d.B) *reinterpret_cast<float2 *>(tb.stats_out + (static_cast<size_t>(co0 + co) * d.B + b) * 2) = make_float2(bsum1, bsum2);
        }
        if constexpr (STATS == 1) {
            // the 16 lanes of a channel hold whole planes: N blocks [i * NBI, (i + 1) * NBI) are image i of the workgroup (MAP 16: one image
            // in four blocks; MAP 8: two images, one block each)
            constexpr int NBI = G::NB / G::IMG;
#pragma unroll
            for (int i = 0; i < G::IMG; ++i) {
                float vals[NBI * 4];
#pragma unroll
                for (int k = 0; k < NBI; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) vals[4 * k + e] = sv[i * NBI + k][e];
                const float2 m = group_moments<16, NBI * 4>(vals);
                if (tl == 0 && b + i < d.B) *reinterpret_cast<float2 *>(tb.stats_out + (static_cast<size_t>(co0 + co) * d.B + b + i) * 2) = m;
            }
        }
    }
    WT_STAMP(3);
    WT_DUMP;
}

constexpr int WN_MAX_KC = 512;  // PRE 1 / 2: the channel-scale table behind the two round buffers (2 KB; the 16x16 kernel has 5 KB to spare)
constexpr int WN_MAX_KC_TRAIN = 256;  // PRE 3: (mean, scale, shift) per reduction channel, 3 KB

template <int PRE>
constexpr size_t wino_table_bytes() {
    return PRE == 4 ? 7 * 128 * sizeof(float) : (PRE == 3 ? 3 * WN_MAX_KC_TRAIN * sizeof(float) : (PRE ? WN_MAX_KC * sizeof(float) : 0));
}

template <int MAP, int PRE, bool POST, int STATS = 0>
int wino_pc_launch(const float *x, const float *u, float *y, const WinoDims &d, const FusePre &pre, const FusePost &post, hipStream_t st,
                   const TrainBn &tb = TrainBn{}) {
    using G = PcGeo<MAP>;
    constexpr size_t bytes = G::lds_bytes + wino_table_bytes<PRE>();
    static_assert(bytes <= 160 * 1024, "fits the LDS of a CU");
    static int ok = hipFuncSetAttribute(reinterpret_cast<const void *>(wino3x3_pc_kernel<MAP, PRE, POST, STATS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(bytes)) == hipSuccess;
    if (!ok) return EE_ERR_UNSUPPORTED;
    EE_LAUNCH((wino3x3_pc_kernel<MAP, PRE, POST, STATS>), dim3(xcd_grid((d.B + G::IMG - 1) / G::IMG, d.RC / WN_CO, d.wl)), dim3(G::NT), bytes,
              st, x, u, y, d, pre, post, tb);
    return launch_status();
}

template <int PRE, bool POST, bool STATS = false>
int wino_map4_launch(const float *x, const float *u, float *y, const WinoDims &d, const FusePre &pre, const FusePost &post, hipStream_t st,
                     const TrainBn &tb = TrainBn{}) {
    constexpr size_t bytes = W4_LDS + wino_table_bytes<PRE>();
    static int ok = hipFuncSetAttribute(reinterpret_cast<const void *>(wino3x3_map4_kernel<PRE, POST, STATS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(bytes)) == hipSuccess;
    if (!ok) return EE_ERR_UNSUPPORTED;
    EE_LAUNCH((wino3x3_map4_kernel<PRE, POST, STATS>), dim3(xcd_grid((d.B + 3) / 4, d.RC / WN_CO, d.wl)), dim3(W4_NT), bytes, st, x, u, y, d, pre, post, tb);
    return launch_status();
}

template <int PRE, bool POST, bool STATS = false>
int wino_dispatch_t(const float *x, const float *u, float *y, const WinoDims &d, int H, const FusePre &pre, const FusePost &post, hipStream_t st, const TrainBn &tb) {
    if (H == 4) return wino_map4_launch<PRE, POST, STATS>(x, u, y, d, pre, post, st, tb);
    if (H == 8) return wino_pc_launch<8, PRE, POST, STATS ? 1 : 0>(x, u, y, d, pre, post, st, tb);
    return wino_pc_launch<16, PRE, POST, STATS ? 1 : 0>(x, u, y, d, pre, post, st, tb);
}

template <int PRE, bool POST>
int wino_dispatch(const float *x, const float *u, float *y, const WinoDims &d, int H, const FusePre &pre, const FusePost &post, hipStream_t st) {
    if (H == 4) return wino_map4_launch<PRE, POST>(x, u, y, d, pre, post, st);
    if (H == 8) return wino_pc_launch<8, PRE, POST>(x, u, y, d, pre, post, st);
    return wino_pc_launch<16, PRE, POST>(x, u, y, d, pre, post, st);
}

int wino_check(const float *x, const float *u, const float *y, int B, int KC, int RC, int H) {
    if (B < 0 || KC < 1 || RC < 1) return EE_ERR_SHAPE;
    if (KC % WN_CK != 0 || RC % WN_CO != 0 || (H != 4 && H != 8 && H != 16)) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!x || !u || !y) return EE_ERR_NULL;
    if (!aligned16(x) || !aligned16(u) || !aligned16(y)) return EE_ERR_ALIGN;
    if (static_cast<int64_t>(B) * (KC > RC ? KC : RC) * H * H > 0x7fffffffLL) return EE_ERR_SHAPE;
    return EE_OK;
}

}  // namespace

// y = conv3x3(x) for H x H maps (H = 4, 8 or 16) with the filters given in the transform domain: u [16][KC][RC], u[4i+j][k][r] =
// (G g G^T)[i][j] of the (r, k) filter pair the product needs (forward: g = w[r][k]; backward-data: g = w[k][r] rotated by 180 degrees).
EE_API int ee_wino3x3_f32(const float *x, const float *u, float *y, int B, int KC, int RC, int H, void *stream) {
    const int rc = wino_check(x, u, y, B, KC, RC, H);
    if (rc != EE_OK || B == 0) return rc;
    const WinoDims d{B, KC, RC, xcd_weights_local(4.0 * B * KC * H * H, 64.0 * KC * RC, RC / WN_CO) ? 1 : 0};
    // the convolution's ALGORITHMIC flops (2 * 9 * KC * RC per output pixel); the kernel executes 4/9 of them (16 multiplies per 2x2 tile)
    ProfScope prof(EE_K_WINO, as_stream(stream), 2.0 * 9.0 * KC * RC * static_cast<double>(B) * H * H);
    return wino_dispatch<0, false>(x, u, y, d, H, FusePre{}, FusePost{}, as_stream(stream));
}

// The same convolution with an eval-mode BatchNorm (running statistics), the block's residual and the ReLU in its output transform:
//     y = [relu]( (conv3x3(x) - mean) * (gamma / sqrt(var + eps)) + beta [+ res] )          resnet.py:44-59 under model.eval()
// - what ee_wino3x3_f32 followed by ee_bn_act_fwd_f32(training = 0) computes, bit for bit, in one launch and without the round trip of the
// convolution's output through memory.  mean / var / gamma / beta: [Cout] (gamma, beta may be null: 1, 0); res: [B][Cout][H][H] or null.
EE_API int ee_wino3x3_bn_eval_fwd_f32(const float *x, const float *u, const float *mean, const float *var, const float *gamma, const float *beta,
                                      float eps, const float *res, int relu, float *y, int B, int Cin, int Cout, int H, void *stream) {
    const int rc = wino_check(x, u, y, B, Cin, Cout, H);
    if (rc != EE_OK || B == 0) return rc;
    const FusePost post{mean, var, gamma, beta, eps, res, relu};
    const int pc = check_post(post);
    if (pc != EE_OK) return pc;
    const WinoDims d{B, Cin, Cout, xcd_weights_local(4.0 * B * Cin * H * H, 64.0 * Cin * Cout, Cout / WN_CO) ? 1 : 0};
    ProfScope prof(EE_K_WINO_FUSED, as_stream(stream), 2.0 * 9.0 * Cin * Cout * static_cast<double>(B) * H * H);
    return wino_dispatch<0, true>(x, u, y, d, H, FusePre{}, post, as_stream(stream));
}

// ... and its backward with respect to x, given the gradient of y in one or two pieces (dy2 may be null) and y itself (the ReLU mask):
//     dz = (y > 0) * (dy + dy2),   dx = conv3x3^T( gamma / sqrt(var + eps) * dz ),   dres = dz (optional: the residual branch's gradient)
// - ee_bn_act_bwd2_f32(training = 0, relu = 1) followed by ee_wino3x3_f32 on the backward filter set `u_b` ([16][Cout][Cin]), in one launch.
// var / gamma: [Cout] of the BatchNorm behind the convolution.  dx_add (optional, [B][Cin][H][H]): added to dx in the output transform - the
// gradient that reaches the block's input through its identity branch, so that ONE summed gradient leaves the block (dx + dx_add in the
// order ee_bn.hip's backward adds the two pieces of a forked output).
EE_API int ee_wino3x3_bn_eval_bwd_f32(const float *dy, const float *dy2, const float *y, const float *u_b, const float *var, const float *gamma, float eps,
                                      float *dres, const float *dx_add, float *dx, int B, int Cin, int Cout, int H, void *stream) {
    const int rc = wino_check(dy, u_b, dx, B, Cout, Cin, H);
    if (rc != EE_OK || B == 0) return rc;
    if (!y) return EE_ERR_NULL;
    if (Cout > WN_MAX_KC) return EE_ERR_UNSUPPORTED;
    const FusePre pre{dy2, y, dres, var, gamma, eps};
    const int pc = check_pre(pre);
    if (pc != EE_OK) return pc;
    const WinoDims d{B, Cout, Cin, xcd_weights_local(4.0 * B * Cout * H * H, 64.0 * Cin * Cout, Cin / WN_CO) ? 1 : 0};
    ProfScope prof(EE_K_WINO_FUSED, as_stream(stream), 2.0 * 9.0 * Cin * Cout * static_cast<double>(B) * H * H);
    if (dx_add) {
        const FusePost post{nullptr, nullptr, nullptr, nullptr, 0.0f, dx_add, 0};
        if (!aligned16(dx_add)) return EE_ERR_ALIGN;
        if (dy2) return wino_dispatch<2, true>(dy, u_b, dx, d, H, pre, post, as_stream(stream));
        return wino_dispatch<1, true>(dy, u_b, dx, d, H, pre, post, as_stream(stream));
    }
    if (dy2) return wino_dispatch<2, false>(dy, u_b, dx, d, H, pre, FusePost{}, as_stream(stream));
    return wino_dispatch<1, false>(dy, u_b, dx, d, H, pre, FusePost{}, as_stream(stream));
}

// TRAIN-mode BatchNorm across the kernel boundary (round 4; ee_fuse.hpp: TrainBn; resnet.py:44-49: conv1 -> bn1 -> relu -> conv2):
//   ee_wino3x3_stats_f32        y = conv3x3(x), and per (channel, image) the plane's (mean, M2) -> stats [Cout][B][2]
//   ee_wino3x3_bn_train_pre_f32 y = conv3x3( relu( batch_norm(x) ) ): x = the RAW output of the producing convolution, its batch statistics
//                               merged from `stats` [Cin][S][2] (S partials of cnt values each); save_mean / save_invstd [Cin] are written
//                               and running_mean / running_var (may be NULL) moved with `momentum` as ee_bn_act_fwd_f32(training = 1) does.
EE_API int ee_wino3x3_stats_f32(const float *x, const float *u, float *y, float *stats, int B, int KC, int RC, int H, void *stream) {
    const int rc = wino_check(x, u, y, B, KC, RC, H);
    if (rc != EE_OK || B == 0) return rc;
    if (!stats) return EE_ERR_NULL;
    const WinoDims d{B, KC, RC, xcd_weights_local(4.0 * B * KC * H * H, 64.0 * KC * RC, RC / WN_CO) ? 1 : 0};
    ProfScope prof(EE_K_WINO_FUSED, as_stream(stream), 2.0 * 9.0 * KC * RC * static_cast<double>(B) * H * H);
    TrainBn tb{};
    tb.stats_out = stats;
    return wino_dispatch_t<0, false, true>(x, u, y, d, H, FusePre{}, FusePost{}, as_stream(stream), tb);
}

EE_API int ee_wino3x3_bn_train_pre_f32(const float *x, const float *stats, int S, int cnt, const float *gamma, const float *beta, float eps, float momentum,
                                       float *running_mean, float *running_var, float *save_mean, float *save_invstd, const float *u, float *y, int B,
                                       int KC, int RC, int H, void *stream) {
    const int rc = wino_check(x, u, y, B, KC, RC, H);
    if (rc != EE_OK || B == 0) return rc;
    if (KC > WN_MAX_KC_TRAIN) return EE_ERR_UNSUPPORTED;
    if (!stats || !save_mean || !save_invstd) return EE_ERR_NULL;
    if (S < 1 || cnt < 1 || (running_mean == nullptr) != (running_var == nullptr)) return EE_ERR_SHAPE;
    if (reinterpret_cast<uintptr_t>(stats) & 7u) return EE_ERR_ALIGN;
    const TrainBn tb{nullptr, stats, S, static_cast<float>(cnt), gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd};
    const WinoDims d{B, KC, RC, xcd_weights_local(4.0 * B * KC * H * H, 64.0 * KC * RC, RC / WN_CO) ? 1 : 0};
    ProfScope prof(EE_K_WINO_FUSED, as_stream(stream), 2.0 * 9.0 * KC * RC * static_cast<double>(B) * H * H);
    return wino_dispatch_t<3, false, false>(x, u, y, d, H, FusePre{}, FusePost{}, as_stream(stream), tb);
}

// The same exchange in the BACKWARD direction (input gradient only; 16x16 maps, channels <= 128):
//   ee_wino3x3_bwd_sums_f32         dy = conv3x3^T(dc) on the backward filter set (the backward-data of the layer BEHIND the BatchNorm), and per
//                                   (channel, image) sums [Cin][B][2] = (sum dz, sum dz * xhat), dz = (bn(x) > 0) * dy, from x = the BatchNorm's
//                                   input [B,Cin,H,H] and its saved statistics
//   ee_wino3x3_bn_train_bwd_pre_f32 dx = conv3x3^T( gamma * invstd * ((dz - mean(dz)) - xhat * mean(dz * xhat)) ): the BatchNorm's own backward
//                                   (ee_bn_act_bwd2_f32, training = 1, relu = 1, mask from x) folded into the staging of the layer in FRONT's
//                                   backward-data convolution; sums as written by ee_wino3x3_bwd_sums_f32
EE_API int ee_wino3x3_bwd_sums_f32(const float *dc, const float *u_b, const float *x, const float *save_mean, const float *save_invstd, const float *gamma,
                                   const float *beta, float *dy, float *sums, int B, int Cin, int Cout, int H, void *stream) {
    const int rc = wino_check(dc, u_b, dy, B, Cout, Cin, H);
    if (rc != EE_OK || B == 0) return rc;
    if (H != 16) return EE_ERR_UNSUPPORTED;
    if (!x || !save_mean || !save_invstd || !sums) return EE_ERR_NULL;
    if ((reinterpret_cast<uintptr_t>(x) & 7u) || (reinterpret_cast<uintptr_t>(sums) & 7u)) return EE_ERR_ALIGN;
    TrainBn tb{};
    tb.stats_out = sums, tb.part = x, tb.gamma = gamma, tb.beta = beta;
    tb.save_mean = const_cast<float *>(save_mean), tb.save_invstd = const_cast<float *>(save_invstd);  // read only by this launch
    const WinoDims d{B, Cout, Cin, xcd_weights_local(4.0 * B * Cout * H * H, 64.0 * Cin * Cout, Cin / WN_CO) ? 1 : 0};
    ProfScope prof(EE_K_WINO_FUSED, as_stream(stream), 2.0 * 9.0 * Cin * Cout * static_cast<double>(B) * H * H);
    return wino_pc_launch<16, 0, false, 2>(dc, u_b, dy, d, FusePre{}, FusePost{}, as_stream(stream), tb);
}

EE_API int ee_wino3x3_bn_train_bwd_pre_f32(const float *dy, const float *x, const float *sums, int S, int cnt, const float *save_mean, const float *save_invstd,
                                           const float *gamma, const float *beta, const float *u_b, float *dx, int B, int Cin, int Cout, int H, void *stream) {
    const int rc = wino_check(dy, u_b, dx, B, Cout, Cin, H);
    if (rc != EE_OK || B == 0) return rc;
    if (H != 16 || Cout > 128) return EE_ERR_UNSUPPORTED;
    if (!x || !sums || !save_mean || !save_invstd) return EE_ERR_NULL;
    if (S < 1 || cnt < 1) return EE_ERR_SHAPE;
    if (!aligned16(x) || (reinterpret_cast<uintptr_t>(sums) & 7u)) return EE_ERR_ALIGN;
    TrainBn tb{};
    tb.part = sums, tb.S = S, tb.cnt = static_cast<float>(cnt), tb.gamma = gamma, tb.beta = beta;
    tb.save_mean = const_cast<float *>(save_mean), tb.save_invstd = const_cast<float *>(save_invstd);
    const FusePre pre{nullptr, x, nullptr, nullptr, nullptr, 0.0f};
    const WinoDims d{B, Cout, Cin, xcd_weights_local(4.0 * B * Cout * H * H, 64.0 * Cin * Cout, Cin / WN_CO) ? 1 : 0};
    ProfScope prof(EE_K_WINO_FUSED, as_stream(stream), 2.0 * 9.0 * Cin * Cout * static_cast<double>(B) * H * H);
    return wino_pc_launch<16, 4, false, 0>(dy, u_b, dx, d, pre, FusePost{}, as_stream(stream), tb);
}

#ifdef EE_WINO_TIMING
EE_API int ee_wino_timing_sub(int on) { return static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(g_wino_sub), &on, sizeof(int))); }
EE_API int ee_wino_timing_read(unsigned long long *host, int n) {
    return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wino_stamps), sizeof(unsigned long long) * n));
}
#endif
