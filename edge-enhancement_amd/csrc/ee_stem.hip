// ee_stem.hip - the LAST launch of an attack iteration's backward pass through a ResNet: from the gradient of the stem's pooled activation
// straight to the gradient of the image (Tiny_ImageNet/models_tinyimagenet/resnet.py:112-117: conv1 -> bn1 -> relu -> maxpool, backwards).
//
// Rounds 2-3 ran it as bn_pool_bwd_kernel<true> (ee_bn.hip: gather the pooled gradient through the argmax codes, ReLU mask recomputed from
// the convolution output x, BatchNorm backward -> the [B,64,32,32] gradient dy, 26 MB written) and stem_bwd_data_mfma_kernel (ee_conv.hip: 26 MB
// read, a 4x4-window correlation on the dy grid producing the 2x2x3 image values of a cell on v_mfma_f32_16x16x4_f32): 16.4 + 45 us at batch 100.
// Here the first kernel's arithmetic happens while the second one's operand is staged:
//   * PRODUCER wavefronts (4) form the dy frame of the next round (16 channels x 5 rows x 36 columns) from x, the codes and the pooled gradient -
//     the same expressions in the same order as bn_pool_bwd_kernel (bit-identical values) - and stage the round's rearranged weights;
//   * CONSUMER wavefronts (4; one 16-cell M-tile each) multiply the current round out of LDS, exactly stem_bwd_data_mfma_kernel's products in its
//     order (channel-major, then the window row u): the image gradient is bit-identical to the two-launch sequence;
//   * PERSISTENT workgroups (2 per CU) walk over units of 2 dy rows x 32 columns x all channels (unit = blockIdx.x, += gridDim.x): 1600 units
//     at batch 100 spread 6-7 per CU, where 400 workgroups of 8 rows left 144 CUs with two workgroups and 112 with one (the kernel is bound by
//     the f32 matrix pipe: 1024 MFMAs per wavefront and 8-row tile).
// One barrier per round: while round g is multiplied the producers form round g + 1's frame (its pooled cells went to LDS a round earlier, its
// x values into the other of two register sets) and put round g + 2's pooled cells into LDS; every global load is issued a whole round before
// its use.  The round's rearranged weights are staged by the CONSUMERS (loads in front of their products, LDS stores behind them).  The two
// roles run separate loops that meet at the same barriers (in one loop the register allocator keeps both roles' state alive: spills).
//
// CNN-body glue, not a row of SURVEY.md section 8.
#include "ee_common.hpp"

namespace {

using namespace ee;

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef EE_STEMB_SKIP
#define EE_STEMB_SKIP 0  // probe builds only (scripts/stem_bwd_phases.py): 1 no products, 2 no frame arithmetic, 4 no global loads in the loop
#endif
constexpr int SP_NT = 512, SP_PT = 256;                    // lanes: 4 consuming + 4 producing wavefronts
constexpr int SP_K = 64, SP_KC = 16, SP_ROUNDS = SP_K / SP_KC;
constexpr int SP_TA = 2, SP_FH = SP_TA + 3, SP_FW = 36, SP_PLANE = SP_FH * SP_FW, SP_FR = SP_KC * SP_PLANE;  // frame of a round
constexpr int SP_OW = 32, SP_PW = 16, SP_PR = 4, SP_CELLS = SP_KC * SP_PR * SP_PW;                           // pooled rows a frame touches
constexpr int SP_CPT = SP_CELLS / SP_PT;                                                                       // pooled cells per producing lane: 4
constexpr int SP_WP = SP_KC * 256;                                                                             // [ch][u][v][j]
constexpr int SP_OUT = 3 * 2 * SP_TA * 2 * SP_OW;                                                              // image tile [3][4][64]
constexpr int SP_TAB = 6;                                                                                      // mean, a, b0, invstd, m1, m2
constexpr size_t SP_LDS = (2 * SP_FR + 2 * SP_WP + 4 * SP_CELLS + SP_K * SP_TAB + SP_OUT) * sizeof(float);

struct StemBwdArgs {
    const float *dyp, *dyp2;      // pooled gradient [B][64][PH][16] (second piece nullable)
    const uint8_t *code;          // argmax codes of the pool
    const float *x;               // the convolution's output [B][64][OH][32]
    const float *gamma, *beta, *mean, *invstd_or_var;
    const float *sums;            // training: [64][G][2] partial (sum dz, sum dz * xhat); else null
    const float *w;               // [64][3][7][7]
    float *dx;                    // [B][3][2 OH][64]
    float *dgamma, *dbeta;        // nullable (training)
    float eps;
    int training, G, B, OH, PH, units, tiles_a;
};

struct Unit {
    int n, t;  // image, row pair: dy rows 2 t, 2 t + 1
};

__global__ __launch_bounds__(SP_NT, 4) void stem_bn_pool_bwd_data_kernel(StemBwdArgs p) {
    extern __shared__ __align__(16) float lds[];
    float *fr = lds;                       // [2][16][5][36]
    float *wp = fr + 2 * SP_FR;            // [2][16][4][4][16]
    float2 *gpc = reinterpret_cast<float2 *>(wp + 2 * SP_WP);  // [2][16][4][16] {pooled gradient, argmax code}
    float *tab = wp + 2 * SP_WP + 4 * SP_CELLS;                // [64][6]
    float *out = tab + SP_K * SP_TAB;                          // [3][4][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int ptid = tid & (SP_PT - 1);
    const int i = lane & 15, kk = lane >> 4;
    const size_t plane_x = static_cast<size_t>(p.OH) * SP_OW, plane_p = static_cast<size_t>(p.PH) * SP_PW;
    // ---- per-channel constants: the expressions of bn_pool_bwd_kernel (its G partial sums per channel through LDS in one parallel load) ---
    if (p.sums) {
        for (int e = tid; e < SP_K * p.G * 2; e += SP_NT) fr[e] = p.sums[e];  // (G <= 64: fits the frame buffers, which are written later)
        __syncthreads();
    }
    if (tid < SP_K) {
        const int c = tid;
        const float mean = p.mean[c];
        const float invstd = p.training ? p.invstd_or_var[c] : 1.0f / sqrtf(p.invstd_or_var[c] + p.eps);
        const float a = invstd * (p.gamma ? p.gamma[c] : 1.0f), b0 = p.beta ? p.beta[c] : 0.0f;
        float m1 = 0.0f, m2 = 0.0f;
        if (p.sums) {
            float sdz = 0.0f, sdzx = 0.0f;
            for (int g = 0; g < p.G; ++g) {
                sdz += fr[(c * p.G + g) * 2];
                sdzx += fr[(c * p.G + g) * 2 + 1];
            }
            if (blockIdx.x == 0) {
                if (p.dgamma) p.dgamma[c] = sdzx;
                if (p.dbeta) p.dbeta[c] = sdz;
            }
            const float n = static_cast<float>(p.B) * static_cast<float>(p.OH * SP_OW);
            m1 = p.training ? sdz / n : 0.0f;
            m2 = p.training ? sdzx / n : 0.0f;
        }
        float *t = tab + c * SP_TAB;
        t[0] = mean, t[1] = a, t[2] = b0, t[3] = invstd, t[4] = m1, t[5] = m2;
    }
    if (p.sums) __syncthreads();  // the partial sums are read: the frame buffers are free
    const int stride = static_cast<int>(gridDim.x);
    const int mine = (p.units - static_cast<int>(blockIdx.x) + stride - 1) / stride;  // units of this workgroup (>= 1: grid <= units)
    const int steps = mine * SP_ROUNDS;
    auto unit_of = [&](int g) {
        const int u = static_cast<int>(blockIdx.x) + (g / SP_ROUNDS) * stride;
        return Unit{u / p.tiles_a, u % p.tiles_a};
    };
    // ---- producer state: one round of global loads in registers ---------------------------------------------------------------------------
    // frame: producing lane `pos` (< 180) owns frame position (frow, fc) of ALL 16 channels of a round - its candidate windows are the same for
    // every channel; pooled cells and weights are spread over all 256 producing lanes
    float XA[SP_KC], XB[SP_KC], PG[SP_CPT], PG2[SP_CPT];  // x of an even / odd round; the pooled cells of one round
    float W[SP_KC];  // CONSUMER lanes: the next round's weights (the producers' registers are full)
    int PC[SP_CPT];
    const int pos = ptid, frow = pos / SP_FW, fc = pos - frow * SP_FW;
    const bool owner = pos < SP_PLANE;
    // weights: lane t fills wp[k][u][v][j] for (u, v, j) = (t >> 6, (t >> 4) & 3, t & 15) and every k of the round (stem_bwd_data_mfma_kernel's roles)
    const int jc = i % 3, jq = i / 3, jph = jq >> 1, jpw = jq & 1;
    const int su = ptid >> 6, sv = (ptid >> 4) & 3;
    const bool wvalid = i < 12 && su < 3 + jph && sv < 3 + jpw;
    const int woff = jc * 49 + (5 + jph - 2 * su) * 7 + (5 + jpw - 2 * sv);
    // does this lane's frame position lie inside the map for the unit of round g?
    auto inside = [&](int g) {
        const Unit un = unit_of(g);
        const int r = 2 * un.t - 1 + frow, c = fc - 1;
        return owner && r >= 0 && r < p.OH && c >= 0 && c < SP_OW;
    };
    auto load_x = [&](int g, float (&X)[SP_KC]) {
        if ((EE_STEMB_SKIP & 4) && g > 1) return;
        const Unit un = unit_of(g);
        const float *xn = p.x + (static_cast<size_t>(un.n) * SP_K + (g % SP_ROUNDS) * SP_KC) * plane_x;  // (uniform)
        const unsigned xo = inside(g) ? static_cast<unsigned>((2 * un.t - 1 + frow) * SP_OW + fc - 1) : 0u;
#pragma unroll
        for (int k = 0; k < SP_KC; ++k) X[k] = xn[static_cast<unsigned>(k) * static_cast<unsigned>(plane_x) + xo];
    };
    auto load_p = [&](int g) {
        if ((EE_STEMB_SKIP & 4) && g > 1) return;
        const Unit un = unit_of(g);
        const size_t pn = (static_cast<size_t>(un.n) * SP_K + (g % SP_ROUNDS) * SP_KC) * plane_p;  // (uniform)
        const float *d1 = p.dyp + pn, *d2 = p.dyp2 ? p.dyp2 + pn : p.dyp + pn;
        const uint8_t *cdp = p.code + pn;
#pragma unroll
        for (int j = 0; j < SP_CPT; ++j) {
            const int q = ptid + j * SP_PT;
            const int ch = q >> 6, prl = (q >> 4) & 3, pc = q & 15;
            const int pr = un.t - 1 + prl;
            const bool in = pr >= 0 && pr < p.PH;
            const unsigned o = in ? static_cast<unsigned>(ch) * static_cast<unsigned>(plane_p) + static_cast<unsigned>(pr * SP_PW + pc) : 0u;
            PG[j] = d1[o];
            PG2[j] = d2[o];
            PC[j] = in ? static_cast<int>(cdp[o]) : 255;
        }
    };
    auto load_w = [&](int g) {
        const float *wk = p.w + static_cast<size_t>((g % SP_ROUNDS) * SP_KC) * 147;
#pragma unroll
        for (int k = 0; k < SP_KC; ++k) W[k] = wk[static_cast<unsigned>(k * 147 + (wvalid ? woff : 0))];
    };
    auto stage_w = [&](int g) {
        float *wb = wp + (g & 1) * SP_WP;
#pragma unroll
        for (int k = 0; k < SP_KC; ++k) wb[k * 256 + ptid] = wvalid ? W[k] : 0.0f;
    };
    // the pooled gradients / codes of round g go to LDS (one round before its frame is formed)
    auto stage_a = [&](int g) {
        float2 *gb = gpc + (g & 1) * SP_CELLS;
#pragma unroll
        for (int j = 0; j < SP_CPT; ++j) {
            const int q = ptid + j * SP_PT;
            gb[q] = make_float2(PC[j] == 255 ? 0.0f : (p.dyp2 ? PG[j] + PG2[j] : PG[j]), __int_as_float(PC[j]));
        }
    };
    // the frame of round g = the BatchNorm + ReLU + max-pool backward of its 16 x 5 x 36 positions (zero outside the map)
    auto stage_b = [&](int g, const float (&X)[SP_KC]) {
        if (!owner || ((EE_STEMB_SKIP & 2) && g > 1)) return;
        const Unit un = unit_of(g);
        const bool xin = inside(g);
        const float2 *gb = gpc + (g & 1) * SP_CELLS;
        const int kc = (g % SP_ROUNDS) * SP_KC;
        float *fb = fr + (g & 1) * SP_FR + pos;
        const int r = 2 * un.t - 1 + frow, c = fc - 1;
        const int oh0 = r >> 1, ow0 = c >> 1, l0 = oh0 - (un.t - 1);
        const bool row2 = (r & 1) && oh0 + 1 < p.PH, col2 = (c & 1) && ow0 + 1 < SP_PW;
        // the candidate windows in ATen's accumulation order (bn_pool_bwd_kernel): pooled row oh0 then oh0 + 1, column ow0 then ow0 + 1; a window
        // that does not exist reads the first one's cell and can never match (code -1).  All LDS reads of a lane first (as `if (code == want)
        // acc += g` chains they were four dependent LDS round trips per element: 6 us per round), then selects: acc + 0.0f == acc bit for bit
        const int w00 = ((r & 1) + 1) * 3 + (c & 1) + 1;
        const int b00 = xin ? l0 * SP_PW + ow0 : 0, b01 = col2 ? b00 + 1 : b00, b10 = row2 ? b00 + SP_PW : b00, b11 = (row2 && col2) ? b00 + SP_PW + 1 : b00;
        const int w01 = col2 ? w00 - 2 : -1, w10 = row2 ? w00 - 6 : -1, w11 = (row2 && col2) ? w00 - 8 : -1;
        constexpr int KB = 4;  // channels per batch of reads (32 registers)
#pragma unroll
        for (int k0 = 0; k0 < SP_KC; k0 += KB) {
            float2 c00[KB], c01[KB], c10[KB], c11[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const float2 *cell = gb + (k0 + k) * (SP_PR * SP_PW);
                c00[k] = cell[b00], c01[k] = cell[b01], c10[k] = cell[b10], c11[k] = cell[b11];
            }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                float acc = 0.0f;
                acc += __float_as_int(c00[k].y) == w00 ? c00[k].x : 0.0f;
                acc += __float_as_int(c01[k].y) == w01 ? c01[k].x : 0.0f;
                acc += __float_as_int(c10[k].y) == w10 ? c10[k].x : 0.0f;
                acc += __float_as_int(c11[k].y) == w11 ? c11[k].x : 0.0f;
                const float *t = tab + (kc + k0 + k) * SP_TAB;
                const float xv = X[k0 + k];
                const float pre = (xv - t[0]) * t[1] + t[2];  // the forward's expression: the same bits, hence the same mask as y > 0
                const float dz = pre > 0.0f ? acc : 0.0f;
                const float xhat = (xv - t[0]) * t[3];
                fb[(k0 + k) * SP_PLANE] = xin ? t[1] * ((dz - t[4]) - xhat * t[5]) : 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);  // the next batch's reads stay behind this batch's arithmetic (all sixteen hoisted: 57 VGPRs spilled)
        }
    };
    // ---- two programs, one barrier count: the producing and the multiplying wavefronts run their own loops (in one loop the register
    // allocator must keep both roles' state alive at once: 50 VGPRs spilled), meeting at the same two barriers per round -----------------
    if (producer) {
        // round g + 1's frame is formed while round g is multiplied; its pooled cells went to LDS a round earlier, its x a round earlier into the
        // other register set: ONE barrier per round, every global load a whole round ahead of its use
        load_p(0);
        load_x(0, XA);
        stage_a(0);
        if (steps > 1) load_p(1);
        __syncthreads();  // (also: the table)
        stage_b(0, XA);
        if (steps > 1) {
            load_x(1, XB);
            stage_a(1);
        }
        if (steps > 2) load_p(2);
        __syncthreads();
        auto step = [&](int g, float (&Xnext)[SP_KC], float (&Xfree)[SP_KC]) {  // Xnext: round g + 1; Xfree: round g's, read in the last step
            if (g + 2 < steps) load_x(g + 2, Xfree);
            if (g + 1 < steps) stage_b(g + 1, Xnext);
            if (g + 2 < steps) stage_a(g + 2);
            if (g + 3 < steps) load_p(g + 3);
            __syncthreads();
        };
        for (int g = 0; g < steps; g += 2) {
            step(g, XB, XA);
            if (g + 1 < steps) step(g + 1, XA, XB);
        }
        return;
    }
    load_w(0);
    stage_w(0);
    __syncthreads();
    __syncthreads();
    const int row = wave >> 1, col0 = (wave & 1) * 16;  // this wavefront's M-tile = dy row `row` of the unit, 16 cells from col0
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    auto multiply = [&](const float *fb, const float *wb) {
        if (EE_STEMB_SKIP & 1) return;
#pragma unroll
        for (int k = 0; k < SP_KC; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[(k * SP_FH + row + u) * SP_FW + col0 + i + kk], wb[((k * 4 + u) * 4 + kk) * 16 + i], acc, 0, 0, 0);
    };
    for (int g = 0; g < steps; ++g) {
        const float *fb = fr + (g & 1) * SP_FR, *wb = wp + (g & 1) * SP_WP;
        const bool last = g % SP_ROUNDS == SP_ROUNDS - 1;
        if (g + 1 < steps) load_w(g + 1);  // behind this round's products; into LDS at the end of the round
        multiply(fb, wb);
        if (g + 1 < steps) stage_w(g + 1);
        if (last) {
            // D[cell kk*4 + r][column i]: column -> (parity, image channel); into the [3][4][64] image tile
            if (i < 12) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(jc * 4 + 2 * row + jph) * 64 + 2 * (col0 + kk * 4 + r) + jpw] = acc[r];
            }
            acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        __syncthreads();
        if (last && tid < 3 * 4 * 16) {  // (the tile is next written four barriers from here)
            const Unit un = unit_of(g);
            const int q4 = tid & 15, orow = (tid >> 4) & 3, c = tid >> 6;
            const int H = 2 * p.OH;
            float *o = p.dx + ((static_cast<size_t>(un.n) * 3 + c) * H + 4 * un.t + orow) * 64 + 4 * q4;
            *reinterpret_cast<float4 *>(o) = *reinterpret_cast<const float4 *>(out + (c * 4 + orow) * 64 + 4 * q4);
        }
    }
}

}  // namespace

// d loss / d image from the gradient of the stem's POOLED activation (resnet.py:112-117 backwards: maxpool, relu, bn1, conv1) in one launch:
//   dy_pool (+ dy_pool2, nullable) [B,64,H/4,W/4], code = the pool's argmax codes, x [B,64,H/2,W/2] = conv1's output, bn1's parameters and
//   statistics (training: save_mean / save_invstd and `sums` = ee_bn_relu_pool_bwd_sums_f32's workspace with G groups; eval: running_mean /
//   running_var, sums = NULL, G = 0), weight [64,3,7,7] -> dx [B,3,H,W].  dgamma / dbeta [64] (nullable, training only) as
//   ee_bn_relu_pool_bwd_f32 writes them.  Bit-identical to ee_bn_relu_pool_bwd_f32 followed by ee_stem7x7s2_bwd_data_f32.
// W must be 64 and H a multiple of 4 (the Tiny-ImageNet stem), 64 channels; else EE_ERR_UNSUPPORTED.
EE_API int ee_stem_bn_pool_bwd_data_f32(const float *dy_pool, const float *dy_pool2, const uint8_t *code, const float *x, const float *gamma, const float *beta,
                                        const float *save_mean, const float *save_invstd, const float *running_mean, const float *running_var, float eps,
                                        int training, const float *sums, int G, float *dgamma, float *dbeta, const float *weight, float *dx, int B,
                                        int K, int H, int W, void *stream) {
    if (B < 0 || K < 1 || H < 4 || W < 4) return EE_ERR_SHAPE;
    if (K != SP_K || W != 2 * SP_OW || H % 4 != 0) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!dy_pool || !code || !x || !weight || !dx) return EE_ERR_NULL;
    if (training ? (!save_mean || !save_invstd || !sums || G < 1) : (!running_mean || !running_var)) return EE_ERR_NULL;
    if (!aligned16(dx)) return EE_ERR_ALIGN;
    const int OH = H / 2;
    const int64_t units = static_cast<int64_t>(B) * (OH / SP_TA);
    if (units > 0x7fffffffLL || static_cast<int64_t>(B) * K * OH * SP_OW > 0x7fffffffLL) return EE_ERR_SHAPE;
    static int ok = hipFuncSetAttribute(reinterpret_cast<const void *>(stem_bn_pool_bwd_data_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(SP_LDS)) == hipSuccess;
    if (!ok) return EE_ERR_UNSUPPORTED;
    const int64_t slots = static_cast<int64_t>(device_cus()) * 2;
    const StemBwdArgs a{dy_pool, dy_pool2, code, x, gamma, beta, training ? save_mean : running_mean, training ? save_invstd : running_var,
                        training ? sums : nullptr, weight, dx, training ? dgamma : nullptr, training ? dbeta : nullptr, eps, training, G, B, OH, OH / 2,
                        static_cast<int>(units), OH / SP_TA};
    EE_LAUNCH(stem_bn_pool_bwd_data_kernel, dim3(static_cast<unsigned>(units < slots ? units : slots)), dim3(SP_NT), SP_LDS, as_stream(stream), a);
    return launch_status();
}
