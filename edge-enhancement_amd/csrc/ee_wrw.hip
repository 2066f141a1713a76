// ee_wrw.hip - the WEIGHT gradients of the classifiers' convolutions (`loss.backward()` of a training step, experiments_tinyimagenet.py:304-306),
// four kernel families with one scheme - the reduction over images and pixels is cut over a few hundred workgroups, their partial gradients meet
// in a workspace, and wrw_sum_kernel adds them IN A FIXED ORDER (MIOpen's solvers use atomics: not reproducible from run to run):
//   wrw_wino_kernel<MAP>  Conv2d(3x3, stride 1, padding 1) on 16x16 / 8x8 / 4x4 / 2x2 maps (resnet.py:26-31 at 64x64 inputs): Winograd F(3x3, 2x2)
//   wrw_s2_kernel<H>      Conv2d(3x3, stride 2, padding 1) from 16x16 / 8x8 / 4x4 maps together with the block's 1x1 / stride 2 shortcut (:50-59, :132-142)
//   wrw_stem_kernel       the stem Conv2d(3, 64, 7, stride 2, padding 3) (:112)
//   wrw1x1_kernel         Conv2d(1x1, stride 1) of the bottleneck blocks (:75-100; ResNet-50 at ImageNet size), operands in their NCHW layout
// The first one in detail:
//
// Why: MIOpen's best solvers for these shapes are its NHWC implicit-GEMM kernels: per layer one 24 - 28 us product plus two to four layout
// transposes and a zero fill, 0.65 ms of the 2.1 ms update of a training step (profiles/round3_h_trace_breakdown.txt), summed with atomics (not
// reproducible from run to run).  Here, per 2x2 tile t of the output map (its 2x2 patch y of dy, the 4x4 input patch d around it):
//     dW = sum_t A^T [ (G y G^T) (.) (B^T d B) ] A        G = [[1,0],[.5,.5],[.5,-.5],[0,1]]   A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]]
// (B^T as in ee_wino.hip), i.e. 16 products per tile instead of 36, and the sum over tiles is 16 small GEMMs
//     M_xi[co][ci] = sum_t Yh_xi[co][t] * Xh_xi[ci][t]          on v_mfma_f32_16x16x4_f32
// A workgroup owns a 32 x 32 block of (co, ci) and a contiguous range of 16-tile chunks (a quarter of a 16x16 image, one 8x8 image, four 4x4
// images, sixteen 2x2 images): per half chunk wavefronts 0-3 transform one input patch per lane and wavefronts 4-7 one dy patch into LDS while
// wavefront w multiplies the xi pair (2w, 2w+1) of the previous half (16 MFMAs); the next chunk's pixels are already on their way (registers ->
// a second raw buffer during the products).
// The reduction over chunks is split over workgroups so that ~256 of them exist; each applies A^T . A to its partial M and writes 3x3 partial
// gradients to a workspace, which a second kernel adds IN A FIXED ORDER (one split - the 512-channel layers - writes the gradient itself):
// the result is reproducible bit for bit.
//
// CNN-body glue, not a row of SURVEY.md section 8.
#include "ee_common.hpp"

namespace {

using namespace ee;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// -DEE_WRW_SKIP=mask (scripts/native/wrw_bench.hip builds its own copies of this file with it; never the product): phase skipping inside the
// loop - 1: no global loads, 2: no raw stores, 4: no transforms, 8: operand reads without products, 16: no operand reads and no products
#ifndef EE_WRW_SKIP
#define EE_WRW_SKIP 0
#endif

constexpr int WR_NT = 512;
constexpr int WR_TS = 10;               // row stride of the transform-domain operands, in float2: rows 10 apart, the four k of a wavefront next to each other -> 32 distinct bank pairs per half wavefront
constexpr int WR_T2 = 8 * 64 * WR_TS;   // one operand buffer = HALF a chunk: [8 xi pairs][32 co rows + 32 ci rows][8 tiles + 2] float2
constexpr int WR_YS = 68;               // dy pixels of a chunk: 64 per channel, channels 68 apart

template <int MAP>
struct WrwGeo {
    static constexpr int TX = MAP / 2, TI = TX * TX;
    static constexpr int IMGS = TI >= 16 ? 1 : 16 / TI;  // images per chunk of 16 tiles
    static constexpr int CPI = TI >= 16 ? TI / 16 : 1;   // chunks per image (16x16 maps: 4 bands of four rows)
    // zero-ringed pixel frames of a chunk; the channel stride is 4 mod 64 floats so that a half wavefront's patch reads (16 channels x 2
    // neighbouring tiles, 8 bytes each) fall on 32 distinct bank pairs
    static constexpr int FW = MAP == 16 ? 20 : MAP == 8 ? 12 : MAP == 4 ? 6 : 4;                // row stride
    static constexpr int FI = MAP == 16 ? 6 * 20 : MAP == 8 ? 10 * 12 : MAP == 4 ? 36 : 4;      // one image's frame (16x16: a band of 4 + 2 rows; 2x2: no ring, the four pixels)
    static constexpr int CS = MAP == 16 ? 132 : MAP == 8 ? 132 : MAP == 4 ? 196 : WR_YS;
    static_assert(IMGS * FI <= CS && CS % 64 == 4, "frames fit, channels on distinct banks");
    static constexpr int RAW = 32 * CS + 32 * WR_YS;  // one raw buffer: x frames, dy pixels
    static constexpr size_t lds_bytes = 2 * WR_T2 * sizeof(float2) + 2 * RAW * sizeof(float);
};

struct WrwDims {
    int B, KC, RC;      // images, input channels (x), output channels (dy)
    int chunks, cpw, S; // 16-tile chunks in all, per workgroup, number of splits
    int grouped;        // workgroup numbering: the (co, ci) blocks of a split on ONE XCD (they read the same pixels)
};

template <int MAP>
__global__ __launch_bounds__(WR_NT) void wrw_wino_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ out, WrwDims d) {
    using G = WrwGeo<MAP>;
    extern __shared__ __align__(16) float lds[];
    float2 *tp = reinterpret_cast<float2 *>(lds);
    float2 *tp1 = tp + WR_T2;
    float *raw0 = lds + 4 * WR_T2, *raw1 = raw0 + G::RAW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int ncb = d.RC / 32, nkb = d.KC / 32, ntile = ncb * nkb;
    int s, tile;
    if (d.grouped) {
        const int g = blockIdx.x & 7, k = blockIdx.x >> 3;
        s = g + 8 * (k / ntile), tile = k % ntile;
        if (s >= d.S) return;
    } else {
        s = blockIdx.x / ntile, tile = blockIdx.x - s * ntile;
    }
    const int co0 = (tile / nkb) * 32, ci0 = (tile % nkb) * 32;
    const int q0 = s * d.cpw, nq = d.chunks - q0 < d.cpw ? d.chunks - q0 : d.cpw;
    // ---- this lane's share of a chunk's pixels: one float4 of x (channel pc, float4 number pf of its 64 pixels), one of dy, and on 16x16 maps
    // one float4 of the two halo rows for the first 256 lanes
    const int pc = threadIdx.x >> 4, pf = threadIdx.x & 15;
    const int hc = (threadIdx.x >> 3) & 31, hh = threadIdx.x & 7;
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 xv = zero4, hv = zero4, yv = zero4;
    auto load = [&](int q) {
        if (MAP == 16) {
            const int img = q >> 2, cc = q & 3;
            xv = *reinterpret_cast<const float4 *>(x + (static_cast<size_t>(img) * d.KC + ci0 + pc) * 256 + cc * 64 + 4 * pf);
            yv = *reinterpret_cast<const float4 *>(dy + (static_cast<size_t>(img) * d.RC + co0 + pc) * 256 + cc * 64 + 4 * pf);
            if (threadIdx.x < 256) {
                const bool up = hh < 4, ok = up ? cc > 0 : cc < 3;
                const int off = up ? cc * 64 - 16 + 4 * hh : cc * 64 + 64 + 4 * (hh - 4);
                hv = ok ? *reinterpret_cast<const float4 *>(x + (static_cast<size_t>(img) * d.KC + ci0 + hc) * 256 + off) : zero4;
            }
        } else if (MAP == 8) {
            xv = *reinterpret_cast<const float4 *>(x + (static_cast<size_t>(q) * d.KC + ci0 + pc) * 64 + 4 * pf);
            yv = *reinterpret_cast<const float4 *>(dy + (static_cast<size_t>(q) * d.RC + co0 + pc) * 64 + 4 * pf);
        } else if (MAP == 4) {
            const int img = 4 * q + (pf >> 2), r = pf & 3;
            const bool ok = img < d.B;
            xv = ok ? *reinterpret_cast<const float4 *>(x + (static_cast<size_t>(img) * d.KC + ci0 + pc) * 16 + 4 * r) : zero4;
            yv = ok ? *reinterpret_cast<const float4 *>(dy + (static_cast<size_t>(img) * d.RC + co0 + pc) * 16 + 4 * r) : zero4;
        } else {
            const int img = 16 * q + pf;
            const bool ok = img < d.B;
            xv = ok ? *reinterpret_cast<const float4 *>(x + (static_cast<size_t>(img) * d.KC + ci0 + pc) * 4) : zero4;
            yv = ok ? *reinterpret_cast<const float4 *>(dy + (static_cast<size_t>(img) * d.RC + co0 + pc) * 4) : zero4;
        }
    };
    auto store = [&](float *buf) {  // the loaded pixels -> frame interiors (the rings stay zero) and the dy rows
        float *xs = buf, *ys = buf + 32 * G::CS;
        float *dst;
        if (MAP == 16) dst = xs + pc * G::CS + (1 + (pf >> 2)) * G::FW + 1 + 4 * (pf & 3);
        else if (MAP == 8) dst = xs + pc * G::CS + (1 + (pf >> 1)) * G::FW + 1 + 4 * (pf & 1);
        else if (MAP == 4) dst = xs + pc * G::CS + (pf >> 2) * G::FI + (1 + (pf & 3)) * G::FW + 1;
        else dst = xs + pc * G::CS + pf * G::FI;
        if (MAP == 2) {
            *reinterpret_cast<float4 *>(dst) = xv;
        } else {
            dst[0] = xv.x, dst[1] = xv.y, dst[2] = xv.z, dst[3] = xv.w;
        }
        if (MAP == 16 && threadIdx.x < 256) {
            float *h = xs + hc * G::CS + (hh < 4 ? 0 : 5) * G::FW + 1 + 4 * (hh & 3);
            h[0] = hv.x, h[1] = hv.y, h[2] = hv.z, h[3] = hv.w;
        }
        *reinterpret_cast<float4 *>(ys + pc * WR_YS + 4 * pf) = yv;
    };
    // ---- this lane's patch of a HALF chunk (8 tiles): wavefronts 0-3 transform the input patches (channel row, tile), 4-7 the dy patches
    const int row = l15 + 16 * (wave & 1), t8 = 4 * ((wave >> 1) & 1) + lq;
    const bool xside = wave < 4;
    auto transform = [&](const float *buf, int half, float2 *tq) {
        const int t = 8 * half + t8;
        if (xside) {
            float dd[4][4];
            if (MAP == 2) {  // the 4x4 patch of a 2x2 map is the map inside a ring of zeros
                const float4 v = *reinterpret_cast<const float4 *>(buf + row * G::CS + 4 * t);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dd[i][j] = 0.0f;
                dd[1][1] = v.x, dd[1][2] = v.y, dd[2][1] = v.z, dd[2][2] = v.w;
            } else {
                int po;
                if (MAP == 16) po = 2 * (t >> 3) * G::FW + 2 * (t & 7);
                else if (MAP == 8) po = 2 * (t >> 2) * G::FW + 2 * (t & 3);
                else po = (t >> 2) * G::FI + 2 * ((t >> 1) & 1) * G::FW + 2 * (t & 1);
                const float *p = buf + row * G::CS + po;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float2 lo = *reinterpret_cast<const float2 *>(p + i * G::FW), hi = *reinterpret_cast<const float2 *>(p + i * G::FW + 2);
                    dd[i][0] = lo.x, dd[i][1] = lo.y, dd[i][2] = hi.x, dd[i][3] = hi.y;
                }
            }
            float tt[4][4];  // B^T d
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tt[0][j] = dd[0][j] - dd[2][j];
                tt[1][j] = dd[1][j] + dd[2][j];
                tt[2][j] = dd[2][j] - dd[1][j];
                tt[3][j] = dd[1][j] - dd[3][j];
            }
            float2 *xo = tq + (32 + row) * WR_TS + t8;  // pair p = 2 a + b / 2 of xi = 4 a + b
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                xo[(2 * a + 0) * 64 * WR_TS] = make_float2(tt[a][0] - tt[a][2], tt[a][1] + tt[a][2]);
                xo[(2 * a + 1) * 64 * WR_TS] = make_float2(tt[a][2] - tt[a][1], tt[a][1] - tt[a][3]);
            }
        } else {
            int yo;
            if (MAP == 16) yo = 2 * (t >> 3) * 16 + 2 * (t & 7);
            else if (MAP == 8) yo = 2 * (t >> 2) * 8 + 2 * (t & 3);
            else if (MAP == 4) yo = (t >> 2) * 16 + 2 * ((t >> 1) & 1) * 4 + 2 * (t & 1);
            else yo = t * 4;
            const float *yp = buf + 32 * G::CS + row * WR_YS + yo;
            const float2 y0 = *reinterpret_cast<const float2 *>(yp), y1 = *reinterpret_cast<const float2 *>(yp + (MAP == 2 ? 2 : MAP));
            float gy[4][2];  // G y
            gy[0][0] = y0.x, gy[0][1] = y0.y;
            gy[1][0] = 0.5f * (y0.x + y1.x), gy[1][1] = 0.5f * (y0.y + y1.y);
            gy[2][0] = 0.5f * (y0.x - y1.x), gy[2][1] = 0.5f * (y0.y - y1.y);
            gy[3][0] = y1.x, gy[3][1] = y1.y;
            float2 *yo2 = tq + row * WR_TS + t8;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                yo2[(2 * a + 0) * 64 * WR_TS] = make_float2(gy[a][0], 0.5f * (gy[a][0] + gy[a][1]));
                yo2[(2 * a + 1) * 64 * WR_TS] = make_float2(0.5f * (gy[a][0] - gy[a][1]), gy[a][1]);
            }
        }
    };
    f32x4 acc[2][2][2];  // [xi of the pair][co tile][ci tile]
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[e][m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    auto multiply = [&](const float2 *tq) {  // M_xi += Yh_xi Xh_xi^T over the half chunk's 8 tiles for xi = 2 wave, 2 wave + 1
        const float2 *ap = tq + (wave * 64 + l15) * WR_TS + lq;
        float2 av[2][2], bv[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) av[m][ks] = ap[16 * m * WR_TS + 4 * ks], bv[m][ks] = ap[(32 + 16 * m) * WR_TS + 4 * ks];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (EE_WRW_SKIP & 8) {  // timing only: the operand reads without the products
                        acc[0][m][n][0] += av[m][ks].x + bv[n][ks].x, acc[1][m][n][0] += av[m][ks].y + bv[n][ks].y;
                        continue;
                    }
                    acc[0][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][ks].x, bv[n][ks].x, acc[0][m][n], 0, 0, 0);
                    acc[1][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][ks].y, bv[n][ks].y, acc[1][m][n], 0, 0, 0);
                }
    };
    // ---- pipeline over HALF chunks h = 0 .. 2 nq - 1 (chunk h / 2, tiles 8 (h & 1) ..): during half round h every wavefront transforms its
    // patch of half h + 1 into the other operand buffer, then multiplies half h - one barrier per half round, and the matrix pipe of a SIMD works
    // for one of its two wavefronts while the other waits for LDS.  Chunk r's pixels sit in raw[r & 1]; chunk r + 1's go from registers into
    // the other raw buffer during the first half of chunk r, and chunk r + 2's loads are issued right after.
    load(q0);
    for (int i = threadIdx.x; i < 2 * G::RAW / 4; i += WR_NT) reinterpret_cast<float4 *>(raw0)[i] = zero4;  // the rings (and everything else) of both buffers
    __syncthreads();
    store(raw0);
    if (nq > 1) load(q0 + 1);
    __syncthreads();
    transform(raw0, 0, tp);
    __syncthreads();
    const int nh = 2 * nq;
    for (int h = 0; h < nh; ++h) {
        const int r = h >> 1;
        // the two wavefronts of a SIMD (w and w + 4) take the half round's two jobs in OPPOSITE order: all eight running the same sequence
        // between barriers meet in the LDS-bound transform, then in the matrix pipe, and the phases simply add up (phase skipping,
        // scripts/native/wrw_bench.hip: 0.8 us transform + 1.2 us products + 0.2 us raw stores + 0.2 us barriers = the 2.5 us a chunk took)
        if (!xside && !(EE_WRW_SKIP & 16)) multiply((h & 1) ? tp1 : tp);
        if (h + 1 < nh && !(EE_WRW_SKIP & 4)) transform(((h + 1) >> 1) & 1 ? raw1 : raw0, (h + 1) & 1, (h & 1) ? tp : tp1);
        if (!(h & 1) && r + 1 < nq) {
            if (!(EE_WRW_SKIP & 2)) store((r & 1) ? raw0 : raw1);
            if (r + 2 < nq && !(EE_WRW_SKIP & 1)) load(q0 + r + 2);
        }
        if (xside && !(EE_WRW_SKIP & 16)) multiply((h & 1) ? tp1 : tp);
        __syncthreads();
    }
    // ---- this workgroup's share of dW: the accumulators meet in LDS (D[row = 4 lq + reg][col = l15] -> ms[xi][co][ci], rows 33 apart), every lane
    // applies A^T . A to two (co, ci) pairs and writes their nine taps: out[s][co][ci][3][3] - the weight gradient itself when there is one split
    float *ms = lds;
    constexpr int MS = 32 * 33;
    static_assert(16 * MS * sizeof(float) <= 2 * WR_T2 * sizeof(float2), "the accumulators fit the operand area");
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int xi = 4 * (wave >> 1) + 2 * (wave & 1) + e;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) ms[xi * MS + (16 * m + 4 * lq + rg) * 33 + 16 * n + l15] = acc[e][m][n][rg];
    }
    __syncthreads();
    float *op = out + static_cast<size_t>(s) * 9 * d.RC * d.KC;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int co = (threadIdx.x >> 5) + 16 * h, ci = threadIdx.x & 31;
        float col[4][3];  // M A
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float m0 = ms[(4 * a + 0) * MS + co * 33 + ci], m1 = ms[(4 * a + 1) * MS + co * 33 + ci], m2 = ms[(4 * a + 2) * MS + co * 33 + ci],
                        m3 = ms[(4 * a + 3) * MS + co * 33 + ci];
            col[a][0] = (m0 + m1) + m2, col[a][1] = m1 - m2, col[a][2] = (m1 + m2) - m3;
        }
        float *o = op + (static_cast<size_t>(co0 + co) * d.KC + ci0 + ci) * 9;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            o[j] = (col[0][j] + col[1][j]) + col[2][j];
            o[3 + j] = col[1][j] - col[2][j];
            o[6 + j] = (col[1][j] + col[2][j]) - col[3][j];
        }
    }
}

// dw = sum_s part[s] (n4 float4 each), always in the same order: a workgroup takes 64 float4 columns, its four wavefronts the splits
// s = 0, 4, 8 ... / 1, 5, 9 ... / ..., and wavefront 0 adds the four sums
__global__ __launch_bounds__(256) void wrw_sum_kernel(const float4 *__restrict__ part, float4 *__restrict__ dw, int S, int n4) {
    __shared__ float4 sm[3][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), sg = threadIdx.x >> 6;
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (col < n4)
        for (int s = sg; s < S; s += 4) {
            const float4 v = part[static_cast<size_t>(s) * n4 + col];
            a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
        }
    if (sg) sm[sg - 1][threadIdx.x & 63] = a;
    __syncthreads();
    if (sg == 0 && col < n4) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float4 v = sm[g][threadIdx.x];
            a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
        }
        dw[col] = a;
    }
}

// how the reduction over tiles is cut: about 256 workgroups in all, every one with the same number of 16-tile chunks (but the last)
void wrw_plan(int B, int KC, int RC, int H, WrwDims &d) {
    d.B = B, d.KC = KC, d.RC = RC;
    d.chunks = H == 16 ? 4 * B : H == 8 ? B : H == 4 ? (B + 3) / 4 : (B + 15) / 16;
    const int ntile = (KC / 32) * (RC / 32);
    const int target = ntile >= 256 ? 1 : 256 / ntile;
    d.cpw = (d.chunks + target - 1) / target;
    d.S = (d.chunks + d.cpw - 1) / d.cpw;
    d.grouped = d.S >= 8;
}

template <int MAP>
int wrw_launch(const float *x, const float *dy, float *out, const WrwDims &d, hipStream_t st) {
    using G = WrwGeo<MAP>;
    static int ok = hipFuncSetAttribute(reinterpret_cast<const void *>(wrw_wino_kernel<MAP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(G::lds_bytes)) == hipSuccess;
    if (!ok) return EE_ERR_UNSUPPORTED;
    const int ntile = (d.KC / 32) * (d.RC / 32);
    const unsigned grid = static_cast<unsigned>(d.grouped ? 8 * ((d.S + 7) / 8) * ntile : d.S * ntile);
    EE_LAUNCH(wrw_wino_kernel<MAP>, dim3(grid), dim3(WR_NT), G::lds_bytes, st, x, dy, out, d);
    return launch_status();
}

// ---- stride 2: the first convolution of a down-sampling block (3x3 / stride 2 / padding 1, H x H -> H/2 x H/2, H = 16, 8 or 4) and, optionally,
// the block's shortcut Conv2d(1x1, stride 2) of the SAME input (resnet.py:50-59, :132-142) in one launch ---------------------------------------
// No Winograd at stride 2: per tap (kh, kw) a plain product dW_t[co][ci] = sum over (image, oh, ow) dy[co][oh][ow] * x[ci][2 oh + kh - 1][2 ow + kw - 1]
// on v_mfma_f32_16x16x4_f32, the MFMA operands read straight from the chunk's raw pixels in LDS (dy rows; zero-ringed x frames, the tap a constant
// offset).  A workgroup owns 32 x 32 (co, ci) and a range of 16-output-pixel chunks (two output rows of a 16x16 input's 8x8 result, one 8x8 input
// image, four 4x4 ones); wavefront w multiplies (co, ci) quarter w & 3 for five of the ten taps (w >> 2: taps 0-4 / 5-8 and the shortcut's 1x1,
// whose input pixel is the centre tap's).  Same split over ~256 workgroups, same fixed-order sum kernel as above.
constexpr int S2W_YS = 18;  // dy pixels of a chunk: 16 per channel, channels 18 apart (the A operand's 32 lanes of a group on 32 banks)

template <int H>
struct WrwS2Geo {
    static constexpr int OH = H / 2, OP = OH * OH;
    static constexpr int IMGS = OP >= 16 ? 1 : 16 / OP;      // images per chunk of 16 output pixels
    static constexpr int FR = H == 16 ? 5 : H == 8 ? 9 : 5;  // frame rows: input rows 4c-1 .. 4c+3 of a band / -1 .. 7 / -1 .. 3
    static constexpr int FW = H == 16 ? 20 : H == 8 ? 10 : 6; // frame row stride: columns -1 .. H-1 (no right or bottom padding is ever read)
    static constexpr int FI = FR * FW;
    static constexpr int CS = IMGS * FI + 1 + ((IMGS * FI) & 1);  // odd channel stride
    static constexpr int RAW = 32 * CS + 2 * 32 * S2W_YS;         // x frames, dy3 rows, dy1 rows
    static constexpr int MSF = 10 * 32 * 33;                      // the epilogue's accumulator exchange
    static constexpr size_t lds_bytes = (2 * RAW > MSF ? 2 * RAW : MSF) * sizeof(float);
};

template <int H>
__global__ __launch_bounds__(WR_NT) void wrw_s2_kernel(const float *__restrict__ x, const float *__restrict__ dy3, const float *__restrict__ dy1,
                                                       float *__restrict__ out, WrwDims d) {
    using G = WrwS2Geo<H>;
    extern __shared__ __align__(16) float lds[];
    float *raw0 = lds, *raw1 = lds + G::RAW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int ncb = d.RC / 32, nkb = d.KC / 32, ntile = ncb * nkb;
    int s, tile;
    if (d.grouped) {
        const int g = blockIdx.x & 7, k = blockIdx.x >> 3;
        s = g + 8 * (k / ntile), tile = k % ntile;
        if (s >= d.S) return;
    } else {
        s = blockIdx.x / ntile, tile = blockIdx.x - s * ntile;
    }
    const int co0 = (tile / nkb) * 32, ci0 = (tile % nkb) * 32;
    const int q0 = s * d.cpw, nq = d.chunks - q0 < d.cpw ? d.chunks - q0 : d.cpw;
    const bool pair = dy1 != nullptr;
    // ---- a lane's share of a chunk's pixels: one float4 of x (channel pc, float4 pf of its 64 pixels); lanes < 128 one float4 of dy3 and of dy1
    // (channel threadIdx.x >> 2, float4 threadIdx.x & 3 of its 16 pixels); 16x16 inputs: lanes 128 .. 255 one float4 of the halo row above the band
    const int pc = threadIdx.x >> 4, pf = threadIdx.x & 15;
    const int yc = (threadIdx.x >> 2) & 31, yf = threadIdx.x & 3;
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 xv = zero4, hv = zero4, y3v = zero4, y1v = zero4;
    auto load = [&](int q) {
        if (H == 16) {
            const int img = q >> 2, c = q & 3;
            xv = *reinterpret_cast<const float4 *>(x + (static_cast<size_t>(img) * d.KC + ci0 + pc) * 256 + c * 64 + 4 * pf);
            if (threadIdx.x < 128) {
                y3v = *reinterpret_cast<const float4 *>(dy3 + (static_cast<size_t>(img) * d.RC + co0 + yc) * 64 + c * 16 + 4 * yf);
                if (pair) y1v = *reinterpret_cast<const float4 *>(dy1 + (static_cast<size_t>(img) * d.RC + co0 + yc) * 64 + c * 16 + 4 * yf);
            } else if (threadIdx.x < 256) {
                hv = c > 0 ? *reinterpret_cast<const float4 *>(x + (static_cast<size_t>(img) * d.KC + ci0 + yc) * 256 + c * 64 - 16 + 4 * yf) : zero4;
            }
        } else if (H == 8) {
            xv = *reinterpret_cast<const float4 *>(x + (static_cast<size_t>(q) * d.KC + ci0 + pc) * 64 + 4 * pf);
            if (threadIdx.x < 128) {
                y3v = *reinterpret_cast<const float4 *>(dy3 + (static_cast<size_t>(q) * d.RC + co0 + yc) * 16 + 4 * yf);
                if (pair) y1v = *reinterpret_cast<const float4 *>(dy1 + (static_cast<size_t>(q) * d.RC + co0 + yc) * 16 + 4 * yf);
            }
        } else {
            const int img = 4 * q + (pf >> 2);
            xv = img < d.B ? *reinterpret_cast<const float4 *>(x + (static_cast<size_t>(img) * d.KC + ci0 + pc) * 16 + 4 * (pf & 3)) : zero4;
            if (threadIdx.x < 128) {
                const int im2 = 4 * q + yf;  // float4 yf of a channel's 16 dy pixels = the 2x2 result of image yf
                const bool ok = im2 < d.B;
                y3v = ok ? *reinterpret_cast<const float4 *>(dy3 + (static_cast<size_t>(im2) * d.RC + co0 + yc) * 4) : zero4;
                if (pair) y1v = ok ? *reinterpret_cast<const float4 *>(dy1 + (static_cast<size_t>(im2) * d.RC + co0 + yc) * 4) : zero4;
            }
        }
    };
    auto store = [&](float *buf) {  // frame row 0 / column 0 = input row (band start - 1) / column -1: the ring stays zero
        float *dst;
        if (H == 16) dst = buf + pc * G::CS + (1 + (pf >> 2)) * G::FW + 1 + 4 * (pf & 3);
        else if (H == 8) dst = buf + pc * G::CS + (1 + (pf >> 1)) * G::FW + 1 + 4 * (pf & 1);
        else dst = buf + pc * G::CS + (pf >> 2) * G::FI + (1 + (pf & 3)) * G::FW + 1;
        dst[0] = xv.x, dst[1] = xv.y, dst[2] = xv.z, dst[3] = xv.w;
        if (threadIdx.x < 128) {
            float *y3 = buf + 32 * G::CS + yc * S2W_YS + 4 * yf;
            y3[0] = y3v.x, y3[1] = y3v.y, y3[2] = y3v.z, y3[3] = y3v.w;
            if (pair) {
                float *y1 = y3 + 32 * S2W_YS;
                y1[0] = y1v.x, y1[1] = y1v.y, y1[2] = y1v.z, y1[3] = y1v.w;
            }
        } else if (H == 16 && threadIdx.x < 256) {
            float *h = buf + yc * G::CS + 1 + 4 * yf;
            h[0] = hv.x, h[1] = hv.y, h[2] = hv.z, h[3] = hv.w;
        }
    };
    // ---- this wavefront's products: quarter (m, n) of the (co, ci) block, taps t0 .. t0 + 4 (tap 9 = the shortcut's 1x1: dy1, centre pixel)
    const int m = (wave >> 1) & 1, n = wave & 1, t0 = 5 * (wave >> 2);
    int pos[4];  // frame offset of output pixel k = lq + 4 ks at tap (0, 0)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int p = lq + 4 * ks;
        if (H == 16) pos[ks] = 2 * (p >> 3) * G::FW + 2 * (p & 7);
        else if (H == 8) pos[ks] = 2 * (p >> 2) * G::FW + 2 * (p & 3);
        else pos[ks] = (p >> 2) * G::FI + 2 * ((p >> 1) & 1) * G::FW + 2 * (p & 1);
    }
    f32x4 acc[5];
#pragma unroll
    for (int a = 0; a < 5; ++a) acc[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    auto multiply = [&](const float *buf) {
        const float *ap = buf + 32 * G::CS + (16 * m + l15) * S2W_YS + lq;
        const float *bp = buf + (16 * n + l15) * G::CS;
        float av[4], a1[4], bv[5][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            av[ks] = ap[4 * ks];
            a1[ks] = ap[32 * S2W_YS + 4 * ks];
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                const int t = t0 + a, tt = t == 9 ? 4 : t;
                bv[a][ks] = bp[pos[ks] + (tt / 3) * G::FW + tt % 3];
            }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int a = 0; a < 5; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(t0 + a == 9 ? a1[ks] : av[ks], bv[a][ks], acc[a], 0, 0, 0);
    };
    // ---- pipeline: chunk r's pixels are in raw[r & 1] when round r starts; chunk r + 1's go from registers into the other buffer during round r
    load(q0);
    for (int i = threadIdx.x; i < 2 * G::RAW; i += WR_NT) raw0[i] = 0.0f;
    __syncthreads();
    store(raw0);
    if (nq > 1) load(q0 + 1);
    __syncthreads();
    for (int r = 0; r < nq; ++r) {
        if (r + 1 < nq) {
            store((r & 1) ? raw0 : raw1);
            if (r + 2 < nq) load(q0 + r + 2);
        }
        multiply((r & 1) ? raw1 : raw0);
        __syncthreads();
    }
    // ---- this workgroup's share: ms[tap][co][ci] in LDS, then rows of [co][ci][9] (and [co][ci] for the shortcut) written whole
    float *ms = lds;
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) ms[((t0 + a) * 32 + 16 * m + 4 * lq + rg) * 33 + 16 * n + l15] = acc[a][rg];
    __syncthreads();
    const size_t plane = static_cast<size_t>(d.RC) * d.KC;
    float *o3 = out + static_cast<size_t>(s) * (pair ? 10 : 9) * plane, *o1 = o3 + 9 * plane;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int co = (threadIdx.x >> 5) + 16 * h, ci = threadIdx.x & 31;
        const size_t e = static_cast<size_t>(co0 + co) * d.KC + ci0 + ci;
#pragma unroll
        for (int t = 0; t < 9; ++t) o3[e * 9 + t] = ms[(t * 32 + co) * 33 + ci];
        if (pair) o1[e] = ms[(9 * 32 + co) * 33 + ci];
    }
}

void wrw_s2_plan(int B, int KC, int RC, int H, WrwDims &d) {
    d.B = B, d.KC = KC, d.RC = RC;
    d.chunks = H == 16 ? 4 * B : H == 8 ? B : (B + 3) / 4;
    const int ntile = (KC / 32) * (RC / 32);
    const int target = ntile >= 256 ? 1 : 256 / ntile;
    d.cpw = (d.chunks + target - 1) / target;
    d.S = (d.chunks + d.cpw - 1) / d.cpw;
    d.grouped = d.S >= 8;
}

template <int H>
int wrw_s2_launch(const float *x, const float *dy3, const float *dy1, float *out, const WrwDims &d, hipStream_t st) {
    using G = WrwS2Geo<H>;
    static int ok = hipFuncSetAttribute(reinterpret_cast<const void *>(wrw_s2_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(G::lds_bytes)) == hipSuccess;
    if (!ok) return EE_ERR_UNSUPPORTED;
    const int ntile = (d.KC / 32) * (d.RC / 32);
    const unsigned grid = static_cast<unsigned>(d.grouped ? 8 * ((d.S + 7) / 8) * ntile : d.S * ntile);
    EE_LAUNCH(wrw_s2_kernel<H>, dim3(grid), dim3(WR_NT), G::lds_bytes, st, x, dy3, dy1, out, d);
    return launch_status();
}

// ---- the stem: Conv2d(3, 64, 7, stride 2, padding 3) (resnet.py:112) ------------------------------------------------------------------------
// dW[co][j] with j = (ci, kh, kw) flattened (147, padded to 160) = sum over output pixels dy[co][pixel] * x[ci][2 oh + kh - 3][2 ow + kw - 3]: a
// [64] x [160] x [B OH OW] product.  A workgroup owns ALL of dW for a range of 16-pixel chunks (16 consecutive output pixels of a row): per chunk
// the 7 input rows x 40 columns around them (3 channels: 840 floats, zero outside the image) and the 64 x 16 dy values go global -> registers ->
// LDS; wavefront w multiplies output-channel tile w & 3 against five of the ten 16-wide tap tiles (w >> 2), the B operand of lane (tap j, pixel k)
// being frame[ci][kh][2 k + kw + 1].  Split over ~256 workgroups, the same fixed-order sum.
constexpr int ST_FW = 40, ST_FI = 7 * ST_FW, ST_XF = 3 * ST_FI;   // one chunk's x frames
constexpr int ST_RAW = ST_XF + 64 * S2W_YS;                       // + its dy rows
constexpr int ST_J = 147, ST_JP = 160;
constexpr int ST_MS = ST_JP * 65;                                 // the epilogue's accumulator exchange [j][co], rows 65 apart

__global__ __launch_bounds__(WR_NT) void wrw_stem_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ out, int B, int H, int W,
                                                        int chunks, int cpw) {
    extern __shared__ __align__(16) float lds[];
    float *raw0 = lds, *raw1 = lds + ST_RAW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int OH = H / 2, OW = W / 2, cpr = OW / 16;  // chunks per output row
    const int s = blockIdx.x;
    const int q0 = s * cpw, nq = chunks - q0 < cpw ? chunks - q0 : cpw;
    // ---- a lane's share of a chunk: lanes 0 .. 209 one float4 of x (channel, frame row, float4 column), lanes 256 .. 511 one float4 of dy
    const int xc = threadIdx.x / 70, xr = (threadIdx.x % 70) / 10, xq = threadIdx.x % 10;
    const int yc = (threadIdx.x >> 2) & 63, yf = threadIdx.x & 3;
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 xv = zero4, yv = zero4;
    auto load = [&](int q) {
        const int row = q / cpr, half = q - row * cpr;  // row = (image, oh)
        const int img = row / OH, oh = row - img * OH, ow0 = 16 * half;
        if (threadIdx.x < 210) {
            const int ih = 2 * oh - 3 + xr, iw = 2 * ow0 - 4 + 4 * xq;  // a float4 of the image or wholly outside it (W is a multiple of 4)
            xv = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? *reinterpret_cast<const float4 *>(x + ((static_cast<size_t>(img) * 3 + xc) * H + ih) * W + iw) : zero4;
        } else if (threadIdx.x >= 256) {
            yv = *reinterpret_cast<const float4 *>(dy + ((static_cast<size_t>(img) * 64 + yc) * OH + oh) * OW + ow0 + 4 * yf);
        }
    };
    auto store = [&](float *buf) {
        if (threadIdx.x < 210) {
            *reinterpret_cast<float4 *>(buf + xc * ST_FI + xr * ST_FW + 4 * xq) = xv;
        } else if (threadIdx.x >= 256) {
            float *y = buf + ST_XF + yc * S2W_YS + 4 * yf;
            y[0] = yv.x, y[1] = yv.y, y[2] = yv.z, y[3] = yv.w;
        }
    };
    const int m = wave & 3, n0 = 5 * (wave >> 2);
    int toff[5];  // frame offset of this lane's tap j = 16 (n0 + a) + l15 at pixel 0; the padding taps (j >= 147) read tap 0 and are never stored
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        const int j = 16 * (n0 + a) + l15, jj = j < ST_J ? j : 0;
        const int ci = jj / 49, kh = (jj % 49) / 7, kw = jj % 7;
        toff[a] = ci * ST_FI + kh * ST_FW + kw + 1 + 2 * lq;
    }
    f32x4 acc[5];
#pragma unroll
    for (int a = 0; a < 5; ++a) acc[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    auto multiply = [&](const float *buf) {
        const float *ap = buf + ST_XF + (16 * m + l15) * S2W_YS + lq;
        float av[4], bv[5][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            av[ks] = ap[4 * ks];
#pragma unroll
            for (int a = 0; a < 5; ++a) bv[a][ks] = buf[toff[a] + 8 * ks];  // pixel k = lq + 4 ks -> column 2 k
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int a = 0; a < 5; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[a][ks], acc[a], 0, 0, 0);
    };
    load(q0);
    __syncthreads();
    store(raw0);
    if (nq > 1) load(q0 + 1);
    __syncthreads();
    for (int r = 0; r < nq; ++r) {
        if (r + 1 < nq) {
            store((r & 1) ? raw0 : raw1);
            if (r + 2 < nq) load(q0 + r + 2);
        }
        multiply((r & 1) ? raw1 : raw0);
        __syncthreads();
    }
    float *ms = lds;  // D[row = co = 4 lq + reg][col = tap l15]
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) ms[(16 * (n0 + a) + l15) * 65 + 16 * m + 4 * lq + rg] = acc[a][rg];
    __syncthreads();
    float *o = out + static_cast<size_t>(s) * 64 * ST_J;
    for (int i = threadIdx.x; i < 64 * ST_J; i += WR_NT) {
        const int co = i / ST_J, j = i - co * ST_J;
        o[i] = ms[j * 65 + co];
    }
}

void wrw_stem_plan(int B, int H, int W, int &chunks, int &cpw, int &S) {
    chunks = B * (H / 2) * (W / 32);
    cpw = (chunks + 255) / 256;
    S = (chunks + cpw - 1) / cpw;
}

// ---- 1x1 / stride 1 (ResNet-50's bottleneck convolutions at ImageNet size, resnet.py:75-100; BASELINE config 5) --------------------------------------
// dW[co][ci] = sum over images and pixels dy[b][co][p] * x[b][ci][p]: a product whose two operands are both contiguous along the reduction (NCHW rows).
// MIOpen's searched solvers for it are NHWC implicit-GEMM kernels behind layout transposes and zero fills - 31 % of a free-AT repeat on ResNet-50
// (profiles/round3_m_free_at_trace_breakdown.txt).  Here a workgroup of four wavefronts owns a 64 x 64 block of (co, ci) and a range of 32-pixel
// chunks (32 consecutive pixels of one image); a chunk's 64 + 64 rows arrive as float4 (8 lanes per row: coalesced), go to LDS transposed -
// [pixel][channel], pixels 65 floats apart, so that both these stores and the MFMA operand reads (32 lanes along the channel) fall on 32 distinct
// banks - and wavefront w multiplies quarter w of the block with sixteen v_mfma_f32_32x32x2_f32.  33 KB of LDS and 256 lanes per workgroup:
// four workgroups per CU hide each other's latencies.  Same split over workgroups, same fixed-order sum.
constexpr int P1_NT = 256, P1_KC = 32, P1_CS = 65;
constexpr int P1_WGS = 1024;  // workgroups aimed at (512: the same times; 2048: 13 % slower - scripts/wrw1x1_probe.py)
constexpr int P1_OP = P1_KC * P1_CS;   // one operand of one buffer
constexpr size_t P1_LDS = 4 * P1_OP * sizeof(float);

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct P1Dims {
    int B, KC, RC, HW;   // images, input channels (x), output channels (dy), pixels per image (a multiple of 4)
    int cpi, chunks, cpw, S, grouped;
};

__global__ __launch_bounds__(P1_NT) void wrw1x1_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ out, P1Dims d) {
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncb = d.RC / 64, nkb = d.KC / 64, ntile = ncb * nkb;
    int s, tile;
    if (d.grouped) {
        const int g = blockIdx.x & 7, k = blockIdx.x >> 3;
        s = g + 8 * (k / ntile), tile = k % ntile;
        if (s >= d.S) return;
    } else {
        s = blockIdx.x / ntile, tile = blockIdx.x - s * ntile;
    }
    const int co0 = (tile / nkb) * 64, ci0 = (tile % nkb) * 64;
    const int q0 = s * d.cpw, nq = d.chunks - q0 < d.cpw ? d.chunks - q0 : d.cpw;
    // a lane's four float4 of a chunk: number i = threadIdx.x + 256 j -> operand i >> 9 (0: dy, 1: x), channel row (i >> 3) & 63, float4 i & 7 of the 32 pixels
    const int row = (threadIdx.x >> 3) & 31, k4 = threadIdx.x & 7;
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 v0 = zero4, v1 = zero4, v2 = zero4, v3 = zero4;  // dy rows `row`, `row + 32`; x rows `row`, `row + 32`
    auto load = [&](int q) {
        const int b = q / d.cpi, p = (q - b * d.cpi) * P1_KC + 4 * k4;
        if (p < d.HW) {
            const float *yp = dy + (static_cast<size_t>(b) * d.RC + co0 + row) * d.HW + p;
            const float *xp = x + (static_cast<size_t>(b) * d.KC + ci0 + row) * d.HW + p;
            v0 = *reinterpret_cast<const float4 *>(yp);
            v1 = *reinterpret_cast<const float4 *>(yp + static_cast<size_t>(32) * d.HW);
            v2 = *reinterpret_cast<const float4 *>(xp);
            v3 = *reinterpret_cast<const float4 *>(xp + static_cast<size_t>(32) * d.HW);
        } else {
            v0 = v1 = v2 = v3 = zero4;
        }
    };
    auto store = [&](float *buf) {
        float *a = buf + 4 * k4 * P1_CS + row, *b = a + P1_OP;
        a[0] = v0.x, a[P1_CS] = v0.y, a[2 * P1_CS] = v0.z, a[3 * P1_CS] = v0.w;
        a[32] = v1.x, a[P1_CS + 32] = v1.y, a[2 * P1_CS + 32] = v1.z, a[3 * P1_CS + 32] = v1.w;
        b[0] = v2.x, b[P1_CS] = v2.y, b[2 * P1_CS] = v2.z, b[3 * P1_CS] = v2.w;
        b[32] = v3.x, b[P1_CS + 32] = v3.y, b[2 * P1_CS + 32] = v3.z, b[3 * P1_CS + 32] = v3.w;
    };
    const int m = wave >> 1, n = wave & 1, l31 = lane & 31, lh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    auto multiply = [&](const float *buf) {
        const float *ap = buf + lh * P1_CS + 32 * m + l31, *bp = buf + P1_OP + lh * P1_CS + 32 * n + l31;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks * P1_CS], bp[2 * ks * P1_CS], acc, 0, 0, 0);
    };
    float *buf0 = lds, *buf1 = lds + 2 * P1_OP;
    load(q0);
    store(buf0);
    __syncthreads();
    for (int r = 0; r < nq; ++r) {
        if (r + 1 < nq) load(q0 + r + 1);
        multiply((r & 1) ? buf1 : buf0);
        if (r + 1 < nq) store((r & 1) ? buf0 : buf1);
        __syncthreads();
    }
    // D[row = 8 (reg / 4) + 4 (lane / 32) + reg % 4][col = lane % 32] of the wavefront's 32 x 32 quarter
    float *o = out + static_cast<size_t>(s) * d.RC * d.KC + static_cast<size_t>(co0 + 32 * m) * d.KC + ci0 + 32 * n + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[static_cast<size_t>(8 * (r >> 2) + 4 * lh + (r & 3)) * d.KC] = acc[r];
}

void wrw1x1_plan(int B, int KC, int RC, int HW, P1Dims &d) {
    d.B = B, d.KC = KC, d.RC = RC, d.HW = HW;
    d.cpi = (HW + P1_KC - 1) / P1_KC;
    d.chunks = B * d.cpi;
    const int ntile = (KC / 64) * (RC / 64);
    const int target = ntile >= P1_WGS ? 1 : P1_WGS / ntile;  // about P1_WGS / 256 workgroups per CU
    d.cpw = (d.chunks + target - 1) / target;
    if (d.cpw < 8) d.cpw = d.chunks < 8 ? d.chunks : 8;   // ... but never so few chunks that the prologue and the partial sums dominate
    d.S = (d.chunks + d.cpw - 1) / d.cpw;
    d.grouped = d.S >= 8;
}

}  // namespace

EE_API int64_t ee_wrw3x3_workspace_floats(int B, int Cin, int Cout, int H) {
    if (B < 1 || Cin < 32 || Cout < 32 || Cin % 32 != 0 || Cout % 32 != 0 || (H != 2 && H != 4 && H != 8 && H != 16)) return 0;
    WrwDims d;
    wrw_plan(B, Cin, Cout, H, d);
    return d.S > 1 ? static_cast<int64_t>(d.S) * 9 * Cin * Cout : 0;
}

// dw [Cout][Cin][3][3] = d loss / d weight of y = conv3x3(x, weight) (stride 1, padding 1) given x [B][Cin][H][H] and dy [B][Cout][H][H];
// workspace: ee_wrw3x3_workspace_floats(...) floats, overwritten
EE_API int ee_wrw3x3_f32(const float *x, const float *dy, float *dw, float *workspace, int B, int Cin, int Cout, int H, void *stream) {
    if (B < 0 || Cin < 1 || Cout < 1) return EE_ERR_SHAPE;
    if (Cin % 32 != 0 || Cout % 32 != 0 || (H != 2 && H != 4 && H != 8 && H != 16)) return EE_ERR_UNSUPPORTED;
    if (!dw) return EE_ERR_NULL;
    if (B == 0) return static_cast<int>(hipMemsetAsync(dw, 0, sizeof(float) * 9 * Cin * Cout, as_stream(stream)));
    if (!x || !dy) return EE_ERR_NULL;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dw)) return EE_ERR_ALIGN;
    if (static_cast<int64_t>(B) * (Cin > Cout ? Cin : Cout) * H * H > 0x7fffffffLL) return EE_ERR_SHAPE;
    WrwDims d;
    wrw_plan(B, Cin, Cout, H, d);
    hipStream_t st = as_stream(stream);
    if (d.S > 1 && (!workspace || !aligned16(workspace))) return workspace ? EE_ERR_ALIGN : EE_ERR_NULL;
    float *out = d.S > 1 ? workspace : dw;  // one split: the kernel's 3x3 results ARE the gradient
    int rc;
    if (H == 16) rc = wrw_launch<16>(x, dy, out, d, st);
    else if (H == 8) rc = wrw_launch<8>(x, dy, out, d, st);
    else if (H == 4) rc = wrw_launch<4>(x, dy, out, d, st);
    else rc = wrw_launch<2>(x, dy, out, d, st);
    if (rc != EE_OK || d.S == 1) return rc;
    const int n4 = 9 * Cin * Cout / 4;
    EE_LAUNCH(wrw_sum_kernel, dim3((n4 + 63) / 64), dim3(256), 0, st, reinterpret_cast<const float4 *>(workspace), reinterpret_cast<float4 *>(dw), d.S, n4);
    return launch_status();
}

EE_API int64_t ee_wrw3x3s2_workspace_floats(int B, int Cin, int Cout, int H, int with_shortcut) {
    if (B < 1 || Cin < 32 || Cout < 32 || Cin % 32 != 0 || Cout % 32 != 0 || (H != 4 && H != 8 && H != 16)) return 0;
    WrwDims d;
    wrw_s2_plan(B, Cin, Cout, H, d);
    return d.S > 1 ? static_cast<int64_t>(d.S) * (with_shortcut ? 10 : 9) * Cin * Cout : 0;
}

// dw = [ d loss / d w3 [Cout][Cin][3][3] | d loss / d w1 [Cout][Cin] (only with dy1) ] of y3 = conv3x3(x, w3) (stride 2, padding 1) and
// y1 = conv1x1(x, w1) (stride 2): x [B][Cin][H][H], dy3 / dy1 [B][Cout][H/2][H/2] (dy1 NULL: the 3x3 alone); workspace as above
EE_API int ee_wrw3x3s2_f32(const float *x, const float *dy3, const float *dy1, float *dw, float *workspace, int B, int Cin, int Cout, int H, void *stream) {
    if (B < 0 || Cin < 1 || Cout < 1) return EE_ERR_SHAPE;
    if (Cin % 32 != 0 || Cout % 32 != 0 || (H != 4 && H != 8 && H != 16)) return EE_ERR_UNSUPPORTED;
    if (!dw) return EE_ERR_NULL;
    const int taps = dy1 ? 10 : 9;
    if (B == 0) return static_cast<int>(hipMemsetAsync(dw, 0, sizeof(float) * taps * Cin * Cout, as_stream(stream)));
    if (!x || !dy3) return EE_ERR_NULL;
    if (!aligned16(x) || !aligned16(dy3) || (dy1 && !aligned16(dy1)) || !aligned16(dw)) return EE_ERR_ALIGN;
    if (static_cast<int64_t>(B) * (Cin > Cout ? Cin : Cout) * H * H > 0x7fffffffLL) return EE_ERR_SHAPE;
    WrwDims d;
    wrw_s2_plan(B, Cin, Cout, H, d);
    hipStream_t st = as_stream(stream);
    if (d.S > 1 && (!workspace || !aligned16(workspace))) return workspace ? EE_ERR_ALIGN : EE_ERR_NULL;
    float *out = d.S > 1 ? workspace : dw;
    int rc;
    if (H == 16) rc = wrw_s2_launch<16>(x, dy3, dy1, out, d, st);
    else if (H == 8) rc = wrw_s2_launch<8>(x, dy3, dy1, out, d, st);
    else rc = wrw_s2_launch<4>(x, dy3, dy1, out, d, st);
    if (rc != EE_OK || d.S == 1) return rc;
    const int n4 = taps * Cin * Cout / 4;
    EE_LAUNCH(wrw_sum_kernel, dim3((n4 + 63) / 64), dim3(256), 0, st, reinterpret_cast<const float4 *>(workspace), reinterpret_cast<float4 *>(dw), d.S, n4);
    return launch_status();
}

EE_API int64_t ee_wrw_stem7x7s2_workspace_floats(int B, int H, int W) {
    if (B < 1 || H < 2 || W < 32 || H % 2 != 0 || W % 32 != 0) return 0;
    int chunks, cpw, S;
    wrw_stem_plan(B, H, W, chunks, cpw, S);
    return S > 1 ? static_cast<int64_t>(S) * 64 * ST_J : 0;
}

// dw [64][3][7][7] = d loss / d weight of the stem Conv2d(3, 64, 7, stride 2, padding 3, bias=False) (resnet.py:112): x [B][3][H][W],
// dy [B][64][H/2][W/2]; H even, W a multiple of 32 (else EE_ERR_UNSUPPORTED); workspace: ee_wrw_stem7x7s2_workspace_floats(B, H, W) floats
EE_API int ee_wrw_stem7x7s2_f32(const float *x, const float *dy, float *dw, float *workspace, int B, int H, int W, void *stream) {
    if (B < 0 || H < 2 || W < 2) return EE_ERR_SHAPE;
    if (H % 2 != 0 || W % 32 != 0) return EE_ERR_UNSUPPORTED;
    if (!dw) return EE_ERR_NULL;
    if (B == 0) return static_cast<int>(hipMemsetAsync(dw, 0, sizeof(float) * 64 * ST_J, as_stream(stream)));
    if (!x || !dy) return EE_ERR_NULL;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dw)) return EE_ERR_ALIGN;
    if (static_cast<int64_t>(B) * 64 * (H / 2) * (W / 2) > 0x7fffffffLL) return EE_ERR_SHAPE;
    int chunks, cpw, S;
    wrw_stem_plan(B, H, W, chunks, cpw, S);
    if (S > 1 && (!workspace || !aligned16(workspace))) return workspace ? EE_ERR_ALIGN : EE_ERR_NULL;
    hipStream_t st = as_stream(stream);
    constexpr size_t lds_bytes = (2 * ST_RAW > ST_MS ? 2 * ST_RAW : ST_MS) * sizeof(float);
    static_assert(lds_bytes <= 64 * 1024, "the stem kernel's LDS fits the default limit");
    EE_LAUNCH(wrw_stem_kernel, dim3(static_cast<unsigned>(S)), dim3(WR_NT), lds_bytes, st, x, dy, S > 1 ? workspace : dw, B, H, W, chunks, cpw);
    int rc = launch_status();
    if (rc != EE_OK || S == 1) return rc;
    const int n4 = 64 * ST_J / 4;
    EE_LAUNCH(wrw_sum_kernel, dim3((n4 + 63) / 64), dim3(256), 0, st, reinterpret_cast<const float4 *>(workspace), reinterpret_cast<float4 *>(dw), S, n4);
    return launch_status();
}

EE_API int64_t ee_wrw1x1_workspace_floats(int B, int Cin, int Cout, int HW) {
    if (B < 1 || Cin < 64 || Cout < 64 || Cin % 64 != 0 || Cout % 64 != 0 || HW < 4 || HW % 4 != 0) return 0;
    P1Dims d;
    wrw1x1_plan(B, Cin, Cout, HW, d);
    return d.S > 1 ? static_cast<int64_t>(d.S) * Cin * Cout : 0;
}

// dw [Cout][Cin] = d loss / d weight of y = conv1x1(x, weight) (stride 1, no padding): x [B][Cin][HW], dy [B][Cout][HW] (HW = H * W pixels, a
// multiple of 4); Cin, Cout multiples of 64 (else EE_ERR_UNSUPPORTED); workspace: ee_wrw1x1_workspace_floats(...) floats
EE_API int ee_wrw1x1_f32(const float *x, const float *dy, float *dw, float *workspace, int B, int Cin, int Cout, int HW, void *stream) {
    if (B < 0 || Cin < 1 || Cout < 1 || HW < 1) return EE_ERR_SHAPE;
    if (Cin % 64 != 0 || Cout % 64 != 0 || HW % 4 != 0) return EE_ERR_UNSUPPORTED;
    if (!dw) return EE_ERR_NULL;
    if (B == 0) return static_cast<int>(hipMemsetAsync(dw, 0, sizeof(float) * Cin * Cout, as_stream(stream)));
    if (!x || !dy) return EE_ERR_NULL;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dw)) return EE_ERR_ALIGN;
    if (static_cast<int64_t>(B) * (Cin > Cout ? Cin : Cout) * HW > 0x7fffffffLL) return EE_ERR_SHAPE;
    P1Dims d;
    wrw1x1_plan(B, Cin, Cout, HW, d);
    if (d.S > 1 && (!workspace || !aligned16(workspace))) return workspace ? EE_ERR_ALIGN : EE_ERR_NULL;
    hipStream_t st = as_stream(stream);
    const int ntile = (Cin / 64) * (Cout / 64);
    const unsigned grid = static_cast<unsigned>(d.grouped ? 8 * ((d.S + 7) / 8) * ntile : d.S * ntile);
    EE_LAUNCH(wrw1x1_kernel, dim3(grid), dim3(P1_NT), P1_LDS, st, x, dy, d.S > 1 ? workspace : dw, d);
    int rc = launch_status();
    if (rc != EE_OK || d.S == 1) return rc;
    const int n4 = Cin * Cout / 4;
    EE_LAUNCH(wrw_sum_kernel, dim3((n4 + 63) / 64), dim3(256), 0, st, reinterpret_cast<const float4 *>(workspace), reinterpret_cast<float4 *>(dw), d.S, n4);
    return launch_status();
}
