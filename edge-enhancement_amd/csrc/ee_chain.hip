// ee_chain.hip - the whole EE front end of one PGD iteration in TWO launches (gfx950).
//
//   forward   x_in = clamp( hfs(add_square(x)) + w * edge(x), 0, 1 )            resnet_EE_square.py:187-206, Net2_EE_square.py:48-63
//             draws (core.py:637,645,648) -> Add_Square (core.py:636-655) -> HighFreqSuppress (core.py:15-55)
//             -> CannyFilter_step125_1 (core.py:549-585) -> combine + clamp gate
//   backward  x   = clamp(min(max(x + dir*alpha*sign(g), x0 - eps), x0 + eps), lo, hi),                         attacks.py:25-27
//             g   = dsquare * hfs(gate * g_in) + edge_adjoint(w * sum_c gate_c * g_in_c)
// where round 1 used seven launches (square_draw, hfs<1>, edge_fwd, | edge_bwd_saved, hfs<2>, pgd_step_bcast) and moved
// x_lp, g_hfs, g_lp, g_edge, stripe through HBM.  Here ONE WORKGROUP OWNS ONE IMAGE (C planes of H x W <= 64 x 64: 48 KB):
// x is read once; x_in, the gate byte and the two saved Sobel responses are written once; in the backward g_in, gate, x, x0
// are read once and x is written once - nothing else touches memory.
//
// Work split inside a workgroup of (4 + C) wavefronts:
//   waves 0..3      ("stencil" waves, VALU): global <-> LDS staging, the edge filter and its adjoint with the exact arithmetic of
//                   ee_edge.hip (same helpers: bit-identical edge bits, Sobel responses and edge-branch gradients), the update
//   waves 4..4+C-1  ("matrix" waves): one per colour plane, the low-pass operator as FOUR CHAINED v_mfma_f32_16x16x4_f32 products
//                   whose intermediate tiles never leave the accumulators (scripts/chain_emulate.py derives and checks the index
//                   algebra): P,Q = X T1 -> R = CS^T [P|Q] -> (lane^8 exchange) -> UV^T = EF^T CS^T -> y = UV T4.  The constant
//                   factors sit in LDS in fragment order (K-permuted to match the accumulator layout); 192 MFMAs per 64x64 plane.
//   The matrix pipe and the vector pipe of a SIMD run concurrently, so the two halves overlap.
//
// Numerics: everything the edge filter decides (edge bits, gx1 / gy1, g_edge incl. NaNs, the gate bit arithmetic, the update
// formula) is bit-identical to the separate kernels; the low-pass values differ from ee_hfs.hip in summation order only
// (f32 MFMA is exact f32 multiply-add; 3e-7 against the float64 operator, same tolerance class as ee_hfs.hip).
//
// gate byte written by the forward: bit 0 = 1[0 <= x_lp + w*e <= 1] (clamp passes the gradient), bits 1-2 = d add_square / dx
// in {0, 1/2, 3/4, 1} coded 0..3 (n_queries <= 1; 3 when the model has no Add_Square) - the backward needs neither x nor the draws.
#include <math.h>
#include <stdlib.h>

#include "ee_common.hpp"
#include "ee_square.hpp"
#include "ee_stencil.hpp"

namespace {

using namespace ee;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// -DEE_CHAIN_SKIP=<bits> (scripts/chain_phases.py builds its own copy of this file with it; never the product): phase skipping in the
// forward kernel - 1: no edge filter, 2: no low-pass products, 4: no combine / store.  Results are garbage, the timings tell the phases apart.
#ifndef EE_CHAIN_SKIP
#define EE_CHAIN_SKIP 0
#endif

constexpr int kSW = 8;             // stencil waves: two per SIMD, the edge loop is latency-bound with one (34 -> 2x fewer rounds)
constexpr int kST = kSW * kWave;   // 512 stencil threads (the helpers of ee_stencil.hpp assume 4-pixel groups, not a block size)

template <int H, int W>
struct Dims {
    static constexpr int HP = (H + 15) / 16 * 16, WP = (W + 15) / 16 * 16, HT = HP / 16, WT = WP / 16;
    static constexpr int N1 = WP / 4, N2 = 2 * HT * 4, N3 = HT * 2 * 4, N4 = WT * 4;
    static constexpr int NTAB = (N1 + N2 + N3 + N4) * 64;  // floats
    static constexpr int AS = WP + 4;                       // row stride of an MFMA A-operand plane: bank = (4 i + g) % 64, conflict-free
};

// ---- the low-pass operator on one plane, one wavefront, in two halves (a workgroup barrier may sit between them) ---------------
// a_at(h, w): the plane's value (h < HP, w < WP; anything FINITE outside the image: the tables are zero there); every A operand
// is consumed by lowpass_front.  lowpass_back: y[ht][wt][r] = result at (h = 16 ht + 4 (lane >> 4) + r, w = 16 wt + (lane & 15)).
template <int H, int W, class AFetch>
__device__ __forceinline__ void lowpass_front(const float *__restrict__ tab, AFetch a_at, f32x4 (&ef)[2]) {
    using D = Dims<H, W>;
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const float *t1 = tab + lane, *t2 = t1 + D::N1 * 64;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 pq[D::HT];
#pragma unroll
    for (int mt = 0; mt < D::HT; ++mt) pq[mt] = zero;
#pragma unroll 4
    for (int s = 0; s < D::N1; ++s) {  // P | Q = X T1: four independent accumulators per k-step
        const float b = t1[s * 64];
#pragma unroll
        for (int mt = 0; mt < D::HT; ++mt) pq[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_at(16 * mt + li, 4 * s + lg), b, pq[mt], 0, 0, 0);
    }
    f32x4 rc[2] = {zero, zero};
#pragma unroll
    for (int t = 0; t < D::HT; ++t)  // R = CS^T [P | Q]: the PQ accumulators are the B operands as they lie (K-permuted table)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                rc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(t2[((mt * D::HT + t) * 4 + r) * 64], pq[t][r], rc[mt], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float pc = __shfl_xor(rc[0][r], 8), ps = __shfl_xor(rc[1][r], 8);
        ef[0][r] = (li < 8) ? rc[0][r] - ps : rc[0][r] + ps;  // E1 = a - d | F1 = c + b
        ef[1][r] = (li < 8) ? rc[1][r] + pc : rc[1][r] - pc;  // E2 = b + c | F2 = d - a
    }
}

template <int H, int W>
__device__ __forceinline__ void lowpass_back(const float *__restrict__ tab, const f32x4 (&ef)[2], f32x4 (&y)[Dims<H, W>::HT][Dims<H, W>::WT]) {
    using D = Dims<H, W>;
    const int lane = threadIdx.x & 63;
    const float *t3 = tab + lane + (D::N1 + D::N2) * 64, *t4 = t3 + D::N3 * 64;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 uv[D::HT];
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht) uv[ht] = zero;
#pragma unroll
    for (int t = 0; t < 2; ++t)  // UV^T = EF^T CS^T / H: the EF registers are the A operands as they lie
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ht = 0; ht < D::HT; ++ht)
                uv[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(ef[t][r], t3[((ht * 2 + t) * 4 + r) * 64], uv[ht], 0, 0, 0);
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht)
#pragma unroll
        for (int wt = 0; wt < D::WT; ++wt) y[ht][wt] = zero;
#pragma unroll
    for (int r = 0; r < 4; ++r)  // y = UV T4: the UV^T registers are the A operands as they lie
#pragma unroll
        for (int wt = 0; wt < D::WT; ++wt) {
            const float b = t4[(wt * 4 + r) * 64];
#pragma unroll
            for (int ht = 0; ht < D::HT; ++ht) y[ht][wt] = __builtin_amdgcn_mfma_f32_16x16x4f32(uv[ht][r], b, y[ht][wt], 0, 0, 0);
        }
}

// d add_square / dx in {0, 1/2, 3/4, 1} <-> 2-bit code
__device__ __forceinline__ unsigned dsq_code(float d) { return static_cast<unsigned>(d * 4.0f) - (d >= 0.5f ? 1u : 0u); }
__device__ __forceinline__ float dsq_value(unsigned code) { return code == 0u ? 0.0f : static_cast<float>(code + 1u) * 0.25f; }

struct FwdParams {
    const float *x;
    float *x_in;
    uint8_t *gate;
    float *gx, *gy, *edge;
    const float *tables;
    const float *stripe_in;       // injected draws (tests): [B,C,1,W] / [1] / [1,C]; NULL = draw on the device
    const int64_t *sq_pos_in;
    const float *sq_sign_in;
    unsigned long long *state;    // {seed, offset, ticket, -}: Philox state of the device-side draws, advanced by the last workgroup
    int B, sq_size;
    float eps, two_eps, alpha, high, w;
};

template <int C, int H, int W, bool SQUARE>
__global__ __launch_bounds__((kSW + C) * kWave) void chain_fwd_kernel(FwdParams p, Weights wt) {
    using D = Dims<H, W>;
    constexpr int FH = H + 4, FW = W + 2 * kColHalo, F4 = FW / 4, PL = FH * FW, W4 = W / 4;
    static_assert(W % 4 == 0 && H <= 64 && W <= 64 && C <= 3, "shape class of the reference configs");
    extern __shared__ __align__(16) float lds[];
    float *xr = lds;                       // [C][FH][FW] clamped frame of x (rows -2..H+1, cols -4..W+3): the edge filter's input
    float *xs = xr + C * PL;               // [C][HP][AS] add_square(x): the MFMA A operand, later the low-pass result in place
    float *tab = xs + C * D::HP * D::AS;   // MFMA constant fragments
    float *stripe = tab + D::NTAB;         // [C][W] stripe signs of this image
    uint8_t *emap = reinterpret_cast<uint8_t *>(stripe + C * W + 4);  // [H][W] edge map, one byte per pixel (4 floats of square draws before it)
    uint8_t *gst = emap + H * W;                                  // [C][H][W] derivative codes (bits 1-2 of the gate byte)
    const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const float *xn = p.x + static_cast<size_t>(n) * C * H * W;

    // ---- x frame: every global load of the stencil waves first, on clamped addresses --------------------------------------------------
    constexpr int TOTAL = C * FH * F4, PER = (TOTAL + kST - 1) / kST;
    float4 v[PER];
    int gjs[PER];
    if (wave < kSW) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx0 = tid + q * kST;
            const int idx = idx0 < TOTAL ? idx0 : TOTAL - 1;
            const int c = idx / (FH * F4), rem = idx - c * (FH * F4);
            const int r = rem / F4, f = rem - r * F4;
            const int gi = clampi(r - 2, 0, H - 1);
            gjs[q] = 4 * f - kColHalo;
            v[q] = *reinterpret_cast<const float4 *>(xn + (static_cast<size_t>(c) * H + gi) * W + clamp_col4(gjs[q], W));
        }
    }

    // ---- draws (core.py:637, :645, :648) by wave 0: the same element <-> Philox mapping as square_draw_kernel (ee_square.hip).
    // stripe[C][W] signs and {vh, 2 eps sign_c} go to LDS.  Device-side draws: every lane of wave 0 reads {seed, offset}, then
    // lane 0 takes a ticket; the workgroup holding the last ticket advances the offset - every workgroup has read it by then
    // (its ticket came after its reads), and the latency of the atomic hides under the rest of the kernel ------------------------
    float *sqp = stripe + C * W;  // [0] = vh (as float bits), [1 + c] = 2 eps sign_c
    const bool rng_mode = SQUARE && !p.stripe_in;
    unsigned long long rng_base = 0ull, ticket = 0ull;
    if (SQUARE && wave == 0) {
        if (p.stripe_in) {
            for (int i = tid; i < C * W; i += kWave) stripe[i] = p.stripe_in[static_cast<size_t>(n) * C * W + i];
            if (tid == 0) sqp[0] = __int_as_float(static_cast<int>(p.sq_pos_in[0]));
            if (tid < C) sqp[1 + tid] = p.two_eps * p.sq_sign_in[tid];
        } else {
            const unsigned long long seed = p.state[0], base = p.state[1];
            const Philox rng(seed);
            const long long n_stripe = static_cast<long long>(p.B) * C * W;
            if (tid < C * W4) {  // C * W / 4 counters per image (W % 4 == 0: an image's stripe starts on a counter boundary)
                const uint4 r = rng(base + static_cast<unsigned long long>(n) * (C * W4) + tid);
                *reinterpret_cast<float4 *>(stripe + 4 * tid) =
                    make_float4(sgn(2.0f * u01(r.x) - 1.0f), sgn(2.0f * u01(r.y) - 1.0f), sgn(2.0f * u01(r.z) - 1.0f), sgn(2.0f * u01(r.w) - 1.0f));
            } else if (tid < C * W4 + 1 + C) {  // the square's offset and per-channel signs, shared by the batch: elements n_stripe + k
                const int k = tid - C * W4;
                const long long e = n_stripe + k;
                const uint4 r = rng(base + static_cast<unsigned long long>(e >> 2));
                const unsigned rr[4] = {r.x, r.y, r.z, r.w};
                const float u = u01(rr[e & 3]);
                if (k == 0) {
                    const float span = static_cast<float>(H) - static_cast<float>(p.sq_size);
                    sqp[0] = __int_as_float(static_cast<int>(static_cast<long long>(0.0f + (span - 0.0f) * u)));
                } else {
                    sqp[k] = p.two_eps * sgn(2.0f * u - 1.0f);
                }
            }
            rng_base = base;
        }
    }
    static_assert(C * (W / 4) + 1 + C <= kWave, "one wavefront makes all draws of an image");
    int vh = 0;
    float sq_delta[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sq_delta[c] = 0.0f;
    if (SQUARE) {
        __syncthreads();  // the draws are in LDS (and this wave's reads of the state have returned: the barrier waits for them)
        vh = __float_as_int(sqp[0]);
#pragma unroll
        for (int c = 0; c < C; ++c) sq_delta[c] = sqp[1 + c];
        // the ticket: issued HERE, its value read at the end of the kernel, so that the atomic's round trip hides behind the edge filter (round
        // 4; taken before the barrier it held the whole workgroup for 2-3 us).  No fence: every read of the state precedes its workgroup's
        // ticket, the tickets are device-scope atomics, and the new offset only has to be visible to the NEXT launch (an agent-scope fence
        // would write the XCD's dirty L2 lines back)
        if (rng_mode && tid == 0) ticket = atomicAdd(p.state + 2, 1ull);
    }

    // ---- stage in: frame -> LDS; its interior also through Add_Square (value -> A plane, derivative code -> gate byte) ----------------
    if (wave < kSW) {
        SquareArgs sa{};
        sa.nq = 1; sa.C = C; sa.H = H; sa.W = W; sa.eps = p.eps; sa.two_eps = p.two_eps;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx = tid + q * kST;
            if (idx < TOTAL) {
                *reinterpret_cast<float4 *>(xr + idx * 4) = replicate4(v[q], gjs[q], W);
                const int c = idx / (FH * F4), rem = idx - c * (FH * F4);
                const int r = rem / F4, f = rem - r * F4;
                const int h = r - 2, w0 = 4 * f - kColHalo;
                if (h >= 0 && h < H && w0 >= 0 && w0 < W) {
                    const float xv[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
                    float o[4];
                    unsigned code[4];
                    if (SQUARE) {
                        SquarePlane pl{};
                        pl.vh[0] = vh; pl.s[0] = p.sq_size;
#pragma unroll
                        for (int cc = 0; cc < C; ++cc)
                            if (cc == c) pl.delta[0] = sq_delta[cc];
                        const float4 st = *reinterpret_cast<const float4 *>(stripe + c * W + w0);
                        const float sv[4] = {st.x, st.y, st.z, st.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float d;
                            o[k] = square_elem<true>(sa, pl, xv[k], sv[k], c, h, w0 + k, d);
                            code[k] = dsq_code(d) << 1;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            o[k] = xv[k];
                            code[k] = 3u << 1;
                        }
                    }
                    *reinterpret_cast<float4 *>(xs + (c * D::HP + h) * D::AS + w0) = make_float4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<uchar4 *>(gst + (c * H + h) * W + w0) = make_uchar4(code[0], code[1], code[2], code[3]);
                }
            }
        }
        if (D::HP != H || D::WP != W) {  // padding rows / columns of the A planes (28 -> 32): zeros (their table entries are zero too)
            for (int idx = tid; idx < C * D::HP * D::AS; idx += kST) {
                const int rem = idx % (D::HP * D::AS);
                const int h = rem / D::AS, w = rem - h * D::AS;
                if (h >= H || w >= W) xs[idx] = 0.0f;
            }
        }
    } else {
        constexpr int MT = C * kWave, NT4 = D::NTAB / 4, PERT = (NT4 + MT - 1) / MT;
        const int mt_id = tid - kST;
        float4 tv[PERT];
#pragma unroll
        for (int q = 0; q < PERT; ++q) {
            const int i = mt_id + q * MT;
            tv[q] = reinterpret_cast<const float4 *>(p.tables)[i < NT4 ? i : NT4 - 1];
        }
#pragma unroll
        for (int q = 0; q < PERT; ++q) {
            const int i = mt_id + q * MT;
            if (i < NT4) reinterpret_cast<float4 *>(tab)[i] = tv[q];
        }
    }
    __syncthreads();

    if (wave < kSW) {
        // ---- edge filter on 4-pixel groups: ee_edge.hip's arithmetic, whole image in the frame ----------------------------------
        for (int idx = tid; idx < ((EE_CHAIN_SKIP & 1) ? 0 : H * W4); idx += kST) {
            const int i = idx / W4, lx = idx - i * W4, jb = 4 * lx;
            float b[C][3][6];
            blur_group<C, FH, FW>(xr, wt, i, jb + kColHalo - 2, i, jb, H, W, b);
            float e[4], gxs[4], gys[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float ax, ay, s2, mag, mag_a;
                sobel_px<C>(b, wt, k, ax, ay);
                edge_from_sums<C>(ax, ay, p.alpha, p.high, gxs[k], gys[k], s2, mag, mag_a, e[k]);
            }
            // e is 0 or 1 (To_compare.forward of a finite magnitude; NaN cannot occur: x is finite): one byte per pixel
            *reinterpret_cast<uchar4 *>(emap + i * W + jb) = make_uchar4(e[0] != 0.0f, e[1] != 0.0f, e[2] != 0.0f, e[3] != 0.0f);
            const size_t pix = (static_cast<size_t>(n) * H + i) * W + jb;
            *reinterpret_cast<float4 *>(p.gx + pix) = make_float4(gxs[0], gxs[1], gxs[2], gxs[3]);
            *reinterpret_cast<float4 *>(p.gy + pix) = make_float4(gys[0], gys[1], gys[2], gys[3]);
            if (p.edge) *reinterpret_cast<float4 *>(p.edge + pix) = make_float4(e[0], e[1], e[2], e[3]);
        }
    } else {
        // ---- low-pass of add_square(x), plane c = wave - kSW; the result replaces the plane (this wave is its only reader) -----------
        const int c = wave - kSW, lane = tid & 63, li = lane & 15, lg = lane >> 4;
        float *xc = xs + c * D::HP * D::AS;
        f32x4 y[D::HT][D::WT], ef[2];
        if (!(EE_CHAIN_SKIP & 2)) {
        lowpass_front<H, W>(tab, [&](int h, int w) -> float { return xc[h * D::AS + w]; }, ef);
        lowpass_back<H, W>(tab, ef, y);
#pragma unroll
        for (int ht = 0; ht < D::HT; ++ht)
#pragma unroll
            for (int wt_ = 0; wt_ < D::WT; ++wt_)
#pragma unroll
                for (int r = 0; r < 4; ++r) xc[(16 * ht + 4 * lg + r) * D::AS + 16 * wt_ + li] = y[ht][wt_][r];
        }
    }
    __syncthreads();
    if (EE_CHAIN_SKIP & 4) return;

    // ---- combine + stage out (everybody): x_in = clamp(x_lp + w * e), gate = code | 1[0 <= x_lp + w * e <= 1]; 16 B / 4 B per lane --------
    constexpr int NT = (kSW + C) * kWave;
    float *xo = p.x_in + static_cast<size_t>(n) * C * H * W;
    uint8_t *go = p.gate + static_cast<size_t>(n) * C * H * W;
    for (int idx = tid; idx < C * H * W4; idx += NT) {
        const int c = idx / (H * W4), rem = idx - c * (H * W4);
        const int h = rem / W4, w0 = 4 * (rem - h * W4);
        const float4 lp = *reinterpret_cast<const float4 *>(xs + (c * D::HP + h) * D::AS + w0);
        const uchar4 e4 = *reinterpret_cast<const uchar4 *>(emap + h * W + w0);
        const uchar4 cd = *reinterpret_cast<const uchar4 *>(gst + (c * H + h) * W + w0);
        const float s0 = lp.x + p.w * (e4.x ? 1.0f : 0.0f), s1 = lp.y + p.w * (e4.y ? 1.0f : 0.0f);
        const float s2 = lp.z + p.w * (e4.z ? 1.0f : 0.0f), s3 = lp.w + p.w * (e4.w ? 1.0f : 0.0f);
        *reinterpret_cast<float4 *>(xo + (static_cast<size_t>(c) * H + h) * W + w0) =
            make_float4(tclamp(s0, 0.0f, 1.0f), tclamp(s1, 0.0f, 1.0f), tclamp(s2, 0.0f, 1.0f), tclamp(s3, 0.0f, 1.0f));
        *reinterpret_cast<uchar4 *>(go + (static_cast<size_t>(c) * H + h) * W + w0) =
            make_uchar4(cd.x | (s0 >= 0.0f && s0 <= 1.0f), cd.y | (s1 >= 0.0f && s1 <= 1.0f), cd.z | (s2 >= 0.0f && s2 <= 1.0f),
                        cd.w | (s3 >= 0.0f && s3 <= 1.0f));
    }
    if (rng_mode && tid == 0 && ticket + 1ull == static_cast<unsigned long long>(p.B)) {  // the last ticket: every workgroup has read the state
        const long long total = static_cast<long long>(p.B) * C * W + 1 + C;
        p.state[1] = rng_base + static_cast<unsigned long long>((total + 3) >> 2);
        p.state[2] = 0ull;
    }
}

// Round 4 priced this kernel (scripts/chain_phases.py, profiles/round4_a_chain_fwd_phases.txt) and a row-band variant of it (one workgroup
// per image and 16 rows, two per CU, 400 at batch 100; commit 614de05, profiles/round4_c_chain_fwd_bands.txt): of the 23 us at 100 x 3 x 64 x 64,
// 9.4 are launch / loads / draws / Add_Square / staging, 8.4 the edge filter (VALU-bound: ~770 K lane-operations per image = 5 us of a CU's
// vector pipes at full rate), 3.7 the low-pass products (hidden behind it), 2.8 combine / store.  Bands spread the edge filter over more CUs
// but every band workgroup needs add_square(x) of EVERY row as low-pass input: 23.1 us against 22.6 with Add_Square (14.5 against 16.2
// without), 210 against 140 us at batch 1600.  Removed; what stayed is the draw-state ticket issued behind the barrier (-0.5 us).

struct BwdParams {
    const float *g_in;
    const uint8_t *gate;
    const float *gx, *gy;
    float *x;
    const float *x0;
    const float *tables;
    float alpha, high, w;       // edge filter / front end
    float step, eps, lo, hi;    // update (step carries the direction's sign)
};

template <int C, int H, int W>
__global__ __launch_bounds__((kSW + C) * kWave) void chain_bwd_kernel(BwdParams p, Weights wt) {
    using D = Dims<H, W>;
    constexpr int FH = H + 8, FW = W + 2 * kColHalo, PL = FH * FW, W4 = W / 4, OI = -4, OJ = -kColHalo;
    constexpr int NPOS = H * W4, PER = (NPOS + kST - 1) / kST;  // 4-pixel groups of the image, PER per stencil thread
    static_assert(W % 4 == 0 && H <= 64 && W <= 64 && C <= 3, "shape class of the reference configs");
    extern __shared__ __align__(16) float lds[];
    float *ggx = lds, *ggy = ggx + PL, *gb = ggy + PL;  // frames with origin at image pixel (-4, -4), zero outside the image
    float *gh = gb + PL;                                // [C][HP][AS]: gate * g_in (the MFMA A operand), later the low-pass result
    float *tab = gh + C * D::HP * D::AS;
    const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const size_t img = static_cast<size_t>(n) * C * H * W, map = static_cast<size_t>(n) * H * W;

    float4 xv[PER][C], x0v[PER][C];
    uchar4 gtv[PER][C];
    if (wave < kSW) {
        // ---- every global load of the stencil waves, back to back on clamped group indices -----------------------------------------------
        float4 gin[PER][C], gxv[PER], gyv[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx0 = tid + q * kST, idx = idx0 < NPOS ? idx0 : NPOS - 1;
            const int i = idx / W4, jb = 4 * (idx - i * W4);
            gxv[q] = *reinterpret_cast<const float4 *>(p.gx + map + i * W + jb);
            gyv[q] = *reinterpret_cast<const float4 *>(p.gy + map + i * W + jb);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const size_t o = img + (static_cast<size_t>(c) * H + i) * W + jb;
                gin[q][c] = *reinterpret_cast<const float4 *>(p.g_in + o);
                gtv[q][c] = *reinterpret_cast<const uchar4 *>(p.gate + o);
            }
        }
        // zero the halo ring of the three frames (the interior is written below / in stage 3 by whoever owns the cell)
        for (int idx = tid; idx < PL; idx += kST) {
            const int r = idx / FW, s = idx - r * FW;
            if (r < 4 || r >= 4 + H || s < kColHalo || s >= kColHalo + W) {
                ggx[idx] = 0.0f;
                ggy[idx] = 0.0f;
                gb[idx] = 0.0f;
            }
        }
        // padding rows / columns of the A planes (28 -> 32): zeros
        if (D::HP != H || D::WP != W) {
            for (int idx = tid; idx < C * D::HP * D::AS; idx += kST) {
                const int rem = idx % (D::HP * D::AS);
                const int h = rem / D::AS, w = rem - h * D::AS;
                if (h >= H || w >= W) gh[idx] = 0.0f;
            }
        }
        // ---- g_hfs = gate ? g_in : 0 -> A planes; u = w * sum_c g_hfs_c; stage 2 of the edge adjoint (ee_edge.hip) on the saved responses -----
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx = tid + q * kST;
            if (idx < NPOS) {
                const int i = idx / W4, jb = 4 * (idx - i * W4);
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float v[4] = {(gtv[q][c].x & 1) ? gin[q][c].x : 0.0f, (gtv[q][c].y & 1) ? gin[q][c].y : 0.0f,
                                        (gtv[q][c].z & 1) ? gin[q][c].z : 0.0f, (gtv[q][c].w & 1) ? gin[q][c].w : 0.0f};
                    *reinterpret_cast<float4 *>(gh + (c * D::HP + i) * D::AS + jb) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = (c == 0) ? v[k] : acc[k] + v[k];
                }
                const float gxa[4] = {gxv[q].x, gxv[q].y, gxv[q].z, gxv[q].w}, gya[4] = {gyv[q].x, gyv[q].y, gyv[q].z, gyv[q].w};
                float ox[4], oy[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float gx1 = gxa[k], gy1 = gya[k];
                    const float s2 = gx1 * gx1 + gy1 * gy1;
                    const float mag = sqrtf(s2);
                    const float mag_a = (mag < p.alpha) ? 0.0f : mag;
                    float gm = acc[k] * p.w;
                    if (mag_a <= p.high) gm = 0.0f;     // To_compare.backward core.py:356
                    if (mag_a > 1.001f) gm = 0.0f;      // core.py:357
                    if (mag < p.alpha) gm = 0.0f;       // where() backward core.py:575
                    const float rs = 1.0f / sqrtf(s2);  // pow(s2, -0.5): 0 -> inf
                    const float gs = gm * (0.5f * rs);  // 0*inf = NaN kept
                    ox[k] = (gs * (2.0f * gx1)) / static_cast<float>(C);
                    oy[k] = (gs * (2.0f * gy1)) / static_cast<float>(C);
                }
                *reinterpret_cast<float4 *>(ggx + (i - OI) * FW + jb - OJ) = make_float4(ox[0], ox[1], ox[2], ox[3]);
                *reinterpret_cast<float4 *>(ggy + (i - OI) * FW + jb - OJ) = make_float4(oy[0], oy[1], oy[2], oy[3]);
            }
            __builtin_amdgcn_sched_barrier(0);  // one group at a time: interleaving the IEEE divisions of all groups spills registers
        }
    } else {
        constexpr int MT = C * kWave, NT4 = D::NTAB / 4, PERT = (NT4 + MT - 1) / MT;
        const int mt_id = tid - kST;
        float4 v[PERT];
#pragma unroll
        for (int q = 0; q < PERT; ++q) {
            const int i = mt_id + q * MT;
            v[q] = reinterpret_cast<const float4 *>(p.tables)[i < NT4 ? i : NT4 - 1];
        }
#pragma unroll
        for (int q = 0; q < PERT; ++q) {
            const int i = mt_id + q * MT;
            if (i < NT4) reinterpret_cast<float4 *>(tab)[i] = v[q];
        }
    }
    __syncthreads();  // A

    if (wave < kSW) {
        // ---- stage 3: gb = pad^T(Sx^T ggx + Sy^T ggy) ------------------------------------------------------------------------------------
#pragma unroll 1
        for (int idx = tid; idx < NPOS; idx += kST) {
            const int i = idx / W4, jb = 4 * (idx - i * W4);
            float cells[3][6];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const bool need = (a == 1) || (a == 0 && i == 0) || (a == 2 && i == H - 1);
#pragma unroll
                for (int k = 0; k < 6; ++k) cells[a][k] = 0.0f;
                if (need) {
                    cells_row<FW>(ggx, wt.sx, i + a, jb, OI, OJ, cells[a]);
                    cells_row<FW>(ggy, wt.sy, i + a, jb, OI, OJ, cells[a]);
                }
            }
            float o[4];
            fold4(cells, i, jb, H, W, o);
            *reinterpret_cast<float4 *>(gb + (i - OI) * FW + jb - OJ) = make_float4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();  // B
        // the update's operands: issued before stage 4 (stage 3 needs the registers), consumed after barrier C
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx0 = tid + q * kST, idx = idx0 < NPOS ? idx0 : NPOS - 1;
            const int i = idx / W4, jb = 4 * (idx - i * W4);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const size_t o = img + (static_cast<size_t>(c) * H + i) * W + jb;
                xv[q][c] = *reinterpret_cast<const float4 *>(p.x + o);
                x0v[q][c] = *reinterpret_cast<const float4 *>(p.x0 + o);
            }
        }
        // ---- stage 4: g_edge = pad^T(G^T gb), parked in the (now dead) ggx frame at this thread's own groups -----------------------------
#pragma unroll 1
        for (int idx = tid; idx < NPOS; idx += kST) {
            const int i = idx / W4, jb = 4 * (idx - i * W4);
            float cells[3][6];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const bool need = (a == 1) || (a == 0 && i == 0) || (a == 2 && i == H - 1);
#pragma unroll
                for (int k = 0; k < 6; ++k) cells[a][k] = 0.0f;
                if (need) cells_row<FW>(gb, wt.g, i + a, jb, OI, OJ, cells[a]);
            }
            float o[4];
            fold4(cells, i, jb, H, W, o);
            *reinterpret_cast<float4 *>(ggx + (i - OI) * FW + jb - OJ) = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else {
        // ---- low-pass of g_hfs, plane c = wave - 4; the result replaces the plane ----------------------------------------------------------
        const int c = wave - kSW, lane = tid & 63, li = lane & 15, lg = lane >> 4;
        float *gc = gh + c * D::HP * D::AS;
        f32x4 y[D::HT][D::WT], ef[2];
        lowpass_front<H, W>(tab, [&](int h, int w) -> float { return gc[h * D::AS + w]; }, ef);
        __syncthreads();  // B - in the middle of the chain, so that neither side waits long for the other
        lowpass_back<H, W>(tab, ef, y);
#pragma unroll
        for (int ht = 0; ht < D::HT; ++ht)
#pragma unroll
            for (int wt_ = 0; wt_ < D::WT; ++wt_)
#pragma unroll
                for (int r = 0; r < 4; ++r) gc[(16 * ht + 4 * lg + r) * D::AS + 16 * wt_ + li] = y[ht][wt_][r];
    }
    __syncthreads();  // C

    // ---- update (stencil waves): g = dsquare * lowpass + g_edge;  attacks.py:25-27 ----------------------------------------------------------
    if (wave < kSW) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx = tid + q * kST;
            if (idx < NPOS) {
                const int i = idx / W4, jb = 4 * (idx - i * W4);
                const float4 ge4 = *reinterpret_cast<const float4 *>(ggx + (i - OI) * FW + jb - OJ);
                const float ge[4] = {ge4.x, ge4.y, ge4.z, ge4.w};
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float4 lp = *reinterpret_cast<const float4 *>(gh + (c * D::HP + i) * D::AS + jb);
                    const float l4[4] = {lp.x, lp.y, lp.z, lp.w};
                    const unsigned cd[4] = {gtv[q][c].x, gtv[q][c].y, gtv[q][c].z, gtv[q][c].w};
                    const float xa[4] = {xv[q][c].x, xv[q][c].y, xv[q][c].z, xv[q][c].w};
                    const float x0a[4] = {x0v[q][c].x, x0v[q][c].y, x0v[q][c].z, x0v[q][c].w};
                    float r4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float g = l4[k] * dsq_value((cd[k] >> 1) & 3u) + ge[k];
                        float t = xa[k] + p.step * sgn(g);
                        t = tmax(t, x0a[k] - p.eps);
                        t = tmin(t, x0a[k] + p.eps);
                        r4[k] = tclamp(t, p.lo, p.hi);
                    }
                    *reinterpret_cast<float4 *>(p.x + img + (static_cast<size_t>(c) * H + i) * W + jb) = make_float4(r4[0], r4[1], r4[2], r4[3]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

Weights load_weights(const float *w27) {
    Weights wt;
    for (int k = 0; k < 9; ++k) {
        wt.g[k] = w27[k];
        wt.sx[k] = w27[9 + k];
        wt.sy[k] = w27[18 + k];
    }
    return wt;
}

template <int H, int W>
size_t fwd_lds_bytes(int C) {
    using D = Dims<H, W>;
    return sizeof(float) * (static_cast<size_t>(C) * (H + 4) * (W + 2 * kColHalo) + static_cast<size_t>(C) * D::HP * D::AS + D::NTAB + C * W + 4) +
           static_cast<size_t>(H) * W + static_cast<size_t>(C) * H * W;
}

template <int H, int W>
size_t bwd_lds_bytes(int C) {
    using D = Dims<H, W>;
    return sizeof(float) * (3 * static_cast<size_t>(H + 8) * (W + 2 * kColHalo) + static_cast<size_t>(C) * D::HP * D::AS + D::NTAB);
}

template <class K>
int opt_in_lds(K kernel, size_t bytes) {
    // above 64 KB of dynamic LDS a kernel has to say so once (gfx950: up to 160 KB per workgroup)
    if (bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)) != hipSuccess)
        (void)hipGetLastError();  // best effort: a runtime that needs no opt-in may reject the attribute; the launch itself reports a real limit
    return EE_OK;
}

template <int C, int H, int W>
int launch_fwd(const FwdParams &p, const Weights &wt, bool square, hipStream_t s) {
    const size_t bytes = fwd_lds_bytes<H, W>(C);
    const dim3 grid(static_cast<unsigned>(p.B)), block((kSW + C) * kWave);
    if (square) {
        static int ok = opt_in_lds(chain_fwd_kernel<C, H, W, true>, fwd_lds_bytes<H, W>(C));
        if (ok != EE_OK) return ok;
        EE_LAUNCH((chain_fwd_kernel<C, H, W, true>), grid, block, bytes, s, p, wt);
    } else {
        static int ok = opt_in_lds(chain_fwd_kernel<C, H, W, false>, fwd_lds_bytes<H, W>(C));
        if (ok != EE_OK) return ok;
        EE_LAUNCH((chain_fwd_kernel<C, H, W, false>), grid, block, bytes, s, p, wt);
    }
    return launch_status();
}

template <int C, int H, int W>
int launch_bwd(const BwdParams &p, const Weights &wt, int B, hipStream_t s) {
    const size_t bytes = bwd_lds_bytes<H, W>(C);
    static int ok = opt_in_lds(chain_bwd_kernel<C, H, W>, bwd_lds_bytes<H, W>(C));
    if (ok != EE_OK) return ok;
    EE_LAUNCH((chain_bwd_kernel<C, H, W>), dim3(static_cast<unsigned>(B)), dim3((kSW + C) * kWave), bytes, s, p, wt);
    return launch_status();
}

// shapes with a fused instantiation: (C, H, W) of the reference configs + a small one for tests
enum Shape { S_NONE, S_3_64, S_1_28, S_3_32, S_1_64 };
Shape shape_of(int C, int H, int W) {
    if (C == 3 && H == 64 && W == 64) return S_3_64;  // Tiny-ImageNet
    if (C == 1 && H == 28 && W == 28) return S_1_28;  // MNIST
    if (C == 3 && H == 32 && W == 32) return S_3_32;
    if (C == 1 && H == 64 && W == 64) return S_1_64;
    return S_NONE;
}

inline bool al16(const void *q) { return !q || aligned16(q); }

}  // namespace

EE_API int ee_chain_supported(int C, int H, int W) { return shape_of(C, H, W) != S_NONE; }

EE_API int ee_chain_table_floats(int H, int W) {
    if (H < 1 || W < 1 || H > 64 || W > 64) return EE_ERR_SHAPE;
    const int hp = (H + 15) / 16 * 16, wp = (W + 15) / 16 * 16;
    return (wp / 4 + 2 * (hp / 16) * 4 + (hp / 16) * 2 * 4 + (wp / 16) * 4) * 64;
}

EE_API int ee_chain_fwd_f32(const float *x, int B, int C, int H, int W, const float *tables, const float *weights27, float alpha, float high,
                            float w, int square, float eps, int sq_size, uint64_t *draw_state, const float *stripe_in, const int64_t *sq_pos_in,
                            const float *sq_sign_in, float *x_in, uint8_t *gate, float *gx, float *gy, float *edge, void *stream) {
    if (B < 0 || C < 1 || H < 1 || W < 1) return EE_ERR_SHAPE;
    const Shape sh = shape_of(C, H, W);
    if (sh == S_NONE) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!x || !tables || !weights27 || !x_in || !gate || !gx || !gy) return EE_ERR_NULL;
    if (square) {
        if (sq_size < 1 || sq_size > H) return EE_ERR_SHAPE;
        const bool injected = stripe_in || sq_pos_in || sq_sign_in;
        if (injected && !(stripe_in && sq_pos_in && sq_sign_in)) return EE_ERR_NULL;
        if (!injected && !draw_state) return EE_ERR_NULL;
    }
    if (!al16(x) || !al16(tables) || !al16(x_in) || !al16(gx) || !al16(gy) || !al16(edge) || (reinterpret_cast<uintptr_t>(gate) & 3u)) return EE_ERR_ALIGN;
    FwdParams p{};
    p.x = x; p.x_in = x_in; p.gate = gate; p.gx = gx; p.gy = gy; p.edge = edge; p.tables = tables;
    p.stripe_in = square ? stripe_in : nullptr; p.sq_pos_in = sq_pos_in; p.sq_sign_in = sq_sign_in;
    p.state = reinterpret_cast<unsigned long long *>(draw_state);
    p.B = B; p.sq_size = sq_size;
    p.eps = eps; p.two_eps = static_cast<float>(2.0 * static_cast<double>(eps));
    p.alpha = alpha; p.high = high; p.w = w;
    const Weights wt = load_weights(weights27);
    hipStream_t st = as_stream(stream);
    ProfScope prof(EE_K_CHAIN_FWD, st);
    switch (sh) {
        case S_3_64: return launch_fwd<3, 64, 64>(p, wt, square != 0, st);
        case S_1_28: return launch_fwd<1, 28, 28>(p, wt, square != 0, st);
        case S_3_32: return launch_fwd<3, 32, 32>(p, wt, square != 0, st);
        default: return launch_fwd<1, 64, 64>(p, wt, square != 0, st);
    }
}

EE_API int ee_chain_bwd_f32(const float *g_in, const uint8_t *gate, const float *gx, const float *gy, float *x, const float *x0, int B, int C,
                            int H, int W, const float *tables, const float *weights27, float alpha, float high, float w, float step, float eps,
                            float lo, float hi, int dir, void *stream) {
    if (B < 0 || C < 1 || H < 1 || W < 1 || (dir != 1 && dir != -1)) return EE_ERR_SHAPE;
    const Shape sh = shape_of(C, H, W);
    if (sh == S_NONE) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!g_in || !gate || !gx || !gy || !x || !x0 || !tables || !weights27) return EE_ERR_NULL;
    if (!al16(g_in) || !al16(gx) || !al16(gy) || !al16(x) || !al16(x0) || !al16(tables) || (reinterpret_cast<uintptr_t>(gate) & 3u)) return EE_ERR_ALIGN;
    BwdParams p{};
    p.g_in = g_in; p.gate = gate; p.gx = gx; p.gy = gy; p.x = x; p.x0 = x0; p.tables = tables;
    p.alpha = alpha; p.high = high; p.w = w;
    p.step = dir > 0 ? step : -step; p.eps = eps; p.lo = lo; p.hi = hi;
    const Weights wt = load_weights(weights27);
    hipStream_t st = as_stream(stream);
    ProfScope prof(EE_K_CHAIN_BWD, st);
    switch (sh) {
        case S_3_64: return launch_bwd<3, 64, 64>(p, wt, B, st);
        case S_1_28: return launch_bwd<1, 28, 28>(p, wt, B, st);
        case S_3_32: return launch_bwd<3, 32, 32>(p, wt, B, st);
        default: return launch_bwd<1, 64, 64>(p, wt, B, st);
    }
}
