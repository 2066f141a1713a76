// ee_bf16x3.hip - f32 matrix products on the BF16 matrix cores: every f32 operand is the exact sum of three bf16 pieces
// (x = hi + mid + lo: 8 + 8 + 8 significant bits, each piece rounded to nearest), a product a * b is the sum of the six piece
// products of weight >= 2^-16 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; each exact in f32), accumulated in f32 by
// v_mfma_f32_16x16x32_bf16.  The three dropped terms are below 2^-23 of |a * b| - the size of one f32 rounding - so the result carries
// f32 accuracy (tests/test_gpu_kernels.py: against float64, beside the f32 fma chain), NOT the bits of an f32 fma chain.
//
// Why: v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate (MI355X_MICROARCH.md, matrix cores), and every convolution / dense product of
// the CNN body is bound by it.  Six bf16 MFMAs per f32-equivalent product are 16 / 6 = 2.7 x the f32 matrix rate.
//
// ROUND 4 PILOT, opt-in: the plain product behind layer4's 2x2-map convolutions (ee_dense.hip's shape, [100 x 2048] . [2048 x 2048]).
//   ee_split_bf16x3_f32      x -> (hi, mid, lo)                                   element-wise
//   ee_gemm_bf16x3_nt_f32    C[M][N] = A[M][K] . B[N][K]^T from the split pieces  (K a multiple of 128, N of 32)
// A workgroup owns a 32 x 32 tile of C; its four wavefronts split the reduction (K / 4 each) and own the whole tile (2 x 2 blocks of
// 16 x 16): operand fragments go global -> registers (a lane's fragment is 16 contiguous bytes of a row; no LDS: nothing is shared
// between wavefronts), two reduction steps in flight; the four partial tiles meet in LDS and are added in wavefront order
// (bit-reproducible).
//
// CNN-body glue, not a row of SURVEY.md section 8.
#include "ee_common.hpp"

namespace {

using namespace ee;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// round-to-nearest-even f32 -> bf16 on the bits (finite inputs; NaN / Inf: the pieces are garbage-in, NaN-out - see the header note)
__device__ __forceinline__ unsigned short bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    return static_cast<unsigned short>((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float(static_cast<unsigned>(h) << 16); }

__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float *__restrict__ x, long n, unsigned short *__restrict__ hi,
                                                           unsigned short *__restrict__ mid, unsigned short *__restrict__ lo) {
    const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const unsigned short h = bf16_rne(v);
    const float r1 = v - bf16_f32(h);  // exact: at most 16 significant bits
    const unsigned short m = bf16_rne(r1);
    const float r2 = r1 - bf16_f32(m);  // exact: at most 8
    hi[i] = h, mid[i] = m, lo[i] = bf16_rne(r2);
}

constexpr int BX_TM = 32, BX_TN = 32, BX_KS = 32;  // tile of C per workgroup, reduction indices per MFMA

struct Frag3 {
    uint4 h, m, l;
};
__device__ __forceinline__ Frag3 load3(const unsigned short *h, const unsigned short *m, const unsigned short *l, size_t off) {
    return Frag3{*reinterpret_cast<const uint4 *>(h + off), *reinterpret_cast<const uint4 *>(m + off), *reinterpret_cast<const uint4 *>(l + off)};
}
__device__ __forceinline__ f32x4 mfma6(const Frag3 &a, const Frag3 &b, f32x4 acc) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.h), am = __builtin_bit_cast(bf16x8, a.m), al = __builtin_bit_cast(bf16x8, a.l);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b.h), bm = __builtin_bit_cast(bf16x8, b.m), bl = __builtin_bit_cast(bf16x8, b.l);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);  // the small terms first
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    return acc;
}

__global__ __launch_bounds__(256) void gemm_bf16x3_nt_kernel(const unsigned short *__restrict__ ah, const unsigned short *__restrict__ am,
                                                             const unsigned short *__restrict__ al, const unsigned short *__restrict__ bh,
                                                             const unsigned short *__restrict__ bm, const unsigned short *__restrict__ bl,
                                                             float *__restrict__ c, int M, int N, int K) {
    __shared__ float red[4][BX_TM][BX_TN + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int mtiles = (M + BX_TM - 1) / BX_TM;
    const int nt = blockIdx.x / mtiles, mt = blockIdx.x - nt * mtiles;  // the few row tiles of one column tile are neighbours (they share B)
    const int m0 = mt * BX_TM, n0 = nt * BX_TN;
    const int kq = K / 4, k0 = wave * kq, steps = kq / BX_KS;
    // fragment rows of this lane: A rows m0 + l15, m0 + 16 + l15 (clamped into the matrix: never stored), B rows (= columns of C) n0 + l15, + 16
    const int ra0 = m0 + l15 < M ? m0 + l15 : M - 1, ra1 = m0 + 16 + l15 < M ? m0 + 16 + l15 : M - 1;
    const size_t oa0 = static_cast<size_t>(ra0) * K + k0 + 8 * lq, oa1 = static_cast<size_t>(ra1) * K + k0 + 8 * lq;
    const size_t ob0 = static_cast<size_t>(n0 + l15) * K + k0 + 8 * lq, ob1 = static_cast<size_t>(n0 + 16 + l15) * K + k0 + 8 * lq;
    f32x4 acc00 = {0.0f, 0.0f, 0.0f, 0.0f}, acc01 = acc00, acc10 = acc00, acc11 = acc00;
    // two register sets: step s + 1 travels while step s is multiplied
    Frag3 A0 = load3(ah, am, al, oa0), A1 = load3(ah, am, al, oa1), B0 = load3(bh, bm, bl, ob0), B1 = load3(bh, bm, bl, ob1);
    for (int s = 0; s < steps; ++s) {
        const size_t kn = static_cast<size_t>(s + 1 < steps ? s + 1 : s) * BX_KS;  // always issued (the last one re-reads its own step)
        const Frag3 A0n = load3(ah, am, al, oa0 + kn), A1n = load3(ah, am, al, oa1 + kn), B0n = load3(bh, bm, bl, ob0 + kn), B1n = load3(bh, bm, bl, ob1 + kn);
        acc00 = mfma6(A0, B0, acc00);
        acc01 = mfma6(A0, B1, acc01);
        acc10 = mfma6(A1, B0, acc10);
        acc11 = mfma6(A1, B1, acc11);
        A0 = A0n, A1 = A1n, B0 = B0n, B1 = B1n;
    }
    // D element r of lane (l15, lq): row 4 lq + r, column l15
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wave][4 * lq + r][l15] = acc00[r];
        red[wave][4 * lq + r][16 + l15] = acc01[r];
        red[wave][16 + 4 * lq + r][l15] = acc10[r];
        red[wave][16 + 4 * lq + r][16 + l15] = acc11[r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < BX_TM * BX_TN; e += 256) {
        const int row = e / BX_TN, col = e - row * BX_TN;
        if (m0 + row < M) c[static_cast<size_t>(m0 + row) * N + n0 + col] = ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
    }
}

}  // namespace

// x [n] f32 -> hi, mid, lo [n] bf16 (as 16-bit words) with x == hi + mid + lo exactly (finite x; each piece rounded to nearest even)
EE_API int ee_split_bf16x3_f32(const float *x, int64_t n, uint16_t *hi, uint16_t *mid, uint16_t *lo, void *stream) {
    if (n < 0) return EE_ERR_SHAPE;
    if (n == 0) return EE_OK;
    if (!x || !hi || !mid || !lo) return EE_ERR_NULL;
    EE_LAUNCH(split_bf16x3_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, static_cast<long>(n), hi, mid, lo);
    return launch_status();
}

// C [M][N] (f32) = A [M][K] . B [N][K]^T, both given as their three bf16 pieces (ee_split_bf16x3_f32); K % 128 == 0, N % 32 == 0, 16-byte
// aligned pieces; else EE_ERR_UNSUPPORTED / EE_ERR_ALIGN
EE_API int ee_gemm_bf16x3_nt_f32(const uint16_t *a_hi, const uint16_t *a_mid, const uint16_t *a_lo, const uint16_t *b_hi, const uint16_t *b_mid,
                                 const uint16_t *b_lo, float *c, int M, int N, int K, void *stream) {
    if (M < 0 || N < 1 || K < 1) return EE_ERR_SHAPE;
    if (K % 128 != 0 || N % BX_TN != 0) return EE_ERR_UNSUPPORTED;
    if (M == 0) return EE_OK;
    if (!a_hi || !a_mid || !a_lo || !b_hi || !b_mid || !b_lo || !c) return EE_ERR_NULL;
    if (!aligned16(a_hi) || !aligned16(a_mid) || !aligned16(a_lo) || !aligned16(b_hi) || !aligned16(b_mid) || !aligned16(b_lo)) return EE_ERR_ALIGN;
    const int64_t grid = static_cast<int64_t>((M + BX_TM - 1) / BX_TM) * (N / BX_TN);
    if (grid > 0x7fffffffLL) return EE_ERR_SHAPE;
    EE_LAUNCH(gemm_bf16x3_nt_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, as_stream(stream), a_hi, a_mid, a_lo, b_hi, b_mid, b_lo, c, M, N, K);
    return launch_status();
}
