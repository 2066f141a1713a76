// ee_hfs_mfma.hip - HighFreqSuppress (utils/core.py:15-55) for planes up to 256 x 256 (ImageNet: 224 x 224, r = 16) on the
// f32 matrix cores, one workgroup per plane, ONE WAVEFRONT PER 16-ROW BAND.
//
// ee_hfs.hip keeps a whole plane in LDS and is limited to 64 x 64; beyond that the host layer used three rocBLAS launches and a
// [B,C,H,2W] intermediate in HBM.  The operator is still the rank-(NU x 2NV) one (eeadv/hfs.py), so here it is the same four
// skinny products as in ee_chain.hip, chained through the v_mfma_f32_16x16x4_f32 accumulators, but split over the bands:
//     S1  [P | Q]   = X_band  T1                 per band: 16 x W times W x 32      (X staged through a wave-private LDS strip)
//     S2  R_band    = CS_band^T [P | Q]          per band: 2NU x 16 times 16 x 32   -> summed over the bands through LDS
//     EF            = (a - d | c + b ; b + c | d - a)                                   (tile-wise register arithmetic)
//     S3  [U | V]^T = EF^T CS_band^T / H         per band
//     S4  y_band    = [U | V] T4                 per band: 16 x 32 times 32 x W
// scripts/chain_emulate.py (big_tables / big_apply) is the specification of the fragment-ordered tables and checks the index
// algebra against the dense operator in float64.  HBM traffic = the algorithmic minimum, read x + write y (8 B per element;
// 12 with sq_mode 2).  The band reduction adds in band order: results are bit-reproducible run to run.
//
// sq_mode as in ee_hfs_f32: 1 applies Add_Square (core.py:636-655) while a strip is staged, 2 multiplies the result by
// d add_square / dx evaluated at sq_x.  Parity: UNPINNED like ee_hfs.hip (torch.rfft is gone); 3e-7 from the float64 operator.
#include "ee_common.hpp"
#include "ee_square.hpp"

namespace {

using namespace ee;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 64;           // columns per staged strip
constexpr int SS = KC + 4;       // strip row stride: bank = (4 i + g + 4 s) % 64, conflict-free A-fragment reads
constexpr int STRIP = 16 * SS;   // floats per wave

struct BandDims {
    int H, W, HP, WP, NB, WT;  // NB = HP / 16 bands (= wavefronts), WT = WP / 16
    int C;
};

template <int SQ, int MT2>
__global__ __launch_bounds__(1024) void hfs_band_kernel(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ tables,
                                                        BandDims d, const float *__restrict__ sq_x, SquareArgs sq) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, band = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int H = d.H, W = d.W, WP = d.WP, NB = d.NB, NT = NB * 64;
    float *strip = lds + band * STRIP;        // this wave's [16][SS] staging strip (later: partial R tiles, output tiles)
    float *tab = lds + NB * STRIP;            // T1, later T4: WP * 32 floats
    float *rsum = tab + WP * 32;              // [MT2][2][4][64] band-summed R
    const int plane = blockIdx.x, c = plane % d.C;
    const size_t poff = static_cast<size_t>(plane) * H * W;
    const float *src = in + poff;
    const int n1 = WP / 4 * 2, n2 = NB * MT2 * 4, n3 = n2;  // fragments of T1, T2, T3
    const float *g1 = tables, *g2 = g1 + n1 * 64, *g3 = g2 + n2 * 64, *g4 = g3 + n3 * 64;
    SquarePlane pl{};
    const float *stripe_row = nullptr;
    if (SQ != 0) {
        pl = square_plane(sq, c);
        stripe_row = sq.stripe + static_cast<size_t>(plane) * W;
    }
    const bool vec = ((W & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) && (SQ == 0 || ((reinterpret_cast<uintptr_t>(stripe_row) & 15u) == 0));

    // ---- T1 -> LDS ----------------------------------------------------------------------------------------------------------
    for (int i = tid; i < WP * 8; i += NT) reinterpret_cast<float4 *>(tab)[i] = reinterpret_cast<const float4 *>(g1)[i];
    __syncthreads();

    // ---- S1: [P | Q] of this band, strips of KC columns staged through the wave's private LDS strip ------------------------------
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 pq[2] = {zero, zero};
    const int row0 = 16 * band;
    auto load_strip = [&](int col0, float4 (&v)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (lane >> 4) + 4 * q, w0 = col0 + 4 * (lane & 15);  // 16 lanes cover 64 columns of one row: 256 B per row
            const int h = row0 + r;
            const int hc = h < H ? h : H - 1;
            if (vec) {
                const int wc = w0 + 3 < W ? w0 : (W - 4 > 0 ? W - 4 : 0);
                v[q] = *reinterpret_cast<const float4 *>(src + static_cast<size_t>(hc) * W + wc);
            } else {
                float t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = src[static_cast<size_t>(hc) * W + (w0 + k < W ? w0 + k : W - 1)];
                v[q] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
    };
    auto stage_strip = [&](int col0, const float4 (&v)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (lane >> 4) + 4 * q, w0 = col0 + 4 * (lane & 15);
            const int h = row0 + r;
            float t[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int w = w0 + k;
                const bool inside = h < H && w < W;
                if (SQ == 1 && inside) {
                    float dd;
                    t[k] = square_elem<false>(sq, pl, t[k], stripe_row[w], c, h, w, dd);
                }
                if (!inside) t[k] = 0.0f;  // padding rows / columns (and the clamped loads that stood in for them)
            }
            *reinterpret_cast<float4 *>(strip + r * SS + 4 * (lane & 15)) = make_float4(t[0], t[1], t[2], t[3]);
        }
    };
    float4 cur[4], nxt[4];
    load_strip(0, cur);
    for (int col0 = 0; col0 < WP; col0 += KC) {
        if (col0 + KC < WP) load_strip(col0 + KC, nxt);  // in flight while this strip is multiplied
        stage_strip(col0, cur);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int ksteps = (WP - col0 < KC ? WP - col0 : KC) / 4;
        for (int s = 0; s < ksteps; ++s) {
            const float a = strip[li * SS + 4 * s + lg];
            const int f = ((col0 >> 2) + s) * 2;
            pq[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, tab[(f + 0) * 64 + lane], pq[0], 0, 0, 0);
            pq[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, tab[(f + 1) * 64 + lane], pq[1], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();  // every lane has read the strip before it is overwritten
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
    }

    // ---- S2: this band's share of R = CS^T [P | Q] (the PQ accumulators are the B operands as they lie) --------------------------
    f32x4 rc[MT2][2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) {
        rc[mt][0] = zero;
        rc[mt][1] = zero;
    }
    {
        const float *t2 = g2 + static_cast<size_t>(band) * MT2 * 4 * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = t2[(mt * 4 + r) * 64];
                rc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pq[0][r], rc[mt][0], 0, 0, 0);
                rc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pq[1][r], rc[mt][1], 0, 0, 0);
            }
    }
    __syncthreads();  // every band is done with T1 and with its strip

    // ---- band reduction of R through the strips (two rounds: the tiles of P, then those of Q), T4 takes T1's place meanwhile ----------
    for (int i = tid; i < WP * 8; i += NT) reinterpret_cast<float4 *>(tab)[i] = reinterpret_cast<const float4 *>(g4)[i];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) strip[(mt * 4 + r) * 64 + lane] = rc[mt][nt][r];
        __syncthreads();
        for (int e = tid; e < MT2 * 256; e += NT) {
            float acc = 0.0f;
            for (int b = 0; b < NB; ++b) acc += lds[b * STRIP + e];  // band order: reproducible
            const int mt = e >> 8, rl = e & 255;
            rsum[(mt * 2 + nt) * 256 + rl] = acc;
        }
        __syncthreads();
    }

    // ---- EF (tile-wise), S3: [U | V]^T of this band, S4: y_band, written out through the strip in 64-column groups --------------------
    constexpr int HALF = MT2 / 2;
    f32x4 ef[MT2][2];
#pragma unroll
    for (int j = 0; j < HALF; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = rsum[((j * 2 + 0) * 4 + r) * 64 + lane], cq = rsum[((j * 2 + 1) * 4 + r) * 64 + lane];
            const float b = rsum[(((HALF + j) * 2 + 0) * 4 + r) * 64 + lane], dq = rsum[(((HALF + j) * 2 + 1) * 4 + r) * 64 + lane];
            ef[j][0][r] = a - dq;
            ef[j][1][r] = cq + b;
            ef[HALF + j][0][r] = b + cq;
            ef[HALF + j][1][r] = dq - a;
        }
    f32x4 uv[2] = {zero, zero};
    {
        const float *t3 = g3 + static_cast<size_t>(band) * MT2 * 4 * 64 + lane;
#pragma unroll
        for (int kt = 0; kt < MT2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float b = t3[(kt * 4 + r) * 64];
                uv[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ef[kt][0][r], b, uv[0], 0, 0, 0);
                uv[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ef[kt][1][r], b, uv[1], 0, 0, 0);
            }
    }
    float *dst = out + poff;
    const bool vec_out = ((W & 3) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) &&
                         (SQ != 2 || (((reinterpret_cast<uintptr_t>(sq_x) | reinterpret_cast<uintptr_t>(stripe_row)) & 15u) == 0));
    for (int wt0 = 0; wt0 < d.WT; wt0 += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int wt = wt0 + q;
            if (wt < d.WT) {
                f32x4 y = zero;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y = __builtin_amdgcn_mfma_f32_16x16x4f32(uv[nt][r], tab[((wt * 2 + nt) * 4 + r) * 64 + lane], y, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) strip[(4 * lg + r) * SS + 16 * q + li] = y[r];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (lane >> 4) + 4 * q, w0 = 16 * wt0 + 4 * (lane & 15);
            const int h = row0 + r;
            if (h < H && w0 < W) {
                const float4 t = *reinterpret_cast<const float4 *>(strip + r * SS + 4 * (lane & 15));
                float o[4] = {t.x, t.y, t.z, t.w};
                if (SQ == 2) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (w0 + k < W) {
                            float dd = 0.0f;
                            (void)square_elem<true>(sq, pl, sq_x[poff + static_cast<size_t>(h) * W + w0 + k], stripe_row[w0 + k], c, h, w0 + k, dd);
                            o[k] = o[k] * dd;
                        }
                }
                if (vec_out && w0 + 3 < W) {
                    *reinterpret_cast<float4 *>(dst + static_cast<size_t>(h) * W + w0) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    for (int k = 0; k < 4 && w0 + k < W; ++k) dst[static_cast<size_t>(h) * W + w0 + k] = o[k];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int table_floats(int H, int W, int nu_pad) {
    const int hp = (H + 15) / 16 * 16, wp = (W + 15) / 16 * 16, mt2 = 2 * nu_pad / 16;
    return (wp / 4 * 2 + 2 * (hp / 16) * mt2 * 4 + (wp / 16) * 8) * 64;
}

}  // namespace

EE_API int ee_hfs_mfma_table_floats(int H, int W, int nu_pad) {
    if (H < 1 || W < 1 || H > 256 || W > 256 || (nu_pad != 16 && nu_pad != 32)) return EE_ERR_SHAPE;
    return table_floats(H, W, nu_pad);
}

EE_API int ee_hfs_mfma_f32(const float *in, float *out, int B, int C, int H, int W, const float *tables, int nu_pad, int sq_mode,
                           const float *sq_x, float eps, const float *stripe, const float *sq_sign, const int64_t *sq_pos, const int32_t *sq_size,
                           int nq, void *stream) {
    if (B < 0 || C < 1 || H < 1 || W < 1 || sq_mode < 0 || sq_mode > 2 || nq < 0) return EE_ERR_SHAPE;
    if (H > 256 || W > 256 || (nu_pad != 16 && nu_pad != 32)) return EE_ERR_UNSUPPORTED;
    if (B == 0) return EE_OK;
    if (!in || !out || !tables) return EE_ERR_NULL;
    if (sq_mode != 0 && (!stripe || (nq > 0 && (!sq_sign || !sq_pos || !sq_size)))) return EE_ERR_NULL;
    if (sq_mode == 2 && !sq_x) return EE_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(tables) & 15u) return EE_ERR_ALIGN;
    BandDims d{H, W, (H + 15) / 16 * 16, (W + 15) / 16 * 16, 0, 0, C};
    d.NB = d.HP / 16;
    d.WT = d.WP / 16;
    const int mt2 = 2 * nu_pad / 16;
    const size_t lds_bytes = sizeof(float) * (static_cast<size_t>(d.NB) * STRIP + static_cast<size_t>(d.WP) * 32 + static_cast<size_t>(mt2) * 512);
    SquareArgs sq{stripe, sq_sign, sq_pos, sq_size, nq, C, H, W, eps, static_cast<float>(2.0 * static_cast<double>(eps))};
    const dim3 grid(static_cast<unsigned>(static_cast<int64_t>(B) * C)), block(static_cast<unsigned>(d.NB * 64));
    hipStream_t st = as_stream(stream);
    ProfScope prof(sq_mode == 0 ? EE_K_HFS : (sq_mode == 1 ? EE_K_HFS_SQ_FWD : EE_K_HFS_SQ_BWD), st);
#define EE_BAND_LAUNCH(SQ_, MT2_)                                                                                              \
    do {                                                                                                                       \
        static bool opted = false;                                                                                             \
        if (!opted && lds_bytes > 64 * 1024) {                                                                                 \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(hfs_band_kernel<SQ_, MT2_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    160 * 1024) != hipSuccess)                                                                 \
                (void)hipGetLastError();                                                                                       \
            opted = true;                                                                                                      \
        }                                                                                                                      \
        EE_LAUNCH((hfs_band_kernel<SQ_, MT2_>), grid, block, lds_bytes, st, in, out, tables, d, sq_x, sq);                     \
    } while (0)
    if (mt2 == 2) {
        if (sq_mode == 0) EE_BAND_LAUNCH(0, 2);
        else if (sq_mode == 1) EE_BAND_LAUNCH(1, 2);
        else EE_BAND_LAUNCH(2, 2);
    } else {
        if (sq_mode == 0) EE_BAND_LAUNCH(0, 4);
        else if (sq_mode == 1) EE_BAND_LAUNCH(1, 4);
        else EE_BAND_LAUNCH(2, 4);
    }
#undef EE_BAND_LAUNCH
    return launch_status();
}
