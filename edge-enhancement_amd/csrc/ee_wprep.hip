// ee_wprep.hip - the convolution kernels' filters in the order their kernels read them, rebuilt from the nn.Conv2d weights once per
// optimiser step (inside the captured update graph): Winograd F(2x2, 3x3) filter transforms U = G g G^T (ee_wino.hip), the tap-major slabs
// of the stride-2 kernels with or without the shortcut's 1x1 filters (ee_s2.hip), the dense [4 Cin, 4 Cout] matrix of a 3x3 convolution
// on a 2x2 map (layer 4: functional.Conv3x3Map2Fn).  As torch expressions (einsum / cat / permute / advanced indexing, the restatement
// in functional._rearranged) these were 120 launches and 0.55 ms of a 12.6 ms training step; here one launch per (weight, kind).
//
// Every kind is a gather with a closed-form index (plus 16 sums for the Winograd ones): one lane per OUTPUT element, or per filter pair.
// CNN-body glue, not a row of SURVEY.md section 8.
#include "ee_common.hpp"

namespace {

using namespace ee;

// U[xi = 4 i + j][k][r] = (G g G^T)[i][j], G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]].
// forward: g = w[r][k] (k = input channel, r = output channel); backward-data: g = w[k][r] rotated by 180 degrees (k = output channel).
// One lane per (k, r), r fastest: the 16 stores of a wavefront are contiguous rows.
__device__ __forceinline__ void wino_filter_body(const float *__restrict__ w, float *__restrict__ u, int Cout, int Cin, int backward, unsigned block) {
    const int K = backward ? Cout : Cin, R = backward ? Cin : Cout;
    const int idx = block * 256 + threadIdx.x;
    if (idx >= K * R) return;
    const int k = idx / R, r = idx - k * R;
    const float *g = w + (static_cast<size_t>(backward ? k : r) * Cin + (backward ? r : k)) * 9;
    float gg[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) gg[a][b] = backward ? g[(2 - a) * 3 + (2 - b)] : g[a * 3 + b];
    float t[4][3];  // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float s = gg[0][b] + gg[2][b];
        t[0][b] = gg[0][b];
        t[1][b] = 0.5f * (s + gg[1][b]);
        t[2][b] = 0.5f * (s - gg[1][b]);
        t[3][b] = gg[2][b];
    }
    float *o = u + static_cast<size_t>(k) * R + r;
    const size_t xs = static_cast<size_t>(K) * R;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float s = t[i][0] + t[i][2];
        o[(4 * i + 0) * xs] = t[i][0];
        o[(4 * i + 1) * xs] = 0.5f * (s + t[i][1]);
        o[(4 * i + 2) * xs] = 0.5f * (s - t[i][1]);
        o[(4 * i + 3) * xs] = t[i][2];
    }
}

// out [R/32][K/16][TAPS][4][2][16][4]: out[rb][rd][t][q][h][m][k] = W(r = 32 rb + 16 h + m, kk = 16 rd + 4 q + k, t), where
// forward (R = Cout, K = Cin): W = w3[r][kk][t] (t < 9) or w1[r][kk]; backward (R = Cin, K = Cout): W = w3[kk][r][t] or w1[kk][r]
__device__ __forceinline__ void s2_slab_body(const float *__restrict__ w3, const float *__restrict__ w1, float *__restrict__ out, int Cout, int Cin,
                                             int backward, int taps, unsigned block) {
    const int K = backward ? Cout : Cin, R = backward ? Cin : Cout;
    const size_t total = static_cast<size_t>(R) * K * taps;
    const size_t idx = static_cast<size_t>(block) * 256 + threadIdx.x;
    if (idx >= total) return;
    size_t v = idx;
    const int k = v & 3;
    v >>= 2;
    const int m = v & 15;
    v >>= 4;
    const int h = v & 1;
    v >>= 1;
    const int q = v & 3;
    v >>= 2;
    const int t = static_cast<int>(v % taps);
    v /= taps;
    const int rd = static_cast<int>(v % (K / 16)), rb = static_cast<int>(v / (K / 16));
    const int r = 32 * rb + 16 * h + m, kk = 16 * rd + 4 * q + k;
    const int co = backward ? kk : r, ci = backward ? r : kk;
    out[idx] = t < 9 ? w3[(static_cast<size_t>(co) * Cin + ci) * 9 + t] : w1[static_cast<size_t>(co) * Cin + ci];
}

// 3x3 / stride 1 / padding 1 on a 2x2 map as a dense product: out[(ci, iy, ix)][(co, oy, ox)] = w[co][ci][iy - oy + 1][ix - ox + 1]
__device__ __forceinline__ void dense_map2_body(const float *__restrict__ w, float *__restrict__ out, int Cout, int Cin, unsigned block) {
    const size_t total = 16 * static_cast<size_t>(Cout) * Cin;
    const size_t idx = static_cast<size_t>(block) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int col = static_cast<int>(idx % (4 * static_cast<size_t>(Cout))), row = static_cast<int>(idx / (4 * static_cast<size_t>(Cout)));
    const int co = col >> 2, oy = (col >> 1) & 1, ox = col & 1, ci = row >> 2, iy = (row >> 1) & 1, ix = row & 1;
    out[idx] = w[(static_cast<size_t>(co) * Cin + ci) * 9 + (iy - oy + 1) * 3 + (ix - ox + 1)];
}

// EE_WPREP_WINO_FB: both Winograd sets of a weight in one pass over it.  wino_filter_body gathers its nine taps with the lanes along the RESULT
// channel - forward that is a stride of 9 Cin floats between lanes (one 4-byte use per fetched sector), and every weight is read twice (forward,
// backward-data).  Here a workgroup reads a 16 x 16 tile of (co, ci) filters ONCE, row by row (16 x 9 contiguous floats per output channel),
// into LDS and writes both sets with the lanes along each set's fastest index.  Same sums in the same order: the same bits.
constexpr int FB_T = 16;               // the tile: 16 x 16 (co, ci) filters per workgroup - (C / 16)^2 workgroups, 256 for the 256-channel layers
constexpr int FB_ROW = FB_T * 9 + 1;   // LDS row of an output channel's 16 filters (+1: lanes along co on distinct banks)

__device__ __forceinline__ void wino_transform_store(const float (&gg)[3][3], float *o, size_t xs) {
    float t[4][3];  // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float s = gg[0][b] + gg[2][b];
        t[0][b] = gg[0][b];
        t[1][b] = 0.5f * (s + gg[1][b]);
        t[2][b] = 0.5f * (s - gg[1][b]);
        t[3][b] = gg[2][b];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float s = t[i][0] + t[i][2];
        o[(4 * i + 0) * xs] = t[i][0];
        o[(4 * i + 1) * xs] = 0.5f * (s + t[i][1]);
        o[(4 * i + 2) * xs] = 0.5f * (s - t[i][1]);
        o[(4 * i + 3) * xs] = t[i][2];
    }
}

__device__ __forceinline__ void wino_filter_fb_body(const float *__restrict__ w, float *__restrict__ uf, float *__restrict__ ub, int Cout, int Cin, unsigned block,
                                                    float *g) {
    const int nci = Cin / FB_T;
    const int co0 = (block / nci) * FB_T, ci0 = (block % nci) * FB_T;
    for (int i = threadIdx.x; i < FB_T * FB_T * 9; i += 256) {
        const int r = i / (FB_T * 9), c = i - r * (FB_T * 9);
        g[r * FB_ROW + c] = w[(static_cast<size_t>(co0 + r) * Cin + ci0) * 9 + c];
    }
    __syncthreads();
    const size_t xs = static_cast<size_t>(Cout) * Cin;
    const int lo = threadIdx.x & (FB_T - 1), hi = threadIdx.x / FB_T;
    {  // forward set u[xi][ci][co]: lanes along co
        const int co = lo, ci = hi;
        const float *p = g + co * FB_ROW + ci * 9;
        float gg[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) gg[a][b] = p[a * 3 + b];
        wino_transform_store(gg, uf + static_cast<size_t>(ci0 + ci) * Cout + co0 + co, xs);
    }
    {  // backward-data set u[xi][co][ci] of the filters rotated by 180 degrees: lanes along ci
        const int ci = lo, co = hi;
        const float *p = g + co * FB_ROW + ci * 9;
        float gg[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) gg[a][b] = p[(2 - a) * 3 + (2 - b)];
        wino_transform_store(gg, ub + static_cast<size_t>(co0 + co) * Cin + ci0 + ci, xs);
    }
}

__global__ __launch_bounds__(256) void wino_filter_fb_kernel(const float *__restrict__ w, float *__restrict__ uf, float *__restrict__ ub, int Cout, int Cin) {
    __shared__ float g[FB_T * FB_ROW];
    wino_filter_fb_body(w, uf, ub, Cout, Cin, blockIdx.x, g);
}

__device__ __forceinline__ void wprep_dispatch(int kind, const float *w, const float *w1, float *out, int Cout, int Cin, unsigned block) {
    if (kind == EE_WPREP_WINO_F || kind == EE_WPREP_WINO_B)
        wino_filter_body(w, out, Cout, Cin, kind == EE_WPREP_WINO_B ? 1 : 0, block);
    else if (kind == EE_WPREP_DENSE_MAP2)
        dense_map2_body(w, out, Cout, Cin, block);
    else
        s2_slab_body(w, w1, out, Cout, Cin, (kind == EE_WPREP_S2M_B || kind == EE_WPREP_S2P_B) ? 1 : 0, (kind == EE_WPREP_S2P_F || kind == EE_WPREP_S2P_B) ? 10 : 9, block);
}

__global__ __launch_bounds__(256) void wprep_kernel(int kind, const float *__restrict__ w, const float *__restrict__ w1, float *__restrict__ out, int Cout, int Cin) {
    wprep_dispatch(kind, w, w1, out, Cout, Cin, blockIdx.x);
}

// ---- every rearranged copy of a model in ONE launch (round 4): the captured update ended with 19 launches of ~5 us for ResNet-18 (one per
// weight and kind: 125 us of a 11.7 ms step).  The descriptors travel BY VALUE in the kernel arguments (the pointers are static: the captured
// graph keeps them); a workgroup finds its item by a scan of the first-block table (<= 64 entries, wavefront-uniform) and runs that item's body.
constexpr int WB_MAX = 64;
struct WprepBatch {
    int n;
    unsigned first[WB_MAX + 1];  // first workgroup of item i; first[n] = the grid
    const float *w[WB_MAX], *w1[WB_MAX];
    float *out[WB_MAX];
    short kind[WB_MAX];
    short cout[WB_MAX], cin[WB_MAX];
};

__global__ __launch_bounds__(256) void wprep_batch_kernel(WprepBatch b) {
    __shared__ float g[FB_T * FB_ROW];
    int i = 0;
    while (i + 1 < b.n && blockIdx.x >= b.first[i + 1]) ++i;
    const unsigned block = blockIdx.x - b.first[i];
    const int kind = b.kind[i], Cout = b.cout[i], Cin = b.cin[i];
    if (kind == EE_WPREP_WINO_FB)
        wino_filter_fb_body(b.w[i], b.out[i], b.out[i] + 16 * static_cast<size_t>(Cout) * Cin, Cout, Cin, block, g);
    else
        wprep_dispatch(kind, b.w[i], b.w1[i], b.out[i], Cout, Cin, block);
}

unsigned wprep_blocks(int kind, size_t pairs) {
    if (kind == EE_WPREP_WINO_F || kind == EE_WPREP_WINO_B) return static_cast<unsigned>((pairs + 255) / 256);
    if (kind == EE_WPREP_DENSE_MAP2) return static_cast<unsigned>((16 * pairs + 255) / 256);
    return static_cast<unsigned>((pairs * ((kind == EE_WPREP_S2P_F || kind == EE_WPREP_S2P_B) ? 10 : 9) + 255) / 256);
}

}  // namespace

// kind: EE_WPREP_* (eeadv.h).  w [Cout][Cin][3][3]; w1 [Cout][Cin] (the shortcut's 1x1 filters, EE_WPREP_S2P_* only, else NULL);
// out: see the kind.  Channel counts: multiples of 32 for the stride-2 slabs (16 on the reduction side), any for the others.
EE_API int ee_conv_weight_prep_f32(int kind, const float *w, const float *w1, float *out, int Cout, int Cin, void *stream) {
    if (Cout < 1 || Cin < 1) return EE_ERR_SHAPE;
    if (!w || !out) return EE_ERR_NULL;
    if (static_cast<int64_t>(Cout) * Cin > (1LL << 26)) return EE_ERR_SHAPE;
    hipStream_t st = as_stream(stream);
    const size_t pairs = static_cast<size_t>(Cout) * Cin;
    if (kind == EE_WPREP_WINO_FB) {
        if (Cout % 32 != 0 || Cin % 32 != 0) return EE_ERR_UNSUPPORTED;
        EE_LAUNCH(wino_filter_fb_kernel, dim3(static_cast<unsigned>((Cout / FB_T) * (Cin / FB_T))), dim3(256), 0, st, w, out, out + 16 * pairs, Cout, Cin);
        return launch_status();
    }
    if (kind < EE_WPREP_WINO_F || kind > EE_WPREP_DENSE_MAP2) return EE_ERR_UNSUPPORTED;
    if (kind >= EE_WPREP_S2M_F && kind <= EE_WPREP_S2P_B) {
        const bool backward = kind == EE_WPREP_S2M_B || kind == EE_WPREP_S2P_B;
        if ((backward ? Cout : Cin) % 16 != 0 || (backward ? Cin : Cout) % 32 != 0) return EE_ERR_UNSUPPORTED;
        if ((kind == EE_WPREP_S2P_F || kind == EE_WPREP_S2P_B) && !w1) return EE_ERR_NULL;
    }
    EE_LAUNCH(wprep_kernel, dim3(wprep_blocks(kind, pairs)), dim3(256), 0, st, kind, w, w1, out, Cout, Cin);
    return launch_status();
}

// n rearranged copies in one launch per 64 items: kinds[i], w[i] [cout[i]][cin[i]][3][3], w1[i] (or NULL), out[i] - each exactly as
// ee_conv_weight_prep_f32 takes them (the same bodies: the same bits).  The arrays are HOST arrays of device pointers / ints.
EE_API int ee_conv_weight_prep_batch_f32(int n, const int *kinds, const float *const *w, const float *const *w1, float *const *out, const int *cout,
                                         const int *cin, void *stream) {
    if (n < 0) return EE_ERR_SHAPE;
    if (n == 0) return EE_OK;
    if (!kinds || !w || !w1 || !out || !cout || !cin) return EE_ERR_NULL;
    hipStream_t st = as_stream(stream);
    for (int base = 0; base < n; base += WB_MAX) {
        WprepBatch b{};
        b.n = n - base < WB_MAX ? n - base : WB_MAX;
        unsigned blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            const int j = base + i, kind = kinds[j], Cout = cout[j], Cin = cin[j];
            if (Cout < 1 || Cin < 1 || Cout > 32767 || Cin > 32767) return EE_ERR_SHAPE;
            if (!w[j] || !out[j]) return EE_ERR_NULL;
            const size_t pairs = static_cast<size_t>(Cout) * Cin;
            unsigned nb;
            if (kind == EE_WPREP_WINO_FB) {
                if (Cout % 32 != 0 || Cin % 32 != 0) return EE_ERR_UNSUPPORTED;
                nb = static_cast<unsigned>((Cout / FB_T) * (Cin / FB_T));
            } else {
                if (kind < EE_WPREP_WINO_F || kind > EE_WPREP_DENSE_MAP2) return EE_ERR_UNSUPPORTED;
                if (kind >= EE_WPREP_S2M_F && kind <= EE_WPREP_S2P_B) {
                    const bool backward = kind == EE_WPREP_S2M_B || kind == EE_WPREP_S2P_B;
                    if ((backward ? Cout : Cin) % 16 != 0 || (backward ? Cin : Cout) % 32 != 0) return EE_ERR_UNSUPPORTED;
                    if ((kind == EE_WPREP_S2P_F || kind == EE_WPREP_S2P_B) && !w1[j]) return EE_ERR_NULL;
                }
                nb = wprep_blocks(kind, pairs);
            }
            b.first[i] = blocks;
            blocks += nb;
            b.w[i] = w[j], b.w1[i] = w1[j], b.out[i] = out[j];
            b.kind[i] = static_cast<short>(kind), b.cout[i] = static_cast<short>(Cout), b.cin[i] = static_cast<short>(Cin);
        }
        b.first[b.n] = blocks;
        EE_LAUNCH(wprep_batch_kernel, dim3(blocks), dim3(256), 0, st, b);
        const int rc = launch_status();
        if (rc != EE_OK) return rc;
    }
    return EE_OK;
}
