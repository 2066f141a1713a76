/*
 * ee_oracle.c - CPU restatement (plain C, scalar, one thread) of the reference's
 * hot-path arithmetic.  TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path never does.
 *
 * Every function states the reference lines it follows (paths relative to the
 * reference repository root).  The floating-point OPERATION ORDER written here
 * is the definition the HIP kernels in edge-enhancement_amd/csrc/ reproduce bit for bit:
 * build with -ffp-contract=off so that only the explicit fmaf() calls fuse.
 *
 * Operation order of the two forward stencils was pinned against the reference
 * itself (torch 2.10 CPU / oneDNN, this container): the 3x3 blur equals an fmaf
 * chain over taps in row-major order starting from 0, the C-channel Sobel
 * equals an fmaf chain over (kh, kw, c) with c innermost - both 100 % bitwise on
 * random inputs (see tests/test_oracle_golden.py and DESIGN.md).
 *
 * Parity status: pinned by tests/golden/{edge125,pgd_steps,losses,avmix_cw}.npz,
 * which tests/golden/make_golden.py generated from the reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* torch.sign: sign(+-0) = 0, sign(NaN) = 0 (utils/attacks.py:25) */
static inline float signf_(float g) { return (float)((g > 0.0f) - (g < 0.0f)); }

/* torch.clamp(x, lo, hi) = min(max(x, lo), hi); NaN propagates */
static inline float clampf_(float v, float lo, float hi) {
    if (v != v) return v;
    v = v < lo ? lo : v;
    return v > hi ? hi : v;
}
/* torch.max / torch.min (binary): NaN in either operand propagates */
static inline float maxf_(float a, float b) { return (a != a || b != b) ? (a + b) : (a > b ? a : b); }
static inline float minf_(float a, float b) { return (a != a || b != b) ? (a + b) : (a < b ? a : b); }

/* ---------------------------------------------------------------------------
 * PGD family (utils/attacks.py)
 * ------------------------------------------------------------------------- */

/* attacks.py:15-17  x = clamp(x0 + noise, lo, hi)  (noise = U(-eps,eps) or 0.001*randn) */
EXPORT void orc_pgd_init_f32(float *x, const float *x0, const float *noise, int64_t n, float lo, float hi) {
    for (int64_t i = 0; i < n; ++i) x[i] = clampf_(x0[i] + noise[i], lo, hi);
}

/* attacks.py:25-27 (same body :52-54, :82-84, :257-259, :298-300, :318-320, :353-355, :414-416, :466-468, :505-507)
 *   t = x + dir*alpha*sign(g); t = max(t, x0 - eps); t = min(t, x0 + eps); x = clamp(t, lo, hi)      in place   */
EXPORT void orc_pgd_step_f32(float *x, const float *g, const float *x0, int64_t n, float alpha, float eps,
                             float lo, float hi, int dir) {
    const float a = dir >= 0 ? alpha : -alpha;
    for (int64_t i = 0; i < n; ++i) {
        float t = x[i] + a * signf_(g[i]);
        t = maxf_(t, x0[i] - eps);
        t = minf_(t, x0[i] + eps);
        x[i] = clampf_(t, lo, hi);
    }
}

/* attacks.py:121-126  FGSM: one signed step, clamp, no eps projection */
EXPORT void orc_fgsm_step_f32(float *out, const float *x, const float *g, int64_t n, float alpha, float lo,
                              float hi, int dir) {
    const float a = dir >= 0 ? alpha : -alpha;
    for (int64_t i = 0; i < n; ++i) out[i] = clampf_(x[i] + a * signf_(g[i]), lo, hi);
}

/* ImageNet/free_imagenet/AT_free_imagenet_ddp.py:289-290   in1 = clamp(input + noise, lo, hi) */
EXPORT void orc_add_clamp_f32(float *out, const float *x, const float *delta, int64_t n, float lo, float hi) {
    for (int64_t i = 0; i < n; ++i) out[i] = clampf_(x[i] + delta[i], lo, hi);
}

/* AT_free_imagenet_ddp.py:305-307   delta[:B] += alpha*sign(g); delta.clamp_(-eps, eps)
 * (rows >= B are untouched and already inside the box, so only the first n elements change) */
EXPORT void orc_freeat_update_f32(float *delta, const float *g, int64_t n, float alpha, float eps) {
    for (int64_t i = 0; i < n; ++i) delta[i] = clampf_(delta[i] + alpha * signf_(g[i]), -eps, eps);
}

/* attacks.py:469-478  AVmixup vertex + per-sample mix.  The mixing weight is float64 (numpy Beta),
 * so the products promote to double and the result is cast back to float (`.to(torch.float)`):
 *   v = clamp(x0 + (x - x0)*gamma, 0, 1);  out = float( double(x0)*w + double(v)*(1 - w) )             */
EXPORT void orc_avmix_f32(float *out, const float *x, const float *x0, const double *wgt, int64_t B,
                          int64_t per, float gamma) {
    for (int64_t b = 0; b < B; ++b) {
        const double w = wgt[b];
        for (int64_t i = 0; i < per; ++i) {
            const int64_t k = b * per + i;
            float p = (x[k] - x0[k]) * gamma;
            float v = clampf_(x0[k] + p, 0.0f, 1.0f);
            out[k] = (float)((double)x0[k] * w + (double)v * (1.0 - w));
        }
    }
}

/* ---------------------------------------------------------------------------
 * CannyFilter_step125_1 (utils/core.py:509-585) + To_compare (core.py:329-358)
 * ------------------------------------------------------------------------- */

/* forward.  x [B,C,H,W]; g9/sx9/sy9 = 3x3 weights row-major (core.py:524-535);
 * outputs (each may be NULL): edge [B,1,H,W], mag (before the alpha mask), gx1, gy1 (after /C). */
EXPORT void orc_edge125_fwd_f32(const float *x, int B, int C, int H, int W, const float *g9, const float *sx9,
                                const float *sy9, float alpha, float high, float *edge, float *mag_out,
                                float *gx_out, float *gy_out) {
    float *b = (float *)malloc(sizeof(float) * (size_t)C * H * W);
    for (int n = 0; n < B; ++n) {
        const float *xn = x + (size_t)n * C * H * W;
        /* core.py:560-563  per channel: replicate-pad 1, 3x3 cross-correlation with G */
        for (int c = 0; c < C; ++c)
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < W; ++j) {
                    float acc = 0.0f;
                    for (int di = 0; di < 3; ++di)
                        for (int dj = 0; dj < 3; ++dj) {
                            int r = clampi(i + di - 1, 0, H - 1), s = clampi(j + dj - 1, 0, W - 1);
                            acc = fmaf(g9[di * 3 + dj], xn[((size_t)c * H + r) * W + s], acc);
                        }
                    b[((size_t)c * H + i) * W + j] = acc;
                }
        /* core.py:565-567  replicate-pad the blurred planes, Sobel x / y summed over channels */
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                float ax = 0.0f, ay = 0.0f;
                for (int di = 0; di < 3; ++di)
                    for (int dj = 0; dj < 3; ++dj) {
                        int r = clampi(i + di - 1, 0, H - 1), s = clampi(j + dj - 1, 0, W - 1);
                        for (int c = 0; c < C; ++c) {
                            float bv = b[((size_t)c * H + r) * W + s];
                            ax = fmaf(sx9[di * 3 + dj], bv, ax);
                            ay = fmaf(sy9[di * 3 + dj], bv, ay);
                        }
                    }
                /* core.py:570-571 */
                float gx1 = ax / (float)C, gy1 = ay / (float)C;
                float s2 = gx1 * gx1 + gy1 * gy1;
                float mag = sqrtf(s2);
                /* core.py:574-575 */
                float mag_a = (mag < alpha) ? 0.0f : mag;
                /* core.py:577-583 + To_compare.forward core.py:343-345 (NaN stays NaN) */
                float e = (mag_a > high) ? 1.0f : ((mag_a <= high) ? 0.0f : mag_a);
                size_t o = ((size_t)n * H + i) * W + j;
                if (edge) edge[o] = e;
                if (mag_out) mag_out[o] = mag;
                if (gx_out) gx_out[o] = gx1;
                if (gy_out) gy_out[o] = gy1;
            }
    }
    free(b);
}

/* adjoint of ReplicationPad2d(1): fold the padded plane gp [(H+2),(W+2)] onto [H,W], accumulating the
 * contributions of one output pixel in raster order of the padded positions, starting from 0. */
static void reppad1_adjoint(const float *gp, int H, int W, float *out) {
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            int p0 = (i == 0) ? 0 : i + 1, p1 = (i == H - 1) ? H + 1 : i + 1;
            int q0 = (j == 0) ? 0 : j + 1, q1 = (j == W - 1) ? W + 1 : j + 1;
            float acc = 0.0f;
            for (int p = p0; p <= p1; ++p)
                for (int q = q0; q <= q1; ++q) acc = acc + gp[(size_t)p * (W + 2) + q];
            out[(size_t)i * W + j] = acc;
        }
}

/* transposed 3x3 cross-correlation: gp[(H+2),(W+2)] (padded domain) from g[H,W] with weights w9:
 *   gp(p,q) = sum_{di,dj} w[di][dj] * g(p-di, q-dj)   (padded index p = image row + 1 - ... see fwd)
 * taps in row-major order, fmaf chain from 0; positions outside the image contribute nothing,
 * but zero-weight taps inside DO multiply (0*NaN = NaN, as in a GEMM-based dgrad). */
static void corr3_transpose_acc(const float *g, int H, int W, const float *w9, float *gp, int first) {
    for (int p = 0; p < H + 2; ++p)
        for (int q = 0; q < W + 2; ++q) {
            float acc = first ? 0.0f : gp[(size_t)p * (W + 2) + q];
            for (int di = 0; di < 3; ++di)
                for (int dj = 0; dj < 3; ++dj) {
                    int i = p - di, j = q - dj; /* fwd: out(i,j) reads padded (i+di, j+dj) */
                    if (i < 0 || i >= H || j < 0 || j >= W) continue;
                    acc = fmaf(w9[di * 3 + dj], g[(size_t)i * W + j], acc);
                }
            gp[(size_t)p * (W + 2) + q] = acc;
        }
}

/* backward.  u = dL/d(edge) [B,1,H,W]; gx_img [B,1,H,W] = the map every channel of dL/dx receives
 * (SURVEY 8(a'): the gradient is identical across input channels).
 *   To_compare.backward core.py:350-358; where core.py:575; pow core.py:571; /C core.py:570;
 *   conv / ReplicationPad2d adjoints of core.py:560-567.  0*inf = NaN at mag == 0 is KEPT. */
EXPORT void orc_edge125_bwd_f32(const float *x, const float *u, int B, int C, int H, int W, const float *g9,
                                const float *sx9, const float *sy9, float alpha, float high, float *gx_img) {
    size_t HW = (size_t)H * W, PW = (size_t)(H + 2) * (W + 2);
    float *mag = (float *)malloc(sizeof(float) * HW), *gx1 = (float *)malloc(sizeof(float) * HW);
    float *gy1 = (float *)malloc(sizeof(float) * HW), *ggx = (float *)malloc(sizeof(float) * HW);
    float *ggy = (float *)malloc(sizeof(float) * HW), *gp = (float *)malloc(sizeof(float) * PW);
    float *gb = (float *)malloc(sizeof(float) * HW);
    for (int n = 0; n < B; ++n) {
        orc_edge125_fwd_f32(x + (size_t)n * C * HW, 1, C, H, W, g9, sx9, sy9, alpha, high, NULL, mag, gx1, gy1);
        for (size_t k = 0; k < HW; ++k) {
            float m = mag[k];
            float mag_a = (m < alpha) ? 0.0f : m;
            float gm = u[(size_t)n * HW + k];
            if (mag_a <= high) gm = 0.0f;  /* core.py:356 */
            if (mag_a > 1.001f) gm = 0.0f; /* core.py:357 */
            if (m < alpha) gm = 0.0f;      /* where() backward, core.py:575 */
            float s2 = gx1[k] * gx1[k] + gy1[k] * gy1[k];
            float r = 1.0f / sqrtf(s2); /* pow(s2, -0.5) = 1/sqrt, two roundings */
            float gs = gm * (0.5f * r); /* d sqrt: grad * (0.5 * s2^-0.5) */
            ggx[k] = (gs * (2.0f * gx1[k])) / (float)C;
            ggy[k] = (gs * (2.0f * gy1[k])) / (float)C;
        }
        corr3_transpose_acc(ggx, H, W, sx9, gp, 1);
        corr3_transpose_acc(ggy, H, W, sy9, gp, 0);
        reppad1_adjoint(gp, H, W, gb);
        corr3_transpose_acc(gb, H, W, g9, gp, 1);
        reppad1_adjoint(gp, H, W, gx_img + (size_t)n * HW);
    }
    free(mag); free(gx1); free(gy1); free(ggx); free(ggy); free(gp); free(gb);
}

/* ---------------------------------------------------------------------------
 * EE front end (Tiny_ImageNet/models_tinyimagenet/resnet_EE.py:176-191, MNIST/models_mnist/Net2_EE.py:36-49)
 *   s = x_hfs + w*edge (edge broadcast over C);  x_in = clamp(s, 0, 1);  gate = (0 <= s <= 1)
 * ------------------------------------------------------------------------- */
EXPORT void orc_frontend_fwd_f32(const float *x, const float *x_hfs, int B, int C, int H, int W, const float *g9,
                                 const float *sx9, const float *sy9, float alpha, float high, float w,
                                 float *x_in, uint8_t *gate, float *edge_out) {
    size_t HW = (size_t)H * W;
    float *edge = (float *)malloc(sizeof(float) * B * HW);
    orc_edge125_fwd_f32(x, B, C, H, W, g9, sx9, sy9, alpha, high, edge, NULL, NULL, NULL);
    for (int n = 0; n < B; ++n)
        for (int c = 0; c < C; ++c)
            for (size_t k = 0; k < HW; ++k) {
                size_t o = ((size_t)n * C + c) * HW + k;
                float s = x_hfs[o] + w * edge[(size_t)n * HW + k];
                x_in[o] = clampf_(s, 0.0f, 1.0f);
                if (gate) gate[o] = (s >= 0.0f && s <= 1.0f) ? 1 : 0;
            }
    if (edge_out) memcpy(edge_out, edge, sizeof(float) * B * HW);
    free(edge);
}

/* backward of the front end.  g_in = dL/dx_in [B,C,H,W].
 *   g_hfs = g_in * gate                        (clamp backward)
 *   u     = w * ((g_hfs_0 + g_hfs_1) + ...)    (broadcast-add backward = sum over C in channel order, then *w)
 *   g_edge[B,1,H,W] = edge125_bwd(u)                                                                   */
EXPORT void orc_frontend_bwd_f32(const float *g_in, const uint8_t *gate, const float *x, int B, int C, int H,
                                 int W, const float *g9, const float *sx9, const float *sy9, float alpha,
                                 float high, float w, float *g_hfs, float *g_edge) {
    size_t HW = (size_t)H * W;
    float *u = (float *)malloc(sizeof(float) * B * HW);
    for (int n = 0; n < B; ++n)
        for (size_t k = 0; k < HW; ++k) {
            float acc = 0.0f;
            for (int c = 0; c < C; ++c) {
                size_t o = ((size_t)n * C + c) * HW + k;
                float v = gate[o] ? g_in[o] : 0.0f;
                g_hfs[o] = v;
                acc = (c == 0) ? v : acc + v;
            }
            u[(size_t)n * HW + k] = acc * w;
        }
    orc_edge125_bwd_f32(x, u, B, C, H, W, g9, sx9, sy9, alpha, high, g_edge);
    free(u);
}

/* ---------------------------------------------------------------------------
 * per-row losses (utils/attacks.py).  Row arithmetic in float like the reference's
 * log_softmax; the batch reduction is carried in double and rounded once.
 * ------------------------------------------------------------------------- */
static float row_logsumexp(const float *z, int K, float *mx_out) {
    float mx = z[0];
    for (int k = 1; k < K; ++k) mx = z[k] > mx ? z[k] : mx;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (double)expf(z[k] - mx);
    *mx_out = mx;
    return logf((float)s);
}

/* F.cross_entropy(logits, y, reduction='sum'|'mean')  attacks.py:23 (sum), :255 (mean);
 * dlogits = scale * (softmax - onehot), scale = 1 (sum) or 1/B (mean). */
EXPORT double orc_ce_f32(const float *logits, const int64_t *y, int B, int K, int mean, float *dlogits) {
    double tot = 0.0;
    const float scale = mean ? 1.0f / (float)B : 1.0f;
    for (int b = 0; b < B; ++b) {
        const float *z = logits + (size_t)b * K;
        float mx;
        float lse = row_logsumexp(z, K, &mx);
        tot += (double)(lse - (z[y[b]] - mx));
        if (dlogits)
            for (int k = 0; k < K; ++k) {
                float p = expf((z[k] - mx) - lse);
                dlogits[(size_t)b * K + k] = (p - (k == y[b] ? 1.0f : 0.0f)) * scale;
            }
    }
    return mean ? tot / B : tot;
}

/* nn.KLDivLoss('batchmean')(log_softmax(zq), softmax(zp))  attacks.py:375,:412,:426
 *   = (1/B) sum_b sum_k p*(log p - log q);  dq = (q - p)/B ;  dp_logits = p*((log p - log q) - KL_b)/B  */
EXPORT double orc_kl_f32(const float *zq, const float *zp, int B, int K, float *dzq, float *dzp) {
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {
        const float *q = zq + (size_t)b * K, *p = zp + (size_t)b * K;
        float mq, mp;
        float lq = row_logsumexp(q, K, &mq), lp = row_logsumexp(p, K, &mp);
        double klb = 0.0;
        for (int k = 0; k < K; ++k) {
            float logp = (p[k] - mp) - lp, logq = (q[k] - mq) - lq;
            float pk = expf(logp);
            if (pk > 0.0f) klb += (double)(pk * (logp - logq));
        }
        tot += klb;
        for (int k = 0; k < K; ++k) {
            float logp = (p[k] - mp) - lp, logq = (q[k] - mq) - lq;
            float pk = expf(logp), qk = expf(logq);
            if (dzq) dzq[(size_t)b * K + k] = (qk - pk) / (float)B;
            if (dzp) dzp[(size_t)b * K + k] = pk * ((logp - logq) - (float)klb) / (float)B;
        }
    }
    return tot / B;
}

/* -sum(log_softmax(z) * t) * scale  with a float64 soft target (attacks.py:462-463; driver loss
 * Tiny_ImageNet/experiments_tinyimagenet.py:292-293 with scale = 1/B).  dz = scale*(softmax*sum_k t - t). */
EXPORT double orc_softce_f64(const float *z, const double *t, int B, int K, double scale, double *dz) {
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {
        const float *zb = z + (size_t)b * K;
        float mx;
        float lse = row_logsumexp(zb, K, &mx);
        double ts = 0.0;
        for (int k = 0; k < K; ++k) {
            float lp = (zb[k] - mx) - lse;
            tot -= (double)lp * t[(size_t)b * K + k];
            ts += t[(size_t)b * K + k];
        }
        if (dz)
            for (int k = 0; k < K; ++k) {
                float lp = (zb[k] - mx) - lse;
                dz[(size_t)b * K + k] = scale * ((double)expf(lp) * ts - t[(size_t)b * K + k]);
            }
    }
    return tot * scale;
}

/* F.mse_loss(a, b) = mean over B*K  (attacks.py:269) ; da = 2(a-b)/(B*K), db = -da */
EXPORT double orc_mse_f32(const float *a, const float *b, int64_t n, float *da) {
    double tot = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        float d = a[i] - b[i];
        tot += (double)(d * d);
        if (da) da[i] = (2.0f * d) / (float)n;
    }
    return tot / (double)n;
}

/* utils/helper.py:39-55 accuracy: sorted top-k indices per row (ties: lower index first),
 * and the count of rows whose label is within the first k.  Integer results. */
EXPORT void orc_topk_i64(const float *logits, const int64_t *y, int B, int K, int k, int64_t *idx,
                         int64_t *correct_at_k) {
    for (int j = 0; j < k; ++j) correct_at_k[j] = 0;
    for (int b = 0; b < B; ++b) {
        const float *z = logits + (size_t)b * K;
        for (int j = 0; j < k; ++j) {
            int best = -1;
            for (int c = 0; c < K; ++c) {
                int taken = 0;
                for (int jj = 0; jj < j; ++jj) taken |= (idx[(size_t)b * k + jj] == c);
                if (taken) continue;
                if (best < 0 || z[c] > z[best]) best = c;
            }
            idx[(size_t)b * k + j] = best;
        }
        for (int j = 0; j < k; ++j)
            if (idx[(size_t)b * k + j] == y[b])
                for (int jj = j; jj < k; ++jj) correct_at_k[jj] += 1;
    }
}

/* ---------------------------------------------------------------------------
 * CannyFilter (full: NMS + STE double threshold + hysteresis)  utils/core.py:148-326
 *   dirs[16] = (drow, dcol) of the -1 tap of the 8 directional kernels (core.py:87-112).  The reference builds them
 *   with cv2 (absent here); the table passed in is DERIVED -> every result of these two functions is PARITY UNPINNED
 *   with respect to that table, and pinned to the reference's own forward/backward code given the table
 *   (tests/golden/canny_full_unpinned.npz).  Only the path every model takes is restated: low_threshold and
 *   high_threshold given, hysteresis=True.
 * ------------------------------------------------------------------------- */
static float magA_at(const float *magA, int H, int W, int i, int j) {
    return (i < 0 || i >= H || j < 0 || j >= W) ? 0.0f : magA[(size_t)i * W + j]; /* conv2d(padding=1): zero padding, core.py:268 */
}

/* per-image intermediate maps shared by forward and backward */
static void canny_maps(const float *x, int C, int H, int W, const float *g9, const float *sx9, const float *sy9, float alpha,
                       float high, const int *dirs, float *mag, float *gx1, float *gy1, float *magA, unsigned char *removed) {
    size_t HW = (size_t)H * W;
    orc_edge125_fwd_f32(x, 1, C, H, W, g9, sx9, sy9, alpha, high, NULL, mag, gx1, gy1);
    for (size_t k = 0; k < HW; ++k) magA[k] = (mag[k] < alpha) ? 0.0f : mag[k]; /* core.py:263-264 */
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            size_t p = (size_t)i * W + j;
            /* core.py:258-260, 270: orientation -> degrees -> multiples of 45 -> index mod 8 (all fp32) */
            float ori = atanf(gy1[p] / gx1[p]) * (float)(360.0 / 3.141592653589793) + 180.0f;
            float ori2 = rintf(ori / 45.0f) * 45.0f;
            float q = ori2 / 45.0f;
            float pidx = q - 8.0f * floorf(q / 8.0f); /* torch remainder */
            unsigned char rem = 0;
            if (pidx == pidx) { /* NaN orientation (gx = gy = 0) matches no direction: never suppressed */
                int kk = ((int)pidx) % 4;
                float d1 = magA[p] - magA_at(magA, H, W, i + dirs[2 * kk], j + dirs[2 * kk + 1]);
                float d2 = magA[p] - magA_at(magA, H, W, i + dirs[2 * (kk + 4)], j + dirs[2 * (kk + 4) + 1]);
                float mn = (d1 != d1 || d2 != d2) ? (d1 + d2) : (d1 < d2 ? d1 : d2);
                rem = !(mn > 0.0f); /* core.py:285-290 */
            }
            removed[p] = rem;
        }
}

EXPORT void orc_canny_fwd_f32(const float *x, int B, int C, int H, int W, const float *g9, const float *sx9, const float *sy9,
                              float alpha, float low, float high, const int *dirs, float *out) {
    size_t HW = (size_t)H * W;
    float *mag = malloc(sizeof(float) * HW), *gx1 = malloc(sizeof(float) * HW), *gy1 = malloc(sizeof(float) * HW);
    float *magA = malloc(sizeof(float) * HW), *t2 = malloc(sizeof(float) * HW), *hb = malloc(sizeof(float) * HW);
    unsigned char *rem = malloc(HW);
    for (int n = 0; n < B; ++n) {
        canny_maps(x + (size_t)n * C * HW, C, H, W, g9, sx9, sy9, alpha, high, dirs, mag, gx1, gy1, magA, rem);
        for (size_t k = 0; k < HW; ++k) {
            float t = rem[k] ? 0.0f : magA[k];
            /* safeSign(t - thr): 0 -> -1, NaN -> sign 0 -> -1   (core.py:115-118, 299-310) */
            float lowb = ((t - low) > 0.0f) ? 1.0f : 0.0f, highb = ((t - high) > 0.0f) ? 1.0f : 0.0f;
            hb[k] = highb;
            t2[k] = lowb * 0.5f + highb * 0.5f; /* core.py:315 */
        }
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                float acc = 0.0f;
                for (int di = -1; di <= 1; ++di)
                    for (int dj = -1; dj <= 1; ++dj) {
                        int r = i + di, s = j + dj;
                        float v = (r < 0 || r >= H || s < 0 || s >= W) ? 0.0f : t2[(size_t)r * W + s];
                        acc = fmaf(1.25f, v, acc);
                    }
                size_t p = (size_t)i * W + j;
                float weak_is_high = (acc > 1.0f && t2[p] == 0.5f) ? 1.0f : 0.0f; /* core.py:319-320 */
                out[(size_t)n * HW + p] = hb[p] + weak_is_high;                  /* core.py:321 */
            }
    }
    free(mag); free(gx1); free(gy1); free(magA); free(t2); free(hb); free(rem);
}

/* backward of the above.  Only `high` carries gradient (low and the hysteresis terms enter through comparisons):
 *   g_t = (u / 2) where |t - high| <= 1.001 (BinaryConnectDeterministic.backward core.py:138-145), 0 at suppressed
 *   pixels (in-place assignment core.py:290), then the alpha mask and the magnitude / Sobel / blur adjoints as in
 *   orc_edge125_bwd_f32 (0 * inf = NaN at mag == 0 kept). */
EXPORT void orc_canny_bwd_f32(const float *x, const float *u, int B, int C, int H, int W, const float *g9, const float *sx9,
                              const float *sy9, float alpha, float low, float high, const int *dirs, float *gx_img) {
    (void)low;
    size_t HW = (size_t)H * W, PW = (size_t)(H + 2) * (W + 2);
    float *mag = malloc(sizeof(float) * HW), *gx1 = malloc(sizeof(float) * HW), *gy1 = malloc(sizeof(float) * HW);
    float *magA = malloc(sizeof(float) * HW), *ggx = malloc(sizeof(float) * HW), *ggy = malloc(sizeof(float) * HW);
    float *gp = malloc(sizeof(float) * PW), *gb = malloc(sizeof(float) * HW);
    unsigned char *rem = malloc(HW);
    for (int n = 0; n < B; ++n) {
        canny_maps(x + (size_t)n * C * HW, C, H, W, g9, sx9, sy9, alpha, high, dirs, mag, gx1, gy1, magA, rem);
        for (size_t k = 0; k < HW; ++k) {
            float t = rem[k] ? 0.0f : magA[k];
            float gm = u[(size_t)n * HW + k] / 2.0f;
            if (fabsf(t - high) > 1.001f) gm = 0.0f;
            if (rem[k]) gm = 0.0f;
            if (mag[k] < alpha) gm = 0.0f;
            float s2 = gx1[k] * gx1[k] + gy1[k] * gy1[k];
            float r = 1.0f / sqrtf(s2);
            float gs = gm * (0.5f * r);
            ggx[k] = (gs * (2.0f * gx1[k])) / (float)C;
            ggy[k] = (gs * (2.0f * gy1[k])) / (float)C;
        }
        corr3_transpose_acc(ggx, H, W, sx9, gp, 1);
        corr3_transpose_acc(ggy, H, W, sy9, gp, 0);
        reppad1_adjoint(gp, H, W, gb);
        corr3_transpose_acc(gb, H, W, g9, gp, 1);
        reppad1_adjoint(gp, H, W, gx_img + (size_t)n * HW);
    }
    free(mag); free(gx1); free(gy1); free(magA); free(ggx); free(ggy); free(gp); free(gb); free(rem);
}

/* ---------------------------------------------------------------------------
 * CannyFilter_BPDA  utils/core.py:386-505 (AWP configs only), low/high thresholds given, hysteresis=True.
 *   Differences from CannyFilter: no alpha mask; NMS multiplies by the keep mask (core.py:480) instead of assigning 0;
 *   thresholds through To_compare (core.py:329-358: x <= thr -> 0, x > thr -> 1, NaN kept; backward passes where
 *   thr < x <= 1.001), hysteresis through To_eq (core.py:361-382) and To_compare(conv, 1).  Same DERIVED direction
 *   table -> PARITY UNPINNED w.r.t. that table, pinned to the reference's own code given the table
 *   (tests/golden/canny_full_unpinned.npz, CannyFilter_BPDA entries).
 * ------------------------------------------------------------------------- */
static float to_compare(float v, float thr) { return (v <= thr) ? 0.0f : ((v > thr) ? 1.0f : v); }

static void bpda_maps(const float *mag, const unsigned char *rem, int H, int W, float low, float high, float *thin, float *hb,
                      float *t2, float *w0) {
    size_t HW = (size_t)H * W;
    for (size_t k = 0; k < HW; ++k) {
        thin[k] = rem[k] ? mag[k] * 0.0f : mag[k]; /* core.py:480 */
        hb[k] = to_compare(thin[k], high);
        t2[k] = to_compare(thin[k], low) * 0.5f + hb[k] * 0.5f; /* core.py:493 */
    }
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            float acc = 0.0f;
            for (int di = -1; di <= 1; ++di)
                for (int dj = -1; dj <= 1; ++dj) {
                    int r = i + di, s = j + dj;
                    float v = (r < 0 || r >= H || s < 0 || s >= W) ? 0.0f : t2[(size_t)r * W + s];
                    acc = fmaf(1.25f, v, acc);
                }
            w0[(size_t)i * W + j] = acc; /* core.py:500 */
        }
}

EXPORT void orc_canny_bpda_fwd_f32(const float *x, int B, int C, int H, int W, const float *g9, const float *sx9, const float *sy9,
                                   float low, float high, const int *dirs, float *out) {
    size_t HW = (size_t)H * W;
    float *mag = malloc(sizeof(float) * HW), *gx1 = malloc(sizeof(float) * HW), *gy1 = malloc(sizeof(float) * HW);
    float *magA = malloc(sizeof(float) * HW), *thin = malloc(sizeof(float) * HW), *hb = malloc(sizeof(float) * HW);
    float *t2 = malloc(sizeof(float) * HW), *w0 = malloc(sizeof(float) * HW);
    unsigned char *rem = malloc(HW);
    for (int n = 0; n < B; ++n) {
        canny_maps(x + (size_t)n * C * HW, C, H, W, g9, sx9, sy9, 0.0f, high, dirs, mag, gx1, gy1, magA, rem);
        bpda_maps(mag, rem, H, W, low, high, thin, hb, t2, w0);
        for (size_t k = 0; k < HW; ++k) {
            float weak = (t2[k] == 0.5f) ? 1.0f : 0.0f;                           /* To_eq, core.py:498 */
            out[(size_t)n * HW + k] = hb[k] * 1.0f + (to_compare(w0[k], 1.0f) * weak) * 1.0f; /* core.py:501-504 */
        }
    }
    free(mag); free(gx1); free(gy1); free(magA); free(thin); free(hb); free(t2); free(w0); free(rem);
}

EXPORT void orc_canny_bpda_bwd_f32(const float *x, const float *u, int B, int C, int H, int W, const float *g9, const float *sx9,
                                   const float *sy9, float low, float high, const int *dirs, float *gx_img) {
    size_t HW = (size_t)H * W, PW = (size_t)(H + 2) * (W + 2);
    float *mag = malloc(sizeof(float) * HW), *gx1 = malloc(sizeof(float) * HW), *gy1 = malloc(sizeof(float) * HW);
    float *magA = malloc(sizeof(float) * HW), *thin = malloc(sizeof(float) * HW), *hb = malloc(sizeof(float) * HW);
    float *t2 = malloc(sizeof(float) * HW), *w0 = malloc(sizeof(float) * HW), *gw0 = malloc(sizeof(float) * HW);
    float *ggx = malloc(sizeof(float) * HW), *ggy = malloc(sizeof(float) * HW);
    float *gp = malloc(sizeof(float) * PW), *gb = malloc(sizeof(float) * HW);
    unsigned char *rem = malloc(HW);
    for (int n = 0; n < B; ++n) {
        const float *un = u + (size_t)n * HW;
        canny_maps(x + (size_t)n * C * HW, C, H, W, g9, sx9, sy9, 0.0f, high, dirs, mag, gx1, gy1, magA, rem);
        bpda_maps(mag, rem, H, W, low, high, thin, hb, t2, w0);
        /* out = high + weak_1 * weak.  d/d weak_0 through To_compare(., 1): passes only where 1 < weak_0 <= 1.001 */
        for (size_t k = 0; k < HW; ++k) {
            float weak = (t2[k] == 0.5f) ? 1.0f : 0.0f;
            float g = un[k] * weak;
            if (w0[k] <= 1.0f) g = 0.0f;
            if (w0[k] > 1.001f) g = 0.0f;
            gw0[k] = g;
        }
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                size_t k = (size_t)i * W + j;
                /* d/d t2: To_eq.backward (core.py:373-382) + the transposed hysteresis convolution */
                float g_eq = un[k] * to_compare(w0[k], 1.0f);
                if (t2[k] != 0.5f) g_eq = 0.0f;
                float g_cv = 0.0f;
                for (int di = -1; di <= 1; ++di)
                    for (int dj = -1; dj <= 1; ++dj) {
                        int r = i - di, s = j - dj;
                        float v = (r < 0 || r >= H || s < 0 || s >= W) ? 0.0f : gw0[(size_t)r * W + s];
                        g_cv = fmaf(1.25f, v, g_cv);
                    }
                float g_t2 = g_eq + g_cv;
                float g_low = g_t2 * 0.5f, g_high = un[k] * 1.0f + g_t2 * 0.5f;
                if (thin[k] <= low) g_low = 0.0f;
                if (thin[k] > 1.001f) g_low = 0.0f;
                if (thin[k] <= high) g_high = 0.0f;
                if (thin[k] > 1.001f) g_high = 0.0f;
                float gm = g_low + g_high;
                if (rem[k]) gm = gm * 0.0f; /* mul by the keep mask, core.py:480 */
                float s2 = gx1[k] * gx1[k] + gy1[k] * gy1[k];
                float r = 1.0f / sqrtf(s2);
                float gs = gm * (0.5f * r);
                ggx[k] = (gs * (2.0f * gx1[k])) / (float)C;
                ggy[k] = (gs * (2.0f * gy1[k])) / (float)C;
            }
        corr3_transpose_acc(ggx, H, W, sx9, gp, 1);
        corr3_transpose_acc(ggy, H, W, sy9, gp, 0);
        reppad1_adjoint(gp, H, W, gb);
        corr3_transpose_acc(gb, H, W, g9, gp, 1);
        reppad1_adjoint(gp, H, W, gx_img + (size_t)n * HW);
    }
    free(mag); free(gx1); free(gy1); free(magA); free(thin); free(hb); free(t2); free(w0); free(gw0);
    free(ggx); free(ggy); free(gp); free(gb); free(rem);
}
