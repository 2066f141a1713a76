/*
 * eeadv.h - C ABI of libeeadv.so: the MI355X (gfx950) hot path of Edge-Enhancement adversarial training.
 *
 * The reference (Aiqz/Edge-Enhancement) is pure Python on PyTorch and has no FFI of its own; the
 * "interface" each entry point replaces is therefore the group of ATen launches behind a few lines of
 * utils/attacks.py / utils/core.py, cited per function (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd or a torch ROCm tensor's data_ptr()) unless a
 *     parameter is documented "host"; tensors are dense, row-major NCHW / [B,K];
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is enqueued
 *     asynchronously on it, nothing synchronises, so every call is hipGraph-capturable;
 *   - return value: 0 = success, < 0 = EE_ERR_* (argument check failed, nothing was launched),
 *     > 0 = the hipError_t reported by the launch;
 *   - fp32 arithmetic follows the operation order of oracle/ee_oracle.c bit for bit (built with
 *     -ffp-contract=off; IEEE division and square root), including sign(NaN) = sign(0) = 0 and the
 *     0*inf = NaN gradient of the edge filter at zero magnitude.
 */
#ifndef EEADV_H
#define EEADV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EE_OK 0
#define EE_ERR_NULL (-1)        /* a required pointer is NULL */
#define EE_ERR_SHAPE (-2)       /* a size / shape argument is out of range */
#define EE_ERR_UNSUPPORTED (-3) /* valid request this build has no kernel for (e.g. C > 4) */
#define EE_ERR_ALIGN (-4)       /* pointer not aligned to the element size */

#define EEADV_ABI_VERSION 1

int ee_abi_version(void);
/* static string for an EE_ERR_* / hipError_t code returned by any entry point */
const char *ee_strerror(int code);
/* name of the device the library would launch on, or NULL (host pointer to a static buffer) */
const char *ee_device_name(void);

/* ---------------------------------------------------------------------------------------------
 * PGD / FGSM / free-AT element-wise updates                              (utils/attacks.py)
 * ------------------------------------------------------------------------------------------- */

/* attacks.py:15-17 (also :42-44, :454-456, :492-494):  x = clamp(x0 + noise, lo, hi).
 * `noise` is caller-supplied (U(-eps,eps) for PGD, 0.001*randn for TRADES/ALP :250,:406 with
 * lo = -inf, hi = +inf because those initialisations are not clamped). */
int ee_pgd_init_f32(float *x, const float *x0, const float *noise, int64_t n, float lo, float hi, void *stream);

/* same, with the noise drawn on the device: Philox4x32-10 keyed by (seed, offset), element i uses
 * counter offset + i/4.  dist 0: U(-scale, scale)  (zeros_like(x).uniform_(-eps, eps), attacks.py:16);
 * dist 1: scale * N(0,1)  (0.001 * randn, attacks.py:250).  Streams differ from torch's generators,
 * so parity tests use ee_pgd_init_f32 with injected noise. */
int ee_pgd_init_rng_f32(float *x, const float *x0, int64_t n, float scale, int dist, uint64_t seed,
                        uint64_t offset, float lo, float hi, void *stream);

/* attacks.py:25-27 (same body :52-54 :82-84 :257-259 :298-300 :318-320 :353-355 :414-416 :466-468
 * :505-507), in place on x:
 *     t = x + dir*alpha*sign(g);  t = max(t, x0 - eps);  t = min(t, x0 + eps);  x = clamp(t, lo, hi)
 * dir = +1 ascent (untargeted), -1 descent (targeted).  16 B of HBM traffic per element. */
int ee_pgd_step_f32(float *x, const float *g, const float *x0, int64_t n, float alpha, float eps, float lo,
                    float hi, int dir, void *stream);

/* attacks.py:389-400 (Trades.PGD_L2), in place on x [B, per_sample]: per sample b
 *     gn = sqrt(mean(g_b^2)) + 1e-8;  t = x_b + step * (g_b / gn);  d = t - x0_b;  dn = sqrt(mean(d^2));
 *     if dn > eps: d *= eps / dn;     x_b = clamp(x0_b + d, lo, hi)
 * (l2_norm is the root of the MEAN of squares, attacks.py:360-366).  The two means are accumulated in double and
 * rounded to fp32 once; ATen's fp32 sum order is not reproduced - results agree to ~1e-7, not bit for bit. */
int ee_l2_step_f32(float *x, const float *g, const float *x0, int64_t B, int64_t per_sample, float step, float eps,
                   float lo, float hi, void *stream);

/* attacks.py:121-126 FGSM:  out = clamp(x + dir*alpha*sign(g), lo, hi)   (no eps projection) */
int ee_fgsm_step_f32(float *out, const float *x, const float *g, int64_t n, float alpha, float lo, float hi,
                     int dir, void *stream);

/* ImageNet/free_imagenet/AT_free_imagenet_ddp.py:289-290:  out = clamp(x + delta, lo, hi) */
int ee_add_clamp_f32(float *out, const float *x, const float *delta, int64_t n, float lo, float hi,
                     void *stream);

/* AT_free_imagenet_ddp.py:305-307:  delta[0:n] = clamp(delta[0:n] + alpha*sign(g), -eps, eps) in place.
 * The reference clamps the whole persistent buffer; rows beyond the batch are unchanged and already
 * inside the box, so touching only the n live elements gives the same buffer. */
int ee_freeat_update_f32(float *delta, const float *g, int64_t n, float alpha, float eps, void *stream);

/* the same update when the caller holds g_in1 = dL/d(in1), in1 = clamp(x + delta, 0, 1) (AT_free_imagenet_ddp.py:289-290):
 * the clamp's gradient mask (0 <= x + delta <= 1) is applied in the kernel, dL/d(delta) never exists as a tensor. */
int ee_freeat_update_masked_f32(float *delta, const float *g_in1, const float *x, int64_t n, float alpha, float eps, void *stream);

/* attacks.py:469-479 AVmixup vertex + per-sample mix.  wgt = float64[B] (numpy Beta(1,1) weights):
 *     v = clamp(x0 + (x - x0)*gamma, 0, 1);   out = float(double(x0)*w_b + double(v)*(1 - w_b)) */
int ee_avmix_f32(float *out, const float *x, const float *x0, const double *wgt, int64_t B, int64_t per_sample,
                 float gamma, void *stream);

/* attacks.py:475-478  mixed soft labels, float64 out[B,K]:
 *     y = ls(onehot, l1)*w_b + ls(onehot, l2)*(1 - w_b),  ls(o,f) = o*f + (o-1)*((f-1)/(K-1))  (:444-445, fp32) */
int ee_avmix_labels_f64(double *out, const int64_t *labels, const double *wgt, int64_t B, int64_t K, float lambda1,
                        float lambda2, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Edge filter CannyFilter_step125_1 (utils/core.py:509-585) + To_compare (core.py:329-358)
 *   weights27 = HOST pointer to 27 floats: Gaussian 3x3, Sobel-x 3x3, Sobel-y 3x3, row-major; copied into the kernel arguments
 *   (core.py:524-535; the caller builds them once with get_gaussian_kernel / get_sobel_kernel).
 * ------------------------------------------------------------------------------------------- */

/* forward: x[B,C,H,W] -> edge[B,1,H,W] in {0,1} (fp32).  mag (nullable) receives the gradient
 * magnitude before the alpha mask.  Traffic (C+1)*4 B per pixel. */
int ee_edge125_fwd_f32(const float *x, int B, int C, int H, int W, const float *weights27, float alpha, float high,
                       float *edge, float *mag, void *stream);

/* backward: u = dL/d(edge) [B,1,H,W] -> g_img[B,1,H,W], the map that EVERY input channel of dL/dx
 * receives (the adjoint is identical across channels).  Recomputes the forward from x. */
int ee_edge125_bwd_f32(const float *x, const float *u, int B, int C, int H, int W, const float *weights27,
                       float alpha, float high, float *g_img, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused EE front end   x_in = clamp(x_hfs + w*edge(x), 0, 1)
 *   (Tiny_ImageNet/models_tinyimagenet/resnet_EE.py:176-191, resnet_EE_square.py:187-206,
 *    MNIST/models_mnist/Net2_EE.py:36-49, Net2_EE_square.py:48-63)
 * ------------------------------------------------------------------------------------------- */

/* forward: reads x, x_hfs [B,C,H,W]; writes x_in [B,C,H,W] and gate [B,C,H,W] (uint8: 1 where
 * 0 <= x_hfs + w*edge <= 1, i.e. where clamp passes the gradient).  edge (nullable) [B,1,H,W].
 * 12*C B (+C gate bytes) per pixel. */
int ee_frontend_fwd_f32(const float *x, const float *x_hfs, int B, int C, int H, int W, const float *weights27,
                        float alpha, float high, float w, float *x_in, uint8_t *gate, float *edge, void *stream);

/* backward: g_in = dL/dx_in.  Writes g_hfs = g_in*gate [B,C,H,W] (the gradient entering the
 * low-pass branch) and g_edge [B,1,H,W] (the edge-branch gradient of every channel of x):
 *     u = w * sum_c g_hfs_c ;  g_edge = edge125_bwd(u).                                        */
int ee_frontend_bwd_f32(const float *g_in, const uint8_t *gate, const float *x, int B, int C, int H, int W,
                        const float *weights27, float alpha, float high, float w, float *g_hfs, float *g_edge,
                        void *stream);

/* The same pair keeping the channel-mean Sobel responses gx, gy [B,1,H,W] between the two: the forward writes them (8 more
 * bytes per pixel), the backward then needs neither x nor the blur / Sobel recomputation (about 60 % of the recomputing
 * kernel's instructions) and is bit-identical to ee_frontend_bwd_f32. */
int ee_frontend_fwd_save_f32(const float *x, const float *x_hfs, int B, int C, int H, int W, const float *weights27,
                             float alpha, float high, float w, float *x_in, uint8_t *gate, float *edge, float *gx, float *gy,
                             void *stream);
int ee_frontend_bwd_saved_f32(const float *g_in, const uint8_t *gate, const float *gx, const float *gy, int B, int C, int H,
                              int W, const float *weights27, float alpha, float high, float w, float *g_hfs, float *g_edge,
                              void *stream);

/* ---------------------------------------------------------------------------------------------
 * Full CannyFilter (utils/core.py:148-326): blur, Sobel, magnitude, alpha mask, orientation-quantised non-maximum
 * suppression, straight-through double threshold, hysteresis - the path every model takes (low and high thresholds
 * given, hysteresis=True).  dirs16 = HOST int[16]: (drow, dcol) of the -1 tap of the 8 directional kernels
 * (core.py:87-112; the reference builds them with cv2, the caller passes the derived table -> parity unpinned).
 * ------------------------------------------------------------------------------------------- */

/* forward.  x_in == NULL: edge[B,1,H,W] only.  x_in != NULL: fused front end, x_in = clamp(x_hfs + w*edge, 0, 1) and
 * gate (nullable) as in ee_frontend_fwd_f32; edge then optional. */
int ee_canny_fwd_f32(const float *x, const float *x_hfs, int B, int C, int H, int W, const float *weights27, const int *dirs16,
                     float alpha, float low, float high, float w, float *edge, float *x_in, uint8_t *gate, void *stream);

/* backward.  g_in == NULL: u = dL/d(edge) [B,1,H,W] -> g_img[B,1,H,W].  g_in != NULL (fused front end): u is formed from
 * g_in * gate as in ee_frontend_bwd_f32 and g_hfs = g_in * gate is written too. */
int ee_canny_bwd_f32(const float *x, const float *u, const float *g_in, const uint8_t *gate, int B, int C, int H, int W,
                     const float *weights27, const int *dirs16, float alpha, float low, float high, float w, float *g_img,
                     float *g_hfs, void *stream);

/* last stage of the attack step for EE models, fused: dL/dx = g_lp + g_edge (g_edge broadcast over C),
 * then the PGD update of ee_pgd_step_f32 on x - the gradient itself never reaches HBM.
 * g_lp [B,C,H,W] = gradient arriving through the low-pass branch; g_edge [B,1,H,W]. */
int ee_pgd_step_bcast_f32(float *x, const float *g_lp, const float *g_edge, const float *x0, int B, int C,
                          int64_t hw, float alpha, float eps, float lo, float hi, int dir, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Per-row losses on logits [B,K]  (K <= 65536; one 64-lane wavefront per row)
 *   row_loss: float64[B] per-row values (nullable when only the gradient is wanted)
 *   ee_reduce_rows_f64 folds them in a fixed order -> bit-reproducible scalars.
 * ------------------------------------------------------------------------------------------- */

/* F.cross_entropy incl. label smoothing (attacks.py:23 sum, :255 mean, :89-99 LabelSmoothLoss):
 *     row_loss_b = -sum_k w_bk log_softmax(z_b)_k,  w = smoothing/(K-1) off target, 1-smoothing on it
 *     dz = gscale * (softmax(z) - w)           gscale = 1 ('sum') or 1/B ('mean')            */
int ee_ce_f32(const float *logits, const int64_t *labels, int B, int K, float smoothing, float gscale,
              double *row_loss, float *dlogits, void *stream);
/* The last two layers of the MNIST classifier and the loss behind them, gradient only (MNIST/models_mnist/Net2.py:27-28 + attacks.py:23):
 * d CrossEntropyLoss(fc2(relu(z1)), labels) / d z1 in one launch.  z1 [B,Hd] = fc1's output, w2 [K,Hd], b2 [K] or NULL, labels [B] ->
 * dz1 [B,Hd] (and the logits [B,K] if logits_out != NULL); gscale = 1 ("sum") or 1/B ("mean").  K <= 64, Hd <= 8192. */
int ee_fc_ce_grad_f32(const float *z1, const float *w2, const float *b2, const int64_t *labels, float *dz1, float *logits_out, int B, int Hd,
                      int K, float gscale, void *stream);

/* nn.KLDivLoss('batchmean')(log_softmax(zq), softmax(zp))  (attacks.py:375, :412, :426):
 *     row_loss_b = sum_k p(log p - log q);  dzq = gscale*(q - p);  dzp = gscale*p*((log p - log q) - row_loss_b)
 * dzq / dzp nullable. */
int ee_kl_f32(const float *zq, const float *zp, int B, int K, float gscale, double *row_loss, float *dzq,
              float *dzp, void *stream);

/* -sum_k log_softmax(z)_k * t_k with float64 soft targets (attacks.py:462-463;
 * Tiny_ImageNet/experiments_tinyimagenet.py:292-293): dz (float64, nullable) = gscale*(softmax*sum_k t - t) */
int ee_softce_f64(const float *z, const double *t, int B, int K, double gscale, double *row_loss, double *dz,
                  void *stream);

/* F.mse_loss(a, b) pieces (attacks.py:269): partial[nblk] float64 block sums of (a-b)^2 (nullable),
 * da = gscale*(a - b) (nullable; db = -da).  nblk = ee_mse_num_partials(n). */
int ee_mse_f32(const float *a, const float *b, int64_t n, float gscale, double *partial, float *da, void *stream);
int64_t ee_mse_num_partials(int64_t n);

/* out[0] (float64) = scale * (rows[0] + rows[1] + ...) in a fixed order (single workgroup) */
int ee_reduce_rows_f64(const double *rows, int64_t n, double scale, double *out, void *stream);

/* utils/helper.py:39-55 accuracy / attacks.py:131-132 predict_from_logits:
 * idx[B,k] = sorted top-k class indices (ties: lower index first), correct[k] (int64, zeroed by the
 * call) = number of rows whose label is among the first j+1 indices.  labels nullable. */
int ee_topk_i64(const float *logits, const int64_t *labels, int B, int K, int k, int64_t *idx, int64_t *correct,
                void *stream);

/* CannyFilter_BPDA (utils/core.py:386-505; AWP configs): no alpha mask, NMS by multiplication, thresholds through
 * To_compare (core.py:329-358), hysteresis through To_eq (core.py:361-382).  thresholds given, hysteresis=True.
 *   forward: edge, thin (the thinned magnitude), t2 (the {0, .5, 1} threshold map) [B,1,H,W]; thin / t2 feed the backward.
 *   backward: u = dL/d(edge) -> g_img [B,1,H,W] (the map every input channel receives); g_thin [B,1,H,W] is scratch.
 *   NaN pixels (NaN / inf input only) are not reproduced.  dirs16 as for ee_canny_fwd_f32 (HOST). */
int ee_canny_bpda_fwd_f32(const float *x, int B, int C, int H, int W, const float *weights27, const int *dirs16, float low,
                          float high, float *edge, float *thin, float *t2, void *stream);
int ee_canny_bpda_bwd_f32(const float *x, const float *u, const float *thin, const float *t2, int B, int C, int H, int W,
                          const float *weights27, const int *dirs16, float low, float high, float *g_thin, float *g_img,
                          void *stream);

/* ---------------------------------------------------------------------------------------------
 * Add_Square (utils/core.py:589-655), element-wise given its random draws
 *   stripe[B,C,W]  = sign(2*rand-1) of core.py:637 ; sq_sign[nq,C] = sign draws of core.py:648 ;
 *   sq_pos[nq] (int64, device: the `.long()` draw) and sq_size[nq] (int32, device) = vh and s of core.py:644-645.
 * ------------------------------------------------------------------------------------------- */
int ee_add_square_fwd_f32(const float *x, int B, int C, int H, int W, float eps, const float *stripe,
                          const float *sq_sign, const int64_t *sq_pos, const int32_t *sq_size, int nq, float *out,
                          void *stream);
/* the draws themselves, one launch: stripe[n_stripe] = sign(2u-1) (core.py:637), sq_pos[q] = long((h - sq_size[q]) * u)
 * (core.py:645), sq_sign[nq*C] = sign(2u-1) (core.py:648), u ~ U[0,1) from Philox4x32-10.  state = DEVICE uint64[2]
 * {seed, offset}; the kernel advances the offset itself, so replays of a captured graph draw fresh numbers. */
int ee_square_draw_f32(float *stripe, int64_t n_stripe, int64_t *sq_pos, float *sq_sign, const int32_t *sq_size, int nq,
                       int C, int h, uint64_t *state, void *stream);
/* backward: g_x = g_out * d(out)/d(x), the derivative autograd assigns (ties of max/min split 1/2) */
int ee_add_square_bwd_f32(const float *g_out, const float *x, int B, int C, int H, int W, float eps,
                          const float *stripe, const float *sq_sign, const int64_t *sq_pos, const int32_t *sq_size,
                          int nq, float *g_x, void *stream);

/* ---------------------------------------------------------------------------------------------
 * HighFreqSuppress (utils/core.py:15-55) as a fixed real, self-adjoint linear operator, one workgroup per plane
 *   in / out [B,C,H,W], H, W <= 64 (EE_ERR_UNSUPPORTED beyond: callers use the dense two-contraction form there).
 *   tables: DEVICE buffer of ee_hfs_table_floats(H, W, NU, NV) floats built by the host (eeadv/hfs.py):
 *       CwT[NVp][Wp], SwT[NVp][Wp]   cos / sin(2 pi v w / W) for the NV kept half-spectrum column frequencies
 *       ChT[NUp][Hp+4], ShT[NUp][Hp+4], ChN[H][NUp], ShN[H][NUp]   cos / sin(2 pi u h / H) for the NU kept row frequencies
 *       dv[NVp] = kappa_v / W          (NUp, NVp, Hp, Wp: rounded up to multiples of 4, zero padded)
 *   inv_h = 1/H.  The operator equals its own adjoint, so the same call is the backward.
 *   sq_mode 0: y = F(in).  1: y = F(add_square(in))  (core.py:636-655 fused into the load).
 *   2: y = F(in) * d add_square / dx evaluated at sq_x  (the backward of mode 1).  Add_Square draws as in
 *   ee_add_square_fwd_f32.  8 B of HBM traffic per element.
 * ------------------------------------------------------------------------------------------- */
int ee_hfs_table_floats(int H, int W, int NU, int NV);
int ee_hfs_f32(const float *in, float *out, int B, int C, int H, int W, const float *tables, int NU, int NV, float inv_h,
               int sq_mode, const float *sq_x, float eps, const float *stripe, const float *sq_sign, const int64_t *sq_pos,
               const int32_t *sq_size, int nq, void *stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d fused with the residual add and the ReLU that follow it in the reference's ResNet blocks
 * (Tiny_ImageNet/models_tinyimagenet/resnet.py:44-59, :90-110: out = relu(bn(conv(x)) [+ identity])).
 *   x, residual (nullable), y: [B,C,HW] fp32 (NCHW, HW = H*W); gamma / beta (nullable = 1 / 0), running_*: [C].
 *   training != 0: batch statistics (two-pass, biased variance for the output, unbiased for running_var), saved to
 *     save_mean / save_invstd [C]; running_* (nullable) updated with `momentum` as nn.BatchNorm2d does.
 *   training == 0: running statistics.      relu != 0: y = max(., 0) (NaN propagates).
 * backward: dz = relu ? dy * (y > 0) : dy ; dresidual (nullable) = dz ; dgamma / dbeta (nullable) = sum dz*xhat / sum dz ;
 *   dx (nullable) = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)) in training mode, gamma*invstd*dz otherwise.
 * Fixed-order reductions: results are bit-reproducible run to run.  One workgroup per channel; a channel too large for one
 * workgroup's registers (B*HW > 28672 values) is split across workgroups in two launches when `workspace` (nullable, DEVICE,
 * ee_bn_workspace_floats(B, C, HW) floats, contents undefined on entry and exit) is given.
 * ------------------------------------------------------------------------------------------- */
int ee_bn_workspace_floats(int B, int C, int HW);
int ee_bn_act_fwd_f32(const float *x, const float *residual, const float *gamma, const float *beta, float *running_mean,
                      float *running_var, float momentum, float eps, int training, int relu, float *y, float *save_mean,
                      float *save_invstd, float *workspace, int B, int C, int HW, void *stream);
int ee_bn_act_bwd_f32(const float *dy, const float *y, const float *x, const float *gamma, const float *save_mean,
                      const float *save_invstd, const float *running_mean, const float *running_var, float eps, int training,
                      int relu, float *dx, float *dresidual, float *dgamma, float *dbeta, float *workspace, int B, int C, int HW,
                      void *stream);
/* the same with (1) the incoming gradient in two pieces, dy + dy2 (dy2 nullable): a residual block's output feeds the next block's
 * convolution and its identity branch (resnet.py:44-59), and the two gradients are added here, on load, instead of in a launch of
 * their own; (2) y nullable when relu = 1 and the forward had NO residual: the ReLU mask is then recomputed from x, gamma and beta
 * with the forward's expression (the same bits) - one tensor less to read */
int ee_bn_act_bwd2_f32(const float *dy, const float *dy2, const float *y, const float *x, const float *gamma, const float *beta, const float *save_mean,
                       const float *save_invstd, const float *running_mean, const float *running_var, float eps, int training,
                       int relu, float *dx, float *dresidual, float *dgamma, float *dbeta, float *workspace, int B, int C, int HW,
                       void *stream);

/* SyncBatchNorm (ImageNet/experiments_imagenet.py:125, ImageNet/free_imagenet/AT_free_imagenet_ddp.py:149: the reference converts with
 * torch.nn.SyncBatchNorm.convert_sync_batchnorm): the LOCAL halves of a BatchNorm whose batch statistics are those of all ranks' batches.
 * The caller exchanges one small tensor per layer and direction between them (eeadv/syncbn.py over torch.distributed / RCCL):
 *   forward:  ee_syncbn_stats_f32 -> moments [C][3] = this rank's (mean, M2 about it, element count)   | all_gather -> all_moments [W][C][3]
 *             ee_syncbn_apply_f32: merges the W ranks' moments in rank order (Chan's update: the same bits on every rank), writes
 *             save_mean / save_invstd of the global batch, updates running_* (nullable) with the global count (unbiased variance) and
 *             y = [relu]((x - mean) * invstd * gamma + beta [+ residual])
 *   backward: ee_syncbn_bwd_sums_f32 -> sums [C][2] = this rank's (sum dz, sum dz * xhat), which are also its dbeta / dgamma | all_reduce
 *             ee_syncbn_bwd_apply_f32: dx = gamma * invstd * (dz - SUM dz / N - xhat * SUM dz xhat / N) with the global sums, N = n_global;
 *             dresidual = dz.  dy2 / y nullable as in ee_bn_act_bwd2_f32.
 * H*W % 4 == 0, 16-byte aligned tensors; workspace: ee_syncbn_workspace_floats(B, C, HW) floats (0 = unsupported shape). */
int ee_syncbn_workspace_floats(int B, int C, int HW);
int ee_syncbn_stats_f32(const float *x, float *workspace, float *moments, int B, int C, int HW, void *stream);
int ee_syncbn_apply_f32(const float *x, const float *residual, const float *gamma, const float *beta, const float *all_moments, int W,
                        float *running_mean, float *running_var, float momentum, float eps, int relu, float *y, float *save_mean,
                        float *save_invstd, int B, int C, int HW, void *stream);
int ee_syncbn_bwd_sums_f32(const float *dy, const float *dy2, const float *y, const float *x, const float *gamma, const float *beta,
                           const float *save_mean, const float *save_invstd, int relu, float *workspace, float *sums, int B, int C, int HW,
                           void *stream);
int ee_syncbn_bwd_apply_f32(const float *dy, const float *dy2, const float *y, const float *x, const float *gamma, const float *beta,
                            const float *save_mean, const float *save_invstd, const float *global_sums, double n_global, int relu, float *dx,
                            float *dresidual, int B, int C, int HW, void *stream);

/* The end of a residual block with a down-sampling shortcut (resnet.py:54-59 with :137-142): y = relu(bn_a(xa) + bn_b(xb)), bn_a = the
 * block's second BatchNorm on its convolution's output, bn_b = the shortcut's BatchNorm on the 1x1 convolution's output - both per
 * channel, so ONE launch each way instead of two; results bit-identical to ee_bn_act_fwd_f32(xb, relu=0) -> ee_bn_act_fwd_f32(xa,
 * residual, relu=1) and their backwards.  ee_bn_dual_supported: 1 for the shapes it takes (one 256-lane workgroup holds a channel in
 * registers), else use the two calls.  Backward: dy (+ dy2, nullable second piece) -> dxa, dxb (nullable), dgamma / dbeta (nullable). */
int ee_bn_dual_supported(int B, int C, int HW);
int ee_bn_dual_fwd_f32(const float *xa, const float *xb, const float *gamma_a, const float *beta_a, float *running_mean_a, float *running_var_a,
                       float momentum_a, float eps_a, float *save_mean_a, float *save_invstd_a, const float *gamma_b, const float *beta_b,
                       float *running_mean_b, float *running_var_b, float momentum_b, float eps_b, float *save_mean_b, float *save_invstd_b,
                       int training, float *y, int B, int C, int HW, void *stream);
int ee_bn_dual_bwd_f32(const float *dy, const float *dy2, const float *y, const float *xa, const float *xb, const float *gamma_a,
                       const float *save_mean_a, const float *save_invstd_a, const float *running_mean_a, const float *running_var_a,
                       float eps_a, const float *gamma_b, const float *save_mean_b, const float *save_invstd_b,
                       const float *running_mean_b, const float *running_var_b, float eps_b, int training, float *dxa, float *dxb,
                       float *dgamma_a, float *dbeta_a, float *dgamma_b, float *dbeta_b, int B, int C, int HW, void *stream);

/* relu(batch_norm(x)) followed by MaxPool2d(3, stride 2, padding 1) - the ResNet stem (resnet.py:113-117 / :148-150) - without
 * the full-resolution activation: forward x [B,C,H,W] -> y_pool [B,C,OH,OW] + one-byte argmax codes (OH = (H-1)/2+1); backward
 * dy_pool (+ dy_pool2, nullable: the second piece of the gradient, see ee_bn_act_bwd2_f32) + codes + x -> dx [B,C,H,W] (nullable),
 * dgamma, dbeta [C] (nullable).  Values, codes and the pooled-gradient gather equal
 * ee_bn_act_fwd_f32(relu=1) -> ee_maxpool3s2_fwd_f32 and their backwards bit for bit; the gradient sums run over a different
 * partition (dgamma / dbeta / dx agree to rounding).  W % 4 == 0, H*W <= 16000 (else EE_ERR_UNSUPPORTED: use the two calls);
 * workspace: ee_bn_relu_pool_workspace_floats(B, C, H, W) floats (0 = unsupported shape).  conv_stats (nullable): the producing
 * convolution's per-workgroup moments [C][conv_stats_slices][3] (ee_stem7x7s2_fwd_stats_f32); training mode then merges those
 * (Chan's update, fixed order) instead of reading x twice. */
int ee_bn_relu_pool_workspace_floats(int B, int C, int H, int W);
int ee_bn_relu_pool_fwd_f32(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                            float momentum, float eps, int training, float *y_pool, uint8_t *code, float *save_mean,
                            float *save_invstd, float *workspace, const float *conv_stats, int conv_stats_slices, int B, int C, int H,
                            int W, void *stream);
int ee_bn_relu_pool_bwd_f32(const float *dy_pool, const float *dy_pool2, const uint8_t *code, const float *x, const float *gamma, const float *beta,
                            const float *save_mean, const float *save_invstd, const float *running_mean, const float *running_var,
                            float eps, int training, float *dx, float *dgamma, float *dbeta, float *workspace, int B, int C, int H,
                            int W, void *stream);
/* The same pair with one more tensor between them (round 4): the forward also writes x_argmax [B,C,OH,OW] = x at every window's argmax, and
 * the training-mode backward takes its two batch sums from dy_pool and x_argmax alone (a position's masked gradient is the sum of the pooled
 * gradients of the windows whose argmax it is) - one pass over the full-resolution map less.  x_argmax NULL in the backward = the form above.
 * Sums in another order: dgamma / dbeta / dx agree with the form above to rounding. */
int ee_bn_relu_pool_fwd_xa_f32(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                               float momentum, float eps, int training, float *y_pool, uint8_t *code, float *x_argmax, float *save_mean,
                               float *save_invstd, float *workspace, const float *conv_stats, int conv_stats_slices, int B, int C, int H,
                               int W, void *stream);
int ee_bn_relu_pool_bwd_xa_f32(const float *dy_pool, const float *dy_pool2, const uint8_t *code, const float *x, const float *x_argmax, const float *gamma,
                               const float *beta, const float *save_mean, const float *save_invstd, const float *running_mean,
                               const float *running_var, float eps, int training, float *dx, float *dgamma, float *dbeta, float *workspace, int B,
                               int C, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The stem's MaxPool2d(3, stride 2, padding 1) (Tiny_ImageNet/models_tinyimagenet/resnet.py:117), bit-identical to ATen's
 * (first maximum wins, NaN always wins).  x [planes,H,W] -> y, code [planes,OH,OW], OH = (H-1)/2+1; code = 3*kh+kw of the
 * argmax inside its window (one byte).  Backward gathers: dx [planes,H,W] from dy, code.
 * ------------------------------------------------------------------------------------------- */
int ee_maxpool3s2_fwd_f32(const float *x, float *y, uint8_t *code, int planes, int H, int W, void *stream);
int ee_maxpool3s2_bwd_f32(const float *dy, const uint8_t *code, float *dx, int planes, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The shortcut convolution Conv2d(Cin, Cout, kernel_size=1, stride=2, bias=False) of the ResNet blocks
 * (Tiny_ImageNet/models_tinyimagenet/resnet.py:137-142) on the exact-f32 matrix cores.
 *   forward : x [B,Cin,H,W], weight [Cout,Cin] -> y [B,Cout,H/2,W/2]
 *   backward: dy [B,Cout,H/2,W/2] -> dx [B,Cin,H,W] (all of it is written: three of every four pixels are zero)
 *   Cin, Cout, H, W even (EE_ERR_UNSUPPORTED otherwise).  The weight gradient is not provided.
 * ------------------------------------------------------------------------------------------- */
int ee_conv1x1s2_fwd_f32(const float *x, const float *weight, float *y, int B, int Cin, int Cout, int H, int W, void *stream);
int ee_conv1x1s2_bwd_f32(const float *dy, const float *weight, float *dx, int B, int Cin, int Cout, int H, int W, void *stream);

/* Conv2d(3x3, stride 1, padding 1, bias=False) of the residual blocks (resnet.py:26-31) on H x H maps, H = 4, 8 or 16 (ResNet-18 layers 3, 2, 1
 * at 64x64 inputs), forward and backward-data, as Winograd F(2x2, 3x3) around the f32 matrix cores: x [B,KC,H,H], u [16][KC][RC] = the filters
 * in the transform domain, u[4i+j][k][r] = (G g G^T)[i][j] with g = weight[r][k] (forward: KC = Cin, RC = Cout) or g = weight[k][r] rotated by
 * 180 degrees (backward-data: KC = Cout, RC = Cin; x = dy) -> y [B,RC,H,H].  KC % 16 == 0, RC % 32 == 0 (else EE_ERR_UNSUPPORTED).  The weight
 * gradient is not provided.  (Round 3 removed the direct implicit-GEMM kernels ee_conv3x3s1_* / ee_conv3x3s2_* that these and the
 * ee_conv3x3s2_small_* kernels below superseded on every shape of the BASELINE configs.) */
int ee_wino3x3_f32(const float *x, const float *u, float *y, int B, int KC, int RC, int H, void *stream);

/* ---- eval-mode BatchNorm folded into the convolutions (round 4) ---------------------------------------------------------------------------
 * Every validate() pass (experiments_tinyimagenet.py:337 `model.eval()`, then PGD-10/50/100 at :354-358) and the inner loops of ALP / TRADES
 * (utils/attacks.py:249, :405) differentiate the classifier in eval mode: BatchNorm reads its RUNNING statistics and is the per-channel
 * constant map (x - mean) * gamma / sqrt(var + eps) + beta - nothing crosses workgroups, so it needs no launch of its own.  These entry
 * points replace [convolution, BatchNorm(+ residual)(+ ReLU)] (forward) and [BatchNorm/ReLU backward, backward-data convolution] of
 * resnet.py:44-59 by ONE launch each; the expressions and their order are those of ee_bn_act_fwd_f32 / ee_bn_act_bwd2_f32 with
 * training = 0, so the results equal the unfused sequence bit for bit.  mean / var / gamma / beta are [channels] device arrays (gamma / beta
 * may be NULL: 1 / 0).
 *   forward   y = [relu]( (conv3x3(x) - mean) * gamma / sqrt(var + eps) + beta [+ res] )           x [B,Cin,H,H], u = EE_WPREP_WINO_F, res / y [B,Cout,H,H]
 *   backward  dz = (y > 0) * (dy [+ dy2]);  dres <- dz (optional);  dx = conv3x3^T( gamma / sqrt(var + eps) * dz ) [+ dx_add]
 *             dy, dy2, y, dres [B,Cout,H,H]; u_b = EE_WPREP_WINO_B; var / gamma [Cout] (Cout <= 512); dx_add / dx [B,Cin,H,H].  dy2 = the second
 *             piece of a forked output's gradient; dres = the gradient of the block's residual input; dx_add = what reaches the block's input
 *             through the identity branch (one summed gradient then leaves the block). */
int ee_wino3x3_bn_eval_fwd_f32(const float *x, const float *u, const float *mean, const float *var, const float *gamma, const float *beta, float eps,
                               const float *res, int relu, float *y, int B, int Cin, int Cout, int H, void *stream);
int ee_wino3x3_bn_eval_bwd_f32(const float *dy, const float *dy2, const float *y, const float *u_b, const float *var, const float *gamma, float eps,
                               float *dres, const float *dx_add, float *dx, int B, int Cin, int Cout, int H, void *stream);
/* TRAIN-mode BatchNorm with the batch statistics exchanged across the KERNEL BOUNDARY (round 4; resnet.py:44-49: conv1 -> bn1 -> relu ->
 * conv2 inside the attack loop of the training drivers, which runs in train mode - experiments_tinyimagenet.py:234-282).  The producing
 * convolution's output transform writes, next to its raw output, per (channel, image) the plane's (mean, M2) to stats [Cout][B][2]; the
 * consuming convolution's prologue merges the S equal-count partials of each of its reduction channels in a fixed order (bit-reproducible)
 * and stages relu((x - mean) * gamma / sqrt(var + eps) + beta); its first workgroup writes save_mean / save_invstd [Cin] (what
 * ee_bn_act_bwd2_f32(training = 1) reads) and moves running_mean / running_var (may both be NULL) by `momentum` with the unbiased variance,
 * as ee_bn_act_fwd_f32(training = 1) does.  One BatchNorm launch less per residual block and forward pass (10.3 -> 5.2 us,
 * profiles/round4_a_bn_boundary_probe.txt); the statistics are summed in another order than ee_bn_act_fwd_f32's: rounding-level difference.
 * H = 4, 8 or 16; Cin <= 256 for the consumer.  S partials of cnt values each (ee_wino3x3_stats_f32: S = B, cnt = H * H). */
int ee_wino3x3_stats_f32(const float *x, const float *u, float *y, float *stats, int B, int KC, int RC, int H, void *stream);
int ee_wino3x3_bn_train_pre_f32(const float *x, const float *stats, int S, int cnt, const float *gamma, const float *beta, float eps, float momentum,
                                float *running_mean, float *running_var, float *save_mean, float *save_invstd, const float *u, float *y, int B, int KC,
                                int RC, int H, void *stream);
/* The same exchange in the BACKWARD direction (input gradient only; 16x16 maps, <= 128 channels): ee_wino3x3_bwd_sums_f32 is the backward-data
 * convolution of the layer BEHIND the BatchNorm (dy = conv3x3^T(dc) on the backward filter set u_b [16][Cout][Cin]) that also writes per (channel,
 * image) sums [Cin][B][2] = (sum dz, sum dz * xhat), dz = (bn(x) > 0) * dy, from x = the BatchNorm's INPUT and its saved statistics;
 * ee_wino3x3_bn_train_bwd_pre_f32 is the backward-data convolution of the layer in FRONT with the BatchNorm's own backward
 * (gamma * invstd * ((dz - mean dz) - xhat * mean(dz xhat)): ee_bn_act_bwd2_f32 with training = 1, relu = 1 and the mask taken from x)
 * folded into its input staging, the means merged from `sums` (S partials of cnt values) in its prologue. */
int ee_wino3x3_bwd_sums_f32(const float *dc, const float *u_b, const float *x, const float *save_mean, const float *save_invstd, const float *gamma,
                            const float *beta, float *dy, float *sums, int B, int Cin, int Cout, int H, void *stream);
int ee_wino3x3_bn_train_bwd_pre_f32(const float *dy, const float *x, const float *sums, int S, int cnt, const float *save_mean, const float *save_invstd,
                                    const float *gamma, const float *beta, const float *u_b, float *dx, int B, int Cin, int Cout, int H, void *stream);

/* Conv2d(3x3, stride 1, padding 1, bias=False) on a 2x2 map (ResNet-18's layer4 at 64x64 inputs, resnet.py:26-31) as ONE dense product on the
 * f32 matrix cores: every input pixel reaches every output pixel, y[n][(co,p)] = sum x[n][(ci,q)] * w2[(ci,q)][(co,p)], w2 [4 Cin][4 Cout] =
 * EE_WPREP_DENSE_S1 (the backward-data product takes its transpose and Cin / Cout exchanged).  x [B,Cin,2,2] -> y [B,Cout,2,2]; Cin a multiple
 * of 128, Cout of 8.  The *_bn_eval_* forms fold the eval-mode BatchNorm, the residual and the ReLU of the block in exactly as
 * ee_wino3x3_bn_eval_fwd_f32 / _bwd_f32 do (same arguments; w2t [4 Cout][4 Cin] = w2 transposed; Cout <= 512 for the backward). */
int ee_dense2x2_f32(const float *x, const float *w2, float *y, int B, int Cin, int Cout, void *stream);
int ee_dense2x2_bn_eval_fwd_f32(const float *x, const float *w2, const float *mean, const float *var, const float *gamma, const float *beta, float eps,
                                const float *res, int relu, float *y, int B, int Cin, int Cout, void *stream);
int ee_dense2x2_bn_eval_bwd_f32(const float *dy, const float *dy2, const float *y, const float *w2t, const float *var, const float *gamma, float eps,
                                float *dres, const float *dx_add, float *dx, int B, int Cin, int Cout, void *stream);

/* The WEIGHT gradient of the same convolution (`loss.backward()` of the training step, experiments_tinyimagenet.py:304-306; resnet.py:26-31) on
 * H x H maps, H = 2, 4, 8 or 16, as Winograd F(3x3, 2x2) around the f32 matrix cores: x [B,Cin,H,H] (the layer's input), dy [B,Cout,H,H] (the
 * gradient of its output) -> dw [Cout,Cin,3,3] (overwritten).  The sum over images and tiles is split over ~256 workgroups whose partial results
 * meet in `workspace` (ee_wrw3x3_workspace_floats(...) floats, contents undefined on entry and exit; 0 = none needed, NULL allowed) and are added
 * in a fixed order: the result is reproducible bit for bit.  Cin, Cout multiples of 32 (else EE_ERR_UNSUPPORTED; the workspace query returns 0). */
int64_t ee_wrw3x3_workspace_floats(int B, int Cin, int Cout, int H);
int ee_wrw3x3_f32(const float *x, const float *dy, float *dw, float *workspace, int B, int Cin, int Cout, int H, void *stream);

/* ... and of the down-sampling blocks' first convolution (3x3 / stride 2 / padding 1 from an H x H map, H = 16, 8 or 4; resnet.py:26-31, :132-137),
 * optionally together with the block's shortcut Conv2d(1x1, stride 2) of the same input (resnet.py:137-142): plain per-tap products on the f32
 * matrix cores, same split + fixed-order sum.  x [B,Cin,H,H], dy3 (and dy1, or NULL) [B,Cout,H/2,H/2] -> dw = [ dw3 [Cout,Cin,3,3] | dw1 [Cout,Cin]
 * (only with dy1) ], overwritten.  Cin, Cout multiples of 32 (else EE_ERR_UNSUPPORTED). */
int64_t ee_wrw3x3s2_workspace_floats(int B, int Cin, int Cout, int H, int with_shortcut);
int ee_wrw3x3s2_f32(const float *x, const float *dy3, const float *dy1, float *dw, float *workspace, int B, int Cin, int Cout, int H, void *stream);

/* ... and of the stem Conv2d(3, 64, 7, stride 2, padding 3, bias=False) (resnet.py:112): x [B,3,H,W], dy [B,64,H/2,W/2] -> dw [64,3,7,7], overwritten;
 * H even, W a multiple of 32 (else EE_ERR_UNSUPPORTED); same split + fixed-order sum. */
int64_t ee_wrw_stem7x7s2_workspace_floats(int B, int H, int W);
int ee_wrw_stem7x7s2_f32(const float *x, const float *dy, float *dw, float *workspace, int B, int H, int W, void *stream);

/* ... and of the 1x1 / stride 1 convolutions of the bottleneck blocks (resnet.py:75-100; ResNet-50 at ImageNet size, BASELINE config 5): both operands
 * are read in their NCHW layout (rows contiguous along the reduction) - x [B,Cin,HW], dy [B,Cout,HW] with HW = H * W pixels, a multiple of 4 ->
 * dw [Cout,Cin], overwritten; Cin, Cout multiples of 64 (else EE_ERR_UNSUPPORTED); same split + fixed-order sum. */
int64_t ee_wrw1x1_workspace_floats(int B, int Cin, int Cout, int HW);
int ee_wrw1x1_f32(const float *x, const float *dy, float *dw, float *workspace, int B, int Cin, int Cout, int HW, void *stream);

/* Conv2d(3x3, stride 2, padding 1, bias=False) between SMALL maps - the first convolution of ResNet-18's layer2 / layer3 / layer4 at 64x64 inputs
 * (resnet.py:26-31, :132-137): H = 16 (16x16 -> 8x8), 8 (8x8 -> 4x4) or 4 (4x4 -> 2x2) - on the f32 matrix cores with the reduction split over a
 * workgroup's wavefronts.  The filters arrive rearranged, w9 [R/32][K/16][9][4][2][16][4] with R = result and K = reduction channels:
 *   forward        x [B,Cin,H,H] -> y [B,Cout,H/2,H/2];  w9[cb][rd][t][q][h][m][k] = weight[32 cb + 16 h + m][16 rd + 4 q + k][t]
 *   backward-data  dy [B,Cout,H/2,H/2] -> dx [B,Cin,H,H] (all of it written, no multiplication by inserted zeros: four parity classes);
 *                  w9[cb][rd][t][q][h][m][k] = weight[16 rd + 4 q + k][32 cb + 16 h + m][t]
 * Cin, Cout multiples of 32 (else EE_ERR_UNSUPPORTED); the weight gradient is not provided. */
int ee_conv3x3s2_small_fwd_f32(const float *x, const float *w9, float *y, int B, int Cin, int Cout, int H, void *stream);
int ee_conv3x3s2_small_bwd_data_f32(const float *dy, const float *w9, float *dx, int B, int Cin, int Cout, int H, void *stream);
/* The same convolution TOGETHER with the shortcut Conv2d(1x1, stride 2, bias=False) of its BasicBlock (resnet.py:50-59, :137-142: both
 * read the block's input, and the 1x1 filter sees exactly the 3x3's centre tap): w10 [R/32][K/16][10][4][2][16][4] = w9 with the 1x1
 * filters as a tenth tap (same index order).  Forward: y3 = conv3x3s2(x), y1 = conv1x1s2(x); backward-data: dx = conv3x3s2^T(dy3) +
 * conv1x1s2^T(dy1) in one pass (the block's input gradient arrives summed). */
int ee_conv3x3s2_pair_fwd_f32(const float *x, const float *w10, float *y3, float *y1, int B, int Cin, int Cout, int H, void *stream);
int ee_conv3x3s2_pair_bwd_data_f32(const float *dy3, const float *dy1, const float *w10, float *dx, int B, int Cin, int Cout, int H, void *stream);
/* ... with the eval-mode BatchNorms behind the two convolutions folded in (see ee_wino3x3_bn_eval_*; resnet.py:50-59, :137-142 under model.eval()):
 *   forward   y3 = relu( bn1(conv3x3s2(x)) ),  y1 = bn_ds(conv1x1s2(x))       (mean3 .. eps3: the block's bn1; mean1 .. eps1: downsample[1])
 *   backward  dx = conv3x3s2^T( gamma3 / sqrt(var3 + eps3) * (y3 > 0) * dy3 ) + conv1x1s2^T( gamma1 / sqrt(var1 + eps1) * dy1 )     (Cout <= 512) */
int ee_conv3x3s2_pair_bn_eval_fwd_f32(const float *x, const float *w10, const float *mean3, const float *var3, const float *gamma3, const float *beta3,
                                      float eps3, const float *mean1, const float *var1, const float *gamma1, const float *beta1, float eps1, float *y3,
                                      float *y1, int B, int Cin, int Cout, int H, void *stream);
int ee_conv3x3s2_pair_bn_eval_bwd_f32(const float *dy3, const float *y3, const float *dy1, const float *w10, const float *var3, const float *gamma3,
                                      float eps3, const float *var1, const float *gamma1, float eps1, float *dx, int B, int Cin, int Cout, int H,
                                      void *stream);
/* ... and, for TRAIN mode, with the statistics of y3 for the BatchNorm behind it (see ee_wino3x3_bn_train_pre_f32): stats [Cout][S][2] = (mean, M2)
 * per result channel and partial - H = 16: S = 2 B half images of 32 values; H = 8: S = B images of 16 values (H = 4: EE_ERR_UNSUPPORTED). */
int ee_conv3x3s2_pair_stats_fwd_f32(const float *x, const float *w10, float *y3, float *y1, float *stats, int B, int Cin, int Cout, int H, void *stream);

/* The filters of the convolution kernels above in the order those kernels read them, from the Conv2d weight [Cout,Cin,3,3] (one launch;
 * the host rebuilds them once per optimiser step, inside the captured update graph):
 *   EE_WPREP_WINO_F / _B   u [16][K][R] for ee_wino3x3_f32, forward (K = Cin, R = Cout) / backward-data (K = Cout, R = Cin, rotated filters)
 *   EE_WPREP_S2M_F / _B    w9 [R/32][K/16][9][4][2][16][4] for ee_conv3x3s2_small_*; EE_WPREP_S2P_F / _B: w10 (10 taps) for ee_conv3x3s2_pair_*,
 *                          w1 [Cout,Cin] = the shortcut's 1x1 filters
 *   EE_WPREP_DENSE_MAP2    [4 Cin][4 Cout]: a 3x3 / stride 1 / padding 1 convolution on a 2x2 map as one dense product (layer 4)
 *   EE_WPREP_WINO_FB       both Winograd sets in one pass over the weight: out = [ u forward [16][Cin][Cout] | u backward-data [16][Cout][Cin] ]
 *                          (2 * 16 * Cin * Cout floats; Cin, Cout multiples of 32), the same values as EE_WPREP_WINO_F / _B */
#define EE_WPREP_WINO_F 0
#define EE_WPREP_WINO_B 1
#define EE_WPREP_S2M_F 2
#define EE_WPREP_S2M_B 3
#define EE_WPREP_S2P_F 4
#define EE_WPREP_S2P_B 5
#define EE_WPREP_DENSE_MAP2 6
#define EE_WPREP_WINO_FB 7
int ee_conv_weight_prep_f32(int kind, const float *w, const float *w1, float *out, int Cout, int Cin, void *stream);
/* ... n of them in ONE launch per 64 items (the captured update of a training step ends with every rearranged copy of the model: 19 launches of
 * ~5 us for ResNet-18 before round 4).  kinds / w / w1 / out / cout / cin: HOST arrays of length n, each item exactly as ee_conv_weight_prep_f32
 * takes it; the same arithmetic, the same bits. */
int ee_conv_weight_prep_batch_f32(int n, const int *kinds, const float *const *w, const float *const *w1, float *const *out, const int *cout, const int *cin,
                                  void *stream);

/* Backward-data of the stem Conv2d(3, K, kernel_size=7, stride=2, padding=3, bias=False) (resnet.py:112-113): the gradient
 * with respect to the image, i.e. the last step of every PGD iteration's backward pass.
 *   dy [B,K,H/2,W/2], weight [K,3,7,7] -> dx [B,3,H,W];  H, W even. */
int ee_stem7x7s2_bwd_data_f32(const float *dy, const float *weight, float *dx, int B, int K, int H, int W, void *stream);
/* The same convolution forward (resnet.py:112-113 / :147): x [B,3,H,W], weight [K,3,7,7] -> y [B,K,H/2,W/2].
 * H, W even, (W/2) % 32 == 0, K % 64 == 0 (else EE_ERR_UNSUPPORTED); weight 16-byte aligned. */
int ee_stem7x7s2_fwd_f32(const float *x, const float *weight, float *y, int B, int K, int H, int W, void *stream);
/* ... and, on the way, the statistics of y for the BatchNorm that follows it (resnet.py:113): stats [K][S][3] = per channel and
 * workgroup (sum, M2 about the tile's own mean, count); ee_stem7x7s2_fwd_stats_floats = K*S*3 (0: unsupported shape).  Hand them to
 * ee_bn_relu_pool_fwd_f32 (conv_stats, conv_stats_slices = S): it then needs no pass over y for its statistics. */
int ee_stem7x7s2_fwd_stats_floats(int B, int K, int H, int W);
int ee_stem7x7s2_fwd_stats_f32(const float *x, const float *weight, float *y, float *stats, int B, int K, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The convolutional half of the MNIST classifier Net_2 (MNIST/models_mnist/Net2.py:13-16), one launch per half each way:
 *   a1 = relu(max_pool2d(conv1(x), 2)),  a2 = relu(max_pool2d(dropmask * conv2(a1), 2))
 *   x [B,1,28,28], w1 [32,1,5,5], b1 [32] (nullable), w2 [64,32,5,5], b2 [64] (nullable), drop [B,64] (nullable: Dropout2d's per-(image,
 *   channel) Bernoulli(keep) draw, 0 or 1, drawn by the caller; the kernels scale by drop / keep, keep = 1 - p).  drop == NULL with
 *   draw_state != NULL (the device-resident {seed, offset, ticket, -} state of ee_square_draw_f32 / ee_chain_fwd_f32): the second kernel makes
 *   the draw itself (Philox4x32-10, stream id 7), writes the mask to drop_out [B,64] for the backward, and its last workgroup advances
 *   the offset - no host-side random launch inside a captured attack iteration
 *   -> a1 [B,32,12,12], a2 [B,64,4,4] and the one-byte argmax codes of the two pools.
 * Backward (input gradient only): da2 [B,64,4,4] -> dx [B,1,28,28] (NULL: skipped); da1 [B,32,12,12] = the gradient of a1, scratch the caller provides.  ReLU backward
 * follows ATen's threshold rule (the gradient passes unless the output is <= 0); pool ties and NaNs follow ATen's max_pool2d. */
int ee_net2_conv_fwd_f32(const float *x, const float *w1, const float *b1, const float *w2, const float *b2, const float *drop, float keep,
                         uint64_t *draw_state, float *drop_out, float *a1, uint8_t *code1, float *a2, uint8_t *code2, int B, void *stream);
int ee_net2_conv_bwd_f32(const float *da2, const float *a2, const uint8_t *code2, const float *drop, float keep, const float *w2,
                         const float *a1, const uint8_t *code1, const float *w1, float *da1, float *dx, int B, void *stream);
/* Parameter gradients of the two halves (a training step's backward): da2 as above, da1 = what ee_net2_conv_bwd_f32 left in its scratch argument
 * -> out = [ dw2 [64,32,5,5] | dw1 [32,1,5,5] | db2 [64] | db1 [32] ] (52096 floats), overwritten; workspace: ee_net2_conv_wrw_workspace_floats(B)
 * floats (0: none needed, NULL allowed).  Groups of ten images are added in order (bit-reproducible). */
int64_t ee_net2_conv_wrw_workspace_floats(int B);
int ee_net2_conv_wrw_f32(const float *x, const float *a1, const uint8_t *code1, const float *da1, const float *a2, const uint8_t *code2, const float *da2,
                         const float *drop, float keep, float *out, float *workspace, int B, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Classifier head: logits = fc(avgpool(feat).view(B,-1)) for a global average pool
 * (Tiny_ImageNet/models_tinyimagenet/resnet.py:157-160).  feat [B,C,HW], weight [K,C], bias [K] (nullable);
 * pooled [B,C] (kept for the weight gradient), logits [B,K].  C <= 4096, K <= 8192 (EE_ERR_UNSUPPORTED beyond).
 * Backward w.r.t. feat: dfeat[b,c,:] = (dlogits[b,:] . weight[:,c]) / HW.
 * ------------------------------------------------------------------------------------------- */
int ee_pool_linear_fwd_f32(const float *feat, const float *weight, const float *bias, float *pooled, float *logits, int B,
                           int C, int HW, int K, void *stream);
int ee_pool_linear_bwd_f32(const float *dlogits, const float *weight, float *dfeat, int B, int C, int HW, int K, void *stream);
/* ... with the cross-entropy gradient formed inside the same launch (the attack loop: attacks.py:23 sum, :255 mean): logits [B,K] as
 * ee_pool_linear_fwd_f32 wrote them, labels [B], gscale = 1 or 1 / B -> dfeat [B,C,HW]; the bits of ee_ce_f32 + ee_pool_linear_bwd_f32. */
int ee_ce_pool_linear_bwd_f32(const float *logits, const int64_t *labels, float gscale, const float *weight, float *dfeat, int B, int C, int HW,
                              int K, void *stream);

/* HighFreqSuppress for planes up to 256 x 256 (ImageNet 224 x 224, r = 16: utils/core.py:15-55 with cize 224) on the f32 matrix
 * cores, one workgroup per plane, one wavefront per 16-row band (csrc/ee_hfs_mfma.hip).  Same operator, same sq_mode fusions and
 * draw arrays as ee_hfs_f32.  `tables`: ee_hfs_mfma_table_floats(H, W, nu_pad) floats in MFMA fragment order (eeadv/hfs.py:
 * band_tables; scripts/chain_emulate.py big_tables is the layout's specification); nu_pad = 16 or 32 = the kept row frequencies
 * rounded up.  8 B of HBM traffic per element (12 with sq_mode 2). */
int ee_hfs_mfma_table_floats(int H, int W, int nu_pad);
int ee_hfs_mfma_f32(const float *in, float *out, int B, int C, int H, int W, const float *tables, int nu_pad, int sq_mode,
                    const float *sq_x, float eps, const float *stripe, const float *sq_sign, const int64_t *sq_pos, const int32_t *sq_size,
                    int nq, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The whole EE front end of one PGD iteration in two launches (csrc/ee_chain.hip): one workgroup per image.
 *   (Tiny_ImageNet/models_tinyimagenet/resnet_EE_square.py:187-206, MNIST/models_mnist/Net2_EE_square.py:48-63,
 *    utils/core.py:15-55, 509-585, 636-655, utils/attacks.py:25-27)
 * Shapes: ee_chain_supported(C, H, W) != 0 - (3,64,64), (1,28,28), (3,32,32), (1,64,64); others EE_ERR_UNSUPPORTED
 * (callers then use ee_hfs_f32 + ee_frontend_* + ee_pgd_step_bcast_f32, same results).
 * `tables`: ee_chain_table_floats(H, W) floats, the constant operands of the low-pass operator in MFMA fragment order
 * (eeadv/hfs.py: chain_tables; scripts/chain_emulate.py is the layout's specification).
 * ------------------------------------------------------------------------------------------- */
int ee_chain_supported(int C, int H, int W);
int ee_chain_table_floats(int H, int W);

/* forward: x[B,C,H,W] -> x_in = clamp(hfs(add_square(x)) + w * edge125(x), 0, 1) [B,C,H,W], gate[B,C,H,W] (uint8: bit 0 = clamp
 * passes the gradient, bits 1-2 = d add_square/dx in {0,1/2,3/4,1} coded 0..3), gx / gy [B,1,H,W] (the Sobel responses the backward
 * starts from), edge (nullable) [B,1,H,W].  square = 0: no Add_Square (resnet_EE.py:176-191).  square = 1: n_queries = 1 with
 * side `sq_size` (core.py:644); the draws (core.py:637, :645, :648) are either injected (stripe_in [B,C,1,W], sq_pos_in [1],
 * sq_sign_in [1,C]: all three) or made in the kernel from Philox4x32-10 with draw_state = uint64[4] {seed, offset, ticket, -}: the
 * element <-> counter mapping is ee_square_draw_f32's, and the last workgroup to finish advances `offset`, so a replayed HIP graph
 * draws fresh numbers.  HBM traffic: read 4C, write 5C + 8 bytes per pixel. */
int ee_chain_fwd_f32(const float *x, int B, int C, int H, int W, const float *tables, const float *weights27, float alpha, float high,
                     float w, int square, float eps, int sq_size, uint64_t *draw_state, const float *stripe_in, const int64_t *sq_pos_in,
                     const float *sq_sign_in, float *x_in, uint8_t *gate, float *gx, float *gy, float *edge, void *stream);

/* backward + update, in place on x: g_in = dL/dx_in [B,C,H,W];
 *     g = dsquare * hfs(gate * g_in) + edge125_adjoint(w * sum_c gate_c * g_in_c)   (NaN where the edge magnitude is 0)
 *     x = clamp(min(max(x + dir*step*sign(g), x0 - eps), x0 + eps), lo, hi)           (attacks.py:25-27)
 * HBM traffic: read 13C + 8, write 4C bytes per pixel. */
int ee_chain_bwd_f32(const float *g_in, const uint8_t *gate, const float *gx, const float *gy, float *x, const float *x0, int B, int C,
                     int H, int W, const float *tables, const float *weights27, float alpha, float high, float w, float step, float eps,
                     float lo, float hi, int dir, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Optional built-in timing of the last launch of each kernel family (HIP events on `stream`).
 * Off by default; bench.py switches it on outside graph capture to measure kernel durations live.
 * ------------------------------------------------------------------------------------------- */
#define EE_K_PGD_STEP 0
#define EE_K_FRONTEND_FWD 1
#define EE_K_FRONTEND_BWD 2
#define EE_K_EDGE_FWD 3
#define EE_K_EDGE_BWD 4
#define EE_K_CE 5
#define EE_K_PGD_STEP_BCAST 6
#define EE_K_EMPTY 7 /* an event pair with nothing in between: the bracket's own cost, see ee_prof_mark_empty */
#define EE_K_HFS 8          /* ee_hfs_f32, sq_mode 0 */
#define EE_K_CHAIN_FWD 9    /* ee_chain_fwd_f32 */
#define EE_K_CHAIN_BWD 10   /* ee_chain_bwd_f32 */
#define EE_K_HFS_SQ_FWD 11  /* ee_hfs_f32, sq_mode 1 (Add_Square on load) */
#define EE_K_HFS_SQ_BWD 12  /* ee_hfs_f32, sq_mode 2 (times d Add_Square / dx on store) */
#define EE_K_SQUARE_DRAW 13 /* ee_square_draw_f32 */
/* 14 - 17: the direct 3x3 kernels removed in round 3 */
#define EE_K_WINO 18        /* ee_wino3x3_f32 (forward and backward-data are the same kernel); work = the convolution's algorithmic flops */
#define EE_K_CONV3S2_FWD 19 /* ee_conv3x3s2_small_fwd_f32; work = flops */
#define EE_K_CONV3S2_BWD 20 /* ee_conv3x3s2_small_bwd_data_f32 */
#define EE_K_WINO_FUSED 21  /* ee_wino3x3_bn_eval_*, ee_wino3x3_stats_f32, ee_wino3x3_bn_train_*, ee_wino3x3_bwd_sums_f32: the same products with the
                             * BatchNorm work of the layer folded into the staging / output stage; work = the convolution's flops only */
#define EE_K_COUNT 22
int ee_prof_enable(int on);
/* records one empty start/stop bracket on `stream` (family EE_K_EMPTY): callers subtract its mean from the other
 * families' means, because a HIP event pair costs ~4-5 us on gfx950 - comparable to the kernels being timed */
int ee_prof_mark_empty(void *stream);
/* synchronises the recorded events and returns accumulated milliseconds / launch count since the last reset */
int ee_prof_read(int kernel_id, double *total_ms, int64_t *launches);
/* sum of the `work` the timed launches of a family declared (floating-point operations for the matrix-core families; 0 for
 * the HBM-bound ones, whose bytes follow from the shapes the caller knows) */
int ee_prof_read_work(int kernel_id, double *work);
int ee_prof_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* EEADV_H */
