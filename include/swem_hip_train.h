/* swem_hip_train.h -- C ABI of the TRAINING step of the SWEM path on MI355X (gfx950): loss, optimizer and the
 * backward kernels of every stage of swem_hip.h.  Same conventions as swem_hip.h (plain device pointers + sizes +
 * hipStream_t, caller-provided workspaces, nothing allocated or kept, never synchronises, 0 / negative status).
 * Replaces, for reference/methods/SWEM/swem_trainer.py:59-108 (SWEMTrainer.one_step):
 *   the ATen autograd graph of the clip loop      -> the *_bwd entry points below, driven by swem_amd/autograd.py
 *   losses/__init__.py:34-63, bce_losses.py       -> swem_vos_loss_*
 *   solver/solver.py:38-41 (optim.AdamW)          -> swem_adamw_f32
 */
#ifndef SWEM_HIP_TRAIN_H
#define SWEM_HIP_TRAIN_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------
 * Loss (losses/__init__.py:34-63): BootstrappedCE (bce_losses.py:7-51) + mask-IoU auxiliary loss (:109-141),
 * evaluated frame by frame; frames are independent up to the final means.
 *   logits [B][N1][HW] (N1 = objects + background), label int64, row b at label + b*label_bs (a frame of a
 *          (B,T,H,W) tensor is passed without a copy: label_bs = T*HW), valid [B][N1] float (> 0.5 = object exists)
 *          or NULL (= all valid).  As in the reference (`scores[b][valid_obj[b] > 0.5]`) the softmax runs over the
 *          valid channels only and label values index the valid channels in order.
 *   prob   [B][N1][HW]   softmax over the valid channels (0 on invalid ones)
 *   raw    [B][HW]       per-pixel cross entropy
 *   rowstat[B][4]        k-th largest value of raw (threshold), #values above it, their sum, #values equal to it
 *                        (k = 0: plain mean -- {0, HW, sum, 0})
 *   iou    [B][N1][2]    sum min(prob, onehot), sum max(prob, onehot) + 1e-6 per valid channel
 *   k = int(HW * p) of bce_losses.py:49 (0 below start_warm); k_dev != NULL: read from device memory instead (the value
 *   changes every iteration, a captured HIP graph must not bake it in) */
size_t swem_vos_loss_workspace(int B, int N1, long long HW);
int swem_vos_loss_frame_fwd_f32(void *stream, const float *logits, const long long *label, long long label_bs,
                                const float *valid,
                                float *prob, float *raw, float *rowstat, float *iou, int B, int N1, long long HW,
                                long long k, const long long *k_dev, void *ws, size_t ws_bytes);
/* losses = {total, main, aux} from the per-frame statistics of T frames (rowstat [T][B][4], iou [T][B][N1][2]) */
int swem_vos_loss_reduce_f32(void *stream, const float *rowstat, const float *iou, const float *valid, float *losses,
                             int B, int N1, int T, long long HW, long long k, const long long *k_dev, float aux_ratio);
/* d total_loss / d logits of one frame, times the device scalar gout[0] (NULL = 1); T = frames in the clip loss */
int swem_vos_loss_frame_bwd_f32(void *stream, const float *prob, const float *raw, const long long *label,
                                long long label_bs, const float *valid, const float *rowstat, const float *iou, float *dlogits, int B,
                                int N1, int T, long long HW, long long k, const long long *k_dev, float aux_ratio,
                                const float *gout);

/* torch.optim.AdamW, one step over a flat parameter buffer (solver/solver.py:38-41):
 *   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps) */
int swem_adamw_f32(void *stream, float *p, const float *g, float *m, float *v, long long n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step);
/* The same update behind a device-side GATE -- the counterpart of GradScaler's found_inf (basic_trainer.py:222-223 steps the
 * optimizer through one): `gate` points at `ngate` device floats; if any is non-zero when the launch runs, p, m and v are left
 * untouched.  applied (optional device int): incremented once by every launch that did update.  The trainer feeds the gate from
 * swem_fault_flags_f32 below, all-reduced with the loss scalars, so every rank skips the same steps and no update is ever made
 * from gradients a faulted launch produced -- without a host synchronisation per step. */
int swem_adamw_gated_f32(void *stream, float *p, const float *g, float *m, float *v, long long n, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int step, const float *gate, int ngate, int *applied);
/* The device's sticky fault word (swem_hip.h, "Asynchronous faults") as two floats, in stream order behind the step's launches:
 * flags2[0] = 1 if SWEM_FAULT_RANGE is set, flags2[1] = 1 if any other fault bit is set (else 0).  Floats, so that they travel in
 * the same SUM all-reduce as the loss scalars (swem_trainer.py:41-43, basic_trainer.py:105-110). */
int swem_fault_flags_f32(void *stream, const unsigned *fault, float *flags2);

/* ------------------------------------------------------------------------------------
 * Convolution backward.  The DATA gradient is swem_conv2d_nhwc_f32 / _bf16x3 themselves with SWEM_CONV_DGRAD
 * (swem_hip.h): x = dY, filters transposed to [Cin][KH][KW][Cout]; SWEM_CONV_MASK_POS applies the input-ReLU mask.
 * The WEIGHT gradient:  dw[co][ci][ky][kx] (+)= sum_pixels dY[b][oy][ox][co] * act(x[b][oy*s-p+ky][ox*s-p+kx][ci])
 *   dy [B][Ho][Wo][Cout]; up to three concatenated NHWC sources as in swem_conv2d_nhwc_f32 (bsK = 0: shared map);
 *   act = ReLU when relu_in; dw in the reference's OIHW layout with cin_store input channels (< c0+c1+c2 only for the
 *   zero-padded stems); accumulate != 0 adds to dw (the gradient buffer of the optimizer). */
size_t swem_conv2d_wgrad_workspace(int B, int H, int W, int c0, int c1, int c2, int Cout, int KH, int KW, int stride,
                                   int pad);
int swem_conv2d_wgrad_f32(void *stream, const float *dy, const float *x0, int c0, long long bs0, const float *x1,
                          int c1, long long bs1, const float *x2, int c2, long long bs2, int B, int H, int W, int Cout,
                          int KH, int KW, int stride, int pad, int relu_in, float *dw, int cin_store, int accumulate,
                          void *ws, size_t ws_bytes);
/* The same weight gradient on the bf16 matrix pipe from PRE-SPLIT operands (swem_split_bf16x3_f32, swem_hip.h): dy3 =
 * plane 0 of the split of dY ([3][Cout/8][B*Ho*Wo][8], plane stride dy_ps elements), xK = plane 0 of the split of source
 * K (act already folded into the split; plane stride psK, batch stride bsK in elements, 0 = shared map).  Channel counts
 * are multiples of 8.  math 1 = bf16x6 (six products of the exact three-way split, fp32-level error), 2 = plain bf16
 * (plane 0, one product: config.AMP).  plan 0 = heuristic, else  tile | pixel_slices << 4 | flip << 12  (tile 1 = 64x64, 2 = 128x128; pixel_slices <= 255; flip = 1 takes the
 * other LDS slab depth: the default is 16 pixels per slab for the six-product 128x128 tile, 32 otherwise). */
size_t swem_conv2d_wgrad_bf16x3_workspace(int B, int H, int W, int c0, int c1, int c2, int Cout, int KH, int KW,
                                          int stride, int pad, int plan);
int swem_conv2d_wgrad_bf16x3(void *stream, const unsigned short *dy3, long long dy_ps, const unsigned short *x0, int c0,
                             long long bs0, long long ps0, const unsigned short *x1, int c1, long long bs1,
                             long long ps1, const unsigned short *x2, int c2, long long bs2, long long ps2, int B, int H,
                             int W, int Cout, int KH, int KW, int stride, int pad, int math, float *dw, int cin_store,
                             int accumulate, int plan, void *ws, size_t ws_bytes);
/* ---- fp32-level training on the f16x3 arithmetic (round 5).  The reference back-propagates in fp32
 * (swem_trainer.py:92-105, `loss.backward()` without autocast unless config.AMP); the six-product bf16 split reproduces that at six
 * MFMA products per fp32 product, the fp16 (hi, mid) pair of the inference path at three -- but gradients span too many binades for
 * an unscaled pair.  swem_split_f16x2_scaled_f32 writes the pair of  dY * 2^s  ([2][C/8][npix][8] fp16, as swem_split_f16x2_f32),
 * s = 13 - floor(log2 max|dY|) chosen ON THE DEVICE from the map itself (two launches: SWEM_AMAX_PARTS block maxima, then the
 * split; no host decision, so a captured graph replays it), and stores 2^-s in scratch[0] for the consumers:
 *   - the data-gradient convolution (swem_conv2d_nhwc_bf16x3_planes_ctr on those planes, SWEM_PLAN_F16): its per-column epilogue
 *     `scale` multiplied by scratch[0] -- swem_vec_scale_f32(in, scratch, out, n): out[i] = (in ? in[i] : 1) * scratch[0];
 *   - swem_conv2d_wgrad_f16x3 (dy_inv_scale = scratch): three products hi.mid + mid.hi + hi.hi on the f16 MFMA, the partial sums
 *     multiplied by 2^-s in the reduce.  x planes: swem_split_f16x2_f32 of the forward activations (unscaled: |x| < 65520, faulted).
 * Powers of two: the scaling is exact.  The largest element lands in [2^13, 2^14); elements within 2^-15 of it keep >= 22 bits,
 * smaller ones carry an absolute error <= 2^-39 of the maximum.  scratch: SWEM_AMAX_PARTS + 1 floats of device memory, contents
 * irrelevant on entry (nothing to zero), nparts = 0.  nparts > 0: scratch[1 .. nparts] ALREADY hold block maxima of |x| that the
 * map's producer wrote (swem_bn_act_bwd_amax_f32): the first pass is skipped, scratch is nparts + 1 floats.  A non-finite element
 * sets SWEM_FAULT_RANGE in `fault` (may be NULL) and leaves s = 0.
 * plan / workspace of the weight gradient: as swem_conv2d_wgrad_bf16x3 (same workspace query). */
#define SWEM_AMAX_PARTS 256
int swem_split_f16x2_scaled_f32(void *stream, const float *x, void *out, long long npix, int C, float *scratch, int nparts,
                                void *fault);
int swem_vec_scale_f32(void *stream, const float *in, const float *factor, float *out, int n);
/* filters w [Cout][K] (K % 8 == 0, K in the pre-split kernel's order) -> the fp16 pair planes [2][K/8][Cout][8] of the f16x3
 * arithmetic, each filter scaled by its own power of two (largest weight in [2^13, 2^14): swem_hip.h, "f16x3"), and
 * scale_out[n] = (scale_in ? scale_in[n] : 1) * 2^-e[n], the epilogue scale that undoes it.  One launch: the training step
 * re-packs every filter every step. */
int swem_pack_filters_f16x2_f32(void *stream, const float *w, void *out, int Cout, int K, const float *scale_in,
                                float *scale_out);
int swem_conv2d_wgrad_f16x3(void *stream, const void *dy2, long long dy_ps, const void *x0, int c0, long long bs0,
                            long long ps0, const void *x1, int c1, long long bs1, long long ps1, const void *x2, int c2,
                            long long bs2, long long ps2, int B, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                            const float *dy_inv_scale, float *dw, int cin_store, int accumulate, int plan, void *ws,
                            size_t ws_bytes);
/* column sums of a [M][C] matrix: out1[c] (+)= sum_m a[m][c], out2[c] (+)= sum_m a[m][c]*b[m][c]  (bias and frozen
 * BatchNorm parameter gradients); either output may be NULL */
size_t swem_colsum_workspace(long long M, int C);
int swem_colsum_f32(void *stream, const float *a, const float *b, float *out1, float *out2, long long M, int C,
                    int accumulate, void *ws, size_t ws_bytes);
/* y[i] (+)= sum_b x[b][i], i < n: gradient of a map shared by the B objects of a frame */
int swem_sum_batch_f32(void *stream, const float *x, float *y, int B, long long n, int accumulate);
/* Clip-batched step (swem_trainer.py:60-90 runs the B clips of a GPU as one tensor): a map shared by the N objects of each of
 * G clips, laid out once per object -- y[g N + j][i] = x[g][i], i < n -- and its gradient y[g][i] = sum_j x[g N + j][i] */
int swem_expand_groups_f32(void *stream, const float *x, float *y, int G, int N, long long n);
int swem_sum_groups_f32(void *stream, const float *x, float *y, int G, int N, long long n);

/* ------------------------------------------------------------------------------------
 * Frozen BatchNorm (+ residual, ReLU) as a stage of its own: training keeps the raw convolution output c for the
 * BatchNorm parameter gradients (mod_resnet.py:58-113 with the trainer's set_bn_eval, swem_trainer.py:37-39).
 *   y = act(c * alpha + shift + res),  alpha = gamma / sqrt(var + eps),  shift = beta - mean * alpha
 *   bwd (one pass + a per-channel finish): dz = dy * (y > 0) (also the residual's gradient, optional), dc = dz * alpha,
 *   dgamma += invstd * (sum dz*c - mean * sum dz), dbeta += sum dz  (either may be NULL; the conv bias's gradient is the
 *   column sum of dc, taken by the convolution's own backward)
 *   planes (optional, C % 8 == 0): also write the bf16 planes [3][C/8][M][8] of y (forward) / of dc (backward) exactly as
 *   swem_split_bf16x3_f32 would, for the convolution that consumes them */
int swem_bn_act_f32(void *stream, const float *c, const float *alpha, const float *shift, const float *res, float *y,
                    long long M, int C, int relu, void *planes);
size_t swem_bn_act_bwd_workspace(long long M, int C);
int swem_bn_act_bwd_f32(void *stream, const float *dy, const float *y, const float *c, const float *alpha,
                        const float *mean, const float *invstd, float *dz, float *dc, float *dgamma, float *dbeta,
                        long long M, int C, int relu, void *planes, void *ws, size_t ws_bytes);
/* ... for consumers on the f16x3 arithmetic (round 5).  swem_bn_act_planes_f32: the forward stage with the format of the planes
 * given -- nplanes = 3 (bf16, as above) or SWEM_PLANES_F16 (the fp16 (hi, mid) pair [2][C/8][M][8], exactly swem_split_f16x2_f32's;
 * `fault`: SWEM_FAULT_RANGE when an output leaves the fp16 range, may be NULL).  swem_bn_act_bwd_amax_f32: the backward stage, no
 * planes, but one float per block -- the largest |dc| the block wrote -- into amax_parts[0 .. swem_bn_act_bwd_amax_parts(M, C)):
 * the first pass of swem_split_f16x2_scaled_f32 for the gradient map dc (pass it scratch with scratch + 1 = amax_parts and
 * nparts = that count). */
int swem_bn_act_planes_f32(void *stream, const float *c, const float *alpha, const float *shift, const float *res, float *y,
                           long long M, int C, int relu, void *planes, int nplanes, void *fault);
int swem_bn_act_bwd_amax_parts(long long M, int C);
int swem_bn_act_bwd_amax_f32(void *stream, const float *dy, const float *y, const float *c, const float *alpha,
                             const float *mean, const float *invstd, float *dz, float *dc, float *dgamma, float *dbeta,
                             long long M, int C, int relu, float *amax_parts, void *ws, size_t ws_bytes);
/* backward of swem_cbam_f32 (y = x + CBAM(x), attentions.py:22-84): dx [B][H][W][C]; the gradients of the six
 * parameters (mlp.1 / mlp.3 weight+bias, spatial conv weight [1][2][7][7] + bias) are ACCUMULATED.  Ties of the two
 * max-pools send the gradient to the first maximum. */
size_t swem_cbam_bwd_workspace(int B, int H, int W, int C);
int swem_cbam_bwd_f32(void *stream, const float *x, const float *w1, const float *b1, const float *w2, const float *b2,
                      const float *w7, const float *b7, const float *dy, float *dx, float *dw1, float *db1, float *dw2,
                      float *db2, float *dw7, float *db7, int B, int H, int W, int C, int hid, void *ws,
                      size_t ws_bytes);
/* alpha = gamma / sqrt(var + eps), shift = beta - mean * alpha, invstd = 1 / sqrt(var + eps) (frozen BatchNorm) */
int swem_bn_fold_f32(void *stream, const float *gamma, const float *beta, const float *mean, const float *var, float eps,
                     float *alpha, float *shift, float *invstd, int C);
/* modules.py:25-26 as two convolutions + a gate: y = f * sigmoid(a) and its gradients; n elements */
int swem_glu_f32(void *stream, const float *f, const float *a, float *y, long long n);
int swem_glu_bwd_f32(void *stream, const float *dy, const float *f, const float *a, float *df, float *da, long long n);
int swem_add_f32(void *stream, const float *a, const float *b, float *y, long long n);
/* backward of swem_maxpool3x3s2_nhwc_f32 (first maximum of a window takes the gradient, as ATen) */
int swem_maxpool3x3s2_bwd_f32(void *stream, const float *x, const float *dy, float *dx, int B, int H, int W, int C);
/* the same gradient with the forward output y = maxpool(x) at hand (a pixel wins a window iff it equals y there and no earlier
 * position does): a quarter of the loads; bit-identical for finite inputs */
int swem_maxpool3x3s2_bwd_y_f32(void *stream, const float *x, const float *y, const float *dy, float *dx, int B, int H, int W,
                               int C);
/* adjoint of the bilinear upsampling of swem_upsample_add_nhwc_f32 (the skip branch's gradient is dy itself) and of
 * swem_resize_planes_f32 mode 1 for upsampling (planes) */
int swem_upsample_bwd_nhwc_f32(void *stream, const float *dy, float *dlow, int B, int Hl, int Wl, int Ho, int Wo,
                               int C);
int swem_resize_bilinear_bwd_f32(void *stream, const float *dy, float *dx, int planes, int Hi, int Wi, int Ho, int Wo);
/* backward of swem_decode_head_f32: dlogits / dprob (either may be NULL) [B][N+1][Ho*Wo] -> dlogit4 [B*N][h4][w4];
 * workspace B*N*Ho*Wo floats */
int swem_decode_head_bwd_f32(void *stream, const float *logit4, const float *valid, const float *dlogits,
                             const float *dprob, float *dlogit4, int B, int N, int h4, int w4, int Ho, int Wo, void *ws,
                             size_t ws_bytes);
/* backward of swem_pred_head_f32: dx [B][H][W][C]; dw (OIHW [1][C][3][3]) and db are ACCUMULATED */
size_t swem_pred_head_bwd_workspace(int B, int H, int W, int C);
int swem_pred_head_bwd_f32(void *stream, const float *x, const float *w, const float *dlogit, float *dx, float *dw,
                           float *db, int B, int H, int W, int C, void *ws, size_t ws_bytes);
/* backward of swem_prep_value_input_f32 w.r.t. the masks: dxin [B*N][H][W][8] -> dmasks [B][N+1][H][W] */
int swem_prep_value_input_bwd_f32(void *stream, const float *dxin, float *dmasks, int B, int N, int H, int W,
                                  int single_obj);

/* ------------------------------------------------------------------------------------
 * EM / matching in the training step.  The E, M and W steps run under no_grad in the reference (modules.py:93,112,122):
 * only the value update nu = (zita_prev*nu_prev + v.z)/zita (:164-165) and the matching carry gradient.
 * swem_memorize_train_f32 = swem_memorize_f32 that also returns the last responsibilities z [N][Pz][2L]
 * (Pz = swem_em_pad(P), rows >= P zero) for swem_nu_update_bwd_f32:
 *   dv [N][P][V] (the NHWC value map's gradient), dnu_prev [2N][V][L] (may be NULL) from dnu [2N][V][L] */
int swem_memorize_train_f32(void *stream, const float *x, const float *v, const float *masks, const float *kappa_prev,
                            const float *nu_prev, const float *zita_prev, float *kappa_out, float *nu_out,
                            float *zita_out, float *z_out, int N, int C, int V, int P, int L, int T, float tau,
                            void *ws, size_t ws_bytes);
size_t swem_nu_update_bwd_workspace(int N, int V, int P, int L);
int swem_nu_update_bwd_f32(void *stream, const float *z, const float *zita_prev, const float *zita, const float *dnu,
                           float *dv, float *dnu_prev, int N, int V, int P, int L, void *ws, size_t ws_bytes);
/* backward of swem_match_f32 for the N <= 7 objects of one clip: dmem [N][Pm][V] and dS [N][P][2*topl] (dS may be NULL)
 * -> dqk [P][C] (summed over objects; through the query's l2norm), dnu_first / dnu_update [N][2][V][L].
 * Ties among the top-l values take the gradient jointly (measure zero). */
size_t swem_match_bwd_workspace(int N, int C, int V, int P, int L, int nbanks);
int swem_match_bwd_f32(void *stream, const float *qk, const float *kappa_first, const float *nu_first,
                       const float *kappa_update, const float *nu_update, const float *dmem, const float *dS,
                       float *dqk, float *dnu_first, float *dnu_update, int N, int C, int V, int P, int L, int topl,
                       float tau, void *ws, size_t ws_bytes);

#ifdef __cplusplus
}
#endif
#endif
