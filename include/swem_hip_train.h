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
 *   k = int(HW * p) of bce_losses.py:49 (0 below start_warm) */
size_t swem_vos_loss_workspace(int B, int N1, long long HW);
int swem_vos_loss_frame_fwd_f32(void *stream, const float *logits, const long long *label, long long label_bs,
                                const float *valid,
                                float *prob, float *raw, float *rowstat, float *iou, int B, int N1, long long HW,
                                long long k, void *ws, size_t ws_bytes);
/* losses = {total, main, aux} from the per-frame statistics of T frames (rowstat [T][B][4], iou [T][B][N1][2]) */
int swem_vos_loss_reduce_f32(void *stream, const float *rowstat, const float *iou, const float *valid, float *losses,
                             int B, int N1, int T, long long HW, long long k, float aux_ratio);
/* d total_loss / d logits of one frame, times the device scalar gout[0] (NULL = 1); T = frames in the clip loss */
int swem_vos_loss_frame_bwd_f32(void *stream, const float *prob, const float *raw, const long long *label,
                                long long label_bs, const float *valid, const float *rowstat, const float *iou, float *dlogits, int B,
                                int N1, int T, long long HW, long long k, float aux_ratio, const float *gout);

/* torch.optim.AdamW, one step over a flat parameter buffer (solver/solver.py:38-41):
 *   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps) */
int swem_adamw_f32(void *stream, float *p, const float *g, float *m, float *v, long long n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step);

#ifdef __cplusplus
}
#endif
#endif
