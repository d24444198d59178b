"""Training loss on HIP kernels: BootstrappedCE + mask-IoU auxiliary loss.

Mirrors the reference's ``losses.VOSLoss`` (losses/__init__.py:15-63 with ``NAME='boots_ce'``, ``AUX='iou'``, the
defaults of configs/config.py:83-89) and ``BootstrappedCE`` (losses/bce_losses.py:7-51): same constructor arguments,
``forward(scores, target, it, valid_obj)`` and the same ``losses`` dict.  The per-pixel softmax / cross entropy, the
top-k selection (a radix select instead of ``torch.topk``), the IoU sums and the whole backward run in
``csrc/train.hip``; nothing here computes with torch.
"""
import torch

from . import _lib, ops


def this_p(it, start_warm, end_warm, top_p):
    """bce_losses.py:44-48 (None below start_warm: plain cross entropy)."""
    if it < start_warm:
        return None
    if it > end_warm:
        return top_p
    return top_p + (1 - top_p) * ((end_warm - it) / (end_warm - start_warm))


class _ClipLoss(torch.autograd.Function):
    """total_loss of a clip from its per-frame logits (B,N+1,H,W); label (B,T,H,W) int64; valid (B,N+1) or None."""

    @staticmethod
    def forward(ctx, label, valid, k, k_dev, aux_ratio, *logits):
        T = len(logits)
        B, N1, H, W = logits[0].shape
        HW = H * W
        dev = logits[0].device
        prob = torch.empty((T, B, N1, HW), dtype=torch.float32, device=dev)
        raw = torch.empty((T, B, HW), dtype=torch.float32, device=dev)
        rowstat = torch.empty((T, B, 4), dtype=torch.float32, device=dev)
        iou = torch.empty((T, B, N1, 2), dtype=torch.float32, device=dev)
        out = torch.empty(3, dtype=torch.float32, device=dev)     # total, main, aux
        wsb = _lib.query('swem_vos_loss_workspace', B, N1, HW)
        ws = ops.workspace(wsb, dev)
        vp = ops._ptr(valid)
        for t, lg in enumerate(logits):
            ops._chk(lg, 'logits')
            _lib.call('swem_vos_loss_frame_fwd_f32', ops._stream(), lg.data_ptr(), label[:, t].data_ptr(),
                      label.stride(0), vp, prob[t].data_ptr(), raw[t].data_ptr(), rowstat[t].data_ptr(), iou[t].data_ptr(),
                      B, N1, HW, k, ops._ptr(k_dev), ws.data_ptr(), wsb)
        _lib.call('swem_vos_loss_reduce_f32', ops._stream(), rowstat.data_ptr(), iou.data_ptr(), vp, out.data_ptr(), B, N1,
                  T, HW, k, ops._ptr(k_dev), aux_ratio)
        ctx.saved = (label, valid, prob, raw, rowstat, iou, k, k_dev, aux_ratio, (B, N1, T, H, W))
        return out

    @staticmethod
    def backward(ctx, gout):
        label, valid, prob, raw, rowstat, iou, k, k_dev, aux_ratio, (B, N1, T, H, W) = ctx.saved
        grads = []
        gout = gout.contiguous()               # (3,): d/d total first (1 for total.backward(); a loss scale otherwise)
        for t in range(T):
            d = torch.empty((B, N1, H, W), dtype=torch.float32, device=prob.device)
            _lib.call('swem_vos_loss_frame_bwd_f32', ops._stream(), prob[t].data_ptr(), raw[t].data_ptr(),
                      label[:, t].data_ptr(), label.stride(0), ops._ptr(valid), rowstat[t].data_ptr(),
                      iou[t].data_ptr(),
                      d.data_ptr(), B, N1, T, H * W, k, ops._ptr(k_dev), aux_ratio, gout.data_ptr())
            grads.append(d)
        return (None, None, None, None, None, *grads)


class VOSLoss:
    """losses/__init__.py:15-63.  ``config_loss`` needs NAME ('boots_ce' or 'ce'), BS_PERIOD, BS_RATIO, AUX ('iou' or
    None), AUX_RATIO; ``max_iter`` and ``device`` are accepted for signature parity."""

    def __init__(self, config_loss, max_iter=1, device=None):
        get = (lambda k: config_loss[k]) if isinstance(config_loss, dict) else (lambda k: getattr(config_loss, k))
        name, aux = get('NAME'), get('AUX')
        if name not in ('boots_ce', 'ce') or aux not in (None, 'iou'):
            raise NotImplementedError('VOSLoss: main loss %r / aux loss %r (the HIP path builds boots_ce|ce + iou)'
                                      % (name, aux))
        assert max_iter > 0
        self.bootstrap = name == 'boots_ce'
        self.start_warm, self.end_warm = get('BS_PERIOD')
        self.top_p = get('BS_RATIO')
        self.aux_alpha = float(get('AUX_RATIO')) if aux is not None else 0.0

    def top_k(self, it, hw):
        """(p, k) of bce_losses.py:44-49 for iteration `it` and hw pixels (k = 0: plain cross entropy)."""
        p = this_p(it, self.start_warm, self.end_warm, self.top_p) if self.bootstrap else None
        return p, (0 if p is None else int(hw * p))

    def clip_loss(self, logits_list, target, it, valid_obj=None, k_dev=None):
        """logits_list: T tensors (B,N+1,H,W); target (B,T,H,W) int64.  Returns the reference's losses dict; total_loss
        carries the gradient, the others are device scalars.  k_dev: int64 device scalar holding k (graph replay)."""
        H, W = logits_list[0].shape[-2:]
        p, k = self.top_k(it, H * W)
        # (the kernels take the clip's label frames through a batch stride: a (G,T,H,W) slice of a larger label tensor -- the
        # clips of one lane, frames 1.. -- is read in place; only the frames themselves must be dense)
        if target.dtype != torch.int64 or target.stride(-1) != 1 or target.stride(-2) != W:
            target = target.long().contiguous()
        out = _ClipLoss.apply(target, valid_obj, k, k_dev, self.aux_alpha, *logits_list)
        det = out.detach()
        return {'total_loss': out[0], 'main_loss': det[1], 'aux_loss': det[2], 'p': 1.0 if p is None else p, '_vec': out}

    def __call__(self, scores, target, it, valid_obj=None):
        """scores (B,N+1,T,H,W) as in the reference (losses/__init__.py:34-41)."""
        frames = [scores[:, :, t].contiguous() for t in range(scores.shape[2])]
        return self.clip_loss(frames, target, it, valid_obj)

    forward = __call__


def get_criterion(config_loss, logger=None, rank=1, max_iter=1, device=None):
    """losses/__init__.py:66-71."""
    return VOSLoss(config_loss, max_iter, device)
