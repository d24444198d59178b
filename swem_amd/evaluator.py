"""Per-sequence inference loop (reference methods/SWEM/swem_evaluator.py:59-102) and the FPS meter
(utils/logger.py:87-108, basic_evaluator.py:171-176).

Same call order as the reference: frame 0 encode_key -> encode_value -> init; then for every frame
encode_key -> match -> segment -> argmax/one-hot -> [bilinear resize -> encode_value -> memorize]
(every frame but the last is memorised).  The reference loop is plain Python over ``model(mode, ...)``;
so is this one, with the resize / argmax / one-hot steps on HIP kernels instead of ATen.
"""
import time

import torch

from . import ops


class FrameSecondMeter:
    """utils/logger.py:87-108: fps = frames / synchronised wall time of the per-sequence loops."""

    def __init__(self):
        self.frame_n = 0
        self.total_time = 1e-12
        self.ti = time.time()

    def tic(self):
        self.ti = time.time()

    def toc(self, frame_n):
        self.frame_n += frame_n
        self.total_time += time.time() - self.ti

    @property
    def fps(self):
        return self.frame_n / self.total_time


def evaluate_davis_seq(model, frames, init_masks, out_size, trace=None):
    """frames (1,T,3,H,W) in [0,1] on the device; init_masks list with a (1,N+1,Ho,Wo) float mask first;
    returns (list of (1,Ho,Wo) int64 index maps, list of (1,N+1,Ho,Wo) probability maps)."""
    preds, pred_scores = [], []
    b, t, c, h, w = frames.shape
    out_size = (int(out_size[0]), int(out_size[1]))
    mk16, _, s16, _, _ = model('encode_key', frames[:, 0])
    init_mask = ops.resize_planes(init_masks[0].float().contiguous(), (h, w), 'nearest')
    mv16 = model('encode_value', frames[:, 0], init_mask, s16)
    model('init', mk16, mv16, init_masks[0])
    for i in range(1, t):
        qk16, qv16, s16, s8, s4 = model('encode_key', frames[:, i])
        context, n = model('match', qk16, qv16)
        logits, pred_mask = model('segment', n, context, s8, s4, None, out_size)
        pred_scores.append(pred_mask)
        pred, hard_pred_mask = ops.argmax_onehot(pred_mask, want_onehot=i < t - 1)
        if trace is not None:
            trace.append({'qk16': qk16, 'qv16': qv16, 's16': s16, 's8': s8, 's4': s4, 'context': context,
                          'logits': logits})
        if i < t - 1:
            pm = ops.resize_planes(pred_mask, (h, w), 'bilinear')
            mv16 = model('encode_value', frames[:, i], pm, s16)
            model('memorize', qk16, mv16, hard_pred_mask, pm)
        preds.append(pred)
    return preds, pred_scores


def frame_step(model, frame, out_size, memorize=True):
    """One steady-state frame (swem_evaluator.py:72-97) for a (1,3,H,W) device tensor; returns the index map."""
    h, w = frame.shape[-2:]
    qk16, qv16, s16, s8, s4 = model('encode_key', frame)
    context, n = model('match', qk16, qv16)
    logits, pred_mask = model('segment', n, context, s8, s4, None, out_size)
    pred, hard = ops.argmax_onehot(pred_mask, want_onehot=memorize)
    if memorize:
        pm = ops.resize_planes(pred_mask, (h, w), 'bilinear')
        mv16 = model('encode_value', frame, pm, s16)
        model('memorize', qk16, mv16, hard, pm)
    return pred


class FrameGraph:
    """The steady-state frame captured once into a HIP graph and replayed per frame.

    A frame is ~300 short kernel launches; replaying them from a graph removes the host launch cost and the
    inter-kernel gaps.  Shapes are static within a sequence, the kernels never synchronise and the library
    allocates nothing, so the capture is a plain ``torch.cuda.graph``.  The recurrent state (the 'update' bank)
    lives in static buffers: the captured memorize writes fresh bases, a device copy moves them into the static
    ones at the end of the graph.  Requires both banks to exist (i.e. at least two frames already processed)."""

    def __init__(self, model, frame_shape, out_size):
        self.model, self.out_size = model, (int(out_size[0]), int(out_size[1]))
        core = model.swem_core
        upd = core.memories['update'].bases
        if upd is None or core.memories['first'].bases is None:
            raise RuntimeError('FrameGraph needs an initialised memory with both banks (run two frames eagerly first)')
        dev = upd['kappa'].device
        self.frame = torch.empty(frame_shape, dtype=torch.float32, device=dev)
        self.state = {k: v.clone() for k, v in upd.items()}
        core.memories['update'].bases = self.state
        self.graph = torch.cuda.CUDAGraph()
        self.pred = None

    def capture(self, example_frame):
        core = self.model.swem_core
        self.frame.copy_(example_frame)
        with torch.no_grad():
            # one eager pass on a side stream (also sizes every workspace), then restore the state it consumed
            saved = {k: v.clone() for k, v in self.state.items()}
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                frame_step(self.model, self.frame, self.out_size)
            torch.cuda.current_stream().wait_stream(s)
            for k in self.state:
                self.state[k].copy_(saved[k])
            core.memories['update'].bases = self.state
            with torch.cuda.graph(self.graph):
                self.pred = frame_step(self.model, self.frame, self.out_size)
                new = core.memories['update'].bases
                for k in self.state:
                    self.state[k].copy_(new[k])
            core.memories['update'].bases = self.state
        return self

    def run(self, frame):
        """Stage the frame into the static input buffer and replay; returns the (static) int64 index map."""
        self.frame.copy_(frame)
        self.graph.replay()
        return self.pred


def run_sequences(model, sequences, meter=None):
    """basic_evaluator.py:149-199 without the disk IO: sequences = iterable of (frames, init_mask, out_size)."""
    meter = meter or FrameSecondMeter()
    results = []
    for frames, init_mask, out_size in sequences:
        with torch.no_grad():
            torch.cuda.synchronize()
            meter.tic()
            preds, _ = evaluate_davis_seq(model, frames, [init_mask] + [None] * (frames.shape[1] - 1), out_size)
            torch.cuda.synchronize()
            meter.toc(frames.shape[1])
        results.append(preds)
    return results, meter
