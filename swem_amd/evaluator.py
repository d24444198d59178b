"""Per-sequence inference loop (reference methods/SWEM/swem_evaluator.py:59-102) and the FPS meter
(utils/logger.py:87-108, basic_evaluator.py:171-176).

Same call order as the reference: frame 0 encode_key -> encode_value -> init; then for every frame
encode_key -> match -> segment -> argmax/one-hot -> [bilinear resize -> encode_value -> memorize]
(every frame but the last is memorised).  The reference loop is plain Python over ``model(mode, ...)``;
so is this one, with the resize / argmax / one-hot steps on HIP kernels instead of ATen.
"""
import time

import os
import sys

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


def range_fallback(model, err, what):
    """A sequence raised SwemRangeError: an activation of THIS model on THIS input left the fp16 range of the f16x3 arithmetic
    (the reference's fp32 inference has no such limit, networks.py:22-32).  The model's book leaves f16x3 for the bf16x6 /
    fp32 kernels (ops.PlanBook.to_full_range: all 24 operand bits, fp32 exponent range) and stays there -- a checkpoint that
    overflows once will again -- and the caller re-runs the sequence, fresh work in this process.  Loud: a warning per model."""
    import warnings
    if model.book.full_range:
        raise err            # (cannot happen: a full-range book produces no fp16 pair)
    n = model.book.to_full_range()
    warnings.warn('swem_amd: %s: %s -- this model now runs the full-range arithmetic (bf16x6 / fp32 kernels; %d tuned f16x3 '
                  'plans converted): about 0.6x the f16x3 frame rate, the reference\'s fp32 range' % (what, err, n), RuntimeWarning)


def evaluate_davis_seq(model, frames, init_masks, out_size, trace=None):
    """frames (1,T,3,H,W) in [0,1] on the device; init_masks list with a (1,N+1,Ho,Wo) float mask first;
    returns (list of (1,Ho,Wo) int64 index maps, list of (1,N+1,Ho,Wo) probability maps).
    A sequence whose activations leave the fp16 range of the default f16x3 arithmetic is re-run in the full-range one
    (`range_fallback`): the result is then the fp32-range result, never a silently wrong mask."""
    try:
        return _davis_seq(model, frames, init_masks, out_size, trace)
    except ops.SwemRangeError as err:
        range_fallback(model, err, 'evaluate_davis_seq')
        if trace is not None:
            del trace[:]
        return _davis_seq(model, frames, init_masks, out_size, trace)


def _davis_seq(model, frames, init_masks, out_size, trace=None):
    ops.drain_faults('evaluate_davis_seq')      # (a fault left by earlier work is not this sequence's: ops.FAULT_OWNERS)
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
    # the sequence boundary is where the host waits anyway (basic_evaluator.py:171-176 synchronises around every sequence):
    # asynchronous faults of the sequence's launches surface here, not as a silently wrong mask
    ops.check_faults()
    return preds, pred_scores


def evaluate_ytvos_seq(model, frames, init_masks, out_size):
    """swem_evaluator.py:104-148: like the DAVIS loop, but objects may be annotated from a later frame on:
    ``init_masks[i]`` (1,N'+1,Ho,Wo) then zeroes the predicted scores where a new object sits and appends the new
    masks as extra channels; the memory grows by random-initialised bases for the new ids (modules.py:140-146).
    Range faults of the f16x3 arithmetic: as evaluate_davis_seq."""
    try:
        return _ytvos_seq(model, frames, init_masks, out_size)
    except ops.SwemRangeError as err:
        range_fallback(model, err, 'evaluate_ytvos_seq')
        return _ytvos_seq(model, frames, init_masks, out_size)


def _ytvos_seq(model, frames, init_masks, out_size):
    ops.drain_faults('evaluate_ytvos_seq')
    preds = []
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
        if init_masks[i] is not None:
            pred_mask = ops.inject_objects(pred_mask, init_masks[i])
            n = pred_mask.shape[1] - 1
        pred, hard_pred_mask = ops.argmax_onehot(pred_mask, want_onehot=i < t - 1)
        if i < t - 1:
            pm = ops.resize_planes(pred_mask, (h, w), 'bilinear')
            mv16 = model('encode_value', frames[:, i], pm, s16)
            model('memorize', qk16, mv16, hard_pred_mask, pm)
        preds.append(pred)
    ops.check_faults()
    return preds


def evaluate_davis_seq_ms(model, frames, init_masks, out_size, scales=(480,), is_flip=False):
    """swem_evaluator.py:34-57: multi-scale / flip test-time augmentation: the probability maps of every pass are
    averaged, the index map is their argmax.  frames (1,T,C,H,W)."""
    assert len(scales) > 0
    final = None
    masks = [m for m in init_masks if m is not None]
    for scale in scales:
        h, w = scale, int((scale / 480) * 864)
        in_frames = ops.resize_planes(frames[0].contiguous(), (h, w), 'bicubic').unsqueeze(0)
        _, scores = evaluate_davis_seq(model, in_frames, init_masks, out_size)
        if is_flip:
            flipped = ops.flip_w(in_frames)
            fmasks = [ops.flip_w(m.float().contiguous()) for m in masks]
            _, fscores = evaluate_davis_seq(model, flipped, fmasks + [None] * (frames.shape[1] - len(fmasks)), out_size)
            scores = [ops.lincomb(a, 0.5, ops.flip_w(bf), 0.5) for a, bf in zip(scores, fscores)]
        k = 1.0 / len(scales)
        final = [ops.lincomb(sc, k) for sc in scores] if final is None else \
            [ops.lincomb(f, 1.0, sc, k) for f, sc in zip(final, scores)]
    return [ops.argmax_onehot(f, want_onehot=False)[0] for f in final]


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

    def __init__(self, model, frame_shape, out_size, streams=None):
        """streams: (warm-up stream, capture stream) to reuse (SequencePool keeps one pair per lane: a DAVIS / YouTube-VOS
        sweep re-captures for every new object count or frame shape, and HIP streams are never garbage collected)."""
        self.model, self.out_size = model, (int(out_size[0]), int(out_size[1]))
        self.streams = streams
        core = model.swem_core
        upd = core.memories['update'].bases
        if upd is None or core.memories['first'].bases is None:
            raise RuntimeError('FrameGraph needs an initialised memory with both banks (run two frames eagerly first)')
        dev = upd['kappa'].device
        self.frame = torch.empty(frame_shape, dtype=torch.float32, device=dev)
        self.state = {k: v.clone() for k, v in upd.items()}
        core.memories['update'].bases = self.state
        self.first = core.memories['first'].bases           # the captured kernels read these tensors in place
        self.graph = torch.cuda.CUDAGraph()
        self.pred = None

    def capture(self, example_frame):
        core = self.model.swem_core
        self.frame.copy_(example_frame)
        with torch.no_grad():
            # one eager pass on a side stream (also sizes every workspace), then restore the state it consumed
            saved = {k: v.clone() for k, v in self.state.items()}
            if self.streams is None:
                self.streams = (ops.new_stream(), ops.new_stream())
            s = self.streams[0]
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                frame_step(self.model, self.frame, self.out_size)
            torch.cuda.current_stream().wait_stream(s)
            for k in self.state:
                self.state[k].copy_(saved[k])
            core.memories['update'].bases = self.state
            # matching's packed banks (SWEMCore._pack) are static buffers of the model: bring them in line with the restored
            # state now, so that the captured frame finds them current and contains no repacking
            self.pack = core.repack()
            # capture on a stream of this graph's own: scratch buffers are per stream (ops.workspace), and graphs that are
            # replayed concurrently must not share one (torch's default capture stream is one object for all captures)
            # (scratch requested while capturing is allocated inside the capture and owned by the graph: ops.workspace)
            self.capture_stream = self.streams[1]
            with torch.cuda.graph(self.graph, stream=self.capture_stream, **ops.graph_capture_kwargs()):
                self.pred = frame_step(self.model, self.frame, self.out_size)
                new = core.memories['update'].bases
                for k in self.state:
                    self.state[k].copy_(new[k])
            core.memories['update'].bases = self.state
            core.restamp()          # the captured memorize rewrites the pack's 'update' half together with the state
        return self

    def rebind(self):
        """Adopt the model's CURRENT memory (a new sequence initialised eagerly, same shapes) into the captured graph's
        static buffers, so one capture serves every sequence of that shape."""
        core = self.model.swem_core
        cur_first, cur_upd = core.memories['first'].bases, core.memories['update'].bases
        if cur_first is None or cur_upd is None or cur_first['kappa'].shape != self.first['kappa'].shape:
            return False
        if core._pack is not self.pack:        # the graph holds the addresses of the pack it was captured with
            return False
        core.repack()                          # (no-op when memorize kept it current, as it does)
        for k in self.first:
            if cur_first[k] is not self.first[k]:
                self.first[k].copy_(cur_first[k])
            if cur_upd[k] is not self.state[k]:
                self.state[k].copy_(cur_upd[k])
        core.memories['first'].bases = self.first
        core.memories['update'].bases = self.state
        core.restamp()
        return True

    def run(self, frame):
        """Stage the frame into the static input buffer and replay; returns the (static) int64 index map."""
        self.frame.copy_(frame)
        self.graph.replay()
        return self.pred


class PipelinedFrameGraph(FrameGraph):
    """FrameGraph with the frame software-pipelined for ONE sequence.  Within a sequence the only thing frame t needs from
    frame t-1 is its memory update, and only at `match`: frame t's key encoder is independent of it.  The captured graph
    therefore forks -- frame t-1's encode_value + memorize (its inputs kept in static "pending" buffers) run on a side branch
    under frame t's encode_key -- and joins in front of match.  Every kernel sees exactly the data of the sequential order:
    the index maps are bit-identical (tests/test_gpu_model.py); measured on config B, one sequence: 3.43 -> ~3.0 ms per frame.

    The first run() after capture() / rebind() processes its frame eagerly with the memorize deferred (there is nothing pending
    yet); flush() applies a pending memorize eagerly (the model's memory then equals the sequential loop's)."""

    def __init__(self, model, frame_shape, out_size, streams=None, side_stream=None):
        super().__init__(model, frame_shape, out_size, streams)
        self.side = side_stream
        self.pend = None            # static buffers: frame, soft masks, one-hot masks, key, 1/16 features of the pending frame
        self.primed = False

    def _front(self, frame):
        """encode_key .. segment of one frame; returns the index map and what its deferred memorize needs."""
        h, w = frame.shape[-2:]
        qk16, qv16, s16, s8, s4 = self.model('encode_key', frame)
        context, n = self.model('match', qk16, qv16)
        _, pred_mask = self.model('segment', n, context, s8, s4, None, self.out_size)
        pred, hard = ops.argmax_onehot(pred_mask, want_onehot=True)
        pm = ops.resize_planes(pred_mask, (h, w), 'bilinear')
        return pred, (frame, pm, hard, qk16, s16)

    def _back(self):
        """The pending frame's encode_value + memorize."""
        frame, pm, hard, qk16, s16 = self.pend
        self.model('memorize', qk16, self.model('encode_value', frame, pm, s16), hard, pm)

    def _stash(self, cur):
        if self.pend is None:
            self.pend = [t.clone() for t in cur]
        else:
            for dst, src in zip(self.pend, cur):
                if dst is not src:
                    dst.copy_(src)

    def _body(self, main, side):
        core = self.model.swem_core
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._back()
            new = core.memories['update'].bases
            for k in self.state:
                self.state[k].copy_(new[k])
        pred, cur = None, None
        qk16, qv16, s16, s8, s4 = self.model('encode_key', self.frame)        # under the side branch
        main.wait_stream(side)
        h, w = self.frame.shape[-2:]
        context, n = self.model('match', qk16, qv16)
        _, pred_mask = self.model('segment', n, context, s8, s4, None, self.out_size)
        pred, hard = ops.argmax_onehot(pred_mask, want_onehot=True)
        pm = ops.resize_planes(pred_mask, (h, w), 'bilinear')
        self._stash((self.frame, pm, hard, qk16, s16))
        return pred

    def capture(self, example_frame):
        core = self.model.swem_core
        self.frame.copy_(example_frame)
        with torch.no_grad():
            if self.streams is None:
                self.streams = (ops.new_stream(), ops.new_stream())
            if self.side is None:
                self.side = overlapping_streams(2)[1]
            warm, cap = self.streams
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):
                # a pending frame to size the buffers (any frame will do: everything it touches is restored below)
                saved = {k: v.clone() for k, v in self.state.items()}
                _, cur = self._front(self.frame)
                self._stash(cur)
                self._body(warm, self.side)                      # one eager pass: sizes every workspace of both branches
                warm.wait_stream(self.side)
            torch.cuda.current_stream().wait_stream(warm)
            for k in self.state:
                self.state[k].copy_(saved[k])
            core.memories['update'].bases = self.state
            self.pack = core.repack()
            self.capture_stream = cap
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.graph(self.graph, stream=cap, **ops.graph_capture_kwargs()):
                self.pred = self._body(cap, self.side)
            core.memories['update'].bases = self.state
            core.restamp()
        self.primed = False
        return self

    def rebind(self):
        ok = super().rebind()
        self.primed = False
        return ok

    def run(self, frame):
        if not self.primed:        # nothing pending yet: this frame eagerly, its memorize deferred
            with torch.no_grad():
                pred, cur = self._front(frame)
                self._stash(cur)
            self.primed = True
            return pred
        self.frame.copy_(frame)
        self.graph.replay()
        return self.pred

    def flush(self):
        """Apply the pending memorize (eagerly): afterwards the model's memory is that of the sequential loop."""
        if self.primed:
            with torch.no_grad():
                self._back()
                new = self.model.swem_core.memories['update'].bases
                for k in self.state:
                    self.state[k].copy_(new[k])
                self.model.swem_core.memories['update'].bases = self.state
                self.model.swem_core.restamp()
            self.primed = False


def frame_chain(model, keys, frame, out_size, memorize=True, em_stream=None):
    """The part of a frame that depends on the memory (swem_evaluator.py:77-97): match -> segment -> argmax / one-hot ->
    [bilinear -> encode_value -> memorize], from the frame's key-encoder outputs `keys` = (qk16, qv16, s16, s8, s4).
    em_stream: run the part of memorize that does not read the value map -- every E, W and key M step, 2T - 1 of its 2T
    launches (modules.py:129-163 need the key, the masks and the prior only) -- on that stream BESIDE encode_value, and only
    the value update behind it (SWEM.memorize_begin / memorize_end: the same blocks on the same data, identical results)."""
    h, w = frame.shape[-2:]
    qk16, qv16, s16, s8, s4 = keys
    context, n = model('match', qk16, qv16)
    _, pred_mask = model('segment', n, context, s8, s4, None, out_size)
    pred, hard = ops.argmax_onehot(pred_mask, want_onehot=memorize)
    if memorize:
        pm = ops.resize_planes(pred_mask, (h, w), 'bilinear')
        tok = None
        if em_stream is not None:
            main = torch.cuda.current_stream()
            em_stream.wait_stream(main)
            with torch.cuda.stream(em_stream):
                tok = model.memorize_begin(qk16, hard, pm)
        mv16 = model('encode_value', frame, pm, s16)
        if em_stream is not None:
            main.wait_stream(em_stream)
        if tok is None:
            model('memorize', qk16, mv16, hard, pm)
        else:
            if not torch.cuda.is_current_stream_capturing():
                for t_ in (tok['kappa'], tok['zita'], tok['z']):
                    t_.record_stream(main)
            model.memorize_end(tok, mv16)
    return pred


def key_item(keys, j):
    """Frame j's share of a batched encode_key result (NCHW-shaped views of NHWC maps), with the planes the batch carries."""
    from .modules import as_nchw, to_pixel_major
    return tuple(as_nchw(ops.batch_item(to_pixel_major(t), j)) for t in keys)


class LookaheadGraph:
    """k frames of ONE sequence per replay, with the key encoder batched over them.

    `encode_key` does not read the memory (swem_evaluator.py:75 vs :77) and the whole sequence is device-resident inside the
    timed region (basic_evaluator.py:157-176), so the key encoder of the NEXT k frames runs as one B = k pass -- a k-th of
    the launches, k times the grid: the B = 1 layers of the ResNet-50 trunk are 10-25 us launches that leave most of the
    chip idle -- while the k frames of the CURRENT group run their memory-dependent chains (match -> segment -> encode_value ->
    memorize) one after the other.  Two HIP graphs per buffer parity: `keys[p]` (stage k frames, one batched pass; its
    outputs live in that graph's pool) and `chain[p]` (the k frame chains reading keys[p]'s outputs); a replay runs
    chain[p] on the lane's stream and keys[1-p] for the following group on a side stream (overlap=True, one lane) or behind
    it on the same stream (several lanes: their streams already fill the hardware queues).  Every kernel sees the data of
    the sequential order; with batch-invariant plans (ops.PlanBook.fallback without a K-split) index maps and memory are those
    of the frame-by-frame loop bit for bit (tests/test_gpu_model.py)."""

    def __init__(self, model, frame_shape, out_size, k, streams=None, side_stream=None, overlap=True, em_overlap=False):
        self.model, self.k, self.out_size = model, int(k), (int(out_size[0]), int(out_size[1]))
        self.streams, self.side, self.overlap = streams, side_stream, overlap
        core = model.swem_core
        upd = core.memories['update'].bases
        if upd is None or core.memories['first'].bases is None:
            raise RuntimeError('LookaheadGraph needs an initialised memory with both banks (run two frames eagerly first)')
        dev = upd['kappa'].device
        self.frame_shape = tuple(frame_shape)                 # (1, 3, H, W)
        self.frames = [torch.empty((self.k,) + self.frame_shape[1:], dtype=torch.float32, device=dev) for _ in range(2)]
        self.state = {key: v.clone() for key, v in upd.items()}
        # the frames of a group write their new bases alternately into the second set and back into the first: no copy of
        # the update bank per frame (an odd k pays one copy back per group)
        self.state2 = {key: torch.empty_like(v) for key, v in self.state.items()}
        core.memories['update'].bases = self.state
        self.first = core.memories['first'].bases
        self.kg = [torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()]
        self.cg = [torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()]
        self.keys = [None, None]
        self.preds = [None, None]
        # em_overlap: the key half of every memorize (9 of its 10 launches: SWEM.memorize_begin) on a side stream beside the
        # frame's value encoder.  Identical results (tested); measured NEUTRAL on config B (chains 2.26 -> 2.23 ms per frame
        # alone, 2.71 -> 2.72 with the next group's key encoder beside them: the frame is bound by the kernel sum), so off
        self.em_stream = ops.new_stream() if em_overlap else None
        self.p = 0                   # parity of the group whose keys are ready (after prime() / run())
        self.primed = False

    def _chains(self, p):
        core = self.model.swem_core
        preds = []
        sets = (self.state, self.state2)
        for j in range(self.k):
            core._next_out = sets[(j + 1) % 2]
            preds.append(frame_chain(self.model, key_item(self.keys[p], j), self.frames[p][j:j + 1], self.out_size,
                                     em_stream=self.em_stream))
            core._next_out = None
            core.memories['update'].bases = sets[(j + 1) % 2]       # (the tensors memorize wrote, under their own names)
            core.restamp()
        if self.k % 2:
            for key in self.state:
                self.state[key].copy_(self.state2[key])
            core.memories['update'].bases = self.state
            core.restamp()
        return preds

    def capture(self, example_frames):
        """example_frames (k,3,H,W): any frames of the sequence's shape (everything the warm-up touches is restored)."""
        core = self.model.swem_core
        with torch.no_grad():
            if self.streams is None:
                self.streams = (ops.new_stream(), ops.new_stream())
            if self.side is None:
                # (the chains and the side branch replay on a PROBED pair: two streams may share a hardware queue, and the
                # stream the caller happens to be on was never probed against anything)
                self.side = tuple(overlapping_streams(2)) if self.overlap else ()
            warm, cap = self.streams
            saved = {key: v.clone() for key, v in self.state.items()}
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):
                for p in (0, 1):
                    self.frames[p].copy_(example_frames)
                # eager passes: size every workspace, let the batched layers' consumers report their split requests (the
                # producers then write the planes themselves) and -- while ops.AUTOTUNE is on -- tune the B = k layer shapes
                for _ in range(2):
                    self.keys[0] = self.model('encode_key', self.frames[0])
                    self._chains(0)
            torch.cuda.current_stream().wait_stream(warm)
            for p in (0, 1):
                for key in self.state:
                    self.state[key].copy_(saved[key])
                core.memories['update'].bases = self.state
                self.pack = core.repack()
                cap.wait_stream(torch.cuda.current_stream())
                with torch.cuda.graph(self.kg[p], stream=cap, **ops.graph_capture_kwargs()):
                    self.keys[p] = self.model('encode_key', self.frames[p])
                with torch.cuda.graph(self.cg[p], stream=cap, **ops.graph_capture_kwargs()):
                    self.preds[p] = self._chains(p)
                torch.cuda.current_stream().wait_stream(cap)
            for key in self.state:
                self.state[key].copy_(saved[key])
            core.memories['update'].bases = self.state
            core.restamp()
        self.primed = False
        return self

    def rebind(self):
        """Adopt the model's CURRENT memory (a new sequence of the same shapes) into the captured graphs' static buffers."""
        core = self.model.swem_core
        cur_first, cur_upd = core.memories['first'].bases, core.memories['update'].bases
        if cur_first is None or cur_upd is None or cur_first['kappa'].shape != self.first['kappa'].shape:
            return False
        if core._pack is not self.pack:
            return False
        core.repack()
        for key in self.first:
            if cur_first[key] is not self.first[key]:
                self.first[key].copy_(cur_first[key])
            if cur_upd[key] is not self.state[key]:
                self.state[key].copy_(cur_upd[key])
        core.memories['first'].bases = self.first
        core.memories['update'].bases = self.state
        core.restamp()
        self.primed = False
        return True

    def prime(self, frames_k):
        """Key-encoder pass of the first group (k,3,H,W)."""
        self.frames[self.p].copy_(frames_k)
        self.kg[self.p].replay()
        self.primed = True

    def run(self, next_frames_k=None):
        """The k frame chains of the group whose keys are ready; `next_frames_k` (k,3,H,W) = the following group, whose key
        encoder runs next to them (None: the sequence ends).  Returns the k (static) int64 index maps."""
        if not self.primed:
            raise RuntimeError('LookaheadGraph.run before prime()')
        p, main = self.p, torch.cuda.current_stream()
        if next_frames_k is not None:
            self.frames[1 - p].copy_(next_frames_k)
        if self.overlap and next_frames_k is not None:
            s0, s1 = self.side
            s0.wait_stream(main)
            s1.wait_stream(main)
            with torch.cuda.stream(s0):
                self.cg[p].replay()
            with torch.cuda.stream(s1):
                self.kg[1 - p].replay()
            main.wait_stream(s0)
            main.wait_stream(s1)
        else:
            self.cg[p].replay()
            if next_frames_k is not None:
                self.kg[1 - p].replay()
        if next_frames_k is not None:
            self.p = 1 - p
        else:
            self.primed = False
        return self.preds[p]


def key_items(keys, j, n):
    """Frames j .. j + n - 1 of a batched encode_key result (see key_item)."""
    from .modules import as_nchw, to_pixel_major
    return tuple(as_nchw(ops.batch_item(to_pixel_major(t), j, n)) for t in keys)


def lockstep_chain(models, keys, keys_each, frames, out_size, forks, outs=None, fuse_batched=False):
    """frame_chain for ONE frame of each of S sequences in lock step (round 6): `models[s]` holds sequence s's memory,
    `keys` = (qk16, qv16, s16, s8, s4) with batch S (models[0]'s key encoder over the S frames), keys_each[s] = sequence s's item of
    them (key_item: with the planes the batch carries), `frames` (S,3,H,W).
    What depends on one sequence's memory only -- match (affinity, top-l, readout, fusion conv) and memorize (EM) -- runs per
    sequence on `forks[s]` (S streams that overlap: evaluator.overlapping_streams), exactly the launches of frame_chain; the
    decoder and the value encoder, whose layers treat the objects as a batch (swem.py:52-53, 94-95), run ONCE for the S * N
    objects of all sequences on the caller's stream through models[0]'s engine (the models are replicas: same weights, one
    PlanBook) -- 4 x 3,240 rows per 1/16-scale layer instead of four launches of 3,240.  Needs the same N in every
    sequence.  outs[s]: the tensors sequence s's new bases go to (LookaheadGraph's alternating state sets).
    Returns the (S,Ho,Wo) int64 index maps."""
    main = torch.cuda.current_stream()
    h, w = frames.shape[-2:]
    qk16, qv16, s16, s8, s4 = keys
    S = len(models)
    ctxs, n = [], None
    forks = forks or [main] * S          # (no forks: the per-sequence parts one after the other on the caller's stream)
    for s, (m, st) in enumerate(zip(models, forks)):
        if st is not main:
            st.wait_stream(main)
        with torch.cuda.stream(st):
            if fuse_batched:        # affinity / top-l / readout per sequence; the fusion conv (modules.py:286-291) below, batched
                core = m.swem_core
                with ops.use_book(m.book):
                    c = core._affinity_readout(keys_each[s][0], core.memories['first'].bases, core.memories['update'].bases)
                n_s = core.memories['first'].bases['kappa'].shape[1]
            else:
                c, n_s = m('match', keys_each[s][0], keys_each[s][1])
        if n is not None and n_s != n:
            raise RuntimeError('lockstep_chain: the sequences hold %d and %d objects' % (n, n_s))
        n = n_s
        ctxs.append(c)
    for st in forks:
        if st is not main:
            main.wait_stream(st)
    m0 = models[0]
    if fuse_batched:
        with ops.use_book(m0.book):
            context = m0.engine().fuse_context(torch.cat([c[1].contiguous() for c in ctxs]), to_pixel_major_(qv16),
                                               torch.cat([c[0] for c in ctxs]))
    else:
        context = torch.cat([to_pixel_major_(c) for c in ctxs])
    _, pred_mask = m0('segment', n, as_nchw_(context), s8, s4, None, out_size)
    pred, hard = ops.argmax_onehot(pred_mask, want_onehot=True)
    pm = ops.resize_planes(pred_mask, (h, w), 'bilinear')
    mv16 = m0('encode_value', frames, pm, s16)                      # (S,N,V,h,w)
    for s, (m, st) in enumerate(zip(models, forks)):
        if st is not main:
            st.wait_stream(main)
        with torch.cuda.stream(st):
            if outs is not None:
                m.swem_core._next_out = outs[s]
            m('memorize', keys_each[s][0], mv16[s:s + 1], hard[s:s + 1], pm[s:s + 1])
            m.swem_core._next_out = None
    for st in forks:
        if st is not main:
            main.wait_stream(st)
    return pred


def lockstep_chain_batched(lane, keys, frames, out_size, nxt):
    """lockstep_chain with EM and matching batched over the lane's sequences as well (LockstepGraph(batched_em=True)): the lane keeps
    the sequences' banks as slices of ONE tensor per bank and their packs as slices of ONE pack, so that affinity / top-l / readout
    and the whole memorize run as single launches over the S * N objects with one key map per sequence
    (swem_match_packed_clips_f32, swem_memorize_packed_clips_f32: per object the same blocks on the same data), and the fusion conv
    takes the matching kernels' own output planes for all objects.  `nxt`: the per-sequence dicts of the state set the new bases go
    to (views of the lane-wide tensors `lane.sets_all[...]`).  Steady state only (both banks, the packs current)."""
    models, S = lane.models, lane.S
    m0 = models[0]
    cores = [m.swem_core for m in models]
    core0 = cores[0]
    h, w = frames.shape[-2:]
    qk16, qv16, s16, s8, s4 = keys
    Ck, h16, w16 = qk16.shape[1:]
    P, L = h16 * w16, core0.n_bases
    cur_all = lane.sets_all[lane.cur]
    nxt_all = lane.sets_all[1 - lane.cur]
    N = cur_all['kappa'].shape[1]
    with ops.use_book(m0.book):
        for c in cores:     # (under the lane's book: whether the packs carry the fp16 value planes depends on the book's readout)
            if not (c._stamped(0, c.memories['first'].bases) and c._stamped(1, c.memories['update'].bases)):
                c.repack()
        xq = to_pixel_major_(qk16).view(S, P, Ck)
        mem_img, s_img = ops.match_packed(xq, lane.pack_all, L, core0.topl, core0.tau, hw=(h16, w16), clips=S)
        context = m0.engine().fuse_context(mem_img, to_pixel_major_(qv16), s_img)          # (S*N,h,w,V)
    _, pred_mask = m0('segment', N, as_nchw_(context), s8, s4, None, out_size)
    pred, hard = ops.argmax_onehot(pred_mask, want_onehot=True)
    pm = ops.resize_planes(pred_mask, (h, w), 'bilinear')
    mv16 = m0('encode_value', frames, pm, s16)                      # (S,N,V,h16,w16)
    with ops.use_book(m0.book):
        masks = ops.mask_prep(hard.contiguous(), pm.float().contiguous(), h16, w16)          # (S*N,2,P)
        vp = to_pixel_major_(mv16.flatten(0, 1)).view(S * N, P, -1)
        ops.memorize(xq, vp, masks, cur_all['kappa'].view(S * N, 2, Ck, L), cur_all['nu'].view(S * N, 2, -1, L),
                     cur_all['zita'].view(S * N, 2, L), core0.n_iters, core0.tau, pack=lane.pack_all, prior_packed=True, bank=1,
                     out=(nxt_all['kappa'].view(S * N, 2, Ck, L), nxt_all['nu'].view(S * N, 2, -1, L),
                          nxt_all['zita'].view(S * N, 2, L)), clips=S)
    for c, st in zip(cores, nxt):
        c.memories['update'].bases = st       # (the lane-wide tensors memorize wrote, under the sequence's own views)
        c.restamp()
    lane.cur = 1 - lane.cur
    return pred


def to_pixel_major_(t):
    from .modules import to_pixel_major
    return to_pixel_major(t)


def as_nchw_(t):
    from .modules import as_nchw
    return as_nchw(t)


class LockstepGraph:
    """k frames of S sequences per replay, the sequences in lock step (round 6; VERDICT r05 item 2b taken to four sequences).

    LookaheadGraph with a second batch axis: ONE key-encoder pass over the k x S frames of the next group (frame-major: the S
    frames of step j are contiguous), and k lockstep_chain steps per replay -- per sequence: match and memorize on S forked
    streams inside the graph; batched over the S * N objects: decoder and value encoder.  `models` are replicas (same
    weights, one PlanBook), each holding one sequence's memory (both banks initialised, the same number of objects)."""

    def __init__(self, models, frame_shape, out_size, k, streams=None, side_stream=None, overlap=True, forks=None, fuse_batched=False,
                 batched_em=False):
        self.models, self.k, self.out_size = list(models), int(k), (int(out_size[0]), int(out_size[1]))
        self.S = len(self.models)
        self.fuse_batched = fuse_batched
        self.batched_em = batched_em
        self.streams, self.side, self.overlap, self.forks = streams, side_stream, overlap, forks
        cores = [m.swem_core for m in self.models]
        for c in cores:
            if c.memories['update'].bases is None or c.memories['first'].bases is None:
                raise RuntimeError('LockstepGraph needs an initialised memory with both banks in every model')
        shapes = {tuple(c.memories['first'].bases['kappa'].shape) for c in cores}
        if len(shapes) != 1:
            raise RuntimeError('LockstepGraph: the sequences must hold the same number of objects')
        for m in self.models[1:]:
            m.book = self.models[0].book
        dev = cores[0].memories['update'].bases['kappa'].device
        self.frame_shape = tuple(frame_shape)                 # (1, 3, H, W)
        self.frames = [torch.empty((self.k, self.S) + self.frame_shape[1:], dtype=torch.float32, device=dev) for _ in range(2)]
        self.state = [{key: v.clone() for key, v in c.memories['update'].bases.items()} for c in cores]
        self.state2 = [{key: torch.empty_like(v) for key, v in st.items()} for st in self.state]
        for c, st in zip(cores, self.state):
            c.memories['update'].bases = st
        self.first = [c.memories['first'].bases for c in cores]
        if batched_em:
            self._share_state(cores, dev)
        self.kg = [torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()]
        self.cg = [torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()]
        self.keys = [None, None]
        self.preds = [None, None]
        self.packs = None
        self.p = 0
        self.primed = False

    def _share_state(self, cores, dev):
        """batched_em: the sequences' banks as slices of ONE tensor per bank (the reference's own layout with B = S:
        kappa (S,N,2,C,L), modules.py:129-168) and their packs as slices of one pack -- every sequence's core keeps working on
        its own views (eager frames, re-binding), the lane's batched launches on the whole."""
        S = self.S
        stack = lambda dicts: {key: torch.cat([d[key] for d in dicts]).contiguous() for key in dicts[0]}
        first_all, state_all = stack(self.first), stack(self.state)
        state2_all = {key: torch.empty_like(v) for key, v in state_all.items()}
        views = lambda whole: [{key: v[s_:s_ + 1] for key, v in whole.items()} for s_ in range(S)]
        self.first, self.state, self.state2 = views(first_all), views(state_all), views(state2_all)
        self.sets_all, self.cur = (state_all, state2_all), 0
        N, _, Ck, L = first_all['kappa'].shape[1:]
        self.pack_all = ops.new_pack(S * N, Ck, cores[0].valdim, L, dev)
        for s_, c in enumerate(cores):
            c.memories['first'].bases = self.first[s_]
            c.memories['update'].bases = self.state[s_]
            c._pack = (self.pack_all[0][2 * N * s_:2 * N * (s_ + 1)], self.pack_all[1][N * s_:N * (s_ + 1)],
                       self.pack_all[2][N * s_:N * (s_ + 1)])
            c._stamp = [None, None]

    def _encode(self, p):
        return self.models[0]('encode_key', self.frames[p].view((self.k * self.S,) + self.frame_shape[1:]))

    def _chains(self, p):
        cores = [m.swem_core for m in self.models]
        preds = []
        sets = (self.state, self.state2)
        if self.batched_em:
            self.cur = 0            # (a group starts from the first state set: `state`)
            for j in range(self.k):
                preds.append(lockstep_chain_batched(self, key_items(self.keys[p], j * self.S, self.S), self.frames[p][j],
                                                    self.out_size, sets[(j + 1) % 2]))
        for j in range(self.k if not self.batched_em else 0):
            nxt = sets[(j + 1) % 2]
            each = [key_item(self.keys[p], j * self.S + s) for s in range(self.S)]
            preds.append(lockstep_chain(self.models, key_items(self.keys[p], j * self.S, self.S), each, self.frames[p][j],
                                        self.out_size, self.forks, outs=nxt, fuse_batched=self.fuse_batched))
            for c, st in zip(cores, nxt):
                c.memories['update'].bases = st       # (the tensors memorize wrote, under their own names)
                c.restamp()
        if self.k % 2:
            for c, a, b in zip(cores, self.state, self.state2):
                for key in a:
                    a[key].copy_(b[key])
                c.memories['update'].bases = a
                c.restamp()
        return preds

    def capture(self, example_frames):
        """example_frames (k,S,3,H,W): any frames of the sequences' shape (everything the warm-up touches is restored)."""
        cores = [m.swem_core for m in self.models]
        with torch.no_grad():
            if self.streams is None:
                self.streams = (ops.new_stream(), ops.new_stream())
            if self.forks is None:
                self.forks = overlapping_streams(self.S)
            elif self.forks == 'none':
                self.forks = ()
            if self.side is None:
                self.side = tuple(overlapping_streams(2)) if self.overlap else ()
            warm, cap = self.streams
            saved = [{key: v.clone() for key, v in st.items()} for st in self.state]

            def restore():
                for c, st, sv in zip(cores, self.state, saved):
                    for key in st:
                        st[key].copy_(sv[key])
                    c.memories['update'].bases = st
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):
                for p in (0, 1):
                    self.frames[p].copy_(example_frames)
                for _ in range(2):       # eager passes: workspaces, split requests, and -- while ops.AUTOTUNE is on -- the batched shapes' plans
                    self.keys[0] = self._encode(0)
                    self._chains(0)
            torch.cuda.current_stream().wait_stream(warm)
            for p in (0, 1):
                restore()
                with ops.use_book(self.models[0].book):      # (the packs with the value planes THIS book's readout reads)
                    self.packs = [c.repack() for c in cores]
                cap.wait_stream(torch.cuda.current_stream())
                with torch.cuda.graph(self.kg[p], stream=cap, **ops.graph_capture_kwargs()):
                    self.keys[p] = self._encode(p)
                with torch.cuda.graph(self.cg[p], stream=cap, **ops.graph_capture_kwargs()):
                    self.preds[p] = self._chains(p)
                torch.cuda.current_stream().wait_stream(cap)
            restore()
            for c in cores:
                c.restamp()
        self.primed = False
        return self

    def rebind(self):
        """Adopt the models' CURRENT memories (new sequences of the same shapes) into the captured graphs' static buffers."""
        cores = [m.swem_core for m in self.models]
        for c, first, pack in zip(cores, self.first, self.packs or [None] * self.S):
            cur_first, cur_upd = c.memories['first'].bases, c.memories['update'].bases
            if cur_first is None or cur_upd is None or cur_first['kappa'].shape != first['kappa'].shape or c._pack is not pack:
                return False
        for c, first, state in zip(cores, self.first, self.state):
            cur_first, cur_upd = c.memories['first'].bases, c.memories['update'].bases
            with ops.use_book(self.models[0].book):
                c.repack()
            for key in first:
                if cur_first[key] is not first[key]:
                    first[key].copy_(cur_first[key])
                if cur_upd[key] is not state[key]:
                    state[key].copy_(cur_upd[key])
            c.memories['first'].bases = first
            c.memories['update'].bases = state
            c.restamp()
        self.primed = False
        return True

    def prime(self, frames_kS):
        """Key-encoder pass of the first group (k,S,3,H,W)."""
        self.frames[self.p].copy_(frames_kS)
        self.kg[self.p].replay()
        self.primed = True

    def run(self, next_frames_kS=None):
        """The k lock-step frames of the group whose keys are ready; `next_frames_kS` (k,S,3,H,W) = the following group, whose key
        encoder runs next to them (None: the sequences end).  Returns k (static) (S,Ho,Wo) int64 index maps."""
        if not self.primed:
            raise RuntimeError('LockstepGraph.run before prime()')
        p, main = self.p, torch.cuda.current_stream()
        if next_frames_kS is not None:
            self.frames[1 - p].copy_(next_frames_kS)
        if self.overlap and next_frames_kS is not None:
            s0, s1 = self.side
            s0.wait_stream(main)
            s1.wait_stream(main)
            with torch.cuda.stream(s0):
                self.cg[p].replay()
            with torch.cuda.stream(s1):
                self.kg[1 - p].replay()
            main.wait_stream(s0)
            main.wait_stream(s1)
        else:
            self.cg[p].replay()
            if next_frames_kS is not None:
                self.kg[1 - p].replay()
        if next_frames_kS is not None:
            self.p = 1 - p
        else:
            self.primed = False
        return self.preds[p]


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


_PROBED = {}


def overlapping_streams(n, device=None, tries=12):
    """n HIP streams that really run concurrently.  HIP multiplexes streams onto a few hardware queues, and two streams on
    one queue execute strictly one after the other (measured: two sequences on two such streams took exactly twice the time
    of one).  There is no query for the queue of a stream, so candidates are probed: a short chain of small kernels is
    replayed on a pair, and a candidate is kept if the pair finishes in well under twice the single-stream time.
    The probe is timed ON THE GPU (HIP events around the replays, which are held back behind a short spin kernel until the
    host has enqueued all of them), not with the host clock: on an 8-GPU node eight ranks share one container's CPU quota,
    and a host that is late by a replay's 0.2 ms would make every pair look serial.  Found streams are
    cached per device for the life of the process."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device
    have = _PROBED.setdefault(dev.index, [])       # streams found earlier in this process: probe only for the missing ones
    if len(have) >= n:
        return list(have[:n])
    buf = [torch.zeros(1 << 18, dtype=torch.float32, device=dev) for _ in range(tries + n)]

    def chain(st, k):
        g = torch.cuda.CUDAGraph()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            ops.lincomb(buf[k], 1.0)
            st.synchronize()
            with torch.cuda.graph(g, stream=st, **ops.graph_capture_kwargs()):
                for _ in range(60):
                    ops.lincomb(buf[k], 1.0)
        return g

    def run(pairs):
        """GPU time (ms) from the common start to the last stream's end, best of five."""
        main = torch.cuda.current_stream()
        best = None
        for _ in range(5):
            torch.cuda.synchronize()
            start = torch.cuda.Event(enable_timing=True)
            ends = [torch.cuda.Event(enable_timing=True) for _ in pairs]
            torch.cuda._sleep(2_000_000)                 # ~1 ms: the replays below are all enqueued before it ends
            start.record(main)
            for (st, g), e in zip(pairs, ends):
                st.wait_event(start)
                with torch.cuda.stream(st):
                    g.replay()
                    e.record(st)
            torch.cuda.synchronize()
            dt = max(start.elapsed_time(e) for e in ends)
            best = dt if best is None else min(best, dt)
        return best
    chosen = [(st, chain(st, i)) for i, st in enumerate(have)]
    k = len(chosen)
    if not chosen:
        first = ops.new_stream()
        chosen.append((first, chain(first, 0)))
        k = 1
    t1 = run(chosen[:1])
    while len(chosen) < n and k < tries + n:
        cand = ops.new_stream()
        pair = (cand, chain(cand, k))
        k += 1
        # (measured with this probe: pairs on different queues 1.3-1.5 x the single-stream time, pairs on one queue 1.8-1.9 x)
        ratios = [run([c, pair]) / t1 for c in chosen]
        if os.environ.get('SWEM_PROBE_DEBUG'):
            print('probe: candidate %d ratios %s' % (k, ' '.join('%.2f' % v for v in ratios)), file=sys.stderr)
        if all(v < 1.65 for v in ratios):
            chosen.append(pair)
    while len(chosen) < n:                     # no further overlapping candidate found: fall back to plain streams
        chosen.append((ops.new_stream(), None))
    have[:] = [st for st, _ in chosen]
    return list(have)


class SequencePool:
    """Several sequences in flight on one GPU, each on its own stream with its own model instance (memory banks):
    sequences are independent and one sequence alone leaves the GPU under-filled (204 blocks in the
    EM kernels, ~300 short launches per frame), so a second one's kernels fill the gaps (+18..25 % frames/s, bench.py).
    The lanes' streams are probed for real concurrency (`overlapping_streams`): two streams can share a hardware queue.
    After a sequence's first two frames (eager: they build the two banks) the steady state runs from HIP graphs captured once
    per lane and re-bound to each new sequence of the same shape: `lookahead` frames per replay with the key encoder batched
    over them (LookaheadGraph; with ONE lane the next group's key encoder runs on a side stream next to the current group's
    frame chains), the last frames of a sequence that do not fill a group eagerly.  lookahead = 0: one frame per replay
    (FrameGraph; a pool of one model then runs the software-pipelined PipelinedFrameGraph)."""

    def __init__(self, models, use_graph=True, lookahead=8, plans='shipped'):
        self.models = list(models)
        n = len(self.models)
        for m in self.models[1:]:            # the lanes run the same layers on the same shapes: one PlanBook for all of them
            m.book = self.models[0].book
        # plans='shipped' (default): a pool whose book holds no tuned conv plan yet loads the plan file that ships with the
        # library (swem_amd/plans/: 480p, K = 256, 1-5 objects on MI355X) -- layer shapes it does not hold run the
        # book's fallback (f16x3 on the heuristic tile).  plans=None: the book as it is; a path: that file.
        # The shipped file is f16x3 plans tuned on an MI355X: it is NOT loaded into a book whose owner chose another default
        # arithmetic (book.fallback = 0: the exact fp32 kernels; a book that has left the fp16 range) nor on a device of
        # another architecture -- those run the book as it is.
        book = self.models[0].book
        dev = next(self.models[0].parameters()).device
        if plans is not None and not book.conv and (plans != 'shipped' or
                                                    ((book.fallback >> 16) & 7 == 7 and not book.full_range)):
            import os
            path = ops.shipped_plans() if plans == 'shipped' else plans
            if os.path.exists(path):
                book.load(path, device=dev if plans == 'shipped' else None)
        self.streams = overlapping_streams(n) if n > 1 else [torch.cuda.current_stream()]
        self.graphs = [None] * n
        self.graph_streams = [None] * n      # per lane: (warm-up stream, capture stream), reused by every re-capture
        self.use_graph = use_graph
        self.lookahead = int(lookahead)

    def _graph_for(self, li, frames, i, out_size):
        """The lane's graph for this sequence's steady state, (re-)captured if its shapes changed; None if it cannot be used."""
        model, g, k = self.models[li], self.graphs[li], self.lookahead
        shape = tuple(frames[:, i].shape)
        if k > 0:
            ok = isinstance(g, LookaheadGraph) and g.frame_shape == shape and g.out_size == out_size and g.k == k
        else:
            ok = g is not None and not isinstance(g, LookaheadGraph) and g.frame.shape == frames[:, i].shape and g.out_size == out_size
        if ok and g.rebind():
            return g
        if model.swem_core.memories['update'].bases is None:
            return None
        self.graphs[li] = None                # the replaced graph (and its private pool) goes first
        if k > 0:
            g = LookaheadGraph(model, shape, out_size, k, streams=self.graph_streams[li], overlap=len(self.models) == 1)
            g.capture(frames[0, i:i + k])
        else:
            # one lane: the frame software-pipelined (+12 % frames/s); several lanes already fill the hardware queues, a
            # forked graph per lane costs ~10 % there (bench.py --pipeline)
            cls = PipelinedFrameGraph if len(self.models) == 1 else FrameGraph
            g = cls(model, shape, out_size, streams=self.graph_streams[li])
            g.capture(frames[:, i])
        self.graphs[li], self.graph_streams[li] = g, g.streams
        return g

    def run(self, sequences, seeds=None):
        """sequences: list of (frames (1,T,3,H,W), init_mask (1,N+1,Ho,Wo), out_size); returns one list of (1,Ho,Wo)
        int64 index maps per sequence (frames 1..T-1), in input order.  seeds: optional torch seed per sequence, set
        right before its memory is initialised (reproducible random bases whatever the interleaving).
        A range fault of the f16x3 arithmetic (ops.SwemRangeError at the final check) moves the lanes' shared book to the
        full-range arithmetic, drops the captured graphs (they hold f16x3 launches) and runs the call's sequences again."""
        try:
            return self._run(sequences, seeds)
        except ops.SwemRangeError as err:
            range_fallback(self.models[0], err, 'SequencePool.run')
            self.graphs = [None] * len(self.models)
            return self._run(sequences, seeds)

    def _run(self, sequences, seeds=None):
        ops.drain_faults('SequencePool.run')
        todo = list(enumerate(sequences))
        results = [None] * len(sequences)
        k = self.lookahead
        lanes = [None] * len(self.models)          # per lane: [seq index, frames, out_size, next frame, preds, bound graph]
        main = torch.cuda.current_stream()
        for st in self.streams:
            st.wait_stream(main)
        with torch.no_grad():
            while todo or any(l is not None for l in lanes):
                for li, (model, st) in enumerate(zip(self.models, self.streams)):
                    with torch.cuda.stream(st):
                        if lanes[li] is None:
                            if not todo:
                                continue
                            si, (frames, init_mask, out_size) = todo.pop(0)
                            if seeds is not None:
                                torch.manual_seed(seeds[si])
                            h, w = frames.shape[-2:]
                            mk16, _, s16, _, _ = model('encode_key', frames[:, 0])
                            m0 = ops.resize_planes(init_mask.float().contiguous(), (h, w), 'nearest')
                            model('init', mk16, model('encode_value', frames[:, 0], m0, s16), init_mask)
                            # (a graph is bound to ONE sequence of ONE run() call: the record below is unique per sequence)
                            lanes[li] = [si, frames, (int(out_size[0]), int(out_size[1])), 1, [], None]
                            if frames.shape[1] <= 1:        # (a one-frame sequence: nothing to segment, swem_evaluator.py:72)
                                results[si], lanes[li] = [], None
                            continue
                        si, frames, out_size, i, preds, bound = lanes[li]
                        t = frames.shape[1]
                        step = 1
                        # A captured frame always memorizes; the reference's loop does not memorize a sequence's LAST frame
                        # (swem_evaluator.py:89: `if i < t - 1`).  The graphs therefore never take the last frame: a group of k
                        # frames runs from the graph only while MORE than k frames remain, the tail (at most k frames) eagerly --
                        # no wasted encode_value + memorize per sequence, and the model's memory after run() is the eager
                        # loop's.  (The software-pipelined one-frame graph defers every memorize by a frame and
                        # simply never applies the last one.)
                        if self.use_graph and i >= 2 and bound is None and t - i > max(k, 1):
                            bound = lanes[li][5] = self._graph_for(li, frames, i, out_size)
                            if bound is not None and k > 0:
                                bound.prime(frames[0, i:i + k])
                        if bound is not None and k > 0 and t - i > k:
                            nxt = frames[0, i + k:i + 2 * k] if t - i > 2 * k else None
                            preds.extend(p_.clone() for p_ in bound.run(nxt))
                            step = k
                            if nxt is None:
                                bound = lanes[li][5] = None        # the rest of the sequence (at most k frames) eagerly
                        elif bound is not None and k == 0 and (i < t - 1 or isinstance(bound, PipelinedFrameGraph)):
                            preds.append(bound.run(frames[:, i]).clone())
                        else:
                            preds.append(frame_step(model, frames[:, i], out_size, memorize=i < t - 1))
                        lanes[li][3] = i + step
                        if i + step >= t:
                            results[si] = preds
                            lanes[li] = None
        for st in self.streams:
            main.wait_stream(st)
        ops.check_faults()       # (synchronises: the results are about to be read)
        return results


class LockstepPool:
    """SequencePool for sequences that can march in LOCK STEP (round 6): len(models) / lockstep lanes, each a pipeline of `lockstep`
    sequences of the same frame size, length, mask shape and output size (LockstepGraph: one key-encoder pass over
    lockstep x lookahead frames, decoder and value encoder batched over the objects of all its sequences; match and memorize per
    sequence).  run() cuts its sequences into such groups; what does not fill a group runs on a SequencePool over the first
    models.  `models` are replicas (same weights); they share models[0]'s PlanBook.  Measured on config B (two lanes of four
    sequences against four independent pipelines): bench.py, DESIGN.md section 5."""

    def __init__(self, models, lockstep=4, use_graph=True, lookahead=10, plans='shipped', batched_em=True):
        # batched_em (default): a lane runs EM and matching -- and with them the fusion conv -- once for the objects of all its
        # sequences (LockstepGraph(batched_em=True)); False: per sequence (bit-identical to the per-sequence loops under
        # batch-invariant plans; the batched fusion conv picks its tile by ITS row count)
        self.batched_em = bool(batched_em)
        self.models, self.S = list(models), int(lockstep)
        if self.S < 2 or len(self.models) % self.S:
            raise ValueError('LockstepPool: %d models do not make lanes of %d sequences' % (len(self.models), self.S))
        self.rest = SequencePool(self.models[:min(len(self.models), 4)], use_graph=use_graph, lookahead=lookahead, plans=plans)
        for m in self.models[1:]:
            m.book = self.models[0].book
        self.lanes = [self.models[i:i + self.S] for i in range(0, len(self.models), self.S)]
        self.streams = overlapping_streams(len(self.lanes)) if len(self.lanes) > 1 else [torch.cuda.current_stream()]
        self.graphs = [None] * len(self.lanes)
        self.graph_streams = [None] * len(self.lanes)
        self.use_graph, self.lookahead = use_graph, int(lookahead)

    def _graph_for(self, li, frames_list, i, out_size):
        models, g, k = self.lanes[li], self.graphs[li], self.lookahead
        shape = tuple(frames_list[0][:, i].shape)
        if isinstance(g, LockstepGraph) and g.frame_shape == shape and g.out_size == out_size and g.k == k and g.rebind():
            return g
        if any(m.swem_core.memories['update'].bases is None for m in models):
            return None
        if len({tuple(m.swem_core.memories['first'].bases['kappa'].shape) for m in models}) != 1:
            return None                       # (sequences with different numbers of objects: the lane runs them frame by frame)
        self.graphs[li] = None
        g = LockstepGraph(models, shape, out_size, k, streams=self.graph_streams[li], overlap=False, forks='none',
                          batched_em=self.batched_em)
        g.capture(torch.stack([f[0, i:i + k] for f in frames_list], dim=1))
        self.graphs[li], self.graph_streams[li] = g, g.streams
        return g

    def run(self, sequences, seeds=None):
        """As SequencePool.run (same arguments, same results layout, same range-fault fallback)."""
        try:
            return self._run(sequences, seeds)
        except ops.SwemRangeError as err:
            range_fallback(self.models[0], err, 'LockstepPool.run')
            self.graphs = [None] * len(self.lanes)
            self.rest.graphs = [None] * len(self.rest.models)
            return self._run(sequences, seeds)

    def _run(self, sequences, seeds=None):
        ops.drain_faults('LockstepPool.run')
        S, k = self.S, self.lookahead
        buckets = {}
        for si, (frames, init_mask, out_size) in enumerate(sequences):
            key = (tuple(frames.shape), tuple(init_mask.shape), (int(out_size[0]), int(out_size[1])))
            buckets.setdefault(key, []).append(si)
        todo, rest = [], []
        for idx in buckets.values():
            while len(idx) >= S and k > 0 and self.use_graph:
                todo.append(idx[:S])
                idx = idx[S:]
            rest += idx
        results = [None] * len(sequences)
        state = [None] * len(self.lanes)            # per lane: [sequence indices, next frame, preds per sequence, bound graph]
        main = torch.cuda.current_stream()
        for st in self.streams:
            st.wait_stream(main)
        with torch.no_grad():
            while todo or any(l is not None for l in state):
                for li, (models, st) in enumerate(zip(self.lanes, self.streams)):
                    with torch.cuda.stream(st):
                        if state[li] is None:
                            if not todo:
                                continue
                            chunk = todo.pop(0)
                            for m, si in zip(models, chunk):
                                frames, init_mask, _ = sequences[si]
                                if seeds is not None:
                                    torch.manual_seed(seeds[si])
                                h, w = frames.shape[-2:]
                                mk16, _, s16, _, _ = m('encode_key', frames[:, 0])
                                m0 = ops.resize_planes(init_mask.float().contiguous(), (h, w), 'nearest')
                                m('init', mk16, m('encode_value', frames[:, 0], m0, s16), init_mask)
                            state[li] = [chunk, 1, [[] for _ in chunk], None]
                            if sequences[chunk[0]][0].shape[1] <= 1:        # (one-frame sequences: nothing to segment)
                                for si in chunk:
                                    results[si] = []
                                state[li] = None
                            continue
                        chunk, i, preds, bound = state[li]
                        fl = [sequences[si][0] for si in chunk]
                        out_size = (int(sequences[chunk[0]][2][0]), int(sequences[chunk[0]][2][1]))
                        t = fl[0].shape[1]
                        step = 1
                        # (as SequencePool: a captured frame always memorizes, the reference's loop does not memorize a sequence's
                        # last frame -- a group runs from the graph only while MORE than k frames remain, the tail eagerly)
                        if i >= 2 and bound is None and t - i > k:
                            bound = state[li][3] = self._graph_for(li, fl, i, out_size)
                            if bound is not None:
                                bound.prime(torch.stack([f[0, i:i + k] for f in fl], dim=1))
                        if bound is not None and t - i > k:
                            nxt = torch.stack([f[0, i + k:i + 2 * k] for f in fl], dim=1) if t - i > 2 * k else None
                            for p_ in bound.run(nxt):
                                for s_ in range(S):
                                    preds[s_].append(p_[s_:s_ + 1].clone())
                            step = k
                            if nxt is None:
                                bound = state[li][3] = None
                        else:
                            for s_, m in enumerate(models):
                                preds[s_].append(frame_step(m, fl[s_][:, i], out_size, memorize=i < t - 1))
                        state[li][1] = i + step
                        if i + step >= t:
                            for s_, si in enumerate(chunk):
                                results[si] = preds[s_]
                            state[li] = None
        for st in self.streams:
            main.wait_stream(st)
        ops.check_faults()       # (synchronises; a range fault of the lanes raises here, before the rest would drain it)
        if rest:
            sub = self.rest._run([sequences[si] for si in rest], None if seeds is None else [seeds[si] for si in rest])
            for si, r in zip(rest, sub):
                results[si] = r
        return results

