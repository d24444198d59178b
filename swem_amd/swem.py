"""SWEM: the drop-in model module (reference methods/SWEM/swem.py:9-133).

Same constructor argument (``config_model`` with KEYDIM, VALDIM, NUM_BASES, NUM_EM_ITERS, EM_TAU, TOPL,
SINGLE_OBJ, BACKBONE), same ``forward(mode, *args)`` dispatcher and mode signatures, same child module
names and therefore the same ``state_dict`` keys, so the reference's evaluator / trainer loops and
checkpoints work against it.  All arithmetic is executed by libswem_hip.so through ``Engine``; the model
must live on a HIP device and fails loudly otherwise.

Feature maps returned by the modes are NCHW-shaped views of channels-last (NHWC) memory.
"""
import weakref

import torch
from torch import nn

from . import ops
from .engine import Engine
from .modules import SWEMCore, as_nchw, to_pixel_major
from .networks import Decoder, KeyEncoder, KeyProjection, ValueEncoder, ValueEncoderSO


_nchw = as_nchw
_KEY_STATE = weakref.WeakKeyDictionary()     # model -> side stream / last event of its eager encode_key calls


class SWEM(nn.Module):
    def __init__(self, config_model):
        super().__init__()
        keydim = config_model.KEYDIM
        valdim = config_model.VALDIM
        self.single_object = config_model.SINGLE_OBJ
        self.key_encoder = KeyEncoder(config_model.BACKBONE)
        nf = self.key_encoder.num_features
        self.value_encoder = ValueEncoderSO(nf[0]) if self.single_object else ValueEncoder(nf[0])
        self.key_proj = KeyProjection(nf[0], keydim=keydim)
        self.key_comp = nn.Conv2d(nf[0], valdim, kernel_size=(3, 3), padding=(1, 1))
        self.swem_core = SWEMCore(n_bases=config_model.NUM_BASES, valdim=valdim, n_iters=config_model.NUM_EM_ITERS,
                                  tau=config_model.EM_TAU, topl=config_model.TOPL)
        self.decoder = Decoder([valdim, nf[1], nf[2]], 256)
        self._eng = None
        self.swem_core._engine = self.engine
        # tuned plans and fused-split hints of THIS model (ops.PlanBook); current for the duration of every forward() call.
        # Models that run the same layers on the same shapes may share one (evaluator.SequencePool does): m.book = other.book
        # A fresh book holds no tuned plan: every layer then runs the heuristic tile in the f16x3 arithmetic (ops.MODEL_FALLBACK;
        # book.fallback = 0: the exact fp32 kernels); book.load_shipped() / ops.AUTOTUNE add tuned plans.
        self.book = ops.PlanBook(fallback=ops.MODEL_FALLBACK)

    # -- packed-weight cache: rebuilt after load_state_dict / .to() / .cuda()
    def engine(self):
        if self._eng is None:
            self._eng = Engine(self)
        return self._eng

    def invalidate(self):
        self._eng = None

    def load_state_dict(self, *a, **k):
        self._eng = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._eng = None
        return super()._apply(fn, *a, **k)

    # ------------------------------------------------------------------ swem.py:39-43
    def encode_key(self, frames):
        x = frames.float().contiguous()
        if ops.ASYNC_KEY_ENCODER and not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing():
            outs = self._encode_key_side(x)
        else:
            outs = self.engine().encode_key(x)
        qk16, qv16, s16, s8, s4 = outs
        return _nchw(qk16), _nchw(qv16), _nchw(s16), _nchw(s8), _nchw(s4)

    def _encode_key_side(self, x):
        """The key encoder on a SIDE stream (round 6, eager calls only).  It reads the frame and the weights -- never the memory
        (swem_evaluator.py:75 vs :77) -- so in a loop that calls the modes one frame at a time (the reference's evaluator on this
        model) frame i + 1's key encoder need not queue behind frame i's match -> segment -> encode_value -> memorize chain: the
        host enqueues it while that chain is still running, and on its own stream the two overlap (the look-ahead graphs do the
        same with a whole group of frames).  The caller's stream waits for the side stream before this returns, so every consumer
        sees finished tensors; the side stream waits for the caller's stream up to the END OF THE PREVIOUS encode_key call when
        this frame lives in the same storage as the previous one (a slice of a clip that was resident before the loop), and for
        everything the caller has queued otherwise (a frame produced just now: no overlap, same result).  `ops.ASYNC_KEY_ENCODER
        = False` / SWEM_ASYNC_KEY=0 turn it off (a caller that overwrites a resident clip buffer in place between calls must)."""
        main = torch.cuda.current_stream()
        state = _KEY_STATE.setdefault(self, {})       # (not in self.__dict__: streams and events do not deep-copy / pickle)
        st = state.get('stream')
        if st is None or st[0] != x.device:
            st = state['stream'] = (x.device, ops.new_stream())
        side = st[1]
        prev = state.get('prev')
        stor = x.untyped_storage().data_ptr()
        if prev is not None and prev[0] == stor and prev[2] == main.cuda_stream:
            side.wait_event(prev[1])
        else:
            side.wait_stream(main)
        x.record_stream(side)
        with torch.cuda.stream(side):
            outs = self.engine().encode_key(x)
        for t in outs:
            ops.record_stream_deep(t, main)        # (allocated on the side stream, consumed -- and released -- on the caller's)
        main.wait_stream(side)
        ev = torch.cuda.Event()
        ev.record(main)
        state['prev'] = (stor, ev, main.cuda_stream)
        return outs

    # ------------------------------------------------------------------ swem.py:45-62
    def encode_value(self, frame, masks, s16):
        B = frame.shape[0]
        N = masks.shape[1] - 1
        mv = self.engine().encode_value(frame.float().contiguous(), masks.float().contiguous(), to_pixel_major(s16))
        mv = _nchw(mv)                                    # (B*N, V, h, w)
        return mv.view(B, N, *mv.shape[1:])

    # ------------------------------------------------------------------ swem.py:64-86
    def init_mem(self, qk16, mv16, mask):
        self.swem_core.empty()
        return self.memorize(qk16, mv16, mask, mask.float())

    def memorize(self, qk16, mv16, masks_hard, masks_soft):
        b, _, h_16, w_16 = qk16.shape
        n = masks_hard.shape[1] - 1
        masks = ops.mask_prep(masks_hard.contiguous(), masks_soft.float().contiguous(), h_16, w_16)
        self.swem_core.memorize(qk16, mv16, masks.view(b, n, 2, h_16, w_16))

    # `memorize` in two calls (not in the reference): the part that does not need the value map, and the value update.
    # memorize_begin returns None where the one-call form has to be used (frame 0, new object ids, a batch of clips).
    def memorize_begin(self, qk16, masks_hard, masks_soft):
        b, _, h_16, w_16 = qk16.shape
        n = masks_hard.shape[1] - 1
        masks = ops.mask_prep(masks_hard.contiguous(), masks_soft.float().contiguous(), h_16, w_16)
        return self.swem_core.memorize_begin(qk16, masks.view(b, n, 2, h_16, w_16))

    def memorize_end(self, token, mv16):
        self.swem_core.memorize_end(token, mv16)

    # ------------------------------------------------------------------ swem.py:88-90
    def match(self, qk16, qv16):
        return self.swem_core.matching(qk16, qv16)

    # ------------------------------------------------------------------ swem.py:92-116
    def decode(self, n, context, s8, s4, valid_obj, out_size):
        B = s8.shape[0]
        logit4 = self.engine().decoder_logit(to_pixel_major(context), to_pixel_major(s8), to_pixel_major(s4))
        logits, pred_mask, _ = ops.decode_head(logit4, B, n, tuple(int(v) for v in out_size), valid=valid_obj)
        return logits, pred_mask

    def forward(self, mode, *args, **kwargs):
        with ops.use_book(self.book):
            return self._dispatch(mode, *args, **kwargs)

    def _dispatch(self, mode, *args, **kwargs):
        if mode == 'encode_key':
            return self.encode_key(*args, **kwargs)
        elif mode == 'encode_value':
            return self.encode_value(*args, **kwargs)
        elif mode == 'init':
            return self.init_mem(*args, **kwargs)
        elif mode == 'memorize':
            return self.memorize(*args, **kwargs)
        elif mode == 'match':
            return self.match(*args, **kwargs)
        elif mode == 'segment':
            return self.decode(*args, **kwargs)
        else:
            raise NotImplementedError
