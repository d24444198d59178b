"""Training step of SWEM on HIP kernels: the counterpart of ``SWEMTrainer.one_step``
(reference methods/SWEM/swem_trainer.py:59-108).

``SWEMTrainer(config, model)`` keeps the reference's ``one_step(frames, init_mask, valid_obj, label, cur_iter)``
signature and return value ``(losses, results)``.  The clip loop, the loss (``losses.VOSLoss``) and the optimizer
(``optim.FlatAdamW`` + ``MultiStepLR``) run through ``swem_amd.autograd`` / ``libswem_hip.so``; the model's parameters
live in one flat buffer and their gradients are accumulated in-kernel.

Data parallel: every rank steps its own clips; the flat gradient buffer is all-reduced over RCCL in two slices -- everything
but the key-encoder trunk as soon as the lanes have back-propagated down to the trunk (in flight during the trunk's
backward), the trunk's slice and the three loss scalars (one 3-float message) at the end -- and the 1/world factor is folded
into the loss gradient.  The reference scales the learning rate by the
number of GPUs only if asked (solver.py:31-34, ``num_gpu``), so does ``SWEMTrainer(num_gpu=...)``.

How the batch is run: clips are independent (frozen BatchNorm, per-clip memory), so every clip is its own forward /
per-clip loss with weight 1/B / backward.  Up to ``lanes`` clips are in flight at once, each on its own stream with its
own flat gradient buffer (the in-kernel accumulation is a plain read-modify-write); the lanes' buffers are summed into
the optimizer's.  That equals the reference's batched step (``total = mean_b``).  Mixed precision (config.AMP): the
convolutions' forward and data gradient take bf16 operands (one MFMA product, fp32 accumulate); everything else is fp32.
"""
import contextlib
import math
import os

import torch

from . import _lib, autograd as A, dist as sdist, losses as L, ops, optim
from .networks import BasicBlock, Bottleneck


def _bn(m):
    return (m.weight, m.bias, m.running_mean, m.running_var)


def _get(cfg, k):
    return cfg[k] if isinstance(cfg, dict) else getattr(cfg, k)


class TrainGraph:
    """The forward graphs of swem.py / networks.py on differentiable HIP stages for G clips of N objects at once -- the reference
    pushes the B clips of a GPU through the model as ONE tensor (swem_trainer.py:60-90: `frames[:, i]` is (B,3,H,W)); here G is
    the share of the batch one lane (stream) steps: G = B with one lane, G = 1 with a lane per clip (rounds 1-5).  Layout: a
    frame's tensors carry the clips on the batch axis (G, ...), per-object tensors the objects of clip 0, then clip 1, ...
    (G*N, ...), as the reference's `.flatten(0, 1)` of (B, N, ...) does (swem.py:57,98); the key encoder takes all T frames of
    the G clips in one pass, frame-major (T*G, ...).  EM and matching run clip by clip inside their stages (autograd._Memorize,
    _Match); maps the N objects of a clip share are laid out per object for the convolutions (autograd.expand_objects)."""

    def __init__(self, model):
        dev = next(model.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('swem_amd trains on a HIP device only (model is on %s); there is no CPU path' % dev)
        self.m = model
        self.single_obj = model.single_object
        self.k_mean, self.k_std = ops._f3(model.key_encoder.mean), ops._f3(model.key_encoder.std)
        self.v_mean, self.v_std = ops._f3(model.value_encoder.mean), ops._f3(model.value_encoder.std)

    # mod_resnet.py:58-113 / torchvision blocks with BatchNorm frozen (swem_trainer.py:37-39)
    @staticmethod
    def _block(blk, x):
        s = blk.stride
        if blk.downsample is not None:
            d = blk.downsample
            res = A.bn_act(A.conv2d([x], d[0].weight, d[0].bias, stride=s, pad=0), _bn(d[1]), relu=False)
        else:
            res = x
        if isinstance(blk, Bottleneck):
            y = A.bn_act(A.conv2d([x], blk.conv1.weight, blk.conv1.bias, pad=0), _bn(blk.bn1))
            y = A.bn_act(A.conv2d([y], blk.conv2.weight, blk.conv2.bias, stride=s, pad=1), _bn(blk.bn2))
            return A.bn_act(A.conv2d([y], blk.conv3.weight, blk.conv3.bias, pad=0), _bn(blk.bn3), res=res)
        assert isinstance(blk, BasicBlock)
        y = A.bn_act(A.conv2d([x], blk.conv1.weight, blk.conv1.bias, stride=s, pad=1), _bn(blk.bn1))
        return A.bn_act(A.conv2d([y], blk.conv2.weight, blk.conv2.bias, pad=1), _bn(blk.bn2), res=res)

    # networks.py:12-32
    @staticmethod
    def _res_block(rb, srcs, batch=None):
        r = A.conv2d(srcs, rb.conv1.weight, rb.conv1.bias, relu_in=True, batch=batch)
        if rb.downsample is None:
            res = srcs[0] if len(srcs) == 1 else A.concat2(srcs[0], srcs[1], batch or srcs[0].shape[0])
        else:
            res = A.conv2d(srcs, rb.downsample.weight, rb.downsample.bias, batch=batch)
        return A.conv2d([r], rb.conv2.weight, rb.conv2.bias, relu_in=True, residual=res)

    # swem.py:39-43 + networks.py:160-182, in two parts: the ResNet trunk (every parameter of `key_encoder`, the FIRST slice of
    # the flat parameter buffer) and the two 3x3 projections on its 1/16 map.  The step back-propagates through the trunk in a
    # phase of its own (SWEMTrainer: the rest of the gradient is all-reduced meanwhile).
    def key_trunk(self, frames):
        """frames: (T*G,3,H,W) frame-major, or a list of T tensors (G,3,H,W) (the frames of a lane's clips: slices of the step's
        (T,B,3,H,W) input buffer)."""
        ke = self.m.key_encoder
        if isinstance(frames, (list, tuple)):
            G, _, H, W = frames[0].shape
            x = torch.empty((len(frames) * G, H, W, 4), dtype=torch.float32, device=frames[0].device)
            for t, f in enumerate(frames):
                ops.prep_key_input(f, self.k_mean, self.k_std, out=x[t * G:(t + 1) * G])
        else:
            x = ops.prep_key_input(frames, self.k_mean, self.k_std)
        x = A.maxpool(A.bn_act(A.conv2d([x], ke.conv1.weight, None, stride=2, pad=3, cin_pad=4), _bn(ke.bn1)))
        feats = []
        for st in (ke.res2, ke.layer2, ke.layer3):
            for blk in st:
                x = self._block(blk, x)
            feats.append(x)
        s4, s8, s16 = feats
        return s16, s8, s4

    def key_project(self, s16):
        m = self.m
        qk16 = A.conv2d([s16], m.key_proj.key_proj.weight, m.key_proj.key_proj.bias)
        qv16 = A.conv2d([s16], m.key_comp.weight, m.key_comp.bias)
        return qk16, qv16

    def encode_key(self, frame):
        s16, s8, s4 = self.key_trunk(frame)
        qk16, qv16 = self.key_project(s16)
        return qk16, qv16, s16, s8, s4

    # swem.py:45-62 + networks.py:113-129, 43-50
    def encode_value(self, frame, masks, s16):
        """frame (G,3,H,W), masks (G,N+1,H,W), s16 (G,h,w,C) -> (G*N,h,w,V)."""
        ve = self.m.value_encoder
        G, N = masks.shape[0], masks.shape[1] - 1
        x = A.prep_value_input(frame, masks, self.v_mean, self.v_std, self.single_obj)
        x = A.maxpool(A.bn_act(A.conv2d([x], ve.conv1.weight, ve.conv1.bias, stride=2, pad=3, cin_pad=8), _bn(ve.bn1)))
        for st in (ve.layer1, ve.layer2, ve.layer3):
            for blk in st:
                x = self._block(blk, x)
        x = self._res_block(ve.fuser.block1, [x, A.expand_objects(s16, N)], batch=G * N)
        att = ve.fuser.attention
        x = A.cbam_residual(x, att.ChannelGate.mlp[1].weight, att.ChannelGate.mlp[1].bias, att.ChannelGate.mlp[3].weight,
                            att.ChannelGate.mlp[3].bias, att.SpatialGate.spatial.conv.weight,
                            att.SpatialGate.spatial.conv.bias)
        return self._res_block(ve.fuser.block2, [x])                      # (G*N, h, w, V)

    # modules.py:278-293
    def match(self, qk16, qv16, first, update):
        """qk16 (G,h,w,C), qv16 (G,h,w,V), banks (G*N, ...) -> context (G*N,h,w,V), N."""
        core = self.m.swem_core
        G, h, w, Cc = qk16.shape
        P = h * w
        mem, S = A.match(qk16.view(G, P, Cc), first['nu'], None if update is None else update['nu'], first['kappa'],
                         None if update is None else update['kappa'], core.topl, core.tau)
        GN = mem.shape[0]
        N = GN // G
        mem = mem[:, :P].view(GN, h, w, -1)
        S = S.view(GN, h, w, -1)
        qv = A.expand_objects(qv16, N)
        fl = core.fusion_layer
        f = A.conv2d([mem, qv, S], fl.layer_f.weight, fl.layer_f.bias, batch=GN)
        a = A.conv2d([mem, qv, S], fl.layer_a.weight, fl.layer_a.bias, batch=GN)
        return A.glu(f, a), N

    # swem.py:92-116 + networks.py:199-216
    def segment(self, n, context, s8, s4, valid_obj, out_size):
        """context (G*n,h,w,V), s8 / s4 (G, ...), valid_obj (G,n+1) or None -> logits, prob (G,n+1,Ho,Wo)."""
        dec = self.m.decoder
        G = s8.shape[0]
        x = self._res_block(dec.compress, [context])
        sk = A.conv2d([s8], dec.up_16_8.skip_conv.weight, dec.up_16_8.skip_conv.bias)
        x = self._res_block(dec.up_16_8.out_conv, [A.upsample_add(A.expand_objects(sk, n), x, batch=G * n)])
        sk = A.conv2d([s4], dec.up_8_4.skip_conv.weight, dec.up_8_4.skip_conv.bias)
        x = self._res_block(dec.up_8_4.out_conv, [A.upsample_add(A.expand_objects(sk, n), x, batch=G * n)])
        logit4 = A.pred_head(x, dec.pred.weight, dec.pred.bias)
        return A.decode_head(logit4, valid_obj, G, n, out_size)

    # swem.py:64-86 + modules.py:129-168, 183-193
    def memorize(self, qk16, mv16, masks_hard, masks_soft, prior):
        """qk16 (G,h,w,C), mv16 (G*N,h,w,V), masks (G,N+1,H,W), prior bases (G*N, ...)."""
        core = self.m.swem_core
        G, h, w, Cc = qk16.shape
        GN = mv16.shape[0]
        masks = ops.mask_prep(masks_hard.contiguous(), masks_soft.detach().float().contiguous(), h, w)   # (G*N,2,P)
        kappa, nu, zita = A.memorize(mv16.view(GN, h * w, -1), prior['nu'], qk16.detach().view(G, h * w, Cc), masks,
                                     prior['kappa'], prior['zita'], core.n_iters, core.tau)
        return {'kappa': kappa, 'nu': nu, 'zita': zita}


class one_cpu_thread:
    """Context: torch CPU ops inside run on ONE thread.  A torch CPU op of a few hundred KB wakes the whole intra-op pool -- one
    OpenMP thread per logical CPU, 128-256 on the hosts of this pool, busy-waiting between ops -- and inside a container with a
    CFS CPU quota (cpu.max = 16 CPUs per 100 ms here) that burns the period's budget in milliseconds: the kernel then stalls
    EVERY thread of the process, the one feeding the GPU included, until the next period.  That was the "bimodal" training rate
    of rounds 1-3 (70-76 vs 99-111 clips/s, step times of 38 / 62 / 100 / 200 ms on a 100 ms grid): with the step's host-side
    draw on one thread every step takes 37.2-37.9 ms (tools/train_bench.py --per-step, profiles/r04_train_step_times.txt)."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(1)

    def __exit__(self, *a):
        torch.set_num_threads(self.n)


def random_init_host(B, N, Cc, Lb):
    """kappa of modules.py:170-178 drawn from the global torch CPU generator (the parity mode of the tests, SWEMCore.init_on_host:
    the reference trainer's fixtures were recorded on the CPU): N(0, sqrt(2/L)) l2-normalised over C, one tensor for the whole
    batch.  (nu = 0 and zita = 1e-6 are constants.)"""
    with one_cpu_thread():
        kappa = torch.zeros(B, N, 2, Cc, Lb)
        kappa.normal_(0, math.sqrt(2.0 / Lb))
        return kappa / (torch.linalg.norm(kappa, dim=-2, keepdim=True) + 1e-6)


# Measured on one box, 4 clips of 3 x 384x384, 2 objects, R50, graph replay (profiles/r06_train_lanes_ab.txt; fp32-level / AMP clips/s):
#   4 lanes x 1 clip 93.1 / 120.5 -- 2 lanes x 2 clips 89.4 / 118.0 -- 1 lane x 4 clips 81.2 / 106.9 (84.1 / 109.5 with the weight
#   gradients on a side stream).  Batching the clips cuts the kernel time per clip from 21.8 to 12.9 ms and the launches from 1,434
#   to 502 (profiles/r06_train_launches_f16x3_lanes1.csv), but one stream of kernels that each fill a fraction of the chip (the
#   1/16-scale layers: 36-72 tiles on 256 CUs) loses the 1.95x overlap four concurrent lanes get: the lanes keep the default.
DEFAULT_LANES = 4


class SWEMTrainer:
    """swem_trainer.py:19-108 without the dataset / logging plumbing: model, criterion, optimizer, scheduler, one_step."""

    def __init__(self, config, model, num_gpu=None, use_graph=True, lanes=None, overlap_allreduce=True, reduce_in_graph=False,
                 wgrad_stream=None):
        self.config = config
        self.model = model
        # config.AMP (configs/config.py:89, basic_trainer.py:83-86,222): the reference runs the forward under fp16
        # autocast with a GradScaler.  Here AMP = the convolutions (forward and data gradient) round their operands to
        # bf16 and take ONE MFMA product with fp32 accumulation (conv math mode 2); activations, EM, matching, the loss,
        # the weight gradient and the optimizer stay fp32.  bf16 keeps fp32's exponent range, so no loss scaling.
        self.amp = bool(_get(config, 'AMP'))
        # fp32-level steps (no AMP) may run their convolutions -- forward, data and weight gradient -- in the f16x3 arithmetic of
        # the inference path (three fp16 MFMA products, 22-23 operand bits) next to fp32 MFMA and bf16x6 (six products): the
        # tuner picks per layer.  Gradient maps are scaled by a power of two chosen on the device (swem_split_f16x2_scaled_f32);
        # activations beyond the fp16 range raise SwemRangeError at the fault check (then: SWEM_TRAIN_F16X3=0 / trainer.f16x3 = False
        # before the first step).
        self.f16x3 = os.environ.get('SWEM_TRAIN_F16X3', '1') != '0'
        self.math_modes = None       # (tests: an explicit set of conv math modes for the step, e.g. (7,) = f16x3 wherever it applies)
        # data parallel: True = the non-trunk gradient slice is all-reduced while the lanes back-propagate through the
        # key-encoder trunk (collective kernels next to the lanes' graphs); False = ONE all-reduce of the whole gradient after
        # the backward pass, nothing of RCCL in flight beside the lanes (the conservative form; same result bit for bit)
        self.overlap_allreduce = bool(overlap_allreduce)
        # reduce_in_graph: the all-reduce of the non-trunk slice is CAPTURED into the step's HIP graph (the node behind the lanes'
        # gradient sum) instead of being issued between two graph replays -- one launch less on the host's critical path per step;
        # needs a collective library that records into a stream capture (RCCL does: tests/_rccl_single_rank_probe.py)
        self.reduce_in_graph = bool(reduce_in_graph)
        dev = next(model.parameters()).device
        model.train()
        for mod in model.modules():                    # BasicTrainer.set_bn_eval (swem_trainer.py:37-39)
            if mod.__class__.__name__.find('BatchNorm') != -1:
                mod.eval()
        # what the step learns about its launches (tuned conv plans, fused-split hints) is the trainer's own: the model's
        # inference book (validation between steps) and other trainers in the process never see it
        self.book = ops.PlanBook()
        A.reset(self.book)
        self.optimizer = optim.make_optimizer(_get(config, 'SOLVER'), model, num_gpu)
        # DistributedDataParallel's constructor broadcasts rank 0's parameters AND buffers (swem_trainer.py:41-43;
        # broadcast_buffers=False only stops the per-iteration re-broadcast): without it rank-local initialisation
        # (FROM_SCRATCH, the orthogonal fifth stem channel of checkpoint.adapt_state_dict) would train W different models
        sdist.broadcast_model_(self.optimizer.param, model)
        # the key-encoder trunk's parameters are the first slice of the flat buffers (model.parameters() order)
        ke = {id(q) for q in model.key_encoder.parameters()}
        idx = [i for i, q in enumerate(self.optimizer.params) if id(q) in ke]
        assert idx == list(range(len(idx))), 'key_encoder must come first in model.parameters()'
        nxt = self.optimizer.params[len(idx)] if len(idx) < len(self.optimizer.params) else None
        self.trunk_end = (nxt.data_ptr() - self.optimizer.param.data_ptr()) // 4 if nxt is not None else self.optimizer.param.numel()
        self._works = []
        self._trunks = []
        self.lr_scheduler = optim.make_lr_scheduler(_get(config, 'SOLVER'), self.optimizer)
        self.criterion = L.get_criterion(_get(config, 'LOSS'), None, 1, 1, dev)
        self.graph = TrainGraph(model)
        self.device = dev
        self.use_graph = use_graph
        # The B clips of a step are cut into `lanes` contiguous shares; a share is stepped as ONE batch (TrainGraph: G clips
        # through every convolution together, as the reference runs its batch, swem_trainer.py:60-90) on a stream of its own, with
        # its own flat gradient buffer.  lanes = B: a clip per stream (rounds 1-5); lanes = 1: the whole batch in one pass.
        # Default (SWEM_TRAIN_LANES overrides): DEFAULT_LANES, the faster on the reference's training shapes -- measured, profiles/
        # r06_train_lanes_ab.txt.
        self.lanes = max(1, int(lanes if lanes is not None else os.environ.get('SWEM_TRAIN_LANES', DEFAULT_LANES)))
        # every lane's weight gradients on a second stream beside its data-gradient chain (autograd.use_lane(side=...)): the
        # backward pass's critical path is dY -> dX -> the previous layer; a third of its kernel time (dW) depends on nothing
        # downstream.  Measured (profiles/r06_train_lanes_ab.txt): + 3.5-4 % with ONE lane (79.5 -> 82.4 / 103.4 -> 107.9 clips/s).
        # (beside OTHER lanes it loses: the forked branches of several lanes' graphs end up on shared hardware queues -- 89 -> 51
        # clips/s with four lanes: the default is on for a single lane only)
        self.wgrad_stream = bool(int(os.environ.get('SWEM_TRAIN_WGRAD_STREAM', 1 if self.lanes == 1 else 0))) if wgrad_stream is None \
            else bool(wgrad_stream)
        self._lane_state = None
        self._foreign_fault = 0
        ops.fault_word(dev)
        ops.register_fault_owner(self)

    def clip_forward(self, frames, init_mask, valid_obj, prior0):
        """swem_trainer.py:63-90 for G clips: frames = T tensors (G,3,H,W) (or one (T*G,3,H,W), frame-major), init_mask
        (G,N+1,H,W), valid_obj (G,N+1) or None, prior0 = random bases (G*N, ...).  Returns T-1 logits (G,N+1,H,W) and index maps."""
        g = self.graph
        G = init_mask.shape[0]
        t = len(frames) if isinstance(frames, (list, tuple)) else frames.shape[0] // G
        frame = (lambda i: frames[i]) if isinstance(frames, (list, tuple)) else (lambda i: frames[i * G:(i + 1) * G])
        out_size = tuple(init_mask.shape[-2:])
        # the key encoder does not depend on the memory: all t frames of the G clips go through it in one pass (a third of its
        # launches, t times larger kernels); frozen BatchNorm keeps every frame's result what a per-frame call gives
        trunk = g.key_trunk(frames)                                          # (s16, s8, s4), all t frames, frame-major
        # the trunk's backward is a phase of its own (`trunk_backward`): everything downstream differentiates towards
        # detached copies, whose .grad the second phase feeds into the trunk
        cut = [o.detach().requires_grad_(True) for o in trunk]
        qk_all, qv_all = g.key_project(cut[0])
        enc = [A.unbatch(e, t) for e in (qk_all, qv_all, cut[0], cut[1], cut[2])]   # [qk16, qv16, s16, s8, s4][frame]: (G, ...)
        self._trunks.append((trunk, cut))
        mk16, s16 = enc[0][0], enc[2][0]
        mv16 = g.encode_value(frame(0), init_mask.float(), s16)
        first = g.memorize(mk16, mv16, init_mask, init_mask.float(), prior0)
        update = None
        logits_list, results = [], []
        for i in range(1, t):
            qk16, qv16, s16, s8, s4 = (e[i] for e in enc)
            context, n = g.match(qk16, qv16, first, update)
            logits, pred_mask = g.segment(n, context, s8, s4, valid_obj, out_size)
            logits_list.append(logits)
            pred, hard = ops.argmax_onehot(pred_mask.detach(), want_onehot=i < t - 1)
            results.append(pred)
            if i < t - 1:
                mv16 = g.encode_value(frame(i), pred_mask, s16)
                update = g.memorize(qk16, mv16, hard, pred_mask, first if update is None else update)
        return logits_list, results

    # ------------------------------------------------------------------ the step
    def _static(self, frames, init_mask, valid_obj, label):
        """Device buffers with fixed addresses for the step's inputs (a captured HIP graph replays on them)."""
        key = (tuple(frames.shape), tuple(init_mask.shape), valid_obj is not None)
        if getattr(self, '_static_key', None) != key:
            core = self.model.swem_core
            B, N = frames.shape[0], init_mask.shape[1] - 1
            Ck = self.model.key_proj.key_proj.weight.shape[0]
            dev = self.device
            T = frames.shape[1]
            self.buf = {
                # frame-major (T,B,3,H,W): frame t of a contiguous share of the clips is one dense (G,3,H,W) block
                'frames': torch.empty((T, B) + tuple(frames.shape[2:]), dtype=torch.float32, device=dev),
                'init_mask': torch.empty(init_mask.shape, dtype=torch.float32, device=dev),
                'label': torch.empty(label.shape, dtype=torch.int64, device=dev),
                'valid': None if valid_obj is None else torch.empty(valid_obj.shape, dtype=torch.float32, device=dev),
                'kappa0': torch.empty((B, N, 2, Ck, core.n_bases), dtype=torch.float32, device=dev),
                'nu0': torch.zeros((B * N, 2, core.valdim, core.n_bases), dtype=torch.float32, device=dev),
                'zita0': torch.full((B * N, 2, core.n_bases), 1e-6, dtype=torch.float32, device=dev),
                'gout': torch.zeros((max(1, min(self.lanes, B)), 3), dtype=torch.float32, device=dev),   # per lane: its share / world
                'k': torch.zeros(1, dtype=torch.int64, device=dev),
                # (total, main, aux) of the batch + the step's two fault flags (range, other: swem_fault_flags_f32) -- ONE message
                'sums': torch.zeros(5, dtype=torch.float32, device=dev),
            }
            self._static_key, self._graph, self._eager_steps = key, None, 0
        return self.buf

    def _lanes(self, B):
        """Streams, per-lane flat gradient buffers and loss accumulators for the clips in flight.  A clip's ~3000 launches are
        short and dependent, so a second clip's kernels fill the gaps; the in-kernel gradient accumulation is a plain
        read-modify-write, hence one gradient buffer per lane, summed into the optimizer's buffer at the end."""
        n = min(self.lanes, B)
        if self._lane_state is None or len(self._lane_state['streams']) != n:
            from . import evaluator
            opt = self.optimizer
            flat = torch.zeros((n, opt.grad.numel()), dtype=torch.float32, device=self.device)
            views = []
            for l in range(n):
                d = {}
                for prm in opt.params:
                    off = prm.grad.data_ptr() - opt.grad.data_ptr()
                    d[id(prm)] = flat[l, off // 4: off // 4 + prm.numel()].view(prm.shape)
                views.append(d)
            # (the lanes' streams and their weight-gradient streams: 2n streams probed to overlap one another)
            sts = evaluator.overlapping_streams(2 * n if self.wgrad_stream else n) if (n > 1 or self.wgrad_stream) else [None]
            self._lane_state = {'streams': (sts[:n] if n > 1 else [None]), 'flat': flat,
                                'side': (sts[n:2 * n] if n > 1 else sts[:1]) if self.wgrad_stream else [None] * n,
                                'views': views, 'sums': torch.zeros((n, 3), dtype=torch.float32, device=self.device)}
        # lane l steps the clips [chunks[l][0], chunks[l][1]) as one batch
        self._lane_state['chunks'] = [((l * B) // n, ((l + 1) * B) // n) for l in range(n)]
        return self._lane_state

    # The step's clip work in three parts, so that it can run eagerly or as HIP graphs: `_pre` (main stream: re-pack the
    # filters once for all lanes, zero the lanes' gradient buffers), `_lane(l)` (lane l's clips: forward / loss / backward
    # on the lane's stream and gradient buffer), `_post` (main stream: lanes' gradients and losses summed).
    def _pre(self):
        ls = self._lanes(self.buf['init_mask'].shape[0])
        A.new_step()                                 # (the packs themselves: `_packs`, dealt out to the lanes' streams)
        ls['flat'].zero_()
        ls['sums'].zero_()
        self._results = [None] * len(ls['streams'])
        self._lane_trunks = [[] for _ in ls['streams']]

    def _packs(self, l):
        """Lane l's share of the step's filter re-pack (autograd.prebuild_packs), on the lane's stream.  Every lane reads every
        pack: the caller joins all lanes' streams between this and `_lane`."""
        A.prebuild_packs(l, len(self._lane_state['streams']))

    def _lane(self, l, cur_iter):
        """Phase A of lane l: its clips' forward (one batch), loss and the backward of everything BUT the key-encoder trunk."""
        bf = self.buf
        B = bf['init_mask'].shape[0]
        ls = self._lanes(B)
        b0, b1 = ls['chunks'][l]
        G, N = b1 - b0, bf['init_mask'].shape[1] - 1
        T = bf['frames'].shape[0]
        A.use_lane(l, ls['views'][l], ls['side'][l])
        self._trunks = self._lane_trunks[l]
        vo = None if bf['valid'] is None else bf['valid'][b0:b1]
        prior = {'kappa': bf['kappa0'][b0:b1].flatten(0, 1), 'nu': bf['nu0'][b0 * N:b1 * N], 'zita': bf['zita0'][b0 * N:b1 * N]}
        # (the whole batch in one lane: the (T,B,...) buffer IS the key encoder's frame-major batch)
        frames = bf['frames'].flatten(0, 1) if G == B else [bf['frames'][t, b0:b1] for t in range(T)]
        logits_list, res = self.clip_forward(frames, bf['init_mask'][b0:b1], vo, prior)
        out = self.criterion.clip_loss(logits_list, bf['label'][b0:b1, 1:], cur_iter, vo, k_dev=bf['k'])
        vec = out['_vec']                                      # (total, main, aux): means over this lane's G clips
        vec.backward(bf['gout'][l])
        ls['sums'][l].copy_(ops.lincomb(vec.detach(), float(G) / B))
        self._results[l] = torch.stack(res, dim=1)             # (G, T-1, H, W)
        A.join_side()
        A.use_lane(0, None)
        return out['p']

    def _lane_trunk(self, l):
        """Phase B of lane l: the key-encoder trunk's backward for the lane's clips (its parameters are the first slice of the
        flat buffer; meanwhile the rest of the gradient is summed over the lanes and all-reduced, `_post_rest`)."""
        ls = self._lane_state
        A.use_lane(l, ls['views'][l], ls['side'][l])
        for trunk, cut in self._lane_trunks[l]:
            pairs = [(o, c.grad) for o, c in zip(trunk, cut) if c.grad is not None]
            torch.autograd.backward([o for o, _ in pairs], [g_ for _, g_ in pairs])
        self._lane_trunks[l] = []
        A.join_side()
        A.use_lane(0, None)

    def _post_rest(self):
        """Lanes' gradients of everything but the trunk -> the optimizer's buffer (slice [trunk_end, end))."""
        self._sum_lanes(self.trunk_end, self.optimizer.grad.numel())

    def _sum_lanes(self, lo, hi):
        """optimizer.grad[lo:hi] = sum over the lanes of their buffers' [lo:hi) (lane order: the same sum every step)."""
        ls = self._lane_state
        out = self.optimizer.grad[lo:hi].view(1, -1)
        for l in range(ls['flat'].shape[0]):
            A.sum_batch(ls['flat'][l:l + 1, lo:hi], out=out, accumulate=l > 0)

    def _post(self):
        bf = self.buf
        ls = self._lanes(bf['init_mask'].shape[0])
        self._sum_lanes(0, self.trunk_end)                                     # the trunk's slice
        tot = ls['sums'][0]
        for l in range(1, len(ls['streams'])):
            tot = ops.lincomb(tot, 1.0, ls['sums'][l], 1.0)
        bf['sums'][:3].copy_(tot)
        return torch.cat(self._results, dim=0)

    def _clips(self, cur_iter):
        """zero_grad + forward / loss / backward of every clip on the static buffers; returns (results, p).
        Data parallel: the all-reduce of the non-trunk gradient (6/7 of the parameters) is started as soon as every lane has
        finished phase A and runs while the lanes back-propagate through the key-encoder trunk."""
        ls = self._lanes(self.buf['init_mask'].shape[0])
        main = torch.cuda.current_stream()
        self._pre()
        p = 1.0
        for st in ls['streams']:
            if st is not None:
                st.wait_stream(main)                                   # fork: the re-pack, a share per lane
        for l, st in enumerate(ls['streams']):
            with torch.cuda.stream(st if st is not None else main):
                self._packs(l)
        for st in ls['streams']:
            if st is not None:
                main.wait_stream(st)                                   # every pack is there ...
        for st in ls['streams']:
            if st is not None:
                st.wait_stream(main)                                   # ... before any lane reads one
        for l, st in enumerate(ls['streams']):
            with torch.cuda.stream(st if st is not None else main):
                p = self._lane(l, cur_iter)
        for st in ls['streams']:
            if st is not None:
                main.wait_stream(st)                                   # phase A of every lane
        self._post_rest()
        self._reduce_rest()
        for l, st in enumerate(ls['streams']):
            with torch.cuda.stream(st if st is not None else main):
                self._lane_trunk(l)
        for st in ls['streams']:
            if st is not None:
                main.wait_stream(st)                                   # join
        return self._post(), p

    # ------------------------------------------------------------------ data-parallel reduction (RCCL over xGMI)
    def _reduce_rest(self, captured=False):
        """Start the all-reduce of the gradient slice that phase A completed; waited for in `_reduce_finish`.
        captured: called from inside the stream capture of the step (reduce_in_graph) -- the collective becomes a graph node, in
        stream order, nothing to wait for; the replayed step then skips the eager call."""
        if captured:
            if self.overlap_allreduce and sdist.active():
                sdist.allreduce_sum_(self.optimizer.grad[self.trunk_end:], sync=True)
            return
        if self.reduce_in_graph and self._graph is not None:
            self._works = []
            return
        self._works = sdist.allreduce_sum_async(self.optimizer.grad[self.trunk_end:]) if self.overlap_allreduce else []

    def _reduce_finish(self):
        """All-reduce the trunk's slice and the three loss scalars (ONE 3-float all-reduce instead of the reference's three,
        basic_trainer.py:105-110,240-243; no host synchronisation), then wait for everything in flight."""
        works = self._works + sdist.allreduce_sum_async(self.optimizer.grad[:self.trunk_end] if self.overlap_allreduce
                                                         else self.optimizer.grad)
        works += sdist.allreduce_sum_async(self.buf['sums'], mean=True)
        for w in works:
            w.wait()
        self._works = []

    def one_step(self, frames, init_mask, valid_obj, label, cur_iter, _redo=False):
        """swem_trainer.py:59-108.  With ``use_graph`` (default) the clips' forward/backward is captured into a HIP graph
        after two eager steps (which also tune the conv plans) and replayed afterwards: ~3000 launches per clip otherwise
        leave the GPU waiting for the host.  Everything that changes between steps enters through device buffers (inputs,
        random bases, the bootstrap k, the loss weight); the optimizer update stays outside the graph (lr, step count)."""
        bf = self._static(frames, init_mask, valid_obj, label)
        B, N = frames.shape[0], init_mask.shape[1] - 1
        core = self.model.swem_core
        bf['frames'].copy_(frames.transpose(0, 1))             # (B,T,...) -> the frame-major buffer
        bf['init_mask'].copy_(init_mask)
        bf['label'].copy_(label)
        if valid_obj is not None:
            bf['valid'].copy_(valid_obj)
        if not core.init_on_host:
            # the random bases as the reference draws them (modules.py:170-178: `normal_` on the DEVICE tensor, the device's
            # generator), straight into the static buffer the captured step reads: no host work, no copy
            k0 = bf['kappa0']
            k0.normal_(0, math.sqrt(2.0 / k0.shape[-1]))
            k0.div_(torch.linalg.norm(k0, dim=-2, keepdim=True).add_(1e-6))
        else:
            # parity mode: drawn on the host (the global torch CPU generator, as the fixtures of the reference step were) and
            # sent through a ring of three pinned buffers -- a pageable copy would block the host until the previous step has
            # drained, i.e. serialise host and device every step
            ring = self.__dict__.setdefault('_kappa_ring', {})
            if ring.get('shape') != tuple(bf['kappa0'].shape):
                ring.update(shape=tuple(bf['kappa0'].shape), i=0, ev=[None] * 3,
                            pin=[torch.empty(bf['kappa0'].shape, dtype=torch.float32).pin_memory() for _ in range(3)])
            i = ring['i'] = (ring['i'] + 1) % 3
            if ring['ev'][i] is not None:
                ring['ev'][i].synchronize()                 # the copy issued three steps ago has long finished
            with one_cpu_thread():
                ring['pin'][i].copy_(random_init_host(B, N, bf['kappa0'].shape[3], core.n_bases))
            bf['kappa0'].copy_(ring['pin'][i], non_blocking=True)
            ring['ev'][i] = torch.cuda.Event()
            ring['ev'][i].record()
        # mean over the clips of this rank and over the ranks (DistributedDataParallel averages, swem_trainer.py:41-43)
        world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        if getattr(self, '_gout_for', None) != (B, world, id(bf['gout'])):
            # a lane's loss vector is the mean over ITS clips: weight (its clips) / (clips of all lanes and ranks)
            chunks = self._lanes(B)['chunks']
            bf['gout'].copy_(torch.tensor([[float(b1 - b0) / (B * world), 0.0, 0.0] for b0, b1 in chunks]))
            self._gout_for = (B, world, id(bf['gout']))
        H, W = init_mask.shape[-2:]
        p, k = self.criterion.top_k(cur_iter, H * W)
        bf['k'].fill_(k)
        if self.use_graph and self._eager_steps >= 2 and self._graph is None and not ops.AUTOTUNE_PENDING():
            self._capture(cur_iter)
        if self._graph is not None:
            self._replay()
            results, p = self._graph_out, (1.0 if p is None else p)
        else:
            with self._math():
                results, p = self._clips(cur_iter)
            self._eager_steps += 1
        # the step's fault flags, in stream order behind every launch of the step, into the message the loss scalars travel in:
        # after the all-reduce every rank holds "some rank faulted" and gates its optimizer launch on it ON THE DEVICE (ADVICE
        # r05: like GradScaler's found_inf -- no update is ever made from gradients a faulted launch produced, on any rank, and
        # no rank steps alone into the next all-reduce)
        _lib.call('swem_fault_flags_f32', ops._stream(), ops._fault_ptr(self.device), bf['sums'][3:].data_ptr())
        if self._foreign_fault:
            # bits another owner drained from the word (on_foreign_fault) are no longer on the device: they enter the flags here,
            # every step until the host has looked -- through the all-reduce, so that every rank decides alike
            ff = self._foreign_fault
            bf['sums'][3:].add_(torch.tensor([float(bool(ff & ops.FAULT_RANGE)), float(bool(ff & ~ops.FAULT_RANGE))], device=self.device))
        self._reduce_finish()                                          # RCCL over xGMI; no-op for one process
        self.optimizer.step(gate=bf['sums'][3:])
        self.lr_scheduler.step()
        # the parameters changed in place: inference through this model (validation, encode_key / segment modes) must not run
        # on the conv packs of the engine built before the step
        self.model.invalidate()
        sums = bf['sums']
        losses = {'total_loss': sums[0], 'main_loss': sums[1], 'aux_loss': sums[2], 'p': p}
        # Looking at the flags synchronises the device, so the HOST looks every `fault_check_every` steps (default 20: the
        # reference's trainer reads its losses -- a synchronisation -- every LOG_PERIOD steps anyway, basic_trainer.py:105-131), and
        # always on the eager first steps.  Nothing is lost in between: the fault word is sticky, so from the faulting step on
        # every optimizer launch finds the gate closed.
        self._steps_seen = getattr(self, '_steps_seen', 0) + 1
        every = getattr(self, 'fault_check_every', 20)
        if every and (self._graph is None or self._steps_seen % every == 0 or (self._foreign_fault and not sdist.active())):
            if self._look_at_faults() and not _redo:
                return self.one_step(frames, init_mask, valid_obj, label, cur_iter, _redo=True)
        return losses, results

    # ------------------------------------------------------------------ asynchronous faults of the step
    def on_foreign_fault(self, bits):
        """ops.drain_faults: another owner (a validation sequence about to start) found these bits in the device's fault word --
        launches of THIS trainer's steps left them.  Remembered; the next step deals with them."""
        self._foreign_fault |= int(bits)

    def _look_at_faults(self):
        """Read the all-reduced fault flags of the last step (synchronises).  Clean: False.  Otherwise the optimizer has not
        moved since the faulting step (device-side gate, every rank alike); the host's step / scheduler counts are wound back to
        what was applied, the local fault word is cleared, and
          * a fault other than SWEM_FAULT_RANGE (a K-split / stream-K wait that expired) raises SwemHipError on EVERY rank;
          * SWEM_FAULT_RANGE -- an activation or a scaled gradient left the fp16 range of the f16x3 arithmetic -- moves the
            trainer to the reference's own range (fp32 MFMA / bf16x6: modes (0, 1)), warns, and returns True: the caller redoes
            the step at hand (the batches of the skipped steps in between are lost, and counted in the warning);
            a trainer that is on the full-range modes already raises SwemRangeError (cannot happen: no fp16 pair is produced)."""
        import warnings
        fl = self.buf['sums'][3:5].tolist()                 # (all-reduced: the same on every rank, and so is what follows)
        rng, other = fl[0] != 0.0, fl[1] != 0.0
        if not (rng or other):
            return False
        foreign, self._foreign_fault = self._foreign_fault, 0
        bits = ops._collect_faults()                       # clear the local word (and, on a WAIT fault, the tile counters)
        skipped = self.optimizer.reconcile()
        self.lr_scheduler.rewind(skipped)
        if other:
            raise _lib.SwemHipError('asynchronous fault in the training step (flags range=%s other=%s, local word %#x): %s -- the '
                                    'optimizer has not been stepped since the fault (%d steps skipped on every rank)'
                                    % (rng, other, bits | foreign, ops._fault_text((bits | foreign) & ~ops.FAULT_RANGE) or
                                       'raised on another rank', skipped))
        modes = self.math_modes or ((2,) if self.amp else ((0, 1, 7) if self.f16x3 else (0, 1)))
        if 7 not in modes:
            raise ops.SwemRangeError('SWEM_FAULT_RANGE in a training step that runs no f16x3 launch (modes %s)' % (modes,))
        self.f16x3, self.math_modes = False, None
        self._graph, self._eager_steps = None, 0
        self.book = ops.PlanBook()
        A.reset(self.book)
        warnings.warn('swem_amd: a training step left the fp16 range of the f16x3 arithmetic (SWEM_FAULT_RANGE on some rank): no '
                      'optimizer update was applied from that step on (%d steps skipped); the trainer now runs the fp32-range '
                      'arithmetic (fp32 MFMA / bf16x6, about 0.8x the step rate) and redoes the step at hand' % skipped,
                      RuntimeWarning)
        return True

    def _math(self):
        """Conv math modes of the step: config.AMP = plain bf16 operands; otherwise the fp32-level modes only (fp32 MFMA,
        bf16x6 and -- round 5, `self.f16x3` -- the fp16-pair f16x3 with gradient maps scaled on the device, include/
        swem_hip_train.h) -- the 16-bit-operand bf16x3 mode the inference tuner may pick is kept out of the gradient path."""
        import contextlib
        st = contextlib.ExitStack()
        st.enter_context(ops.use_book(self.book))
        modes = self.math_modes or ((2,) if self.amp else ((0, 1, 7) if self.f16x3 else (0, 1)))
        st.enter_context(ops.conv_math(modes))
        # conv epilogues do not write operand planes here and the tuner keeps to the non-persistent kernel forms (ops.py,
        # TUNE_ROUND3_FORMS: measured on the four-lane step); the frozen-BN stages' own planes (autograd._planes_for) stay
        # (re-measured in round 4 with the step's timing stable to +-0.3 %: FUSE_SPLIT on or off is the same 112 / 69 clips/s --
        # round 3's "71 against 108" was the CPU-quota throttling of that round's processes, not the planes)
        st.enter_context(ops.flags(FUSE_SPLIT=False, TUNE_ROUND3_FORMS=False))
        return st

    def _capture(self, cur_iter):
        """One HIP graph per part: `_pre` and `_post` on the main stream, one graph per lane on the lane's own (probed)
        stream.  A single graph with forked branches leaves the placement of the branches to the runtime, and two lanes on
        one hardware queue serialise (measured: 40 to 56 clips/s from run to run); separate graphs replay on the streams
        that `evaluator.overlapping_streams` found to overlap."""
        torch.cuda.synchronize()
        ls = self._lanes(self.buf['init_mask'].shape[0])
        main = torch.cuda.current_stream()
        # (scratch buffers of the captured launches are allocated inside the captures, ops.workspace: the eager warm-up
        # steps left cached workspaces on these very streams, and a graph must not point into a cache entry that a later,
        # larger eager request -- e.g. 480p validation between steps -- replaces)
        cap = ops.graph_capture_kwargs()       # ("thread_local" while a process group's watchdog thread is alive: ops.py)
        with self._math():
            g_pre = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_pre, **cap):
                self._pre()
            packs, lanes_a, lanes_b = [], [], []
            for l, st in enumerate(ls['streams']):
                g = torch.cuda.CUDAGraph()
                if st is None:
                    with torch.cuda.graph(g, **cap):
                        self._packs(l)
                else:
                    st.wait_stream(main)
                    with torch.cuda.graph(g, stream=st, **cap):
                        self._packs(l)
                packs.append(g)
            for st in ls['streams']:
                if st is not None:
                    main.wait_stream(st)
            for l, st in enumerate(ls['streams']):
                g = torch.cuda.CUDAGraph()
                if st is None:
                    with torch.cuda.graph(g, **cap):
                        self._lane(l, cur_iter)
                else:
                    st.wait_stream(main)
                    with torch.cuda.graph(g, stream=st, **cap):
                        self._lane(l, cur_iter)
                lanes_a.append(g)
            g_rest = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_rest, **cap):
                self._post_rest()
                if self.reduce_in_graph:
                    self._reduce_rest(captured=True)
            for l, st in enumerate(ls['streams']):
                # phase B reads tensors phase A allocated (the trunk's activations, the cut's gradients): same memory pool
                g = torch.cuda.CUDAGraph()
                if st is None:
                    with torch.cuda.graph(g, pool=lanes_a[l].pool(), **cap):
                        self._lane_trunk(l)
                else:
                    with torch.cuda.graph(g, stream=st, pool=lanes_a[l].pool(), **cap):
                        self._lane_trunk(l)
                lanes_b.append(g)
            g_post = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_post, **cap):
                out = self._post()
        torch.cuda.synchronize()
        self._graph, self._graph_out = (g_pre, packs, lanes_a, g_rest, lanes_b, g_post), out

    def _replay(self):
        g_pre, packs, lanes_a, g_rest, lanes_b, g_post = self._graph
        ls = self._lane_state
        main = torch.cuda.current_stream()
        g_pre.replay()
        for st, g in zip(ls['streams'], packs):            # the re-pack, a share per lane's stream
            if st is None:
                g.replay()
            else:
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    g.replay()
        for st in ls['streams']:
            if st is not None:
                main.wait_stream(st)                       # every pack is there before any lane reads one
        for st, g in zip(ls['streams'], lanes_a):
            if st is None:
                g.replay()
            else:
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    g.replay()
        for st in ls['streams']:
            if st is not None:
                main.wait_stream(st)                       # phase A of every lane
        g_rest.replay()
        self._reduce_rest()                                # in flight while the lanes run phase B
        for st, g in zip(ls['streams'], lanes_b):
            if st is None:
                g.replay()
            else:
                with torch.cuda.stream(st):
                    g.replay()
        for st in ls['streams']:
            if st is not None:
                main.wait_stream(st)
        g_post.replay()
