"""Differentiable stages of the SWEM training step: ``torch.autograd.Function`` wrappers whose forward AND backward are
HIP kernels (C ABI: include/swem_hip.h, include/swem_hip_train.h).

torch's autograd engine is used for what it is here -- the tape: it records which stage produced which tensor and calls
the ``backward`` below in reverse order on the forward stream.  No arithmetic is done by torch.  Activations are NHWC
fp32 tensors (B, H, W, C) as in the inference engine.

Parameter gradients are ACCUMULATED IN-KERNEL into ``param.grad`` (views of the optimizer's flat gradient buffer,
``optim.make_optimizer``): the ``backward`` functions return ``None`` for parameters.  ``param.grad`` must exist and be
zeroed before the step (``optimizer.zero_grad()``); ``ensure_grads`` creates missing ones.

Reference graph: methods/SWEM/swem_trainer.py:59-108 through swem.py / networks.py / modules.py (cited per stage).
"""
import ctypes as C

import torch
from torch.autograd import Function

from . import _lib, ops

_PACKS = {}      # (id(weight), tag) -> packed operand; parameters change every optimizer step: new_step() clears it
_RECIPES = {}    # (id(weight), tag) -> builder of that pack, recorded the first time a step needs it


def new_step(prebuild=False):
    """Drop the packed filters of the previous step.  prebuild: re-pack everything the previous steps used right away on
    the current stream -- the training step calls this before it forks its lanes, so the clips in flight share ONE set
    of packs (built lazily inside a lane they would be private to it: another lane's stream could not see them safely)."""
    _PACKS.clear()
    if prebuild:
        prebuild_packs(0, 1)


def prebuild_packs(part, nparts):
    """Re-pack, on the CURRENT stream, every n-th recorded recipe (index % nparts == part): the step's filter packs, frozen-BN
    folds and the operand planes their plans read last step.  The per-step re-pack is ~320 small dependent launches (2.6 ms of a
    45 ms step when one stream does it alone, before anything else can start); the training step deals it out to its lanes'
    streams and joins them before the first lane reads a pack (train.py, round 6)."""
    for i, (key, build) in enumerate(list(_RECIPES.items())):
        if i % nparts != part:
            continue
        pk = _PACKS[key] = _keyed(build(), key)
        if isinstance(pk, ops.ConvPack):
            # (the planes the layer's plan read last step are built now, before the lanes fork -- not lazily inside one lane)
            if pk.site_key in ops.F16_PACK_SITES:
                pk.planes16()
            if pk.site_key in ops.BF16_PACK_SITES:
                pk.w3


def reset(book=None):
    """Forget the recorded packs (a new trainer / model) and the fused-split hints of `book` (default: the current book) --
    the hints of the training step are keyed by its packs' recipes, i.e. by parameter identity: a trainer passes ITS book
    (SWEMTrainer.book), so that a model rebuilt at a recycled address never inherits a dead layer's hints."""
    _PACKS.clear()
    _RECIPES.clear()
    ops.F16_PACK_SITES.clear()
    ops.BF16_PACK_SITES.clear()
    b = book if book is not None else ops.BOOK
    b.hints.clear()
    b.hint_epoch.clear()


def ensure_grads(params):
    for p in params:
        if p.requires_grad and p.grad is None:
            p.grad = torch.zeros_like(p)


_LANE = 0          # clips of a batch may run on several streams at once (train.py): each lane accumulates into its own
_LANE_GRADS = None  # gradient buffer ({id(param): tensor view}), and packs its own copies of the filters


_SIDE = None        # the lane's WEIGHT-GRADIENT stream (train.py, round 6), or None: everything on the lane's own stream
_SIDE_KEEP = []     # tensors the side stream reads, kept alive until `join_side` (no reuse of their memory before it is done)


def use_lane(lane, grads=None, side=None):
    """Route the in-kernel parameter-gradient accumulation of the following backward calls to `grads` ({id(param):
    view of that lane's flat buffer}); None = the parameters' own .grad.  side: a second stream for the lane's weight gradients
    (`_Conv.backward`): a layer's dW needs dY and the layer's input only -- nothing downstream of it waits for dW -- so it runs
    beside the data-gradient chain (dY -> dX -> the previous layer), which is the backward pass's critical path."""
    global _LANE, _LANE_GRADS, _SIDE
    _LANE, _LANE_GRADS, _SIDE = lane, grads, side


def join_side():
    """The lane's stream waits for its weight-gradient stream; what that stream was reading may now be freed."""
    if _SIDE is not None:
        torch.cuda.current_stream().wait_stream(_SIDE)
    del _SIDE_KEEP[:]


def _grad(p):
    if _LANE_GRADS is not None:
        return _LANE_GRADS[id(p)]
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    if not p.grad.is_contiguous():
        raise _lib.SwemHipError('parameter gradient buffers must be contiguous')
    return p.grad


def _ws(nbytes, dev):
    return ops.workspace(nbytes, dev)


def colsum(a, b=None, out1=None, out2=None, accumulate=True):
    """out1[c] (+)= sum_m a[m][c]; out2[c] (+)= sum_m a[m][c]*b[m][c] over all leading dimensions."""
    Cc = a.shape[-1]
    M = a.numel() // Cc
    wsb = _lib.query('swem_colsum_workspace', M, Cc)
    ws = _ws(wsb, a.device)
    _lib.call('swem_colsum_f32', ops._stream(), a.data_ptr(), ops._ptr(b), ops._ptr(out1), ops._ptr(out2), M, Cc,
              int(accumulate), ws.data_ptr(), wsb)


def sum_batch(x, out=None, accumulate=False):
    """(B, ...) -> (1, ...): gradient of a map shared by the objects of a frame."""
    B = x.shape[0]
    n = x.numel() // B
    y = out if out is not None else torch.empty((1,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    _lib.call('swem_sum_batch_f32', ops._stream(), x.data_ptr(), y.data_ptr(), B, n, int(accumulate))
    return y


# --------------------------------------------------------------------------------------------- convolution
def _keyed(pk, key):
    """Filters are re-packed every step; the layer's name in the fused-split hints (ops.PlanBook.hints) must not change
    with them: it is the pack's recipe key (the parameter's identity + role), the same for every step and every lane."""
    if isinstance(pk, ops.ConvPack):
        pk.site_key = ('train',) + tuple(key)
    return pk


def _shared_pack(key, build):
    """The step's shared pack if new_step(prebuild=True) made it, else this lane's own (and remember how to build it)."""
    pk = _PACKS.get(key)
    if pk is None:
        _RECIPES.setdefault(key, build)
        pk = _PACKS.get(key + (_LANE,))
        if pk is None:
            pk = _PACKS[key + (_LANE,)] = _keyed(build(), key)
    return pk


def _fwd_pack(weight, bias, stride, pad, cin_pad):
    def build():
        pk = ops.pack_conv(weight, bias, None, stride, pad, cin_pad=cin_pad, lazy_planes=True)
        pk.fast16 = True           # (re-packed every step: the fp16 planes of an f16x3 plan in one launch)
        return pk
    return _shared_pack((id(weight), 'fwd', cin_pad), build)


def _dgrad_pack(weight, off, c, stride, pad, cin_pad):
    """Filters of the data-gradient GEMM for the source that owns input channels [off, off+c): [c][KH][KW][Cout]."""
    def build():
        co, ci, kh, kw = weight.shape
        w = weight.detach()
        if cin_pad is not None and cin_pad != ci:
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cin_pad - ci))
        wt = w[:, off:off + c].permute(1, 2, 3, 0).contiguous()        # [c][KH][KW][Cout]
        pk = ops.ConvPack(wt, None, None, c, kh, kw, stride, pad, lazy_planes=True)
        pk.fast16 = True
        return pk
    return _shared_pack((id(weight), 'dgrad', off, c, cin_pad), build)


def wgrad_math(cs, cout, kh, kw, M):
    """Which kernel takes a layer's weight gradient: 0 = fp32 MFMA on the fp32 tensors, 1 = bf16x6 / 2 = plain bf16 / 3 = f16x3 on
    the pre-split planes (include/swem_hip_train.h).  Under ops.conv_math((2,)) (config.AMP) every layer the bf16 kernel
    is not slower on takes plain bf16; otherwise the large 3x3 layers take bf16x6 (fp32-level error) -- measured with
    tools/wgrad_bench.py on the training shapes."""
    if any(c % 8 for c in cs) or cout % 8:
        return 0
    big = cout >= 128 and all(c >= 128 for c in cs)
    if ops._PLAN_TAG == ('math', 2):
        return 2 if (kh * kw > 1 or (big and M >= 2048)) else 0
    if 7 in ops.CONV_MATH_MODES and ops._PLAN_TAG:
        # fp32-level training on the f16x3 arithmetic (round 5): the plane kernel with fp16 pairs, dY scaled (three products
        # instead of bf16x6's six)
        return 3                   # (every layer: faster than the fp32-MFMA kernel on all training shapes, tools/wgrad_bench.py)
    return 1 if (big and kh * kw > 1 and 1 in ops.CONV_MATH_MODES) else 0


def _wgrad(dy, srcs, weight, stride, pad, relu_in):
    """dW (+)= dY^T * act(x) into the parameter's gradient view (swem_conv2d_wgrad_f32 / _bf16x3)."""
    B, Ho, Wo, Cout = dy.shape
    H, W = srcs[0].shape[1:3]
    dev = dy.device
    kh, kw = weight.shape[2:]
    cs = [s.shape[3] for s in srcs]
    math = wgrad_math(cs, Cout, kh, kw, B * Ho * Wo)
    if ops.FLOPS is not None:
        ops.FLOPS['conv_wgrad'] = ops.FLOPS.get('conv_wgrad', 0.0) + 2.0 * B * Ho * Wo * Cout * kh * kw * sum(cs)
        ops.FLOPS['wgrad_launches'] = ops.FLOPS.get('wgrad_launches', 0) + 1
    args = []
    npl = ops.PLANES_F16 if math == 3 else 3
    for s in srcs:
        bs = 0 if (s.shape[0] == 1 and B > 1) else (s.stride(0) if s.shape[0] > 1 else H * W * s.shape[3])
        if math:
            sp = ops.presplit(s, relu_in, npl)
            args += [sp.data_ptr(), s.shape[3], bs, sp.stride(0)]
        else:
            args += [s.data_ptr(), s.shape[3], bs]
    for _ in range(3 - len(srcs)):
        args += [0, 0, 0, 0] if math else [0, 0, 0]
    cs = cs + [0, 0]
    if math == 3:
        d2 = ops.presplit(dy, False, ops.PLANES_F16)             # (dy is marked a gradient map: the scaled pair)
        wsb = _lib.query('swem_conv2d_wgrad_bf16x3_workspace', B, H, W, cs[0], cs[1], cs[2], Cout, kh, kw, stride, pad, 0)
        ws = _ws(wsb, dev)
        _lib.call('swem_conv2d_wgrad_f16x3', ops._stream(), d2.data_ptr(), d2.stride(0), *args, B, H, W, Cout, kh, kw,
                  stride, pad, dy.__dict__['_swem_inv'].data_ptr(), _grad(weight).data_ptr(), weight.shape[1], 1, 0,
                  ws.data_ptr(), wsb)
        return
    if math:
        d3 = ops.presplit(dy, False)
        wsb = _lib.query('swem_conv2d_wgrad_bf16x3_workspace', B, H, W, cs[0], cs[1], cs[2], Cout, kh, kw, stride, pad, 0)
        ws = _ws(wsb, dev)
        _lib.call('swem_conv2d_wgrad_bf16x3', ops._stream(), d3.data_ptr(), d3.stride(0), *args, B, H, W, Cout, kh, kw,
                  stride, pad, math, _grad(weight).data_ptr(), weight.shape[1], 1, 0, ws.data_ptr(), wsb)
        return
    wsb = _lib.query('swem_conv2d_wgrad_workspace', B, H, W, cs[0], cs[1], cs[2], Cout, kh, kw, stride, pad)
    ws = _ws(wsb, dev)
    _lib.call('swem_conv2d_wgrad_f32', ops._stream(), dy.data_ptr(), *args, B, H, W, Cout, kh, kw, stride, pad,
              int(relu_in), _grad(weight).data_ptr(), weight.shape[1], 1, ws.data_ptr(), wsb)


class _Conv(Function):
    """y = conv(act(cat(srcs)), W) + b (+ residual); act = ReLU when relu_in (networks.py:22-32 pre-activation blocks,
    mod_resnet convs, modules.py:25-26, swem.py:33).  A source with batch 1 is shared by all `batch` items."""

    @staticmethod
    def forward(ctx, weight, bias, residual, meta, *srcs):
        stride, pad, relu_in, batch, cin_pad = meta
        pk = _fwd_pack(weight, bias, stride, pad, cin_pad)
        y = ops.conv2d(list(srcs), pk, relu_in=relu_in, residual=residual, batch=batch)
        ctx.meta = meta
        ctx.save_for_backward(weight, bias, *srcs)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, pad, relu_in, batch, cin_pad = ctx.meta
        weight, bias, *srcs = ctx.saved_tensors
        dy = dy.contiguous()
        dy.__dict__['_swem_grad'] = True      # (an fp16-pair consumer of this map gets the SCALED pair: ops.presplit)
        B, Ho, Wo, Cout = dy.shape
        dev = dy.device
        H, W = srcs[0].shape[1:3]
        want_b = bias is not None and bias.requires_grad
        if _SIDE is None:
            if want_b:
                colsum(dy, out1=_grad(bias))
            if weight.requires_grad:
                _wgrad(dy, srcs, weight, stride, pad, relu_in)
        elif want_b or weight.requires_grad:
            # dW and db on the lane's side stream.  dY's operand planes are shared with the data gradient below (and with whoever
            # receives dY as a residual gradient): they are made HERE, on the lane's stream, before the fork -- a cache entry
            # written on the side stream would be read by this stream without an order between the two.
            if weight.requires_grad:
                wm = wgrad_math([s_.shape[3] for s_ in srcs], Cout, weight.shape[2], weight.shape[3], B * Ho * Wo)
                if wm:
                    ops.presplit(dy, False, ops.PLANES_F16 if wm == 3 else 3)
            cur = torch.cuda.current_stream()
            _SIDE.wait_stream(cur)
            _SIDE_KEEP.append((dy, srcs))
            with torch.cuda.stream(_SIDE):
                if want_b:
                    colsum(dy, out1=_grad(bias))
                if weight.requires_grad:
                    _wgrad(dy, srcs, weight, stride, pad, relu_in)
        grads = []
        off = 0
        for i, s in enumerate(srcs):
            c = s.shape[3]
            g = None
            if ctx.needs_input_grad[4 + i]:
                pk = _dgrad_pack(weight, off, c, stride, pad, cin_pad)
                g = ops.conv2d([dy], pk, dgrad=(H, W), mask=s if relu_in else None, batch=B)
                if s.shape[0] == 1 and B > 1:
                    g = sum_batch(g)
            grads.append(g)
            off += c
        return (None, None, dy if ctx.has_res else None, None, *grads)


def conv2d(srcs, weight, bias=None, stride=1, pad=None, relu_in=False, residual=None, batch=None, cin_pad=None):
    pad = weight.shape[-1] // 2 if pad is None else pad
    return _Conv.apply(weight, bias, residual, (stride, pad, relu_in, batch, cin_pad), *srcs)


# --------------------------------------------------------------------------------------------- frozen BatchNorm (+res, ReLU)
class _BNAct(Function):
    """y = act(bn_eval(c) + res): mod_resnet.py:58-113 with BatchNorm in eval mode (swem_trainer.py:37-39); gamma and
    beta still train."""

    @staticmethod
    def forward(ctx, c, gamma, beta, mean, var, res, relu, eps):
        Cc = c.shape[-1]
        M = c.numel() // Cc

        def build():
            f = torch.empty((3, Cc), dtype=torch.float32, device=c.device)     # alpha, shift, invstd
            _lib.call('swem_bn_fold_f32', ops._stream(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                      var.data_ptr(), eps, f[0].data_ptr(), f[1].data_ptr(), f[2].data_ptr(), Cc)
            return f
        fold = _shared_pack((id(gamma), 'bnfold', eps), build)     # once per step, shared by the frames and the lanes
        y = torch.empty_like(c)
        planes = _planes_for((id(gamma), 'y'), y, M, Cc)
        npl = ops.PLANES_F16 if (planes is not None and planes.dtype == torch.float16) else 3
        _lib.call('swem_bn_act_planes_f32', ops._stream(), c.data_ptr(), fold[0].data_ptr(), fold[1].data_ptr(), ops._ptr(res),
                  y.data_ptr(), M, Cc, int(relu), ops._ptr(planes), npl, ops._fault_ptr(c.device))
        ctx.save_for_backward(c, y, fold, gamma, beta, mean)     # (saved_tensors keeps the output without a cycle)
        ctx.flags = (relu, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        c, y, fold, gamma, beta, mean = ctx.saved_tensors
        relu, has_res = ctx.flags
        dy = dy.contiguous()
        Cc = c.shape[-1]
        M = c.numel() // Cc
        dz = torch.empty_like(c) if has_res else None
        dc = torch.empty_like(c)
        want = gamma.requires_grad or beta.requires_grad
        wsb = _lib.query('swem_bn_act_bwd_workspace', M, Cc) if want else 0
        ws = _ws(wsb, c.device) if want else None
        site = (id(gamma), 'dc')
        if Cc % 8 == 0 and ops.BOOK.hints.get(site, {}).get(False, 0) == ops.PLANES_F16:
            # the convolution behind this stage back-propagates in f16x3: it wants the SCALED fp16 pair of dc, whose first pass
            # (the map's largest magnitude) this kernel delivers as block maxima (ops.presplit, `_swem_amax`)
            nparts = _lib.query('swem_bn_act_bwd_amax_parts', M, Cc)
            scratch = torch.empty(nparts + 1, dtype=torch.float32, device=c.device)
            dc.__dict__['_swem_site'] = site
            dc.__dict__['_swem_amax'] = (scratch, nparts)
            _lib.call('swem_bn_act_bwd_amax_f32', ops._stream(), dy.data_ptr(), y.data_ptr(), c.data_ptr(), fold[0].data_ptr(),
                      mean.data_ptr(), fold[2].data_ptr(), ops._ptr(dz), dc.data_ptr(),
                      _grad(gamma).data_ptr() if gamma.requires_grad else 0, _grad(beta).data_ptr() if beta.requires_grad else 0,
                      M, Cc, int(relu), scratch[1:].data_ptr(), ops._ptr(ws), wsb)
            return dc, None, None, None, None, (dz if has_res else None), None, None
        planes = _planes_for(site, dc, M, Cc)
        _lib.call('swem_bn_act_bwd_f32', ops._stream(), dy.data_ptr(), y.data_ptr(), c.data_ptr(), fold[0].data_ptr(),
                  mean.data_ptr(), fold[2].data_ptr(), ops._ptr(dz), dc.data_ptr(),
                  _grad(gamma).data_ptr() if gamma.requires_grad else 0, _grad(beta).data_ptr() if beta.requires_grad else 0,
                  M, Cc, int(relu), ops._ptr(planes), ops._ptr(ws), wsb)
        return dc, None, None, None, None, (dz if has_res else None), None, None


def _planes_for(site, t, M, Cc):
    """The bf16 planes of a stage's output, written by the stage itself when an earlier step saw a convolution split this
    output (ops.SPLIT_HINTS): attached to the tensor where ops.presplit looks for them.  Otherwise the tensor is tagged
    with its producer so that a later split records the hint."""
    want = ops.BOOK.hints.get(site, {}).get(False, 0) if Cc % 8 == 0 else 0
    if want == ops.PLANES_F16 and site[1] == 'y':
        # (forward maps only: the fp16 pair of a GRADIENT map is scaled by its maximum, which its producer cannot know while it
        # writes -- _BNAct.backward hands the consumer the block maxima instead)
        planes = torch.empty((2, M * Cc), dtype=torch.float16, device=t.device)
        t.__dict__['_swem_split'] = {ops._pkey(False, ops.PLANES_F16): (planes, ops.PLANES_F16)}
        t.__dict__['_swem_split_ver'] = t._version
        t.__dict__['_swem_site'] = site
        return planes
    if want in (1, 2, 3):
        planes = torch.empty((3, M * Cc), dtype=torch.bfloat16, device=t.device)
        t.__dict__['_swem_split'] = {False: (planes, 3)}
        return planes
    t.__dict__['_swem_site'] = site
    return None


def bn_act(c, bn, res=None, relu=True, eps=1e-5):
    """bn = (weight, bias, running_mean, running_var) of the frozen BatchNorm2d."""
    return _BNAct.apply(c, bn[0], bn[1], bn[2], bn[3], res, relu, eps)


# --------------------------------------------------------------------------------------------- small stages
class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.maxpool(x)
        ctx.save_for_backward(x, y)       # (y is kept alive by its consumer anyway; with it the backward needs a quarter of the loads)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        B, H, W, Cc = x.shape
        dx = torch.empty_like(x)
        _lib.call('swem_maxpool3x3s2_bwd_y_f32', ops._stream(), x.data_ptr(), y.data_ptr(), dy.contiguous().data_ptr(),
                  dx.data_ptr(), B, H, W, Cc)
        return dx


def maxpool(x):
    return _MaxPool.apply(x)


class _UpsampleAdd(Function):
    """networks.py:193-194: skip + bilinear x2 of the low-resolution map (skip may be shared by the objects)."""

    @staticmethod
    def forward(ctx, skip, low, batch):
        ctx.shapes = (skip.shape, low.shape)
        return ops.upsample_add(skip, low, batch=batch)

    @staticmethod
    def backward(ctx, dy):
        sshape, lshape = ctx.shapes
        dy = dy.contiguous()
        B, Ho, Wo, Cc = dy.shape
        dlow = torch.empty(lshape, dtype=torch.float32, device=dy.device)
        _lib.call('swem_upsample_bwd_nhwc_f32', ops._stream(), dy.data_ptr(), dlow.data_ptr(), B, lshape[1], lshape[2],
                  Ho, Wo, Cc)
        dskip = sum_batch(dy) if (sshape[0] == 1 and B > 1) else dy
        return dskip, dlow, None


def upsample_add(skip, low, batch=None):
    return _UpsampleAdd.apply(skip, low, batch)


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        y = torch.empty_like(a)
        _lib.call('swem_add_f32', ops._stream(), a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel())
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _Add.apply(a, b)


class _Concat2(Function):
    """torch.cat([x0, x1], C) where the concatenation itself is a residual (networks.py:44-45 with ResNet-18)."""

    @staticmethod
    def forward(ctx, x0, x1, batch):
        ctx.meta = (x0.shape, x1.shape)
        return ops.concat2(x0, x1, batch)

    @staticmethod
    def backward(ctx, dy):
        s0, s1 = ctx.meta
        dy = dy.contiguous()
        B = dy.shape[0]
        g0 = dy[..., :s0[3]].contiguous()
        g1 = dy[..., s0[3]:].contiguous()
        if s0[0] == 1 and B > 1:
            g0 = sum_batch(g0)
        if s1[0] == 1 and B > 1:
            g1 = sum_batch(g1)
        return g0, g1, None


def concat2(x0, x1, batch):
    return _Concat2.apply(x0, x1, batch)


class _Unbatch(Function):
    """(n * G, ...) -> n tensors (G, ...): the frames of a clip -- of the G clips a lane steps as one batch, frame-major -- go
    through the key encoder in ONE pass (BatchNorm is frozen, so a sample's result does not depend on its batch mates) and are
    handed to the frame loop one by one, as views (no copy: nothing modifies them in place); the backward puts the frames'
    gradients back side by side (zeros for a frame whose output was not used)."""

    @staticmethod
    def forward(ctx, x, n):
        G = x.shape[0] // n
        ctx.meta = (x.shape, n, G)
        return tuple(x[i * G:(i + 1) * G] for i in range(n))

    @staticmethod
    def backward(ctx, *grads):
        shape, n, G = ctx.meta
        ref = next(g for g in grads if g is not None)
        parts = [g if g is not None else torch.zeros((G,) + tuple(shape[1:]), dtype=ref.dtype, device=ref.device)
                 for g in grads]
        return torch.cat(parts, 0), None


def unbatch(x, n):
    return _Unbatch.apply(x, n)


class _ExpandObjects(Function):
    """(G, ...) -> (G * N, ...), item g repeated for its N objects (train.py, clip-batched step): the reference broadcasts such
    maps inside its ops (networks.py:119-121 `f16.unsqueeze(1).expand`, modules.py:287-289 `qv.unsqueeze(1).expand`); here the G
    clips' objects form ONE convolution batch, so the shared map is laid out per object.  Backward: the N copies' gradients
    summed in a fixed order."""

    @staticmethod
    def forward(ctx, x, N):
        G = x.shape[0]
        n = x.numel() // G
        ctx.meta = (x.shape, N)
        y = torch.empty((G * N,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
        _lib.call('swem_expand_groups_f32', ops._stream(), x.contiguous().data_ptr(), y.data_ptr(), G, N, n)
        return y

    @staticmethod
    def backward(ctx, dy):
        shape, N = ctx.meta
        G = shape[0]
        dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
        _lib.call('swem_sum_groups_f32', ops._stream(), dy.contiguous().data_ptr(), dx.data_ptr(), G, N, dx.numel() // G)
        return dx, None


def expand_objects(x, N):
    """x (G, ...) shared by the N objects of each of its G items -> (G * N, ...); G == 1 or N == 1: x itself (conv2d / concat2 /
    upsample_add broadcast a batch-1 source over their `batch` without a copy)."""
    if x.shape[0] == 1 or N == 1:
        return x
    return _ExpandObjects.apply(x, N)


class _GLU(Function):
    """modules.py:25-26: layer_f(x) * sigmoid(layer_a(x)) on the two convolutions' outputs."""

    @staticmethod
    def forward(ctx, f, a):
        y = torch.empty_like(f)
        _lib.call('swem_glu_f32', ops._stream(), f.data_ptr(), a.data_ptr(), y.data_ptr(), f.numel())
        ctx.saved = (f, a)
        return y

    @staticmethod
    def backward(ctx, dy):
        f, a = ctx.saved
        df, da = torch.empty_like(f), torch.empty_like(a)
        _lib.call('swem_glu_bwd_f32', ops._stream(), dy.contiguous().data_ptr(), f.data_ptr(), a.data_ptr(),
                  df.data_ptr(), da.data_ptr(), f.numel())
        return df, da


def glu(f, a):
    return _GLU.apply(f, a)


class _CBAM(Function):
    """x + CBAM(x) (attentions.py:22-84, networks.py:46-47)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w7, b7):
        ctx.saved = (x, w1, b1, w2, b2, w7, b7)
        return ops.cbam_residual(x, w1.detach(), b1.detach(), w2.detach(), b2.detach(), w7.detach(), b7.detach())

    @staticmethod
    def backward(ctx, dy):
        x, w1, b1, w2, b2, w7, b7 = ctx.saved
        B, H, W, Cc = x.shape
        dx = torch.empty_like(x)
        wsb = _lib.query('swem_cbam_bwd_workspace', B, H, W, Cc)
        ws = _ws(wsb, x.device)
        _lib.call('swem_cbam_bwd_f32', ops._stream(), x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                  b2.data_ptr(), w7.data_ptr(), b7.data_ptr(), dy.contiguous().data_ptr(), dx.data_ptr(),
                  _grad(w1).data_ptr(), _grad(b1).data_ptr(), _grad(w2).data_ptr(), _grad(b2).data_ptr(),
                  _grad(w7).data_ptr(), _grad(b7).data_ptr(), B, H, W, Cc, w1.shape[0], ws.data_ptr(), wsb)
        return dx, None, None, None, None, None, None


def cbam_residual(x, w1, b1, w2, b2, w7, b7):
    return _CBAM.apply(x, w1, b1, w2, b2, w7, b7)


class _PredHead(Function):
    """networks.py:213: conv3x3(relu(x)) -> one channel; returns (B, H, W) logits at 1/4 scale."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        key = (id(weight), 'pred', _LANE)
        w = _PACKS.get(key)
        if w is None:
            w = _PACKS[key] = weight.detach().permute(0, 2, 3, 1).contiguous()   # [1][3][3][C]
        ctx.saved = (x, weight, bias, w)
        return ops.pred_head(x, w, bias.detach())

    @staticmethod
    def backward(ctx, dl):
        x, weight, bias, w = ctx.saved
        B, H, W, Cc = x.shape
        dx = torch.empty_like(x)
        wsb = _lib.query('swem_pred_head_bwd_workspace', B, H, W, Cc)
        ws = _ws(wsb, x.device)
        _lib.call('swem_pred_head_bwd_f32', ops._stream(), x.data_ptr(), w.data_ptr(), dl.contiguous().data_ptr(),
                  dx.data_ptr(), _grad(weight).data_ptr(), _grad(bias).data_ptr(), B, H, W, Cc, ws.data_ptr(), wsb)
        return dx, None, None


def pred_head(x, weight, bias):
    return _PredHead.apply(x, weight, bias)


class _DecodeHead(Function):
    """swem.py:99-108 + aggregate :110-116: bilinear to out_size, sigmoid, valid_obj, soft aggregation, softmax.
    logit4 (B*N, h4, w4) -> logits, prob (B, N+1, Ho, Wo)."""

    @staticmethod
    def forward(ctx, logit4, valid, B, N, out_size):
        logits, prob, _ = ops.decode_head(logit4, B, N, out_size, valid=valid)
        ctx.saved = (logit4, valid, B, N, out_size)
        return logits, prob

    @staticmethod
    def backward(ctx, dlogits, dprob):
        logit4, valid, B, N, (Ho, Wo) = ctx.saved
        h4, w4 = logit4.shape[-2:]
        d4 = torch.empty_like(logit4)
        wsb = B * N * Ho * Wo * 4
        ws = _ws(wsb, logit4.device)
        _lib.call('swem_decode_head_bwd_f32', ops._stream(), logit4.data_ptr(), ops._ptr(valid),
                  ops._ptr(None if dlogits is None else dlogits.contiguous()),
                  ops._ptr(None if dprob is None else dprob.contiguous()), d4.data_ptr(), B, N, h4, w4, Ho, Wo,
                  ws.data_ptr(), wsb)
        return d4, None, None, None, None


def decode_head(logit4, valid, B, N, out_size):
    return _DecodeHead.apply(logit4, valid, B, N, (int(out_size[0]), int(out_size[1])))


class _PrepValueInput(Function):
    """swem.py:48-53 + networks.py:115-117: [normalised image, object mask, other objects' mask] per object (padded
    to 8 channels); the masks carry gradient (the predicted soft masks feed the value encoder, swem_trainer.py:88)."""

    @staticmethod
    def forward(ctx, frame, masks, mean3, std3, single_obj):
        ctx.meta = (masks.shape, single_obj)
        return ops.prep_value_input(frame, masks, mean3, std3, single_obj)

    @staticmethod
    def backward(ctx, dx):
        (B, N1, H, W), single_obj = ctx.meta
        dm = torch.empty((B, N1, H, W), dtype=torch.float32, device=dx.device)
        _lib.call('swem_prep_value_input_bwd_f32', ops._stream(), dx.contiguous().data_ptr(), dm.data_ptr(), B, N1 - 1, H,
                  W, int(single_obj))
        return None, dm, None, None, None


def prep_value_input(frame, masks, mean3, std3, single_obj):
    return _PrepValueInput.apply(frame, masks, mean3, std3, single_obj)


# --------------------------------------------------------------------------------------------- EM value update, matching
class _Memorize(Function):
    """SWEMCore.swem (modules.py:129-168) for the N objects of each of G clips (the reference's `B` dimension; G = 1 when x is
    (P,C)).  x (G,P,C) raw key (no gradient: the E/M/W steps run under no_grad), v (G*N,P,V) value map, masks (G*N,2,P), prior
    bases (G*N, ...).  Gradient flows v -> nu and nu_prev -> nu.  Clips are independent problems: one library call per clip on
    that clip's slices (EM + matching are 2 % of the step; the convolutions around them run the G clips as one batch)."""

    @staticmethod
    def forward(ctx, v, nu_prev, x, masks, kappa_prev, zita_prev, T, tau):
        GN, P, V = v.shape
        G = 1 if x.dim() == 2 else x.shape[0]
        N = GN // G
        Cc = x.shape[-1]
        L = kappa_prev.shape[-1]
        dev = v.device
        for t_, nm in ((v, 'v'), (nu_prev, 'nu_prev'), (x, 'x'), (masks, 'masks'), (kappa_prev, 'kappa_prev'), (zita_prev, 'zita_prev')):
            if not t_.is_contiguous():
                raise _lib.SwemHipError('memorize: %s must be contiguous' % nm)
        kappa = torch.empty_like(kappa_prev)
        nu = torch.empty_like(nu_prev)
        zita = torch.empty_like(zita_prev)
        Pz = _lib.query('swem_em_pad', P)
        zT = torch.empty((GN, Pz, 2 * L), dtype=torch.float32, device=dev)   # pixel-major z
        wsb = _lib.query('swem_memorize_workspace', N, Cc, V, P, L)
        ws = _ws(wsb, dev)
        st = ops._stream()
        for g in range(G):
            o = 4 * g * N                       # byte offset of clip g's first object, per element of an object's slice
            _lib.call('swem_memorize_train_f32', st, x.data_ptr() + 4 * g * P * Cc, v.data_ptr() + o * P * V,
                      masks.data_ptr() + o * 2 * P, kappa_prev.data_ptr() + o * 2 * Cc * L, nu_prev.data_ptr() + o * 2 * V * L,
                      zita_prev.data_ptr() + o * 2 * L, kappa.data_ptr() + o * 2 * Cc * L, nu.data_ptr() + o * 2 * V * L,
                      zita.data_ptr() + o * 2 * L, zT.data_ptr() + o * Pz * 2 * L, N, Cc, V, P, L, T, tau, ws.data_ptr(), wsb)
        if ops.FLOPS is not None:     # F_mem = 4 P L (C (3T - 1) + V) per object; the value update's backward: two GEMMs of 4 V P L
            ops.FLOPS['em_match'] = ops.FLOPS.get('em_match', 0.0) + GN * 4.0 * P * L * (Cc * (3 * T - 1) + V) + GN * 8.0 * V * P * L
        ctx.save_for_backward(zT, zita_prev, zita)
        ctx.dims = (G, N, P, V, L)
        ctx.mark_non_differentiable(kappa, zita)
        return kappa, nu, zita

    @staticmethod
    def backward(ctx, dkappa, dnu, dzita):
        zT, zita_prev, zita = ctx.saved_tensors
        G, N, P, V, L = ctx.dims
        dev = zT.device
        dv = torch.empty((G * N, P, V), dtype=torch.float32, device=dev)
        want_prev = ctx.needs_input_grad[1]
        dnu = dnu.contiguous()
        dnu_prev = torch.empty_like(dnu) if want_prev else None
        wsb = _lib.query('swem_nu_update_bwd_workspace', N, V, P, L)
        ws = _ws(wsb, dev)
        st = ops._stream()
        Pz = zT.shape[1]
        for g in range(G):
            o = 4 * g * N
            _lib.call('swem_nu_update_bwd_f32', st, zT.data_ptr() + o * Pz * 2 * L, zita_prev.data_ptr() + o * 2 * L,
                      zita.data_ptr() + o * 2 * L, dnu.data_ptr() + o * 2 * V * L, dv.data_ptr() + o * P * V,
                      (dnu_prev.data_ptr() + o * 2 * V * L) if want_prev else 0, N, V, P, L, ws.data_ptr(), wsb)
        return dv, dnu_prev, None, None, None, None, None, None


def memorize(v, nu_prev, x, masks, kappa_prev, zita_prev, T, tau):
    return _Memorize.apply(v, nu_prev, x, masks, kappa_prev, zita_prev, T, tau)


class _Match(Function):
    """get_affinity + perm_inv_feat (modules.py:198-208, 232-276) for the N objects of each of G clips (G = 1 when qk is (P,C)).
    qk (G,P,C) raw query key; banks: kappa (G*N,2,C,L) no gradient, nu (G*N,2,V,L) with gradient.
    -> mem_out (G*N,Pm,V) (rows >= P zero), S (G*N,P,2*topl).  One library call per clip (see _Memorize)."""

    @staticmethod
    def forward(ctx, qk, nu_first, nu_update, kappa_first, kappa_update, topl, tau):
        G = 1 if qk.dim() == 2 else qk.shape[0]
        P, Cc = qk.shape[-2:]
        GN, _, V, L = nu_first.shape
        N = GN // G
        dev = qk.device
        for t_ in (qk, nu_first, nu_update, kappa_first, kappa_update):
            if t_ is not None and not t_.is_contiguous():
                raise _lib.SwemHipError('match: operands must be contiguous')
        nb = 1 if kappa_update is None else 2
        Pm = _lib.query('swem_match_pad', P)
        mem = torch.empty((GN, Pm, V), dtype=torch.float32, device=dev)
        S = torch.empty((GN, P, 2 * topl), dtype=torch.float32, device=dev)
        wsb = _lib.query('swem_match_workspace', N, Cc, V, P, L, nb, 0)
        ws = _ws(wsb, dev)
        st = ops._stream()
        for g in range(G):
            o = 4 * g * N
            _lib.call('swem_match_f32', st, qk.data_ptr() + 4 * g * P * Cc, kappa_first.data_ptr() + o * 2 * Cc * L,
                      nu_first.data_ptr() + o * 2 * V * L, 0 if kappa_update is None else kappa_update.data_ptr() + o * 2 * Cc * L,
                      0 if nu_update is None else nu_update.data_ptr() + o * 2 * V * L, mem.data_ptr() + o * Pm * V,
                      S.data_ptr() + o * P * 2 * topl, N, Cc, V, P, L, topl, tau, 0, ws.data_ptr(), wsb)
        if ops.FLOPS is not None:     # F_match = 4 Lm P (C + V) per object, Lm = nb L; backward: twice that
            ops.FLOPS['em_match'] = ops.FLOPS.get('em_match', 0.0) + 3.0 * GN * 4.0 * nb * L * P * (Cc + V)
        ctx.saved = (qk, nu_first, nu_update, kappa_first, kappa_update, topl, tau)
        return mem, S

    @staticmethod
    def backward(ctx, dmem, dS):
        qk, nu_first, nu_update, kappa_first, kappa_update, topl, tau = ctx.saved
        G = 1 if qk.dim() == 2 else qk.shape[0]
        P, Cc = qk.shape[-2:]
        GN, _, V, L = nu_first.shape
        N = GN // G
        dev = qk.device
        nb = 1 if kappa_update is None else 2
        Pm = _lib.query('swem_match_pad', P)
        if dmem is None:
            dmem = torch.zeros((GN, Pm, V), dtype=torch.float32, device=dev)
        dmem = dmem.contiguous()
        dS = None if dS is None else dS.contiguous()
        dqk = torch.empty_like(qk)
        dn1 = torch.empty_like(nu_first)
        dn2 = torch.empty_like(nu_update) if nu_update is not None else None
        wsb = _lib.query('swem_match_bwd_workspace', N, Cc, V, P, L, nb)
        ws = _ws(wsb, dev)
        st = ops._stream()
        for g in range(G):
            o = 4 * g * N
            _lib.call('swem_match_bwd_f32', st, qk.data_ptr() + 4 * g * P * Cc, kappa_first.data_ptr() + o * 2 * Cc * L,
                      nu_first.data_ptr() + o * 2 * V * L, 0 if kappa_update is None else kappa_update.data_ptr() + o * 2 * Cc * L,
                      0 if nu_update is None else nu_update.data_ptr() + o * 2 * V * L, dmem.data_ptr() + o * Pm * V,
                      0 if dS is None else dS.data_ptr() + o * P * 2 * topl, dqk.data_ptr() + 4 * g * P * Cc,
                      dn1.data_ptr() + o * 2 * V * L, 0 if dn2 is None else dn2.data_ptr() + o * 2 * V * L, N, Cc,
                      V, P, L, topl, tau, ws.data_ptr(), wsb)
        return dqk, dn1, dn2, None, None, None, None


def match(qk, nu_first, nu_update, kappa_first, kappa_update, topl, tau):
    return _Match.apply(qk, nu_first, nu_update, kappa_first, kappa_update, topl, tau)
