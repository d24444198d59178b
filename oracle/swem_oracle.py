"""CPU oracle for the SWEM hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (torch CPU ops, fp32) of the reference algorithm
for the path BASELINE.json names.  It exists so that the HIP path can be checked
against something that is *not* the HIP path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product package ``swem_amd`` never does.

Parity status: PINNED for everything except the torchvision trunk.
  * EM / matching / mask prep / value encoder / decoder / frame loop are checked
    against the reference's own Python (imported from /root/reference with the
    shims in ``tests/golden/make_golden.py``); the resulting vectors are
    committed under ``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py``.
  * key-encoder trunk: the reference takes it from ``torchvision.models.resnet18/50``
    (requirements.txt:10, unpinned, not vendored, not installed here).  The
    topology below restates the published torchvision v1.5 ResNet
    (stride on the 3x3 of the bottleneck, bias-free convs, BN after every conv);
    it is checked against the reference's ``mod_resnet`` trunk with zero conv
    biases (same topology), which is as far as this container allows:
    "parity unpinned" for torchvision itself.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  State dict keys are the reference's.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

EPS_L2 = 1e-6
BN_EPS = 1e-5


# --------------------------------------------------------------------------- #
# EM core (methods/SWEM/modules.py)
# --------------------------------------------------------------------------- #
def l2norm(t, dim):
    """modules.py:7-9 -- eps is added AFTER the sqrt."""
    return t / (torch.linalg.norm(t, dim=dim, keepdim=True) + EPS_L2)


def random_init(size, valdim, dtype=torch.float32):
    """modules.py:170-178.  Draws from the global torch CPU generator, like the
    reference does for CPU tensors, so a shared manual_seed gives equal bases."""
    b, n, _, _, nb = size
    kappa = torch.zeros(size, dtype=dtype)
    kappa.normal_(0, math.sqrt(2.0 / nb))
    kappa = l2norm(kappa, dim=-2)
    nu = torch.zeros(b, n, 2, valdim, nb, dtype=dtype)
    zita = torch.zeros(b, n, 2, 1, nb, dtype=dtype) + 1e-6
    return kappa, nu, zita


@torch.no_grad()
def e_step(x_t, kappa, weights, tau):
    """modules.py:112-120.  x_t (B,1,1,P,C) is the RAW key; kappa (B,N,2,C,L).  Like the M and W steps it runs under
    no_grad in the reference (decorators at :93,:112,:122): in training only the value update carries gradient."""
    s = torch.matmul(x_t, l2norm(kappa, dim=-2))
    s = (s - s.max(dim=-1, keepdim=True)[0]) / tau
    return F.softmax(s, dim=-1) * weights


@torch.no_grad()
def m_step(z, x, kappa_prev, zita_prev):
    """modules.py:122-127.  The prior is the previous FRAME's (kappa_, zita_)."""
    zita = zita_prev + z.sum(dim=-2, keepdim=True)
    kappa = (zita_prev * kappa_prev + torch.matmul(x, z)) / zita
    return kappa, zita


@torch.no_grad()
def w_step(kappa, x_t, masks, tau):
    """modules.py:93-110.  Joint {bg,fg} max, literal ``1 - p``."""
    c = torch.matmul(l2norm(x_t, dim=-1), l2norm(kappa, dim=-2))
    m = c.max(dim=-1, keepdim=True)[0].max(dim=2, keepdim=True)[0]
    se = torch.exp((c - m) / tau).sum(dim=-1, keepdim=True)
    p = se / se.sum(dim=2, keepdim=True)
    return masks * (1 - p)


def swem(x, v, masks, bases_prev, n_bases, n_iters, tau, valdim):
    """modules.py:129-168.  x (B,C,h,w), v (B,N,V,h,w), masks (B,N,2,h,w)."""
    b, ck, _, _ = x.shape
    n = masks.shape[1]
    if bases_prev is None:
        kappa_, nu_, zita_ = random_init((b, n, 2, ck, n_bases), valdim, x.dtype)
    else:
        kappa_, nu_, zita_ = bases_prev['kappa'], bases_prev['nu'], bases_prev['zita']
    n_new = n - kappa_.shape[1]
    if n_new > 0:  # modules.py:140-146 (objects that appear later)
        k2, n2, z2 = random_init((b, n_new, 2, ck, n_bases), valdim, x.dtype)
        kappa_ = torch.cat([kappa_, k2], 1)
        nu_ = torch.cat([nu_, n2], 1)
        zita_ = torch.cat([zita_, z2], 1)

    xf = x.flatten(2)[:, None, None]            # B,1,1,C,P
    x_t = xf.transpose(-2, -1)                   # B,1,1,P,C
    mk = masks.flatten(3).unsqueeze(-1)          # B,N,2,P,1
    weights = mk.clone()
    kappa = kappa_.clone()
    z = zita = None
    for it in range(n_iters):
        z = e_step(x_t, kappa, weights, tau)
        kappa, zita = m_step(z, xf, kappa_, zita_)
        if it < n_iters - 1:
            weights = w_step(kappa, x_t, mk, tau)
    mv = v.flatten(3).unsqueeze(2)               # B,N,1,V,P
    nu = (zita_ * nu_ + torch.matmul(mv, z)) / zita   # modules.py:164-165
    return {'kappa': kappa, 'nu': nu, 'zita': zita}


def perm_inv_feat(e, topl):
    """modules.py:198-208.  e (BN,2,Lm,h,w) after exp.  The prefix sums are added
    strictly left to right like the reference's Python loop (torch.cumsum rounds
    differently by ~2e-7, checked in make_golden.py), so this stays bit-exact."""
    top = torch.topk(e, k=topl, dim=2)[0]
    planes = [top[:, :, 0]]
    for i in range(1, topl):
        planes.append(planes[-1] + top[:, :, i])
    c = torch.stack(planes, dim=2)
    f = c[:, 0] / (c[:, 0] + c[:, 1])
    return torch.cat([f, 1 - f], dim=1)


def get_affinity(qk, mk, mv, tau, topl):
    """modules.py:232-276, default branch (:264-266).  qk and mk already l2-normed."""
    b, _, h, w = qk.shape
    n = mk.shape[1]
    q = qk.flatten(2)[:, None, None]                           # B,1,1,C,P
    aff = torch.matmul(mk.transpose(-2, -1), q)                # B,N,2,Lm,P
    m = aff.max(dim=2, keepdim=True)[0].max(dim=3, keepdim=True)[0]
    e = torch.exp((aff - m) / tau)
    p = (e / e.sum(dim=[2, 3], keepdim=True)).flatten(2, 3)    # B,N,2Lm,P
    s_feat = perm_inv_feat(e.view(b * n, 2, -1, h, w), topl)
    mvf = mv.transpose(2, 3).flatten(-2)                       # B,N,V,2Lm
    mem_out = torch.matmul(mvf, p).view(b, n, -1, h, w)
    return s_feat, mem_out


class MemoryBank:
    """modules.py:29-60."""

    def __init__(self, fixed):
        self.fixed = fixed
        self.bases = None
        self.n_objs = 0

    def reset(self):
        self.bases, self.n_objs = None, 0

    def update(self, bases):
        if not self.fixed:
            self.bases = bases
            return
        if self.bases is None:
            self.bases = bases
        else:
            n = bases['kappa'].shape[1]
            if n > self.n_objs:
                self.bases = {k: torch.cat([self.bases[k], bases[k][:, self.n_objs:]], 1)
                              for k in bases}
        self.n_objs = bases['kappa'].shape[1]


class Core:
    """SWEMCore bank policy: modules.py:63-88, 183-193, 278-306."""

    def __init__(self, n_bases=256, valdim=512, n_iters=4, tau=0.05, topl=64):
        self.n_bases, self.valdim, self.n_iters, self.tau = n_bases, valdim, n_iters, tau
        self.topl = int(min(n_bases, topl))
        self.first = MemoryBank(fixed=True)
        self.upd = MemoryBank(fixed=False)

    def empty(self):
        self.first.reset()
        self.upd.reset()

    def memorize(self, qk, qv, masks):
        prior = self.first.bases if self.upd.bases is None else self.upd.bases
        bases = swem(qk, qv, masks, prior, self.n_bases, self.n_iters, self.tau, self.valdim)
        had_first = self.first.bases is not None
        self.first.update(bases)
        if had_first:
            self.upd.update(bases)
        return bases

    def get_mem(self):
        banks = [b.bases for b in (self.first, self.upd) if b.bases is not None]
        return (torch.cat([b['kappa'] for b in banks], -1),
                torch.cat([b['nu'] for b in banks], -1))

    def match_features(self, qk, qv):
        """modules.py:278-289 up to (not including) the fusion conv."""
        mk, mv = self.get_mem()
        s_feat, mem_out = get_affinity(l2norm(qk, 1), l2norm(mk, -2), mv, self.tau, self.topl)
        qv_e = qv.unsqueeze(1).expand_as(mem_out).flatten(0, 1)
        return mem_out.flatten(0, 1), qv_e, s_feat, mk.shape[1]


# --------------------------------------------------------------------------- #
# Networks (methods/basic_modules/{networks,mod_resnet,attentions}.py)
# --------------------------------------------------------------------------- #
# Mixed-precision restatement (test_gpu_train.py::test_amp_step_vs_rounded_operand_oracle).  The reference's config.AMP is
# fp16 autocast (basic_trainer.py:83-86,222); the build's AMP rounds the CONVOLUTION operands to bf16 (round to nearest
# even) and accumulates in fp32.  ROUNDED_CONV, when set, is a policy  f(name, x, w, stride) -> (fwd, dgrad, wgrad)  that
# says which of a layer's three GEMMs take rounded operands; the arithmetic below is then exactly "fp32 conv on rounded
# operands" for those.  None (the default) = the reference's fp32 path, untouched.
ROUNDED_CONV = None


def _bf16r(t):
    return t.bfloat16().to(t.dtype)


class _ConvRounded(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, flags):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, flags, b is not None)
        xf, wf = (_bf16r(x), _bf16r(w)) if flags[0] else (x, w)
        return F.conv2d(xf, wf, b, stride=stride, padding=pad)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, flags, has_b = ctx.cfg
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            d, ww = (_bf16r(dy), _bf16r(w)) if flags[1] else (dy, w)
            dx = torch.nn.grad.conv2d_input(x.shape, ww, d, stride=stride, padding=pad)
        if ctx.needs_input_grad[1]:
            d, xx = (_bf16r(dy), _bf16r(x)) if flags[2] else (dy, x)
            dw = torch.nn.grad.conv2d_weight(xx, w.shape, d, stride=stride, padding=pad)
        if has_b and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dw, db, None, None, None


def conv(sd, name, x, stride=1, pad=None):
    w = sd[name + '.weight']
    if pad is None:
        pad = w.shape[-1] // 2
    if ROUNDED_CONV is not None:
        flags = ROUNDED_CONV(name, x, w, stride)
        if any(flags):
            return _ConvRounded.apply(x, w, sd.get(name + '.bias'), stride, pad, tuple(flags))
    return F.conv2d(x, w, sd.get(name + '.bias'), stride=stride, padding=pad)


def bn(sd, name, x):
    """Frozen (eval-mode) BatchNorm2d; the trainer also forces eval (swem_trainer.py:39)."""
    return F.batch_norm(x, sd[name + '.running_mean'], sd[name + '.running_var'],
                        sd[name + '.weight'], sd[name + '.bias'], False, 0.0, BN_EPS)


def basic_block(sd, p, x, stride):
    """mod_resnet.py:45-74 / torchvision BasicBlock."""
    out = F.relu(bn(sd, p + '.bn1', conv(sd, p + '.conv1', x, stride)))
    out = bn(sd, p + '.bn2', conv(sd, p + '.conv2', out))
    if (p + '.downsample.0.weight') in sd:
        x = bn(sd, p + '.downsample.1', conv(sd, p + '.downsample.0', x, stride, 0))
    return F.relu(out + x)


def bottleneck(sd, p, x, stride):
    """mod_resnet.py:77-113 / torchvision v1.5 Bottleneck (stride on the 3x3)."""
    out = F.relu(bn(sd, p + '.bn1', conv(sd, p + '.conv1', x, 1, 0)))
    out = F.relu(bn(sd, p + '.bn2', conv(sd, p + '.conv2', out, stride)))
    out = bn(sd, p + '.bn3', conv(sd, p + '.conv3', out, 1, 0))
    if (p + '.downsample.0.weight') in sd:
        x = bn(sd, p + '.downsample.1', conv(sd, p + '.downsample.0', x, stride, 0))
    return F.relu(out + x)


RESNET_LAYERS = {'resnet18': (basic_block, (2, 2, 2)), 'resnet50': (bottleneck, (3, 4, 6))}


def resnet_stage(sd, p, x, block, nblocks, stride):
    for i in range(nblocks):
        x = block(sd, f'{p}.{i}', x, stride if i == 0 else 1)
    return x


def trunk(sd, p, x, backbone, layer_names):
    """stem + three stages: networks.py:160-170 (key) and :119-125 (value)."""
    block, nb = RESNET_LAYERS[backbone]
    x = F.relu(bn(sd, p + '.bn1', conv(sd, p + '.conv1', x, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    f4 = resnet_stage(sd, f'{p}.{layer_names[0]}', x, block, nb[0], 1)
    f8 = resnet_stage(sd, f'{p}.{layer_names[1]}', f4, block, nb[1], 2)
    f16 = resnet_stage(sd, f'{p}.{layer_names[2]}', f8, block, nb[2], 2)
    return f16, f8, f4


def res_block(sd, p, x):
    """networks.py:12-32 (pre-activation residual block, optional 3x3 downsample)."""
    r = conv(sd, p + '.conv1', F.relu(x))
    r = conv(sd, p + '.conv2', F.relu(r))
    if (p + '.downsample.weight') in sd:
        x = conv(sd, p + '.downsample', x)
    return x + r


def cbam(sd, p, x):
    """attentions.py:22-84."""
    def mlp(t):
        t = F.linear(t.flatten(1), sd[p + '.ChannelGate.mlp.1.weight'], sd[p + '.ChannelGate.mlp.1.bias'])
        return F.linear(F.relu(t), sd[p + '.ChannelGate.mlp.3.weight'], sd[p + '.ChannelGate.mlp.3.bias'])
    hw = x.shape[-2:]
    att = mlp(F.avg_pool2d(x, hw, stride=hw)) + mlp(F.max_pool2d(x, hw, stride=hw))
    x = x * torch.sigmoid(att)[:, :, None, None]
    comp = torch.cat([x.max(1, keepdim=True)[0], x.mean(1, keepdim=True)], 1)
    return x * torch.sigmoid(conv(sd, p + '.SpatialGate.spatial.conv', comp, 1, 3))


def encode_key(sd, cfg, frames):
    """swem.py:39-43 + networks.py:160-182."""
    f = (frames - sd['key_encoder.mean']) / sd['key_encoder.std']
    s16, s8, s4 = trunk(sd, 'key_encoder', f, cfg.BACKBONE, ('res2', 'layer2', 'layer3'))
    return conv(sd, 'key_proj.key_proj', s16), conv(sd, 'key_comp', s16), s16, s8, s4


def encode_value(sd, cfg, frame, masks, s16):
    """swem.py:45-62 + networks.py:56-129 + :35-50."""
    n = masks.shape[1] - 1
    others = 1 - masks - masks[:, 0:1]
    m_fg = masks[:, 1:].flatten(0, 1).unsqueeze(1)
    m_ot = others[:, 1:].flatten(0, 1).unsqueeze(1)
    fr = frame.unsqueeze(1).expand(-1, n, -1, -1, -1).flatten(0, 1)
    s16e = s16.unsqueeze(1).expand(-1, n, -1, -1, -1).flatten(0, 1)
    img = (fr - sd['value_encoder.mean']) / sd['value_encoder.std']
    f = torch.cat([img, m_fg] if cfg.SINGLE_OBJ else [img, m_fg, m_ot], 1)
    x, _, _ = trunk(sd, 'value_encoder', f, 'resnet18', ('layer1', 'layer2', 'layer3'))
    x = res_block(sd, 'value_encoder.fuser.block1', torch.cat([x, s16e], 1))
    x = res_block(sd, 'value_encoder.fuser.block2', x + cbam(sd, 'value_encoder.fuser.attention', x))
    return x.view(-1, n, *x.shape[1:])


def mask_prep(masks_hard, masks_soft, h16, w16):
    """swem.py:77-84."""
    mh = F.interpolate(masks_hard[:, 1:].float(), size=(h16, w16), mode='nearest')
    ms = F.interpolate(masks_soft[:, 1:], size=(h16, w16), mode='bilinear')
    return torch.stack([(1 - mh) * (1 - ms), mh * ms], dim=2)


def fusion_layer(sd, x):
    """modules.py:13-26 (GLU)."""
    return conv(sd, 'swem_core.fusion_layer.layer_f', x) * torch.sigmoid(conv(sd, 'swem_core.fusion_layer.layer_a', x))


def aggregate(prob):
    """swem.py:110-116."""
    p = torch.cat([torch.prod(1 - prob, dim=1, keepdim=True), prob], 1).clamp(1e-7, 1 - 1e-7)
    return torch.log(p / (1 - p))


def decoder_logit(sd, context, s8e, s4e):
    """networks.py:199-213, up to the single-channel 1/4-scale logit."""
    x = res_block(sd, 'decoder.compress', context)
    sk = conv(sd, 'decoder.up_16_8.skip_conv', s8e)
    x = res_block(sd, 'decoder.up_16_8.out_conv',
                  sk + F.interpolate(x, size=sk.shape[-2:], mode='bilinear', align_corners=False))
    sk = conv(sd, 'decoder.up_8_4.skip_conv', s4e)
    x = res_block(sd, 'decoder.up_8_4.out_conv',
                  sk + F.interpolate(x, size=sk.shape[-2:], mode='bilinear', align_corners=False))
    return conv(sd, 'decoder.pred', F.relu(x))


def decode(sd, n, context, s8, s4, valid_obj, out_size):
    """swem.py:92-108."""
    s8e = s8.unsqueeze(1).expand(-1, n, -1, -1, -1).flatten(0, 1)
    s4e = s4.unsqueeze(1).expand(-1, n, -1, -1, -1).flatten(0, 1)
    lg = decoder_logit(sd, context, s8e, s4e)
    lg = F.interpolate(lg, size=out_size, mode='bilinear', align_corners=False)
    preds = torch.sigmoid(lg).view(-1, n, *lg.shape[-2:])
    if valid_obj is not None:
        preds = preds * valid_obj[:, 1:, None, None]
    logits = aggregate(preds)
    return logits, F.softmax(logits, dim=1)


class Model:
    """Functional counterpart of SWEM(nn.Module) (swem.py:9-133) over a plain state dict."""

    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg
        self.core = Core(cfg.NUM_BASES, cfg.VALDIM, cfg.NUM_EM_ITERS, cfg.EM_TAU, cfg.TOPL)

    def __call__(self, mode, *a):
        sd, cfg = self.sd, self.cfg
        if mode == 'encode_key':
            return encode_key(sd, cfg, *a)
        if mode == 'encode_value':
            return encode_value(sd, cfg, *a)
        if mode == 'init':
            qk, mv, mask = a
            self.core.empty()
            return self('memorize', qk, mv, mask, mask.float())
        if mode == 'memorize':
            qk, mv, mh, ms = a
            return self.core.memorize(qk, mv, mask_prep(mh, ms, qk.shape[-2], qk.shape[-1]))
        if mode == 'match':
            mem_out, qv_e, s_feat, n = self.core.match_features(*a)
            return fusion_layer(sd, torch.cat([mem_out, qv_e, s_feat], 1)), n
        if mode == 'segment':
            return decode(sd, *a)
        raise NotImplementedError(mode)


def evaluate_seq(model, frames, init_masks, out_size, trace=None):
    """swem_evaluator.py:59-102: encode_key -> match -> segment -> argmax/one-hot ->
    [bilinear -> encode_value -> memorize] for every frame but the last."""
    b, t, _, h, w = frames.shape
    preds, scores = [], []
    mk16, _, s16, _, _ = model('encode_key', frames[:, 0])
    m0 = F.interpolate(init_masks[0], size=(h, w), mode='nearest')
    mv16 = model('encode_value', frames[:, 0], m0.float(), s16)
    model('init', mk16, mv16, init_masks[0])
    for i in range(1, t):
        qk16, qv16, s16, s8, s4 = model('encode_key', frames[:, i])
        context, n = model('match', qk16, qv16)
        logits, pred_mask = model('segment', n, context, s8, s4, None, out_size)
        scores.append(pred_mask.clone())
        pred = torch.argmax(pred_mask, dim=1, keepdim=True)
        hard = (pred.expand(-1, n + 1, -1, -1) ==
                torch.arange(n + 1).view(1, -1, 1, 1)).type_as(pred)
        if trace is not None:
            trace.append({'qk16': qk16, 'context': context, 'logits': logits})
        if i < t - 1:
            pm = F.interpolate(pred_mask, size=(h, w), mode='bilinear', align_corners=False)
            mv16 = model('encode_value', frames[:, i], pm, s16)
            model('memorize', qk16, mv16, hard, pm)
        preds.append(pred[:, 0])
    return preds, scores


def evaluate_ytvos_seq(model, frames, init_masks, out_size):
    """swem_evaluator.py:104-148 (new objects injected from init_masks[i], i > 0)."""
    b, t, _, h, w = frames.shape
    preds = []
    mk16, _, s16, _, _ = model('encode_key', frames[:, 0])
    m0 = F.interpolate(init_masks[0], size=(h, w), mode='nearest')
    mv16 = model('encode_value', frames[:, 0], m0.float(), s16)
    model('init', mk16, mv16, init_masks[0])
    for i in range(1, t):
        qk16, qv16, s16, s8, s4 = model('encode_key', frames[:, i])
        context, n = model('match', qk16, qv16)
        logits, pred_mask = model('segment', n, context, s8, s4, None, out_size)
        if init_masks[i] is not None:
            new_objects = init_masks[i][:, 1:].sum(dim=1, keepdim=True).expand_as(pred_mask)
            pred_mask[new_objects > 0] = 0
            pred_mask = torch.cat([pred_mask, init_masks[i][:, 1:]], dim=1)
            n = pred_mask.shape[1] - 1
        pred = torch.argmax(pred_mask, dim=1, keepdim=True)
        hard = (pred.expand(-1, n + 1, -1, -1) == torch.arange(n + 1).view(1, -1, 1, 1)).type_as(pred)
        if i < t - 1:
            pm = F.interpolate(pred_mask, size=(h, w), mode='bilinear', align_corners=False)
            mv16 = model('encode_value', frames[:, i], pm, s16)
            model('memorize', qk16, mv16, hard, pm)
        preds.append(pred[:, 0])
    return preds


def evaluate_seq_ms(model, frames, init_masks, out_size, scales=(480,), is_flip=False):
    """swem_evaluator.py:34-57 (multi-scale / flip averaging of the probability maps)."""
    final = [0 for _ in range(frames.shape[1] - 1)]
    for scale in scales:
        h, w = scale, int((scale / 480) * 864)
        fr = F.interpolate(frames[0], size=(h, w), mode='bicubic', align_corners=False).unsqueeze(0)
        _, scores = evaluate_seq(model, fr, init_masks, out_size)
        if is_flip:
            ff = torch.flip(fr, dims=[-1])
            fm = [torch.flip(m, dims=[-1]) for m in init_masks if m is not None]
            _, fs = evaluate_seq(model, ff, fm + [None] * (frames.shape[1] - len(fm)), out_size)
            scores = [(a + torch.flip(b, dims=[-1])) / 2 for a, b in zip(scores, fs)]
        final = [f + s_ / len(scales) for f, s_ in zip(final, scores)]
    return [torch.argmax(f, dim=1) for f in final]


# --------------------------------------------------------------------------- #
# Training step (methods/SWEM/swem_trainer.py:59-108, losses/, solver/solver.py)
# --------------------------------------------------------------------------- #
def bootstrapped_ce(scores, target, it, valid_obj, start_warm, end_warm, top_p):
    """losses/bce_losses.py:7-51.  scores (B,N+1,T,H,W) logits, target (B,T,H,W) int64, valid_obj (B,N+1) or None."""
    b, t, h, w = target.shape
    if valid_obj is not None:
        if it < start_warm:
            tot = 0.0
            for i in range(b):
                tot = tot + F.cross_entropy(scores[i][valid_obj[i] > 0.5].unsqueeze(0), target[i].unsqueeze(0))
            return tot / b, 1.0
        raw = torch.cat([F.cross_entropy(scores[i][valid_obj[i] > 0.5].unsqueeze(0), target[i].unsqueeze(0),
                                         reduction='none').view(1, t, -1) for i in range(b)], 0)
    else:
        if it < start_warm:
            return F.cross_entropy(scores, target), 1.0
        raw = F.cross_entropy(scores, target, reduction='none').view(b, t, -1)
    if it > end_warm:
        this_p = top_p
    else:
        this_p = top_p + (1 - top_p) * ((end_warm - it) / (end_warm - start_warm))
    loss, _ = torch.topk(raw, k=int(h * w * this_p), dim=-1, sorted=False)
    return loss.mean(), this_p


def mask_iou_loss(pred, label):
    """losses/bce_losses.py:109-141.  pred (B,N,H,W) probabilities, label (B,H,W) indices."""
    b, n = pred.shape[:2]
    target = torch.stack([(label == i).type(pred.dtype) for i in range(n)], 1)
    inter = torch.min(pred, target).sum(dim=(-1, -2))
    union = torch.max(pred, target).sum(dim=(-1, -2)) + 1e-6
    return 1.0 - torch.sum(inter / union) / (b * n)


def vos_loss(scores, target, it, valid_obj, loss_cfg):
    """losses/__init__.py:15-63 with NAME='boots_ce', AUX='iou' (configs/config.py:83-89)."""
    main, p = bootstrapped_ce(scores, target, it, valid_obj, loss_cfg['BS_PERIOD'][0], loss_cfg['BS_PERIOD'][1],
                              loss_cfg['BS_RATIO'])
    b, n, t, h, w = scores.shape
    if valid_obj is None:
        aux = mask_iou_loss(F.softmax(scores.transpose(1, 2), dim=2).reshape(b * t, n, h, w),
                            target.contiguous().view(b * t, h, w))
    else:
        aux = 0.0
        for i in range(b):
            cur = F.softmax(scores[i][valid_obj[i] > 0.5].transpose(0, 1), dim=1)
            aux = aux + mask_iou_loss(cur, target[i])
        aux = aux / b
    return {'total_loss': main + loss_cfg['AUX_RATIO'] * aux, 'main_loss': main, 'aux_loss': aux, 'p': p}


def train_forward(model, frames, init_mask, valid_obj):
    """swem_trainer.py:59-90: the clip loop with gradients (bases restart from random_init every step)."""
    b, t, _, h, w = frames.shape
    out_size = tuple(init_mask.shape[-2:])
    mk16, _, s16, _, _ = model('encode_key', frames[:, 0])
    mv16 = model('encode_value', frames[:, 0], init_mask.float(), s16)
    model('init', mk16, mv16, init_mask)
    logits_list, results = [], []
    for i in range(1, t):
        qk16, qv16, s16, s8, s4 = model('encode_key', frames[:, i])
        context, n = model('match', qk16, qv16)
        logits, pred_mask = model('segment', n, context, s8, s4, valid_obj, out_size)
        logits_list.append(logits)
        pred = torch.argmax(pred_mask, dim=1, keepdim=True)
        results.append(pred)
        hard = (pred.expand(-1, n + 1, -1, -1) == torch.arange(n + 1).view(1, -1, 1, 1)).type_as(pred)
        if i < t - 1:
            mv16 = model('encode_value', frames[:, i], pred_mask, s16)
            model('memorize', qk16, mv16, hard, pred_mask)
    return torch.stack(logits_list, dim=2), torch.cat(results, dim=1)


def adamw_step(params, grads, state, lr, weight_decay, step, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.AdamW (solver/solver.py:38-41) for one step, single-tensor form:
    p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    for k, p in params.items():
        g = grads[k]
        m, v = state.setdefault(k, (torch.zeros_like(p), torch.zeros_like(p)))
        p.mul_(1 - lr * weight_decay)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


def multistep_lr(base_lr, milestones, gamma, it):
    """optim.lr_scheduler.MultiStepLR (solver/solver.py:62-70): lr during iteration ``it`` (0-based)."""
    return base_lr * gamma ** sum(1 for m in milestones if it >= m)


def train_one_step(sd, cfg, frames, init_mask, valid_obj, label, cur_iter, loss_cfg):
    """swem_trainer.py:59-108 up to (not including) the optimizer step.  sd: name -> tensor; parameters that train
    carry requires_grad (frozen-BN running stats do not).  Returns losses, results, {name: grad}."""
    model = Model(sd, cfg)
    logits, results = train_forward(model, frames, init_mask, valid_obj)
    losses = vos_loss(logits, label[:, 1:], cur_iter, valid_obj, loss_cfg)
    names = [k for k, v in sd.items() if v.requires_grad]
    grads = torch.autograd.grad(losses['total_loss'], [sd[k] for k in names], allow_unused=True)
    return losses, results, {k: g for k, g in zip(names, grads)}, logits


def make_cfg(**kw):
    base = dict(KEYDIM=128, VALDIM=512, NUM_BASES=256, NUM_EM_ITERS=4, EM_TAU=0.05, TOPL=64,
                SINGLE_OBJ=False, BACKBONE='resnet50')
    base.update(kw)
    return SimpleNamespace(**base)
