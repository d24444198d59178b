"""Runs the SWEM encoders / decoder on the HIP kernels.

``Engine`` walks the parameter containers of ``swem_amd.networks`` once, packs every conv into the
kernel layout (OHWI weights, frozen BatchNorm folded into a per-filter scale/shift) and then executes
the reference's forward graphs (citations per method) as a sequence of C-ABI calls.  Activations are
NHWC fp32 tensors; nothing here computes with torch.
"""
import ctypes as C

import torch

from . import ops
from .networks import BasicBlock, Bottleneck


def _bn(m):
    return (m.weight, m.bias, m.running_mean, m.running_var)


class _Block:
    """One ResNet block (mod_resnet.py:58-113) as packed convs."""

    def __init__(self, blk):
        s = blk.stride
        self.bottleneck = isinstance(blk, Bottleneck)
        if self.bottleneck:
            self.c1 = ops.pack_conv(blk.conv1.weight, blk.conv1.bias, _bn(blk.bn1), 1, 0)
            self.c2 = ops.pack_conv(blk.conv2.weight, blk.conv2.bias, _bn(blk.bn2), s, 1)
            self.c3 = ops.pack_conv(blk.conv3.weight, blk.conv3.bias, _bn(blk.bn3), 1, 0)
        else:
            assert isinstance(blk, BasicBlock)
            self.c1 = ops.pack_conv(blk.conv1.weight, blk.conv1.bias, _bn(blk.bn1), s, 1)
            self.c2 = ops.pack_conv(blk.conv2.weight, blk.conv2.bias, _bn(blk.bn2), 1, 1)
        self.down = None
        if blk.downsample is not None:
            d = blk.downsample
            self.down = ops.pack_conv(d[0].weight, d[0].bias, _bn(d[1]), s, 0)
        # set by the Engine for every block but the last of its stage: the output is consumed by the next block only -- its
        # convolutions and its residual add (mod_resnet.py:77-113) -- and may be left as operand planes (ops.conv2d, 'block');
        # a stage's LAST output is handed to the caller (s4 / s8 / s16) and keeps its fp32 map
        self.inner = False

    def __call__(self, x):
        res = x if self.down is None else ops.conv2d([x], self.down)
        only = 'block' if self.inner else False
        # (conv1's and conv2's outputs have exactly one consumer, the next convolution: planes-only once it reads planes)
        if self.bottleneck and self.down is None and ops.bottleneck_ok(x, self.c1, self.c2, self.c3):
            return ops.bottleneck(x, self.c1, self.c2, self.c3, planes_only=only)     # (layer1's identity blocks: one launch)
        if self.bottleneck:
            y = ops.conv2d([x], self.c1, relu_out=True, planes_only=True)
            y = ops.conv2d([y], self.c2, relu_out=True, planes_only=True)
            return ops.conv2d([y], self.c3, relu_out=True, residual=res, planes_only=only)
        y = ops.conv2d([x], self.c1, relu_out=True, planes_only=True)
        return ops.conv2d([y], self.c2, relu_out=True, residual=res, planes_only=only)


class _ResBlock:
    """networks.py:12-32: r = conv2(relu(conv1(relu(x)))); out = downsample(x) + r."""

    def __init__(self, rb):
        self.c1 = ops.pack_conv(rb.conv1.weight, rb.conv1.bias)
        self.c2 = ops.pack_conv(rb.conv2.weight, rb.conv2.bias)
        self.down = None if rb.downsample is None else ops.pack_conv(rb.downsample.weight, rb.downsample.bias)

    def __call__(self, srcs, batch=None):
        r = ops.conv2d(srcs, self.c1, relu_in=True, batch=batch, planes_only=True)     # (only conv2 reads it)
        if self.down is None:
            # identity shortcut: with two sources the concatenated tensor itself is the residual
            res = srcs[0] if len(srcs) == 1 else ops.concat2(srcs[0], srcs[1], batch or srcs[0].shape[0])
        else:
            res = ops.conv2d(srcs, self.down, batch=batch)
        return ops.conv2d([r], self.c2, relu_in=True, residual=res)


class _SharedSourceSplit:
    """ResBlock(cat[x, f16]) of the value encoder's fusion block (networks.py:35-50, 113-129) with the part of its two 1280-channel
    convolutions that reads the CLIP's key feature f16 computed once per clip instead of once per object (round 6).  The cat's second
    source is the same map for the N objects of a clip (swem.py:52-53 .expand), a convolution is linear in its input channels and the
    block's input ReLU acts per element, so

        conv(relu(cat[x_n, f16])) = conv_x(relu(x_n)) + conv_s(relu(f16)),      likewise the downsample conv without the ReLU:

    conv_s (1,024 of the 1,280 input channels at config B) runs on B clips, conv_x on B * N objects with conv_s's result as its
    residual addend (one image for the whole batch when B = 1, else repeated per object: a 512-channel 1/16-scale map).  For N = 2
    that is 40 % fewer FLOPs in the block's two big launches, 53 % for N = 3; the results differ from the one-launch form by fp32
    summation order only (two partial sums added in fp32).  N = 1 keeps the one-launch form (nothing is shared)."""

    def __init__(self, rb, cx):
        w1, wd = rb.conv1.weight, rb.downsample.weight
        self.cx = cx
        self.c1x = ops.pack_conv(w1[:, :cx], rb.conv1.bias)
        self.c1s = ops.pack_conv(w1[:, cx:], None)
        self.dx = ops.pack_conv(wd[:, :cx], rb.downsample.bias)
        self.ds = ops.pack_conv(wd[:, cx:], None)
        # both shared halves as ONE convolution (conv1's filters, then the downsample's): valid where relu(f16) = f16
        self.cs = ops.pack_conv(torch.cat([w1[:, cx:], wd[:, cx:]], dim=0), None)
        self.cout = w1.shape[0]

    def __call__(self, x, shared, n, c2):
        """x (B*n,h,w,cx) per object, shared (B,h,w,cs) per clip, c2: the block's second conv pack."""
        if ops.SPLIT_SHARED_MERGE and shared.__dict__.get('_swem_nonneg') == shared._version:
            # f16 as THIS engine's key encoder wrote it ends in a ReLU (mod_resnet.py:108-113): relu(f16) = f16, so conv1's and the
            # downsample's shared halves read the same input and run as one launch with twice the output columns
            rs = ops.conv2d([shared], self.cs)
            rs1, rs2 = rs[..., :self.cout], rs[..., self.cout:]
            if shared.shape[0] > 1:
                rs1, rs2 = per_object(rs1, n), per_object(rs2, n)
            else:
                rs1, rs2 = rs1.contiguous(), rs2.contiguous()
        else:
            rs1 = ops.conv2d([shared], self.c1s, relu_in=True)
            rs2 = ops.conv2d([shared], self.ds)
            if shared.shape[0] > 1:
                rs1, rs2 = per_object(rs1, n), per_object(rs2, n)
        r = ops.conv2d([x], self.c1x, relu_in=True, residual=rs1, planes_only=True)     # (only conv2 reads it)
        res = ops.conv2d([x], self.dx, residual=rs2)
        return ops.conv2d([r], c2, relu_in=True, residual=res)


def per_object(t, n):
    """(B, ...) -> (B*n, ...): item b repeated for each of its n objects, clip-major like the reference's
    ``unsqueeze(1).expand(-1, n, ...).flatten(0, 1)`` (swem.py:52-53, 94-95).  The conv kernels broadcast ONE image over a
    batch (batch stride 0); with several clips in the batch the per-clip repeat is materialised (a device copy of a 1/16-
    or 1/4-scale map)."""
    return t.unsqueeze(1).expand(t.shape[0], n, *t.shape[1:]).reshape(t.shape[0] * n, *t.shape[1:]).contiguous()


class Engine:
    def __init__(self, model):
        dev = next(model.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('swem_amd runs on a HIP device only (model is on %s); there is no CPU path' % dev)
        self.device = dev
        self.single_obj = model.single_object
        ke, ve, dec = model.key_encoder, model.value_encoder, model.decoder
        with ops.pack_keys('engine'):       # layer names for the model's PlanBook: the same for every re-built Engine
            self._pack_all(model, ke, ve, dec)

    def _pack_all(self, model, ke, ve, dec):
        # --- key encoder (networks.py:132-170)
        self.k_mean, self.k_std = ops._f3(ke.mean), ops._f3(ke.std)
        self.k_stem = ops.pack_conv(ke.conv1.weight, None, _bn(ke.bn1), 2, 3, cin_pad=4)
        self.k_stem_s2d = ops.pack_stem_s2d(ke.conv1.weight, None, _bn(ke.bn1))
        self.k_stages = [[_Block(b) for b in st] for st in (ke.res2, ke.layer2, ke.layer3)]
        for st in self.k_stages:
            for b_ in st[:-1]:
                b_.inner = True
        self.key_proj = ops.pack_conv(model.key_proj.key_proj.weight, model.key_proj.key_proj.bias)
        self.key_comp = ops.pack_conv(model.key_comp.weight, model.key_comp.bias)
        # --- value encoder (networks.py:94-129)
        self.v_mean, self.v_std = ops._f3(ve.mean), ops._f3(ve.std)
        self.v_stem = ops.pack_conv(ve.conv1.weight, ve.conv1.bias, _bn(ve.bn1), 2, 3, cin_pad=8)
        self.v_stem_s2d = ops.pack_stem_s2d(ve.conv1.weight, ve.conv1.bias, _bn(ve.bn1))
        self.v_stages = [[_Block(b) for b in st] for st in (ve.layer1, ve.layer2, ve.layer3)]
        for st in self.v_stages:
            for b_ in st[:-1]:
                b_.inner = True
        self.fuse1 = _ResBlock(ve.fuser.block1)
        self.fuse2 = _ResBlock(ve.fuser.block2)
        att = ve.fuser.attention
        f32 = lambda t: t.detach().float().contiguous()
        self.cbam = [f32(att.ChannelGate.mlp[1].weight), f32(att.ChannelGate.mlp[1].bias),
                     f32(att.ChannelGate.mlp[3].weight), f32(att.ChannelGate.mlp[3].bias),
                     f32(att.SpatialGate.spatial.conv.weight), f32(att.SpatialGate.spatial.conv.bias)]
        # --- fusion layer (modules.py:13-26)
        fl = model.swem_core.fusion_layer
        self.glu = ops.pack_glu(fl.layer_f.weight, fl.layer_f.bias, fl.layer_a.weight, fl.layer_a.bias)
        # --- decoder (networks.py:199-216)
        self.compress = _ResBlock(dec.compress)
        self.skip8 = ops.pack_conv(dec.up_16_8.skip_conv.weight, dec.up_16_8.skip_conv.bias)
        self.out8 = _ResBlock(dec.up_16_8.out_conv)
        self.skip4 = ops.pack_conv(dec.up_8_4.skip_conv.weight, dec.up_8_4.skip_conv.bias)
        self.out4 = _ResBlock(dec.up_8_4.out_conv)
        self.pred_w = f32(dec.pred.weight.permute(0, 2, 3, 1))  # [1][3][3][C]
        self.pred_b = f32(dec.pred.bias)
        # --- (packed last: the layer names of everything above stay what they were) the fusion block's first ResBlock with the clip's
        # key feature split off its two convolutions (_SharedSourceSplit); needs the pre-split kernels' channel granularity
        b1 = ve.fuser.block1
        cx = b1.conv1.weight.shape[1] - ke.num_features[0] if hasattr(ke, 'num_features') else 0
        self.fuse1_split = None
        if ops.SPLIT_SHARED_SOURCE and b1.downsample is not None and cx > 0 and cx % 32 == 0 and (b1.conv1.weight.shape[1] - cx) % 32 == 0:
            self.fuse1_split = _SharedSourceSplit(b1, cx)

    # swem.py:39-43 + networks.py:160-182
    def encode_key(self, frames):
        if ops.S2D_STEMS and frames.shape[2] % 2 == 0 and frames.shape[3] % 2 == 0:
            x = ops.conv2d([ops.prep_input_s2d(frames, None, self.k_mean, self.k_std)], self.k_stem_s2d, relu_out=True)
        else:
            x = ops.conv2d([ops.prep_key_input(frames, self.k_mean, self.k_std)], self.k_stem, relu_out=True)
        x = ops.maxpool(x)
        feats = []
        for st in self.k_stages:
            for blk in st:
                x = blk(x)
            feats.append(x)
        s4, s8, s16 = feats
        if ops.SKIP_IN_KEY_PASS:
            # the decoder's skip convolutions depend on the frame alone (ops.SKIP_IN_KEY_PASS): computed here, handed on with s8 / s4
            for t_, pk in ((s8, self.skip8), (s4, self.skip4)):
                t_.__dict__['_swem_skip'] = (ops.conv2d([t_], pk), t_._version, self)
        qk16 = ops.conv2d([s16], self.key_proj)
        qv16 = ops.conv2d([s16], self.key_comp)
        s16.__dict__['_swem_nonneg'] = s16._version       # (the trunk's last block ends in a ReLU: _SharedSourceSplit)
        return qk16, qv16, s16, s8, s4

    # swem.py:45-62 + networks.py:113-129, 43-50
    def encode_value(self, frame, masks, s16):
        """frame NCHW (B,3,H,W), masks NCHW (B,N+1,H,W), s16 NHWC (B,h,w,Cs) -> NHWC (B*N,h,w,512)."""
        B = frame.shape[0]
        N = masks.shape[1] - 1
        split = self.fuse1_split if (N > 1 and ops.SPLIT_SHARED_SOURCE) else None
        if B != 1 and N != 1 and split is None:
            s16 = per_object(s16, N)      # clip b's feature map for each of its N objects (swem.py:52-53 .expand)
        if ops.S2D_STEMS and frame.shape[2] % 2 == 0 and frame.shape[3] % 2 == 0:
            x = ops.conv2d([ops.prep_input_s2d(frame, masks, self.v_mean, self.v_std, self.single_obj)], self.v_stem_s2d,
                           relu_out=True)
        else:
            x = ops.conv2d([ops.prep_value_input(frame, masks, self.v_mean, self.v_std, self.single_obj)], self.v_stem,
                           relu_out=True)
        x = ops.maxpool(x)
        for st in self.v_stages:
            for blk in st:
                x = blk(x)
        if split is not None:
            x = split(x, s16, N, self.fuse1.c2)            # the clip's f16 part of the block's two big convolutions once per clip
        else:
            x = self.fuse1([x, s16], batch=B * N)          # cat([x, f16]) never materialised
        x = ops.cbam_residual(x, *self.cbam)                # x + CBAM(x)
        return self.fuse2([x])

    # modules.py:286-291
    def fuse_context(self, mem_out, qv16, s_feat):
        BN = mem_out.shape[0]
        if qv16.shape[0] not in (1, BN):
            qv16 = per_object(qv16, BN // qv16.shape[0])     # (B,...) -> (B*N,...): modules.py:287 .expand_as
        return ops.conv2d([mem_out, qv16, s_feat], self.glu, batch=BN)

    def _skip(self, s, pack):
        """skip_conv(s) (networks.py:190-196): what THIS engine's encode_key computed for exactly this tensor, else computed now."""
        c = s.__dict__.get('_swem_skip')
        if c is not None and c[1] == s._version and c[2] is self and c[0].shape[:3] == s.shape[:3]:
            return c[0]
        return ops.conv2d([s], pack)

    # networks.py:208-213 ; the skip convs see the same s8/s4 for every object (swem.py:94-95): computed once
    def decoder_logit(self, context, s8, s4):
        BN, B = context.shape[0], s8.shape[0]
        x = self.compress([context])
        sk = self._skip(s8, self.skip8)
        x = self.out8([ops.upsample_add(sk, x)])          # (B skip images for B * N objects: image b // N, no per-object copy)
        sk = self._skip(s4, self.skip4)
        x = self.out4([ops.upsample_add(sk, x)])
        return ops.pred_head(x, self.pred_w, self.pred_b)
