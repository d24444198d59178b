"""GPU: the training-step kernels (include/swem_hip_train.h) against the oracle's autograd (oracle/swem_oracle.py,
pinned to the reference trainer by tests/golden/g9_*).  Loss / optimizer first, then every backward kernel with
identical inputs, then the whole step."""
import math

import numpy as np
import pytest
import torch

from oracle import swem_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, tol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = float((a - b).abs().max())
    scale = float(b.abs().max()) + 1e-30
    assert err <= tol * scale, '%s: max err %.3e (scale %.3e, rel %.3e > %.1e)' % (what, err, scale, err / scale, tol)


LOSS_CFG = dict(NAME='boots_ce', BS_RATIO=0.30, BS_PERIOD=[20, 70], AUX='iou', AUX_RATIO=1.0)


@pytest.mark.parametrize('it', [5, 45, 90])
@pytest.mark.parametrize('use_valid', [True, False])
def test_vos_loss_forward_backward(lib, it, use_valid):
    """losses/__init__.py:34-63: BootstrappedCE (plain CE below start_warm, annealed top-p inside, top-p after) + IoU
    auxiliary loss, values and d/d logits."""
    from swem_amd import losses
    g = torch.Generator().manual_seed(3 + it)
    B, N1, T, Hh, Ww = 2, 3, 2, 72, 80
    scores = (torch.randn(B, N1, T, Hh, Ww, generator=g) * 3).requires_grad_(True)
    valid = torch.tensor([[1., 1., 1.], [1., 1., 0.]]) if use_valid else None
    target = torch.randint(0, N1, (B, T, Hh, Ww), generator=g)
    if use_valid:
        target[1] = target[1].clamp(max=1)
    ref = O.vos_loss(scores, target, it, valid, LOSS_CFG)
    ref['total_loss'].backward()
    crit = losses.VOSLoss(LOSS_CFG, 100, DEV)
    frames = [scores.detach()[:, :, t].contiguous().to(DEV).requires_grad_(True) for t in range(T)]
    out = crit.clip_loss(frames, target.to(DEV), it, None if valid is None else valid.to(DEV))
    assert out['p'] == pytest.approx(ref['p'])
    for k in ('total_loss', 'main_loss', 'aux_loss'):
        assert float(out[k].detach()) == pytest.approx(float(ref[k].detach()), rel=2e-6), k
    out['total_loss'].backward()
    for t in range(T):
        close(frames[t].grad, scores.grad[:, :, t], 2e-5, 'dlogits frame %d' % t)
    # the reference-shaped call (stacked scores) gives the same numbers
    out2 = crit(scores.detach().to(DEV), target.to(DEV), it, None if valid is None else valid.to(DEV))
    assert float(out2['total_loss']) == pytest.approx(float(out['total_loss']), rel=1e-7)


def test_adamw_matches_torch_optimizer(lib):
    """solver/solver.py:38-41: three AdamW steps on a flat buffer vs torch.optim.AdamW on CPU."""
    from swem_amd import optim
    g = torch.Generator().manual_seed(9)
    n = 100003
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * (10.0 ** -i) for i in range(3)]
    pr = p0.clone().requires_grad_(True)
    ref = torch.optim.AdamW([pr], lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    opt = optim.FlatAdamW(p0.to(DEV), lr=2e-5, weight_decay=5e-4)
    for gr in grads:
        pr.grad = gr.clone()
        ref.step()
        opt.grad.copy_(gr.to(DEV))
        opt.step()
    # one fp32 ulp of a parameter of magnitude <= 4 is 2.4e-7; the three updates are ~6e-5 each
    assert float((opt.param.cpu() - pr.detach()).abs().max()) <= 2.5e-7
    close(opt.param.cpu() - p0, pr.detach() - p0, 5e-3, 'AdamW parameter update')
    assert opt.step_count == 3


# ----------------------------------------------------------------------------------------------------------------
# backward kernels, one stage at a time, against torch's CPU autograd of the same stage with identical inputs
# ----------------------------------------------------------------------------------------------------------------
import torch.nn.functional as F  # noqa: E402


def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).contiguous().to(DEV)


def back(t):
    return t.detach().permute(0, 3, 1, 2).cpu()


def leaf(t):
    return t.detach().clone().requires_grad_(True)


def param(t):
    p = torch.nn.Parameter(t.detach().clone().to(DEV))
    p.grad = torch.zeros_like(p)
    return p


@pytest.mark.parametrize('case', [
    dict(cins=[(64, 2)], cout=64, k=3, s=1, hw=(12, 20), relu_in=True, res=True),
    dict(cins=[(128, 2)], cout=128, k=3, s=2, hw=(17, 24), relu_in=False, res=False),
    dict(cins=[(64, 1)], cout=128, k=1, s=2, hw=(16, 15), relu_in=False, res=False, pad=0),
    dict(cins=[(64, 3), (128, 1)], cout=64, k=3, s=1, hw=(9, 11), relu_in=True, res=False),
    dict(cins=[(256, 2), (128, 1), (64, 2)], cout=96, k=3, s=1, hw=(8, 8), relu_in=False, res=False),
    dict(cins=[(5, 2)], cout=64, k=7, s=2, hw=(32, 40), relu_in=False, res=False, cin_pad=8, pad=3),
], ids=['3x3_relu_res', '3x3_s2_odd', '1x1_s2', 'two_src_broadcast', 'three_src', 'stem7x7'])
def test_conv_backward(lib, case):
    """Data gradient (SWEM_CONV_DGRAD on the forward kernels, input-ReLU mask, shared sources summed over the batch),
    weight gradient (swem_conv2d_wgrad_f32, accumulated into .grad) and bias gradient against F.conv2d's autograd."""
    from swem_amd import autograd as A
    A.new_step()
    g = torch.Generator().manual_seed(7)
    k, s = case['k'], case['s']
    pad = case.get('pad', k // 2)
    H, W = case['hw']
    B = max(b for _, b in case['cins'])
    xs = [leaf(torch.randn(b, c, H, W, generator=g)) for c, b in case['cins']]
    cin = sum(c for c, _ in case['cins'])
    w = leaf(torch.randn(case['cout'], cin, k, k, generator=g) * 0.05)
    bias = leaf(torch.randn(case['cout'], generator=g))
    xcat = torch.cat([x.expand(B, -1, -1, -1) for x in xs], 1)
    y = F.conv2d(F.relu(xcat) if case['relu_in'] else xcat, w, bias, stride=s, padding=pad)
    res = leaf(torch.randn(*y.shape, generator=g)) if case['res'] else None
    if res is not None:
        y = y + res
    dy = torch.randn(*y.shape, generator=g)
    y.backward(dy)
    # HIP
    cin_pad = case.get('cin_pad')
    srcs = []
    for x in xs:
        t = x.detach()
        if cin_pad:
            t = F.pad(t, (0, 0, 0, 0, 0, cin_pad - t.shape[1]))
        srcs.append(nhwc(t).requires_grad_(True))
    wp, bp = param(w), param(bias)
    rs = nhwc(res).requires_grad_(True) if res is not None else None
    out = A.conv2d(srcs, wp, bp, stride=s, pad=pad, relu_in=case['relu_in'], residual=rs, batch=B, cin_pad=cin_pad)
    close(back(out), y, 2e-5, 'conv forward')
    out.backward(nhwc(dy))
    close(wp.grad, w.grad, 3e-5, 'dW')
    close(bp.grad, bias.grad, 3e-5, 'dbias')
    for x, sx in zip(xs, srcs):
        gx = back(sx.grad)[:, :x.shape[1]]
        close(gx, x.grad, 3e-5, 'dX (%d ch, batch %d)' % (x.shape[1], x.shape[0]))
    if res is not None:
        close(back(rs.grad), res.grad, 1e-6, 'dres')
    # a second backward accumulates into .grad (the optimizer's flat buffer is zeroed once per step)
    out2 = A.conv2d([t_.detach().requires_grad_(True) for t_ in srcs], wp, bp, stride=s, pad=pad,
                    relu_in=case['relu_in'], residual=None, batch=B, cin_pad=cin_pad)
    out2.backward(nhwc(dy))
    close(wp.grad, 2 * w.grad, 3e-5, 'dW accumulated')


@pytest.mark.parametrize('relu,with_res', [(True, True), (True, False), (False, False)])
def test_bn_act_backward(lib, relu, with_res):
    """Frozen BatchNorm (+residual, ReLU) as a stage: dc, dres, dgamma, dbeta vs F.batch_norm(eval) autograd."""
    from swem_amd import autograd as A
    g = torch.Generator().manual_seed(8)
    B, Cc, H, W = 2, 64, 9, 13
    c = leaf(torch.randn(B, Cc, H, W, generator=g))
    gamma, beta = leaf(torch.rand(Cc, generator=g) + 0.5), leaf(torch.randn(Cc, generator=g))
    mean, var = torch.randn(Cc, generator=g), torch.rand(Cc, generator=g) + 0.5
    res = leaf(torch.randn(B, Cc, H, W, generator=g)) if with_res else None
    y = F.batch_norm(c, mean, var, gamma, beta, False, 0.0, 1e-5)
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    dy = torch.randn(*y.shape, generator=g)
    y.backward(dy)
    cx = nhwc(c).requires_grad_(True)
    rx = nhwc(res).requires_grad_(True) if with_res else None
    gp, bp = param(gamma), param(beta)
    out = A.bn_act(cx, (gp, bp, mean.to(DEV), var.to(DEV)), res=rx, relu=relu)
    close(back(out), y, 1e-5, 'bn_act forward')
    out.backward(nhwc(dy))
    close(back(cx.grad), c.grad, 1e-5, 'dc')
    close(gp.grad, gamma.grad, 3e-5, 'dgamma')
    close(bp.grad, beta.grad, 3e-5, 'dbeta')
    if with_res:
        close(back(rx.grad), res.grad, 1e-6, 'dres')


def test_bn_stages_write_the_operand_planes(lib):
    """When a convolution split a BatchNorm stage's output in an earlier step (ops.SPLIT_HINTS), the stage writes the bf16
    planes itself: they are BIT-identical to what swem_split_bf16x3_f32 makes of the same tensor (forward: y; backward: dc),
    and ops.presplit finds them on the tensor instead of launching the split."""
    from swem_amd import autograd as A, ops
    A.reset()
    g = torch.Generator().manual_seed(12)
    B, Cc, H, W = 2, 64, 9, 13
    mean, var = torch.randn(Cc, generator=g).to(DEV), (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    gp, bp = param(torch.rand(Cc, generator=g) + 0.5), param(torch.randn(Cc, generator=g))
    c = nhwc(torch.randn(B, Cc, H, W, generator=g))
    dy = nhwc(torch.randn(B, Cc, H, W, generator=g))

    def run():
        A.new_step()
        cx = c.clone().requires_grad_(True)
        y = A.bn_act(cx, (gp, bp, mean, var), relu=True)
        hinted = '_swem_split' in y.__dict__
        py = ops.presplit(y, False).clone()                       # the next layer's conv asks for the split of y
        seen = {}

        def hook(gr):                                              # the conv backward asks for the split of dc
            seen['dc'] = (gr.clone(), '_swem_split' in gr.__dict__, ops.presplit(gr, False).clone())
        cx.register_hook(hook)
        y.backward(dy)
        return hinted, py, seen['dc']
    h0, py0, (dc0, hd0, pd0) = run()                               # first step: the split kernel runs, the hints are recorded
    assert not h0 and not hd0 and (id(gp), 'y') in ops.SPLIT_HINTS and (id(gp), 'dc') in ops.SPLIT_HINTS
    h1, py1, (dc1, hd1, pd1) = run()                               # second step: the stages write the planes
    assert h1 and hd1
    assert torch.equal(py0.view(torch.int16), py1.view(torch.int16)) and torch.equal(dc0, dc1)
    assert torch.equal(pd0.view(torch.int16), pd1.view(torch.int16))
    A.reset()


def test_bn_stages_serve_an_f16x3_consumer(lib):
    """The same for a consumer on the f16x3 arithmetic (round 5): the forward stage writes the fp16 (hi, mid) pair of y -- bit
    for bit swem_split_f16x2_f32's -- and the backward stage hands the consumer of dc the block maxima of |dc|, from which the
    scaled split (swem_split_f16x2_scaled_f32, first pass skipped) makes the SAME planes and the same 2^-s as with its own pass."""
    from swem_amd import autograd as A, ops
    A.reset()
    g = torch.Generator().manual_seed(12)
    B, Cc, H, W = 2, 64, 9, 13
    mean, var = torch.randn(Cc, generator=g).to(DEV), (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    gp, bp = param(torch.rand(Cc, generator=g) + 0.5), param(torch.randn(Cc, generator=g))
    c = nhwc(torch.randn(B, Cc, H, W, generator=g))
    dy = nhwc(torch.randn(B, Cc, H, W, generator=g) * 3.0e-5)

    def run():
        A.new_step()
        cx = c.clone().requires_grad_(True)
        y = A.bn_act(cx, (gp, bp, mean, var), relu=True)
        hinted = '_swem_split' in y.__dict__
        py = ops.presplit(y, False, ops.PLANES_F16).clone()
        seen = {}

        def hook(gr):
            gr.__dict__['_swem_grad'] = True                       # (what _Conv.backward does with the gradient it is handed)
            have = '_swem_amax' in gr.__dict__
            pl = ops.presplit(gr, False, ops.PLANES_F16).clone()
            seen['dc'] = (gr.clone(), have, pl, float(gr.__dict__['_swem_inv'][0]))
        cx.register_hook(hook)
        y.backward(dy)
        return hinted, py, seen['dc']
    h0, py0, (dc0, hd0, pd0, inv0) = run()
    assert not h0 and not hd0
    h1, py1, (dc1, hd1, pd1, inv1) = run()
    assert h1 and hd1, 'the stages did not serve the fp16 consumer'
    assert torch.equal(py0.view(torch.int16), py1.view(torch.int16)) and torch.equal(dc0, dc1)
    assert inv0 == inv1 and 2.0 ** -14 < float(dc0.abs().max()) / inv0 <= 2.0 ** 14 and float(dc0.abs().max()) / inv0 >= 2.0 ** 13
    assert torch.equal(pd0.view(torch.int16), pd1.view(torch.int16))
    ops.check_faults()
    A.reset()


def test_maxpool_upsample_glu_backward(lib):
    from swem_amd import autograd as A
    g = torch.Generator().manual_seed(9)
    x = leaf(torch.randn(2, 64, 15, 18, generator=g))
    y = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(*y.shape, generator=g)
    y.backward(dy)
    hx = nhwc(x).requires_grad_(True)
    out = A.maxpool(hx)
    assert torch.equal(back(out), y)
    out.backward(nhwc(dy))
    close(back(hx.grad), x.grad, 1e-6, 'maxpool dx')
    # ties: a constant map sends each window's gradient to its first element, like ATen
    xc = leaf(torch.zeros(1, 4, 6, 6))
    F.max_pool2d(xc, 3, 2, 1).sum().backward()
    hc = nhwc(xc).requires_grad_(True)
    A.maxpool(hc).sum().backward()
    assert torch.equal(back(hc.grad), xc.grad)
    # round 6: the product's backward reads the forward output (swem_maxpool3x3s2_bwd_y_f32); the window-scanning form of rounds 1-5
    # stays in the ABI -- both give the same gradient bit for bit, also on a map FULL of ties (values quantised to four levels)
    from swem_amd import _lib as L, ops
    for quant in (False, True):
        xt = torch.randn(3, 17, 22, 64, generator=g)
        xt = (xt * 1.5).round().clamp(-2, 1) if quant else xt
        xt = xt.to(DEV).contiguous()
        yt = ops.maxpool(xt)
        dyt = torch.randn(*yt.shape, generator=g).to(DEV)
        d_old, d_new = torch.empty_like(xt), torch.empty_like(xt)
        L.call('swem_maxpool3x3s2_bwd_f32', ops._stream(), xt.data_ptr(), dyt.data_ptr(), d_old.data_ptr(), 3, 17, 22, 64)
        L.call('swem_maxpool3x3s2_bwd_y_f32', ops._stream(), xt.data_ptr(), yt.data_ptr(), dyt.data_ptr(), d_new.data_ptr(), 3, 17, 22, 64)
        assert torch.equal(d_old, d_new), quant
        xr = leaf(back(xt))
        F.max_pool2d(xr, 3, 2, 1).backward(back(dyt))
        close(back(d_new), xr.grad, 1e-6, 'maxpool dx vs ATen, ties %s' % quant)
    # skip (shared by the 3 objects) + bilinear x2 of the low map
    skip, low = leaf(torch.randn(1, 32, 12, 16, generator=g)), leaf(torch.randn(3, 32, 6, 8, generator=g))
    y = skip + F.interpolate(low, size=(12, 16), mode='bilinear', align_corners=False)
    dy = torch.randn(*y.shape, generator=g)
    y.backward(dy)
    hs, hl = nhwc(skip).requires_grad_(True), nhwc(low).requires_grad_(True)
    out = A.upsample_add(hs, hl, batch=3)
    close(back(out), y, 1e-6, 'upsample_add forward')
    out.backward(nhwc(dy))
    close(back(hl.grad), low.grad, 2e-6, 'dlow')
    close(back(hs.grad), skip.grad, 2e-6, 'dskip (summed over objects)')
    # odd -> even sizes (7 -> 13 is not x2: the adjoint is generic)
    low2 = leaf(torch.randn(1, 8, 7, 5, generator=g))
    sk2 = leaf(torch.randn(1, 8, 13, 10, generator=g))
    y = sk2 + F.interpolate(low2, size=(13, 10), mode='bilinear', align_corners=False)
    dy = torch.randn(*y.shape, generator=g)
    y.backward(dy)
    hl2 = nhwc(low2).requires_grad_(True)
    A.upsample_add(nhwc(sk2), hl2).backward(nhwc(dy))
    close(back(hl2.grad), low2.grad, 2e-6, 'dlow generic scale')
    # GLU gate
    f, a = leaf(torch.randn(2, 32, 5, 7, generator=g)), leaf(torch.randn(2, 32, 5, 7, generator=g))
    y = f * torch.sigmoid(a)
    dy = torch.randn(*y.shape, generator=g)
    y.backward(dy)
    hf, ha = nhwc(f).requires_grad_(True), nhwc(a).requires_grad_(True)
    out = A.glu(hf, ha)
    close(back(out), y, 1e-6, 'glu forward')
    out.backward(nhwc(dy))
    close(back(hf.grad), f.grad, 2e-6, 'glu df')
    close(back(ha.grad), a.grad, 2e-6, 'glu da')


def test_cbam_backward(lib):
    """x + CBAM(x): dx and the six parameter gradients vs the oracle's cbam (attentions.py:22-84) under autograd."""
    from swem_amd import autograd as A
    g = torch.Generator().manual_seed(10)
    B, Cc, H, W, hid = 2, 64, 6, 9, 4
    x = leaf(torch.randn(B, Cc, H, W, generator=g))
    names = ['ChannelGate.mlp.1.weight', 'ChannelGate.mlp.1.bias', 'ChannelGate.mlp.3.weight', 'ChannelGate.mlp.3.bias',
             'SpatialGate.spatial.conv.weight', 'SpatialGate.spatial.conv.bias']
    shapes = [(hid, Cc), (hid,), (Cc, hid), (Cc,), (1, 2, 7, 7), (1,)]
    sd = {'a.' + n: leaf(torch.randn(*s, generator=g) * 0.3) for n, s in zip(names, shapes)}
    y = x + O.cbam(sd, 'a', x)
    dy = torch.randn(*y.shape, generator=g)
    y.backward(dy)
    hx = nhwc(x).requires_grad_(True)
    ps = [param(sd['a.' + n]) for n in names]
    out = A.cbam_residual(hx, *ps)
    close(back(out), y, 1e-5, 'cbam forward')
    out.backward(nhwc(dy))
    close(back(hx.grad), x.grad, 5e-5, 'cbam dx')
    for n, p_ in zip(names, ps):
        close(p_.grad, sd['a.' + n].grad, 1e-4, 'cbam d' + n)


def test_heads_backward(lib):
    """pred head, decode head (bilinear -> sigmoid -> valid -> aggregate -> softmax; gradient through logits AND the
    probabilities), value-encoder input packing."""
    from swem_amd import autograd as A
    A.new_step()
    g = torch.Generator().manual_seed(11)
    N, Cc, h4, w4 = 2, 64, 10, 12
    x = leaf(torch.randn(N, Cc, h4, w4, generator=g))
    w, b = leaf(torch.randn(1, Cc, 3, 3, generator=g) * 0.1), leaf(torch.randn(1, generator=g))
    valid = torch.tensor([[1., 1., 0.]])
    lg4 = F.conv2d(F.relu(x), w, b, padding=1)                                     # (N,1,h4,w4)
    Ho, Wo = 40, 48
    up = F.interpolate(lg4, size=(Ho, Wo), mode='bilinear', align_corners=False)
    preds = torch.sigmoid(up).view(1, N, Ho, Wo) * valid[:, 1:, None, None]
    logits = O.aggregate(preds)
    prob = F.softmax(logits, dim=1)
    dl, dp = torch.randn(1, N + 1, Ho, Wo, generator=g), torch.randn(1, N + 1, Ho, Wo, generator=g)
    ((logits * dl).sum() + (prob * dp).sum()).backward()
    hx = nhwc(x).requires_grad_(True)
    wp, bp = param(w), param(b)
    l4 = A.pred_head(hx, wp, bp)
    hl, hp = A.decode_head(l4, valid.to(DEV), 1, N, (Ho, Wo))
    close(hl.cpu(), logits, 1e-4, 'logits')
    ((hl * dl.to(DEV)).sum() + (hp * dp.to(DEV)).sum()).backward()
    close(back(hx.grad), x.grad, 1e-4, 'pred/decode dx')
    close(wp.grad, w.grad, 1e-4, 'pred dw')
    close(bp.grad, b.grad, 1e-4, 'pred db')
    # value-encoder input: channels [img, m, 1 - m - m_bg]
    frame = torch.rand(1, 3, 16, 20, generator=g)
    masks = leaf(torch.rand(1, 3, 16, 20, generator=g))
    others = 1 - masks - masks[:, 0:1]
    f = torch.cat([masks[:, 1:].flatten(0, 1).unsqueeze(1), others[:, 1:].flatten(0, 1).unsqueeze(1)], 1)   # (N,2,H,W)
    dy = torch.randn(2, 8, 16, 20, generator=g)
    (f * dy[:, 3:5]).sum().backward()
    hm = masks.detach().to(DEV).requires_grad_(True)
    import ctypes as C
    z3, o3 = (C.c_float * 3)(0, 0, 0), (C.c_float * 3)(1, 1, 1)
    out = A.prep_value_input(frame.to(DEV), hm, z3, o3, False)
    out.backward(nhwc(dy))
    close(hm.grad.cpu(), masks.grad, 1e-6, 'd masks')


@pytest.mark.parametrize('L,banks,topl,N', [(64, 1, 32, 2), (64, 2, 64, 2), (128, 2, 64, 2), (64, 2, 32, 5), (256, 2, 64, 2)])
def test_match_backward(lib, L, banks, topl, N):
    """get_affinity + perm_inv_feat: d qk (through the l2norm, the joint softmax and the top-l prefix features) and
    d nu for both banks, against the oracle under autograd."""
    from swem_amd import autograd as A
    g = torch.Generator().manual_seed(12 + L + banks)
    Cc, V, h, w = 128, 64, 6, 9
    P = h * w
    xk, _ = H.structured_keys(P, Cc, 5, g)
    qk = leaf(xk.t().reshape(1, Cc, h, w).contiguous())
    kap = [O.l2norm(torch.randn(1, N, 2, Cc, L, generator=g) * 0.3 + xk[torch.randint(0, P, (L,), generator=g)].t(), -2)
           for _ in range(banks)]
    nus = [leaf(torch.randn(1, N, 2, V, L, generator=g)) for _ in range(banks)]
    mk, mv = torch.cat(kap, -1), torch.cat(nus, -1)
    S, mem = O.get_affinity(O.l2norm(qk, 1), O.l2norm(mk, -2), mv, 0.05, topl)
    dS, dmem = torch.randn(*S.shape, generator=g), torch.randn(*mem.shape, generator=g)
    ((S * dS).sum() + (mem * dmem).sum()).backward()
    hq = qk.detach().permute(0, 2, 3, 1).reshape(P, Cc).contiguous().to(DEV).requires_grad_(True)
    hn = [n_.detach()[0].contiguous().to(DEV).requires_grad_(True) for n_ in nus]
    hk = [k_[0].contiguous().to(DEV) for k_ in kap]
    m_, s_ = A.match(hq, hn[0], hn[1] if banks == 2 else None, hk[0], hk[1] if banks == 2 else None, topl, 0.05)
    close(s_.cpu(), S.view(N, 2 * topl, P).permute(0, 2, 1), 1e-4, 'S')
    close(m_[:, :P].cpu(), mem[0].flatten(2).permute(0, 2, 1), 1e-4, 'mem_out')
    dm = torch.zeros_like(m_)
    dm[:, :P] = dmem[0].flatten(2).permute(0, 2, 1).to(DEV)
    ds = dS.view(N, 2 * topl, P).permute(0, 2, 1).contiguous().to(DEV)
    ((m_ * dm).sum() + (s_ * ds).sum()).backward()
    # d qk passes through p (g - sum p g) / tau with strong cancellation: judge it against an fp64 evaluation, with
    # the fp32 CPU oracle's own distance to fp64 as the yardstick
    q64 = qk.detach().double().requires_grad_(True)
    S64, mem64 = O.get_affinity(O.l2norm(q64, 1), O.l2norm(mk.double(), -2), mv.detach().double(), 0.05, topl)
    ((S64 * dS.double()).sum() + (mem64 * dmem.double()).sum()).backward()
    ref64 = q64.grad[0].flatten(1).t()
    # Ties are discontinuities of this gradient: d c_i goes to the element of RANK i, so where two of a (pixel, object,
    # class)'s top-l (+1: the cut) values are equal to fp32 rounding, two correct evaluations order them differently (torch.topk
    # picks one order, the kernel gives tied elements the same rank: include/swem_hip_train.h) and d qk of that PIXEL changes
    # by O(d c_i).  Measure zero in exact arithmetic, but a fixed seed can sit on one: pixels whose sorted affinities come
    # closer than 5e-6 exponent units (~4 fp32 ulps of the logit) are left out of the comparison.
    Lm = mk.shape[-1]
    with torch.no_grad():
        qn = O.l2norm(q64.detach(), 1).flatten(2)[:, None, None]
        aff = torch.matmul(O.l2norm(mk.double(), -2).transpose(-2, -1), qn)                  # 1,N,2,Lm,P
        srt = aff.sort(dim=3, descending=True)[0]
        kk = min(topl, Lm - 1)
        gap = ((srt[:, :, :, :kk] - srt[:, :, :, 1:kk + 1]) / 0.05).amin(dim=3)                  # 1,N,2,P
        keep = (gap > 5e-6).all(dim=1).all(dim=1)[0]
    print('pixels compared: %d of %d' % (int(keep.sum()), P))
    assert float(keep.float().mean()) > 0.5
    ref64 = ref64[keep]
    scale = float(ref64.abs().max())
    err_cpu = float((qk.grad[0].flatten(1).t().double()[keep] - ref64).abs().max())
    err_hip = float((hq.grad.cpu().double()[keep] - ref64).abs().max())
    assert err_hip <= max(3 * err_cpu, 2e-4 * scale), 'd qk: HIP %.3e vs fp64, CPU fp32 %.3e, scale %.3e' % (
        err_hip, err_cpu, scale)
    for i in range(banks):
        close(hn[i].grad.cpu(), nus[i].grad[0], 1e-4, 'd nu bank %d' % i)


def test_memorize_backward(lib):
    """swem(): bases as the inference kernel gives them, and the value update's gradient (d v, d nu_prev)."""
    from swem_amd import autograd as A
    g = torch.Generator().manual_seed(13)
    N, Cc, V, h, w, L, T = 2, 128, 128, 6, 9, 64, 4
    P = h * w
    x, v, m = H.em_inputs(h, w, Cc, V, N, g)
    v = leaf(v)
    prior = {'kappa': O.l2norm(torch.randn(1, N, 2, Cc, L, generator=g), -2),
             'nu': leaf(torch.randn(1, N, 2, V, L, generator=g)), 'zita': torch.rand(1, N, 2, 1, L, generator=g) * 3 + 1e-6}
    ref = O.swem(x, v, m, prior, L, T, 0.05, V)
    dnu = torch.randn(*ref['nu'].shape, generator=g)
    (ref['nu'] * dnu).sum().backward()
    hv = v.detach()[0].flatten(2).permute(0, 2, 1).contiguous().to(DEV).requires_grad_(True)       # (N,P,V)
    hnu = prior['nu'].detach()[0].contiguous().to(DEV).requires_grad_(True)
    hx = x[0].flatten(1).t().contiguous().to(DEV)
    kap, nu, zita = A.memorize(hv, hnu, hx, m[0].flatten(2).contiguous().to(DEV), prior['kappa'][0].contiguous().to(DEV),
                               prior['zita'][0, :, :, 0].contiguous().to(DEV), T, 0.05)
    # the forward against float64 (T iterations amplify rounding: two fp32 evaluations of this input -- the reference with
    # its key channels re-ordered -- differ from each other by up to 1.8e-4 and from float64 by 2e-5 .. 1e-4): the HIP
    # result may be as far from float64 as twice the reference's own fp32 arithmetic is
    with torch.no_grad():
        r64 = O.swem(x.double(), v.detach().double(), m.double(), {k: t.detach().double() for k, t in prior.items()}, L, T,
                     0.05, V)
    mass = r64['zita'][0]

    def err64(nu_, zita_):
        return (float(((nu_.double() - r64['nu'][0]) * mass).abs().max() / (r64['nu'][0] * mass).abs().max()),
                float((zita_.double() - r64['zita'][0, :, :, 0]).abs().max() / r64['zita'].abs().max()))
    floor = err64(ref['nu'][0].detach(), ref['zita'][0, :, :, 0])
    got = err64(nu.detach().cpu(), zita.cpu())
    print('memorize vs float64: hip nu*zita %.3g zita %.3g   reference fp32 %.3g %.3g' % (got + floor))
    assert got[0] <= max(1e-4, 2 * floor[0]) and got[1] <= max(1e-4, 2 * floor[1]), (got, floor)
    (nu * dnu[0].to(DEV)).sum().backward()
    # identical z would make these exact; the fp32 EM differs by rounding, so compare mass-weighted like the forward
    close(hnu.grad.cpu(), prior['nu'].grad[0], 1e-3, 'd nu_prev')
    close(hv.grad.cpu(), v.grad[0].flatten(2).permute(0, 2, 1), 1e-3, 'd v')


@pytest.mark.parametrize('tag,it', [('r18', 5), ('r18', 45), ('r50', 45), ('r50k256', 45), ('r50k256n5', 45)])
def test_one_step_matches_reference_trainer(golden, lib, tag, it):
    """a18: SWEMTrainer.one_step on HIP against the losses, index maps, per-parameter gradient norms and the AdamW
    update recorded from the REFERENCE trainer (tests/golden/make_golden_train.py)."""
    _one_step_vs_reference(golden, tag, it, None)


@pytest.mark.parametrize('tag,it', [('r18', 45), ('r50', 45), ('r50k256', 45), ('r50k256n5', 45)])
def test_one_step_f16x3_matches_reference_trainer(golden, lib, tag, it):
    """The same step with EVERY convolution the pre-split kernel can take in the f16x3 arithmetic (round 5: forward, data gradient
    on the device-scaled fp16 pair of dY, weight gradient swem_conv2d_wgrad_f16x3) -- held to the same bars against the reference
    trainer's record as the fp32 / bf16x6 step above (the tuner mixes the three per layer; this is the all-f16x3 corner)."""
    from swem_amd import ops
    ran = {}
    with ops.flags(MATH_RAN=ran):
        _one_step_vs_reference(golden, tag, it, (7,))
    assert ran.get(7, 0) > 50 and ran.get(7, 0) > 5 * ran.get(0, 0), ran        # (the stems and the heads stay on the fp32 kernels)
    ops.check_faults()


@pytest.mark.parametrize('side', [False, True], ids=['one_stream', 'wgrad_side_stream'])
@pytest.mark.parametrize('modes', [None, (7,)], ids=['fp32', 'f16x3'])
def test_one_step_clip_batched_matches_reference_trainer(golden, lib, modes, side):
    """Round 6: the clips of a lane go through the conv stack as ONE batch, as the reference's step does (swem_trainer.py:60-90:
    `frames[:, i]` is (B,3,H,W)).  The two-clip fixture of the REFERENCE trainer (different valid_obj per clip) with one lane =
    both clips in one pass -- key encoder on 6 frames, value encoder / decoder on 4 objects, EM and matching clip by clip inside
    their stages, the shared maps laid out per object (autograd.expand_objects) -- held to the same bars as the lane-per-clip step;
    `wgrad_side_stream`: the weight gradients on the lane's second stream beside the data-gradient chain."""
    _one_step_vs_reference(golden, 'r18', 45, modes, lanes=1, wgrad_stream=side, record='_batched%s' % ('_side' if side else ''))


def test_clip_batched_step_equals_lane_per_clip_step(lib):
    """The same two clips stepped as one batch (one lane) and as a clip per lane: the forward is the same arithmetic image by image
    (losses and index maps equal to rounding), the parameter gradients differ only by the order of their sums over the clips."""
    from swem_amd import train
    from swem_amd.train import SWEMTrainer
    tc = H.train_cases()
    case = dict(tc['cases']['r18'], hw=[128, 128])
    cfg = O.make_cfg(**case['cfg'])
    frames, init_mask, label, valid = [t.to(DEV) for t in H.train_batch(case)]
    torch.manual_seed(4)
    fixed = train.random_init_host(2, case['n'], 128, cfg.NUM_BASES)
    real = train.random_init_host
    train.random_init_host = lambda B, N, Cc, Lb: fixed.clone()
    out = {}
    try:
        for lanes, side in ((2, False), (1, False), (1, True)):
            model, _ = H.make_model_and_sd(cfg, case['wseed'], DEV, pred_scale=tc['pred_scale'])
            tr = SWEMTrainer(dict(SOLVER=tc['solver_cfg'], LOSS=tc['loss_cfg'], AMP=False), model, use_graph=False, lanes=lanes,
                             wgrad_stream=side)
            losses, results = tr.one_step(frames, init_mask, valid, label, 45)
            assert [b1 - b0 for b0, b1 in tr._lane_state['chunks']] == ([1, 1] if lanes == 2 else [2])
            out[(lanes, side)] = ([float(losses[k]) for k in ('total_loss', 'main_loss', 'aux_loss')], results.clone(),
                                  tr.optimizer.grad.clone())
    finally:
        train.random_init_host = real
    ref = out[(2, False)]
    for key in ((1, False), (1, True)):
        got = out[key]
        assert got[0] == pytest.approx(ref[0], rel=1e-5), key
        # (the fixtures' shrunken prediction head leaves the classes' probabilities close: the argmax of a few pixels is rounding)
        assert float((got[1] == ref[1]).float().mean()) >= 0.999, key
        rel = float((got[2] - ref[2]).norm() / ref[2].norm())
        print('clip-batched %s vs lane-per-clip: losses %s vs %s, gradient rel diff %.3g' % (key, got[0], ref[0], rel))
        assert rel < 2e-4, (key, rel)
    # the side stream changes no arithmetic: bit for bit the one-stream batched step
    assert torch.equal(out[(1, True)][2], out[(1, False)][2]) and out[(1, True)][0] == out[(1, False)][0]


def _one_step_vs_reference(golden, tag, it, modes, lanes=None, wgrad_stream=None, record=''):
    from swem_amd.train import SWEMTrainer
    tc = H.train_cases()
    case = tc['cases'][tag]
    fx = golden('g9_train_%s_it%d.npz' % (tag, it))
    cfg = O.make_cfg(**case['cfg'])
    model, sd = H.make_model_and_sd(cfg, case['wseed'], DEV, pred_scale=tc['pred_scale'])
    trainer = SWEMTrainer(dict(SOLVER=tc['solver_cfg'], LOSS=tc['loss_cfg'], AMP=False), model, lanes=lanes, wgrad_stream=wgrad_stream)
    trainer.math_modes = modes
    frames, init_mask, label, valid = H.train_batch(case)
    assert H.checksum(frames) == pytest.approx(float(fx['frames_sum']), rel=1e-12)
    torch.manual_seed(91)
    losses, results = trainer.one_step(frames.to(DEV), init_mask.to(DEV), valid.to(DEV), label.to(DEV), it)
    got = {k: float(losses[k]) for k in ('total_loss', 'main_loss', 'aux_loss')}
    print(tag, it, 'losses', got, 'ref', {k: float(fx[k]) for k in got})
    agree = float((results.cpu().to(torch.uint8) == fx['results']).float().mean())
    print('   index maps agree %.5f' % agree)
    # yardstick: the reference's own fp32-vs-fp64 distance on this step (floor64_*, measured when the fixture was made):
    # the EM amplifies rounding, so two correct fp32 implementations differ by about that much
    names = fx['grad_names']
    params = dict(model.named_parameters())
    floors = fx['floor64_norm'].tolist()
    rels, bad = [], []
    for n, ref_norm, fl in zip(names, fx['grad_norms'].tolist(), floors):
        gn = float(params[n].grad.double().norm())
        rel = abs(gn - ref_norm) / (ref_norm + 1e-12)
        rels.append(rel)
        if rel > max(5e-3, 5 * fl):
            bad.append((n, rel, fl))
    srt = sorted(rels)
    print('   grad norms: median rel err %.2e (floor %.2e), 90%% %.2e, worst %.2e (floor max %.2e); outside the bound: %s'
          % (srt[len(srt) // 2], sorted(floors)[len(floors) // 2], srt[int(len(srt) * 0.9)], srt[-1], max(floors), bad))
    errs = []
    for key, fxk in (('decoder.pred.weight', 'g_pred_weight'), ('key_proj.key_proj.bias', 'g_key_proj_bias'),
                     ('value_encoder.conv1.weight', 'g_v_conv1_weight')):
        errs.append(float((params[key].grad.cpu() - fx[fxk]).abs().max() / fx[fxk].abs().max()))
    fe = fx['floor64_elem'].tolist()
    print('   elementwise rel err: pred.weight %.2e (floor %.1e), key_proj.bias %.2e (%.1e), value conv1.weight %.2e (%.1e)'
          % (errs[0], fe[0], errs[1], fe[1], errs[2], fe[2]))
    H.record_parity('train_step_%s_it%d%s%s' % (tag, it, '' if modes is None else '_f16x3', record), {
        'losses': got, 'reference_losses': {k: float(fx[k]) for k in got}, 'reference_fp32_vs_fp64_loss_floor': float(fx['floor64_loss']),
        'index_agreement': agree, 'reference_fp32_vs_fp64_agreement': float(fx['agree64']),
        'grad_norm_rel_err': {'median': srt[len(srt) // 2], 'p90': srt[int(len(srt) * 0.9)], 'worst': srt[-1]},
        'reference_fp32_vs_fp64_grad_norm_floor': {'median': sorted(floors)[len(floors) // 2], 'max': max(floors)},
        'elementwise_rel_err': dict(zip(('decoder.pred.weight', 'key_proj.key_proj.bias', 'value_encoder.conv1.weight'), errs)),
        'parameters_compared': len(names)})
    for k in got:
        assert got[k] == pytest.approx(float(fx[k]), rel=max(1e-4, 5 * float(fx['floor64_loss']))), k
    assert losses['p'] == pytest.approx(float(fx['p']))
    assert agree >= min(0.9995, 1 - 3 * (1 - float(fx['agree64'])))
    assert not bad, bad
    assert srt[len(srt) // 2] < max(2e-4, 5 * sorted(floors)[len(floors) // 2])
    for e, f in zip(errs, fe):
        assert e < max(2e-3, 5 * f)
    # optimizer: the parameter moved exactly like the reference's (lr 2e-5: the update is lr-sized whatever the gradient)
    dw = (params['decoder.pred.weight'].detach().cpu() - fx['w_after_pred_weight']).abs().max()
    assert float(dw) < 2e-6
    for n in fx['unused']:
        assert float(params[n].grad.abs().max()) == 0.0, n


def test_graph_replay_equals_eager_steps(lib):
    """The captured step (HIP graph of all clips' forward/backward, replayed on static buffers with the bootstrap k, the
    random bases and the loss weight entering through device memory) gives the same losses and the same parameters as
    eager steps, iteration after iteration (the kernels are deterministic: no atomics)."""
    from swem_amd.train import SWEMTrainer
    tc = H.train_cases()
    case = dict(tc['cases']['r18'], hw=[128, 128])
    cfg = O.make_cfg(**case['cfg'])
    frames, init_mask, label, valid = [t.to(DEV) for t in H.train_batch(case)]
    out = {}
    for mode in (False, True):
        model, _ = H.make_model_and_sd(cfg, case['wseed'], DEV, pred_scale=tc['pred_scale'])
        solver = dict(tc['solver_cfg'], BASE_LR=1e-4)        # a visible update per step
        tr = SWEMTrainer(dict(SOLVER=solver, LOSS=tc['loss_cfg'], AMP=False), model, use_graph=mode)
        torch.manual_seed(5)
        hist = []
        for it in (18, 19, 20, 45, 71, 5):                   # plain CE, annealed top-k, final top-k, plain again
            losses, results = tr.one_step(frames, init_mask, valid, label, it)
            hist.append((float(losses['total_loss']), float(losses['main_loss']), losses['p'], int(results.sum())))
        assert (tr._graph is not None) == mode
        out[mode] = (hist, tr.optimizer.param.detach().clone())
    for a, b in zip(out[False][0], out[True][0]):
        assert a == b, (a, b)
    assert torch.equal(out[False][1], out[True][1])
    assert len({h[0] for h in out[True][0]}) == 6 and out[True][0][-1][3] > 0   # the loss moves, the masks are alive


def test_one_step_single_object_vs_oracle(lib):
    """Stage-0 style step (SINGLE_OBJ model, one object, valid_obj=None, plain CE below start_warm) against the oracle's
    autograd run in the test itself: exercises ValueEncoderSO, the 4-channel stem and the valid_obj=None branches."""
    from swem_amd.train import SWEMTrainer
    tc = H.train_cases()
    cfg = O.make_cfg(BACKBONE='resnet18', NUM_BASES=64, TOPL=32, NUM_EM_ITERS=4, SINGLE_OBJ=True)
    model, sd = H.make_model_and_sd(cfg, 6, DEV, pred_scale=tc['pred_scale'])
    case = dict(b=1, t=3, hw=(128, 160), n=1, seed=77, valid=[[1, 1]])
    frames, init_mask, label, _ = H.train_batch(case)
    torch.manual_seed(17)
    ref_l, ref_res, ref_g, _ = O.train_one_step(H.trainable_sd(sd, model), cfg, frames, init_mask, None, label, 3,
                                                tc['loss_cfg'])
    tr = SWEMTrainer(dict(SOLVER=tc['solver_cfg'], LOSS=tc['loss_cfg'], AMP=False), model, use_graph=False)
    torch.manual_seed(17)
    losses, results = tr.one_step(frames.to(DEV), init_mask.to(DEV), None, label.to(DEV), 3)
    assert float(losses['total_loss']) == pytest.approx(float(ref_l['total_loss']), rel=2e-4)
    assert float((results.cpu() == ref_res).float().mean()) > 0.998
    params = dict(model.named_parameters())
    rels = sorted(abs(float(params[k].grad.double().norm()) - float(g.double().norm())) / (float(g.double().norm()) + 1e-12)
                  for k, g in ref_g.items() if g is not None)
    assert rels[len(rels) // 2] < 1e-3 and rels[int(len(rels) * 0.9)] < 2e-2, (rels[len(rels) // 2], rels[-5:])
    assert params['value_encoder.conv1.weight'].shape[1] == 4


def test_one_step_five_objects_vs_oracle(lib):
    """YouTube-VOS main-training style step: five object slots, two of them invalid in this clip (valid_obj), against the
    oracle's autograd run in the test."""
    from swem_amd.train import SWEMTrainer
    tc = H.train_cases()
    cfg = O.make_cfg(BACKBONE='resnet18', NUM_BASES=64, TOPL=32, NUM_EM_ITERS=4)
    model, sd = H.make_model_and_sd(cfg, 8, DEV, pred_scale=tc['pred_scale'])
    case = dict(b=1, t=3, hw=(128, 160), n=5, seed=91, valid=[[1, 1, 1, 1, 0, 0]])
    frames, init_mask, label, valid = H.train_batch(case)
    torch.manual_seed(19)
    ref_l, ref_res, ref_g, _ = O.train_one_step(H.trainable_sd(sd, model), cfg, frames, init_mask, valid, label, 45,
                                                tc['loss_cfg'])
    tr = SWEMTrainer(dict(SOLVER=tc['solver_cfg'], LOSS=tc['loss_cfg'], AMP=False), model, use_graph=False)
    torch.manual_seed(19)
    losses, results = tr.one_step(frames.to(DEV), init_mask.to(DEV), valid.to(DEV), label.to(DEV), 45)
    assert float(losses['total_loss'].detach()) == pytest.approx(float(ref_l['total_loss'].detach()), rel=5e-4)
    # (the shrunken prediction head leaves the five slots' probabilities nearly equal: the argmax itself is noise-level;
    # the measured agreement goes to the parity report)
    agree = float((results.cpu() == ref_res).float().mean())
    assert agree > 0.97
    params = dict(model.named_parameters())
    rels = sorted(abs(float(params[k].grad.double().norm()) - float(g.double().norm())) / (float(g.double().norm()) + 1e-12)
                  for k, g in ref_g.items() if g is not None)
    H.record_parity('train_step_five_objects_vs_oracle', {
        'index_agreement': agree, 'total_loss': float(losses['total_loss'].detach()),
        'total_loss_oracle': float(ref_l['total_loss'].detach()),
        'grad_norm_rel_err': {'median': rels[len(rels) // 2], 'p90': rels[int(len(rels) * 0.9)], 'worst': rels[-1]}})
    assert rels[len(rels) // 2] < 2e-3 and rels[int(len(rels) * 0.9)] < 5e-2, (rels[len(rels) // 2], rels[-5:])


def test_training_reduces_the_loss_on_a_fixed_batch(lib):
    """Twelve optimizer steps on one fixed batch (graph replay from the third step on): the loss goes down -- the whole loop (forward, loss, backward, all-lanes gradient sum, AdamW, scheduler) pulls in one direction."""
    from swem_amd.train import SWEMTrainer
    tc = H.train_cases()
    case = dict(tc['cases']['r18'], hw=[128, 128])
    cfg = O.make_cfg(**case['cfg'])
    model, _ = H.make_model_and_sd(cfg, case['wseed'], DEV, pred_scale=tc['pred_scale'])
    solver = dict(tc['solver_cfg'], BASE_LR=2e-4, PRETRAIN_ITERS=[1000, 2000])
    tr = SWEMTrainer(dict(SOLVER=solver, LOSS=tc['loss_cfg'], AMP=False), model)
    frames, init_mask, label, valid = [t.to(DEV) for t in H.train_batch(case)]
    torch.manual_seed(23)
    hist = []
    for it in range(12):
        losses, _ = tr.one_step(frames, init_mask, valid, label, 5)      # plain CE + IoU (below start_warm)
        hist.append(float(losses['total_loss']))
    assert tr._graph is not None
    assert hist[-1] < 0.8 * hist[0], hist
    # (every step draws new random bases, so single steps may go up)
    assert sum(1 for a, b in zip(hist, hist[1:]) if b < a) >= 6, hist


@pytest.mark.parametrize('math', [1, 2], ids=['bf16x6', 'bf16'])
@pytest.mark.parametrize('case', [
    dict(cins=[(256, 2)], cout=256, k=3, s=1, hw=(24, 24), relu_in=True),          # 128x128 tiles, whole
    dict(cins=[(136, 2)], cout=200, k=3, s=2, hw=(21, 19), relu_in=False),         # 128x128 tiles, ragged; odd size
    dict(cins=[(72, 3)], cout=40, k=3, s=1, hw=(25, 23), relu_in=False),           # 64x64 tiles, ragged both ways
    dict(cins=[(64, 1)], cout=128, k=1, s=2, hw=(16, 15), relu_in=True, pad=0),    # 1x1 stride 2 without padding
    dict(cins=[(128, 3), (256, 1), (128, 3)], cout=128, k=3, s=1, hw=(9, 11), relu_in=False),   # shared middle source
    dict(cins=[(8, 2)], cout=64, k=7, s=2, hw=(32, 40), relu_in=False, pad=3, cin_store=5),     # value-encoder stem
], ids=['whole128', 'ragged128_s2', 'ragged64', '1x1_s2', 'three_src_shared', 'stem7x7'])
def test_conv_wgrad_bf16_pipe(lib, case, math):
    """swem_conv2d_wgrad_bf16x3: the weight gradient from the pre-split bf16 planes (LDS images read through the
    hardware transpose).  math 1 (six products) carries fp32-level error against F.conv2d's autograd in fp64; math 2
    (config.AMP) equals the fp64 gradient of the bf16-ROUNDED operands to fp32 level and the true one to bf16 level."""
    from swem_amd import _lib, ops
    g = torch.Generator().manual_seed(11)
    k, s = case['k'], case['s']
    pad = case.get('pad', k // 2)
    H, W = case['hw']
    B = max(b for _, b in case['cins'])
    xs = [torch.randn(b, c, H, W, generator=g) for c, b in case['cins']]
    cin = sum(c for c, _ in case['cins'])
    cout = case['cout']
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    dy = torch.randn(B, cout, Ho, Wo, generator=g)

    def ref(rnd):
        xcat = torch.cat([x.expand(B, -1, -1, -1) for x in xs], 1)
        xcat = F.relu(xcat) if case['relu_in'] else xcat
        w = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
        F.conv2d(rnd(xcat).double(), w, stride=s, padding=pad).backward(rnd(dy).double())
        return w.grad
    srcs = [nhwc(x) for x in xs]
    d = nhwc(dy)
    d3 = ops.presplit(d, False)
    args = []
    for t in srcs:
        sp = ops.presplit(t, case['relu_in'])
        args += [sp.data_ptr(), t.shape[3], 0 if (t.shape[0] == 1 and B > 1) else H * W * t.shape[3], sp.stride(0)]
    cs = [t.shape[3] for t in srcs] + [0, 0]
    for _ in range(3 - len(srcs)):
        args += [0, 0, 0, 0]
    cin_store = case.get('cin_store', cin)
    for plan in (0, 1 | 3 << 4, 2 | 1 << 4, 2 | 2 << 4 | 1 << 12, 1 | 1 << 12):
        wsb = _lib.query('swem_conv2d_wgrad_bf16x3_workspace', B, H, W, cs[0], cs[1], cs[2], cout, k, k, s, pad, plan)
        ws = ops.workspace(wsb, d.device)
        dw = torch.full((cout, cin_store, k, k), 0.5, device=DEV)
        for acc in (0, 1):                        # plain store, then accumulate on top of it
            _lib.call('swem_conv2d_wgrad_bf16x3', ops._stream(), d3.data_ptr(), d3.stride(0), *args, B, H, W, cout, k, k, s,
                      pad, math, dw.data_ptr(), cin_store, acc, plan, ws.data_ptr(), wsb)
        got = dw.cpu().double() / 2
        exact = ref(lambda t: t)[:, :cin_store]
        if math == 1:
            close(got, exact, 3e-5, 'dW bf16x6 plan %#x' % plan)
        else:
            close(got, ref(lambda t: t.bfloat16().float())[:, :cin_store], 3e-5, 'dW bf16 (rounded operands) plan %#x' % plan)
            err = float((got - exact).abs().max() / exact.abs().max())
            assert 1e-5 < err < 2e-2, err


@pytest.mark.parametrize('gscale,tail', [(1.0, 1e3), (3.0e-7, 1e3), (2.5e4, 1e3), (1.0, 1e6)], ids=['unit', 'tiny', 'huge', 'outliers_1e6'])
@pytest.mark.parametrize('case', [
    dict(cins=[(256, 2)], cout=256, k=3, s=1, hw=(24, 24), relu_in=True),
    dict(cins=[(136, 2)], cout=200, k=3, s=2, hw=(21, 19), relu_in=False),
    dict(cins=[(72, 3)], cout=40, k=3, s=1, hw=(25, 23), relu_in=False),
    dict(cins=[(64, 1)], cout=128, k=1, s=2, hw=(16, 15), relu_in=True, pad=0),
    dict(cins=[(128, 3), (256, 1), (128, 3)], cout=128, k=3, s=1, hw=(9, 11), relu_in=False),
], ids=['whole128', 'ragged128_s2', 'ragged64', '1x1_s2', 'three_src_shared'])
def test_conv_wgrad_f16x3_scaled(lib, case, gscale, tail):
    """swem_split_f16x2_scaled_f32 + swem_conv2d_wgrad_f16x3 (round 5): the weight gradient from fp16
    (hi, mid) pairs -- dY scaled by a power of two chosen ON THE DEVICE from its largest magnitude, the activations unscaled --
    carries fp32-level error against F.conv2d's autograd in fp64 whatever the gradient's magnitude: a dY of 3e-7 (its pair would
    be all subnormal unscaled) or 2.5e4 x N(0,1) (beyond the fp16 range unscaled).  dY has a heavy tail on purpose (a few
    elements `tail` = 1e3 x the rest: the scale follows the maximum, the bulk sits ten binades below it; `outliers_1e6` (ADVICE
    r05): twenty binades -- the bulk's pairs keep ~17 bits there, an absolute error <= 2^-39 of the map's maximum, which is what
    every sum the map enters is measured against: the same bar holds)."""
    from swem_amd import _lib, ops
    ops.check_faults()
    g = torch.Generator().manual_seed(13)
    k, s = case['k'], case['s']
    pad = case.get('pad', k // 2)
    H_, W_ = case['hw']
    B = max(b for _, b in case['cins'])
    xs = [torch.randn(b, c, H_, W_, generator=g) for c, b in case['cins']]
    cin = sum(c for c, _ in case['cins'])
    cout = case['cout']
    Ho, Wo = (H_ + 2 * pad - k) // s + 1, (W_ + 2 * pad - k) // s + 1
    dy = torch.randn(B, cout, Ho, Wo, generator=g)
    dy[torch.rand(dy.shape, generator=g) < 1e-3] *= tail
    dy = dy * gscale
    xcat = torch.cat([x.expand(B, -1, -1, -1) for x in xs], 1)
    xcat = F.relu(xcat) if case['relu_in'] else xcat
    w = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(xcat.double(), w, stride=s, padding=pad).backward(dy.double())
    exact = w.grad
    srcs = [nhwc(x) for x in xs]
    d = nhwc(dy)
    d.__dict__['_swem_grad'] = True
    d2 = ops.presplit(d, False, ops.PLANES_F16)
    inv = d.__dict__['_swem_inv']
    amax = float(dy.abs().max())
    s_exp = 13 - int(np.floor(np.log2(amax)))
    assert float(inv[0]) == 2.0 ** -s_exp, (float(inv[0]), s_exp)
    # the planes ARE the pair of dy * 2^s: hi + mid reproduces it to 2^-22 of the element or 2^-25 absolute (scaled units)
    rec = (d2[0].float() + d2[1].float()).view(cout // 8, B * Ho * Wo, 8).permute(1, 0, 2).reshape(B, Ho, Wo, cout)
    want = d.float() * 2.0 ** s_exp
    assert float(((rec - want).abs() - want.abs() * 2.0 ** -22).clamp_min(0).max()) <= 2.0 ** -25
    args = []
    for t in srcs:
        sp = ops.presplit(t, case['relu_in'], ops.PLANES_F16)
        args += [sp.data_ptr(), t.shape[3], 0 if (t.shape[0] == 1 and B > 1) else H_ * W_ * t.shape[3], sp.stride(0)]
    cs = [t.shape[3] for t in srcs] + [0, 0]
    for _ in range(3 - len(srcs)):
        args += [0, 0, 0, 0]
    for plan in (0, 1 | 3 << 4, 2 | 1 << 4, 2 | 2 << 4 | 1 << 12):
        wsb = _lib.query('swem_conv2d_wgrad_bf16x3_workspace', B, H_, W_, cs[0], cs[1], cs[2], cout, k, k, s, pad, plan)
        ws = ops.workspace(wsb, d.device)
        dw = torch.full((cout, cin, k, k), 0.5 * gscale, device=DEV)
        for acc in (0, 1):
            _lib.call('swem_conv2d_wgrad_f16x3', ops._stream(), d2.data_ptr(), d2.stride(0), *args, B, H_, W_, cout, k, k, s, pad,
                      inv.data_ptr(), dw.data_ptr(), cin, acc, plan, ws.data_ptr(), wsb)
        close(dw.cpu().double() / 2, exact, 3e-5, 'dW f16x3 plan %#x' % plan)
        err = float((dw.cpu().double() / 2 - exact).abs().max() / exact.abs().max())
        assert err < 4e-6, err              # (fp32-level: bf16x6 measures 1-2e-6 on these shapes)
    ops.check_faults()


def test_scaled_gradient_planes_through_the_data_gradient_conv(lib):
    """A gradient map marked by autograd (`_swem_grad`) that reaches an f16x3 data-gradient convolution is split into the SCALED
    pair and the 2^-s goes into the epilogue scale on the device (ops.conv2d / swem_vec_scale_f32): dX against autograd in fp64
    for gradients of 1e-7 and 1e4, with the input-ReLU mask, next to bf16x6 on the same data.  A non-finite gradient sets the
    range fault."""
    from swem_amd import autograd as A, ops
    ops.check_faults()
    A.new_step()
    g = torch.Generator().manual_seed(17)
    for gscale in (1.0, 1.0e-7, 1.0e4):
        x = leaf(torch.randn(2, 64, 20, 28, generator=g))
        w = leaf(torch.randn(96, 64, 3, 3, generator=g) * 0.05)
        y = F.conv2d(F.relu(x.double()), w.double(), padding=1)
        dy = torch.randn(*y.shape, generator=g) * gscale
        gx_ref, gw_ref = torch.autograd.grad(y, (x, w), dy.double())
        outs = {}
        for modes in ((7,), (1,)):
            with ops.use_book(ops.PlanBook()), ops.conv_math(modes):
                A.new_step()
                wp = param(w)
                sx = nhwc(x.detach()).requires_grad_(True)
                out = A.conv2d([sx], wp, None, stride=1, pad=1, relu_in=True)
                out.backward(nhwc(dy))
                outs[modes] = (back(sx.grad).double(), wp.grad.cpu().double())
                ops.check_faults()
        for modes, (gx, gw) in outs.items():
            ex = float((gx - gx_ref).abs().max() / gx_ref.abs().max())
            ew = float((gw - gw_ref).abs().max() / gw_ref.abs().max())
            print('gradient scale %g, modes %s: dX rel err %.2e, dW %.2e' % (gscale, modes, ex, ew))
            assert ex < 3e-6 and ew < 4e-6, (gscale, modes, ex, ew)
    with ops.use_book(ops.PlanBook()), ops.conv_math((7,)):
        A.new_step()
        sx = nhwc(torch.randn(1, 64, 8, 8, generator=g)).requires_grad_(True)
        out = A.conv2d([sx], param(torch.randn(64, 64, 3, 3, generator=g) * 0.05), None, stride=1, pad=1)
        bad = torch.randn(*out.shape, generator=g)
        bad[0, 3, 3, 3] = float('inf')
        out.backward(bad.to(DEV))
        with pytest.raises(ops.SwemRangeError):
            ops.check_faults()


def test_pack_filters_f16x2_kernel_equals_the_host_recipe(lib):
    """swem_pack_filters_f16x2_f32 (the training step's per-step filter packs, one launch) against ConvPack.planes16's torch
    recipe: the same planes and epilogue scales, bit for bit, wherever log2 does not round across a power of two."""
    from swem_amd import ops
    g = torch.Generator().manual_seed(3)
    for co, ci, k in ((64, 64, 3), (200, 96, 3), (128, 256, 1), (32, 32, 7)):
        w = (torch.randn(co, ci, k, k, generator=g) * 0.03).to(DEV)
        w[1] *= 1.0e-6
        w[2] = 0.0
        w[3] *= 300.0
        bn = [t.to(DEV) for t in (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g), torch.randn(co, generator=g),
                                  torch.rand(co, generator=g) + 0.5)]
        a, b = ops.pack_conv(w, None, bn, 1, k // 2), ops.pack_conv(w, None, bn, 1, k // 2)
        b.fast16 = True
        (wa, sa), (wb, sb) = a.planes16(), b.planes16()
        assert torch.equal(wa.view(torch.int16), wb.view(torch.int16)) and torch.equal(sa, sb), (co, ci, k)


def test_amp_step_tracks_the_fp32_step(lib):
    """config.AMP (configs/config.py:89): the convolutions' forward / data gradient run on bf16 operands (one MFMA
    product, fp32 accumulate; the reference autocasts to fp16 with a GradScaler, basic_trainer.py:83-86,222).  The step
    must stay a bf16-level perturbation of the fp32 step -- same loss to a percent, flat gradient aligned -- and must
    really be the reduced-precision one (not bit-equal to fp32).  Graph replay of the AMP step equals its eager form."""
    from swem_amd.train import SWEMTrainer
    tc = H.train_cases()
    case = dict(tc['cases']['r18'], hw=[128, 128])
    cfg = O.make_cfg(**case['cfg'])
    frames, init_mask, label, valid = [t.to(DEV) for t in H.train_batch(case)]
    out = {}
    for name, amp, graph in (('fp32', False, False), ('amp', True, False), ('amp_graph', True, True)):
        model, _ = H.make_model_and_sd(cfg, case['wseed'], DEV, pred_scale=tc['pred_scale'])
        tr = SWEMTrainer(dict(SOLVER=dict(tc['solver_cfg'], BASE_LR=1e-4), LOSS=tc['loss_cfg'], AMP=amp), model,
                         use_graph=graph)
        torch.manual_seed(31)
        hist = []
        for it in (5, 6, 7, 8):
            losses, _ = tr.one_step(frames, init_mask, valid, label, it)
            hist.append(float(losses['total_loss']))
            if it == 5:
                g0 = tr.optimizer.grad.detach().clone()
        assert (tr._graph is not None) == graph
        out[name] = (hist, g0, tr.optimizer.param.detach().clone())
    (h32, g32, _), (ha, ga, pa), (hg, gg, pg) = out['fp32'], out['amp'], out['amp_graph']
    assert abs(ha[0] - h32[0]) < 1e-2 * abs(h32[0]), (ha, h32)
    assert ha[0] != h32[0] and not torch.equal(ga, g32)
    cos = float((ga.double() * g32.double()).sum() / (ga.double().norm() * g32.double().norm()))
    assert cos > 0.98, cos
    assert abs(float(ga.norm()) / float(g32.norm()) - 1) < 0.05
    assert hg == ha and torch.equal(gg, ga) and torch.equal(pg, pa)
    assert all(abs(a - b) < 0.05 * abs(b) for a, b in zip(ha, h32)), (ha, h32)


def amp_policy(topl):
    """Which GEMMs of a conv layer the build's config.AMP runs on bf16-rounded operands (mirrors ops.conv2d's pre-split
    condition and autograd.wgrad_math under ops.conv_math((2,))): forward when every source has a multiple of 32
    channels, data gradient when Cout is one, weight gradient for every 3x3 / 7x7 layer whose channel counts are
    multiples of 8 and for the large 1x1 layers; the prediction head and the two stems (3 / 5 input channels) stay fp32."""
    multi = {'value_encoder.fuser.block1.conv1': None, 'value_encoder.fuser.block1.downsample': None,
             'swem_core.fusion_layer.layer_f': (512, 512, 2 * topl), 'swem_core.fusion_layer.layer_a': (512, 512, 2 * topl)}

    def policy(name, x, w, stride):
        co, ci, kh, kw = w.shape
        if name == 'decoder.pred' or ci < 8:
            return (False, False, False)
        cs = multi.get(name) or ((256, ci - 256) if name in multi else (ci,))
        M = x.shape[0] * (x.shape[2] // stride) * (x.shape[3] // stride)
        fwd = all(c % 32 == 0 for c in cs) and (ci * kh * kw) % 8 == 0
        dgrad = co % 32 == 0
        big = co >= 128 and all(c >= 128 for c in cs)
        wgrad = all(c % 8 == 0 for c in cs) and co % 8 == 0 and (kh * kw > 1 or (big and M >= 2048))
        return (fwd, dgrad, wgrad)
    return policy


def test_amp_step_vs_rounded_operand_oracle(lib):
    """config.AMP against an ORACLE of the same arithmetic: the CPU restatement with every bf16-mode GEMM replaced by an
    fp32 convolution on bf16-ROUNDED operands (oracle.ROUNDED_CONV), forward, data gradient and weight gradient."""
    from swem_amd.train import SWEMTrainer
    tc = H.train_cases()
    case = dict(tc['cases']['r18'], hw=[128, 128], b=1, valid=[[1, 1, 1]])
    cfg = O.make_cfg(**case['cfg'])
    frames, init_mask, label, valid = H.train_batch(case)
    model, sd = H.make_model_and_sd(cfg, case['wseed'], DEV, pred_scale=tc['pred_scale'])
    res = {}
    for name, pol in (('fp32', None), ('rounded', amp_policy(cfg.TOPL))):
        O.ROUNDED_CONV = pol
        try:
            torch.manual_seed(91)
            ol, _, og, _ = O.train_one_step(H.trainable_sd(sd, model), cfg, frames, init_mask, valid, label, 45, tc['loss_cfg'])
        finally:
            O.ROUNDED_CONV = None
        res[name] = ({k: float(v) for k, v in ol.items() if k != 'p'}, {k: g for k, g in og.items() if g is not None})
    tr = SWEMTrainer(dict(SOLVER=tc['solver_cfg'], LOSS=tc['loss_cfg'], AMP=True), model, use_graph=False)
    torch.manual_seed(91)
    losses, _ = tr.one_step(frames.to(DEV), init_mask.to(DEV), valid.to(DEV), label.to(DEV), 45)
    params = dict(model.named_parameters())
    out = {}
    for name, (ol, og) in res.items():
        rel = sorted(abs(float(params[k].grad.double().norm()) - float(g.double().norm())) / (float(g.double().norm()) + 1e-12)
                     for k, g in og.items())
        flat_h = torch.cat([params[k].grad.flatten().cpu().double() for k in og])
        flat_o = torch.cat([g.flatten().double() for g in og.values()])
        out[name] = {'loss_rel': abs(float(losses['total_loss']) - ol['total_loss']) / ol['total_loss'],
                     'grad_norm_rel_median': rel[len(rel) // 2], 'grad_norm_rel_p90': rel[int(0.9 * len(rel))],
                     'flat_grad_rel_l2': float((flat_h - flat_o).norm() / flat_o.norm())}
    # the two oracles against each other: the size of the bf16 effect itself
    (l32, g32), (lr, gr) = res['fp32'], res['rounded']
    a = torch.cat([g32[k].flatten().double() for k in gr]); b = torch.cat([gr[k].flatten().double() for k in gr])
    out['oracle_rounded_vs_oracle_fp32'] = {'loss_rel': abs(lr['total_loss'] - l32['total_loss']) / l32['total_loss'],
                                            'flat_grad_rel_l2': float((a - b).norm() / a.norm())}
    print('AMP step vs oracles:', out)
    H.record_parity('amp_step_vs_rounded_operand_oracle', out)
    # Rounding to bf16 is a step function: an fp32-level difference in a layer's output (another summation order) flips the
    # rounding of ~1/256 of the next layer's operands by 2^-9 relative each, i.e. two correct implementations of the SAME
    # rounded-operand arithmetic drift apart ~250 x faster than two fp32 ones (5e-5 on this step, test_one_step_*) -- to the
    # 1e-2 level after ~60 layers and the EM.  The HIP step is therefore held to that level against the rounded-operand
    # oracle (per-layer the bf16 kernels are exact to 3e-5 on rounded operands: test_conv2d_plain_bf16_mode,
    # test_conv_wgrad_bf16_pipe), must not be further from it than from the fp32 oracle, and both distances are recorded.
    r, f = out['rounded'], out['fp32']
    assert r['loss_rel'] < 3e-3 and r['grad_norm_rel_median'] < 1e-2 and r['flat_grad_rel_l2'] < 2e-2, out
    assert r['flat_grad_rel_l2'] <= 1.1 * f['flat_grad_rel_l2'], out


def test_inference_after_a_step_sees_the_updated_weights(lib):
    """SWEM.engine() caches packed conv filters; an optimizer step changes the parameters in place, so the trainer
    invalidates the cache: inference through the trained model equals a fresh model loaded from its state_dict()."""
    from swem_amd import evaluator, synth
    from swem_amd.swem import SWEM
    from swem_amd.train import SWEMTrainer
    tc = H.train_cases()
    case = dict(tc['cases']['r18'], hw=[128, 128])
    cfg = O.make_cfg(**case['cfg'])
    model, _ = H.make_model_and_sd(cfg, case['wseed'], DEV, pred_scale=tc['pred_scale'])
    frames, m0 = synth.make_clip(t=3, h=128, w=192, n_obj=2, seed=8)
    frames, m0 = frames.to(DEV), m0.to(DEV)

    def infer(m):
        m.swem_core.init_on_host = True
        with torch.no_grad():
            torch.manual_seed(5)
            _, scores = evaluator.evaluate_davis_seq(m, frames, [m0, None, None], (128, 192))
        return scores[-1].clone()
    before = infer(model)                                   # builds (and caches) the engine on the initial weights
    solver = dict(tc['solver_cfg'], BASE_LR=1e-2)           # a step large enough to move the outputs visibly
    trainer = SWEMTrainer(dict(SOLVER=solver, LOSS=tc['loss_cfg'], AMP=False), model, use_graph=False)
    fr, im, lb, va = [t.to(DEV) for t in H.train_batch(case)]
    torch.manual_seed(91)
    trainer.one_step(fr, im, va, lb, 5)
    model.eval()
    after = infer(model)
    fresh = SWEM(cfg)
    fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    fresh = fresh.eval().to(DEV)
    fresh.book.fallback = model.book.fallback               # (the same arithmetic as the test model: helpers.make_model_and_sd)
    ref = infer(fresh)
    assert float((after - before).abs().max()) > 1e-3, 'the step did not move the outputs: the test is vacuous'
    assert torch.equal(after, ref)


@pytest.mark.parametrize('graph,opts', [(False, ()), (True, ()), (True, ('single_allreduce',)), (True, ('nccl',)), (False, ('nccl',))],
                         ids=['eager', 'graph', 'graph_single_allreduce', 'graph_rccl_two_gpus', 'eager_rccl_two_gpus'])
def test_two_ranks_one_clip_each_equal_one_rank_two_clips(lib, tmp_path, graph, opts):
    """Data parallel (swem_trainer.py:41-43: DistributedDataParallel): two ranks that step one clip each -- parameters
    broadcast from rank 0 at start-up, gradient all-reduced in two overlapped slices, loss scalars in one 3-float message --
    end with the parameters of ONE rank stepping both clips, on every rank, and report the batch's mean losses.
    Rehearsed on one GPU over gloo (the RCCL path is the same torch.distributed calls); the `rccl_two_gpus` variants run the
    same comparison with one rank per GPU over RCCL -- collective kernels beside the lanes' graph replays -- wherever the box
    has two GPUs (skipped on the one-GPU boxes of this pool); `single_allreduce` = SWEMTrainer(overlap_allreduce=False)."""
    import os
    from swem_amd import dist as sdist, train
    from swem_amd.train import SWEMTrainer
    if 'nccl' in opts and torch.cuda.device_count() < 2:
        pytest.skip('one rank per GPU over RCCL needs two GPUs')
    steps = 4 if graph else 2                        # (the graph is captured after two eager steps)
    out = str(tmp_path / 'ranks.pt')
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_dist_train_probe.py')
    env = dict(os.environ)
    env.pop('SWEM_DIST_BACKEND', None)
    rc, text = sdist.launch_ranks(2, [probe, out, str(steps)] + (['graph'] if graph else []) + list(opts), env=env, timeout=900)
    assert rc == 0, text
    got = torch.load(out)
    assert got['world'] == 2 and got['same_on_all_ranks']
    # the same two clips on one rank
    tc = H.train_cases()
    case = dict(tc['cases']['r18'], hw=[128, 128])
    cfg = O.make_cfg(**case['cfg'])
    model, _ = H.make_model_and_sd(cfg, case['wseed'], DEV, pred_scale=tc['pred_scale'])
    frames, init_mask, label, valid = [t.to(DEV) for t in H.train_batch(case)]
    tr = SWEMTrainer(dict(SOLVER=dict(tc['solver_cfg'], BASE_LR=1e-4), LOSS=tc['loss_cfg'], AMP=False), model, use_graph=graph)
    real = train.random_init_host
    hist = []
    try:
        for it in range(steps):
            torch.manual_seed(1000 + it)
            full = real(2, case['n'], 128, cfg.NUM_BASES)
            train.random_init_host = lambda B, N, Cc, Lb, _f=full: _f.clone()
            losses, _ = tr.one_step(frames, init_mask, valid, label, 5 + it)
            hist.append([float(losses[k]) for k in ('total_loss', 'main_loss', 'aux_loss')])
    finally:
        train.random_init_host = real
    one = tr.optimizer.param.detach().cpu()
    d = float((one - got['param']).abs().max())
    print('two ranks vs one rank: max |dparam| %.3g, losses %s vs %s' % (d, got['hist'][-1], hist[-1]))
    # each clip's gradient is the same deterministic kernel sequence on both sides; the two-addend sums commute
    assert torch.equal(one, got['param']), d
    for a, b in zip(got['hist'], hist):
        assert a == pytest.approx(b, rel=1e-6)


def test_range_fault_in_a_training_step_never_reaches_the_weights(lib):
    """fp32-level training runs f16x3 by default, and its asynchronous faults used to be read every 20 steps
    AFTER optimizer.step().  Now the step's fault flags gate the AdamW launch on the device (swem_adamw_gated_f32, fed by
    swem_fault_flags_f32 through the loss scalars' all-reduce): (a) an eager step that leaves the fp16 range -- the key encoder's
    stem scaled by 3e4 -- moves nothing, the trainer switches to the reference's fp32 range (modes 0 / 1), warns and redoes the
    step: parameters equal, bit for bit, to a trainer that never ran f16x3; (b) a fault inside REPLAYED steps that the host does
    not look at is held back just the same: parameters and moments frozen from the faulting step until the host looks, step and
    scheduler counts wound back, then the step at hand redone."""
    from swem_amd import ops, train
    from swem_amd.train import SWEMTrainer
    tc = H.train_cases()
    case = dict(tc['cases']['r18'], hw=[128, 128])
    cfg = O.make_cfg(**case['cfg'])
    frames, init_mask, label, valid = [t.to(DEV) for t in H.train_batch(case)]
    torch.manual_seed(4)
    fixed = train.random_init_host(2, case['n'], 128, cfg.NUM_BASES)
    real = train.random_init_host
    train.random_init_host = lambda B, N, Cc, Lb: fixed.clone()

    def make(scale):
        model, sd = H.make_model_and_sd(cfg, case['wseed'], DEV, pred_scale=tc['pred_scale'])
        if scale != 1.0:
            sd = dict(sd)
            sd['key_encoder.conv1.weight'] = sd['key_encoder.conv1.weight'] * scale
            model.load_state_dict(sd, strict=True)
        return model
    conf = dict(SOLVER=dict(tc['solver_cfg'], BASE_LR=1e-4), LOSS=tc['loss_cfg'], AMP=False)
    try:
        # (a) a real range fault in the first, eager step
        tr = SWEMTrainer(conf, make(3.0e4), use_graph=False)
        assert tr.f16x3
        tr.math_modes = (7,)                # (the tuner is off in the tests: every layer the pre-split kernel takes, in f16x3)
        with pytest.warns(RuntimeWarning, match='fp32-range arithmetic'):
            losses, _ = tr.one_step(frames, init_mask, valid, label, 45)
        assert not tr.f16x3 and tr.optimizer.step_count == 1 and tr.lr_scheduler.last_epoch == 1
        ref = SWEMTrainer(conf, make(3.0e4), use_graph=False)
        ref.f16x3 = False
        rl, _ = ref.one_step(frames, init_mask, valid, label, 45)
        assert torch.equal(tr.optimizer.param, ref.optimizer.param)
        assert float(losses['total_loss']) == float(rl['total_loss'])
        ops.check_faults()
        # (b) replayed steps, a fault the host does not look at for two steps
        tr = SWEMTrainer(conf, make(1.0), use_graph=True)
        tr.fault_check_every = 3
        for it in range(3):
            tr.one_step(frames, init_mask, valid, label, 45)
        assert tr._graph is not None and tr.optimizer.step_count == 3
        p3, m3 = tr.optimizer.param.clone(), tr.optimizer.m.clone()
        ops.fault_word(torch.device(DEV)).fill_(ops.FAULT_RANGE)      # as a launch of step 4 would have
        for it in range(2):
            tr.one_step(frames, init_mask, valid, label, 45)          # steps 4, 5: not looked at
        assert torch.equal(tr.optimizer.param, p3) and torch.equal(tr.optimizer.m, m3)
        with pytest.warns(RuntimeWarning, match='3 steps skipped'):
            tr.one_step(frames, init_mask, valid, label, 45)          # step 6: looked at, wound back, redone
        assert tr.optimizer.step_count == 4 and tr.lr_scheduler.last_epoch == 4 and int(tr.optimizer.applied.item()) == 4
        assert not torch.equal(tr.optimizer.param, p3)
        ops.check_faults()
    finally:
        train.random_init_host = real


def test_fault_word_ownership(lib):
    """ONE sticky fault word per device, several owners of launches.  A fault a training step left must not
    be read by the validation sequence that runs next and blamed on ITS model (book converted to the full-range arithmetic for
    good): the evaluator drains the word before its first launch and hands what it finds to the registered trainer; with no
    trainer alive the same drain raises a stale-fault error instead of blaming the sequence."""
    import warnings
    from swem_amd import evaluator, ops, synth
    from swem_amd.train import SWEMTrainer
    tc = H.train_cases()
    case = dict(tc['cases']['r18'], hw=[128, 128])
    cfg = O.make_cfg(**case['cfg'])
    model, _ = H.make_model_and_sd(cfg, case['wseed'], DEV, pred_scale=tc['pred_scale'])
    tr = SWEMTrainer(dict(SOLVER=tc['solver_cfg'], LOSS=tc['loss_cfg'], AMP=False), model, use_graph=False)
    val, _ = H.make_model_and_sd(cfg, 3, DEV)
    val.book.fallback = ops.MODEL_FALLBACK
    frames, m0 = synth.make_clip(t=3, h=128, w=192, n_obj=2, seed=9)
    frames, m0 = frames.to(DEV), m0.to(DEV)
    word = ops.fault_word(torch.device(DEV))
    word.fill_(ops.FAULT_RANGE)                              # "a training step faulted and nobody has looked yet"
    with torch.no_grad(), pytest.warns(RuntimeWarning, match='handed it to 1 registered owner'):
        evaluator.evaluate_davis_seq(H.SeededInit(val, 3), frames, [m0, None, None], (128, 192))
    assert not val.book.full_range, 'the validation model was blamed for the trainer\'s fault'
    assert tr._foreign_fault == ops.FAULT_RANGE and int(word.item()) == 0
    # nobody to take it: loud, and named for what it is
    del tr
    ops.FAULT_OWNERS.clear()
    word.fill_(ops.FAULT_KSPLIT)
    with torch.no_grad(), pytest.raises(ops._lib.SwemHipError, match='stale asynchronous fault'):
        evaluator.evaluate_davis_seq(H.SeededInit(val, 3), frames, [m0, None, None], (128, 192))
    assert int(word.item()) == 0
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter('error')
        evaluator.evaluate_davis_seq(H.SeededInit(val, 3), frames, [m0, None, None], (128, 192))


@pytest.mark.parametrize('opts', [(), ('graph',), ('graph', 'reduce_in_graph')], ids=['eager', 'graph', 'graph_allreduce_captured'])
def test_rccl_single_rank_training_step(lib, tmp_path, opts):
    """RCCL met once on the hardware there is.  ONE rank with a real "nccl" (= RCCL) process group
    (SWEM_DIST_SINGLE_RANK=1, swem_amd.dist.single_rank_group): communicator creation, barrier(device_ids), the (frames, seconds)
    reduction, an all-reduce recorded into a HIP graph and replayed, and SWEMTrainer.one_step with its bucketed gradient
    all-reduces + the 3-float loss all-reduce really issued (train.py: swem_trainer.py:41-43's DistributedDataParallel) -- eager,
    between the replays of the step's graphs, and (`reduce_in_graph`) as a node INSIDE the captured step.  With one rank SUM is
    the identity: the parameters after the steps equal those of a process without any process group, bit for bit."""
    import json
    import os
    import subprocess
    import sys
    from swem_amd import train
    from swem_amd.train import SWEMTrainer
    steps = 4 if 'graph' in opts else 2
    out = str(tmp_path / 'rccl1.pt')
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_rccl_single_rank_probe.py')
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'SWEM_DIST_BACKEND'):
        env.pop(k, None)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    run = subprocess.run([sys.executable, probe, out, str(steps)] + list(opts), env=env, capture_output=True, text=True, timeout=900)
    if run.returncode != 0:                            # (the whole story, where the GPU box's run can be read afterwards)
        log = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'rccl_single_rank_%s.err' % '_'.join(opts or ('eager',)))
        os.makedirs(os.path.dirname(log), exist_ok=True)
        with open(log, 'w') as f:
            f.write(run.stdout + '\n---- stderr ----\n' + run.stderr)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    got = torch.load(out)
    rep = got['report']
    print('RCCL single rank:', json.dumps(rep))
    assert rep['backend'] == 'nccl' and rep['world'] == 1 and rep['active']
    assert rep['counters'] == [10, 1.5]
    assert rep['captured_allreduce'] == [[1.0, 1.0], [3.0, 3.0], [5.0, 5.0]]
    assert rep['graph'] == ('graph' in opts)
    # two gradient slices + the loss scalars per step, every step (the captured form records its slice once and replays it)
    assert rep['all_reduce_calls'] >= (2 * steps if 'reduce_in_graph' in opts else 3 * steps), rep
    tc = H.train_cases()
    case = dict(tc['cases']['r18'], hw=[128, 128])
    cfg = O.make_cfg(**case['cfg'])
    model, _ = H.make_model_and_sd(cfg, case['wseed'], DEV, pred_scale=tc['pred_scale'])
    frames, init_mask, label, valid = [t.to(DEV) for t in H.train_batch(case)]
    tr = SWEMTrainer(dict(SOLVER=dict(tc['solver_cfg'], BASE_LR=1e-4), LOSS=tc['loss_cfg'], AMP=False), model,
                     use_graph='graph' in opts)
    real = train.random_init_host
    hist = []
    try:
        for it in range(steps):
            torch.manual_seed(1000 + it)
            full = real(2, case['n'], 128, cfg.NUM_BASES)
            train.random_init_host = lambda B, N, Cc, Lb, _f=full: _f.clone()
            losses, _ = tr.one_step(frames, init_mask, valid, label, 5 + it)
            hist.append([float(losses[k]) for k in ('total_loss', 'main_loss', 'aux_loss')])
    finally:
        train.random_init_host = real
    assert torch.equal(tr.optimizer.param.detach().cpu(), got['param'])
    assert hist == rep['hist']
    H.record_parity('rccl_single_rank_training_step[%s]' % ('+'.join(opts) or 'eager'), rep)


@pytest.mark.parametrize('lanes', [0, 2], ids=['independent_pipelines', 'lockstep_lanes'])
def test_bench_under_a_single_rank_rccl_group(lib, lanes):
    """`bench.py` with a one-rank RCCL process group (SWEM_DIST_SINGLE_RANK=1): barrier(device_ids) and the counter all-reduce of
    the timed regions go through librccl; the line says rccl_ranks = 1.  Both launch forms: two independent pipelines, and the
    default's form -- lock-step lanes (here two lanes of two sequences), whose graphs are captured while the process group's
    watchdog thread is alive."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    env = dict(os.environ, SWEM_DIST_SINGLE_RANK='1', SWEM_DIST_BACKEND='nccl')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '4', '--warmup', '2', '--regions', '3',
                          '--lookahead', '2', '--no-autotune', '--no-cpu-baseline', '--no-em', '--no-legs']
                         + (['--seqs', '4', '--lockstep', '2'] if lanes else ['--seqs', '2', '--lockstep', '0']),
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['rccl_ranks'] == 1 and line['value'] > 0
    assert line['config']['lockstep_lanes'] == lanes and line['config']['frames_per_step'] == (4 if lanes else 2)


AMP_LAYERS = ('key_encoder.res2.0.conv2', 'key_encoder.layer2.0.conv2', 'key_encoder.layer3.5.conv3', 'key_proj.key_proj',
              'value_encoder.layer2.0.conv1', 'value_encoder.fuser.block1.conv1', 'value_encoder.fuser.block2.conv2',
              'decoder.compress.conv1', 'decoder.up_16_8.out_conv.conv1', 'decoder.up_8_4.skip_conv',
              'decoder.up_8_4.out_conv.conv2')


@pytest.mark.parametrize('case_name', ['r50k256', 'r50k256n5'], ids=['two_objects', 'five_objects_one_invalid'])
def test_amp_step_stage_by_stage_at_the_training_shapes(lib, case_name):
    """BASELINE configs C / D at their STATED precision: one config.AMP training step at the reference's training shapes
    (ResNet-50, K = 256, 3 x 384x384 frames, 2 objects / 5 objects with one invalid), checked where the rounding step function
    cannot compound -- stage by stage, TEACHER-FORCED: for eleven layers that span every stage of the step (trunk 3x3 /
    strided / 1x1, key projection, value encoder, the two-source fuser, decoder at 1/16, 1/8 and 1/4) the first call's inputs
    (activation, filters, output gradient) are taken out of the running step, and the three GEMMs the step ran on them
    (forward, data gradient, weight gradient) are compared with the fp32 convolution of the bf16-ROUNDED operands
    (the arithmetic of oracle.ROUNDED_CONV / _ConvRounded): <= 1e-4.  The check discriminates: the same GEMMs on the
    UNROUNDED operands are 6e-5 .. 3e-3 away, and every bf16-mode result must be at least 20 x its own error from them."""
    import torch.nn.functional as F
    from swem_amd import autograd as A, ops
    from swem_amd.train import SWEMTrainer
    tc = H.train_cases()
    case = tc['cases'][case_name]
    cfg = O.make_cfg(**case['cfg'])
    frames, init_mask, label, valid = [t.to(DEV) for t in H.train_batch(case)]
    model, _ = H.make_model_and_sd(cfg, case['wseed'], DEV, pred_scale=tc['pred_scale'])
    names = {id(p): n[:-len('.weight')] for n, p in model.named_parameters() if n.endswith('.weight')}
    want = set(AMP_LAYERS)
    rec, seen = {}, set()
    real_fwd, real_bwd = A._Conv.forward, A._Conv.backward

    def fwd(ctx, weight, bias, residual, meta, *srcs):
        y = real_fwd(ctx, weight, bias, residual, meta, *srcs)
        name = names.get(id(weight))
        if name in want and name not in seen:
            seen.add(name)
            ctx._amp_rec = name
            rec[name] = dict(meta=meta, w=weight.detach().clone(), b=None if bias is None else bias.detach().clone(),
                             res=None if residual is None else residual.detach().clone(), srcs=[s.detach().clone() for s in srcs],
                             y=y.detach().clone(), fwd_math=(ops.MATH_RAN or {}).copy())
        return y

    def bwd(ctx, dy):
        out = real_bwd(ctx, dy)
        name = getattr(ctx, '_amp_rec', None)
        if name is not None:
            r = rec[name]
            stride, pad, relu_in, batch, cin_pad = r['meta']
            weight = ctx.saved_tensors[0]
            srcs = list(ctx.saved_tensors[2:])
            r['dy'] = dy.detach().clone()
            r['dx'] = [None if g is None else g.detach().clone() for g in out[4:]]
            buf = torch.zeros_like(weight)
            saved = (A._LANE, A._LANE_GRADS)
            A.use_lane(saved[0], {id(weight): buf})
            try:
                A._wgrad(dy.contiguous(), srcs, weight, stride, pad, relu_in)
            finally:
                A.use_lane(*saved)
            r['dw'] = buf
            cs = [s.shape[3] for s in srcs]
            r['wgrad_math'] = A.wgrad_math(cs, weight.shape[0], weight.shape[2], weight.shape[3], dy.shape[0] * dy.shape[1] * dy.shape[2])
        return out
    A._Conv.forward, A._Conv.backward = staticmethod(fwd), staticmethod(bwd)
    try:
        tr = SWEMTrainer(dict(SOLVER=tc['solver_cfg'], LOSS=tc['loss_cfg'], AMP=True), model, use_graph=False, lanes=1)
        torch.manual_seed(91)
        losses, _ = tr.one_step(frames, init_mask, valid, label, 45)
    finally:
        A._Conv.forward, A._Conv.backward = real_fwd, real_bwd
    torch.cuda.synchronize()
    assert math.isfinite(float(losses['total_loss'])) and seen == want, sorted(want - seen)
    r16 = lambda t: t.bfloat16().float()
    nchw = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()
    rows = {}
    for name in AMP_LAYERS:
        r = rec[name]
        stride, pad, relu_in, batch, cin_pad = r['meta']
        w = r['w'].cpu()
        B = r['y'].shape[0]
        xs = [nchw(s) for s in r['srcs']]
        x = torch.cat([t.expand(B, -1, -1, -1) if t.shape[0] == 1 and B > 1 else t for t in xs], 1)
        xa = F.relu(x) if relu_in else x
        dy = nchw(r['dy'])
        presplit = all(s.shape[3] % 32 == 0 for s in r['srcs'])          # (ops.conv2d: the pre-split kernel's condition)
        out = {'amp_forward': presplit, 'amp_dgrad': w.shape[0] % 32 == 0, 'amp_wgrad': r['wgrad_math'] == 2}

        def refs(rounded_x, rounded_w, rounded_d):
            xx, ww, dd = (r16(xa) if rounded_x else xa), (r16(w) if rounded_w else w), (r16(dy) if rounded_d else dy)
            return xx, ww, dd
        # forward
        for tag, rd in (('rounded', True), ('fp32', False)):
            xx, ww, _ = refs(rd, rd, False)
            y = F.conv2d(xx, ww, None if r['b'] is None else r['b'].cpu(), stride=stride, padding=pad)
            if r['res'] is not None:
                y = y + nchw(r['res'])
            out['fwd_vs_' + tag] = H.rel_err(nchw(r['y']), y)
            # data gradient (towards the first source that asked for one), input-ReLU mask folded in
            _, ww, dd = refs(False, rd, rd)
            dx = torch.nn.grad.conv2d_input(x.shape, ww, dd, stride=stride, padding=pad)
            if relu_in:
                dx = dx * (x > 0)
            off = 0
            for s, g in zip(xs, r['dx']):
                c = s.shape[1]
                if g is not None and s.shape[0] == B:
                    out['dgrad_vs_' + tag] = H.rel_err(nchw(g), dx[:, off:off + c])
                    break
                off += c
            xx, _, dd = refs(rd, False, rd)
            dw = torch.nn.grad.conv2d_weight(xx, w.shape, dd, stride=stride, padding=pad)
            out['wgrad_vs_' + tag] = H.rel_err(r['dw'].cpu(), dw)
        rows[name] = out
        print(name, {k: ('%.2e' % v if isinstance(v, float) else v) for k, v in out.items()})
        for gemm, flag in (('fwd', out['amp_forward']), ('dgrad', out['amp_dgrad']), ('wgrad', out['amp_wgrad'])):
            if gemm + '_vs_rounded' not in out:
                continue
            good, other = ('rounded', 'fp32') if flag else ('fp32', 'rounded')
            assert out['%s_vs_%s' % (gemm, good)] < 1e-4, (name, gemm, out)
            if flag:     # ... and really that arithmetic: the unrounded operands' result is far (bf16 rounding: 3e-3 on
                # activations and data gradients, 6e-5 .. 1e-3 on weight gradients, whose sums over the pixels average it out)
                # (weight gradients: an order of magnitude -- the decoder's last layer sums 5 x 147k pixels, which averages the
                # rounding down to 4e-5 against 2.5e-6; activations and data gradients: 20 x)
                assert out['%s_vs_%s' % (gemm, other)] > (10 if gemm == 'wgrad' else 20) * out['%s_vs_%s' % (gemm, good)], (
                    name, gemm, 'not the bf16-operand arithmetic', out)
    H.record_parity('amp_stages_%s' % case_name, {'total_loss': float(losses['total_loss']), 'layers': rows})


def test_amp_training_throughput_property_run(lib):
    """Configs C / D, driver-visible: four clips of 3 x 384x384 frames, 2 objects, ResNet-50, K = 256 per step under config.AMP
    (graph replay after the two eager tuning steps).  Properties: every loss finite, the loss falls over the run on this
    fixed batch, the graph-replayed steps are deterministic in time order; the sustained clips/s go into the parity report."""
    import time
    from swem_amd import ops
    from swem_amd.train import SWEMTrainer
    tc = H.train_cases()
    case = dict(tc['cases']['r50k256'], b=4, valid=[[1, 1, 1]] * 4)
    cfg = O.make_cfg(**case['cfg'])
    frames, init_mask, label, valid = [t.to(DEV) for t in H.train_batch(case)]
    model, _ = H.make_model_and_sd(cfg, case['wseed'], DEV, pred_scale=tc['pred_scale'])
    tr = SWEMTrainer(dict(SOLVER=dict(tc['solver_cfg'], BASE_LR=1e-4, PRETRAIN_ITERS=[1000, 2000]), LOSS=tc['loss_cfg'], AMP=True),
                     model, lanes=4)
    torch.manual_seed(7)
    hist = []
    ops.AUTOTUNE = True
    for it in range(2):
        hist.append(float(tr.one_step(frames, init_mask, valid, label, 45 + it)[0]['total_loss']))
    ops.AUTOTUNE = False
    for it in range(2, 4):
        hist.append(float(tr.one_step(frames, init_mask, valid, label, 45 + it)[0]['total_loss']))
    assert tr._graph is not None
    torch.cuda.synchronize()
    n = 12
    t0 = time.perf_counter()
    last = None
    for it in range(n):
        last = tr.one_step(frames, init_mask, valid, label, 49 + it)[0]['total_loss']
    ops.spin_sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hist.append(float(last))
    assert all(math.isfinite(v) for v in hist), hist
    assert hist[-1] < hist[0], hist
    H.record_parity('amp_training_throughput', {'clips_per_s': 4 * n / dt, 'ms_per_step': 1e3 * dt / n, 'clips_per_step': 4,
                                                 'steps_timed': n, 'loss_first_last': [hist[0], hist[-1]],
                                                 'config': 'ResNet-50, K = 256, 3 x 384x384 frames, 2 objects, config.AMP, 4 lanes, graph replay'})
