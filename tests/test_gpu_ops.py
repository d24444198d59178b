"""GPU: every conv / pointwise kernel of libswem_hip.so, through the C ABI, against the same op in torch on the
CPU (fp32) with identical seeded inputs.  Tolerances are stated per test; index outputs must be exact."""
import pytest
import torch
import torch.nn.functional as F

from swem_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def back(t):
    return t.permute(0, 3, 1, 2).cpu()


def close(a, b, rtol, what=''):
    err = float((a - b).abs().max())
    ref = float(b.abs().max())
    assert err <= rtol * max(ref, 1e-6), '%s: max abs err %.3g vs max |ref| %.3g (rtol %.1g)' % (what, err, ref, rtol)


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, bias, bn, relu_in, relu_out, residual
    (1, 64, 30, 54, 64, 1, 1, False, True, False, True, False),      # bottleneck 1x1 + bn + relu
    (1, 64, 30, 54, 256, 1, 1, False, True, False, True, True),      # 1x1 + bn + residual + relu
    (2, 64, 33, 47, 128, 3, 2, True, True, False, True, False),      # strided 3x3, ragged size
    (1, 128, 60, 108, 128, 3, 1, True, False, True, False, True),    # ResBlock conv: relu on input + residual
    (1, 256, 15, 27, 512, 3, 1, True, False, False, False, False),   # tiny grid -> split-K path
    (1, 1024, 15, 27, 128, 3, 1, True, False, False, False, False),  # key_proj shape (config A size) split-K
    (3, 256, 60, 108, 256, 3, 1, True, False, True, False, False),   # big grid -> 128x128 tiles
    (1, 8, 64, 96, 64, 7, 2, True, True, False, True, False),        # stem 7x7/2, padded Cin
    (1, 4, 31, 45, 64, 7, 2, False, True, False, True, False),       # key stem (3+1 pad channels), odd size
    (1, 256, 30, 54, 128, 1, 2, False, True, False, False, False),   # downsample 1x1 stride 2
]


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'B%d_ci%d_%dx%d_co%d_k%d_s%d' % c[:7])
def test_conv2d_matches_torch(lib, case):
    B, Cin, H, W, Cout, k, stride, bias, bn, relu_in, relu_out, residual = case
    g = torch.Generator().manual_seed(hash(case) % 10000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1 if bias else None
    bnp = None
    if bn:
        bnp = (torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1,
               torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5)
    ref = F.conv2d(F.relu(x) if relu_in else x, w, b, stride=stride, padding=k // 2)
    if bn:
        ref = F.batch_norm(ref, bnp[2], bnp[3], bnp[0], bnp[1], False, 0.0, 1e-5)
    res = torch.randn(ref.shape, generator=g) if residual else None
    if residual:
        ref = ref + res
    if relu_out:
        ref = F.relu(ref)
    pack = ops.pack_conv(w.to(DEV), None if b is None else b.to(DEV), None if bnp is None else [t.to(DEV) for t in bnp],
                         stride, k // 2)
    y = ops.conv2d([nhwc(x)], pack, relu_in=relu_in, relu_out=relu_out, residual=None if res is None else nhwc(res))
    torch.cuda.synchronize()
    close(back(y), ref, 2e-5, 'conv2d')


def test_conv2d_three_sources_broadcast_and_glu(lib):
    """The fusion layer call (modules.py:291): cat[mem_out, qv (shared by all objects), S] -> f * sigmoid(a)."""
    g = torch.Generator().manual_seed(3)
    N, H, W, V, S2 = 3, 15, 27, 128, 64
    mem, qv, s = torch.randn(N, V, H, W, generator=g), torch.randn(1, V, H, W, generator=g), torch.rand(N, S2, H, W, generator=g)
    cin = 2 * V + S2
    wf, wa = torch.randn(V, cin, 3, 3, generator=g) * 0.03, torch.randn(V, cin, 3, 3, generator=g) * 0.03
    bf, ba = torch.randn(V, generator=g) * 0.1, torch.randn(V, generator=g) * 0.1
    xin = torch.cat([mem, qv.expand(N, -1, -1, -1), s], 1)
    ref = F.conv2d(xin, wf, bf, padding=1) * torch.sigmoid(F.conv2d(xin, wa, ba, padding=1))
    pack = ops.pack_glu(wf.to(DEV), bf.to(DEV), wa.to(DEV), ba.to(DEV))
    y = ops.conv2d([nhwc(mem), nhwc(qv), nhwc(s)], pack, batch=N)
    close(back(y), ref, 2e-5, 'glu conv')
    # plain conv over two sources with a broadcast residual (decoder skip shared by all objects)
    w = torch.randn(64, 2 * V, 3, 3, generator=g) * 0.03
    res = torch.randn(1, 64, H, W, generator=g)
    ref2 = F.conv2d(torch.cat([mem, qv.expand(N, -1, -1, -1)], 1), w, None, padding=1) + res
    y2 = ops.conv2d([nhwc(mem), nhwc(qv)], ops.pack_conv(w.to(DEV)), residual=nhwc(res), batch=N)
    close(back(y2), ref2, 2e-5, 'two-source conv + broadcast residual')


def test_conv2d_rejects_bad_shapes(lib):
    from swem_amd._lib import SwemHipError
    pack = ops.pack_conv(torch.randn(32, 8, 3, 3, device=DEV))
    with pytest.raises(SwemHipError, match='channels'):
        ops.conv2d([torch.zeros(1, 8, 8, 12, device=DEV)], pack)


def test_prep_inputs_and_maxpool(lib):
    g = torch.Generator().manual_seed(5)
    B, N, H, W = 1, 3, 37, 52
    frames = torch.rand(B, 3, H, W, generator=g)
    masks = torch.rand(B, N + 1, H, W, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    m3, s3 = ops._f3(mean), ops._f3(std)
    k = ops.prep_key_input(frames.to(DEV), m3, s3).cpu()
    img = (frames - mean) / std
    assert torch.allclose(k[..., :3].permute(0, 3, 1, 2), img, rtol=1e-6, atol=1e-6) and (k[..., 3] == 0).all()
    v = ops.prep_value_input(frames.to(DEV), masks.to(DEV), m3, s3, False).cpu()
    others = 1 - masks - masks[:, 0:1]
    for n in range(N):
        assert torch.allclose(v[n, ..., :3].permute(2, 0, 1), img[0], rtol=1e-6, atol=1e-6)
        assert torch.equal(v[n, ..., 3], masks[0, n + 1]) and torch.equal(v[n, ..., 4], others[0, n + 1])
        assert (v[n, ..., 5:] == 0).all()
    x = torch.randn(2, 64, 37, 51, generator=g)
    assert torch.equal(back(ops.maxpool(nhwc(x))), F.max_pool2d(x, 3, 2, 1))


def test_resampling_matches_torch(lib):
    g = torch.Generator().manual_seed(6)
    skip, low = torch.randn(2, 32, 30, 54, generator=g), torch.randn(2, 32, 15, 27, generator=g)
    ref = skip + F.interpolate(low, size=(30, 54), mode='bilinear', align_corners=False)
    close(back(ops.upsample_add(nhwc(skip), nhwc(low))), ref, 1e-6, 'upsample_add')
    ref = skip[:1] + F.interpolate(low, size=(30, 54), mode='bilinear', align_corners=False)
    close(back(ops.upsample_add(nhwc(skip[:1]), nhwc(low))), ref, 1e-6, 'upsample_add shared skip')
    # a skip image per GROUP of batch items (the objects of a clip share the clip's skip feature, several clips in the batch):
    # bit-identical to the per-object copy of the skip maps, with and without the fused output planes
    low6 = torch.randn(6, 32, 15, 27, generator=g)
    rep = skip.repeat_interleave(3, dim=0)
    full = ops.upsample_add(nhwc(rep), nhwc(low6))
    assert torch.equal(ops.upsample_add(nhwc(skip), nhwc(low6)), full)
    ops.SPLIT_HINTS.clear()
    try:
        ops.SPLIT_HINTS[full._swem_site] = {False: 2, True: 3}
        a, b = ops.upsample_add(nhwc(skip), nhwc(low6)), ops.upsample_add(nhwc(rep), nhwc(low6))
        assert torch.equal(a, full) and torch.equal(b, full)
        for relu in (False, True):
            n = a.__dict__['_swem_split'][relu][1]               # (planes written: 2 without the ReLU, 3 with it; the rest is not touched)
            assert n == b.__dict__['_swem_split'][relu][1] == (3 if relu else 2)
            assert torch.equal(a.__dict__['_swem_split'][relu][0][:n].view(torch.int16), b.__dict__['_swem_split'][relu][0][:n].view(torch.int16))
    finally:
        ops.SPLIT_HINTS.clear()
    with pytest.raises(Exception):
        ops.upsample_add(nhwc(skip), nhwc(low6[:5]))
    m = torch.rand(1, 3, 48, 85, generator=g)
    for size in ((48, 86), (30, 54), (96, 170), (48, 85)):
        assert torch.equal(ops.resize_planes(m.to(DEV), size, 'nearest').cpu(), F.interpolate(m, size=size, mode='nearest'))
        close(ops.resize_planes(m.to(DEV), size, 'bilinear').cpu(),
              F.interpolate(m, size=size, mode='bilinear', align_corners=False), 1e-6, 'bilinear %s' % (size,))


@pytest.mark.parametrize('hard_dtype', [torch.int64, torch.float32])
def test_mask_prep(lib, hard_dtype):
    """swem.py:79-84 at the DAVIS sizes: masks at 480x854, memory at 30x54."""
    g = torch.Generator().manual_seed(7)
    B, N, Ho, Wo, h, w = 1, 2, 480, 854, 30, 54
    idx = torch.randint(0, N + 1, (B, Ho, Wo), generator=g)
    hard = F.one_hot(idx, N + 1).permute(0, 3, 1, 2).contiguous().to(hard_dtype)
    soft = torch.rand(B, N + 1, 480, 864, generator=g)
    mh = F.interpolate(hard[:, 1:].float(), size=(h, w), mode='nearest')
    ms = F.interpolate(soft[:, 1:], size=(h, w), mode='bilinear')
    ref = torch.stack([(1 - mh) * (1 - ms), mh * ms], dim=2).view(B * N, 2, h * w)
    out = ops.mask_prep(hard.to(DEV), soft.to(DEV), h, w).cpu()
    close(out, ref, 1e-6, 'mask_prep')


def test_cbam_residual(lib):
    g = torch.Generator().manual_seed(8)
    B, Cc, H, W, hid = 2, 512, 15, 27, 32
    x = torch.randn(B, Cc, H, W, generator=g)
    w1, b1 = torch.randn(hid, Cc, generator=g) * 0.05, torch.randn(hid, generator=g) * 0.1
    w2, b2 = torch.randn(Cc, hid, generator=g) * 0.2, torch.randn(Cc, generator=g) * 0.1
    w7, b7 = torch.randn(1, 2, 7, 7, generator=g) * 0.1, torch.randn(1, generator=g) * 0.1

    def mlp(t):
        return F.linear(F.relu(F.linear(t.flatten(1), w1, b1)), w2, b2)
    att = mlp(F.avg_pool2d(x, (H, W))) + mlp(F.max_pool2d(x, (H, W)))
    xc = x * torch.sigmoid(att)[:, :, None, None]
    comp = torch.cat([xc.max(1, keepdim=True)[0], xc.mean(1, keepdim=True)], 1)
    ref = x + xc * torch.sigmoid(F.conv2d(comp, w7, b7, padding=3))
    y = ops.cbam_residual(nhwc(x), *[t.to(DEV).contiguous() for t in (w1, b1, w2, b2, w7, b7)])
    close(back(y), ref, 1e-5, 'x + CBAM(x)')


def test_decoder_heads(lib):
    g = torch.Generator().manual_seed(9)
    B, N, Cc, h4, w4, Ho, Wo = 1, 3, 256, 60, 108, 240, 427
    x = torch.randn(B * N, Cc, h4, w4, generator=g)
    w, b = torch.randn(1, Cc, 3, 3, generator=g) * 0.03, torch.randn(1, generator=g)
    ref_l = F.conv2d(F.relu(x), w, b, padding=1)
    lg = ops.pred_head(nhwc(x), w.permute(0, 2, 3, 1).contiguous().to(DEV), b.to(DEV))
    close(lg.cpu(), ref_l[:, 0], 2e-5, 'pred head')
    for valid in (None, torch.tensor([[1.0, 1.0, 0.0, 1.0]])):
        up = F.interpolate(ref_l, size=(Ho, Wo), mode='bilinear', align_corners=False)
        p = torch.sigmoid(up).view(B, N, Ho, Wo)
        if valid is not None:
            p = p * valid[:, 1:, None, None]
        newp = torch.cat([torch.prod(1 - p, dim=1, keepdim=True), p], 1).clamp(1e-7, 1 - 1e-7)
        ref_logits = torch.log(newp / (1 - newp))
        ref_prob = F.softmax(ref_logits, dim=1)
        logits, prob, amax = ops.decode_head(ref_l[:, 0].contiguous().to(DEV), B, N, (Ho, Wo),
                                             valid=None if valid is None else valid.to(DEV), want_argmax=True)
        close(logits.cpu(), ref_logits, 2e-5, 'aggregate logits')
        close(prob.cpu(), ref_prob, 2e-5, 'softmax')
        # the index map must be the argmax of the probabilities the kernel itself produced (bit exact)
        assert torch.equal(amax.cpu(), prob.cpu().argmax(1))
        am2, onehot = ops.argmax_onehot(prob)
        assert torch.equal(am2, amax)
        assert torch.equal(onehot.cpu(), F.one_hot(amax.cpu(), N + 1).permute(0, 3, 1, 2))
        agree = (amax.cpu() == ref_prob.argmax(1)).float().mean()
        assert agree > 0.9999, 'argmax agreement with the CPU reference %.6f' % agree


def test_transpose(lib):
    x = torch.randn(3, 405, 128)
    y = ops.transpose(x.to(DEV), ld=408).cpu()
    assert torch.equal(y[:, :, :405], x.transpose(1, 2)) and (y[:, :, 405:] == 0).all()


def test_bicubic_flip_lincomb_inject(lib):
    g = torch.Generator().manual_seed(12)
    x = torch.rand(1, 3, 48, 85, generator=g)
    for size in ((48, 86), (96, 172), (30, 54)):
        ref = F.interpolate(x, size=size, mode='bicubic', align_corners=False)
        close(ops.resize_planes(x.to(DEV), size, 'bicubic').cpu(), ref, 2e-6, 'bicubic %s' % (size,))
    assert torch.equal(ops.flip_w(x.to(DEV)).cpu(), torch.flip(x, dims=[-1]))
    a, b = torch.rand(2, 3, 40, 50, generator=g), torch.rand(2, 3, 40, 50, generator=g)
    assert torch.allclose(ops.lincomb(a.to(DEV), 0.5, b.to(DEV), 0.25).cpu(), 0.5 * a + 0.25 * b, rtol=1e-6, atol=1e-7)
    assert torch.allclose(ops.lincomb(a.to(DEV), 0.5).cpu(), 0.5 * a)
    # swem_evaluator.py:124-130
    prob = torch.rand(1, 3, 40, 50, generator=g)
    newm = torch.zeros(1, 3, 40, 50)
    newm[0, 1, 5:15, 5:20] = 1
    newm[0, 2, 20:30, 30:45] = 1
    ref = prob.clone()
    ref[newm[:, 1:].sum(1, keepdim=True).expand_as(ref) > 0] = 0
    ref = torch.cat([ref, newm[:, 1:]], 1)
    assert torch.equal(ops.inject_objects(prob.to(DEV), newm.to(DEV)).cpu(), ref)


@pytest.mark.parametrize('plan', [0x00011, 0x00021, 0x00022, 0x00211, 0x10011, 0x10021, 0x10022, 0x10321,
                                  0x110011, 0x110021, 0x110022, 0x210022, 0x310022, 0x210222,
                                  0x410011, 0x410021, 0x410022, 0x610022, 0x410221, 0x810022, 0x910022, 0x810222, 0x910322,
                                  0x10012, 0x110012, 0x410012, 0x10212],
                         ids=lambda p: 'math%d_wm%d_wn%d_ns%d' % (p >> 16, p & 15, (p >> 4) & 15, (p >> 8) & 255))
def test_conv2d_plans_and_math_modes(lib, plan):
    """Every tiling / K-split / math mode of the fast conv kernel gives the same convolution: fp32 MFMA and the
    bf16x6 mode (exact three-way bf16 split of both operands, six products, fp32 accumulate) to the same 2e-5."""
    g = torch.Generator().manual_seed(77)
    B, Cin, H, W, Cout = 2, 160, 21, 37, 192
    x = torch.randn(B, Cin, H, W, generator=g) * 3
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.03
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(B, Cout, H, W, generator=g)
    ref = F.relu(F.conv2d(F.relu(x), w, b, padding=1) + res)
    pack = ops.pack_conv(w.to(DEV), b.to(DEV))
    y = ops.conv2d([nhwc(x)], pack, relu_in=True, relu_out=True, residual=nhwc(res), plan=plan)
    close(back(y), ref, 2e-5, 'conv2d plan %#x' % plan)
    # three sources + GLU through the same plan (wn must be 2 for the gate pairing)
    if (plan >> 4) & 15 == 2:
        a, q, s = torch.randn(2, 64, 9, 14, generator=g), torch.randn(1, 64, 9, 14, generator=g), torch.rand(2, 32, 9, 14, generator=g)
        wf, wa = torch.randn(64, 160, 3, 3, generator=g) * 0.05, torch.randn(64, 160, 3, 3, generator=g) * 0.05
        bf, ba = torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1
        xin = torch.cat([a, q.expand(2, -1, -1, -1), s], 1)
        ref = F.conv2d(xin, wf, bf, padding=1) * torch.sigmoid(F.conv2d(xin, wa, ba, padding=1))
        y = ops.conv2d([nhwc(a), nhwc(q), nhwc(s)], ops.pack_glu(wf.to(DEV), bf.to(DEV), wa.to(DEV), ba.to(DEV)),
                       batch=2, plan=plan)
        close(back(y), ref, 2e-5, 'glu plan %#x' % plan)


@pytest.mark.parametrize('plan', [0x20011, 0x20021, 0x20022, 0x120021, 0x220022, 0x420011, 0x420021, 0x620022, 0x20221, 0x820022, 0x920022,
                                  0x4020021, 0xa20011, 0xd20022, 0x520022, 0x720022, 0xf20022, 0x1620022, 0x1020011, 0x20012, 0x420012], ids=lambda p: '%#x' % p)
def test_conv2d_plain_bf16_mode(lib, plan):
    """Math mode 2 (mixed-precision training, config.AMP): operands rounded to bf16 once, ONE MFMA product, fp32
    accumulate.  On bf16-representable operands it is the fp32 convolution up to summation order; on general ones it
    is the convolution of the rounded operands (what torch.autocast computes for this layer)."""
    g = torch.Generator().manual_seed(78)
    B, Cin, H, W, Cout = 2, 160, 21, 37, 192
    x = torch.randn(B, Cin, H, W, generator=g) * 3
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.03
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(B, Cout, H, W, generator=g)
    xr, wr = x.bfloat16().float(), w.bfloat16().float()
    ref = F.relu(F.conv2d(F.relu(xr), wr, b, padding=1) + res)
    y = ops.conv2d([nhwc(xr)], ops.pack_conv(wr.to(DEV), b.to(DEV)), relu_in=True, relu_out=True, residual=nhwc(res),
                   plan=plan)
    close(back(y), ref, 2e-5, 'bf16 conv on representable operands, plan %#x' % plan)
    y = ops.conv2d([nhwc(x)], ops.pack_conv(w.to(DEV), b.to(DEV)), relu_in=True, relu_out=True, residual=nhwc(res),
                   plan=plan)
    close(back(y), ref, 2e-5, 'bf16 conv rounds its operands to nearest-even, plan %#x' % plan)
    full = F.relu(F.conv2d(F.relu(x), w, b, padding=1) + res)
    err = float((back(y) - full).abs().max() / full.abs().max())
    assert 1e-5 < err < 1e-2, err             # really one bf16 product (not the exact six), and no worse than bf16


@pytest.mark.parametrize('plan', [0x00011, 0x00022, 0x00211, 0x10021, 0x10022, 0x10321, 0x30011, 0x30021, 0x30022, 0x230022,
                                  0x430011, 0x430021, 0x630022, 0x30221, 0x830022, 0x4030021, 0x8030022, 0xb30011, 0xd30022, 0xe30022,
                                  0x530022, 0x730022, 0xf30022, 0x1630022, 0x1030011, 0x1530022, 0x30012, 0x430012, 0x30212],
                         ids=lambda p: '%#x' % p)
def test_conv2d_fused_output_planes(lib, plan):
    """The conv epilogues (32x32 and 16x16 accumulator layouts, the split-K reduce kernel, the tail split; fp32, bf16x6 and
    bf16x3 kernels) write the output's bf16 planes themselves once a consumer has asked for them (ops.SPLIT_HINTS): y is
    unchanged and the planes are BIT-IDENTICAL to swem_split_bf16x3_f32 on y -- both variants (planes of y, planes of
    relu(y)), two or three planes."""
    g = torch.Generator().manual_seed(81)
    B, Cin, H, W, Cout = 2, 160, 21, 37, 192
    x = nhwc(torch.randn(B, Cin, H, W, generator=g) * 3)
    res = nhwc(torch.randn(B, Cout, H, W, generator=g))
    pack = ops.pack_conv((torch.randn(Cout, Cin, 3, 3, generator=g) * 0.03).to(DEV), (torch.randn(Cout, generator=g) * 0.1).to(DEV))
    ops.SPLIT_HINTS.clear()
    y0 = ops.conv2d([x], pack, relu_in=True, residual=res, plan=plan)
    assert '_swem_split' not in y0.__dict__
    M = B * H * W

    def split(t, relu):
        sp = torch.empty((3, M * Cout), dtype=torch.bfloat16, device=DEV)
        lib_call = __import__('swem_amd')._lib.call
        lib_call('swem_split_bf16x3_f32', ops._stream(), t.data_ptr(), sp.data_ptr(), M, Cout, int(relu))
        return sp
    try:
        ops.SPLIT_HINTS[y0._swem_site] = {False: 3, True: 2}
        y1 = ops.conv2d([x], pack, relu_in=True, residual=res, plan=plan)
        assert torch.equal(y1, y0)
        got = y1.__dict__['_swem_split']
        assert got[False][1] == 3 and got[True][1] == 2
        assert torch.equal(got[False][0].view(torch.int16), split(y0, False).view(torch.int16))
        assert torch.equal(got[True][0][:2].view(torch.int16), split(y0, True)[:2].view(torch.int16))
        # a consumer finds them (no split launch), one that needs the third plane of the relu variant re-splits
        fused_relu = got[True][0]
        assert ops.presplit(y1, False, 3) is got[False][0] and ops.presplit(y1, True, 2) is fused_relu
        assert ops.presplit(y1, True, 3) is not fused_relu
    finally:
        ops.SPLIT_HINTS.clear()


@pytest.mark.parametrize('value', [False, True], ids=['key_stem', 'value_stem'])
def test_space_to_depth_stem_equals_the_7x7_stem(lib, value):
    """conv1 of the encoders (networks.py:115-117,161; mod_resnet.py conv1 7x7 / stride 2 / pad 3) as a 4x4 / stride-1
    convolution on the space-to-depth input (swem_prep_input_s2d_f32 + ops.pack_stem_s2d): the same sums in another order --
    equal to the 7x7 form to fp32 rounding on the fp32 kernels, to the bf16x3 tolerance on the pre-split kernel (whose planes
    the prep kernel writes itself)."""
    g = torch.Generator().manual_seed(3)
    B, N, Hh, Ww = 1, 2, 44, 76
    frame = torch.rand(B, 3, Hh, Ww, generator=g).to(DEV)
    masks = torch.softmax(torch.randn(B, N + 1, Hh, Ww, generator=g) * 3, 1).to(DEV)
    ci = 5 if value else 3
    w = (torch.randn(64, ci, 7, 7, generator=g) * 0.05).to(DEV)
    bias = (torch.randn(64, generator=g) * 0.1).to(DEV) if value else None
    bn = [t.to(DEV) for t in (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1,
                              torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5)]
    mean3, std3 = ops._f3(torch.tensor([0.485, 0.456, 0.406])), ops._f3(torch.tensor([0.229, 0.224, 0.225]))
    if value:
        x7 = ops.prep_value_input(frame, masks, mean3, std3, False)
        xs = ops.prep_input_s2d(frame, masks, mean3, std3, False)
    else:
        x7 = ops.prep_key_input(frame, mean3, std3)
        xs = ops.prep_input_s2d(frame, None, mean3, std3)
    ref = ops.conv2d([x7], ops.pack_conv(w, bias, bn, 2, 3, cin_pad=8 if value else 4), relu_out=True)
    pk = ops.pack_stem_s2d(w, bias, bn)
    got = ops.conv2d([xs], pk, relu_out=True, plan=0x11)                      # fp32 kernel
    assert got.shape == ref.shape
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 2e-6 * scale
    # the planes the prep kernel attached are the split of the fp32 image
    sp = torch.empty((3, xs.numel()), dtype=torch.bfloat16, device=DEV)
    __import__('swem_amd')._lib.call('swem_split_bf16x3_f32', ops._stream(), xs.data_ptr(), sp.data_ptr(), xs.numel() // 32, 32, 0)
    assert torch.equal(xs.__dict__['_swem_split'][False][0].view(torch.int16), sp.view(torch.int16))
    got3 = ops.conv2d([xs], pk, relu_out=True, plan=0x30011)                  # bf16x3 on the pre-split kernel
    assert float((got3 - ref).abs().max()) < 3e-5 * scale


def _split_of(t, M, C, relu, npl=3):
    if npl == ops.PLANES_F16:
        sp = torch.empty((2, M * C), dtype=torch.float16, device=DEV)
        __import__('swem_amd')._lib.call('swem_split_f16x2_f32', ops._stream(), t.data_ptr(), sp.data_ptr(), M, C, int(relu), 0)
        return sp
    sp = torch.empty((3, M * C), dtype=torch.bfloat16, device=DEV)
    __import__('swem_amd')._lib.call('swem_split_bf16x3_f32', ops._stream(), t.data_ptr(), sp.data_ptr(), M, C, int(relu))
    return sp


def _check_fused_planes(make, want, M, C):
    """make() runs a producer; with a consumer's request recorded under its site the same call returns the same y plus
    planes bit-identical to swem_split_bf16x3_f32 on y."""
    ops.SPLIT_HINTS.clear()
    y0 = make()
    assert '_swem_split' not in y0.__dict__
    try:
        ops.SPLIT_HINTS[y0._swem_site] = dict(want)
        y1 = make()
        assert torch.equal(y1, y0)
        got = y1.__dict__['_swem_split']
        for relu, n in want.items():
            key, rows = ops._pkey(relu, n), (2 if n == ops.PLANES_F16 else n)
            assert got[key][1] == n
            assert torch.equal(got[key][0][:rows].view(torch.int16), _split_of(y0, M, C, relu, n)[:rows].view(torch.int16))
            assert ops.presplit(y1, relu, n) is got[key][0]
    finally:
        ops.SPLIT_HINTS.clear()


@pytest.mark.parametrize('plan', [0x00022, 0x10022, 0x30022, 0x230022, 0x630022, 0x630122, 0x30222, 0x830022, 0x8030022,
                                  0xd30022, 0x530022, 0x1630022], ids=lambda p: '%#x' % p)
def test_conv2d_glu_fused_output_planes(lib, plan):
    """modules.py:25-26 (f * sigmoid(a)) with the gated output's bf16 planes written by the conv's own epilogue (32x32 and
    16x16 accumulator layouts, separate and fused K-split reduce): the decoder's first ResBlock consumes it pre-split."""
    g = torch.Generator().manual_seed(83)
    a, q, s = torch.randn(2, 64, 19, 23, generator=g), torch.randn(1, 64, 19, 23, generator=g), torch.rand(2, 32, 19, 23, generator=g)
    wf, wa = torch.randn(96, 160, 3, 3, generator=g) * 0.05, torch.randn(96, 160, 3, 3, generator=g) * 0.05
    bf, ba = torch.randn(96, generator=g) * 0.1, torch.randn(96, generator=g) * 0.1
    xin = torch.cat([a, q.expand(2, -1, -1, -1), s], 1)
    ref = F.conv2d(xin, wf, bf, padding=1) * torch.sigmoid(F.conv2d(xin, wa, ba, padding=1))
    pack = ops.pack_glu(wf.to(DEV), bf.to(DEV), wa.to(DEV), ba.to(DEV))
    srcs = [nhwc(a), nhwc(q), nhwc(s)]
    close(back(ops.conv2d(srcs, pack, batch=2, plan=plan)), ref, 2e-5, 'glu plan %#x' % plan)
    _check_fused_planes(lambda: ops.conv2d(srcs, pack, batch=2, plan=plan), {False: 3, True: 2}, 2 * 19 * 23, 96)
    _check_fused_planes(lambda: ops.conv2d(srcs, pack, batch=2, plan=plan), {False: ops.PLANES_F16, True: 3}, 2 * 19 * 23, 96)


def test_maxpool_and_cbam_fused_output_planes(lib):
    """mod_resnet.py:123 and networks.py:46-47 with the result's bf16 planes written by the producing launch."""
    g = torch.Generator().manual_seed(6)
    x = nhwc(torch.randn(2, 72, 21, 37, generator=g))
    _check_fused_planes(lambda: ops.maxpool(x), {False: 2, True: 3}, 2 * 11 * 19, 72)
    _check_fused_planes(lambda: ops.maxpool(x), {False: ops.PLANES_F16, True: ops.PLANES_F16}, 2 * 11 * 19, 72)
    B, Cc, H, W, hid = 2, 136, 15, 27, 16
    x = nhwc(torch.randn(B, Cc, H, W, generator=g))
    par = [t.to(DEV).contiguous() for t in (torch.randn(hid, Cc, generator=g) * 0.05, torch.randn(hid, generator=g) * 0.1,
                                            torch.randn(Cc, hid, generator=g) * 0.2, torch.randn(Cc, generator=g) * 0.1,
                                            torch.randn(1, 2, 7, 7, generator=g) * 0.1, torch.randn(1, generator=g) * 0.1)]
    _check_fused_planes(lambda: ops.cbam_residual(x, *par), {True: 2}, B * H * W, Cc)
    _check_fused_planes(lambda: ops.cbam_residual(x, *par), {False: 3, True: 3}, B * H * W, Cc)
    _check_fused_planes(lambda: ops.cbam_residual(x, *par), {False: ops.PLANES_F16}, B * H * W, Cc)


@pytest.mark.parametrize('plan', [0x00011, 0x10022, 0x70011, 0x70022, 0x670022, 0x70221, 0x870022, 0x4070021, 0xa70211, 0x70012],
                         ids=lambda p: '%#x' % p)
def test_conv2d_fused_output_planes_f16(lib, plan):
    """The fp16 (hi, mid) pair of a conv output from the conv's own epilogue (every kernel family, fused K-split, tail split,
    residual + ReLU variants): bit-identical to swem_split_f16x2_f32 on y; mixed with a bf16 request for the other variant."""
    g = torch.Generator().manual_seed(81)
    B, Cin, H, W, Cout = 2, 160, 21, 37, 192
    x = nhwc(torch.randn(B, Cin, H, W, generator=g) * 3)
    res = nhwc(torch.randn(B, Cout, H, W, generator=g))
    pack = ops.pack_conv((torch.randn(Cout, Cin, 3, 3, generator=g) * 0.03).to(DEV), (torch.randn(Cout, generator=g) * 0.1).to(DEV))
    make = lambda: ops.conv2d([x], pack, relu_in=True, residual=res, plan=plan)
    _check_fused_planes(make, {False: ops.PLANES_F16, True: ops.PLANES_F16}, B * H * W, Cout)
    _check_fused_planes(make, {False: 3, True: ops.PLANES_F16}, B * H * W, Cout)
    # a consumer of the other format does not find them: it splits for itself, into its own cache entry
    ops.SPLIT_HINTS.clear()
    y0 = make()
    try:
        ops.SPLIT_HINTS[y0._swem_site] = {False: ops.PLANES_F16}
        y1 = make()
        h = y1.__dict__['_swem_split'][ops._pkey(False, ops.PLANES_F16)][0]
        b3 = ops.presplit(y1, False, 2)
        assert b3.dtype == torch.bfloat16 and ops.presplit(y1, False, ops.PLANES_F16) is h and ops.presplit(y1, False, 3) is b3
    finally:
        ops.SPLIT_HINTS.clear()


def test_conv2d_planes_only_output(lib):
    """ops.conv2d(planes_only=True) (the conv1 -> conv2 -> conv3 chains inside a block: mod_resnet.py:58-113, networks.py:22-32):
    once the single consumer has asked for the planes the producer leaves the fp32 map out (y = NULL through the C ABI) and
    the consumer's result is BIT-IDENTICAL to the one computed from a fully written producer; the promise is withdrawn by any
    plan change until the consumer asks again; every other use of such a tensor raises."""
    from swem_amd import _lib
    g = torch.Generator().manual_seed(91)
    B, Cin, H, W, Cmid, Cout = 2, 64, 21, 37, 64, 96
    x = nhwc(torch.randn(B, Cin, H, W, generator=g) * 2)
    p1 = ops.pack_conv((torch.randn(Cmid, Cin, 3, 3, generator=g) * 0.05).to(DEV), (torch.randn(Cmid, generator=g) * 0.1).to(DEV))
    p2 = ops.pack_conv((torch.randn(Cout, Cmid, 3, 3, generator=g) * 0.05).to(DEV), (torch.randn(Cout, generator=g) * 0.1).to(DEV))
    book = ops.PlanBook(fallback=0x30011)          # both layers on the pre-split kernel (bf16x3)

    def chain(only):
        y = ops.conv2d([x], p1, relu_out=True, planes_only=only)
        return y, ops.conv2d([y], p2, relu_out=True)
    with ops.use_book(book):
        y_ref, z_ref = chain(False)                 # frame 1: nothing known yet, y written, the consumer splits it
        y_a, z_a = chain(False)                     # frame 2, ordinary fused planes: y written AND planes
        assert '_swem_split' in y_a.__dict__ and not y_a.__dict__.get('_swem_planes_only') and torch.equal(y_a, y_ref)
        y_b, z_b = chain(True)                      # the same with the promise: planes only
        assert y_b.__dict__.get('_swem_planes_only') and torch.equal(z_b, z_ref) and torch.equal(z_a, z_ref)
        assert torch.equal(y_b.__dict__['_swem_split'][False][0][:2].view(torch.int16),
                           y_a.__dict__['_swem_split'][False][0][:2].view(torch.int16))
        # no fp32 map exists behind it: the tensor is ONE NaN expanded to the shape (nothing allocated, nothing uninitialised)
        assert not y_b.is_contiguous() and y_b.untyped_storage().nbytes() == 4 and bool(torch.isnan(y_b).all())
        with pytest.raises(_lib.SwemHipError):
            ops.maxpool(y_b)                        # an fp32 consumer: loud
        with pytest.raises(_lib.SwemHipError):
            ops.conv2d([y_b], p2, relu_in=True)     # planes its producer did not write (the ReLU variant)
        with pytest.raises(_lib.SwemHipError):
            ops.conv2d([y_b], p2, relu_out=True, plan=0x00011)   # a convolution on the fp32 path
        book.fallback = 0x30021                     # a plan change: the consumer might now take another path
        y_c, z_c = chain(True)
        assert not y_c.__dict__.get('_swem_planes_only') and torch.equal(y_c, y_ref)
        y_d, z_d = chain(True)                      # ... it asked again: planes only
        assert y_d.__dict__.get('_swem_planes_only') and torch.equal(z_d, z_c)
        with ops.conv_math((3,)):                   # a mode change withdraws the promise too, for one frame
            y_e, _ = chain(True)
            assert not y_e.__dict__.get('_swem_planes_only')
        chain(True)
        assert chain(True)[0].__dict__.get('_swem_planes_only')
        try:                                        # ... and so does switching the tuner on (its candidates read the fp32 map)
            ops.AUTOTUNE = True
            assert not chain(True)[0].__dict__.get('_swem_planes_only')
        finally:
            ops.AUTOTUNE = False
        book.fallback = 0x00011                     # the consumer on the fp32 kernels: it never asks, the map is always written
        for _ in range(3):
            y_f, z_f = chain(True)
            assert not y_f.__dict__.get('_swem_planes_only')
        close(z_f, z_ref, 2e-5, 'fp32 path after the planes-only frames')


def test_upsample_add_fused_output_planes(lib):
    """networks.py:193-194 with the result's bf16 planes written by the same launch (the decoder's ResBlocks consume it
    pre-split, with and without their input ReLU): y bit-identical to the plain kernel, planes bit-identical to
    swem_split_bf16x3_f32 on y."""
    g = torch.Generator().manual_seed(5)
    B, C, Hl, Wl, Ho, Wo = 2, 72, 15, 27, 30, 54
    low = nhwc(torch.randn(B, C, Hl, Wl, generator=g))
    skip = nhwc(torch.randn(1, C, Ho, Wo, generator=g))
    ops.SPLIT_HINTS.clear()
    y0 = ops.upsample_add(skip, low)
    assert '_swem_split' not in y0.__dict__
    M = B * Ho * Wo
    try:
        ops.SPLIT_HINTS[y0._swem_site] = {False: 2, True: 3}
        y1 = ops.upsample_add(skip, low)
        assert torch.equal(y1, y0)
        got = y1.__dict__['_swem_split']
        for relu in (False, True):
            sp = torch.empty((3, M * C), dtype=torch.bfloat16, device=DEV)
            __import__('swem_amd')._lib.call('swem_split_bf16x3_f32', ops._stream(), y0.data_ptr(), sp.data_ptr(), M, C, int(relu))
            n = got[relu][1]
            assert n == (3 if relu else 2)
            assert torch.equal(got[relu][0][:n].view(torch.int16), sp[:n].view(torch.int16))
        fused = got[True][0]
        assert ops.presplit(y1, True, 3) is fused
        # an in-place change of the tensor invalidates the planes a producer attached: the next consumer splits again
        y1.mul_(2.0)
        again = ops.presplit(y1, True, 3)
        assert again is not fused
        sp = torch.empty((3, M * C), dtype=torch.bfloat16, device=DEV)
        __import__('swem_amd')._lib.call('swem_split_bf16x3_f32', ops._stream(), y1.data_ptr(), sp.data_ptr(), M, C, 1)
        assert torch.equal(again.view(torch.int16), sp.view(torch.int16))
    finally:
        ops.SPLIT_HINTS.clear()


@pytest.mark.parametrize('plan', [0x30011, 0x30021, 0x30022, 0x130021, 0x230022, 0x430011, 0x430021, 0x630022, 0x30221, 0x830022,
                                  0x930022, 0x4030021, 0xa30011, 0xb30021, 0xa30022, 0xc30022, 0xd30022, 0xe30022, 0xa30211,
                                  0x530022, 0x730022, 0xf30022, 0x530222, 0x4530022,
                                  0x1630022, 0x1030011, 0x1530022, 0x1030021, 0x1230022, 0x1830022, 0x1430011,
                                  0x30012, 0x130012, 0x430012, 0xa30012, 0xb30012, 0x30212, 0x4030012], ids=lambda p: '%#x' % p)
def test_conv2d_bf16x3_mode(lib, plan):
    """Math mode 3 ("bf16x3"): each operand is taken as hi + mid (two bf16 terms = 16 significant bits) and the product is
    hi.hi + hi.mid + mid.hi in fp32 -- half the matrix-core work of bf16x6.  On operands that HAVE only 16 significant bits it
    is the fp32 convolution up to the dropped mid.mid term (2^-16 of a product); on general fp32 operands the error is that of
    truncating the operands to 16 bits: measured here against the fp32 convolution and held to 2e-5 of the output's range."""
    g = torch.Generator().manual_seed(79)
    B, Cin, H, W, Cout = 2, 160, 21, 37, 192
    x = torch.randn(B, Cin, H, W, generator=g) * 3
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.03
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(B, Cout, H, W, generator=g)

    def two_terms(t):                       # hi + mid: what planes 0 and 1 of the exact split hold
        hi = t.bfloat16().float()
        return hi + (t - hi).bfloat16().float()
    x2, w2 = two_terms(x), two_terms(w)
    ref2 = F.relu(F.conv2d(F.relu(x2).double(), w2.double(), b.double(), padding=1) + res.double()).float()
    y = ops.conv2d([nhwc(x)], ops.pack_conv(w.to(DEV), b.to(DEV)), relu_in=True, relu_out=True, residual=nhwc(res), plan=plan)
    close(back(y), ref2, 1e-5, 'bf16x3 conv = conv of the 16-bit operands, plan %#x' % plan)
    full = F.relu(F.conv2d(F.relu(x).double(), w.double(), b.double(), padding=1) + res.double()).float()
    err = float((back(y) - full).abs().max() / full.abs().max())
    y6 = ops.conv2d([nhwc(x)], ops.pack_conv(w.to(DEV), b.to(DEV)), relu_in=True, relu_out=True, residual=nhwc(res),
                    plan=(plan & ~0x30000) | 0x10000)
    err6 = float((back(y6) - full).abs().max() / full.abs().max())
    print('plan %#x: bf16x3 error %.2e of the output range (bf16x6: %.2e)' % (plan, err, err6))
    assert err6 < 2e-6 < err < 2e-5, (err, err6)     # really the three-product mode, and within its error budget


def _two_f16_terms(t, scale=None):
    """hi + mid as the fp16 pair holds them (swem_split_f16x2_f32; filters: per output column scaled by a power of two)."""
    s = 1.0 if scale is None else scale
    ts = (t * s).float()
    hi = ts.half().float()
    return (hi + (ts - hi).half().float()) / s


@pytest.mark.parametrize('plan', [0x70011, 0x70021, 0x70022, 0x170021, 0x270022, 0x470011, 0x470021, 0x670022, 0x70221, 0x870022,
                                  0x970022, 0x4070021, 0xa70011, 0xb70021, 0xa70022, 0xc70022, 0xd70022, 0xe70022, 0xa70211,
                                  0x570022, 0x770022, 0xf70022, 0x570222, 0x70012, 0x470012, 0x70212, 0x4070012],
                         ids=lambda p: '%#x' % p)
def test_conv2d_f16x3_mode(lib, plan):
    """Math mode 7 ("f16x3", round 4): the bf16x3 kernel on fp16 (hi, mid) planes -- each operand carries 22-23 significant bits
    and the product is hi.hi + hi.mid + mid.hi in fp32.  (1) It IS the convolution of those two-term operands (the dropped
    mid.mid term is 2^-24 of a product).  (2) Against the float64 convolution of the fp32 operands its error is that of the
    fp32 kernels -- measured beside the exact fp32 MFMA kernel and bf16x6 on the same inputs, and an order of magnitude below
    bf16x3's."""
    g = torch.Generator().manual_seed(79)
    B, Cin, H, W, Cout = 2, 160, 21, 37, 192
    x = torch.randn(B, Cin, H, W, generator=g) * 3
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.03
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(B, Cout, H, W, generator=g)
    wscale = torch.exp2(13 - torch.floor(torch.log2(w.abs().amax(dim=(1, 2, 3), keepdim=True))))
    x2, w2 = _two_f16_terms(F.relu(x)), _two_f16_terms(w, wscale)
    ref2 = F.relu(F.conv2d(x2.double(), w2.double(), b.double(), padding=1) + res.double()).float()
    full = F.relu(F.conv2d(F.relu(x).double(), w.double(), b.double(), padding=1) + res.double()).float()

    def run(pl):
        y = ops.conv2d([nhwc(x)], ops.pack_conv(w.to(DEV), b.to(DEV)), relu_in=True, relu_out=True, residual=nhwc(res), plan=pl)
        return back(y), float((back(y) - full).abs().max() / full.abs().max())
    y7, err7 = run(plan)
    close(y7, ref2, 2e-6, 'f16x3 conv = conv of the two-term fp16 operands, plan %#x' % plan)
    tile = plan & 0xffff if (plan & 0xff) != 0x12 else (plan & 0xff00) | 0x11     # (the 128x64 tile is a pre-split tile)
    _, err32 = run(tile)
    _, err6 = run((plan & ~0x70000) | 0x10000)
    _, err3 = run((plan & ~0x70000) | 0x30000)
    print('plan %#x: error against float64, of the output range: f16x3 %.2e, fp32 MFMA %.2e, bf16x6 %.2e, bf16x3 %.2e'
          % (plan, err7, err32, err6, err3))
    assert err7 < 1.5 * max(err32, err6) and err7 < 2e-6 and err3 > 3 * err7, (err7, err32, err6, err3)


def test_f16x3_small_and_large_operands(lib):
    """The fp16 pair across the range (include/swem_hip.h, swem_split_f16x2_f32).  Filters of any magnitude, columns decades
    apart: scaled per output column, always full precision.  Activations: wherever most of a tensor's energy sits above 2^-2
    (every conv operand of the bench configuration: rms 0.56-4.5, tools/act_ranges.py) the error against float64 is the fp32
    kernel's; a tensor of TINY values degrades gracefully -- `mid` is subnormal, the absolute error per element is bounded by
    2^-25, i.e. per output by 2^-25 * sum |w| -- and never overflows below 65520."""
    g = torch.Generator().manual_seed(5)
    B, Cin, H, W, Cout = 1, 64, 24, 40, 64
    w0 = torch.randn(Cout, Cin, 3, 3, generator=g)
    for xs, ws in ((1e-3, 1e-4), (0.02, 0.03), (0.5, 0.03), (1.0, 30.0), (2e3, 1e-2), (3e4, 1e-6)):
        x = (torch.randn(B, Cin, H, W, generator=g).abs() * xs).clamp(max=6.0e4)
        w = w0 * ws * torch.logspace(-3, 0, Cout).view(-1, 1, 1, 1)          # columns three decades apart
        full = F.conv2d(x.double(), w.double(), None, padding=1)
        col = full.abs().amax(dim=(0, 2, 3), keepdim=True)                   # per-column range: every filter on its own scale
        errs = {}
        for name, pl in (('f16x3', 0x70011), ('fp32', 0x11), ('bf16x3', 0x30011)):
            y = back(ops.conv2d([nhwc(x)], ops.pack_conv(w.to(DEV)), plan=pl)).double()
            assert torch.isfinite(y).all()
            errs[name] = float(((y - full).abs() / col).max())
        # the documented bound of the subnormal regime, per column, on that column's scale
        sub = float((2.0 ** -25 * w.double().abs().sum(dim=(1, 2, 3)).view(1, -1, 1, 1) / col).max())
        print('x ~ %.0e, w ~ %.0e: %s, subnormal-regime bound %.2e' % (xs, ws, errs, sub))
        assert errs['f16x3'] < max(3 * errs['fp32'], 2e-6) + sub, (xs, ws, errs, sub)
        if xs >= 0.5:
            assert errs['f16x3'] < max(3 * errs['fp32'], 2e-6) and 3 * errs['f16x3'] < errs['bf16x3'], (xs, ws, errs)


def _expect_range_fault():
    with pytest.raises(ops.SwemRangeError, match='fp16 range'):
        ops.check_faults()
    ops.check_faults()                 # cleared by the raise: clean again


def test_f16x3_out_of_range_raises_through_a_relu_epilogue(lib):
    """An activation beyond the fp16 range gives inf / NaN planes; WITHOUT a ReLU the consumer's output is NaN, but a ReLU
    epilogue (v_max_f32 0, x) maps that NaN to 0 -- a finite, wrong feature map.  What makes it loud is
    the producer: whoever writes a value with |x| >= 65520 into an fp16 pair sets SWEM_FAULT_RANGE in the caller's sticky
    fault word and ops.check_faults() raises SwemRangeError.  Every producer is exercised: the stand-alone split (with and
    without its input ReLU), a conv epilogue (both plane variants, K-split reducer included), max-pool, upsample-add, CBAM,
    the space-to-depth input, matching's value planes."""
    ops.check_faults()
    x = torch.ones(1, 32, 8, 8)
    x[0, 3, 2, 2] = 1.0e5
    w = torch.ones(32, 32, 1, 1) * 0.01
    pack = ops.pack_conv(w.to(DEV))
    # (1) the split launch of an fp32 map, consumer WITH a ReLU epilogue: the wrong pixel comes out as a clean 0
    y = back(ops.conv2d([nhwc(x)], pack, relu_out=True, plan=0x70011))
    assert torch.isfinite(y).all() and float(y[0, :, 2, 2].abs().max()) == 0.0 and float(y[0, 0, 0, 0]) > 0.3
    _expect_range_fault()
    # ... without the ReLU it is NaN (and faults all the same)
    y = back(ops.conv2d([nhwc(x)], pack, plan=0x70011))
    assert not torch.isfinite(y[0, :, 2, 2]).any() and torch.isfinite(y[0, :, 0, 0]).all()
    _expect_range_fault()
    # a large NEGATIVE value under the consumer's input ReLU is a plain 0 in the reference too: no fault, right result
    xn = torch.ones(1, 32, 8, 8)
    xn[0, 3, 2, 2] = -1.0e5
    y = back(ops.conv2d([nhwc(xn)], pack, relu_in=True, plan=0x70011))
    ops.check_faults()
    close(y, F.conv2d(F.relu(xn), w), 1e-6, 'negative under relu')
    # the bf16 arithmetics have the fp32 range: no fault, finite right result
    for pl in (0x10011, 0x30011, 0x11):
        y = back(ops.conv2d([nhwc(x)], pack, relu_out=True, plan=pl))
        ops.check_faults()
        close(y, F.relu(F.conv2d(x, w)), 1e-2 if pl == 0x30011 else 1e-5, 'plan %#x' % pl)
    # (2) a conv EPILOGUE that writes its output as an fp16 pair (the fused operand split): inputs in range, output not
    big = ops.pack_conv((torch.ones(32, 32, 1, 1) * 100.0).to(DEV))
    xin = nhwc(torch.ones(1, 32, 8, 8) * 30.0)            # y = 96000
    site_flags = []
    for relu_out, plan in ((True, 0x70011), (False, 0x70011), (False, 0x11), (True, 0x70211)):
        with ops.use_book(ops.PlanBook()):
            for it in range(2):                           # second pass: the producer writes the planes its consumer asked for
                y1 = ops.conv2d([xin], big, relu_out=relu_out, plan=plan)
                ops.conv2d([y1], pack, relu_out=True, plan=0x70011)
                if it == 0:
                    _expect_range_fault()                 # (first pass: the stand-alone split of y1 faults)
            assert y1.__dict__.get('_swem_split'), 'the producer did not write planes'
            _expect_range_fault()
    # (3) the pointwise producers
    with ops.use_book(ops.PlanBook()):
        xb = nhwc(torch.ones(1, 32, 16, 16))
        xb[0, 5, 5, 7] = 7.0e4
        skip, low = nhwc(torch.zeros(1, 32, 16, 16)), nhwc(torch.ones(1, 32, 8, 8) * 7.0e4)
        for it in range(2):
            for prod in (lambda: ops.maxpool(xb), lambda: ops.upsample_add(skip, low)):
                t = prod()
                ops.conv2d([t], pack, relu_out=True, plan=0x70011)
                if it:
                    assert t.__dict__.get('_swem_split')
                _expect_range_fault()
    # the stand-alone split with relu: +1e5 faults, -1e5 does not
    for val, want in ((1.0e5, True), (-1.0e5, False), (float('inf'), True), (float('nan'), True), (65519.0, False), (65520.0, True)):
        t = torch.zeros(64, 32, device=DEV)
        t[3, 9] = val
        sp = torch.empty((2, 64 * 32), dtype=torch.float16, device=DEV)
        __import__('swem_amd')._lib.call('swem_split_f16x2_f32', ops._stream(), t.data_ptr(), sp.data_ptr(), 64, 32, 1,
                                         ops.fault_word(t.device).data_ptr())
        if want:
            _expect_range_fault()
        else:
            ops.check_faults()


def test_nan_does_not_leave_through_a_relu_plane_without_a_fault(lib):
    """fmaxf(NaN, 0) = 0 -- a producer that writes the fp16 pair of relu(y) tested the value AFTER the ReLU, so a NaN
    (or -inf) in y left as a clean 0 without setting SWEM_FAULT_RANGE, and with a planes-only output nothing else carried it.
    The producers now look at the value before the ReLU: a conv epilogue (ReLU output, and the ReLU plane variant its consumer
    asks for), the pointwise plane writer (max-pool's consumer with an input ReLU) and the training step's frozen-BN stage."""
    from swem_amd import _lib as L
    ops.check_faults()
    w = torch.ones(32, 32, 1, 1) * 0.01
    pack = ops.pack_conv(w.to(DEV))
    res = nhwc(torch.zeros(1, 32, 8, 8))
    res[0, 2, 2, 5] = float('nan')                      # arrives through the fp32 residual: the operand planes are clean
    xin = nhwc(torch.ones(1, 32, 8, 8))
    for relu_out, relu_in in ((True, False), (False, True)):
        with ops.use_book(ops.PlanBook()):
            for it in range(2):                         # second pass: the producer writes the planes its consumer asked for
                y1 = ops.conv2d([xin], pack, relu_out=relu_out, residual=res, plan=0x70011)
                ops.conv2d([y1], pack, relu_in=relu_in, relu_out=True, plan=0x70011)
                if it == 0:
                    if relu_out:
                        ops.check_faults()              # (the fp32 map holds a clean 0 already: the stand-alone split sees nothing)
                    else:
                        _expect_range_fault()           # (the stand-alone split tests before ITS input ReLU)
            assert y1.__dict__.get('_swem_split'), 'the producer did not write planes'
            _expect_range_fault()
    # the training step's frozen-BN stage writing the fp16 pair of relu(bn(c))
    c = torch.ones(64, 32, device=DEV)
    c[5, 7] = float('nan')
    one, zero = torch.ones(32, device=DEV), torch.zeros(32, device=DEV)
    y = torch.empty_like(c)
    planes = torch.empty((2, 64 * 32), dtype=torch.float16, device=DEV)
    L.call('swem_bn_act_planes_f32', ops._stream(), c.data_ptr(), one.data_ptr(), zero.data_ptr(), 0, y.data_ptr(), 64, 32, 1,
           planes.data_ptr(), ops.PLANES_F16, ops.fault_word(c.device).data_ptr())
    assert float(y[5, 7]) == 0.0
    _expect_range_fault()


def test_fault_word_survives_graph_replays(lib):
    """the fault word used to be the last word of the per-capture counter buffer, whose zero fill is a node at the
    head of the captured graph -- a fault of replay k was erased by replay k + 1, and the words of every graph but the newest
    were unreachable.  It is now one persistent word per device outside every graph: fault in one replay of an OLDER graph, run
    clean replays of it and of a newer graph, and check_faults still raises."""
    ops.check_faults()
    w = torch.ones(32, 32, 1, 1) * 0.01
    pack = ops.pack_conv(w.to(DEV))
    xs = [nhwc(torch.ones(2, 32, 30, 54)) for _ in range(2)]
    graphs = []
    for x in xs:
        st = ops.new_stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            ops.conv2d([x], pack, relu_out=True, plan=0x70211)       # eager warm-up (K-split: the launch uses the counters)
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                x.__dict__.pop('_swem_split', None)
                y = ops.conv2d([x], pack, relu_out=True, plan=0x70211)
        graphs.append((g, y))
    torch.cuda.current_stream().wait_stream(st)
    graphs[0][0].replay()
    ops.check_faults()
    xs[0][1, 7, 3, 4] = 1.0e5
    graphs[0][0].replay()                     # faults
    xs[0][1, 7, 3, 4] = 1.0
    graphs[0][0].replay()                     # clean replay of the same graph: its memset node runs again
    graphs[1][0].replay()                     # a newer graph's replay
    _expect_range_fault()
    graphs[0][0].replay()
    ops.check_faults()
    close(back(graphs[0][1]), F.relu(F.conv2d(torch.ones(2, 32, 30, 54), w)), 1e-5, 'replayed conv')


@pytest.mark.parametrize('plan', [0x10010011, 0x20010011, 0x30010011, 0x10010021, 0x20010022, 0x30210022, 0x24010021, 0x20810022,
                                  0x20020021, 0x10010211], ids=lambda p: '%#x' % p)
def test_conv2d_xcd_partition(lib, plan):
    """Plan bits 28-29: the tile grid is dealt to the 8 XCDs in 2 / 4 / 8 groups of N tiles instead of ranges of M tiles.
    Placement never changes a tile's arithmetic: the result is BIT-identical to the default placement."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 96, 30, 54, generator=g)
    w = torch.randn(512, 96, 3, 3, generator=g) * 0.05
    b = torch.randn(512, generator=g)
    pack = ops.pack_conv(w.to(DEV), b.to(DEV))
    xs = nhwc(x)
    base = ops.conv2d([xs], pack, relu_in=True, plan=plan & 0x0fffffff)
    y = ops.conv2d([xs], pack, relu_in=True, plan=plan)
    assert torch.equal(y, base)
    ref = F.conv2d(F.relu(x), w, b, padding=1)
    close(back(y), ref, 2e-5 if (plan >> 16) & 3 == 1 else 1e-2, 'xcd-partitioned conv')


@pytest.mark.parametrize('plan', [0x4010021, 0x8010011, 0x4210022, 0x4010022, 0x4810022], ids=lambda p: '%#x' % p)
def test_conv2d_tail_split(lib, plan):
    """Plan bits 24-27: the last, partly filled round of tiles is launched a second time split over K and reduced over
    its rows only; the result equals the plain launch's up to fp32 summation order."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 120, 216, generator=g)
    w = torch.randn(128, 64, 3, 3, generator=g) * 0.05
    b = torch.randn(128, generator=g)
    pack = ops.pack_conv(w.to(DEV), b.to(DEV))
    xs = nhwc(x)
    base = ops.conv2d([xs], pack, relu_in=True, plan=plan & 0xffffff)
    y = ops.conv2d([xs], pack, relu_in=True, plan=plan)
    ref = F.conv2d(F.relu(x), w, b, padding=1)
    close(back(y), ref, 2e-5, 'tail-split conv')
    assert float((y - base).abs().max()) < 1e-4
    # the workspace query covers the tail's partial sums
    assert ops._lib.query('swem_conv2d_workspace', 2, 120, 216, 64, 128, 3, 3, 1, 1, 1, plan) > 0


@pytest.mark.parametrize('shape', [(2, 120, 216, 256, 256, 3), (2, 60, 108, 512, 256, 3), (2, 30, 54, 1280, 512, 3), (1, 30, 54, 256, 1024, 1),
                                   (1, 37, 53, 96, 160, 3)], ids=lambda v: 'x'.join(str(e) for e in v))
@pytest.mark.parametrize('plan', [0x1630022, 0x1030011], ids=lambda p: '%#x' % p)
def test_conv2d_stream_k(lib, shape, plan):
    """Plan bits 24-27 == 1: stream-K.  Persistent workers share the tiles x k-blocks iteration space in equal ranges; a
    tile that straddles workers is finished by the worker that holds its head, which adds the others' partial sums in a fixed
    order.  The result equals the plain launch's up to fp32 summation order (a straddled tile's sum is cut at other k than
    a K-split would cut it), is the SAME on every launch (deterministic: no atomics on the data), and the config-B layer
    shapes -- one to five workers per tile, tiles that no worker shares -- all agree with the fp32 convolution."""
    B, H, W, Cin, Cout, k = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (0.5 / (Cin * k * k) ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    pack = ops.pack_conv(w.to(DEV), b.to(DEV), None, 1, k // 2)
    xs = nhwc(x)
    base = ops.conv2d([xs], pack, relu_in=True, relu_out=True, plan=plan & 0xffffff)
    y = ops.conv2d([xs], pack, relu_in=True, relu_out=True, plan=plan)
    y2 = ops.conv2d([xs], pack, relu_in=True, relu_out=True, plan=plan)
    assert torch.equal(y, y2), 'stream-K must be deterministic'
    err = float((y - base).abs().max() / base.abs().max())
    assert err < 1e-5, err
    ref = F.relu(F.conv2d(F.relu(x), w, b, padding=k // 2))
    close(back(y), ref, 2e-5, 'stream-K conv')


def test_async_fault_word_reaches_the_host(lib):
    """A K-split reducer whose bounded wait for the other splits' partial tiles expires sets the sticky fault word of the
    device (include/swem_hip.h, SWEM_FAULT_KSPLIT_WAIT; ops.fault_word) instead of silently reducing tiles that were never
    written; ops.check_faults() raises on the host, zeroes the counters, and the next launch is clean.  The wait is shortened
    to zero polls through SWEM_SPIN_LIMIT in a child process (the library reads it once): with every reducer arriving
    together with its producers, some tile of a 4-way split always finds its partials missing."""
    import os
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    code = '''
import torch
from swem_amd import ops, _lib
g = torch.Generator().manual_seed(1)
x = torch.randn(2, 30, 54, 512, generator=g).cuda()
pack = ops.pack_conv((torch.randn(512, 512, 3, 3, generator=g) * 0.02).cuda())
ref = ops.conv2d([x], pack, plan=0x30122)                     # no K-split: no waiting
ops.check_faults()
y = ops.conv2d([x], pack, plan=0x30422)                       # four splits, the last one reduces
torch.cuda.synchronize()
ctr = ops.counters(x.device)
word = int(ops.fault_word(x.device))
y2 = ops.conv2d([x], pack, plan=0x30122)                      # later clean launches do not clear it
torch.cuda.synchronize()
assert int(ops.fault_word(x.device)) == word
try:
    ops.check_faults()
    raised = False
except _lib.SwemHipError as e:
    raised = 'K-split' in str(e)
print('FAULT_WORD', word, 'RAISED', raised, 'ZEROED', int(ctr.abs().sum()) == 0)
ops.check_faults()                                            # reset: clean again
'''
    env = dict(os.environ, SWEM_SPIN_LIMIT='0', PYTHONPATH=root)
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('FAULT_WORD')][-1].split()
    assert int(line[1]) & 1 and line[3] == 'True' and line[5] == 'True', out.stdout
    # the same launches with the default wait: no fault, K-split result = unsplit result up to the summation order
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 30, 54, 512, generator=g).to(DEV)
    pack = ops.pack_conv((torch.randn(512, 512, 3, 3, generator=g) * 0.02).to(DEV))
    ref = ops.conv2d([x], pack, plan=0x30122)
    y = ops.conv2d([x], pack, plan=0x30422)
    ops.check_faults()
    assert int(ops.counters(x.device).abs().sum()) == 0
    close(y.cpu(), ref.cpu(), 1e-5, 'fused K-split')


@pytest.mark.parametrize('plan', [0x70011, 0x70022, 0x670022, 0x70221, 0x4070021, 0x470012], ids=lambda p: '%#x' % p)
def test_conv2d_block_output_as_planes_and_plane_residual(lib, plan):
    """Two chained ResNet-style blocks (mod_resnet.py:77-113: out = relu(conv(x) + identity)): with planes_only='block' the FIRST
    block's output is written as operand planes only once both of its consumers in the second block -- the convolution that reads
    it (planes) and the convolution that ADDS it (swem_conv2d_nhwc_bf16x3_planes_res: the addend read back from the planes) -- have
    asked, for the current plans.  The addend is hi + mid of the fp16 pair (22-23 bits of the fp32 value): the second block's output
    stays within 2e-6 of the all-fp32-map run.  A consumer on the fp32 kernels or a plan change withdraw the promise.  (The entry
    point also takes three bf16 planes -- the value exactly; the policy does not use them: 6 bytes per element against the map's 4
    made the exact-split leg slower -- checked here through the C ABI directly.)"""
    from swem_amd import _lib
    g = torch.Generator().manual_seed(17)
    B, C, H, W = 2, 64, 21, 37
    x = nhwc(torch.randn(B, C, H, W, generator=g))
    mk = lambda: ops.pack_conv((torch.randn(C, C, 3, 3, generator=g) * 0.05).to(DEV), (torch.randn(C, generator=g) * 0.1).to(DEV))
    pa, pb, pc = mk(), mk(), mk()
    book = ops.PlanBook(fallback=plan)

    def two_blocks(only):
        y1 = ops.conv2d([x], pa, relu_out=True, residual=x, planes_only=only)          # block 1: relu(conv(x) + x)
        t = ops.conv2d([y1], pb, relu_out=True, planes_only=True)                      # block 2, conv1 reads y1
        y2 = ops.conv2d([t], pc, relu_out=True, residual=y1)                           # block 2, conv2 adds y1
        return y1, y2
    exact = False
    with ops.use_book(book):
        y1_ref, y2_ref = two_blocks(False)           # frame 1: everything written; both consumers of y1 have asked
        assert not y1_ref.__dict__.get('_swem_planes_only')
        y1_b, y2_b = two_blocks('block')             # frame 2: planes only
        assert y1_b.__dict__.get('_swem_planes_only') and bool(torch.isnan(y1_b).all())
        if exact:
            assert torch.equal(y2_b, y2_ref)
        else:
            close(back(y2_b), back(y2_ref), 2e-6, 'second block, addend from the fp16 pair, plan %#x' % plan)
            assert not torch.equal(y2_b, y2_ref)     # (really the plane path: the addend differs in its last bit or two)
        with pytest.raises(_lib.SwemHipError):
            ops.conv2d([t_ := ops.conv2d([y1_b], pb, relu_out=True)], pc, residual=y1_b, plan=0x11)    # an fp32-kernel adder: loud
        book.fallback = plan ^ 0x100                 # a plan change: written again until both consumers have asked anew
        y1_c, _ = two_blocks('block')
        assert not y1_c.__dict__.get('_swem_planes_only')
        assert two_blocks('block')[0].__dict__.get('_swem_planes_only')
    # the adding convolution on the fp32 kernels (its own tuned plan says so) never says it can read planes: no promise, ever
    book2 = ops.PlanBook(fallback=plan)
    book2.conv[(C, C, 3, 3, 1, 1, 0, B, H, W)] = 0x11           # (the adder below: no output ReLU, hence its own signature)
    with ops.use_book(book2):
        for _ in range(3):
            y1_d = ops.conv2d([x], pa, relu_out=True, residual=x, planes_only='block')
            t = ops.conv2d([y1_d], pb, relu_out=True, planes_only=True)
            ops.conv2d([t], pc, residual=y1_d)
            assert not y1_d.__dict__.get('_swem_planes_only')


@pytest.mark.parametrize('B,H,W', [(1, 8, 16), (2, 21, 37), (1, 120, 216)], ids=lambda v: str(v))
def test_fused_bottleneck_matches_three_launches(lib, B, H, W):
    """swem_bottleneck_f16x3 (csrc/bneck.hip): an identity bottleneck block of the key encoder's layer1 (mod_resnet.py:77-113:
    1x1 256 -> 64, 3x3 64 -> 64, 1x1 64 -> 256, frozen BN + ReLU after each, + identity) in ONE launch, against (a) the same block
    as three launches of the f16x3 convolution kernel in its 16x16x32 form -- same operand pairs, same k order: agreement to a
    few fp32 ulps of the accumulation -- and (b) torch fp32 on the CPU.  Ragged sizes (tiles of 8 x 16 pixels that hang over the
    image on both axes), the output as the fp32 map, as map + fp16 pair, and as planes only; the pair bit-identical to
    swem_split_f16x2_f32 of the map."""
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, 256, H, W, generator=g).abs()           # (a block's input is a ReLU output)

    def conv_bn(co, ci, k):
        w = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
        bn = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1,
              torch.rand(co, generator=g) + 0.5)
        return w, bn, ops.pack_conv(w.to(DEV), None, [t.to(DEV) for t in bn], 1, k // 2)
    (w1, bn1, c1), (w2, bn2, c2), (w3, bn3, c3) = conv_bn(64, 256, 1), conv_bn(64, 64, 3), conv_bn(256, 64, 1)

    def bnf(t, bn):
        return F.batch_norm(t, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    ref = F.relu(bnf(F.conv2d(x, w1), bn1))
    ref = F.relu(bnf(F.conv2d(ref, w2, padding=1), bn2))
    ref = F.relu(bnf(F.conv2d(ref, w3), bn3) + x)
    xs = nhwc(x)
    plan = 0x470111                                             # f16x3, 64 x 64 tile, 16x16x32 MFMA, no K-split
    with ops.use_book(ops.PlanBook(fallback=ops.MODEL_FALLBACK)):
        assert not ops.bottleneck_ok(xs, c1, c2, c3)            # off by default (profiles/r05_bottleneck_fusion.txt)
    with ops.use_book(ops.PlanBook(fallback=ops.MODEL_FALLBACK)) as book, ops.flags(FUSE_BOTTLENECK=True):
        assert ops.bottleneck_ok(xs, c1, c2, c3)
        y1 = ops.conv2d([xs], c1, relu_out=True, plan=plan)
        y2 = ops.conv2d([y1], c2, relu_out=True, plan=plan)
        y3 = ops.conv2d([y2], c3, relu_out=True, residual=xs, plan=plan)
        # (the unfused path adds the fp32 identity, the fused one hi + mid of its fp16 pair: 2^-23 |x| apart at most)
        yf = ops.bottleneck(xs, c1, c2, c3)
        ops.check_faults()
        assert '_swem_split' not in yf.__dict__
        close(back(yf), ref, 2e-5, 'fused bottleneck vs torch')
        close(yf.cpu(), y3.cpu(), 2e-6, 'fused bottleneck vs three launches')
        # a consumer asks for the fp16 pair: from the next call on the launch writes it, bit-identical to the split of the map
        ops.presplit(yf, False, ops.PLANES_F16)
        yg = ops.bottleneck(xs, c1, c2, c3)
        assert torch.equal(yg, yf)
        sp = yg.__dict__['_swem_split'][ops._pkey(False, ops.PLANES_F16)][0]
        want = torch.empty_like(sp)
        __import__('swem_amd')._lib.call('swem_split_f16x2_f32', ops._stream(), yg.data_ptr(), want.data_ptr(), B * H * W, 256, 0, 0)
        assert torch.equal(sp.view(torch.int16), want.view(torch.int16))
        # inside a stage: two blocks chained, the first one's output as planes only once both of its consumers -- the second block's
        # convolutions and its identity -- read planes (the fused kernel does both)
        for _ in range(3):
            a = ops.bottleneck(xs, c1, c2, c3, planes_only='block')
            bb = ops.bottleneck(a, c1, c2, c3)
        assert a.__dict__.get('_swem_planes_only')
        ref2 = F.relu(bnf(F.conv2d(ref, w1), bn1))
        ref2 = F.relu(bnf(F.conv2d(ref2, w2, padding=1), bn2))
        ref2 = F.relu(bnf(F.conv2d(ref2, w3), bn3) + ref)
        close(back(bb), ref2, 3e-5, 'two fused blocks')
        # ... and an unfused consumer of a planes-only fused output (the next stage's first block: a strided 3x3 elsewhere; here
        # the plain 1x1) reads the same planes
        z = ops.conv2d([a], c1, relu_out=True, plan=plan)
        close(back(z), F.relu(bnf(F.conv2d(ref, w1), bn1)), 2e-5, 'unfused consumer of the planes')
        ops.check_faults()
    # a book on another arithmetic keeps the three launches
    with ops.use_book(ops.PlanBook(fallback=0)), ops.flags(FUSE_BOTTLENECK=True):
        assert not ops.bottleneck_ok(xs, c1, c2, c3)
    with ops.use_book(ops.PlanBook(fallback=ops.MODEL_FALLBACK)), ops.conv_math((3,)), ops.flags(FUSE_BOTTLENECK=True):
        assert not ops.bottleneck_ok(xs, c1, c2, c3)
    # the range fault of the shipped arithmetic reaches through the fused kernel too (an intermediate beyond 65520)
    big = ops.pack_conv((w1 * 3.0e4).to(DEV), None, [t.to(DEV) for t in bn1], 1, 0)
    with ops.use_book(ops.PlanBook(fallback=ops.MODEL_FALLBACK)):
        ops.bottleneck(xs, big, c2, c3)
        with pytest.raises(ops.SwemRangeError):
            ops.check_faults()

@pytest.mark.parametrize('case', [
    dict(B=2, H=120, W=216, cs=[64], co=256, k=3, s=1, ns=1, relu=True, res=False),
    dict(B=2, H=30, W=54, cs=[512], co=512, k=3, s=1, ns=4, relu=False, res=True),
    dict(B=3, H=17, W=5, cs=[32, 64], co=96, k=3, s=1, ns=1, relu=True, res=True),
    dict(B=5, H=11, W=13, cs=[64, 32, 32], co=200, k=3, s=1, ns=2, relu=False, res=False),
    dict(B=2, H=33, W=41, cs=[128], co=256, k=3, s=2, ns=1, relu=False, res=False),
    dict(B=1, H=64, W=64, cs=[1024], co=512, k=1, s=1, ns=8, relu=False, res=False),
    dict(B=4, H=64, W=64, cs=[128], co=300, k=3, s=1, ns=1, relu=False, res=True),
], ids=['big_relu', 'ksplit4_res', 'two_src', 'three_src_ksplit', 'stride2', '1x1_ksplit8', 'ragged_cols'])
def test_conv2d_256_column_tiles_equal_the_128_tile(lib, case):
    """conv_t256_kernel (plan tile 4 x 4; round 5): every tile height (256 rows on the 2 x 4 wave grid; 128 / 160 / 192 / 224 on
    the 1 x 8 grid) against the 128x128 eight-wave kernel (variant 6) on the same planes -- the same k order and products, so
    IDENTICAL bits without a K-split and 1e-6 of the range with one (another partition of the fp32 sums) -- and against fp64;
    with output planes, residual, ReLU, several sources, stride 2, a ragged column tile."""
    g = torch.Generator().manual_seed(11)
    B, H_, W_, cs, co, k, s_, ns = (case[n] for n in ('B', 'H', 'W', 'cs', 'co', 'k', 's', 'ns'))
    C = sum(cs)
    w = (torch.randn(co, C, k, k, generator=g) * (2.0 / (C * k * k)) ** 0.5)
    bias = torch.randn(co, generator=g)
    xs = [torch.randn(B, c, H_, W_, generator=g) for c in cs]
    pack = ops.pack_conv(w.to(DEV), bias.to(DEV), None, s_, k // 2)
    ref = F.conv2d(torch.cat(xs, 1).double(), w.double(), bias.double(), stride=s_, padding=k // 2)
    r = torch.randn(*ref.shape, generator=g) if case['res'] else None
    if r is not None:
        ref = ref + r.double()
    if case['relu']:
        ref = ref.relu()
    srcs = [nhwc(x) for x in xs]
    rs = nhwc(r) if r is not None else None
    base = ops.conv2d(srcs, pack, relu_out=case['relu'], residual=rs, plan=0x670022 | ns << 8)
    for v in (0, 4, 5, 6, 7):
        plan = 0x70044 | ns << 8 | v << 20
        y = ops.conv2d(srcs, pack, relu_out=case['relu'], residual=rs, plan=plan)
        close(back(y), ref, 2e-6, 'plan %#x against fp64' % plan)
        if ns == 1:
            assert torch.equal(y, base), 'plan %#x differs from the 128x128 kernel' % plan
        else:
            assert float((y - base).abs().max()) <= 1e-6 * float(base.abs().max()), hex(plan)
    ops.check_faults()


def test_conv2d_256_column_tile_glu_and_planes(lib):
    """The 256-row form is the only one that takes SWEM_CONV_GLU (the fusion layer, modules.py:13-26: packed f | gate columns);
    its epilogue writes the output's operand planes like every other tile's: both against the 128x128 kernel, bit for bit.
    A height below 256 with GLU or a tail-split field is refused."""
    g = torch.Generator().manual_seed(5)
    B, H_, W_, ci, co = 2, 30, 54, 256, 256
    x = nhwc(torch.randn(B, ci, H_, W_, generator=g))
    pk = ops.pack_glu(*(t.to(DEV) for t in (torch.randn(co, ci, 3, 3, generator=g) * 0.03, torch.randn(co, generator=g),
                                            torch.randn(co, ci, 3, 3, generator=g) * 0.03, torch.randn(co, generator=g))))
    y0 = ops.conv2d([x], pk, plan=0x670022)
    y1 = ops.conv2d([x], pk, plan=0x70044)
    y2 = ops.conv2d([x], pk, plan=0x70244)
    assert torch.equal(y0, y1), 'GLU: the 256-row tile differs from the 128x128 kernel'
    # (a K-split is another partition of the fp32 sums of BOTH factors of the gated product)
    assert float((y2 - y0).abs().max()) <= 4e-6 * float(y0.abs().max()), float((y2 - y0).abs().max()) / float(y0.abs().max())
    with pytest.raises(ops._lib.SwemHipError):
        ops.conv2d([x], pk, plan=0x670044)          # 192 rows with GLU
    pc = ops.pack_conv((torch.randn(co, ci, 3, 3, generator=g) * 0.03).to(DEV))
    with pytest.raises(ops._lib.SwemHipError):
        ops.conv2d([x], pc, plan=0x4070044)         # a tail-split field
    # (a 4 x 4 tile under another arithmetic is no plan of this kernel: the library's own heuristic runs, as for any unknown tile)
    close(back(ops.conv2d([x], pc, plan=0x10044)), back(ops.conv2d([x], pc, plan=0x10022)), 4e-6, 'bf16x6 with a 4 x 4 tile field')
    # output planes from the epilogue (fused operand split): a consumer's request makes the producer write them
    with ops.use_book(ops.PlanBook()), ops.flags(FUSE_SPLIT=True):
        outs = {}
        for plan in (0x670022, 0x770044, 0x70044):
            for it in range(2):
                y = ops.conv2d([x], pc, relu_out=True, plan=plan)
                ops.conv2d([y], pc, plan=0x70011)
            assert y.__dict__.get('_swem_split'), hex(plan)
            outs[plan] = (y, {kk: vv[0].clone() for kk, vv in y.__dict__['_swem_split'].items()})
        for plan in (0x770044, 0x70044):
            assert torch.equal(outs[plan][0], outs[0x670022][0])
            for kk, vv in outs[0x670022][1].items():
                assert torch.equal(outs[plan][1][kk].view(torch.int16), vv.view(torch.int16)), (hex(plan), kk)
    ops.check_faults()

def test_conv2d_256_column_tile_properties_at_config_b_size(lib):
    """Size-independent properties of the 256-column tile kernel at the bench's largest layer (2 x 120 x 216, 256 -> 256, 3x3; 232
    tiles of 224 rows) where an fp64 reference is not worth its time: scaling the input by a power of two scales every output bit-
    exactly (fp16 pairs and fp32 accumulation are exact under it), swapping the two images swaps the outputs bit-exactly (a tile
    straddles the image boundary: pixel 25,920 is row 160 of tile 115), and the result equals the 128x128 kernel's bit for bit."""
    g = torch.Generator().manual_seed(23)
    r = torch.randn(2, 120, 216, 256, generator=g)
    x = (torch.sign(r) * (0.5 + r.abs())).to(DEV)      # (|x| >= 2^-2: the fp16 pair of x is exact to half an fp32 ulp, so 4x splits as 4 x the pair)
    pk = ops.pack_conv((torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(DEV))
    y = ops.conv2d([x], pk, relu_out=True, plan=0x770144)
    assert torch.equal(y, ops.conv2d([x], pk, relu_out=True, plan=0x670122))
    assert torch.equal(ops.conv2d([x * 4.0], pk, relu_out=True, plan=0x770144), y * 4.0)
    assert torch.equal(ops.conv2d([x.flip(0).contiguous()], pk, relu_out=True, plan=0x770144), y.flip(0))
    assert torch.isfinite(y).all() and float(y.max()) > 0
    ops.check_faults()

@pytest.mark.parametrize('case', [
    dict(B=2, H=60, W=108, cs=[256], co=256, k=3, s=1, ns=1, relu=True),
    dict(B=2, H=30, W=54, cs=[512], co=512, k=3, s=1, ns=4, relu=False),
    dict(B=3, H=17, W=5, cs=[32, 64], co=96, k=3, s=1, ns=1, relu=True),
    dict(B=2, H=33, W=41, cs=[128], co=256, k=3, s=2, ns=2, relu=False),
], ids=['mid', 'ksplit4', 'two_src', 'stride2_ksplit'])
def test_conv2d_256_column_tile_bf16x6(lib, case):
    """The three-plane (bf16x6: all 24 operand bits) form of conv_t256_kernel -- 128-row tiles, plan math 1, tile 4 x 4, bits 20-23
    = 4 -- against fp64 and against the library's other bf16x6 kernels on the same planes (same products; the MFMA shape and the
    partition of the sums may differ: 1e-6 of the range)."""
    g = torch.Generator().manual_seed(29)
    B, H_, W_, cs, co, k, s_, ns = (case[n] for n in ('B', 'H', 'W', 'cs', 'co', 'k', 's', 'ns'))
    C = sum(cs)
    w = (torch.randn(co, C, k, k, generator=g) * (2.0 / (C * k * k)) ** 0.5)
    bias = torch.randn(co, generator=g)
    xs = [torch.randn(B, c, H_, W_, generator=g) for c in cs]
    pack = ops.pack_conv(w.to(DEV), bias.to(DEV), None, s_, k // 2)
    ref = F.conv2d(torch.cat(xs, 1).double(), w.double(), bias.double(), stride=s_, padding=k // 2)
    if case['relu']:
        ref = ref.relu()
    srcs = [nhwc(x) for x in xs]
    y = ops.conv2d(srcs, pack, relu_out=case['relu'], plan=0x410044 | ns << 8)
    base = ops.conv2d(srcs, pack, relu_out=case['relu'], plan=0x10022 | ns << 8)
    close(back(y), ref, 2e-6, 'bf16x6 on 128-row tiles against fp64')
    assert float((y - base).abs().max()) <= 1e-6 * float(base.abs().max())
    ops.check_faults()

def test_conv2d_256_column_tiles_reproduce_bitwise_on_concurrent_streams(lib):
    """Race screen of conv_t256_kernel's hand-over (one barrier per k-block, the next stage's transfers awaited a k-block after they
    were requested): launches of every form -- tile heights, K-splits (fixed reduction order), bf16x6 -- on two streams at once
    reproduce, bit for bit, what the same plan gives alone (tools/t256_race_screen.py: the 80-round version)."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 60, 108, 256, generator=g).to(DEV)
    x2 = torch.randn(2, 30, 54, 512, generator=g).to(DEV)
    pk = ops.pack_conv((torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(DEV))
    pk2 = ops.pack_conv((torch.randn(512, 512, 3, 3, generator=g) * 0.02).to(DEV))
    p1, p2 = (0x770144, 0x70144, 0x570244, 0x470244, 0x410244), (0x570344, 0x70444, 0x470444, 0x410444)
    r1 = [ops.conv2d([x], pk, plan=p).clone() for p in p1]
    r2 = [ops.conv2d([x2], pk2, plan=p).clone() for p in p2]
    torch.cuda.synchronize()
    s1, s2 = ops.new_stream(), ops.new_stream()
    for _ in range(12):
        with torch.cuda.stream(s1):
            ys = [ops.conv2d([x], pk, plan=p) for p in p1]
        with torch.cuda.stream(s2):
            zs = [ops.conv2d([x2], pk2, plan=p) for p in p2]
        torch.cuda.synchronize()
        for p, y, r in list(zip(p1, ys, r1)) + list(zip(p2, zs, r2)):
            assert torch.equal(y, r), hex(p)
    ops.check_faults()
