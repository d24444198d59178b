#!/usr/bin/env python
"""Golden vectors G4 / G5: module-level outputs of the REFERENCE's own classes on small seeded
inputs -- ResBlock (with and without downsample), UpsampleBlock, FeatureFusionBlock (+CBAM), FeatureFusionLayer, Decoder
and SWEM.decode / aggregate (with valid_obj) -- with the oracle asserted bit-identical while they are recorded.

Run in the build container only:   python tests/golden/make_golden_modules.py
Weights are the modules' own seeded initialisation, stored in the fixture (they are small); data only, no source.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from make_golden import O  # noqa: E402


def sd_of(mod, prefix):
    return {prefix + k: v.detach().clone() for k, v in mod.state_dict().items()}


def main():
    torch.set_num_threads(8)
    R, Rswem = MG.import_reference()
    N = sys.modules['methods.basic_modules.networks']
    g = torch.Generator().manual_seed(21)
    out = {}

    def rnd(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale

    def randomize(mod):
        for p in mod.parameters():
            p.data = rnd(*p.shape, scale=0.2 if p.dim() > 1 else 0.5)

    torch.manual_seed(3)
    with torch.no_grad():
        # ---- ResBlock, identity and 3x3-downsample shortcut (networks.py:12-32)
        for tag, (ci, co) in (('rb_same', (32, 32)), ('rb_down', (64, 32))):
            m = N.ResBlock(ci, co)
            randomize(m)
            x = rnd(2, ci, 9, 12)
            y = m(x)
            sd = sd_of(m, 'm.')
            assert torch.equal(O.res_block(sd, 'm', x), y), tag
            out.update({tag + '_x': x, tag + '_y': y, **{tag + '_' + k: v for k, v in sd.items()}})
        # ---- UpsampleBlock (networks.py:186-196)
        m = N.UpsampleBlock(32, 64, 32)
        randomize(m)
        skip, up = rnd(2, 32, 12, 16), rnd(2, 64, 6, 8)
        y = m(skip, up)
        sd = sd_of(m, 'm.')
        sk = O.conv(sd, 'm.skip_conv', skip)
        yo = O.res_block(sd, 'm.out_conv', sk + torch.nn.functional.interpolate(up, size=sk.shape[-2:], mode='bilinear',
                                                                                  align_corners=False))
        assert torch.equal(yo, y)
        out.update({'up_skip': skip, 'up_low': up, 'up_y': y, **{'up_' + k: v for k, v in sd.items()}})
        # ---- FeatureFusionBlock with CBAM (networks.py:35-50, attentions.py)
        m = N.FeatureFusionBlock(32 + 32, 64)
        randomize(m)
        x, f16 = rnd(2, 32, 6, 9), rnd(2, 32, 6, 9)
        y = m(x, f16)
        sd = sd_of(m, 'm.')
        xo = O.res_block(sd, 'm.block1', torch.cat([x, f16], 1))
        yo = O.res_block(sd, 'm.block2', xo + O.cbam(sd, 'm.attention', xo))
        assert torch.equal(yo, y)
        out.update({'ffb_x': x, 'ffb_f16': f16, 'ffb_y': y, **{'ffb_' + k: v for k, v in sd.items()}})
        # ---- FeatureFusionLayer (modules.py:13-26)
        m = R.FeatureFusionLayer(96, 32)
        x = rnd(2, 96, 6, 9)
        y = m(x)
        sd = {'swem_core.fusion_layer.' + k: v.detach().clone() for k, v in m.state_dict().items()}
        assert torch.equal(O.fusion_layer(sd, x), y)
        out.update({'ffl_x': x, 'ffl_y': y, **{'ffl_' + k.split('fusion_layer.')[1]: v for k, v in sd.items()}})
        # ---- Decoder + decode / aggregate with valid_obj (networks.py:199-216, swem.py:92-116)
        cfg = O.make_cfg(BACKBONE='resnet18', NUM_BASES=64)
        ref = Rswem.SWEM(cfg)
        # (the decoder's weights are too large for a fixture: the seeded generator of swem_amd.weights recreates them)
        full = MG.weights.fill_state_dict(MG.HipSWEM(cfg).state_dict(), seed=9, backbone='resnet18')
        full['decoder.pred.weight'] = full['decoder.pred.weight'] * 0.2
        ref.load_state_dict(full, strict=False)
        n = 2
        ctx, s8, s4 = rnd(n, 512, 4, 6), rnd(1, 128, 8, 12), rnd(1, 64, 16, 24)
        valid = torch.tensor([[1., 1., 0.]])
        logits, prob = ref.decode(n, ctx, s8, s4, valid, (61, 90))
        sd = {k: v for k, v in full.items() if k.startswith('decoder.')}
        ol, op = O.decode(sd, n, ctx, s8, s4, valid, (61, 90))
        assert torch.equal(ol, logits) and torch.equal(op, prob)
        logits2, prob2 = ref.decode(n, ctx, s8, s4, None, (64, 96))
        out.update({'dec_ctx': ctx, 'dec_s8': s8, 'dec_s4': s4, 'dec_valid': valid, 'dec_logits': logits, 'dec_prob': prob,
                    'dec_logits_novalid': logits2, 'dec_wseed': 9, 'dec_pred_scale': 0.2})
        agg_in = torch.tensor([[[[0.0, 1.0, 0.5, 1e-9]], [[0.0, 0.0, 0.25, 1.0]]]])      # clamp edges of aggregate
        out.update({'agg_in': agg_in, 'agg_out': ref.aggregate(agg_in)})
        assert torch.equal(O.aggregate(agg_in), out['agg_out'])
    MG.save('g45_modules.npz', **out)
    print('done: %d arrays' % len(out))


if __name__ == '__main__':
    main()
