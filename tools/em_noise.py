#!/usr/bin/env python
"""Rounding-noise yardstick of the EM (the W step's 1 - p cancels and the iterations amplify
rounding): the five-object edge clip's first memorize, the frame on which fp32 evaluations disagree most.  The same
memorize is run in float64 (the truth), then DRAWS times each by the fp32 oracle and by the HIP kernels under
mathematically neutral re-orderings of the sums (key channels permuted in x and in the prior bases, result permuted
back).  Prints both error distributions against float64: the HIP path is as accurate as the reference's arithmetic when
they overlap.   python tools/em_noise.py [--draws 16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def mass_err(got, ref, zita):
    z = zita.squeeze(-2).unsqueeze(-2)
    return float(((got - ref) * z).abs().max() / (ref * z).abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--draws', type=int, default=16)
    a = ap.parse_args()
    from oracle import swem_oracle as O          # tools/ run the oracle as the checker, like tests/
    from tests import helpers as H
    from swem_amd import synth
    from swem_amd.modules import SWEMCore
    n_obj, h, w, bases, topl = 5, 112, 176, 64, 32
    cfg = O.make_cfg(BACKBONE='resnet18', NUM_BASES=bases, NUM_EM_ITERS=3, TOPL=topl)
    _, sd = H.make_model_and_sd(cfg, wseed=21 + n_obj, device='cuda:0')
    om = O.Model(sd, cfg)
    frames, m0 = synth.make_clip(t=4, h=h, w=w, n_obj=n_obj, seed=40 + n_obj)
    m0[:, 0] += m0[:, n_obj]
    m0[:, n_obj] = 0
    core = SWEMCore(n_bases=bases, valdim=om.core.valdim, n_iters=3, tau=om.core.tau, topl=topl)
    with torch.no_grad():
        torch.manual_seed(3)
        mk16, _, s16, _, _ = om('encode_key', frames[:, 0])
        mfull = F.interpolate(m0, size=(h, w), mode='nearest')
        om('init', mk16, om('encode_value', frames[:, 0], mfull.float(), s16), m0)
        oqk, oqv, os16, os8, os4 = om('encode_key', frames[:, 1])
        octx, on = om('match', oqk, oqv)
        _, oprob = om('segment', on, octx, os8, os4, None, (h, w))
        opred = oprob.argmax(1)
        opm = F.interpolate(oprob, size=(h, w), mode='bilinear', align_corners=False)
        ohard = (opred.unsqueeze(1) == torch.arange(on + 1).view(1, -1, 1, 1)).long()
        omv = om('encode_value', frames[:, 1], opm, os16)
        prior = om.core.first.bases
        mk = O.mask_prep(ohard, opm, oqk.shape[-2], oqk.shape[-1])
        args = (om.core.n_bases, om.core.n_iters, om.core.tau, om.core.valdim)
        b64 = O.swem(oqk.double(), omv.double(), mk.double(), {k: v.double() for k, v in prior.items()}, *args)
        gp = torch.Generator().manual_seed(1)
        rows = {'reference fp32': [], 'hip': []}
        for k in range(a.draws):
            perm = torch.arange(oqk.shape[1]) if k == 0 else torch.randperm(oqk.shape[1], generator=gp)
            inv = torch.argsort(perm)
            pr = dict(prior, kappa=prior['kappa'][..., perm, :].contiguous())
            xp = oqk[:, perm].contiguous()
            bo = O.swem(xp, omv, mk, pr, *args)
            bh = core.swem(xp.cuda(), omv.cuda(), mk.cuda(), {k_: v.cuda() for k_, v in pr.items()})
            for name, b in (('reference fp32', bo), ('hip', bh)):
                b = {k_: v.cpu().double() for k_, v in b.items()}
                rows[name].append((mass_err(b['kappa'][..., inv, :], b64['kappa'], b64['zita']),
                                   mass_err(b['nu'], b64['nu'], b64['zita']),
                                   float((b['zita'] - b64['zita']).abs().max() / b64['zita'].abs().max())))
    for name, r in rows.items():
        t = torch.tensor(r)
        print('%-15s vs float64 over %d re-orderings:' % (name, a.draws))
        for j, q in enumerate(('kappa (mass-weighted)', 'nu (mass-weighted)', 'zita')):
            c = t[:, j]
            print('   %-22s min %.2e  median %.2e  max %.2e' % (q, c.min(), c.median(), c.max()))


if __name__ == '__main__':
    main()
