#!/usr/bin/env python
"""Alternative plan files for the layer shapes of the lock-step lanes (round 6), for tools/tune_in_context.py: the per-layer tuner
picked their plans ALONE on an idle chip; two lanes share the chip in the frame, and some near ties go the other way there.

    python tools/lane_alt_plans.py OUT_PREFIX [--batches 8 40] [--slack 1.06]
Every untagged f16x3 conv entry of the shipped plan file whose batch is one of --batches (8 = four sequences x two objects, 40 = four
sequences x ten frames) and that the pre-split kernels can run with >= 128 output columns is timed (HIP-graph replays) over both kernel
families: the 128x128 tile (eight waves, 16x16x32 MFMA: variant 6) with K-split 1 / 2 / 4, and the 256-column tiles at every tile height
with the K-splits that fill the chip about once.  OUT_PREFIX_family.json holds, per entry, the best plan of the OTHER family than the
shipped plan's if it is within `slack` of the shipped plan's time; OUT_PREFIX_second.json the best plan of the SAME family that differs
from the shipped one, within `slack`."""
import argparse
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from swem_amd import ops  # noqa: E402
from t256_alt_plans import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('prefix')
    ap.add_argument('--batches', type=int, nargs='*', default=[8, 40])
    ap.add_argument('--slack', type=float, default=1.06)
    a = ap.parse_args()
    d = json.load(open(os.path.join(ROOT, 'swem_amd', 'plans', 'mi355x_480p_k256.json')))
    dev = 'cuda:0'
    fam_alt, sec_alt = [], []
    nf = ns_ = 0
    with torch.no_grad():
        for k, v in d['conv']:
            keep = [k, v]
            cin, cout, kh, kw, stride, pad, flags, B, H, W = k[:10]
            glu = bool(flags & 4)
            ncols = cout * (2 if glu else 1)
            ok = (len(k) == 10 and B in a.batches and (v >> 16) & 7 == 7 and (v >> 24) & 15 == 0 and cin % 32 == 0 and ncols >= 128
                  and not flags & 8)
            if not ok:
                fam_alt.append(keep)
                sec_alt.append(keep)
                continue
            x = torch.randn(B, H, W, cin, device=dev)
            if glu:
                pack = ops.pack_glu(torch.randn(cout, cin, kh, kw, device=dev) * 0.02, torch.zeros(cout, device=dev),
                                    torch.randn(cout, cin, kh, kw, device=dev) * 0.02, torch.zeros(cout, device=dev))
            else:
                pack = ops.pack_conv(torch.randn(cout, cin, kh, kw, device=dev) * 0.02, torch.zeros(cout, device=dev), None, stride, pad)
            run = lambda plan: ops.conv2d([x], pack, relu_in=bool(flags & 1), relu_out=bool(flags & 2), plan=plan)
            Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
            M, nkb = B * Ho * Wo, kh * kw * cin // 32
            cands = {}
            for ns in (1, 2, 4):
                if ns == 1 or nkb // ns >= 2:
                    cands[2 | 2 << 4 | ns << 8 | 7 << 16 | 6 << 20] = 'tile'
            if ncols >= 192:
                for hv in ((0,) if glu else (0, 4, 5, 6, 7)):
                    rows = 32 * hv if hv else 256
                    tiles = -(-M // rows) * -(-ncols // 256)
                    for ns in sorted({1, max(1, min(nkb // 4, 256 // tiles)), max(1, min(nkb // 4, -(-256 // tiles)))}):
                        cands[4 | 4 << 4 | ns << 8 | 7 << 16 | hv << 20] = 't256'
            fam_cur = 't256' if (v & 0xff) == 0x44 else 'tile'
            t_cur = timed(lambda: run(v))
            times = {p: timed(lambda: run(p)) for p in cands if p != v}
            other = sorted((t, p) for p, t in times.items() if cands[p] != fam_cur)
            same = sorted((t, p) for p, t in times.items() if cands[p] == fam_cur)
            msg = '%s: shipped %#x %.1f us' % (k, v, t_cur)
            f_, s_ = list(keep), list(keep)
            if other and other[0][0] <= a.slack * t_cur:
                f_ = [k, other[0][1]]
                nf += 1
                msg += '; other family %#x %.1f us (%+.1f %%) -> alt' % (other[0][1], other[0][0], 100 * (other[0][0] / t_cur - 1))
            elif other:
                msg += '; other family %#x %.1f us (%+.1f %%)' % (other[0][1], other[0][0], 100 * (other[0][0] / t_cur - 1))
            if same and same[0][0] <= a.slack * t_cur:
                s_ = [k, same[0][1]]
                ns_ += 1
                msg += '; same family %#x %.1f us (%+.1f %%) -> alt' % (same[0][1], same[0][0], 100 * (same[0][0] / t_cur - 1))
            print(msg, flush=True)
            fam_alt.append(f_)
            sec_alt.append(s_)
    json.dump(dict(d, conv=fam_alt), open(a.prefix + '_family.json', 'w'))
    json.dump(dict(d, conv=sec_alt), open(a.prefix + '_second.json', 'w'))
    print('%d other-family and %d same-family alternatives written (%s_family.json, %s_second.json)' % (nf, ns_, a.prefix, a.prefix))


if __name__ == '__main__':
    main()
