#!/usr/bin/env python
"""Alternative plan file with the f16x3 STREAM-K form (round 6) wherever it beats the shipped plan of a layer
shape ALONE on an idle chip; tools/tune_in_context.py then keeps the flips that raise the whole-job frame rate.

    python tools/streamk_alt_plans.py [BASE.json] OUT.json [--slack 1.0]
Every conv entry of BASE (default: the shipped plans) whose plan is an f16x3 128x128 tile (any variant, any K-split, no tail
split) without per-batch filters is timed (HIP-graph replays) with its plan and with plan bits 24-27 = 1 on the eight-wave
16x16x32 instantiation (variant 6: the only f16 kernel with a stream-K form) and no K-split: persistent workers, one per
resident-block slot, equal shares of tiles x k-blocks.

NEEDS the f16 stream-K instantiation, which the shipped library does not carry: apply profiles/r06_experiments/streamk_f16.patch
and rebuild first (the launcher otherwise answers "the f16x3 arithmetic has no stream-K form").  Result of the round-6 run:
profiles/r06_streamk_f16_ab.txt -- slower on every config-B layer, not shipped."""
import argparse
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from swem_amd import ops  # noqa: E402
from t256_alt_plans import timed  # noqa: E402

SK_PLAN = 2 | 2 << 4 | 7 << 16 | 6 << 20 | 1 << 24


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('files', nargs='+')
    ap.add_argument('--slack', type=float, default=1.0)
    a = ap.parse_args()
    base = a.files[0] if len(a.files) > 1 else os.path.join(ROOT, 'swem_amd', 'plans', 'mi355x_480p_k256.json')
    out = a.files[-1]
    d = json.load(open(base))
    dev = 'cuda:0'
    flips = 0
    new = []
    with torch.no_grad():
        for k, v in d['conv']:
            glu = bool(k[6] & 4)
            ok = len(k) == 10 and (v >> 16) & 7 == 7 and (v & 0xff) == 0x22 and (v >> 24) & 15 == 0 and not k[6] & 8
            cin, cout, kh, kw, stride, pad, flags, B, H, W = k[:10]
            if ok and kh * kw * cin // 32 < 16:
                ok = False                       # (the launcher wants at least four k-blocks per worker)
            if not ok:
                new.append([k, v])
                continue
            x = torch.randn(B, H, W, cin, device=dev)
            if glu:
                pack = ops.pack_glu(torch.randn(cout, cin, kh, kw, device=dev) * 0.02, torch.zeros(cout, device=dev),
                                    torch.randn(cout, cin, kh, kw, device=dev) * 0.02, torch.zeros(cout, device=dev))
            else:
                pack = ops.pack_conv(torch.randn(cout, cin, kh, kw, device=dev) * 0.02, torch.zeros(cout, device=dev), None, stride, pad)
            run = lambda plan: ops.conv2d([x], pack, relu_in=bool(flags & 1), relu_out=bool(flags & 2), plan=plan)
            t_cur = timed(lambda: run(v))
            t_sk = timed(lambda: run(SK_PLAN))
            keep = t_sk <= a.slack * t_cur
            flips += keep
            print('%s: shipped %#x %.1f us; stream-K %#x %.1f us (%+.1f %%)%s'
                  % (k, v, t_cur, SK_PLAN, t_sk, 100 * (t_sk / t_cur - 1), '  -> alternative' if keep else ''), flush=True)
            new.append([k, SK_PLAN if keep else v])
    json.dump(dict(d, conv=new), open(out, 'w'))
    print('%d alternatives written to %s' % (flips, out))


if __name__ == '__main__':
    main()
