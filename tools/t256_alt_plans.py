#!/usr/bin/env python
"""Alternative plan file with the 256-column tiles of conv_t256_kernel (round 5) wherever they beat the shipped plan of a layer
shape ALONE on an idle chip; tools/tune_in_context.py then keeps the flips that raise the whole-job frame rate.

    python tools/t256_alt_plans.py [BASE.json] OUT.json [--slack 1.0]
Every untagged conv entry of BASE (default: the shipped plans) whose plan is an f16x3 128x128 tile and whose layer has >= 192
output columns is timed (HIP-graph replays) with its plan and with the t256 candidates (tile heights 128 .. 256, K-splits that
fill the 256 CUs about once); the best candidate replaces the plan in OUT if it takes <= slack x the shipped plan's time."""
import argparse
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from swem_amd import ops  # noqa: E402


def timed(fn, reps=20):
    for _ in range(2):
        fn()
    st = ops.new_stream()
    st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
        g.replay()
        st.synchronize()
        e0.record(st)
        g.replay()
        e1.record(st)
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


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
            exact = len(k) == 13 and (v >> 16) & 7 == 1          # the exact-split leg's entries (tag ('math', 0, 1)): bf16x6
            ok = ((len(k) == 10 and (v >> 16) & 7 == 7 or exact and not glu) and (v & 0xff) == 0x22 and (v >> 24) & 15 == 0
                  and k[0] % 32 == 0 and k[1] * (2 if glu else 1) >= 192 and not k[6] & 8)
            if not ok:
                new.append([k, v])
                continue
            cin, cout, kh, kw, stride, pad, flags, B, H, W = k[:10]
            x = torch.randn(B, H, W, cin, device=dev)
            if glu:
                pack = ops.pack_glu(torch.randn(cout, cin, kh, kw, device=dev) * 0.02, torch.zeros(cout, device=dev),
                                    torch.randn(cout, cin, kh, kw, device=dev) * 0.02, torch.zeros(cout, device=dev))
            else:
                pack = ops.pack_conv(torch.randn(cout, cin, kh, kw, device=dev) * 0.02, torch.zeros(cout, device=dev), None, stride, pad)
            run = lambda plan: ops.conv2d([x], pack, relu_in=bool(flags & 1), relu_out=bool(flags & 2), plan=plan)
            t_cur = timed(lambda: run(v))
            Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
            M, ncols, nkb = B * Ho * Wo, cout * (2 if glu else 1), kh * kw * cin // 32
            best = None
            for hv in ((4,) if exact else (0,) if glu else (0, 4, 5, 6, 7)):
                rows = 32 * hv if hv else 256
                tiles = -(-M // rows) * -(-ncols // 256)
                for ns in sorted({1, max(1, min(nkb // 4, 256 // tiles)), max(1, min(nkb // 4, -(-256 // tiles)))}):
                    plan = 4 | 4 << 4 | ns << 8 | (1 if exact else 7) << 16 | hv << 20
                    t = timed(lambda: run(plan))
                    if best is None or t < best[0]:
                        best = (t, plan)
            keep = best[0] <= a.slack * t_cur
            flips += keep
            print('%s: shipped %#x %.1f us; best 256-column tile %#x %.1f us (%+.1f %%)%s'
                  % (k, v, t_cur, best[1], best[0], 100 * (best[0] / t_cur - 1), '  -> alternative' if keep else ''), flush=True)
            new.append([k, best[1] if keep else v])
    json.dump(dict(d, conv=new), open(out, 'w'))
    print('%d alternatives written to %s' % (flips, out))


if __name__ == '__main__':
    main()
