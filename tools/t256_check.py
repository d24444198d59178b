#!/usr/bin/env python
"""The 256x256-tile f16x3 kernel (conv.hip, conv_t256_kernel; plan tile 4 x 4) against the 128x128 kernel (variant 6) and fp64.
    python tools/t256_check.py
Same k order and products: without K-split the outputs must be IDENTICAL; with K-split equal to 1e-6 of the output scale."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from swem_amd import ops  # noqa: E402

# (B, H, W, [source channels], Cout, k, stride, nsplit, relu_out, residual, dgrad)
CASES = [(1, 30, 54, [64], 64, 3, 1, 1, False, False), (2, 120, 216, [64], 256, 3, 1, 1, True, False),
         (2, 30, 54, [512], 512, 3, 1, 4, False, True), (1, 7, 9, [32], 128, 3, 1, 1, False, False),
         (3, 17, 5, [32, 64], 96, 3, 1, 1, True, True), (1, 1, 1, [32], 128, 3, 1, 1, False, False),
         (2, 60, 108, [256], 256, 3, 1, 2, True, True), (5, 11, 13, [64, 32, 32], 200, 3, 1, 2, False, False),
         (2, 33, 41, [64], 512, 1, 1, 1, True, False), (2, 33, 41, [128], 256, 3, 2, 1, False, False),
         (1, 64, 64, [1024], 512, 1, 1, 8, False, False), (4, 64, 64, [128], 300, 3, 1, 1, False, True)]


def main():
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(11)
    bad = 0
    with torch.no_grad():
        for ci, (B, H, W, cs, co, k, s, ns, relu, use_res) in enumerate(CASES):
            C = sum(cs)
            w = (torch.randn(co, C, k, k, generator=g) * (2.0 / (C * k * k)) ** 0.5).to(dev)
            bias = torch.randn(co, generator=g).to(dev)
            xs = [torch.randn(B, H, W, c, generator=g).to(dev) for c in cs]
            pack = ops.pack_conv(w, bias, None, s, k // 2)
            Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
            r = torch.randn(B, Ho, Wo, co, generator=g).to(dev) if use_res else None
            ys = {}
            for name, plan in [('t256', 0x70044 | ns << 8), ('v6', 0x670022 | ns << 8)] + [('h%d' % (32 * v), 0x70044 | ns << 8 | v << 20) for v in (4, 5, 6, 7)]:
                ys[name] = ops.conv2d(xs, pack, relu_out=relu, residual=r, plan=plan)
                ops.check_faults()
            ref = F.conv2d(torch.cat(xs, 3).permute(0, 3, 1, 2).double(), w.double(), bias.double(), stride=s,
                           padding=k // 2).permute(0, 2, 3, 1)
            if r is not None:
                ref = ref + r.double()
            if relu:
                ref = ref.relu()
            err = float((ys['t256'].double() - ref).abs().max() / ref.abs().max())
            same = torch.equal(ys['t256'], ys['v6'])
            d = float((ys['t256'] - ys['v6']).abs().max() / ys['v6'].abs().max())
            hs = [torch.equal(ys['h%d' % (32 * v)], ys['v6']) or (ns > 1 and float((ys['h%d' % (32 * v)] - ys['v6']).abs().max() / ys['v6'].abs().max()) < 1e-6) for v in (4, 5, 6, 7)]
            ok = err < 2e-6 and (same or (ns > 1 and d < 1e-6)) and all(hs)
            bad += not ok
            print('case %d %s: rel err vs fp64 %.2e; vs the 128x128 kernel %s (%.1e); heights 128..224 %s %s'
                  % (ci, CASES[ci], err, 'IDENTICAL' if same else 'differs', d, hs, '' if ok else 'FAILED'), flush=True)
        # data gradient (SWEM_CONV_DGRAD) through the same kernel
        for (B, H, W, c, co, k, s) in ((2, 24, 24, 64, 128, 3, 1), (2, 25, 23, 64, 256, 3, 2)):
            x = torch.randn(B, c, H, W, generator=g, dtype=torch.float64, requires_grad=True)
            w = (torch.randn(co, c, k, k, generator=g) * 0.05)
            with torch.enable_grad():
                y = F.conv2d(x, w.double(), stride=s, padding=k // 2)
                dy = torch.randn(*y.shape, generator=g)
                gx, = torch.autograd.grad(y, x, dy.double())
            wt = w.permute(1, 2, 3, 0).contiguous().to(dev)
            pk = ops.ConvPack(wt, None, None, c, k, k, s, k // 2)
            d = dy.permute(0, 2, 3, 1).contiguous().to(dev)
            got = {}
            for name, plan in (('t256', 0x70044), ('v6', 0x670022)):
                got[name] = ops.conv2d([d], pk, dgrad=(H, W), plan=plan, batch=B)
            err = float((got['t256'].double().cpu() - gx.permute(0, 2, 3, 1)).abs().max() / gx.abs().max())
            same = torch.equal(got['t256'], got['v6'])
            ok = err < 3e-6 and same
            bad += not ok
            print('dgrad %s: rel err vs fp64 %.2e; vs the 128x128 kernel %s %s' % ((B, H, W, c, co, k, s), err,
                                                                                  'IDENTICAL' if same else 'differs', '' if ok else 'FAILED'))
        # bf16x6 (all 24 operand bits) on 128-row tiles: plan math 1, tile 4 x 4, bits 20-23 = 4 -- against the eight-wave 16x16x32
        # three-plane kernel (variant 6: the same MFMA shape and order) and fp64
        for ci, (B, H, W, cs, co, k, s, ns, relu, use_res) in enumerate(CASES):
            C = sum(cs)
            w = (torch.randn(co, C, k, k, generator=g) * (2.0 / (C * k * k)) ** 0.5).to(dev)
            xs = [torch.randn(B, H, W, c, generator=g).to(dev) for c in cs]
            pack = ops.pack_conv(w, None, None, s, k // 2)
            y6 = ops.conv2d(xs, pack, relu_out=relu, plan=0x410044 | ns << 8)
            try:
                yb = ops.conv2d(xs, pack, relu_out=relu, plan=0x610022 | ns << 8)
            except Exception:
                yb = None
            ref = F.conv2d(torch.cat(xs, 3).permute(0, 3, 1, 2).double(), w.double(), None, stride=s, padding=k // 2).permute(0, 2, 3, 1)
            if relu:
                ref = ref.relu()
            err = float((y6.double() - ref).abs().max() / ref.abs().max())
            same = yb is not None and torch.equal(y6, yb)
            d = float((y6 - yb).abs().max() / yb.abs().max()) if yb is not None else float('nan')
            ok = err < 2e-6 and (yb is None or same or (ns > 1 and d < 1e-6))
            bad += not ok
            print('bf16x6 case %d: rel err vs fp64 %.2e; vs the 128x128 three-plane kernel %s (%.1e) %s'
                  % (ci, err, 'IDENTICAL' if same else ('differs' if yb is not None else 'n/a'), d, '' if ok else 'FAILED'), flush=True)
    print('t256_check: %s' % ('OK' if not bad else '%d cases FAILED' % bad))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
