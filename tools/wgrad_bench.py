#!/usr/bin/env python
"""Times swem_conv2d_wgrad_f32 on the training shapes (3 x 384x384 crops, 2 objects).   python tools/wgrad_bench.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from swem_amd import _lib, ops  # noqa: E402

SHAPES = [  # B, H, W, Cin, Cout, k, stride
    (2, 96, 96, 256, 256, 3, 1), (2, 48, 48, 512, 256, 3, 1), (2, 24, 24, 512, 512, 3, 1), (2, 24, 24, 1280, 512, 3, 1),
    (2, 24, 24, 1152, 512, 3, 1), (1, 24, 24, 1024, 256, 1, 1), (1, 24, 24, 256, 256, 3, 1), (1, 96, 96, 64, 64, 3, 1),
    (2, 96, 96, 64, 64, 3, 1), (1, 48, 48, 128, 128, 3, 1), (1, 96, 96, 64, 256, 1, 1), (2, 48, 48, 256, 512, 3, 2), (2, 25, 23, 72, 40, 3, 1),
    (2, 96, 96, 256, 64, 1, 1), (1, 192, 192, 64, 64, 3, 1), (3, 96, 96, 64, 256, 1, 1), (3, 96, 96, 256, 64, 1, 1),
    (3, 48, 48, 128, 512, 1, 1), (3, 48, 48, 512, 128, 1, 1), (3, 24, 24, 256, 1024, 1, 1), (2, 24, 24, 1024, 256, 1, 1), (3, 96, 96, 64, 64, 1, 1),
]


# plan: tile (bits 0-3) | pixel slices (4-11) | slab flip (12) | ring stages (13-14: 0 = the library's default, 2, 3)
PLANS = ([0, 1 << 12] + [wt | z << 4 | f << 12 for wt in (1, 2) for z in (1, 2, 4, 8, 16, 32, 64, 128) for f in (0, 1)]) if '--tune' in sys.argv else [0]
if '--nst' in sys.argv:      # A/B of the slab ring depth (round 6) on the default tile / slices: two stages, three, the default
    PLANS = [2 << 13, 3 << 13, 0, 2 << 13 | 1 << 12, 3 << 13 | 1 << 12]


def main():
    dev = 'cuda:0'
    for B, H, W, ci, co, k, s in SHAPES:
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        x = torch.randn(B, H, W, ci, device=dev)
        dy = torch.randn(B, Ho, Wo, co, device=dev)
        dw = torch.zeros(co, ci, k, k, device=dev)
        wsb = _lib.query('swem_conv2d_wgrad_workspace', B, H, W, ci, 0, 0, co, k, k, s, pad)
        ws = ops.workspace(wsb, x.device)

        def run():
            _lib.call('swem_conv2d_wgrad_f32', ops._stream(), dy.data_ptr(), x.data_ptr(), ci, H * W * ci, 0, 0, 0, 0, 0, 0,
                      B, H, W, co, k, k, s, pad, 0, dw.data_ptr(), ci, 0, ws.data_ptr(), wsb)
        def timed(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return 1e3 * e0.elapsed_time(e1) / 20
        us = timed(run)
        fl = 2.0 * B * Ho * Wo * co * k * k * ci
        ref = dw.clone()
        line = '%dx%dx%d k%d s%d %4d->%4d  fp32 %7.1f us %6.1f TF |' % (B, H, W, k, s, ci, co, us, fl / us / 1e6)
        if ci % 8 == 0 and co % 8 == 0:
            x3, d3 = ops.presplit(x), ops.presplit(dy)
            for math, plans in ((1, PLANS), (2, PLANS)):
                best = None
                for plan in plans:
                    wsb2 = _lib.query('swem_conv2d_wgrad_bf16x3_workspace', B, H, W, ci, 0, 0, co, k, k, s, pad, plan)
                    ws2 = ops.workspace(wsb2, x.device)

                    def runb():
                        _lib.call('swem_conv2d_wgrad_bf16x3', ops._stream(), d3.data_ptr(), d3.stride(0), x3.data_ptr(), ci,
                                  H * W * ci, x3.stride(0), 0, 0, 0, 0, 0, 0, 0, 0, B, H, W, co, k, k, s, pad, math,
                                  dw.data_ptr(), ci, 0, plan, ws2.data_ptr(), wsb2)
                    t = timed(runb)
                    err = float((dw - ref).abs().max() / ref.abs().max())
                    if best is None or t < best[0]:
                        best = (t, plan, err)
                line += ' %s %7.1f us %6.1f TF plan %#x err %.1e |' % ('bf16x6' if math == 1 else 'bf16', best[0],
                                                                       fl / best[0] / 1e6, best[1], best[2])
            # f16x3 (round 5): fp16 pairs, dY scaled on the device (swem_split_f16x2_scaled_f32), three products
            dy.__dict__['_swem_grad'] = True
            x2, d2 = ops.presplit(x, False, ops.PLANES_F16), ops.presplit(dy, False, ops.PLANES_F16)
            inv = dy.__dict__['_swem_inv']
            def timed_plan(plan_):
                wsb3 = _lib.query('swem_conv2d_wgrad_bf16x3_workspace', B, H, W, ci, 0, 0, co, k, k, s, pad, plan_)
                ws3 = ops.workspace(wsb3, x.device)
                return timed(lambda: _lib.call('swem_conv2d_wgrad_f16x3', ops._stream(), d2.data_ptr(), d2.stride(0), x2.data_ptr(), ci,
                                               H * W * ci, x2.stride(0), 0, 0, 0, 0, 0, 0, 0, 0, B, H, W, co, k, k, s, pad, inv.data_ptr(),
                                               dw.data_ptr(), ci, 0, plan_, ws3.data_ptr(), wsb3))
            best = None
            for plan in PLANS:
                wsb2 = _lib.query('swem_conv2d_wgrad_bf16x3_workspace', B, H, W, ci, 0, 0, co, k, k, s, pad, plan)
                ws2 = ops.workspace(wsb2, x.device)

                def runh():
                    _lib.call('swem_conv2d_wgrad_f16x3', ops._stream(), d2.data_ptr(), d2.stride(0), x2.data_ptr(), ci,
                              H * W * ci, x2.stride(0), 0, 0, 0, 0, 0, 0, 0, 0, B, H, W, co, k, k, s, pad, inv.data_ptr(),
                              dw.data_ptr(), ci, 0, plan, ws2.data_ptr(), wsb2)
                t = timed(runh)
                err = float((dw - ref).abs().max() / ref.abs().max())
                if best is None or t < best[0]:
                    best = (t, plan, err)
            line += ' f16x3 %7.1f us %6.1f TF plan %#x err %.1e |' % (best[0], fl / best[0] / 1e6, best[1], best[2])
            if '--nst' in sys.argv:
                line += ' f16x3 by plan: ' + ' '.join('%#x=%.1f' % (pl_, timed_plan(pl_)) for pl_ in PLANS)
        print(line, flush=True)


if __name__ == '__main__':
    main()
