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
    (2, 96, 96, 64, 64, 3, 1), (1, 48, 48, 128, 128, 3, 1), (1, 96, 96, 64, 256, 1, 1),
]


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
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 20
        fl = 2.0 * B * Ho * Wo * co * k * k * ci
        print('%dx%dx%d k%d %4d->%4d  %8.1f us  %6.1f TFLOP/s  (ws %.0f MB)' % (B, H, W, k, ci, co, us, fl / us / 1e6, wsb / 2 ** 20))


if __name__ == '__main__':
    main()
