#!/usr/bin/env python
"""The 1x1 layers of the batched key encoder (B = 4, config B) as they run in the frame -- shipped plan, residual + ReLU, the
output's relu planes written by the epilogue (or planes only for conv1) -- against the bytes they must move:
    python tools/pointwise_layers.py
GB/s = (input planes 4 B + output 4 B fp32 [+ 4 B planes] [+ 4 B residual]) per element / time."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from swem_amd import ops  # noqa: E402

LAYERS = [  # B, H, W, Cin, Cout, kind: 'c1' = conv1 (planes only), 'c3' = conv3 (+ residual, y and relu planes)
    (4, 120, 216, 64, 64, 'c1'), (4, 120, 216, 64, 256, 'c3'), (4, 120, 216, 256, 64, 'c1'),
    (4, 60, 108, 512, 128, 'c1'), (4, 60, 108, 128, 512, 'c3'),
    (4, 30, 54, 1024, 256, 'c1'), (4, 30, 54, 256, 1024, 'c3'),
]


def main():
    dev = 'cuda:0'
    book = ops.PlanBook().load_shipped()
    with ops.use_book(book):
        for B, H, W, ci, co, kind in LAYERS:
            x = torch.randn(B, H, W, ci, device=dev)
            pack = ops.pack_conv(torch.randn(co, ci, 1, 1, device=dev) * 0.05, torch.zeros(co, device=dev), None, 1, 0)
            res = torch.randn(B, H, W, co, device=dev) if kind == 'c3' else None
            nxt = ops.pack_conv(torch.randn(64, co, 1, 1, device=dev) * 0.05, torch.zeros(64, device=dev), None, 1, 0)

            def run():
                y = ops.conv2d([x], pack, relu_out=True, residual=res, planes_only=(kind == 'c1'))
                return y
            for _ in range(3):                       # frames 1-2: the consumer asks for the planes; from then on the frame's form
                y = run()
                ops.conv2d([y], nxt, plan=0x30011)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(2_000_000)
            e0.record()
            for _ in range(20):
                y = run()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 20
            M = B * H * W
            sig = (ci, co, 1, 1, 1, 0, 2, B, H, W)
            byt = M * ci * 4 + M * co * 4 + (0 if y.__dict__.get('_swem_planes_only') else M * co * 4) + (M * co * 4 if res is not None else 0)
            print('%dx%dx%d k1 %4d->%4d %s  plan %#9x  %6.1f us  %6.1f MB  %5.2f TB/s  %6.1f TFLOP/s%s' % (
                B, H, W, ci, co, kind, book.conv.get(sig, 0), us, byt / 1e6, byt / us / 1e6, 2.0 * M * ci * co / us / 1e6,
                '  (planes only)' if y.__dict__.get('_swem_planes_only') else ''))


if __name__ == '__main__':
    main()
