#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM conv on the shapes of one config-B frame (GPU box only).
   python tools/conv_bench.py [--reps 20] [--shapes big|all]      (SWEM_CONV_PLAN=wm,wn,ns forces a plan)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from swem_amd import ops  # noqa: E402

SHAPES = [  # B, H, W, Cin, Cout, k, stride, relu_in
    (2, 120, 216, 256, 256, 3, 1, True),
    (2, 30, 54, 512, 512, 3, 1, True),
    (2, 30, 54, 1280, 512, 3, 1, True),
    (2, 60, 108, 512, 256, 3, 1, True),
    (1, 60, 108, 512, 512, 3, 1, False),
    (1, 120, 216, 256, 256, 3, 1, False),
    (2, 120, 216, 64, 64, 3, 1, False),
    (1, 30, 54, 256, 256, 3, 1, False),
    (1, 30, 54, 1024, 256, 1, 1, False),
    (1, 30, 54, 256, 1024, 1, 1, False),
    (1, 120, 216, 64, 256, 1, 1, False),
    (1, 30, 54, 1024, 512, 3, 1, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', type=int, default=-1)
    ap.add_argument('--plan', type=lambda v: int(v, 0), default=None, help='explicit plan hint, e.g. 0x10021')
    ap.add_argument('--fresh', action='store_true', help='re-split the input every launch (bf16x6 plans)')
    ap.add_argument('--graph', action='store_true', help='time a HIP-graph replay of the launches (device time, no host launch cost)')
    ap.add_argument('--small', action='store_true', help='like --dominant, for the small layers (64x64 tiles, 1x1 convolutions)')
    ap.add_argument('--dominant', action='store_true',
                    help='the layers of the dominant f16x3 instantiation (128x128 tile, eight waves, 16x16x32 MFMA) with their '
                         'shipped plans: the set the round-5 kernel experiments are judged on (SWEM_HIP_LIB picks the build)')
    ap.add_argument('--t256', action='store_true', help='with --dominant: the 256x256-tile kernel (plan tile 4 x 4) with the K-split given by --ns')
    ap.add_argument('--ns', type=int, default=0, help='K-split of the --t256 plans (0: as many as fill 256 CUs)')
    ap.add_argument('--sk', action='store_true', help='with --dominant: the stream-K form of each plan (no K-split, plan bits 24-27 = 1); f16x3 plans need profiles/r06_experiments/streamk_f16.patch')
    ap.add_argument('--res', action='store_true', help='with a residual addend (a ResBlock / bottleneck closing convolution)')
    ap.add_argument('--bmul', type=int, default=1, help='multiply every batch size (the look-ahead graphs run ten frames per launch)')
    a = ap.parse_args()
    dominant = {0: 0x670122, 1: 0x670422, 2: 0x670422, 3: 0x670222, 4: 0x670222, 5: 0x670122, 11: 0x670822}
    if a.small:      # the small / byte-bound layers with their shipped f16x3 plans (kernel variants the subset builds hold)
        a.dominant, dominant = True, {6: 0x170111, 7: 0x70411, 8: 0x70211, 9: 0x70111, 10: 0x670122}
    dev = 'cuda:0'
    print('plan=%s' % os.environ.get('SWEM_CONV_PLAN', 'auto'))
    for idx, (B, H, W, ci, co, k, s, relu) in enumerate(SHAPES):
        if a.only >= 0 and idx != a.only:
            continue
        B *= a.bmul
        if a.dominant:
            if idx not in dominant:
                continue
            a.plan = dominant[idx]
            if a.sk:
                a.plan = (a.plan & ~0xf00) | 1 << 24
            if a.t256:
                a.plan = None            # (timed below over tile heights and K-splits: what the tuner would do)
        x = torch.randn(B, H, W, ci, device=dev)
        pack = ops.pack_conv(torch.randn(co, ci, k, k, device=dev) * 0.02, torch.zeros(co, device=dev), None, s, k // 2)
        resid = torch.randn(B, (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1, co, device=dev) if a.res else None

        def run():
            if a.fresh:
                x.__dict__.pop('_swem_split', None)
            return ops.conv2d([x], pack, relu_in=relu, residual=resid, plan=a.plan)
        if a.dominant and a.t256:
            nkb = k * k * ci // 32
            best = None
            for v in (0, 4, 5, 6, 7):
                rows = 32 * v if v else 256
                tiles = -(-B * H * W // rows) * -(-co // 256)
                cands = sorted({1, max(1, min(nkb // 4, 256 // tiles)), max(1, min(nkb // 4, -(-256 // tiles)))}) if not a.ns else [a.ns]
                for ns in cands:
                    a.plan = 0x70044 | ns << 8 | v << 20
                    for _ in range(2):
                        run()
                    torch.cuda.synchronize()
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda._sleep(400_000)
                    t0.record()
                    for _ in range(8):
                        run()
                    t1.record()
                    torch.cuda.synchronize()
                    t = t0.elapsed_time(t1)
                    if best is None or t < best[0]:
                        best = (t, a.plan)
            a.plan = best[1]
        for _ in range(3):
            y = run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if a.graph:      # (small layers: an eager loop measures the host's ~16 us per Python launch, not the kernel)
            st = ops.new_stream()
            st.wait_stream(torch.cuda.current_stream())
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st):
                with torch.cuda.graph(gr, stream=st):
                    for _ in range(a.reps):
                        y = run()
                gr.replay()
                st.synchronize()
                e0.record(st)
                gr.replay()
                e1.record(st)
            torch.cuda.synchronize()
        else:
            e0.record()
            for _ in range(a.reps):
                y = run()
            e1.record()
            torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / a.reps
        fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * co * k * k * ci
        print('%dx%dx%d k%d s%d %4d->%4d  %8.1f us  %6.1f TFLOP/s%s' % (B, H, W, k, s, ci, co, us, fl / us / 1e6,
                                                                       '  plan %#x' % a.plan if a.plan is not None else ''))


if __name__ == '__main__':
    main()
