#!/usr/bin/env python
"""What a FUSED ResNet-50 bottleneck block could gain, measured with real kernels on both sides ("measure,
don't estimate").

    python tools/bottleneck_fusion_bound.py [--frames 1 10] [--out profiles/r05_bottleneck_fusion_bound.json]

A bottleneck (mod_resnet.py:77-113: 1x1 C->C/4, 3x3 C/4->C/4, 1x1 C/4->C + residual + ReLU) runs here as THREE launches of the
pre-split convolution kernel; inside a stage every tensor between them is an fp16 operand pair (planes only), the block's input
and output too.  A fused kernel -- the 64/128-channel intermediates kept in LDS -- still has to READ the block's input planes and
WRITE its output planes once.  So its time is bounded below by a device copy of one to the other (same bytes, no arithmetic), and

      upper bound of the fusion speed-up of a block  =  time of its three launches  /  time of that copy.

Both are timed here, on the layer shapes of the key encoder's layer1 / layer2 identity blocks (the byte-bound ones, 5 of the
encoder's 16 blocks; layer3's run on 30x54 maps with 1024 channels and are MFMA / launch bound), for one frame (the reference's
loop as written) and for the ten-frame batch the look-ahead graphs run -- eager launches behind a spin kernel, HIP events on the
launch stream (the intervals are device time between back-to-back packets).  The frame-level bound follows from the per-frame
conv time of bench.py's roofline leg."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from swem_amd import ops  # noqa: E402

# (channels C, H, W, blocks of this shape per frame in the ResNet-50 key encoder: layer1 has 3 blocks, layer2 has 4; the first
# block of a stage has a down-sampling branch and another input width -- not counted: 2 + 3 identity blocks)
BLOCKS = [(256, 120, 216, 2), (512, 60, 108, 3)]


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(4_000_000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, nargs='*', default=[1, 10])
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    dev = 'cuda:0'
    book = ops.PlanBook(fallback=ops.MODEL_FALLBACK)
    book.load_shipped()
    g = torch.Generator().manual_seed(5)
    rows = []
    with ops.use_book(book), torch.no_grad():
        for C_, H, W, per_frame in BLOCKS:
            Cm = C_ // 4
            mk = lambda co, ci, k: ops.pack_conv((torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5).to(dev),
                                                 None, [t.to(dev) for t in (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1,
                                                                            torch.randn(co, generator=g) * 0.1, torch.rand(co, generator=g) + 0.5)],
                                                 1, k // 2)
            c1, c2, c3 = mk(Cm, C_, 1), mk(Cm, Cm, 3), mk(C_, Cm, 1)
            c0 = mk(C_, Cm, 1)                  # a producer in front, so that the block's input arrives as planes like in the stage
            for B in a.frames:
                x_in = torch.randn(B, H, W, Cm, generator=g).to(dev)

                def chain(n):
                    x = ops.conv2d([x_in], c0, relu_out=True, planes_only='block')       # (the stage's first block, stand-in)
                    for _ in range(n):              # n identity blocks, as Engine._Block runs them inside a stage
                        y = ops.conv2d([x], c1, relu_out=True, planes_only=True)
                        y = ops.conv2d([y], c2, relu_out=True, planes_only=True)
                        x = ops.conv2d([y], c3, relu_out=True, residual=x, planes_only='block')
                    return x
                with ops.flags(FUSE_BOTTLENECK=False):
                    for _ in range(3):              # the consumers' plane requests reach the producers (PlanBook.hints)
                        x = chain(4)
                only = bool(x.__dict__.get('_swem_planes_only'))
                planes_in = x.__dict__.get('_swem_split', {})
                # a MIDDLE block of the stage (input and output both operand planes): half the difference of 4 and 2 blocks
                with ops.flags(FUSE_BOTTLENECK=False):
                    t_block = 0.5 * (timed(lambda: chain(4), a.reps) - timed(lambda: chain(2), a.reps))
                # the same middle block as ONE launch (swem_bottleneck_f16x3; layer1's geometry only)
                t_fused = None
                fuse_on = ops.flags(FUSE_BOTTLENECK=True)
                fuse_on.__enter__()
                if ops.bottleneck_ok(x_in.new_empty((B, H, W, C_)), c1, c2, c3):
                    def fchain(n):
                        x = ops.conv2d([x_in], c0, relu_out=True, planes_only='block')
                        for _ in range(n):
                            x = ops.bottleneck(x, c1, c2, c3, planes_only='block')
                        return x
                    for _ in range(3):
                        fchain(4)
                    t_fused = 0.5 * (timed(lambda: fchain(4), a.reps) - timed(lambda: fchain(2), a.reps))
                fuse_on.__exit__(None, None, None)
                numel = B * H * W * C_
                src = torch.empty((2, numel), dtype=torch.float16, device=dev).normal_()
                dst = torch.empty_like(src)
                t_copy = timed(lambda: dst.copy_(src), a.reps)
                # a second, tighter-to-reality floor: a read-modify-write pass of the library itself over the same bytes (the
                # split kernel reads fp32 and writes a pair: 4 + 4 bytes per element, the same traffic as planes in -> planes out)
                xf = torch.randn(B * H * W, C_, device=dev)
                sp = torch.empty((2, numel), dtype=torch.float16, device=dev)
                from swem_amd import _lib
                t_split = timed(lambda: _lib.call('swem_split_f16x2_f32', ops._stream(), xf.data_ptr(), sp.data_ptr(), B * H * W, C_, 0, 0),
                                a.reps)
                rows.append({'block': 'bottleneck %d -> %d -> %d on %dx%dx%d' % (C_, Cm, C_, B, H, W), 'frames_in_batch': B,
                             'identity_blocks_of_this_shape_per_frame': per_frame,
                             'input_arrives_as_planes_only': only, 'input_plane_formats': sorted(planes_in),
                             'three_launches_us': round(t_block, 2), 'copy_planes_in_to_planes_out_us': round(t_copy, 2),
                             'fused_one_launch_us': None if t_fused is None else round(t_fused, 2),
                             'fused_speedup': None if t_fused is None else round(t_block / t_fused, 2),
                             'library_streaming_pass_same_bytes_us': round(t_split, 2),
                             'bytes_in_plus_out_MB': round(2 * numel * 4 / 1e6, 1),
                             'fusion_speedup_upper_bound': round(t_block / t_copy, 2),
                             'us_per_frame_now': round(t_block * per_frame / B, 2),
                             'us_per_frame_saved_at_most': round((t_block - t_copy) * per_frame / B, 2)})
                print(rows[-1])
    out = {'what': __doc__.split('\n\n')[0], 'blocks': rows}
    for B in a.frames:
        sel = [r for r in rows if r['frames_in_batch'] == B]
        out['frames_in_batch_%d' % B] = {'us_per_frame_now': round(sum(r['us_per_frame_now'] for r in sel), 1),
                                         'us_per_frame_saved_at_most': round(sum(r['us_per_frame_saved_at_most'] for r in sel), 1)}
    print(json.dumps(out))
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
