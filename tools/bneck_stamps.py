#!/usr/bin/env python
"""In-kernel clock stamps of the fused bottleneck kernel (csrc/bneck.hip; debug build -DSWEM_EM_STAMPS, tools/conv_stamps.py):
block SWEM_STAMP_BLOCK / wave 0: 0 start, 1 first k-block of phase 1 landed, 2 phase-1 loop done, 3 y1 written + first w2 k-block
landed, 4 phase-2 loop done, 5 phase-3 MFMAs done, 6 epilogue issued, 7 stores complete.   python tools/bneck_stamps.py [--frames 10]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=10)
    a = ap.parse_args()
    from swem_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, 'swem_amd', 'libswem_hip_stamps.so')
    import torch
    from swem_amd import ops
    lib = _lib.load()
    lib.swem_debug_set_stamps.argtypes = [C.c_void_p]
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(5)
    B, H, W = a.frames, 120, 216
    mk = lambda co, ci, k: ops.pack_conv((torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5).to(dev), None,
                                         [t.to(dev) for t in (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1,
                                                              torch.randn(co, generator=g) * 0.1, torch.rand(co, generator=g) + 0.5)], 1, k // 2)
    c1, c2, c3 = mk(64, 256, 1), mk(64, 64, 3), mk(256, 64, 1)
    x = torch.randn(B, H, W, 256, generator=g).abs().to(dev)
    stamps = torch.zeros(64 * 16, dtype=torch.int64, device=dev)
    with ops.use_book(ops.PlanBook(fallback=ops.MODEL_FALLBACK)), ops.flags(FUSE_BOTTLENECK=True):
        y = ops.bottleneck(x, c1, c2, c3)
        ops.presplit(y, False, ops.PLANES_F16)
        for _ in range(3):
            ops.bottleneck(x, c1, c2, c3, planes_only=True)
        torch.cuda.synchronize()
        lib.swem_debug_set_stamps(stamps.data_ptr())
        for _ in range(2):
            ops.bottleneck(x, c1, c2, c3, planes_only=True)
        lib.swem_debug_set_stamps(None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.bottleneck(x, c1, c2, c3, planes_only=True)
        e1.record()
        torch.cuda.synchronize()
    print('%dx%dx%d: %.1f us per launch' % (B, H, W, 1e3 * e0.elapsed_time(e1) / 10))
    st = stamps.cpu().view(64, 8, 2)
    for i in range(2):
        row = st[i]
        print('launch %d: cycles %s | ns %s' % (i, [int(row[j, 0] - row[0, 0]) for j in range(8)],
                                                 [(int(row[j, 1]) - int(row[0, 1])) * 10 for j in range(8)]))


if __name__ == '__main__':
    main()
