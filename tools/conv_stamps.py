#!/usr/bin/env python
"""In-kernel clock stamps of the pre-split conv kernel on ONE layer shape (debug build -DSWEM_EM_STAMPS, see
tools/em_stamps.py): where a small layer's 16-24 us go.   python tools/conv_stamps.py [--shape 9] [--plan 0x30011]
Stamps of block 0 / wave 0: 0 kernel start, 1 prologue done (descriptors, tap masks), 2 first k-block landed (after the
barrier), 3 k-loop done, 4 epilogue issued, 5 stores complete."""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def build():
    csrc = os.path.join(ROOT, 'swem_amd', 'csrc')
    out = os.path.join(ROOT, 'swem_amd', 'libswem_hip_stamps.so')
    objs = []
    for name in ('api', 'pointwise', 'train', 'train_conv', 'match'):
        objs.append(os.path.join(csrc, name + '.o'))
    for name in ('em', 'conv', 'bneck'):
        o = '/tmp/%s_stamps.o' % name
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
                               '-DSWEM_EM_STAMPS', '-DSWEM_STAMP_BLOCK=%d' % int(os.environ.get('SWEM_STAMP_BLOCK', '0')),
                               *(['-DSWEM_PROLOGUE_STAMPS'] if os.environ.get('SWEM_PROLOGUE_STAMPS') else []),
                               # (SWEM_STAMPS_SUBSET=1: a fifth of conv.hip's instantiations -- every tile, two-plane kernels,
                               # variants 0 / 1 / 4 / 6 / 8 -- builds in a minute instead of five)
                               *(['-DSWEM_ISA_SUBSET'] if name == 'conv' and os.environ.get('SWEM_STAMPS_SUBSET') else []),
                               '-c', os.path.join(csrc, name + '.hip'), '-o', o])
        objs.append(o)
        if name == 'conv':          # (its second unit: conv_t256_kernel alone)
            o2 = '/tmp/conv_t256_stamps.o'
            subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DSWEM_EM_STAMPS',
                                   '-DSWEM_STAMP_BLOCK=%d' % int(os.environ.get('SWEM_STAMP_BLOCK', '0')), '-DSWEM_CONV_T256_ONLY',
                                   '-c', os.path.join(csrc, 'conv.hip'), '-o', o2])
            objs.append(o2)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', type=int, default=9)
    ap.add_argument('--plan', type=lambda v: int(v, 0), default=0x30011)
    ap.add_argument('--build-only', action='store_true')
    ap.add_argument('--bmul', type=int, default=1, help='multiply the batch size of the shape')
    a = ap.parse_args()
    out = os.path.join(ROOT, 'swem_amd', 'libswem_hip_stamps.so')
    if a.build_only or not os.path.exists(out):
        build()
        if a.build_only:
            return
    from swem_amd import _lib
    _lib.LIB_PATH = out
    import torch
    import conv_bench
    from swem_amd import ops
    lib = _lib.load()
    lib.swem_debug_set_stamps.argtypes = [C.c_void_p]
    dev = 'cuda:0'
    B, H, W, ci, co, k, s, relu = conv_bench.SHAPES[a.shape]
    B *= a.bmul
    x = torch.randn(B, H, W, ci, device=dev)
    pack = ops.pack_conv(torch.randn(co, ci, k, k, device=dev) * 0.02, torch.zeros(co, device=dev), None, s, k // 2)
    stamps = torch.zeros(64 * 16, dtype=torch.int64, device=dev)
    for _ in range(3):
        ops.conv2d([x], pack, relu_in=relu, plan=a.plan)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(4):
            ops.conv2d([x], pack, relu_in=relu, plan=a.plan)
        lib.swem_debug_set_stamps(stamps.data_ptr())
        for _ in range(3):
            ops.conv2d([x], pack, relu_in=relu, plan=a.plan)
        lib.swem_debug_set_stamps(None)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    print('shape %s plan %#x: %.1f us per launch (graph of 7)' % (conv_bench.SHAPES[a.shape], a.plan, 1e3 * e0.elapsed_time(e1) / 7))
    st = stamps.cpu().view(64, 8, 2)
    prev = None
    for i in range(3):
        row = st[i]
        kk = int((row[:, 0] != 0).sum())
        if not kk:
            break
        acc = {j: int(row[j, 0]) for j in range(8) if int(row[j, 1]) == -1}      # accumulated regions (STAMP_ACC_OUT)
        pts = sorted((j for j in range(8) if int(row[j, 0]) != 0 and j not in acc), key=lambda j: int(row[j, 0]))
        print('   stamp order:', pts)
        print('launch %d: cycles %s | ns %s | gap since previous launch\'s last stamp: %s ns | cycles inside the counted vmcnt '
              'waits %s, inside lgkmcnt(0) + s_barrier %s (wave 0 of the block, k-loop)'
              % (i, [int(row[j, 0] - row[0, 0]) for j in pts], [(int(row[j, 1]) - int(row[0, 1])) * 10 for j in pts],
                 '-' if prev is None else (int(row[0, 1]) - prev) * 10, acc.get(6), acc.get(7)))
        prev = int(row[pts[-1], 1])


if __name__ == '__main__':
    main()
