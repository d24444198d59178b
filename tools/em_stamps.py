#!/usr/bin/env python
"""In-kernel clock stamps of one memorize at config-B sizes, from a DEBUG build of the library (-DSWEM_EM_STAMPS, built
into /tmp by this script; the product library contains no stamp code).  Per launch: shader-clock deltas between the
stamps of block 0 / wave 0 and the 100 MHz wall clock, which also gives the gaps between launches.
   python tools/em_stamps.py [--objects 2]"""
import argparse
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--objects', type=int, default=2)
    a = ap.parse_args()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import conv_stamps
    out = os.path.join(ROOT, 'swem_amd', 'libswem_hip_stamps.so')
    if not os.path.exists(out):            # (built here when missing: cross-compile it in the build container, it travels)
        out = conv_stamps.build()
    from swem_amd import _lib
    _lib.LIB_PATH = out
    import torch
    from swem_amd import ops
    lib = _lib.load()
    dev = 'cuda:0'
    N, P, Cc, V, L, T, tau = a.objects, 1620, 128, 512, 256, 5, 0.05
    g = torch.Generator().manual_seed(1)
    x = torch.randn(P, Cc, generator=g).to(dev)
    v = torch.randn(N, P, V, generator=g).to(dev)
    masks = torch.rand(N, 2, P, generator=g).to(dev)
    kappa = torch.nn.functional.normalize(torch.randn(N, 2, Cc, L, generator=g), dim=2).to(dev)
    nu = torch.randn(N, 2, V, L, generator=g).to(dev)
    zita = (torch.rand(N, 2, L, generator=g) * 3 + 0.1).to(dev)
    pack = ops.new_pack(N, Cc, V, L, dev)
    ops.pack_bank(kappa, nu, pack, 0)
    ops.pack_bank(kappa, nu, pack, 1)
    stamps = torch.zeros(64 * 16, dtype=torch.int64, device=dev)
    for _ in range(5):
        ops.memorize(x, v, masks, kappa, nu, zita, T, tau, pack=pack, prior_packed=True, bank=1)
    torch.cuda.synchronize()
    # back-to-back memorize calls so that the stamped one runs behind a full queue (no host launch gaps)
    gr = torch.cuda.CUDAGraph()
    lib.swem_debug_set_stamps.argtypes = [C.c_void_p]
    with torch.cuda.graph(gr):
        ops.memorize(x, v, masks, kappa, nu, zita, T, tau, pack=pack, prior_packed=True, bank=1)
        lib.swem_debug_set_stamps(stamps.data_ptr())
        ops.memorize(x, v, masks, kappa, nu, zita, T, tau, pack=pack, prior_packed=True, bank=1)
        lib.swem_debug_set_stamps(None)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    st = stamps.cpu().view(64, 8, 2)
    names = ['ew', 'mstep'] * T
    prev_end = None
    for i, nm in enumerate(names):
        row = st[i]
        k = int((row[:, 0] != 0).sum())
        if k == 0:
            break
        cyc = [int(row[j, 0] - row[0, 0]) for j in range(k)]
        wall = [(int(row[j, 1]) - int(row[0, 1])) * 10 for j in range(k)]          # ns (100 MHz ticks)
        gap = '' if prev_end is None else 'gap since previous launch\'s last stamp %5d ns' % ((int(row[0, 1]) - prev_end) * 10)
        print('%-6s cycles %s | ns %s | %s' % (nm, cyc, wall, gap))
        prev_end = int(row[k - 1, 1])


if __name__ == '__main__':
    main()
