#!/usr/bin/env python
"""memorize + match on config-B sizes in a loop (for rocprofv3 --kernel-trace --stats: per-kernel durations of the EM /
matching launches).   python tools/em_loop.py [--objects 2] [--reps 50]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from swem_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--objects', type=int, default=2)
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--plan', type=lambda t: int(t, 0), default=0x8030111,
                    help='readout GEMM plan (include/swem_hip.h); default: the shipped plan file\'s entry for config B with two objects '
                         '(pre-split f16x3 readout, 64x64 tile, tail split 8), 0 = heuristic fp32')
    a = ap.parse_args()
    dev = 'cuda:0'
    N, P, C, V, L, T, tau, topl = a.objects, 1620, 128, 512, 256, 5, 0.05, 64
    g = torch.Generator().manual_seed(1)
    x = torch.randn(P, C, generator=g).to(dev)
    v = torch.randn(N, P, V, generator=g).to(dev)
    masks = torch.rand(N, 2, P, generator=g).to(dev)
    kappa = torch.nn.functional.normalize(torch.randn(N, 2, C, L, generator=g), dim=2).to(dev)
    nu = torch.randn(N, 2, V, L, generator=g).to(dev)
    zita = (torch.rand(N, 2, L, generator=g) * 3 + 0.1).to(dev)
    if a.plan:
        ops._MATCH_PLANS[(N, C, V, P, L, 2)] = a.plan
    pack = ops.new_pack(N, C, V, L, dev)
    ops.pack_bank(kappa, nu, pack, 0)
    ops.pack_bank(kappa, nu, pack, 1)
    for _ in range(a.reps):
        ops.memorize(x, v, masks, kappa, nu, zita, T, tau, pack=pack, prior_packed=True, bank=1)
        ops.match_packed(x, pack, L, topl, tau)
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
