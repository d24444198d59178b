#!/usr/bin/env python
"""Per-entry-point timing of the EM / matching kernels at config-B sizes (P = 1620, C = 128, V = 512, L = 256, N objects):
HIP events around 50 back-to-back launches of each entry point.   python tools/em_bench.py [--objects 2]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from swem_amd import ops  # noqa: E402


def timeit(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()          # graph replay: GPU time without the host's launch cost
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--objects', type=int, default=2)
    ap.add_argument('--autotune', action='store_true', help='let the on-device tuner pick the readout GEMM plan first')
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
    kn = ops.em_pack_bases(kappa.view(2 * N, C, L))
    w, zT = ops.em_ew(x, kn, masks.view(2 * N, P), masks.view(2 * N, P), tau, True, True)
    fl_e = 2.0 * P * C * 2 * L * N
    pack = ops.new_pack(N, C, V, L, dev)
    ops.pack_bank(kappa, nu, pack, 0)
    ops.pack_bank(kappa, nu, pack, 1)
    if a.autotune:
        ops.AUTOTUNE = True
        ops.match_packed(x, pack, L, topl, tau)
        ops.AUTOTUNE = False
        print('tuned readout plans:', {k: hex(v) for k, v in ops._MATCH_PLANS.items()})
    rows = [
        ('em_norm_bases', lambda: ops.em_norm_bases(kappa.view(2 * N, C, L)), 0),
        ('em_ew (W+E)', lambda: ops.em_ew(x, kn, masks.view(2 * N, P), masks.view(2 * N, P), tau, True, True), 2 * fl_e),
        ('em_ew (W only)', lambda: ops.em_ew(x, kn, masks.view(2 * N, P), None, tau, True, False), fl_e),
        ('em_ew (E only)', lambda: ops.em_ew(x, kn, None, masks.view(2 * N, P), tau, False, True), fl_e),
        ('em_mstep keys (one launch)', lambda: ops.em_mstep(x, False, zT, kappa.view(2 * N, C, L),
                                                                          zita.view(2 * N, L), P, True), fl_e),
        ('em_mstep values (one launch)', lambda: ops.em_mstep(v, True, zT, nu.view(2 * N, V, L),
                                                                            zita.view(2 * N, L), P), 2.0 * P * V * 2 * L * N),
        ('memorize (T=5)', lambda: ops.memorize(x, v, masks, kappa, nu, zita, T, tau), 0),
        ('match (2 banks, packed in the call)', lambda: ops.match(x, kappa, nu, kappa, nu, topl, tau), 0),
        ('memorize (T=5, packed banks kept)', lambda: ops.memorize(x, v, masks, kappa, nu, zita, T, tau, pack=pack,
                                                                  prior_packed=True, bank=1),
         4.0 * P * L * (C * (3 * T - 1) + V) * N),
        ('match (persistent pack)', lambda: ops.match_packed(x, pack, L, topl, tau), 4.0 * 2 * L * P * (C + V) * N),
    ]
    tot_t = tot_f = 0.0
    for name, fn, fl in rows:
        us = timeit(fn)
        print('%-44s %8.1f us  %6.1f TFLOP/s' % (name, us, fl / us / 1e6 if fl else 0))
        if name.startswith(('memorize', 'match')) and fl:
            tot_t += us
            tot_f += fl
    print('%-44s %8.1f us  %6.1f TFLOP/s = %.1f %% of 157.3' % ('memorize + match', tot_t, tot_f / tot_t / 1e6,
                                                                 100 * tot_f / tot_t / 1e6 / 157.3))


def concurrent(n_streams, objects=2, reps=30, prio=False):
    """memorize + match of n_streams independent sequences, one HIP graph per stream, replayed together (how the product
    runs: bench.py --seqs): aggregate algorithmic TFLOP/s of the EM/matching phase when the GPU is shared."""
    dev = 'cuda:0'
    N, P, C, V, L, T, tau, topl = objects, 1620, 128, 512, 256, 5, 0.05, 64
    graphs, streams, keep = [], [], []      # keep: the graphs replay on these tensors
    for si in range(n_streams):
        g = torch.Generator().manual_seed(10 + si)
        x = torch.randn(P, C, generator=g).to(dev)
        v = torch.randn(N, P, V, generator=g).to(dev)
        masks = torch.rand(N, 2, P, generator=g).to(dev)
        kappa = torch.nn.functional.normalize(torch.randn(N, 2, C, L, generator=g), dim=2).to(dev)
        nu = torch.randn(N, 2, V, L, generator=g).to(dev)
        zita = (torch.rand(N, 2, L, generator=g) * 3 + 0.1).to(dev)
        st = torch.cuda.Stream(priority=-(si % 2)) if prio else ops.new_stream()
        with torch.cuda.stream(st):
            pack = ops.new_pack(N, C, V, L, dev)
            ops.pack_bank(kappa, nu, pack, 0)
            ops.pack_bank(kappa, nu, pack, 1)

            def fn(x=x, v=v, masks=masks, kappa=kappa, nu=nu, zita=zita, pack=pack):
                ops.memorize(x, v, masks, kappa, nu, zita, T, tau, pack=pack, prior_packed=True, bank=1)
                ops.match_packed(x, pack, L, topl, tau)
            for _ in range(2):
                fn()
            st.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(reps):
                    fn()
        graphs.append(gr)
        streams.append(st)
        keep.append(fn)
    torch.cuda.synchronize()
    import time
    for _ in range(2):
        t0 = time.perf_counter()
        for gr, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                gr.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    fl = (4.0 * P * L * (C * (3 * T - 1) + V) + 4.0 * 2 * L * P * (C + V)) * N * n_streams * reps
    print('%d concurrent sequence(s)%s: %7.1f us per memorize+match round, %6.1f TFLOP/s = %.1f %% of 157.3'
          % (n_streams, ' (alternating stream priorities)' if prio else '', 1e6 * dt / reps, fl / dt / 1e12, 100 * fl / dt / 1e12 / 157.3))


if __name__ == '__main__':
    main()
    for ns in (1, 2, 3, 4):
        concurrent(ns)
