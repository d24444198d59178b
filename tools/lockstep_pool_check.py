#!/usr/bin/env python
"""evaluator.LockstepPool against evaluator.SequencePool at config-B size (480x864, R50, K = 256), the pools' own run() API -- first
frames, graph capture / re-binding and tails included, which bench.py's steady-state loop leaves out.

    python tools/lockstep_pool_check.py [--objects 1 2 3] [--seqs 8] [--frames 33] [--tune]
For every object count: `seqs` synthetic sequences of `frames` frames are evaluated by a SequencePool of four models (round 5's form)
and by a LockstepPool of eight models (two lanes of four sequences); printed: frames/s of both (second run() of each pool: graphs
captured, plans tuned) and the agreement of their index maps.  The shipped plan file holds the lock-step layer shapes of TWO objects;
other object counts run the heuristic tile unless --tune lets the first run() tune them."""
import argparse
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import bench  # noqa: E402
from swem_amd import evaluator, ops, synth, weights  # noqa: E402
from swem_amd.swem import SWEM  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--objects', type=int, nargs='*', default=[2])
    ap.add_argument('--seqs', type=int, default=8)
    ap.add_argument('--frames', type=int, default=33)
    ap.add_argument('--lookahead', type=int, default=10)
    ap.add_argument('--tune', action='store_true')
    ap.add_argument('--save-plans', default=None, help='write the pools\' (shared) plan book there afterwards')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    sd = [None]

    def models(n):
        out = []
        for _ in range(n):
            m = SWEM(SimpleNamespace(**bench.CFG))
            if sd[0] is None:
                sd[0] = weights.fill_state_dict(m.state_dict(), seed=3, backbone='resnet50')
            m.load_state_dict(sd[0])
            out.append(m.eval().to(dev))
        return out

    pool_a = evaluator.SequencePool(models(4), lookahead=a.lookahead)
    pool_b = evaluator.LockstepPool(models(8), lockstep=4, lookahead=a.lookahead)
    pool_b.models[0].book = pool_a.models[0].book          # one book for both pools (same plans, tuned once)
    for m in pool_b.models:
        m.book = pool_a.models[0].book
    for n_obj in a.objects:
        seqs = []
        for si in range(a.seqs):
            frames, m0 = synth.make_clip(t=8, h=bench.H, w=bench.W, n_obj=n_obj, out_hw=bench.OUT_HW, seed=300 + si)
            reps = -(-a.frames // 7)
            frames = torch.cat([frames[:, :1]] + [frames[:, 1:]] * reps, dim=1)[:, :a.frames].contiguous()      # frames 1..7 cycle
            seqs.append((frames.to(dev), m0.to(dev), bench.OUT_HW))
        seeds = list(range(50, 50 + a.seqs))
        res = {}
        for name, pool in (('SequencePool, 4 models', pool_a), ('LockstepPool, 2 lanes x 4', pool_b)):
            ops.AUTOTUNE = a.tune
            pool.run(seqs, seeds=seeds)                   # captures (and tunes)
            ops.AUTOTUNE = False
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = pool.run(seqs, seeds=seeds)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[name] = (out, a.seqs * a.frames / dt)
        (oa, fa), (ob, fb) = res.values()
        same = sum(int((x == y).sum()) for sa, sb in zip(oa, ob) for x, y in zip(sa, sb))
        tot = sum(x.numel() for sa in oa for x in sa)
        print('%d object(s), %d sequences x %d frames (first frames, tails and re-binding included): SequencePool of four %.1f frames/s, '
              'LockstepPool of two lanes x four %.1f frames/s (%+.1f %%); index maps agree on %.6f of the pixels'
              % (n_obj, a.seqs, a.frames, fa, fb, 100 * (fb / fa - 1), same / tot), flush=True)
    if a.save_plans:
        pool_a.models[0].book.save(a.save_plans)


if __name__ == '__main__':
    main()
