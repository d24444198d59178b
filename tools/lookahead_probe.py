#!/usr/bin/env python
"""Where a look-ahead group's time goes (config B, one sequence): the batched key-encoder graph alone, the k frame chains
alone, both on the probed stream pair.   python tools/lookahead_probe.py [--k 4] [--load-plans profiles/r03_tuned_plans.json]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import bench  # noqa: E402
from swem_amd import evaluator, ops, synth, weights  # noqa: E402
from swem_amd.swem import SWEM  # noqa: E402
from types import SimpleNamespace  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--k', type=int, default=4)
    ap.add_argument('--load-plans', default=None)
    ap.add_argument('--reps', type=int, default=20)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    model = SWEM(SimpleNamespace(**bench.CFG))
    model.load_state_dict(weights.fill_state_dict(model.state_dict(), seed=3, backbone='resnet50'))
    model = model.eval().to(dev)
    if a.load_plans:
        model.book.load(a.load_plans)
    ops.AUTOTUNE = not a.load_plans
    frames, m0 = synth.make_clip(t=8, h=bench.H, w=bench.W, n_obj=2, out_hw=bench.OUT_HW, seed=123)
    runner = bench.FrameRunner(model, frames.to(dev), m0.to(dev))
    for _ in range(3):
        runner.step()
    runner.enable_graph(pipelined=True, lookahead=a.k)
    ops.AUTOTUNE = False
    g = runner.look
    for _ in range(2 * a.k):
        runner.step()
    torch.cuda.synchronize()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        ops.spin_sync()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / a.reps
    t_keys = timed(lambda: g.kg[0].replay())
    t_chain = timed(lambda: g.cg[g.p].replay())
    grp = runner.groups[0]
    t_both = timed(lambda: g.run(grp))
    g.overlap = False
    t_serial = timed(lambda: g.run(grp))
    print('k = %d: key-encoder graph (B = %d) %.3f ms = %.3f ms/frame; %d frame chains %.3f ms = %.3f ms/frame; both on the probed '
          'pair %.3f ms = %.3f ms/frame (%.1f frames/s); one stream %.3f ms = %.3f ms/frame (%.1f frames/s)'
          % (a.k, a.k, t_keys, t_keys / a.k, a.k, t_chain, t_chain / a.k, t_both, t_both / a.k, 1e3 * a.k / t_both,
             t_serial, t_serial / a.k, 1e3 * a.k / t_serial))


if __name__ == '__main__':
    main()
