#!/usr/bin/env python
"""Per-frame completion times of the graph-replayed steady-state frame (one sequence), polled: shows whether the box
delivers a steady GPU (platform stalls were seen as completions snapping to a 100 ms grid).
   python tools/step_jitter.py [--frames 200] [--load-plans profiles/r01_tuned_plans.json]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import bench  # noqa: E402
from swem_amd import ops, synth, weights  # noqa: E402
from swem_amd.swem import SWEM  # noqa: E402
from types import SimpleNamespace  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=200)
    ap.add_argument('--load-plans', default=None)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    ops.AUTOTUNE = not a.load_plans
    model = SWEM(SimpleNamespace(**bench.CFG))
    model.load_state_dict(weights.fill_state_dict(model.state_dict(), seed=3, backbone='resnet50'))
    model = model.eval().to(dev)
    if a.load_plans:
        model.book.load(a.load_plans)
    frames, m0 = synth.make_clip(t=8, h=bench.H, w=bench.W, n_obj=2, out_hw=bench.OUT_HW, seed=123)
    runner = bench.FrameRunner(model, frames.to(dev), m0.to(dev))
    for _ in range(3):
        runner.step()
    ops.AUTOTUNE = False
    runner.enable_graph()
    ops.spin_sync()
    ts = []
    for _ in range(a.frames):
        t0 = time.perf_counter()
        runner.step()
        ops.spin_sync()
        ts.append(1e3 * (time.perf_counter() - t0))
    ts_sorted = sorted(ts)
    med = ts_sorted[len(ts) // 2]
    slow = [(i, round(v, 1)) for i, v in enumerate(ts) if v > 1.5 * med]
    print('frames %d  median %.2f ms  min %.2f  max %.2f  total %.1f ms (ideal %.1f)  slow frames (>1.5x median): %s'
          % (len(ts), med, ts_sorted[0], ts_sorted[-1], sum(ts), med * len(ts), slow[:20]))


if __name__ == '__main__':
    main()
