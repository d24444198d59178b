#!/usr/bin/env python
"""Which tensors of a steady-state frame are still split by a separate swem_split_bf16x3_f32 / swem_split_f16x2_f32 launch (no producer wrote their
planes), and every libswem_hip.so call of the frame by name.   python tools/split_sites.py [--lookahead 4]"""
import argparse
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import bench  # noqa: E402
from swem_amd import _lib, ops, synth, weights  # noqa: E402
from swem_amd.swem import SWEM  # noqa: E402
from types import SimpleNamespace  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lookahead', type=int, default=4)
    ap.add_argument('--load-plans', default=None)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    model = SWEM(SimpleNamespace(**bench.CFG))
    model.load_state_dict(weights.fill_state_dict(model.state_dict(), seed=3, backbone='resnet50'))
    model = model.eval().to(dev)
    if a.load_plans:
        model.book.load(a.load_plans)
    else:
        model.book.load_shipped()
    frames, m0 = synth.make_clip(t=8, h=bench.H, w=bench.W, n_obj=2, out_hw=bench.OUT_HW, seed=123)
    runner = bench.FrameRunner(model, frames.to(dev), m0.to(dev))
    for _ in range(3):
        runner.step()
    for _ in range(3):
        runner.eager_group(a.lookahead)
    torch.cuda.synchronize()
    calls, splits = collections.Counter(), collections.Counter()
    real = _lib.call

    def counting(name, *args):
        calls[name] += 1
        if name in ('swem_split_bf16x3_f32', 'swem_split_f16x2_f32'):
            where = [f for f in traceback.extract_stack() if 'swem_amd' in f.filename and 'ops.py' not in f.filename]
            w = where[-1] if where else None
            splits['%s:%d %s  [npix %d x C %d]' % (os.path.basename(w.filename), w.lineno, w.line, args[3], args[4]) if w else '?'] += 1
        return real(name, *args)
    _lib.call = counting
    try:
        runner.eager_group(a.lookahead)
    finally:
        _lib.call = real
    torch.cuda.synchronize()
    k = a.lookahead
    print('library calls in %d eager frames (one look-ahead group): %d = %.1f per frame' % (k, sum(calls.values()), sum(calls.values()) / k))
    for n, c in calls.most_common():
        print('  %4d  %s' % (c, n))
    print('separate operand-split launches:')
    for n, c in splits.most_common():
        print('  %4d  %s' % (c, n))


if __name__ == '__main__':
    main()
