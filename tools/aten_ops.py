#!/usr/bin/env python
"""Which ATen ops (i.e. torch kernels, not libswem_hip.so launches) one steady-state frame still issues, and from where:
a TorchDispatchMode around evaluator.frame_step / the look-ahead chain.   python tools/aten_ops.py [--config a|b]"""
import argparse
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402
import bench  # noqa: E402
from swem_amd import evaluator, synth, weights  # noqa: E402
from swem_amd.swem import SWEM  # noqa: E402
from types import SimpleNamespace  # noqa: E402

NO_KERNEL = ('aten.empty', 'aten.view', 'aten.reshape', 'aten.permute', 'aten.slice', 'aten.select', 'aten.unsqueeze',
             'aten.squeeze', 'aten.expand', 'aten.as_strided', 'aten.detach', 'aten.alias', 'aten.unflatten', 'aten._unsafe_view',
             'aten.t.', 'aten.transpose', 'aten.unbind', 'aten.split', 'aten.flatten', 'aten.sym_', 'aten.is_', 'aten.size',
             'aten.stride', 'aten.lift_fresh', 'aten._local_scalar')


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in NO_KERNEL):
            where = [f for f in traceback.extract_stack() if 'swem_amd' in f.filename or 'bench.py' in f.filename]
            w = where[-1] if where else None
            self.seen[(name, '%s:%d %s' % (os.path.basename(w.filename), w.lineno, w.line) if w else '?')] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lookahead', type=int, default=0)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    model = SWEM(SimpleNamespace(**bench.CFG))
    model.load_state_dict(weights.fill_state_dict(model.state_dict(), seed=3, backbone='resnet50'))
    model = model.eval().to(dev)
    frames, m0 = synth.make_clip(t=8, h=bench.H, w=bench.W, n_obj=2, out_hw=bench.OUT_HW, seed=123)
    runner = bench.FrameRunner(model, frames.to(dev), m0.to(dev))
    for _ in range(3):
        runner.step()
    torch.cuda.synchronize()
    log = Log()
    with torch.no_grad(), log:
        if a.lookahead:
            runner.eager_group(a.lookahead)
        else:
            runner.step()
    torch.cuda.synchronize()
    n = max(a.lookahead, 1)
    print('ATen ops with a kernel behind them in %d eager frame(s):' % n)
    for (name, where), c in sorted(log.seen.items(), key=lambda kv: -kv[1]):
        print('  %3d  %-28s %s' % (c, name, where))


if __name__ == '__main__':
    main()
