#!/usr/bin/env python
"""Magnitudes of every convolution operand of one steady-state frame at the bench configuration (what bounds the fp16-pair
operand format of the f16x3 arithmetic: `mid` is a normal fp16 number for |x| >= 2^-2, the pair overflows at 65520):
per conv source, max |x|, rms, and the share of its energy carried by elements below 2^-2 / 2^-8.   python tools/act_ranges.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import bench  # noqa: E402
from swem_amd import ops, synth, weights  # noqa: E402
from swem_amd.swem import SWEM  # noqa: E402
from types import SimpleNamespace  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    model = SWEM(SimpleNamespace(**bench.CFG))
    model.load_state_dict(weights.fill_state_dict(model.state_dict(), seed=3, backbone='resnet50'))
    model = model.eval().to(dev)
    frames, m0 = synth.make_clip(t=8, h=bench.H, w=bench.W, n_obj=2, out_hw=bench.OUT_HW, seed=123)
    runner = bench.FrameRunner(model, frames.to(dev), m0.to(dev))
    with ops.flags(PLANES_ONLY=False):
        for _ in range(3):
            runner.step()
        rows = []
        real = ops.conv2d

        def spy(srcs, pack, relu_in=False, **kw):
            for i, s in enumerate(srcs):
                x = torch.relu(s) if relu_in else s
                a = x.abs().double()
                e = float((a * a).sum())
                rows.append(('%dx%dx%dx%d k%d -> %d src%d' % (tuple(s.shape) + (pack.kh, pack.cout, i)), float(a.max()),
                             (e / a.numel()) ** 0.5, float((a[a < 0.25] ** 2).sum()) / max(e, 1e-300),
                             float((a[a < 2.0 ** -8] ** 2).sum()) / max(e, 1e-300)))
            wmax = pack.w.abs().amax(dim=(1, 2, 3))
            rows.append(('    filters: column max |w| from %.2e to %.2e' % (float(wmax.min()), float(wmax.max())), None))
            return real(srcs, pack, relu_in=relu_in, **kw)
        ops.conv2d = spy
        with torch.no_grad():
            runner.step()
        ops.conv2d = real
    torch.cuda.synchronize()
    print('%-44s %10s %10s %12s %12s' % ('conv source', 'max |x|', 'rms', 'E(|x|<2^-2)', 'E(|x|<2^-8)'))
    for r in rows:
        if r[1] is None:
            print(r[0])
        else:
            print('%-44s %10.3g %10.3g %12.3g %12.3g' % r)
    vals = [r for r in rows if r[1] is not None]
    print('over all sources: max |x| %.3g, smallest rms %.3g' % (max(r[1] for r in vals), min(r[2] for r in vals)))


if __name__ == '__main__':
    main()
