#!/usr/bin/env python
"""Where the HOST time of an eager frame goes (the reference's loop as written is host-bound: ~300 library calls per frame):
cProfile over a few frames of tests/helpers.py::aten_glue_loop at config B.   python tools/host_profile.py [--frames 6]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import bench  # noqa: E402
from swem_amd import synth, weights  # noqa: E402
from swem_amd.swem import SWEM  # noqa: E402
from tests import helpers as H  # noqa: E402
from types import SimpleNamespace  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=8)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    model = SWEM(SimpleNamespace(**bench.CFG))
    model.load_state_dict(weights.fill_state_dict(model.state_dict(), seed=3, backbone='resnet50'))
    model = model.eval().to(dev)
    model.book.load_shipped()
    frames, m0 = synth.make_clip(t=a.frames, h=bench.H, w=bench.W, n_obj=2, out_hw=bench.OUT_HW, seed=123)
    frames, m0 = frames.to(dev), m0.to(dev)
    masks = [m0] + [None] * (a.frames - 1)
    with torch.no_grad():
        H.aten_glue_loop(model, frames, masks, bench.OUT_HW)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        H.aten_glue_loop(model, frames, masks, bench.OUT_HW)
        t_host = time.perf_counter() - t0          # (enqueue only)
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print('%d frames: host enqueue %.2f ms per frame, wall %.2f ms per frame = %.1f frames/s'
              % (a.frames, 1e3 * t_host / a.frames, 1e3 * t_all / a.frames, a.frames / t_all))
        pr = cProfile.Profile()
        pr.enable()
        H.aten_glue_loop(model, frames, masks, bench.OUT_HW)
        pr.disable()
        torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(28)


if __name__ == '__main__':
    main()
