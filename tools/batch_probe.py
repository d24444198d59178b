#!/usr/bin/env python
"""What batching the memory-dependent conv stages of SEVERAL sequences into one launch would be worth ("next").

    python tools/batch_probe.py [--clips 4]
Today four sequences run as four independent graph pipelines (evaluator.SequencePool): their value encoder, fusion and decoder run
with B = N objects of ONE frame per launch.  Here the same stages (Engine.encode_value + Engine.decoder_logit, config-B sizes, 2
objects) are timed as `clips` back-to-back single-clip calls and as ONE call on a batch of `clips` clips (the Engine supports it:
engine.py per_object), each from a HIP graph on one stream, with the shipped plans and the on-device tuner for the batched shapes
(which the plan file does not hold).  EM / matching are per sequence and not part of this probe."""
import argparse
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import bench  # noqa: E402
from swem_amd import ops, weights  # noqa: E402
from swem_amd.swem import SWEM  # noqa: E402


def graph_time(fn, reps=5):
    for _ in range(2):
        fn()
    st = ops.new_stream()
    st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
        g.replay()
        st.synchronize()
        e0.record(st)
        g.replay()
        e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=4)
    ap.add_argument('--concurrent', action='store_true', help='also: the stages as concurrent pipelines of 1 / 2 / all clips per stream')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    model = SWEM(SimpleNamespace(**bench.CFG))
    model.load_state_dict(weights.fill_state_dict(model.state_dict(), seed=3, backbone='resnet50'))
    model = model.eval().to(dev)
    model.book.load_shipped()
    eng = model.engine()
    N, H, W = 2, bench.H, bench.W
    g = torch.Generator().manual_seed(5)

    def inputs(B):
        masks = torch.rand(B, N + 1, H, W, generator=g)
        masks = (masks / masks.sum(1, keepdim=True)).to(dev)
        return dict(frame=torch.rand(B, 3, H, W, generator=g).to(dev), masks=masks,
                    s16=torch.rand(B, H // 16, W // 16, 1024, generator=g).to(dev),
                    s8=torch.rand(B, H // 8, W // 8, 512, generator=g).to(dev),
                    s4=torch.rand(B, H // 4, W // 4, 256, generator=g).to(dev),
                    ctx=torch.randn(B * N, H // 16, W // 16, 512, generator=g).to(dev))
    one = [inputs(1) for _ in range(a.clips)]
    many = inputs(a.clips)

    def stages(i):
        mv = eng.encode_value(i['frame'], i['masks'], i['s16'])
        return mv, eng.decoder_logit(i['ctx'], i['s8'], i['s4'])
    with torch.no_grad(), ops.use_book(model.book):
        ops.AUTOTUNE = True                      # (the batched shapes are not in the shipped file: tuned here, 256-column tiles included)
        for _ in range(3):                       # tuning + the consumers' plane requests reaching the producers
            stages(many)
            for i in one:
                stages(i)
        ops.AUTOTUNE = False
        t_one = graph_time(lambda: [stages(i) for i in one])
        t_many = graph_time(lambda: stages(many))
        if a.concurrent:
            # the same work as CONCURRENT pipelines (what SequencePool does): `clips` streams of one clip each, clips / 2 streams
            # of two-clip batches, one stream of all clips -- every stream replays its own graph, the host
            # starts them back to back, time = start of the first to end of the last
            def lanes_time(groups, reps=5):
                graphs = []
                for grp in groups:
                    st = ops.new_stream()
                    st.wait_stream(torch.cuda.current_stream())
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.stream(st):
                        with torch.cuda.graph(g, stream=st, **ops.graph_capture_kwargs()):
                            for _ in range(reps):
                                stages(grp)
                    graphs.append((st, g))
                best = None
                for _ in range(4):
                    torch.cuda.synchronize()
                    main = torch.cuda.current_stream()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(main)
                    for st, g in graphs:
                        st.wait_stream(main)
                        with torch.cuda.stream(st):
                            g.replay()
                    for st, g in graphs:
                        main.wait_stream(st)
                    e1.record(main)
                    torch.cuda.synchronize()
                    t = e0.elapsed_time(e1) / reps
                    best = t if best is None else min(best, t)
                return best
            pairs = [inputs(2) for _ in range(a.clips // 2)]
            for _ in range(3):
                ops.AUTOTUNE = True
                for i in pairs:
                    stages(i)
                ops.AUTOTUNE = False
            t_a = lanes_time(one)
            t_b = lanes_time(pairs)
            t_c = lanes_time([many])
            print('the same stages of %d clips as concurrent pipelines: %d streams x 1 clip %.3f ms, %d streams x 2 clips %.3f ms (%+.1f %%), '
                  '1 stream x %d clips %.3f ms (%+.1f %%)' % (a.clips, a.clips, t_a, a.clips // 2, t_b, 100 * (t_b / t_a - 1), a.clips, t_c,
                                                             100 * (t_c / t_a - 1)))
    print('value encoder + decoder of %d clips (2 objects each, 480x864): %d single-clip passes %.3f ms, ONE batched pass %.3f ms '
          '(%.2f x); per clip %.3f -> %.3f ms' % (a.clips, a.clips, t_one, t_many, t_one / t_many, t_one / a.clips, t_many / a.clips))


if __name__ == '__main__':
    main()
