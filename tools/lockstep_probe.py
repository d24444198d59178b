#!/usr/bin/env python
"""Four sequences per GPU: four independent look-ahead pipelines (what bench.py runs) against ONE lock-step pipeline that batches
the decoder and the value encoder over the objects of all sequences (evaluator.LockstepGraph, round 6).

    python tools/lockstep_probe.py [--seqs 4] [--lookahead 10] [--rounds 6] [--save-plans FILE]
Same models, weights, clips, seeds and plan book as bench.py's default leg; both forms replay HIP graphs of `lookahead` frames per
sequence and are timed over the same number of frames (rounds x lookahead x seqs), the median of three regions each.  Layer shapes
the shipped plan file does not hold (the S x N-object layers, the S x lookahead-frame key encoder) are tuned on the device first.
Also printed: the agreement of the two forms' index maps on the first group (the batched layers run other tiles: equal up to
fp32 summation order, not bitwise)."""
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
    ap.add_argument('--seqs', type=int, default=4)
    ap.add_argument('--lookahead', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=6)
    ap.add_argument('--objects', type=int, default=2)
    ap.add_argument('--lanes', type=int, default=1, help='lock-step pipelines side by side (seqs sequences EACH; the independent form then runs 4 sequences as bench.py does)')
    ap.add_argument('--skip-independent', action='store_true')
    ap.add_argument('--no-forks', action='store_true', help='the per-sequence parts of a lock-step frame one after the other (a linear graph)')
    ap.add_argument('--no-overlap', action='store_true', help="a lane's key-encoder graph behind its chain graph on ONE stream instead of beside it")
    ap.add_argument('--stagger-ms', type=float, default=0.0, help='hold lane i back by i x this many ms once, before the timed replays (phase of the lanes against each other)')
    ap.add_argument('--drop-per-sequence', action='store_true',
                    help='TIMING BOUND ONLY (wrong results): the lanes without their per-sequence parts -- match returns a cached context, '
                         'memorize does nothing -- i.e. what batching EM and matching over a lane could gain at the very most')
    ap.add_argument('--decompose', action='store_true', help='also: the chain graph and the key-encoder graph of each form replayed ALONE on the idle chip')
    ap.add_argument('--save-plans', default=None)
    ap.add_argument('--load-plans', default=None)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    S, k = a.seqs, a.lookahead
    book = ops.PlanBook(fallback=ops.MODEL_FALLBACK)
    if a.load_plans:
        book.load(a.load_plans)
    else:
        book.load_shipped()
    sd = [None]

    def fresh(seed_base):
        """S models with initialised two-bank memories (two eager steps each), as bench.make_runners builds them."""
        rs = []
        for si in range(S):
            model = SWEM(SimpleNamespace(**bench.CFG))
            if sd[0] is None:
                sd[0] = weights.fill_state_dict(model.state_dict(), seed=3, backbone='resnet50')
            model.load_state_dict(sd[0])
            model = model.eval().to(dev)
            model.book = book
            frames_cpu, m0_cpu = synth.make_clip(t=8, h=bench.H, w=bench.W, n_obj=a.objects, out_hw=bench.OUT_HW, seed=123 + si)
            torch.manual_seed(seed_base + si)
            rn = bench.FrameRunner(model, frames_cpu.to(dev), m0_cpu.to(dev))
            for _ in range(2):
                rn.step()
            rs.append(rn)
        torch.cuda.synchronize()
        return rs

    def groups_of(rn):
        cyc = [1 + (rn.i + j) % (rn.t - 1) for j in range(k * (rn.t - 1))]
        return [torch.cat([rn.frames[:, c] for c in cyc[g * k:(g + 1) * k]]).contiguous() for g in range(rn.t - 1)]

    def region(fn, rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(rounds):
            fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def alone(g, reps=3):
        """ms of one replay of a captured graph on the idle chip"""
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def median3(fn):
        ts = sorted(region(fn, a.rounds) for _ in range(3))
        return S * k * a.rounds / ts[1]

    with torch.no_grad():
        # ---------------- A: independent pipelines
        fps_a, first_a = float('nan'), None
        if not a.skip_independent:
            rs = fresh(1234)
            sts = evaluator.overlapping_streams(S)
            ops.AUTOTUNE = True
            for rn, st in zip(rs, sts):
                with torch.cuda.stream(st):
                    rn.enable_graph(pipelined=False, lookahead=k)
                ops.AUTOTUNE = False
            torch.cuda.synchronize()
            first_a = []
            for rn, st in zip(rs, sts):
                with torch.cuda.stream(st):
                    rn.grp = (rn.grp + 1) % len(rn.groups)
                    first_a.append([p.clone() for p in rn.look.run(rn.groups[rn.grp])])
            torch.cuda.synchronize()

            def step_a():
                for rn, st in zip(rs, sts):
                    with torch.cuda.stream(st):
                        rn.grp = (rn.grp + 1) % len(rn.groups)
                        rn.look.run(rn.groups[rn.grp])
            for _ in range(2):
                step_a()
            fps_a = median3(step_a)
            if a.decompose:
                lk = rs[0].look
                print('independent form, ONE sequence alone: %d frame chains %.2f ms, key encoder of %d frames %.2f ms -> x %d sequences = %.2f ms '
                      'of single-stream work per group; wall per group with %d pipelines overlapping: %.2f ms'
                      % (k, alone(lk.cg[lk.p]), k, alone(lk.kg[lk.p]), S, S * (alone(lk.cg[lk.p]) + alone(lk.kg[lk.p])), S, 1e3 * S * k / fps_a))
            del rs
            torch.cuda.empty_cache()
        # ---------------- B: lock step (a.lanes pipelines of S sequences each)
        lanes = []
        sides = evaluator.overlapping_streams(max(2 * a.lanes, S))
        for ln in range(a.lanes):
            rs = fresh(1234 + 100 * ln)
            groups = [groups_of(rn) for rn in rs]
            ng = len(groups[0])
            both = [torch.stack([groups[s][g] for s in range(S)], dim=1).contiguous() for g in range(ng)]       # (k,S,3,H,W)
            if a.drop_per_sequence:
                for rn in rs:
                    m = rn.model
                    real = m._dispatch
                    cache = {}

                    def fake(mode, *args, _real=real, _cache=cache, **kw):
                        if mode == 'match':
                            if 'ctx' not in _cache:
                                _cache['ctx'] = _real(mode, *args, **kw)
                            return _cache['ctx']
                        if mode == 'memorize':
                            return None
                        return _real(mode, *args, **kw)
                    m._dispatch = fake
            ops.AUTOTUNE = ln == 0
            look = evaluator.LockstepGraph([rn.model for rn in rs], rs[0].frames[:, 1].shape, bench.OUT_HW, k,
                                           side_stream=tuple(sides[2 * ln:2 * ln + 2]), forks='none' if a.no_forks else sides[:S], overlap=not a.no_overlap).capture(both[0])
            ops.AUTOTUNE = False
            torch.cuda.synchronize()
            lanes.append([look, both, 0, rs, ops.new_stream()])
        for ln in lanes:
            with torch.cuda.stream(ln[4]):
                ln[0].prime(ln[1][0])
        torch.cuda.synchronize()
        ng = len(lanes[0][1])
        lanes[0][2] = 1
        with torch.cuda.stream(lanes[0][4]):
            first_b = [p.clone() for p in lanes[0][0].run(lanes[0][1][1])]
        torch.cuda.synchronize()

        def step_b():
            for ln in lanes:
                with torch.cuda.stream(ln[4]):          # (a lane's own stream: run() joins its side streams into the caller's)
                    ln[2] = (ln[2] + 1) % ng
                    ln[0].run(ln[1][ln[2]])
        for ln in lanes[1:]:
            ln[2] = 1
            with torch.cuda.stream(ln[4]):
                ln[0].run(ln[1][1])
        for _ in range(2):
            step_b()
        if a.stagger_ms:
            plain, calls = step_b, [0]

            def step_b():          # (every timed region starts from a synchronize: the stagger is applied at its first replay, inside the clock)
                if calls[0] % a.rounds == 0:
                    for i, ln in enumerate(lanes):
                        with torch.cuda.stream(ln[4]):
                            torch.cuda._sleep(int(i * a.stagger_ms * 2.0e6))       # (~2 GHz: cycles per ms)
                calls[0] += 1
                plain()
        fps_b = median3(step_b) * a.lanes
        if a.decompose:
            lk = lanes[0][0]
            print('lock-step form alone: %d lock-step frame chains (%d sequences) %.2f ms, key encoder of %d x %d frames %.2f ms = %.2f ms of '
                  'single-stream work per group; wall per group: %.2f ms'
                  % (k, S, alone(lk.cg[lk.p]), k, S, alone(lk.kg[lk.p]), alone(lk.cg[lk.p]) + alone(lk.kg[lk.p]), 1e3 * a.lanes * S * k / fps_b))
        ops.check_faults()
    same, tot = 0, 0
    for s in range(S if first_a is not None else 0):
        for j in range(k):
            same += int((first_a[s][j][0] == first_b[j][s]).sum())
            tot += first_b[j][s].numel()
    if a.save_plans:
        book.save(a.save_plans)
    print('%d sequences x %d objects, %d frames per replay: independent pipelines %.1f frames/s; %d lock-step pipeline(s) of %d sequences '
          '%.1f frames/s (%+.1f %%); index maps of the first group agree on %.6f of the pixels'
          % (S, a.objects, k, fps_a, a.lanes, S, fps_b, 100 * (fps_b / fps_a - 1), same / max(tot, 1)))


if __name__ == '__main__':
    main()
