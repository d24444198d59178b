#!/usr/bin/env python
"""Is the process-to-process spread of the training rate a property of the GRAPH INSTANTIATION?  One process, one trainer: measure
the replayed step, drop the captured graphs, let the trainer capture them again, measure again -- several times.  If the rate moves
from capture to capture as it does from process to process, a trainer can keep the fastest of a few instantiations.
   python tools/train_recapture.py [--amp] [--rounds 5] [--steps 30]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from swem_amd import ops, synth, weights  # noqa: E402
from swem_amd.swem import SWEM  # noqa: E402
from swem_amd.train import SWEMTrainer  # noqa: E402
from types import SimpleNamespace  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--amp', action='store_true')
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--load-plans', default=None)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    cfg = SimpleNamespace(KEYDIM=128, VALDIM=512, NUM_BASES=256, NUM_EM_ITERS=4, EM_TAU=0.05, TOPL=64, SINGLE_OBJ=False, BACKBONE='resnet50')
    model = SWEM(cfg)
    sd = weights.fill_state_dict(model.state_dict(), seed=1, backbone='resnet50')
    sd['decoder.pred.weight'] = sd['decoder.pred.weight'] * 0.02
    model.load_state_dict(sd)
    model = model.to(dev)
    ops.AUTOTUNE = not a.load_plans
    tr = SWEMTrainer(dict(SOLVER=dict(STAGE=0, BASE_LR=2e-5, PRETRAIN_ITERS=[150000, 300000], GAMMA=0.1, OPTIMIZER='AdamW', WEIGHT_DECAY=5e-4),
                          LOSS=dict(NAME='boots_ce', BS_RATIO=0.3, BS_PERIOD=[20000, 70000], AUX='iou', AUX_RATIO=1.0), AMP=a.amp), model, lanes=4)
    if a.load_plans:
        tr.book.load(a.load_plans)
    fr, im, lb = [], [], []
    for i in range(4):
        frames, per = synth.make_clip(t=3, h=384, w=384, n_obj=2, out_hw=(384, 384), seed=50 + i, all_masks=True)
        fr.append(frames[0]); im.append(per[0][0]); lb.append(torch.stack([m[0].argmax(0) for m in per]))
    frames, init_mask, label = torch.stack(fr).to(dev), torch.stack(im).to(dev), torch.stack(lb).to(dev)
    valid = torch.ones(4, 3, device=dev)
    for it in range(2):
        tr.one_step(frames, init_mask, valid, label, 30000 + it)
    ops.AUTOTUNE = False
    for r in range(a.rounds):
        tr._graph = None                     # (the trainer captures again on its next step: _eager_steps is already >= 2)
        for it in range(2):
            tr.one_step(frames, init_mask, valid, label, 30000 + it)
        assert tr._graph is not None
        torch.cuda.synchronize()
        t0 = time.time()
        for it in range(a.steps):
            tr.one_step(frames, init_mask, valid, label, 30000 + it)
        ops.spin_sync()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / a.steps
        print('capture %d: %.1f clips/s (%.1f ms per step)' % (r, 4 / dt, 1e3 * dt), flush=True)


if __name__ == '__main__':
    main()
