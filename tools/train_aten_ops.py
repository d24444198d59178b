#!/usr/bin/env python
"""Which ATen ops (torch kernels, not libswem_hip.so launches) one training step still issues, and from where: a
TorchDispatchMode around one eager SWEMTrainer.one_step on the reference's training shapes.
   python tools/train_aten_ops.py [--lanes 1] [--amp]"""
import argparse
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402
from swem_amd import ops, synth, weights  # noqa: E402
from swem_amd.swem import SWEM  # noqa: E402
from swem_amd.train import SWEMTrainer  # noqa: E402
from types import SimpleNamespace  # noqa: E402

NO_KERNEL = ('aten.empty', 'aten.view', 'aten.reshape', 'aten.permute', 'aten.slice', 'aten.select', 'aten.unsqueeze',
             'aten.squeeze', 'aten.expand', 'aten.as_strided', 'aten.detach', 'aten.alias', 'aten.unflatten', 'aten._unsafe_view',
             'aten.t.', 'aten.transpose', 'aten.unbind', 'aten.split', 'aten.flatten', 'aten.sym_', 'aten.is_', 'aten.size',
             'aten.stride', 'aten.lift_fresh', 'aten._local_scalar', 'aten.narrow', 'aten.record_stream', 'aten.set_')


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in NO_KERNEL):
            where = [f for f in traceback.extract_stack() if 'swem_amd' in f.filename]
            w = where[-1] if where else None
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), None)
            self.seen[(name, '%s:%d %s' % (os.path.basename(w.filename), w.lineno, w.line) if w else '(autograd engine)', shp if w is None else None)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lanes', type=int, default=1)
    ap.add_argument('--clips', type=int, default=4)
    ap.add_argument('--amp', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    cfg = SimpleNamespace(KEYDIM=128, VALDIM=512, NUM_BASES=256, NUM_EM_ITERS=4, EM_TAU=0.05, TOPL=64, SINGLE_OBJ=False, BACKBONE='resnet50')
    model = SWEM(cfg)
    sd = weights.fill_state_dict(model.state_dict(), seed=1, backbone='resnet50')
    sd['decoder.pred.weight'] = sd['decoder.pred.weight'] * 0.02
    model.load_state_dict(sd)
    model = model.to(dev)
    tr = SWEMTrainer(dict(SOLVER=dict(STAGE=0, BASE_LR=2e-5, PRETRAIN_ITERS=[150000, 300000], GAMMA=0.1, OPTIMIZER='AdamW', WEIGHT_DECAY=5e-4),
                          LOSS=dict(NAME='boots_ce', BS_RATIO=0.3, BS_PERIOD=[20000, 70000], AUX='iou', AUX_RATIO=1.0), AMP=a.amp),
                     model, lanes=a.lanes, use_graph=False)
    fr, im, lb = [], [], []
    for i in range(a.clips):
        frames, per = synth.make_clip(t=3, h=384, w=384, n_obj=2, out_hw=(384, 384), seed=50 + i, all_masks=True)
        fr.append(frames[0]); im.append(per[0][0]); lb.append(torch.stack([m[0].argmax(0) for m in per]))
    frames, init_mask, label = torch.stack(fr).to(dev), torch.stack(im).to(dev), torch.stack(lb).to(dev)
    valid = torch.ones(a.clips, 3, device=dev)
    for it in range(2):
        tr.one_step(frames, init_mask, valid, label, 30000 + it)
    torch.cuda.synchronize()
    log = Log()
    with log:
        tr.one_step(frames, init_mask, valid, label, 30002)
    torch.cuda.synchronize()
    total = sum(log.seen.values())
    print('ATen ops with a kernel behind them in one eager step of %d clips, %d lane(s): %d' % (a.clips, a.lanes, total))
    for (name, where, shp), c in sorted(log.seen.items(), key=lambda kv: -kv[1]):
        print('  %3d  %-30s %s %s' % (c, name, where, shp or ''))


if __name__ == '__main__':
    main()
