"""Launched by tests/test_gpu_train.py through swem_amd.dist.launch_ranks: every rank runs SWEMTrainer.one_step on ITS clip of
a two-clip batch (data parallel, gloo rehearsal on one GPU) for a few steps and rank 0 saves the flat parameters + losses."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import swem_oracle as O  # noqa: E402  (only its make_cfg: plain config defaults)
from swem_amd import dist as sdist, train  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    out_path, steps = sys.argv[1], int(sys.argv[2])
    opts = set(sys.argv[3:])
    rank, local_rank, world = sdist.env_world()
    rccl = 'nccl' in opts                    # one rank per GPU over RCCL; default: both ranks on device 0 over gloo
    dev = 'cuda:%d' % (local_rank if rccl else 0)
    torch.cuda.set_device(local_rank if rccl else 0)
    sdist.init(backend='nccl' if rccl else 'gloo')
    tc = H.train_cases()
    case = dict(tc['cases']['r18'], hw=[128, 128])
    cfg = O.make_cfg(**case['cfg'])
    # rank-local initialisation differs on purpose: the trainer's start-up broadcast must make rank 1 train rank 0's model
    model, _ = H.make_model_and_sd(cfg, case['wseed'] + 7 * rank, dev, pred_scale=tc['pred_scale'])
    frames, init_mask, label, valid = [t.to(dev) for t in H.train_batch(case)]
    tr = train.SWEMTrainer(dict(SOLVER=dict(tc['solver_cfg'], BASE_LR=1e-4), LOSS=tc['loss_cfg'], AMP=False), model,
                           use_graph='graph' in opts, overlap_allreduce='single_allreduce' not in opts)
    hist = []
    real_init = train.random_init_host
    for it in range(steps):
        # every rank draws the bases of the WHOLE batch and keeps its clip's (the single-process run draws them in one call)
        torch.manual_seed(1000 + it)
        full = real_init(2, case['n'], 128, cfg.NUM_BASES)
        train.random_init_host = lambda B, N, Cc, Lb, _f=full, _r=rank: _f[_r:_r + 1].clone()
        losses, _ = tr.one_step(frames[rank:rank + 1], init_mask[rank:rank + 1], valid[rank:rank + 1], label[rank:rank + 1],
                                5 + it)
        hist.append([float(losses[k]) for k in ('total_loss', 'main_loss', 'aux_loss')])
    torch.cuda.synchronize()
    flat = tr.optimizer.param.detach()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    if not rccl:
        flat = flat.cpu()
        gathered = [g.cpu() for g in gathered]
    dist.all_gather(gathered, flat)
    flat, gathered = flat.cpu(), [g.cpu() for g in gathered]
    if rank == 0:
        torch.save({'param': flat, 'same_on_all_ranks': all(torch.equal(g, flat) for g in gathered), 'hist': hist,
                    'world': dist.get_world_size()}, out_path)
        print(json.dumps({'ranks': dist.get_world_size()}))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
