"""Launched by tests/test_gpu_train.py::test_rccl_single_rank_*: ONE process, SWEM_DIST_SINGLE_RANK=1, torch backend "nccl"
(= RCCL on ROCm).  A one-GPU box cannot run a two-rank RCCL job, but it can put every RCCL call this package makes in front
of librccl once: communicator creation, barrier(device_ids), the counter reduction, the bucketed all-reduces of the training
step eager AND recorded into a HIP graph.  With one rank a SUM all-reduce is the identity, so the step must reproduce the
no-process-group step bit for bit (the parent test compares)."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
os.environ['SWEM_DIST_SINGLE_RANK'] = '1'
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import swem_oracle as O  # noqa: E402  (only its make_cfg: plain config defaults)
from swem_amd import dist as sdist, train  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    out_path, steps = sys.argv[1], int(sys.argv[2])
    opts = set(sys.argv[3:])
    torch.cuda.set_device(0)
    dev = 'cuda:0'
    sdist.init(backend='nccl')
    report = {'backend': dist.get_backend(), 'world': dist.get_world_size(), 'active': sdist.active()}
    sdist.barrier()                                                   # barrier(device_ids=[0]): creates the communicator
    report['counters'] = list(sdist.reduce_counters(10, 1.5, device=dev))
    # an all-reduce recorded into a HIP graph and replayed on changing contents
    x = torch.arange(1 << 20, dtype=torch.float32, device=dev)
    dist.all_reduce(x)                                                # (eager once first: lazy communicator / kernel set-up)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        x.mul_(2.0)
        sdist.allreduce_sum_(x, bucket_bytes=1 << 20, sync=True)      # four buckets
        x.add_(1.0)
    vals = []
    for k in range(3):
        x.fill_(float(k))
        g.replay()
        torch.cuda.synchronize()
        vals.append([float(x.min()), float(x.max())])
    report['captured_allreduce'] = vals                               # expect [2k + 1, 2k + 1]
    # the training step with its collectives really issued
    tc = H.train_cases()
    case = dict(tc['cases']['r18'], hw=[128, 128])
    cfg = O.make_cfg(**case['cfg'])
    model, _ = H.make_model_and_sd(cfg, case['wseed'], dev, pred_scale=tc['pred_scale'])
    frames, init_mask, label, valid = [t.to(dev) for t in H.train_batch(case)]
    tr = train.SWEMTrainer(dict(SOLVER=dict(tc['solver_cfg'], BASE_LR=1e-4), LOSS=tc['loss_cfg'], AMP=False), model,
                           use_graph='graph' in opts, reduce_in_graph='reduce_in_graph' in opts)
    calls = {'n': 0}
    real_ar = dist.all_reduce

    def counting(*a, **k):
        calls['n'] += 1
        return real_ar(*a, **k)
    dist.all_reduce = counting
    hist = []
    real_init = train.random_init_host
    for it in range(steps):
        torch.manual_seed(1000 + it)
        full = real_init(2, case['n'], 128, cfg.NUM_BASES)
        train.random_init_host = lambda B, N, Cc, Lb, _f=full: _f.clone()
        losses, _ = tr.one_step(frames, init_mask, valid, label, 5 + it)
        hist.append([float(losses[k]) for k in ('total_loss', 'main_loss', 'aux_loss')])
    torch.cuda.synchronize()
    dist.all_reduce = real_ar
    report.update(all_reduce_calls=calls['n'], graph=tr._graph is not None, hist=hist)
    torch.save({'param': tr.optimizer.param.detach().cpu(), 'report': report}, out_path)
    print(json.dumps(report))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
