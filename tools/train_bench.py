#!/usr/bin/env python
"""Times SWEMTrainer.one_step (reference swem_trainer.py:59-108) on the reference's training shapes
(configs/config.py: 3 frames of 384x384 per clip, MAX_NUM_OBJS = 2, ResNet-50, K = 256, 4 EM iterations).
   python tools/train_bench.py [--clips 4] [--steps 5] [--warmup 2] [--size 384] [--objects 2] [--amp] [--no-autotune]
Data parallel (BASELINE configs C / D: batch 32 = 4 clips on each of 8 GPUs): one process per GPU under torchrun,
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_bench.py --clips 4
every rank steps its own clips, the flat gradient buffer is all-reduced over RCCL before the optimizer step
(swem_amd.dist.allreduce_sum_); rank 0 prints the whole-job clips/s (clips of all ranks / slowest rank's time)."""
import argparse
import json
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
    ap.add_argument('--clips', type=int, default=4)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--size', type=int, default=384)
    ap.add_argument('--objects', type=int, default=2)
    ap.add_argument('--backbone', default='resnet50')
    ap.add_argument('--no-autotune', action='store_true')
    ap.add_argument('--lanes', type=int, default=None,
                    help='streams the batch is cut over; each steps its share of the clips as ONE batch (default: train.DEFAULT_LANES)')
    ap.add_argument('--cpu-baseline', action='store_true', help='time the oracle\'s training step (CPU) on one clip beside it')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--amp', action='store_true', help='config.AMP: bf16-operand convolutions')
    ap.add_argument('--per-step', action='store_true', help='synchronise and print every step time (stderr)')
    ap.add_argument('--save-plans', default=None)
    ap.add_argument('--load-plans', default=None, help='reuse tuned conv plans (profiler runs)')
    ap.add_argument('--gpus', type=int, default=1, help='ranks (one per GPU); without torchrun the ranks are started here')
    a = ap.parse_args()
    from swem_amd import dist as sdist
    rank, local_rank, world = sdist.env_world()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:      # before any GPU call: the ranks are child processes
        rc, text = sdist.launch_ranks(a.gpus, [os.path.abspath(__file__)] + sys.argv[1:])
        sys.stdout.write(text)
        raise SystemExit(rc)
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    sdist.init()
    sdist.respect_cpu_quota(world)
    cfg = SimpleNamespace(KEYDIM=128, VALDIM=512, NUM_BASES=256, NUM_EM_ITERS=4, EM_TAU=0.05, TOPL=64, SINGLE_OBJ=False,
                          BACKBONE=a.backbone)
    model = SWEM(cfg)
    sd = weights.fill_state_dict(model.state_dict(), seed=1, backbone=a.backbone)
    sd['decoder.pred.weight'] = sd['decoder.pred.weight'] * 0.02
    model.load_state_dict(sd)
    model = model.to(dev)
    ops.AUTOTUNE = not a.no_autotune and not a.load_plans
    tr = SWEMTrainer(dict(SOLVER=dict(STAGE=0, BASE_LR=2e-5, PRETRAIN_ITERS=[150000, 300000], GAMMA=0.1,
                                      OPTIMIZER='AdamW', WEIGHT_DECAY=5e-4),
                          LOSS=dict(NAME='boots_ce', BS_RATIO=0.3, BS_PERIOD=[20000, 70000], AUX='iou', AUX_RATIO=1.0),
                          AMP=a.amp), model, lanes=a.lanes)
    if a.load_plans:
        tr.book.load(a.load_plans)
    fr, im, lb = [], [], []
    for i in range(a.clips):
        frames, per = synth.make_clip(t=3, h=a.size, w=a.size, n_obj=a.objects, out_hw=(a.size, a.size), seed=50 + i + 16 * rank,
                                      all_masks=True)
        lab = torch.stack([m[0].argmax(0) for m in per])
        fr.append(frames[0])
        im.append(per[0][0])
        lb.append(lab)
    frames, init_mask, label = torch.stack(fr).to(dev), torch.stack(im).to(dev), torch.stack(lb).to(dev)
    valid = torch.ones(a.clips, a.objects + 1, device=dev)
    for it in range(a.warmup):
        losses, _ = tr.one_step(frames, init_mask, valid, label, 30000 + it)
    if a.save_plans:
        tr.book.save(a.save_plans)
    ops.AUTOTUNE = False
    for it in range(2):                      # the first step after tuning captures the HIP graph, the second replays it
        losses, _ = tr.one_step(frames, init_mask, valid, label, 30000 + it)
    sdist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    per_step = []
    for it in range(a.steps):
        ts = time.time()
        losses, _ = tr.one_step(frames, init_mask, valid, label, 30000 + it)
        if a.per_step:
            ops.spin_sync()
            per_step.append(1e3 * (time.time() - ts))
    ops.spin_sync()
    torch.cuda.synchronize()
    elapsed = time.time() - t0
    if per_step and rank == 0:
        print('per-step ms: ' + ' '.join('%.1f' % v for v in per_step), file=sys.stderr)
        srt = sorted(per_step)
        print('per-step min %.1f  median %.1f ms' % (srt[0], srt[len(srt) // 2]), file=sys.stderr)
    sdist.barrier()
    total, elapsed = sdist.reduce_counters(a.clips * a.steps, elapsed, device=dev)
    dt = elapsed / a.steps
    a.clips = total // a.steps
    # ---- roofline: useful FLOPs of one step (counted from the step's own launch list, one eager pass: ops.FLOPS) / the replayed
    # step's time, against the dense MFMA peak of the arithmetic the convolutions run in (MI355X_MICROARCH.md: 2500 TFLOP/s bf16 /
    # fp16; f16x3 = three products per useful product: 833; bf16x6: 417; fp32 MFMA: 157.3)
    roof = None
    if not a.no_roofline:
        fl, calls = {}, {'n': 0}
        from swem_amd import _lib
        real_call = _lib.call

        def counting(name, *args_):
            calls['n'] += 1
            return real_call(name, *args_)
        graph, tr._graph = tr._graph, None                     # one EAGER pass of the same step (same plans, same kernels)
        use_graph, tr.use_graph = tr.use_graph, False
        _lib.call = counting
        try:
            with ops.flags(FLOPS=fl):
                tr.one_step(frames, init_mask, valid, label, 30000)
        finally:
            _lib.call = real_call
            tr._graph, tr.use_graph = graph, use_graph
        torch.cuda.synchronize()
        per_rank_clips = a.clips // world
        conv = fl.get('conv_fwd', 0.0) + fl.get('conv_dgrad', 0.0) + fl.get('conv_wgrad', 0.0)
        modes = tr.math_modes or ((2,) if tr.amp else ((0, 1, 7) if tr.f16x3 else (0, 1)))
        hist = tr.book.math_histogram(('math',) + tuple(modes))
        peak, pipe = (2500.0, 'bf16 (config.AMP: one MFMA product per useful product)') if a.amp else \
            ((2500.0 / 3, 'f16x3 (three fp16 MFMA products per useful product)') if hist.get('f16x3', 0) >= max(hist.get('bf16x6', 0), 1)
             else (2500.0 / 6, 'bf16x6 (six bf16 MFMA products per useful product)'))
        ach = (conv + fl.get('em_match', 0.0)) / per_rank_clips * (a.clips / dt) / world / 1e12
        roof = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                'traffic': None, 'pipe': pipe,
                'useful_gflop_per_clip': {k: round(v / per_rank_clips / 1e9, 2) for k, v in fl.items() if not k.endswith('launches')},
                'useful_gflop_per_clip_total': round((conv + fl.get('em_match', 0.0)) / per_rank_clips / 1e9, 2),
                'conv_layer_shapes_by_math': {k: v for k, v in hist.items() if v},
                'library_calls_per_clip': round(calls['n'] / per_rank_clips, 1),
                'conv_launches_per_clip': round((fl.get('conv_launches', 0) + fl.get('wgrad_launches', 0)) / per_rank_clips, 1),
                'note': 'achieved = useful FLOPs of one step (2 x multiply-adds of every forward / data-gradient / weight-gradient '
                        'convolution, unpadded, + the EM / matching GEMMs by SURVEY 8d\'s formulas; counted from the step\'s own launches '
                        'in one eager pass) x steps/s of the replayed step, per GPU; peak = the dense matrix peak of the arithmetic '
                        'most layers run in (the fp32-level step mixes fp32 MFMA / bf16x6 / f16x3 per layer: priced at the fastest '
                        'of them, so frac is a lower bound); kernel-by-kernel times and launches per clip: tools/train_launches.sh '
                        '-> profiles/r06_train_launches_*.csv'}
    cpu = None
    if a.cpu_baseline and rank == 0:
        # the oracle's training step (the reference's autograd graph on torch CPU ops) on ONE clip of the same batch: forward,
        # loss, backward -- what `one_step` does up to the optimizer launch
        from oracle import swem_oracle as O
        import bench
        cpuinfo = bench.host_cpu()
        threads = max(1, min(cpuinfo['physical_cores_available'] or torch.get_num_threads(), torch.get_num_threads()))
        torch.set_num_threads(threads)
        sd_cpu = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        names = {k for k, _ in model.named_parameters()}
        for k, v in sd_cpu.items():
            if k in names and v.dtype.is_floating_point:
                v.requires_grad_(True)
        ocfg = O.make_cfg(**vars(cfg))
        loss_cfg = dict(NAME='boots_ce', BS_RATIO=0.3, BS_PERIOD=[20000, 70000], AUX='iou', AUX_RATIO=1.0, ONLY_VALID_OBJ=True)
        t0c = time.time()
        torch.manual_seed(0)
        O.train_one_step(sd_cpu, ocfg, frames[:1].cpu(), init_mask[:1].cpu(), valid[:1].cpu(), label[:1].cpu(), 30000, loss_cfg)
        dtc = time.time() - t0c
        cpu = {'value': round(1.0 / dtc, 4), 'unit': 'clips/s', 'cores': threads, 'kind': 'port', 'cpu': cpuinfo,
               'sample': 'oracle/swem_oracle.py::train_one_step (forward, loss, backward; no optimizer launch) on ONE clip of the '
                         'same batch: %.1f s on %d torch CPU threads' % (dtc, threads)}
    if rank == 0:
        line = {'metric': 'training clips/s (3 x %dx%d frames, %d objects, %s, %s)' % (
            a.size, a.size, a.objects, a.backbone, 'AMP: bf16 conv operands' if a.amp else 'fp32-accurate'), 'value': a.clips / dt,
            'unit': 'clips/s', 'ms_per_step': dt * 1e3, 'clips_per_step': a.clips,
            'total_loss': float(losses['total_loss']), 'graph': tr._graph is not None, 'lanes': tr.lanes,
            'clips_per_lane': [b1 - b0 for b0, b1 in tr._lane_state['chunks']], 'n_gpus': world,
            ('rccl_ranks' if (torch.distributed.is_initialized() and torch.distributed.get_backend() == 'nccl') else 'ranks'):
                torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
            'optimizer_steps_applied': None if tr.optimizer.applied is None else int(tr.optimizer.applied.item()),
            'optimizer_steps_issued': tr.optimizer.step_count,
            'peak_mem_GB': torch.cuda.max_memory_allocated() / 2 ** 30}
        if roof is not None:
            line['roofline'] = roof
        if cpu is not None:
            line['cpu_baseline'] = cpu
        print(json.dumps(line))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
