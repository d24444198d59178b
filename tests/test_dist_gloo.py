"""CPU, world_size 2 over gloo: the multi-GPU path (sequence sharding + counter reduction) of bench.py / dist.py."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from swem_amd import dist as sdist


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    r, lr, w = sdist.init(backend='gloo')
    seqs = list(range(7))
    mine = sdist.shard(seqs, r, w)
    sdist.barrier()
    frames, secs = sdist.reduce_counters(10 * len(mine), 1.0 + r)
    # training: bucketed all-reduce of a flat gradient buffer (several buckets, ragged tail)
    flat = torch.full((1000003,), float(r + 1))
    sdist.allreduce_sum_(flat, bucket_bytes=1 << 20)
    assert float(flat.min()) == float(flat.max()) == 3.0
    # training start: rank 0's parameters and buffers reach every rank (DistributedDataParallel's constructor broadcast)
    lin = torch.nn.BatchNorm1d(5)
    lin.running_mean.fill_(float(r + 7))
    fp = torch.full((1001,), float(r + 1))
    sdist.broadcast_model_(fp, lin)
    assert float(fp.min()) == float(fp.max()) == 1.0 and float(lin.running_mean.max()) == 7.0
    q.put((r, mine, frames, secs))
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, f0, s0), (r1, m1, f1, s1) = out
    assert sorted(m0 + m1) == list(range(7)) and not set(m0) & set(m1)   # a partition: no sequence twice, none lost
    assert f0 == f1 == 70 and s0 == s1 == 2.0                            # SUM of frames, MAX of seconds


def test_single_process_is_a_no_op():
    assert sdist.shard([1, 2, 3], 0, 1) == [1, 2, 3]
    assert sdist.reduce_counters(5, 0.5) == (5, 0.5)
    t = torch.ones(10)
    assert sdist.allreduce_sum_(t) is t and float(t.sum()) == 10.0


def test_launch_ranks_starts_a_real_two_rank_group():
    """bench.py --gpus N without torchrun starts its N ranks through dist.launch_ranks (reference: torch.distributed.launch,
    train.py:22-41): the children must form ONE process group of N ranks and rank 0's line must come back."""
    import json
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_dist_probe.py')
    env = dict(os.environ, RANK='5', WORLD_SIZE='1')          # a stale environment must not leak into the ranks
    rc, text = sdist.launch_ranks(2, [probe, '--gpus', '2'], env=env, timeout=300)
    assert rc == 0, text
    line = json.loads([ln for ln in text.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['ranks'] == 2
    assert line['frames'] == 6 and line['seconds'] == 2.0 and line['argv'] == ['--gpus', '2']
    assert sys.executable


def test_bench_self_launch_comes_before_any_gpu_call():
    """The parent of `python bench.py --gpus 2` must hand over to child ranks before it queries the GPU."""
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py')).read()
    main = src[src.index('def main():'):]
    assert main.index('sdist.launch_ranks(') < main.index('torch.cuda.')


def _worker8(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    r, lr, w = sdist.init(backend='gloo')
    n = sdist.respect_cpu_quota(w)
    mine = sdist.shard(list(range(30)), r, w)            # DAVIS17-val has 30 sequences
    sdist.barrier()
    frames, secs = sdist.reduce_counters(len(mine), 0.5 + 0.125 * r)
    q.put((r, mine, frames, secs, n))
    dist.destroy_process_group()


def test_eight_rank_partition_and_counter_reduction():
    """The 8-GPU inference launch on CPU ranks: sequences i -> rank i mod 8 is a partition of the 30 DAVIS17-val sequences,
    the (frames, seconds) reduction is SUM / MAX over all eight, and every rank sizes its torch pool to an eighth of the
    container's CPU quota (never zero)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    seen = sorted(i for _, mine, _, _, _ in out for i in mine)
    assert seen == list(range(30))
    assert all(len(mine) in (3, 4) for _, mine, _, _, _ in out)
    assert all(f == 30 and s == 0.5 + 0.125 * 7 for _, _, f, s, _ in out)
    assert all(n >= 1 for *_, n in out) and len({n for *_, n in out}) == 1
