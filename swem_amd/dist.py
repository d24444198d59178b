"""Multi-GPU: one process per GPU.  Inference: sequences sharded across ranks, no data-path collective.
Training (data parallel over clips, train.py): one bucketed gradient all-reduce per step over the flat gradient buffer.

Within a sequence frame t needs frame t-1's bases (reference modules.py:183-193), so the time axis does not
shard; sequences are independent.  Rank r takes sequences r, r+W, r+2W, ... (round
robin; the reference's dormant ``val_loader`` uses contiguous ranges, datasets/dataloader.py:39-51 -- either is
a pure partition).  The only communication is one all-reduce of the (frames, seconds) counters at the end
(RCCL on GPUs -- torch backend "nccl" -- or gloo in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def single_rank_group():
    """SWEM_DIST_SINGLE_RANK=1: a process group is created even for ONE process, and the collectives of this module are really
    issued on it instead of being skipped (round 6: the only way a one-GPU box can put the RCCL code path -- communicator
    creation, barrier(device_ids), the bucketed all-reduces, the all-reduce captured inside a HIP graph -- in front of librccl;
    tests/test_gpu_train.py::test_rccl_single_rank_*).  With N = 1 a SUM all-reduce is the identity, so results are unchanged."""
    return os.environ.get('SWEM_DIST_SINGLE_RANK', '0') == '1'


def active():
    """True when this module's collectives are to be issued: a process group of > 1 ranks, or the single-rank rehearsal."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or single_rank_group())


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for a single process, unless
    SWEM_DIST_SINGLE_RANK=1 asks for a one-rank group: `single_rank_group`)."""
    rank, local_rank, world = env_world()
    if (world > 1 or single_rank_group()) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1 and 'MASTER_PORT' not in os.environ:
            import socket                                 # (a one-rank group has nobody to agree a port with: any free one)
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('SWEM_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        dist.init_process_group(backend=backend, init_method='env://', rank=rank, world_size=world)
    return rank, local_rank, world


def launch_ranks(n, argv, env=None, timeout=None):
    """Start `n` ranks of the script `argv[0]` (one process per GPU) as CHILD processes and wait for them: the
    reference is launched through torch.distributed.launch (train.py:22-41, train_swem_s3.sh:23); here a plain
    ``python bench.py --gpus N`` does that launch itself.  Must be called BEFORE the calling process touches the GPU
    (no HIP call, no torch.cuda query): the children are fresh interpreters started by torch.distributed.run, the
    parent only relays their output and exit code -- it never execs over a GPU-initialised process.
    Returns (exit code, captured stdout of the job)."""
    import signal
    import subprocess
    import sys
    import threading
    e = dict(os.environ if env is None else env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    if int(n) > 1 and 'HSA_ENABLE_IPC_MODE_LEGACY' not in e:
        # The image exports HSA_ENABLE_IPC_MODE_LEGACY=0 (the host driver supports dmabuf IPC only; without it RCCL's
        # hipIpcGetMemHandle fails with "invalid argument").  Set it only where the caller's environment does not say anything,
        # and say so: this build has never met RCCL with N > 1, the tweak is taken on the image's word.
        e['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
        sys.stderr.write('[swem_amd.dist] HSA_ENABLE_IPC_MODE_LEGACY was unset: the %d ranks run with =0 (dmabuf IPC)\n' % int(n))
    # --standalone: torchrun picks a free rendezvous port itself (no bind / close / reuse race); --local-addr keeps the
    # rendezvous on the loopback (the container's host name may not resolve)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           '--nproc-per-node', str(int(n))] + list(argv)
    # its own session: on a timeout / Ctrl-C the WHOLE job (torchrun and the rank processes that hold the GPUs) is
    # signalled through its process group, not only the torchrun parent
    p = subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE, start_new_session=True)
    chunks = []

    def pump():                                          # stream the job's stdout as it comes (a hung job shows where)
        for line in iter(p.stdout.readline, b''):
            chunks.append(line)
            if os.environ.get('SWEM_DIST_ECHO'):
                sys.stderr.write(line.decode(errors='replace'))
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    try:
        p.wait(timeout=timeout)
    except (subprocess.TimeoutExpired, KeyboardInterrupt):
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(p.pid, sig)                    # exactly the session this call started
            except ProcessLookupError:
                break
            try:
                p.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        raise
    th.join(timeout=10)
    return p.returncode, b''.join(chunks).decode(errors='replace')


def shard(items, rank, world):
    """Round-robin partition: every item goes to exactly one rank."""
    return [it for i, it in enumerate(items) if i % world == rank]


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == 'nccl':
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def reduce_counters(frames, seconds, device='cpu'):
    """Whole-job totals: SUM of frames over ranks, MAX of the elapsed time over ranks."""
    if not dist.is_initialized():
        return int(frames), float(seconds)
    f = torch.tensor([float(frames)], dtype=torch.float64, device=device)
    s = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(s, op=dist.ReduceOp.MAX)
    return int(round(f.item())), float(s.item())


def allreduce_sum_(flat, bucket_bytes=64 << 20, sync=False):
    """In-place SUM all-reduce of a flat gradient buffer in buckets (training: the reference wraps the model in
    DistributedDataParallel, swem_trainer.py:41-43; here the parameters' gradients already live in ONE buffer, so the
    'buckets' are plain slices).  All buckets are launched asynchronously and waited for together: on xGMI the ring is
    per-link bound, 64 MB slices keep every link busy without serialising the launch latency (58.6 M fp32 gradients =
    4 slices).  The mean over ranks is folded into the loss gradient by the caller (1 / (clips * world))."""
    if not active():
        return flat
    n = max(1, bucket_bytes // flat.element_size())
    if sync:
        # in stream order on the CURRENT stream's dependency chain (no work handles): the form a stream capture records
        for i in range(0, flat.numel(), n):
            dist.all_reduce(flat[i:i + n], op=dist.ReduceOp.SUM)
        return flat
    works = [dist.all_reduce(flat[i:i + n], op=dist.ReduceOp.SUM, async_op=True) for i in range(0, flat.numel(), n)]
    for w in works:
        w.wait()
    return flat


def broadcast_model_(flat_param, module=None, src=0):
    """Rank `src`'s parameters (one flat buffer, optim.flatten_parameters) and the module's buffers (BatchNorm running
    statistics, mean / std) to every rank: what DistributedDataParallel does once in its constructor
    (swem_trainer.py:41-43).  No-op for one process."""
    if not active():
        return flat_param
    dist.broadcast(flat_param, src=src)
    if module is not None:
        for b in module.buffers():
            if b.numel():
                dist.broadcast(b, src=src)
    return flat_param


def allreduce_sum_async(t, bucket_bytes=64 << 20, mean=False):
    """Start the in-place SUM all-reduce of a flat device tensor in buckets and return the work handles (empty for one
    process).  The collectives run on the process group's own stream behind what the CURRENT stream has queued so far; the
    caller overlaps them with later launches and calls ``.wait()`` on the handles before it consumes the result.
    mean: divide by the world size first (loss scalars: basic_trainer.py:105-110)."""
    if not active():
        return []
    if mean:
        t.div_(dist.get_world_size())
    n = max(1, bucket_bytes // t.element_size())
    return [dist.all_reduce(t[i:i + n], op=dist.ReduceOp.SUM, async_op=True) for i in range(0, t.numel(), n)]


def respect_cpu_quota(ranks=1):
    """Size torch's intra-op CPU pool by what the CONTAINER may use, not by the host's logical CPUs.  torch starts one OpenMP
    thread per logical CPU (128-256 on the GPU hosts of this pool) and the threads busy-wait between ops; under a cgroup CPU
    quota (cpu.max, e.g. 16 CPUs per 100 ms period) a single torch CPU op of a few hundred KB then burns the period's budget and
    the kernel stalls EVERY thread of the process until the next period -- the thread feeding the GPU included (the bimodal
    training rate of rounds 1-3).  Call once at start-up of a driver script (bench.py, tools/train_bench.py,
    tests/conftest.py do); `ranks` = processes sharing the container.  Returns the thread count now in force."""
    import torch
    n = torch.get_num_threads()
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, per = f.read().split()[:2]
        if q != 'max':
            n = max(1, min(n, int(int(q) / int(per)) // max(1, int(ranks))))
    except (OSError, ValueError):
        pass
    if hasattr(os, 'sched_getaffinity'):
        n = max(1, min(n, len(os.sched_getaffinity(0)) // max(1, int(ranks))))
    torch.set_num_threads(n)
    return n
