"""Launched by tests/test_dist_gloo.py through swem_amd.dist.launch_ranks: every rank joins a gloo group, rank 0 prints
one JSON line with the world size the group really has."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch.distributed as dist  # noqa: E402

from swem_amd import dist as sdist  # noqa: E402

rank, _, world = sdist.init(backend='gloo')
frames, secs = sdist.reduce_counters(3, 1.0 + rank)
if rank == 0:
    print(json.dumps({'n_gpus': world, 'ranks': dist.get_world_size(), 'frames': frames, 'seconds': secs,
                      'argv': sys.argv[1:]}))
dist.destroy_process_group()
