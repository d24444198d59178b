"""Deterministic random weights for a SWEM state dict.

There is no network in the build/test environment, so neither the ImageNet trunks
(reference mod_resnet.py:35-38) nor a trained SWEM checkpoint can be fetched.
Benchmarks and parity tests therefore use random weights of the reference's
architecture.  Each tensor is drawn from a torch *CPU* generator seeded by
(base seed, crc32 of the state-dict key), so the same key gives the same numbers
on every machine and independently of module construction order.  Scales are chosen
so activations stay O(1) through the residual stacks (frozen BN with running stats
that do not match the data would otherwise blow the features up).
"""
import math
import zlib

import torch

_IMAGENET = {'mean': [0.485, 0.456, 0.406], 'std': [0.229, 0.224, 0.225]}


def _gen(seed, key):
    g = torch.Generator(device='cpu')
    g.manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))
    return g


def _is_block_tail(key, backbone):
    """Last conv / BN of a residual branch: damped so the trunk variance stays bounded."""
    parts = key.split('.')
    name = parts[-2]
    if parts[0] == 'key_encoder' and parts[1] in ('res2', 'layer2', 'layer3'):
        if 'downsample' in key:
            return False
        return name in (('conv3', 'bn3') if backbone == 'resnet50' else ('conv2', 'bn2'))
    if parts[0] == 'value_encoder' and parts[1].startswith('layer'):
        return 'downsample' not in key and name in ('conv2', 'bn2')
    return name == 'conv2'         # networks.ResBlock.conv2 (fuser, decoder)


def fill_state_dict(sd, seed=0, backbone='resnet50'):
    """Return a new state dict with the keys/shapes/dtypes of ``sd`` and seeded values."""
    out = {}
    for key, ref in sd.items():
        g = _gen(seed, key)
        shape = tuple(ref.shape)
        leaf = key.split('.')[-1]
        name = key.rsplit('.', 1)[0].split('.')[-1]
        tail = _is_block_tail(key, backbone)
        if leaf == 'num_batches_tracked':
            t = torch.zeros(shape, dtype=ref.dtype)
        elif leaf in ('mean', 'std') and len(shape) == 4 and shape[1] == 3:
            t = torch.tensor(_IMAGENET[leaf], dtype=torch.float32).view(1, 3, 1, 1)
        elif leaf == 'running_mean':
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == 'running_var':
            t = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif len(shape) == 4:                                   # conv weight
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 0.5 if tail else 1.0
            if key.endswith('decoder.pred.weight'):
                gain = 2.0
            t = torch.randn(shape, generator=g) * (gain * math.sqrt(2.0 / fan_in))
        elif len(shape) == 2:                                   # linear weight
            t = torch.randn(shape, generator=g) * math.sqrt(1.0 / shape[1])
        elif leaf == 'weight':                                  # BN gamma
            t = (torch.rand(shape, generator=g) * 0.4 + 0.8) * (0.5 if tail else 1.0)
        elif leaf == 'bias':
            is_bn = name.startswith('bn') or name == '1' and 'downsample' in key
            t = torch.randn(shape, generator=g) * (0.05 if is_bn else 0.02)
        else:
            raise KeyError('no init rule for %s %s' % (key, shape))
        out[key] = t.to(ref.dtype)
    return out
