#!/usr/bin/env python
"""Writes profiles/r06_profile_stamp.json: the hash of the convolution sources the round's counter profiles (profiles/r06_conv_pmc*,
r06_conv_traffic*) were measured on.  bench.py copies mfma_busy / traffic from those files and flags them `stale` when the sources
it runs on hash differently.   python tools/profile_stamp.py  (run where the profiles are produced)"""
import hashlib
import json
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
SOURCES = ('swem_amd/csrc/conv.hip', 'swem_amd/csrc/lds_dma.h', 'swem_amd/csrc/bf16_split.h', 'swem_amd/csrc/common.h',
           'include/swem_hip.h')


def conv_sources_sha1():
    h = hashlib.sha1()
    for rel in SOURCES:
        with open(os.path.join(ROOT, rel), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


if __name__ == '__main__':
    out = os.path.join(ROOT, 'gpurun_out', 'r06_profile_stamp.json')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, 'w') as f:
        json.dump({'conv_sources_sha1': conv_sources_sha1(), 'sources': list(SOURCES)}, f, indent=1)
    print(out)
