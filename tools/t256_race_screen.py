#!/usr/bin/env python
"""Race screen of conv_t256_kernel's sync structure (cdna_hip_programming.md: "screen it for races over many runs"): eight
256-column tile launches per round -- every tile height, K-splits, the bf16x6 form -- on TWO streams at once, 80 rounds, each result
compared BITWISE with the same plan run alone (the kernels are deterministic: any difference is a race).   python tools/t256_race_screen.py"""
import torch, sys
sys.path.insert(0, ".")
from swem_amd import ops
g = torch.Generator().manual_seed(1)
x = torch.randn(2, 120, 216, 256, generator=g).cuda(); x2 = torch.randn(2, 60, 108, 512, generator=g).cuda()
pk = ops.pack_conv((torch.randn(256, 256, 3, 3, generator=g) * 0.03).cuda()); pk2 = ops.pack_conv((torch.randn(256, 512, 3, 3, generator=g) * 0.02).cuda())
P1 = (0x770144, 0x70144, 0x570144, 0x470244, 0x410144); P2 = (0x570344, 0x70244, 0x670244)
# sequential references of the SAME plans (deterministic kernels: concurrent runs must reproduce them bit for bit)
r1 = [ops.conv2d([x], pk, plan=p).clone() for p in P1]; r2 = [ops.conv2d([x2], pk2, plan=p).clone() for p in P2]
torch.cuda.synchronize()
s1, s2 = ops.new_stream(), ops.new_stream()
bad = {}
for it in range(80):
    with torch.cuda.stream(s1):
        ys = [ops.conv2d([x], pk, plan=p) for p in P1]
    with torch.cuda.stream(s2):
        zs = [ops.conv2d([x2], pk2, plan=p) for p in P2]
    torch.cuda.synchronize()
    for p, y, r in list(zip(P1, ys, r1)) + list(zip(P2, zs, r2)):
        if not torch.equal(y, r):
            bad[hex(p)] = bad.get(hex(p), 0) + 1
print("concurrent-stream screen, 80 rounds x 8 launches, bitwise against the same plan run alone: mismatches", bad or "none")
ops.check_faults()
