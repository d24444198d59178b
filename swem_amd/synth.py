"""Synthetic structured clips: smooth random background plus
N moving textured blobs, with the one-hot initial mask of the blobs at t=0.
iid-noise frames make the EM chaotic, so benchmarks and
end-to-end parity fixtures use these instead.  Pure torch-CPU, seeded, no file IO.
"""
import math

import torch
import torch.nn.functional as F


def make_clip(t=4, h=480, w=864, n_obj=2, out_hw=None, seed=123, all_masks=False):
    """Returns frames (1,T,3,h,w) in [0,1] and init_mask (1,N+1,Ho,Wo) one-hot float
    (all_masks=True: the list of one-hot masks of every frame instead)."""
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    ho, wo = out_hw if out_hw is not None else (h, w)
    low = torch.rand(1, 3, h // 32 + 2, w // 32 + 2, generator=g)
    bg = F.interpolate(low, size=(h, w), mode='bicubic', align_corners=False).clamp(0, 1)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32),
                            torch.arange(w, dtype=torch.float32), indexing='ij')
    frames = []
    masks0 = None
    per_frame = []
    col = torch.rand(n_obj, 3, generator=g) * 0.6 + 0.2
    cx0 = (torch.rand(n_obj, generator=g) * 0.5 + 0.25) * w
    cy0 = (torch.rand(n_obj, generator=g) * 0.5 + 0.25) * h
    vel = (torch.rand(n_obj, 2, generator=g) - 0.5) * 12.0
    rad = (torch.rand(n_obj, generator=g) * 0.08 + 0.08) * h
    freq = torch.rand(n_obj, generator=g) * 0.15 + 0.1
    for ti in range(t):
        img = bg.clone()[0]
        occ = torch.zeros(h, w, dtype=torch.long)
        for o in range(n_obj):
            cx, cy = cx0[o] + vel[o, 0] * ti, cy0[o] + vel[o, 1] * ti
            inside = ((xx - cx) / (1.3 * rad[o])) ** 2 + ((yy - cy) / rad[o]) ** 2 < 1.0
            tex = 0.5 + 0.5 * torch.sin(freq[o] * (xx - cx)) * torch.cos(freq[o] * (yy - cy))
            for c in range(3):
                img[c] = torch.where(inside, (0.5 * col[o, c] + 0.5 * tex).clamp(0, 1), img[c])
            occ = torch.where(inside, torch.full_like(occ, o + 1), occ)
        frames.append(img)
        if ti == 0 or all_masks:
            oh = F.one_hot(occ, n_obj + 1).permute(2, 0, 1).float()[None]
            mk = (F.interpolate(oh, size=(ho, wo), mode='nearest') if (ho, wo) != (h, w) else oh).contiguous()
            if ti == 0:
                masks0 = mk
            per_frame.append(mk)
    frames = torch.stack(frames)[None].contiguous()
    return frames, (per_frame if all_masks else masks0)
