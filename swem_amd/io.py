"""Input / output staging either side of the timed loop (reference methods/basic_modules/basic_evaluator.py:149-199,
utils/visualization.py:40-43).

* ``stage_sequence``: the evaluator's bicubic resize of the clip to 480x864 (basic_evaluator.py:160) on the device.
* ``IndexMapWriter``: predicted int64 index maps -> uint8 on the device (8x less PCIe traffic), asynchronous copy into
  pinned host memory on a side stream, palette-PNG encoding (``save_seg_mask``) on a small thread pool so that neither the
  copy nor the zlib work sits between two sequences.
Nothing here touches the model; the dataset classes of the reference stay out of scope.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import ops


def stage_sequence(frames, masks, size=(480, 864)):
    """frames (1,T,3,H,W) fp32 in [0,1] on the device, masks (1,1,N+1,H,W) -> (in_frames (1,T,3,480,864), in_masks list
    with the first-frame mask), as basic_evaluator.py:157-163."""
    in_frames = ops.resize_planes(frames[0].float().contiguous(), tuple(size), 'bicubic').unsqueeze(0)
    in_masks = [None] * frames.shape[1]
    in_masks[0] = masks[:, 0].float()
    return in_frames, in_masks


def save_seg_mask(pred, seg_path, palette):
    """utils/visualization.py:40-43: uint8 index map -> palette PNG."""
    from PIL import Image
    img = Image.fromarray(pred)
    if palette is not None:
        img.putpalette(palette)
    img.save(seg_path)


def default_palette(n=256):
    """DAVIS-style bit-interleaved colour map (the reference reads it from assets/davis_palette.png, not shipped here)."""
    pal = np.zeros((n, 3), dtype=np.uint8)
    for i in range(n):
        c, r, g, b = i, 0, 0, 0
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        pal[i] = (r, g, b)
    return pal.flatten().tolist()


class IndexMapWriter:
    def __init__(self, out_dir, palette=None, workers=4):
        self.out_dir = out_dir
        self.palette = default_palette() if palette is None else palette
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.copy_stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.pending = []

    def submit(self, seq_name, preds, first_index=1):
        """preds: list of (1,H,W) int64 device index maps of one sequence (frames first_index, first_index+1, ...).
        Returns immediately; files appear as ``out_dir/seq_name/%05d.png``."""
        d = os.path.join(self.out_dir, seq_name)
        os.makedirs(d, exist_ok=True)
        stacked = torch.cat(preds, dim=0).contiguous()                 # (T-1,H,W) int64, device
        u8 = ops.pack_u8(stacked)
        host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
        ev = torch.cuda.Event()
        self.copy_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.copy_stream):
            host.copy_(u8, non_blocking=True)
            ev.record()
        u8.record_stream(self.copy_stream)

        def write():
            ev.synchronize()
            arr = host.numpy()
            for t in range(arr.shape[0]):
                save_seg_mask(arr[t], os.path.join(d, '%05d.png' % (first_index + t)), self.palette)
            return arr.shape[0]
        self.pending.append(self.pool.submit(write))

    def save_first(self, seq_name, mask_onehot):
        """Frame 0's given annotation (basic_evaluator.py:180-182)."""
        d = os.path.join(self.out_dir, seq_name)
        os.makedirs(d, exist_ok=True)
        idx, _ = ops.argmax_onehot(mask_onehot.float().contiguous(), want_onehot=False)
        save_seg_mask(ops.pack_u8(idx).cpu().numpy()[0], os.path.join(d, '00000.png'), self.palette)

    def close(self):
        n = sum(f.result() for f in self.pending)
        self.pending = []
        self.pool.shutdown()
        return n
