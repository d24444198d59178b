import sys; sys.path.insert(0, '.')
import torch
from oracle import swem_oracle as O
from swem_amd import losses
LOSS_CFG = dict(NAME='boots_ce', BS_RATIO=0.30, BS_PERIOD=[20, 70], AUX='iou', AUX_RATIO=1.0)
for it in (45, 90):
  for use_valid in (True, False):
    g = torch.Generator().manual_seed(3 + it)
    B, N1, T, Hh, Ww = 2, 3, 2, 72, 80
    scores = (torch.randn(B, N1, T, Hh, Ww, generator=g) * 3)
    valid = torch.tensor([[1., 1., 1.], [1., 1., 0.]]) if use_valid else None
    target = torch.randint(0, N1, (B, T, Hh, Ww), generator=g)
    if use_valid: target[1] = target[1].clamp(max=1)
    ref = O.vos_loss(scores, target, it, valid, LOSS_CFG)
    crit = losses.VOSLoss(LOSS_CFG, 100, 'cuda:0')
    frames = [scores[:, :, t].contiguous().cuda() for t in range(T)]
    out = crit.clip_loss(frames, target.cuda(), it, None if valid is None else valid.cuda())
    print(it, use_valid, [float(ref[k]) for k in ('total_loss','main_loss','aux_loss')], [float(out[k]) for k in ('total_loss','main_loss','aux_loss')])
    # per-row check
    raw = torch.nn.functional.cross_entropy(scores, target, reduction='none').view(B, T, -1) if valid is None else None
    if raw is not None:
        k = int(Hh*Ww*ref['p'])
        tk = torch.topk(raw, k, dim=-1)[0]
        print('  ref rows mean', tk.mean(-1), 'kth', tk.min(-1)[0])
