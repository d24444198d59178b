"""GPU: the training-step kernels (include/swem_hip_train.h) against the oracle's autograd (oracle/swem_oracle.py,
pinned to the reference trainer by tests/golden/g9_*).  Loss / optimizer first, then every backward kernel with
identical inputs, then the whole step."""
import math

import pytest
import torch

from oracle import swem_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, tol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = float((a - b).abs().max())
    scale = float(b.abs().max()) + 1e-30
    assert err <= tol * scale, '%s: max err %.3e (scale %.3e, rel %.3e > %.1e)' % (what, err, scale, err / scale, tol)


LOSS_CFG = dict(NAME='boots_ce', BS_RATIO=0.30, BS_PERIOD=[20, 70], AUX='iou', AUX_RATIO=1.0)


@pytest.mark.parametrize('it', [5, 45, 90])
@pytest.mark.parametrize('use_valid', [True, False])
def test_vos_loss_forward_backward(lib, it, use_valid):
    """losses/__init__.py:34-63: BootstrappedCE (plain CE below start_warm, annealed top-p inside, top-p after) + IoU
    auxiliary loss, values and d/d logits."""
    from swem_amd import losses
    g = torch.Generator().manual_seed(3 + it)
    B, N1, T, Hh, Ww = 2, 3, 2, 72, 80
    scores = (torch.randn(B, N1, T, Hh, Ww, generator=g) * 3).requires_grad_(True)
    valid = torch.tensor([[1., 1., 1.], [1., 1., 0.]]) if use_valid else None
    target = torch.randint(0, N1, (B, T, Hh, Ww), generator=g)
    if use_valid:
        target[1] = target[1].clamp(max=1)
    ref = O.vos_loss(scores, target, it, valid, LOSS_CFG)
    ref['total_loss'].backward()
    crit = losses.VOSLoss(LOSS_CFG, 100, DEV)
    frames = [scores.detach()[:, :, t].contiguous().to(DEV).requires_grad_(True) for t in range(T)]
    out = crit.clip_loss(frames, target.to(DEV), it, None if valid is None else valid.to(DEV))
    assert out['p'] == pytest.approx(ref['p'])
    for k in ('total_loss', 'main_loss', 'aux_loss'):
        assert float(out[k].detach()) == pytest.approx(float(ref[k].detach()), rel=2e-6), k
    out['total_loss'].backward()
    for t in range(T):
        close(frames[t].grad, scores.grad[:, :, t], 2e-5, 'dlogits frame %d' % t)
    # the reference-shaped call (stacked scores) gives the same numbers
    out2 = crit(scores.detach().to(DEV), target.to(DEV), it, None if valid is None else valid.to(DEV))
    assert float(out2['total_loss']) == pytest.approx(float(out['total_loss']), rel=1e-7)


def test_adamw_matches_torch_optimizer(lib):
    """solver/solver.py:38-41: three AdamW steps on a flat buffer vs torch.optim.AdamW on CPU."""
    from swem_amd import optim
    g = torch.Generator().manual_seed(9)
    n = 100003
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * (10.0 ** -i) for i in range(3)]
    pr = p0.clone().requires_grad_(True)
    ref = torch.optim.AdamW([pr], lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    opt = optim.FlatAdamW(p0.to(DEV), lr=2e-5, weight_decay=5e-4)
    for gr in grads:
        pr.grad = gr.clone()
        ref.step()
        opt.grad.copy_(gr.to(DEV))
        opt.step()
    # one fp32 ulp of a parameter of magnitude <= 4 is 2.4e-7; the three updates are ~6e-5 each
    assert float((opt.param.cpu() - pr.detach()).abs().max()) <= 2.5e-7
    close(opt.param.cpu() - p0, pr.detach() - p0, 5e-3, 'AdamW parameter update')
    assert opt.step_count == 3
