"""Optimizer of the training step on one HIP kernel (reference: solver/solver.py).

The reference builds ``optim.AdamW(filter(requires_grad, model.parameters()), lr, betas=(0.9, 0.999), eps=1e-8,
weight_decay)`` (solver/solver.py:38-41) and a ``MultiStepLR`` (:62-70).  Here every trainable parameter of the model is
a view into ONE flat fp32 buffer (and its ``.grad`` a view into a second one), so the whole update is a single
elementwise launch of ``swem_adamw_f32`` over 58.6 M floats (28 bytes per element = 1.6 GB of HBM traffic).
"""
import torch

from . import _lib, ops


class FlatAdamW:
    def __init__(self, param, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8):
        if param.device.type != 'cuda':
            raise RuntimeError('FlatAdamW runs on a HIP device only')
        self.param = param
        self.grad = torch.zeros_like(param)
        self.m = torch.zeros_like(param)
        self.v = torch.zeros_like(param)
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.step_count = 0
        self.applied = None          # device int32: launches of `step(gate=...)` that updated

    def zero_grad(self, set_to_none=False):
        """The gradient buffer is one allocation shared by all parameter views: it is zeroed, never dropped."""
        self.grad.zero_()

    def step(self, gate=None):
        """gate: optional device float tensor; if any element is non-zero when the launch runs, the update is skipped ON THE
        DEVICE (swem_adamw_gated_f32: the found_inf gate of the reference's GradScaler step, basic_trainer.py:222-223) --
        `self.applied` counts the launches that did update, `reconcile()` brings the host's step count back in line."""
        self.step_count += 1
        if gate is None:
            _lib.call('swem_adamw_f32', ops._stream(), self.param.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(),
                      self.v.data_ptr(), self.param.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                      self.weight_decay, self.step_count)
            return
        if self.applied is None:
            # (bias correction uses the HOST's step count: it equals the applied count for every launch that updates, because
            # once the gate closes it stays closed until the host has looked -- the flags come from a sticky word)
            self.applied = torch.full((1,), self.step_count - 1, dtype=torch.int32, device=self.param.device)
        _lib.call('swem_adamw_gated_f32', ops._stream(), self.param.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(),
                  self.v.data_ptr(), self.param.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                  self.weight_decay, self.step_count, gate.data_ptr(), gate.numel(), self.applied.data_ptr())

    def reconcile(self):
        """After a closed gate was seen: the step count the device really applied (synchronises).  Returns the skipped steps."""
        if self.applied is None:
            return 0
        done = int(self.applied.item())
        skipped, self.step_count = self.step_count - done, done
        return skipped

    def state_dict(self):
        return {'m': self.m, 'v': self.v, 'step': self.step_count, 'lr': self.lr}

    def load_state_dict(self, sd):
        self.m.copy_(sd['m'])
        self.v.copy_(sd['v'])
        self.step_count, self.lr = int(sd['step']), float(sd['lr'])
        if self.applied is not None:
            self.applied.fill_(self.step_count)


class MultiStepLR:
    """optim.lr_scheduler.MultiStepLR(optimizer, milestones, gamma) (solver/solver.py:62-70)."""

    def __init__(self, optimizer, milestones, gamma):
        self.optimizer, self.milestones, self.gamma = optimizer, sorted(milestones), gamma
        self.base_lr = optimizer.lr
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        self.optimizer.lr = self.base_lr * self.gamma ** sum(1 for m in self.milestones if self.last_epoch >= m)

    def rewind(self, n):
        """n scheduler steps were taken for optimizer steps the device-side gate skipped (FlatAdamW.reconcile): undo them."""
        if n > 0:
            self.last_epoch -= n + 1
            self.step()

    def state_dict(self):
        return {'last_epoch': self.last_epoch, 'base_lr': self.base_lr}

    def load_state_dict(self, sd):
        self.last_epoch, self.base_lr = int(sd['last_epoch']), float(sd['base_lr'])


def flatten_parameters(params):
    """Re-point every parameter at a view of one flat buffer (16-byte aligned slots) and give it a ``.grad`` view of a
    second flat buffer.  Returns (flat_param, offsets {id(param): (offset, numel)})."""
    params = [p for p in params if p.requires_grad]
    dev = params[0].device
    sizes = [(p.numel() + 3) // 4 * 4 for p in params]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    off, table = 0, {}
    for p, n in zip(params, sizes):
        view = flat[off:off + p.numel()].view(p.shape)
        view.copy_(p.data)
        p.data = view
        table[id(p)] = (off, p.numel())
        off += n
    return flat, table


def make_optimizer(config_solver, model, num_gpu=None):
    """solver/solver.py:30-54 for OPTIMIZER == 'AdamW' (the reference default, configs/config.py:78)."""
    get = (lambda k: config_solver[k]) if isinstance(config_solver, dict) else (lambda k: getattr(config_solver, k))
    if get('OPTIMIZER') != 'AdamW':
        raise NotImplementedError("only the reference's default optimizer (AdamW) is built on HIP")
    lr = get('BASE_LR') * (num_gpu or 1)
    params = [p for p in model.parameters() if p.requires_grad]
    flat, table = flatten_parameters(params)
    opt = FlatAdamW(flat, lr, get('WEIGHT_DECAY'))
    for p in params:
        off, n = table[id(p)]
        p.grad = opt.grad[off:off + n].view(p.shape)
    opt.params = params
    return opt


def make_lr_scheduler(config_solver, optimizer):
    get = (lambda k: config_solver[k]) if isinstance(config_solver, dict) else (lambda k: getattr(config_solver, k))
    stage = get('STAGE')
    steps = get('PRETRAIN_ITERS') if stage == 0 else get('DAVIS_ITERS') if stage == 1 else get('MAINTRAIN_ITERS')
    return MultiStepLR(optimizer, steps, get('GAMMA'))
