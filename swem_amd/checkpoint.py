"""Checkpoint loading with the single-object -> multi-object weight surgery of the reference
(methods/basic_modules/basic_evaluator.py:104-124, basic_trainer.py:125-131): a stage-0 checkpoint has a 4-channel
value-encoder stem (RGB + mask); the multi-object model takes 5 (RGB + mask + other objects), so one orthogonally
initialised input channel is appended."""
import torch


def adapt_state_dict(state, single_object):
    state = dict(state)
    k = 'value_encoder.conv1.weight'
    if k in state and state[k].shape[1] == 4 and not single_object:
        pads = torch.zeros((64, 1, 7, 7), device=state[k].device)
        torch.nn.init.orthogonal_(pads)
        state[k] = torch.cat([state[k], pads], 1)
    return state


def load_model(model, path_or_state, strict=True, cpu=False):
    """model: swem_amd.SWEM.  Accepts a path (torch.load) or a state dict; unwraps DataParallel-style wrappers."""
    state = path_or_state
    if isinstance(path_or_state, str):
        state = torch.load(path_or_state, map_location='cpu' if cpu else None)
    target = model.module if hasattr(model, 'module') else model
    return target.load_state_dict(adapt_state_dict(state, target.single_object), strict=strict)
