#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ by RUNNING THE REFERENCE (read-only, imported from
/root/reference by file path with the shims of SURVEY.md section 8c) on seeded inputs, and checks the CPU
oracle (oracle/swem_oracle.py) against every one of them while doing so.

Run in the build container only (the reference never travels):   python tests/golden/make_golden.py
The fixtures hold inputs/outputs (data), never reference source.  Inputs that are cheap to regenerate
(synthetic clips, seeded weights) are stored as seeds + a checksum instead of the tensors.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
REF = os.environ.get('SWEM_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)

from oracle import swem_oracle as O  # noqa: E402
from swem_amd import synth, weights  # noqa: E402
from swem_amd.swem import SWEM as HipSWEM  # noqa: E402  (parameter containers only; nothing is executed)


# --------------------------------------------------------------------------- reference import shims
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    for pkg, sub in (('methods', 'methods'), ('methods.basic_modules', 'methods/basic_modules'),
                     ('methods.SWEM', 'methods/SWEM')):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[pkg] = m
    mod_resnet = _load('methods.basic_modules.mod_resnet', os.path.join(REF, 'methods/basic_modules/mod_resnet.py'))
    mod_resnet.model_zoo.load_url = lambda *a, **k: {}            # no network: keep the module's own init
    mod_resnet.model_dirs = {'resnet18': '<stub:resnet18>', 'resnet50': '<stub:resnet50>'}   # reference bug, networks.py:8
    # torchvision is not installed: same topology from the reference's own mod_resnet (zero conv biases)
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tvm.resnet18 = lambda pretrained=False: mod_resnet.ResNet(mod_resnet.BasicBlock, [2, 2, 2, 2], 0)
    tvm.resnet50 = lambda pretrained=False: mod_resnet.ResNet(mod_resnet.Bottleneck, [3, 4, 6, 3], 0)
    tv.models = tvm
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = tvm
    real_load = torch.load

    def fake_load(path, *a, **k):
        if isinstance(path, str) and path.startswith('<stub:'):
            name = path[6:-1]
            return getattr(tvm, name)().state_dict()
        return real_load(path, *a, **k)
    torch.load = fake_load
    _load('methods.basic_modules.attentions', os.path.join(REF, 'methods/basic_modules/attentions.py'))
    _load('methods.basic_modules.networks', os.path.join(REF, 'methods/basic_modules/networks.py'))
    modules = _load('methods.SWEM.modules', os.path.join(REF, 'methods/SWEM/modules.py'))
    swem = _load('methods.SWEM.swem', os.path.join(REF, 'methods/SWEM/swem.py'))
    return modules, swem


def import_reference_evaluator():
    """methods/SWEM/swem_evaluator.py itself, so that the fixtures' frame loops are the REFERENCE's own methods
    (SWEMEvaluator.evaluate_davis_seq :58-102, evaluate_ytvos_seq :104-148) and not a restatement of them.  Its base class
    (methods/basic_modules/basic_evaluator.py) imports the repo's dataset / utils packages at module level, which pull cv2,
    tensorboardX, easydict ... -- none of it is used by the two loop methods: empty stand-in MODULES are registered under those
    names for the duration of the import (build container only; nothing of the reference is written anywhere)."""
    saved = {k: sys.modules.get(k) for k in ('datasets', 'utils')}
    ds = types.ModuleType('datasets')
    ds.DAVIS_Test = ds.YTVOS_Test = None
    ut = types.ModuleType('utils')
    for name in ('mkdir', 'init_random_seed', 'setup_logger', 'FrameSecondMeter', 'save_seg_mask', 'save_overlay'):
        setattr(ut, name, None)
    sys.modules['datasets'], sys.modules['utils'] = ds, ut
    try:
        _load('methods.basic_modules.basic_evaluator', os.path.join(REF, 'methods/basic_modules/basic_evaluator.py'))
        ev = _load('methods.SWEM.swem_evaluator', os.path.join(REF, 'methods/SWEM/swem_evaluator.py'))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ev.SWEMEvaluator


class _Recorder:
    """Stands between the reference's loop and the reference's model: forwards every `model(mode, ...)` call unchanged and
    keeps what the modes returned, frame by frame (the per-stage tensors the fixtures store)."""

    def __init__(self, model):
        self.model, self.calls = model, []

    def __call__(self, mode, *args):
        out = self.model(mode, *args)
        self.calls.append((mode, out))
        return out

    def trace(self):
        """One dict per frame 1..t-1 in the layout the fixtures were always written in."""
        rows, cur = [], None
        for mode, out in self.calls[3:]:            # (frame 0: encode_key, encode_value, init)
            if mode == 'encode_key':
                cur = dict(zip(('qk16', 'qv16', 's16', 's8', 's4'), out), mv16=None)
                rows.append(cur)
            elif mode == 'match':
                cur['context'] = out[0]
            elif mode == 'segment':
                cur['logits'] = out[0]
            elif mode == 'encode_value':
                cur['mv16'] = out
        return rows


class _Quiet:
    def info(self, *a, **k):
        pass


_EVALUATOR = []


def _reference_self(model):
    """An instance of the reference's SWEMEvaluator WITHOUT its constructor (which builds datasets, log directories and a
    model from a config file): the loop methods touch self.model and self.logger.info only."""
    if not _EVALUATOR:
        _EVALUATOR.append(import_reference_evaluator())
    ev = object.__new__(_EVALUATOR[0])
    ev.model, ev.logger = model, _Quiet()
    return ev


def ref_evaluate_seq(model, frames, init_masks, out_size, trace=None):
    """SWEMEvaluator.evaluate_davis_seq (swem_evaluator.py:58-102), the reference's own method, on `model`."""
    rec = _Recorder(model)
    preds, scores = _reference_self(rec).evaluate_davis_seq(frames, init_masks, out_size)
    if trace is not None:
        trace.extend(rec.trace())
    return preds, scores


def ref_evaluate_ytvos(model, frames, init_masks, out_size):
    """SWEMEvaluator.evaluate_ytvos_seq (swem_evaluator.py:104-148), the reference's own method, on `model`."""
    return _reference_self(model).evaluate_ytvos_seq(frames, init_masks, out_size)


class SeededInit:
    """Re-seeds the global generator before every 'init' (the only mode that draws: modules.py:170-178), so that each pass of a
    multi-pass evaluation starts from the same random bases."""

    def __init__(self, model, seed):
        self.model, self.seed = model, seed

    def __call__(self, mode, *a):
        if mode == 'init':
            torch.manual_seed(self.seed)
        return self.model(mode, *a)


def ref_evaluate_ms(model, frames, init_masks, out_size, scales, is_flip):
    """SWEMEvaluator.evaluate_davis_seq_ms (swem_evaluator.py:34-57): multi-scale / flip test-time augmentation."""
    return _reference_self(model).evaluate_davis_seq_ms(frames, init_masks, out_size, scales=list(scales), is_flip=is_flip)


def ytvos_masks(per_frame, appear_at):
    """Object 1 is annotated at frame 0, object 2 only from frame `appear_at` on (YouTube-VOS style)."""
    m0 = per_frame[0]
    first = torch.stack([1 - m0[:, 1], m0[:, 1]], 1)                 # (1,2,H,W): bg, obj1
    late = torch.stack([torch.zeros_like(per_frame[appear_at][:, 2]), per_frame[appear_at][:, 2]], 1)   # (1,2,H,W): -, obj2
    masks = [first] + [None] * (len(per_frame) - 1)
    masks[appear_at] = late
    return masks


def structured_keys(P, C, n_clusters, g, noise=0.15, scale=1.0):
    """Keys that look like encoder output: a few cluster centres + noise (iid noise makes EM chaotic).  The
    cluster of a pixel is independent of its position, so foreground and background share appearance and the
    W step's  1 - p_own  stays O(0.1..1) instead of collapsing to rounding noise."""
    centres = torch.randn(n_clusters, C, generator=g)
    assign = torch.randint(0, n_clusters, (P,), generator=g)
    return (centres[assign] + noise * torch.randn(P, C, generator=g)) * scale, assign


def blob_masks(N, h, w, g):
    """Soft fg/bg masks of N objects: rectangles with soft values, independent of the key clusters."""
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    fg = torch.stack([((xx >= (n * w) // (N + 1)) & (xx < ((n + 1) * w) // (N + 1)) & (yy >= h // 5)).float().flatten()
                      for n in range(N)])
    soft = (fg * 0.9 + 0.1 * torch.rand(N, h * w, generator=g)).clamp(0, 1)
    return fg, soft


def maxdiff(a, b):
    return float((a - b).abs().max())


OUT = os.environ.get('SWEM_GOLDEN_OUT', HERE)     # another directory = regenerate beside the committed fixtures and compare


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print('  wrote %-28s %7.1f KB' % (name, os.path.getsize(path) / 1024))


def checksum(t):
    return float(t.double().abs().sum())


def dump_keys(Rswem):
    """State-dict keys and shapes of the reference model for both backbones."""
    import json
    out = {}
    for tag, kw in (('resnet50_mo', dict(BACKBONE='resnet50')),
                    ('resnet18_so', dict(BACKBONE='resnet18', SINGLE_OBJ=True, NUM_BASES=64))):
        ref = Rswem.SWEM(O.make_cfg(**kw))
        # conv biases of the key-encoder trunk exist only because torchvision is stubbed with mod_resnet
        out[tag] = {k: list(v.shape) for k, v in ref.state_dict().items()
                    if not (k.startswith('key_encoder.') and k.endswith('.bias') and '.bn' not in k
                            and 'downsample.1' not in k)}
    with open(os.path.join(OUT, 'g0_state_dict_keys.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print('  wrote g0_state_dict_keys.json', {k: len(v) for k, v in out.items()})


def main():
    torch.set_num_threads(8)
    R, Rswem = import_reference()
    torch.manual_seed(0)
    if '--keys-only' in sys.argv:
        dump_keys(Rswem)
        return
    dump_keys(Rswem)

    # ------------------------------------------------------------------ G1: single E / M / W / nu steps
    print('G1 per-step KATs')
    g = torch.Generator().manual_seed(11)
    h, w, C, V, K, N = 15, 27, 128, 128, 64, 2
    P = h * w
    core = R.SWEMCore(n_bases=K, valdim=V, n_iters=4, tau=0.05, topl=64)
    xk, assign = structured_keys(P, C, 6, g)
    x = xk.t().reshape(1, C, h, w).contiguous()
    xf = x.flatten(2)[:, None, None]
    x_t = xf.transpose(-2, -1)
    kappa = O.l2norm(torch.randn(1, N, 2, C, K, generator=g) * 0.2 + xk[torch.randint(0, P, (K,), generator=g)].t(), -2)
    fg, soft = blob_masks(N, h, w, g)                                                           # N,P
    masks = torch.stack([(1 - fg) * (1 - soft), fg * soft], 1)[None].unsqueeze(-1)                # 1,N,2,P,1
    zita_prev = torch.rand(1, N, 2, 1, K, generator=g) * 5 + 1e-6
    kappa_prev = O.l2norm(torch.randn(1, N, 2, C, K, generator=g), -2)
    with torch.no_grad():
        z = core.swe_step(x_t, kappa, masks)
        kap_m, zita_m = core.swm_step(z, xf, kappa_prev, zita_prev)
        wgt = core.sww_step(kappa, x_t, masks)
    v = torch.randn(1, N, V, h, w, generator=g)
    nu_prev = torch.randn(1, N, 2, V, K, generator=g)
    nu = (zita_prev * nu_prev + torch.matmul(v.flatten(3).unsqueeze(2), z)) / zita_m
    assert maxdiff(O.e_step(x_t, kappa, masks, 0.05), z) == 0
    ok, oz = O.m_step(z, xf, kappa_prev, zita_prev)
    assert maxdiff(ok, kap_m) == 0 and maxdiff(oz, zita_m) == 0
    assert maxdiff(O.w_step(kappa, x_t, masks, 0.05), wgt) == 0
    save('g1_steps.npz', x=x, kappa=kappa, masks=masks, z=z, kappa_prev=kappa_prev, zita_prev=zita_prev,
         kappa_m=kap_m, zita_m=zita_m, weights=wgt, v=v, nu_prev=nu_prev, nu=nu, tau=0.05)

    # ------------------------------------------------------------------ G2: swem() over two frames, G3: matching
    print('G2/G3 two-frame memorize + matching')
    core = R.SWEMCore(n_bases=K, valdim=V, n_iters=4, tau=0.05, topl=64)
    ocore = O.Core(K, V, 4, 0.05, 64)
    inits = []
    orig_init = core.random_init

    def capture(*a, **k):
        out = orig_init(*a, **k)
        inits.append([t.clone() for t in out])
        return out
    core.random_init = capture
    frames_in = []
    for f in range(2):
        xk, assign = structured_keys(P, C, 6, g)
        xx = xk.t().reshape(1, C, h, w).contiguous()
        vv = torch.randn(1, N, V, h, w, generator=g)
        fg, sf = blob_masks(N, h, w, g)
        mm = torch.stack([(1 - fg) * (1 - sf), fg * sf], 1).view(1, N, 2, h, w)
        frames_in.append((xx, vv, mm))
    qx, _ = structured_keys(P, C, 6, g)
    qk = qx.t().reshape(1, C, h, w).contiguous()
    qv = torch.randn(1, V, h, w, generator=g)
    out = {}
    with torch.no_grad():
        torch.manual_seed(5)
        core.memorize(*frames_in[0])
        b0 = {k: t.clone() for k, t in core.memories['first'].bases.items()}
        S1, mem1 = core.get_affinity(R.l2norm(qk, 1), R.l2norm(core.get_mem()[0], -2), core.get_mem()[1])
        ctx1, _ = core.matching(qk, qv)
        core.memorize(*frames_in[1])
        b1 = {k: t.clone() for k, t in core.memories['update'].bases.items()}
        S2, mem2 = core.get_affinity(R.l2norm(qk, 1), R.l2norm(core.get_mem()[0], -2), core.get_mem()[1])
        ctx2, _ = core.matching(qk, qv)
        torch.manual_seed(5)
        ob0 = ocore.memorize(*frames_in[0])
        om1, oq1, os1, _ = ocore.match_features(qk, qv)
        ob1 = ocore.memorize(*frames_in[1])
        om2, oq2, os2, _ = ocore.match_features(qk, qv)
    print('   zita frame0: max %.3g, live fraction %.3f; frame1: max %.3g, live %.3f' % (
        float(b0['zita'].max()), float((b0['zita'] >= 1e-3).float().mean()), float(b1['zita'].max()),
        float((b1['zita'] >= 1e-3).float().mean())))
    for k in b0:
        assert maxdiff(ob0[k], b0[k]) == 0, k
        assert maxdiff(ob1[k], b1[k]) == 0, k
    # the oracle's cumsum vs the reference's python prefix loop
    print('   oracle vs reference: S', maxdiff(os1, S1), maxdiff(os2, S2), ' mem_out', maxdiff(om1, mem1.flatten(0, 1)),
          maxdiff(om2, mem2.flatten(0, 1)))
    assert maxdiff(os1, S1) == 0 and maxdiff(os2, S2) == 0
    assert maxdiff(om1, mem1.flatten(0, 1)) == 0 and maxdiff(om2, mem2.flatten(0, 1)) == 0
    fl_sd = {'swem_core.fusion_layer.' + k: t for k, t in core.fusion_layer.state_dict().items()}
    octx2 = O.fusion_layer(fl_sd, torch.cat([om2, oq2, os2], 1))
    assert maxdiff(octx2, ctx2) < 1e-5
    save('g2_memorize.npz', x0=frames_in[0][0], v0=frames_in[0][1], m0=frames_in[0][2], x1=frames_in[1][0],
         v1=frames_in[1][1], m1=frames_in[1][2], init_kappa=inits[0][0], init_nu=inits[0][1], init_zita=inits[0][2],
         kappa0=b0['kappa'], nu0=b0['nu'], zita0=b0['zita'], kappa1=b1['kappa'], nu1=b1['nu'], zita1=b1['zita'])
    # (the GLU fusion conv on top of these is pinned by the full-network clips below)
    save('g3_matching.npz', qk=qk, S1=S1, mem1=mem1, S2=S2, mem2=mem2)

    # ------------------------------------------------------------------ G6: config A, 2-frame clip, full network
    def run_clip(cfgkw, t, hh, ww, n_obj, out_hw, seed, wseed, tag, sub, double_floor=False):
        cfg = O.make_cfg(**cfgkw)
        ref = Rswem.SWEM(cfg)
        ref.eval()
        mine = HipSWEM(cfg)
        sd = weights.fill_state_dict(mine.state_dict(), seed=wseed, backbone=cfg.BACKBONE)
        missing, unexpected = ref.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        # only the zero biases of the torchvision-stub trunk may be missing (torchvision convs have none)
        assert all(k.startswith('key_encoder.') and k.endswith('.bias') for k in missing), missing
        assert set(ref.state_dict().keys()) - set(missing) == set(sd.keys())
        frames, m0 = synth.make_clip(t=t, h=hh, w=ww, n_obj=n_obj, out_hw=out_hw, seed=seed)
        init_masks = [m0] + [None] * (t - 1)
        tr_ref, tr_orc = [], []
        with torch.no_grad():
            torch.manual_seed(77)
            preds, scores = ref_evaluate_seq(ref, frames, init_masks, out_hw, tr_ref)
            torch.manual_seed(77)
            om = O.Model(sd, cfg)
            opreds, oscores = O.evaluate_seq(om, frames, init_masks, out_hw, tr_orc)
        for i in range(t - 1):
            d = maxdiff(tr_orc[i]['logits'], tr_ref[i]['logits'])
            agree = float((opreds[i] == preds[i]).float().mean())
            print('   %s frame %d: oracle vs reference  |dlogits| %.3g  |dqk16| %.3g  |dctx| %.3g  argmax agree %.6f'
                  % (tag, i + 1, d, maxdiff(tr_orc[i]['qk16'], tr_ref[i]['qk16']),
                     maxdiff(tr_orc[i]['context'], tr_ref[i]['context']), agree))
            assert d < 2e-3 and agree > 0.9995, (d, agree)
        out = {'t': t, 'h': hh, 'w': ww, 'n_obj': n_obj, 'out_h': out_hw[0], 'out_w': out_hw[1], 'seed': seed,
               'wseed': wseed, 'frames_sum': checksum(frames), 'mask_sum': checksum(m0),
               'w_sum': sum(checksum(v) for k, v in sd.items() if v.dtype.is_floating_point)}
        floor = None
        if double_floor:
            ref64 = Rswem.SWEM(cfg).double()
            ref64.eval()
            ref64.load_state_dict({k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()},
                                  strict=False)
            # same random bases as the fp32 run (normal_() draws different numbers in fp64): the floor must
            # measure rounding only
            core64 = ref64.swem_core
            init64 = core64.random_init
            core64.random_init = lambda size, **k: tuple(
                t.double() for t in init64(size, **dict(k, dtype=torch.FloatTensor().type())))
            tr64 = []
            with torch.no_grad():
                torch.manual_seed(77)
                p64, _ = ref_evaluate_seq(ref64, frames.double(), [m0.double()] + [None] * (t - 1), out_hw, tr64)
            floor = [maxdiff(tr64[i]['logits'].float(), tr_ref[i]['logits']) for i in range(t - 1)]
            agree64 = [float((p64[i] == preds[i]).float().mean()) for i in range(t - 1)]
            print('   %s reference fp32-vs-fp64 noise floor |dlogits| per frame:' % tag, floor, ' argmax agree', agree64)
            out['floor64'] = np.array(floor)
            out['agree64'] = np.array(agree64)
        s = sub
        for i in range(t - 1):
            out['logits%d' % i] = tr_ref[i]['logits'][:, :, ::s, ::s]
            out['pred%d' % i] = preds[i].to(torch.uint8)
            out['ctx%d' % i] = tr_ref[i]['context'][:, ::8]
            out['qk16_%d' % i] = tr_ref[i]['qk16'] if i == 0 else tr_ref[i]['qk16'][:, ::8]
            if i == 0:
                out['qv16'] = tr_ref[0]['qv16'][:, ::8]
                out['s16'] = tr_ref[0]['s16'][:, ::8]
                out['s8'] = tr_ref[0]['s8'][:, ::16, ::2, ::2]
                out['s4'] = tr_ref[0]['s4'][:, ::16, ::4, ::4]
                if tr_ref[0]['mv16'] is not None:
                    out['mv16'] = tr_ref[0]['mv16'][:, :, ::8]
        save(tag + '.npz', **out)

    print('G8 YouTube-VOS loop: object 2 appears at frame 2 (config A sizes), + multi-scale/flip TTA')
    cfg = O.make_cfg(BACKBONE='resnet18', NUM_BASES=64, NUM_EM_ITERS=4, SINGLE_OBJ=False)
    ref = Rswem.SWEM(cfg)
    ref.eval()
    sd = weights.fill_state_dict(HipSWEM(cfg).state_dict(), seed=5, backbone='resnet18')
    ref.load_state_dict(sd, strict=False)
    frames, per_frame = synth.make_clip(t=5, h=240, w=432, n_obj=2, out_hw=(240, 432), seed=31, all_masks=True)
    ymasks = ytvos_masks(per_frame, 2)
    with torch.no_grad():
        torch.manual_seed(78)
        rp = ref_evaluate_ytvos(ref, frames, ymasks, (240, 432))
        torch.manual_seed(78)
        op = O.evaluate_ytvos_seq(O.Model(sd, cfg), frames, ymasks, (240, 432))
        ref64 = Rswem.SWEM(cfg).double()
        ref64.eval()
        ref64.load_state_dict({k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}, strict=False)
        init64 = ref64.swem_core.random_init
        ref64.swem_core.random_init = lambda size, **k: tuple(
            t_.double() for t_ in init64(size, **dict(k, dtype=torch.FloatTensor().type())))
        torch.manual_seed(78)
        rp64 = ref_evaluate_ytvos(ref64, frames.double(), [None if m is None else m.double() for m in ymasks], (240, 432))
    for a, b_ in zip(rp, op):
        assert torch.equal(a, b_)
    agree64 = [float((a == b_).float().mean()) for a, b_ in zip(rp, rp64)]
    print('   ytvos: oracle == reference (bit exact); reference fp32-vs-fp64 index agreement per frame', agree64,
          ' objects in last frame', int(rp[-1].max()))
    out = {'seed': 31, 'wseed': 5, 'agree64': np.array(agree64)}
    for i, pr in enumerate(rp):
        out['pred%d' % i] = pr.to(torch.uint8)
    # test-time augmentation on the 2-object DAVIS-style clip: the reference's own evaluate_davis_seq_ms (swem_evaluator.py:34-57),
    # every pass from the same seeded random bases
    m0 = per_frame[0]
    tfr = frames[:, :3]
    with torch.no_grad():
        tta_ref = ref_evaluate_ms(SeededInit(ref, 79), tfr, [m0, None, None], (240, 432), scales=(240, 288), is_flip=True)
        tta_orc = O.evaluate_seq_ms(SeededInit(O.Model(sd, cfg), 79), tfr, [m0, None, None], (240, 432), scales=(240, 288),
                                    is_flip=True)
    for a, b_ in zip(tta_ref, tta_orc):
        assert torch.equal(a, b_)
    for i, pr in enumerate(tta_ref):
        out['tta%d' % i] = pr.to(torch.uint8)
    save('g8_ytvos_tta.npz', **out)

    print('G6 config A (240x432, R18, K=64, single object, 2 frames)')
    run_clip(dict(BACKBONE='resnet18', NUM_BASES=64, NUM_EM_ITERS=4, SINGLE_OBJ=True), 2, 240, 432, 1, (240, 432),
             seed=123, wseed=1, tag='g6_configA', sub=2, double_floor=True)
    print('G6b config A multi-object variant (3 frames, 2 objects)')
    run_clip(dict(BACKBONE='resnet18', NUM_BASES=64, NUM_EM_ITERS=4, SINGLE_OBJ=False), 3, 240, 432, 2, (240, 427),
             seed=124, wseed=2, tag='g6_configA_mo', sub=2, double_floor=True)
    if os.environ.get('SWEM_GOLDEN_SKIP_B') != '1':
        print('G7 config B (480x864, R50, K=256, T=5, 2 objects, 4 frames) -- takes a few minutes')
        run_clip(dict(BACKBONE='resnet50', NUM_BASES=256, NUM_EM_ITERS=5, SINGLE_OBJ=False), 4, 480, 864, 2,
                 (480, 854), seed=123, wseed=3, tag='g7_configB', sub=8, double_floor=True)
    print('done')
    if OUT != HERE:
        compare_with_committed()


def compare_with_committed():
    """Every array of every regenerated fixture against the committed file of the same name, bit for bit (the .npz containers
    themselves differ in their zip timestamps)."""
    bad = 0
    for name in sorted(os.listdir(OUT)):
        if not name.endswith('.npz'):
            continue
        new, old = np.load(os.path.join(OUT, name)), np.load(os.path.join(HERE, name))
        same = set(new.files) == set(old.files) and all(
            new[k].dtype == old[k].dtype and new[k].shape == old[k].shape and np.array_equal(new[k], old[k], equal_nan=True)
            for k in new.files)
        print('  %-28s %3d arrays  %s' % (name, len(new.files), 'IDENTICAL to the committed fixture' if same else 'DIFFERS'))
        bad += not same
    if bad:
        raise SystemExit('%d regenerated fixtures differ from the committed ones' % bad)


if __name__ == '__main__':
    main()
