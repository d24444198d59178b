"""CPU: the oracle (oracle/swem_oracle.py) against the golden vectors generated from the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle; the GPU tests then pin the HIP path to the oracle."""
import os

import pytest
import torch

from oracle import swem_oracle as O
from tests import helpers as H


def test_single_steps_bit_exact(golden):
    g = golden('g1_steps.npz')
    x = g['x']
    xf = x.flatten(2)[:, None, None]
    x_t = xf.transpose(-2, -1)
    assert torch.equal(O.e_step(x_t, g['kappa'], g['masks'], g['tau']), g['z'])
    k, z = O.m_step(g['z'], xf, g['kappa_prev'], g['zita_prev'])
    assert torch.equal(k, g['kappa_m']) and torch.equal(z, g['zita_m'])
    assert torch.equal(O.w_step(g['kappa'], x_t, g['masks'], g['tau']), g['weights'])
    nu = (g['zita_prev'] * g['nu_prev'] + torch.matmul(g['v'].flatten(3).unsqueeze(2), g['z'])) / g['zita_m']
    assert torch.equal(nu, g['nu'])


def test_two_frame_memorize_and_matching_bit_exact(golden):
    g, m = golden('g2_memorize.npz'), golden('g3_matching.npz')
    core = O.Core(64, 128, 4, 0.05, 64)
    init = {'kappa': g['init_kappa'], 'nu': g['init_nu'], 'zita': g['init_zita']}
    b0 = O.swem(g['x0'], g['v0'], g['m0'], init, 64, 4, 0.05, 128)
    for k in ('kappa', 'nu', 'zita'):
        assert torch.equal(b0[k], g[k + '0']), k
    core.first.update(b0)
    qv = torch.zeros(1, 128, *m['qk'].shape[-2:])
    mem, _, S, n = core.match_features(m['qk'], qv)
    assert n == 2 and torch.equal(S, m['S1']) and torch.equal(mem, m['mem1'].flatten(0, 1))
    b1 = core.memorize(g['x1'], g['v1'], g['m1'])
    for k in ('kappa', 'nu', 'zita'):
        assert torch.equal(b1[k], g[k + '1']), k
    mem, _, S, _ = core.match_features(m['qk'], qv)
    assert torch.equal(S, m['S2']) and torch.equal(mem, m['mem2'].flatten(0, 1))


def _run_clip(fx, cfg):
    frames, m0 = H.clip_from_fixture(fx)
    assert H.checksum(frames) == pytest.approx(fx['frames_sum'], rel=1e-12)
    assert H.checksum(m0) == pytest.approx(fx['mask_sum'], rel=1e-12)
    _, sd = H.make_model_and_sd(cfg, int(fx['wseed']))
    wsum = sum(H.checksum(v) for v in sd.values() if v.dtype.is_floating_point)
    assert wsum == pytest.approx(fx['w_sum'], rel=1e-12)
    t = frames.shape[1]
    trace = []
    with torch.no_grad():
        torch.manual_seed(77)
        preds, _ = O.evaluate_seq(O.Model(sd, cfg), frames, [m0] + [None] * (t - 1),
                                  (int(fx['out_h']), int(fx['out_w'])), trace)
    return preds, trace


@pytest.mark.parametrize('name,kw,sub', [
    ('g6_configA.npz', dict(BACKBONE='resnet18', NUM_BASES=64, NUM_EM_ITERS=4, SINGLE_OBJ=True), 2),
    ('g6_configA_mo.npz', dict(BACKBONE='resnet18', NUM_BASES=64, NUM_EM_ITERS=4, SINGLE_OBJ=False), 2),
])
def test_full_clip_config_a(golden, name, kw, sub):
    fx = golden(name)
    preds, trace = _run_clip(fx, O.make_cfg(**kw))
    for i, tr in enumerate(trace):
        assert torch.equal(tr['logits'][:, :, ::sub, ::sub], fx['logits%d' % i])
        assert torch.equal(preds[i].to(torch.uint8), fx['pred%d' % i])
        assert torch.equal(tr['context'][:, ::8], fx['ctx%d' % i])
    assert torch.equal(trace[0]['qk16'], fx['qk16_0'])


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), 'golden', 'g7_configB.npz')),
                    reason='config-B fixture not generated')
def test_full_clip_config_b(golden):
    fx = golden('g7_configB.npz')
    cfg = O.make_cfg(BACKBONE='resnet50', NUM_BASES=256, NUM_EM_ITERS=5, SINGLE_OBJ=False)
    preds, trace = _run_clip(fx, cfg)
    for i, tr in enumerate(trace):
        assert torch.equal(tr['logits'][:, :, ::8, ::8], fx['logits%d' % i])
        assert torch.equal(preds[i].to(torch.uint8), fx['pred%d' % i])


def test_ytvos_loop_and_tta(golden):
    """swem_evaluator.py:104-148 (objects appearing later -> MemoryBank.add_new / random_init for new ids) and
    :34-57 (multi-scale + flip averaging) against the reference's index maps."""
    from swem_amd import synth
    fx = golden('g8_ytvos_tta.npz')
    cfg = O.make_cfg(BACKBONE='resnet18', NUM_BASES=64, NUM_EM_ITERS=4, SINGLE_OBJ=False)
    _, sd = H.make_model_and_sd(cfg, int(fx['wseed']))
    frames, per_frame = synth.make_clip(t=5, h=240, w=432, n_obj=2, out_hw=(240, 432), seed=int(fx['seed']), all_masks=True)
    with torch.no_grad():
        torch.manual_seed(78)
        preds = O.evaluate_ytvos_seq(O.Model(sd, cfg), frames, H.ytvos_masks(per_frame, 2), (240, 432))
        tta = O.evaluate_seq_ms(H.SeededInit(O.Model(sd, cfg), 79), frames[:, :3], [per_frame[0], None, None],
                                (240, 432), scales=(240, 288), is_flip=True)
    for i, p in enumerate(preds):
        assert torch.equal(p.to(torch.uint8), fx['pred%d' % i])
    assert int(preds[-1].max()) == 2 and int(preds[0].max()) == 1      # the second object exists only after frame 2
    for i, p in enumerate(tta):
        assert torch.equal(p.to(torch.uint8), fx['tta%d' % i])


@pytest.mark.parametrize('tag,it', [('r18', 5), ('r18', 45), ('r50', 45), ('r50k256', 45), ('r50k256n5', 45)])
def test_train_step_matches_reference_trainer(golden, tag, it):
    """a18 / f3: the oracle's training step (forward with gradients, BootstrappedCE + IoU loss, autograd) against the
    losses, index maps and per-parameter gradient norms recorded from the reference's SWEMTrainer.one_step."""
    tc = H.train_cases()
    case = tc['cases'][tag]
    fx = golden('g9_train_%s_it%d.npz' % (tag, it))
    cfg = O.make_cfg(**case['cfg'])
    model, sd = H.make_model_and_sd(cfg, case['wseed'], pred_scale=tc['pred_scale'])
    frames, init_mask, label, valid = H.train_batch(case)
    assert H.checksum(frames) == pytest.approx(float(fx['frames_sum']), rel=1e-12)
    torch.manual_seed(91)
    losses, results, grads, logits = O.train_one_step(H.trainable_sd(sd, model), cfg, frames, init_mask, valid, label,
                                                      it, tc['loss_cfg'])
    assert float(losses['total_loss']) == pytest.approx(float(fx['total_loss']), abs=1e-6)
    assert float(losses['main_loss']) == pytest.approx(float(fx['main_loss']), abs=1e-6)
    assert float(losses['aux_loss']) == pytest.approx(float(fx['aux_loss']), abs=1e-6)
    assert float(losses['p']) == pytest.approx(float(fx['p']), abs=1e-12)
    assert torch.equal(results.to(torch.uint8), fx['results'])
    names = fx['grad_names']
    assert names == sorted(k for k, g in grads.items() if g is not None)
    for n, ref_norm in zip(names, fx['grad_norms'].tolist()):
        assert float(grads[n].double().norm()) == pytest.approx(ref_norm, rel=1e-5, abs=1e-12), n
    assert torch.allclose(grads['decoder.pred.weight'], fx['g_pred_weight'], rtol=1e-5, atol=1e-9)
    # AdamW (solver/solver.py:38-41) on these gradients reproduces the reference optimizer's parameters
    sc = tc['solver_cfg']
    p = {'decoder.pred.weight': sd['decoder.pred.weight'].clone()}
    O.adamw_step(p, {'decoder.pred.weight': grads['decoder.pred.weight']}, {},
                 O.multistep_lr(sc['BASE_LR'], sc['PRETRAIN_ITERS'], sc['GAMMA'], 0), sc['WEIGHT_DECAY'], 1)
    assert torch.allclose(p['decoder.pred.weight'], fx['w_after_pred_weight'], rtol=0, atol=1e-8)


def _module_sd(fx, tag, prefix):
    return {prefix + k[len(tag) + 1:]: v for k, v in fx.items()
            if k.startswith(tag + '_') and torch.is_tensor(v) and ('weight' in k or 'bias' in k)}


def test_module_level_vectors_bit_exact(golden):
    """G4 / G5: the oracle against outputs of the reference's own ResBlock, UpsampleBlock, FeatureFusionBlock(+CBAM),
    FeatureFusionLayer, Decoder + decode/aggregate (tests/golden/make_golden_modules.py)."""
    from swem_amd import weights
    from swem_amd.swem import SWEM
    fx = golden('g45_modules.npz')
    for tag in ('rb_same', 'rb_down'):
        assert torch.equal(O.res_block(_module_sd(fx, tag, ''), 'm', fx[tag + '_x']), fx[tag + '_y'])
    sd = _module_sd(fx, 'up', '')
    sk = O.conv(sd, 'm.skip_conv', fx['up_skip'])
    y = O.res_block(sd, 'm.out_conv', sk + torch.nn.functional.interpolate(fx['up_low'], size=sk.shape[-2:],
                                                                            mode='bilinear', align_corners=False))
    assert torch.equal(y, fx['up_y'])
    sd = _module_sd(fx, 'ffb', '')
    x = O.res_block(sd, 'm.block1', torch.cat([fx['ffb_x'], fx['ffb_f16']], 1))
    assert torch.equal(O.res_block(sd, 'm.block2', x + O.cbam(sd, 'm.attention', x)), fx['ffb_y'])
    sd = {'swem_core.fusion_layer.' + k[4:]: v for k, v in fx.items() if k.startswith('ffl_layer')}
    assert torch.equal(O.fusion_layer(sd, fx['ffl_x']), fx['ffl_y'])
    cfg = O.make_cfg(BACKBONE='resnet18', NUM_BASES=64)
    full = weights.fill_state_dict(SWEM(cfg).state_dict(), seed=int(fx['dec_wseed']), backbone='resnet18')
    full['decoder.pred.weight'] = full['decoder.pred.weight'] * float(fx['dec_pred_scale'])
    lg, pr = O.decode(full, 2, fx['dec_ctx'], fx['dec_s8'], fx['dec_s4'], fx['dec_valid'], (61, 90))
    assert torch.equal(lg, fx['dec_logits']) and torch.equal(pr, fx['dec_prob'])
    lg2, _ = O.decode(full, 2, fx['dec_ctx'], fx['dec_s8'], fx['dec_s4'], None, (64, 96))
    assert torch.equal(lg2, fx['dec_logits_novalid'])
    assert torch.equal(O.aggregate(fx['agg_in']), fx['agg_out'])
