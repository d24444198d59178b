"""GPU: per-frame TEACHER-FORCED parity at config B (480x864, ResNet-50, K = 256, 5 EM iterations) and the long-video
case (BASELINE config E).

The free-running clips of test_gpu_model.py can only be held to the reference's own fp32-vs-fp64 floor (0.4-0.7 in the
logits: low-mass bases are ratios of rounding noise and matching does not weight bases by mass).
Here the chaotic feedback is cut: before every frame the HIP model's memory banks are overwritten with the ORACLE's
(bit-identical to the reference's, tests/golden/make_golden.py), so each frame's match -> segment is compared at the
north star's tolerance: 1e-3 on the logits, index maps >= 0.9995; every memorize is compared from identical priors."""
import pytest
import torch
import torch.nn.functional as F

from oracle import swem_oracle as O
from swem_amd import evaluator, ops
from tests import helpers as H
from tests.test_gpu_model import CFG_B, DEV, logit_bound, logits_close, relmax

pytestmark = pytest.mark.gpu
REORDER_DRAWS = 6      # fp32 evaluations of the reference's memorize with permuted key channels (the EM's noise yardstick)


def _to_dev(bases):
    return None if bases is None else {k: v.to(DEV) for k, v in bases.items()}


def _mass_err(got, ref, zita):
    z = zita.squeeze(-2).unsqueeze(-2)
    return float(((got.cpu() - ref) * z).abs().max() / (ref * z).abs().max())


def teacher_forced_clip(model, om, frames, m0, out, fixture=None, seed=77, tol=1e-3):
    """Runs the ORACLE free (it is the reference, bit for bit, where the fixtures were made) and, frame by frame, the HIP
    model from the oracle's memory: banks injected before the frame, then encode_key -> match -> segment on the HIP side
    (logits, index map) and memorize from the oracle's inputs (bases).  Returns one dict of measured errors per frame and
    asserts the north star's bars: logits within `tol` (+ the ulp slack of logit_bound), index maps >= 0.9995."""
    t, (h, w) = frames.shape[1], frames.shape[-2:]
    core = model.swem_core
    rows = []
    with torch.no_grad():
        torch.manual_seed(seed)
        mk16, _, s16, _, _ = om('encode_key', frames[:, 0])
        mfull = F.interpolate(m0, size=(h, w), mode='nearest')
        om('init', mk16, om('encode_value', frames[:, 0], mfull.float(), s16), m0)
        for i in range(1, t):
            core.memories['first'].bases = _to_dev(om.core.first.bases)
            core.memories['first'].n_objs = om.core.first.n_objs
            core.memories['update'].bases = _to_dev(om.core.upd.bases)
            oqk, oqv, os16, os8, os4 = om('encode_key', frames[:, i])
            octx, on = om('match', oqk, oqv)
            ologits, oprob = om('segment', on, octx, os8, os4, None, out)
            row = {'frame': i}
            if fixture is not None:
                # on another host CPU (other BLAS kernels / thread count) the oracle is a second fp32 evaluation of the same
                # chaotic recursion: against the fixture it is only held to the reference's own fp32-vs-fp64 floor
                d_fix = float((ologits[:, :, ::8, ::8] - fixture['logits%d' % (i - 1)]).abs().max())
                assert d_fix <= max(1e-3, 2 * float(fixture['floor64'][i - 1])), d_fix
                row['oracle_on_this_host_vs_fixture_dlogits'] = d_fix
                row['reference_fp32_vs_fp64_free_running_floor'] = float(fixture['floor64'][i - 1])
            fr = frames[:, i].to(DEV)
            qk, qv, s16h, s8, s4 = model('encode_key', fr)
            ctx, n = model('match', qk, qv)
            logits, prob = model('segment', n, ctx, s8, s4, None, out)
            pred, hard = ops.argmax_onehot(prob, want_onehot=True)
            opred = oprob.argmax(1)
            dl = float((logits.cpu() - ologits).abs().max())
            excess = float(((logits.cpu().double() - ologits.double()).abs() - logit_bound(ologits, 0.0)).max())
            agree = float((pred.cpu() == opred).float().mean())
            ctx_s, _ = model('match', oqk.to(DEV), oqv.to(DEV))          # stage-wise: the oracle's inputs
            lg_s, _ = model('segment', n, octx.to(DEV), os8.to(DEV), os4.to(DEV), None, out)
            row.update({'dlogits_max': dl, 'dlogits_beyond_ulp_slack': excess, 'index_agreement': agree,
                        'qk16_rel': relmax(qk, oqk), 'context_rel': relmax(ctx, octx),
                        'context_rel_stage': relmax(ctx_s, octx),
                        'dlogits_stage_max': float((lg_s.cpu() - ologits).abs().max())})
            if i < t - 1:
                opm = F.interpolate(oprob, size=(h, w), mode='bilinear', align_corners=False)
                ohard = (opred.unsqueeze(1) == torch.arange(on + 1).view(1, -1, 1, 1)).long()
                omv = om('encode_value', frames[:, i], opm, os16)
                mv = model('encode_value', fr, opm.to(DEV), os16.to(DEV))
                row['encode_value_rel'] = relmax(mv, omv)
                model('memorize', oqk.to(DEV), omv.to(DEV), ohard.to(DEV), opm.to(DEV))
                # yardstick for the EM (the W step's 1 - p cancels, rounding is amplified over the
                # iterations): the same memorize in float64 from the same fp32 inputs; the reference's own fp32 result
                # differs from it by `floor`, a correct fp32 implementation by about as much
                prior = om.core.first.bases if om.core.upd.bases is None else om.core.upd.bases
                mk = O.mask_prep(ohard, opm, oqk.shape[-2], oqk.shape[-1])
                b64 = O.swem(oqk.double(), omv.double(), mk.double(), {k: v.double() for k, v in prior.items()},
                             om.core.n_bases, om.core.n_iters, om.core.tau, om.core.valdim)
                om('memorize', oqk, omv, ohard, opm)
                ob, hb = om.core.upd.bases, core.memories['update'].bases
                # how far the reference's OWN fp32 arithmetic lands from float64 is one draw of an amplified rounding error
                # (the same frame of the five-object edge clip: 3.2e-4 on the GPU box's CPU, 2.0e-3 on the build container's).  A steadier yardstick: the same fp32 memorize under REORDER_DRAWS
                # mathematically neutral re-orderings of its sums (the key channels permuted in x and in the prior bases
                # alike, the result permuted back) -- the spread of the reference against itself
                draws = {'kappa': [], 'nu': [], 'zita': []}
                gperm = torch.Generator().manual_seed(11 * i)
                with torch.random.fork_rng():
                    for k in range(REORDER_DRAWS):
                        perm = torch.randperm(oqk.shape[1], generator=gperm)
                        inv = torch.argsort(perm)
                        pr = dict(prior, kappa=prior['kappa'][..., perm, :].contiguous())
                        bp = O.swem(oqk[:, perm].contiguous(), omv, mk, pr, om.core.n_bases, om.core.n_iters, om.core.tau,
                                    om.core.valdim)
                        draws['kappa'].append(_mass_err(bp['kappa'][..., inv, :].double(), b64['kappa'], b64['zita']))
                        draws['nu'].append(_mass_err(bp['nu'].double(), b64['nu'], b64['zita']))
                        draws['zita'].append(relmax(bp['zita'].double(), b64['zita']))
                for name in ('kappa', 'nu'):
                    row[name + '_mass_rel'] = _mass_err(hb[name], ob[name], ob['zita'])
                    row[name + '_mass_rel_vs_fp64'] = _mass_err(hb[name].double(), b64[name], b64['zita'])
                    row[name + '_mass_rel_reference_fp32_vs_fp64'] = _mass_err(ob[name].double(), b64[name], b64['zita'])
                    row[name + '_mass_rel_reference_fp32_reordered_vs_fp64'] = draws[name]
                row['zita_rel'] = relmax(hb['zita'], ob['zita'])
                row['zita_rel_vs_fp64'] = relmax(hb['zita'].double(), b64['zita'])
                row['zita_rel_reference_fp32_vs_fp64'] = relmax(ob['zita'].double(), b64['zita'])
                row['zita_rel_reference_fp32_reordered_vs_fp64'] = draws['zita']
            rows.append(row)
            print('teacher-forced frame %d: %s' % (i, {k: ('%.3g' % v if isinstance(v, float) else
                                                           ['%.3g' % e for e in v] if isinstance(v, list) else v)
                                                       for k, v in row.items()}))
            assert logits_close(lg_s, ologits, tol), 'stage logits frame %d: %.3g' % (i, row['dlogits_stage_max'])
            assert row['context_rel_stage'] < 1e-4
            assert logits_close(logits, ologits, tol), 'frame %d: |dlogits| %.3g (beyond the ulp slack: %.3g)' % (i, dl, excess)
            assert agree >= 0.9995, 'frame %d index agreement %.6f' % (i, agree)
            if i < t - 1:
                assert row['encode_value_rel'] < 1e-4
                for name in ('kappa_mass_rel', 'nu_mass_rel', 'zita_rel'):
                    floor = max([row[name + '_reference_fp32_vs_fp64']] + row[name + '_reference_fp32_reordered_vs_fp64'])
                    assert row[name + '_vs_fp64'] <= max(1e-4, 2 * floor), (name, row)
    return rows


def test_teacher_forced_config_b_clip(lib, golden):
    fx = golden('g7_configB.npz')
    cfg = O.make_cfg(**CFG_B)
    model, sd = H.make_model_and_sd(cfg, int(fx['wseed']), device=DEV)
    frames, m0 = H.clip_from_fixture(fx)
    rows = teacher_forced_clip(model, O.Model(sd, cfg), frames, m0, (int(fx['out_h']), int(fx['out_w'])), fixture=fx)
    H.record_parity('teacher_forced_configB_g7', rows)


@pytest.mark.parametrize('n_obj,h,w,bases,topl', [(1, 96, 160, 64, 64), (5, 112, 176, 64, 32), (3, 80, 144, 128, 64)],
                         ids=['one_object', 'five_objects_topl32', 'three_objects_k128'])
def test_teacher_forced_edge_shapes(lib, n_obj, h, w, bases, topl):
    """The edge cases of test_gpu_model.py::test_edge_shapes_free_running (one object; the reference's maximum of five
    with a top-l smaller than the bank; 1/16 grids that are no multiple of the pixel tile; an object with an EMPTY first
    mask) held to the tight bars, frame by frame, from the oracle's memory."""
    from swem_amd import synth
    cfg = O.make_cfg(BACKBONE='resnet18', NUM_BASES=bases, NUM_EM_ITERS=3, TOPL=topl)
    model, sd = H.make_model_and_sd(cfg, wseed=21 + n_obj, device=DEV)
    frames, m0 = synth.make_clip(t=4, h=h, w=w, n_obj=n_obj, seed=40 + n_obj)
    if n_obj >= 3:
        m0[:, 0] += m0[:, n_obj]
        m0[:, n_obj] = 0
    rows = teacher_forced_clip(model, O.Model(sd, cfg), frames, m0, (h, w), seed=3)
    H.record_parity('teacher_forced_edge_%dobj_k%d' % (n_obj, bases), rows)


def test_config_e_long_video(lib, golden):
    """BASELINE config E: >= 1000 frames at 480x864, the memory re-estimated on EVERY frame (sequential base merging),
    one sequence.  The state never grows (allocation constant from the captured frame on), stays finite, the labels stay
    alive, and the first three index maps are those of the 4-frame config-B clip (g7) evaluated by the plain loop."""
    import time
    from swem_amd import synth
    fx = golden('g7_configB.npz')
    cfg = O.make_cfg(**CFG_B)
    n_obj, out = int(fx['n_obj']), (int(fx['out_h']), int(fx['out_w']))
    frames, m0 = synth.make_clip(t=8, h=int(fx['h']), w=int(fx['w']), n_obj=n_obj, out_hw=out, seed=int(fx['seed']))
    short, m0s = H.clip_from_fixture(fx)
    assert torch.equal(frames[:, :short.shape[1]], short) and torch.equal(m0, m0s)
    frames, m0 = frames.to(DEV), m0.to(DEV)
    with torch.no_grad():
        model, _ = H.make_model_and_sd(cfg, int(fx['wseed']), device=DEV)
        torch.manual_seed(77)
        ref, _ = evaluator.evaluate_davis_seq(model, frames[:, :4], [m0, None, None, None], out)
        ref = [p.clone() for p in ref]
        model, _ = H.make_model_and_sd(cfg, int(fx['wseed']), device=DEV)
        torch.manual_seed(77)
        h, w = frames.shape[-2:]
        mk16, _, s16, _, _ = model('encode_key', frames[:, 0])
        model('init', mk16, model('encode_value', frames[:, 0], ops.resize_planes(m0, (h, w), 'nearest'), s16), m0)
        preds = [evaluator.frame_step(model, frames[:, i], out).clone() for i in (1, 2, 3)]
        for a, b in zip(preds, ref):
            assert torch.equal(a, b)
        g = evaluator.FrameGraph(model, frames[:, 1].shape, out).capture(frames[:, 4])
        torch.cuda.synchronize()
        mem0 = torch.cuda.memory_allocated()
        total = 1000
        t0 = time.perf_counter()
        for k in range(4, total):
            pred = g.run(frames[:, 1 + k % 7])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        mem1 = torch.cuda.memory_allocated()
    mem = model.swem_core.memories
    assert mem1 == mem0, 'allocation grew over the sequence: %d -> %d bytes' % (mem0, mem1)
    assert mem['first'].bases['kappa'].shape == mem['update'].bases['kappa'].shape == (1, n_obj, 2, 128, 256)
    assert all(torch.isfinite(v).all() for v in mem['update'].bases.values())
    assert float(mem['update'].bases['zita'].sum()) > 0
    labels = sorted(int(v) for v in torch.unique(pred).tolist())
    assert labels == list(range(n_obj + 1)), labels
    H.record_parity('config_e_long_video', {'frames': total, 'frames_per_s_graph_replay_default_plans': (total - 4) / dt,
                                             'allocated_bytes': mem1, 'labels_last_frame': labels,
                                             'first_three_index_maps_equal_plain_loop': True})
