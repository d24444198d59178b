"""GPU: per-frame TEACHER-FORCED parity at config B (480x864, ResNet-50, K = 256, 5 EM iterations) and the long-video
case (BASELINE config E).

The free-running clips of test_gpu_model.py can only be held to the reference's own fp32-vs-fp64 floor (0.4-0.7 in the
logits: low-mass bases are ratios of rounding noise and matching does not weight bases by mass).
Here the chaotic feedback is cut: before every frame the HIP model's memory banks are overwritten with the ORACLE's
(bit-identical to the reference's, tests/golden/make_golden.py), so each frame's match -> segment is compared at the
north star's tolerance: 1e-3 on the logits outside saturation, index maps >= 0.9995 -- and, at config B, at regression bars
set just above what is measured (TIGHT_B); every memorize is compared from identical priors."""
import pytest
import torch
import torch.nn.functional as F

from oracle import swem_oracle as O
from swem_amd import evaluator, ops
from tests import helpers as H
from tests.test_gpu_model import CFG_B, DEV, logit_bound, logits_close, probs_close, relmax

pytestmark = pytest.mark.gpu
REORDER_DRAWS = 6      # fp32 evaluations of the reference's memorize with permuted key channels (the EM's noise yardstick)


def _to_dev(bases):
    return None if bases is None else {k: v.to(DEV) for k, v in bases.items()}


def _mass_err(got, ref, zita):
    z = zita.squeeze(-2).unsqueeze(-2)
    return float(((got.cpu() - ref) * z).abs().max() / (ref * z).abs().max())


_TRAJ = {}


def oracle_trajectory(key, om, frames, m0, out, fixture=None, seed=77):
    """The ORACLE side of a teacher-forced clip, computed once per clip and shared by the arithmetic modes the test is
    parametrised over: the oracle runs free (it is the reference, bit for bit, where the fixtures were made); per frame the
    memory it matched against, its encoder / match / segment outputs, and for every memorize the inputs, its own result, the
    float64 result from the same fp32 inputs and REORDER_DRAWS re-ordered fp32 evaluations (the EM's noise yardstick)."""
    if key in _TRAJ:
        return _TRAJ[key]
    t, (h, w) = frames.shape[1], frames.shape[-2:]
    steps = []
    with torch.no_grad():
        torch.manual_seed(seed)
        mk16, _, s16, _, _ = om('encode_key', frames[:, 0])
        mfull = F.interpolate(m0, size=(h, w), mode='nearest')
        om('init', mk16, om('encode_value', frames[:, 0], mfull.float(), s16), m0)
        for i in range(1, t):
            st = {'frame': i, 'first': {k: v.clone() for k, v in om.core.first.bases.items()}, 'first_n': om.core.first.n_objs,
                  'upd': None if om.core.upd.bases is None else {k: v.clone() for k, v in om.core.upd.bases.items()}}
            oqk, oqv, os16, os8, os4 = om('encode_key', frames[:, i])
            octx, on = om('match', oqk, oqv)
            ologits, oprob = om('segment', on, octx, os8, os4, None, out)
            st.update(oqk=oqk, oqv=oqv, os16=os16, os8=os8, os4=os4, octx=octx, on=on, ologits=ologits, oprob=oprob,
                      opred=oprob.argmax(1))
            if fixture is not None:
                # on another host CPU (other BLAS kernels / thread count) the oracle is a second fp32 evaluation of the same
                # chaotic recursion: against the fixture it is only held to the reference's own fp32-vs-fp64 floor
                d_fix = float((ologits[:, :, ::8, ::8] - fixture['logits%d' % (i - 1)]).abs().max())
                assert d_fix <= max(1e-3, 2 * float(fixture['floor64'][i - 1])), d_fix
                st['oracle_on_this_host_vs_fixture_dlogits'] = d_fix
                st['reference_fp32_vs_fp64_free_running_floor'] = float(fixture['floor64'][i - 1])
            if i < t - 1:
                opm = F.interpolate(oprob, size=(h, w), mode='bilinear', align_corners=False)
                ohard = (st['opred'].unsqueeze(1) == torch.arange(on + 1).view(1, -1, 1, 1)).long()
                omv = om('encode_value', frames[:, i], opm, os16)
                # yardstick for the EM (the W step's 1 - p cancels, rounding is amplified over the
                # iterations): the same memorize in float64 from the same fp32 inputs; the reference's own fp32 result
                # differs from it by `floor`, a correct fp32 implementation by about as much
                prior = om.core.first.bases if om.core.upd.bases is None else om.core.upd.bases
                mk = O.mask_prep(ohard, opm, oqk.shape[-2], oqk.shape[-1])
                b64 = O.swem(oqk.double(), omv.double(), mk.double(), {k: v.double() for k, v in prior.items()},
                             om.core.n_bases, om.core.n_iters, om.core.tau, om.core.valdim)
                # how far the reference's OWN fp32 arithmetic lands from float64 is one draw of an amplified rounding error
                # (the same frame of the five-object edge clip: 3.2e-4 on the GPU box's CPU, 2.0e-3 on the build container's).
                # A steadier yardstick: the same fp32 memorize under REORDER_DRAWS mathematically neutral re-orderings of
                # its sums (the key channels permuted in x and in the prior bases alike, the result permuted back) -- the
                # spread of the reference against itself
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
                om('memorize', oqk, omv, ohard, opm)
                st.update(opm=opm, ohard=ohard, omv=omv, b64=b64, draws=draws,
                          ob={k: v.clone() for k, v in om.core.upd.bases.items()})
            steps.append(st)
    _TRAJ[key] = steps
    return steps


# Regression bars at config B: the north star's bars below (1e-3 outside saturation, 0.9995) leave two
# orders of magnitude between what is measured and what would fail.  Measured in rounds 3-4 (profiles/r04_parity.json), per
# frame of the teacher-forced config-B clips: logit excess over the ulp term 5.5e-5..9.3e-5 (bf16x3: 5.3e-4..5.8e-4), 0-2
# differing pixels of 409,920 (bf16x3: 4), context 4.3e-6..6.3e-6 of its range (bf16x3: 2.8e-5).  These are asserted IN
# ADDITION to the outer bars: (logit excess, differing pixels per index map, context_rel).
TIGHT_B = {'fp32': (2e-4, 4, 2e-5), 'f16x3': (2e-4, 4, 2e-5), 'tuned': (2e-4, 4, 2e-5), 'bf16x3': (1e-3, 8, 6e-5)}


def index_ties(pred, opred, oprob, dprob, what):
    """"Bit-exact index maps" with ties told apart from bugs: every pixel whose index differs from the
    oracle's must be an argmax NEAR-TIE of the oracle itself -- the gap between the oracle's two largest probabilities at that
    pixel no larger than twice the probability difference this very frame measured between the two implementations (`dprob`).
    A differing pixel anywhere else is a wrong index, whatever the count.  Returns the measured gaps for the parity record."""
    diff = pred != opred
    if not bool(diff.any()):
        return {'index_diff_oracle_top2_gap_max': 0.0, 'index_diffs_all_near_ties': True}
    top2 = oprob.topk(2, dim=1).values
    gaps = (top2[:, 0] - top2[:, 1])[diff]
    worst = float(gaps.max())
    assert worst <= 2.0 * dprob, ('%s: %d index-map pixels differ and the oracle\'s top-2 probability gap there reaches %.3g > 2 x '
                                  '|dprob| = %.3g: not a tie' % (what, int(diff.sum()), worst, 2.0 * dprob))
    return {'index_diff_oracle_top2_gap_max': worst, 'index_diffs_all_near_ties': True}


def teacher_forced_clip(model, steps, frames, out, tol=1e-3, tight=None):
    """Frame by frame, the HIP model from the oracle's memory: banks injected before the frame, then encode_key -> match ->
    segment on the HIP side (logits, probabilities, index map) and memorize from the oracle's inputs (bases).  Returns one
    dict of measured errors per frame and asserts the north star's bars: logits within `tol` OUTSIDE SATURATION (+ the ulp
    slack of logit_bound where log(p / (1 - p)) saturates: 109-140 of 1,229,760 logits per frame exceed a flat 1e-3, every one
    at |reference logit| > 7), probabilities within half that bound (probs_close), index maps >= 0.9995 -- and, with
    `tight` = (logit excess over the ulp term, differing pixels per index map, context_rel), the regression bars above."""
    t = frames.shape[1]
    core = model.swem_core
    rows = []
    with torch.no_grad():
        for st in steps:
            i = st['frame']
            core.memories['first'].bases = _to_dev(st['first'])
            core.memories['first'].n_objs = st['first_n']
            core.memories['update'].bases = _to_dev(st['upd'])
            oqk, oqv, os16, os8, os4, octx, on = (st[k] for k in ('oqk', 'oqv', 'os16', 'os8', 'os4', 'octx', 'on'))
            ologits, oprob, opred = st['ologits'], st['oprob'], st['opred']
            row = {'frame': i}
            for k in ('oracle_on_this_host_vs_fixture_dlogits', 'reference_fp32_vs_fp64_free_running_floor'):
                if k in st:
                    row[k] = st[k]
            fr = frames[:, i].to(DEV)
            qk, qv, s16h, s8, s4 = model('encode_key', fr)
            ctx, n = model('match', qk, qv)
            logits, prob = model('segment', n, ctx, s8, s4, None, out)
            pred, hard = ops.argmax_onehot(prob, want_onehot=True)
            dl = float((logits.cpu() - ologits).abs().max())
            dabs = (logits.cpu().double() - ologits.double()).abs()
            excess = float((dabs - logit_bound(ologits, 0.0)).max())
            # how many logits needed the saturation slack at all (a flat 1e-3 would have failed them), and how many of those sit
            # where the reference's own logit is saturated (|logit| > 7: one ulp of its fp32 probability moves it by > 1e-3)
            over = dabs > 1e-3
            n_over, n_over_sat = int(over.sum()), int((over & (ologits.double().abs() > 7.0)).sum())
            agree = float((pred.cpu() == opred).float().mean())
            ctx_s, _ = model('match', oqk.to(DEV), oqv.to(DEV))          # stage-wise: the oracle's inputs
            lg_s, _ = model('segment', n, octx.to(DEV), os8.to(DEV), os4.to(DEV), None, out)
            row.update({'logits_total': int(dabs.numel()), 'logits_beyond_flat_1e-3': n_over,
                        'logits_beyond_flat_1e-3_where_reference_logit_saturated': n_over_sat})
            row.update({'dlogits_max': dl, 'dlogits_beyond_ulp_slack': excess, 'index_agreement': agree,
                        'dprob_max': float((prob.cpu() - oprob).abs().max()),
                        # softmax is 1/2-Lipschitz in the max-norm of the logits: excess over half the channel-wise logit bound
                        # (where two objects' probabilities both sit at the 1 - 1e-7 clamp the reference's own logits are
                        # ulp noise of +-0.05, and the softmax over them moves by +-0.01: swem.py:111-116)
                        'dprob_beyond_bound': float(((prob.cpu().double() - oprob.double()).abs()
                                                     - 0.5 * logit_bound(ologits, tol).max(dim=1, keepdim=True)[0]).max()),
                        'qk16_rel': relmax(qk, oqk), 'context_rel': relmax(ctx, octx),
                        'context_rel_stage': relmax(ctx_s, octx),
                        'dlogits_stage_max': float((lg_s.cpu() - ologits).abs().max())})
            if i < t - 1:
                opm, ohard, omv, b64, draws, ob = (st[k] for k in ('opm', 'ohard', 'omv', 'b64', 'draws', 'ob'))
                mv = model('encode_value', fr, opm.to(DEV), os16.to(DEV))
                row['encode_value_rel'] = relmax(mv, omv)
                model('memorize', oqk.to(DEV), omv.to(DEV), ohard.to(DEV), opm.to(DEV))
                hb = core.memories['update'].bases
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
            assert probs_close(prob, oprob, ologits, tol), 'frame %d: |dprob| %.3g (beyond its bound: %.3g)' % (
                i, row['dprob_max'], row['dprob_beyond_bound'])
            assert agree >= 0.9995, 'frame %d index agreement %.6f' % (i, agree)
            row['index_pixels_differing'] = int((pred.cpu() != opred).sum())
            row.update(index_ties(pred.cpu(), opred, oprob, row['dprob_max'], 'frame %d' % i))
            if tight is not None:
                t_excess, t_pix, t_ctx = tight
                assert excess <= t_excess, 'frame %d: logit excess over the ulp term %.3g > %.1g (regression bar)' % (i, excess, t_excess)
                assert row['index_pixels_differing'] <= t_pix, 'frame %d: %d pixels of the index map differ (regression bar %d)' % (
                    i, row['index_pixels_differing'], t_pix)
                assert row['context_rel'] <= t_ctx, 'frame %d: context_rel %.3g > %.1g (regression bar)' % (i, row['context_rel'], t_ctx)
            if i < t - 1:
                assert row['encode_value_rel'] < 1e-4
                for name in ('kappa_mass_rel', 'nu_mass_rel', 'zita_rel'):
                    floor = max([row[name + '_reference_fp32_vs_fp64']] + row[name + '_reference_fp32_reordered_vs_fp64'])
                    assert row[name + '_vs_fp64'] <= max(1e-4, 2 * floor), (name, row)
    return rows


@pytest.mark.parametrize('mode', H.ARITH_MODES)
def test_teacher_forced_config_b_clip(lib, golden, mode):
    """BASELINE configs[1] in each conv arithmetic the product can run: exact fp32 MFMA, bf16x3 FORCED on every layer, and
    the plans the bench loads.  The bars (1e-3 logits / probabilities, 0.9995 index maps) are the same in all three."""
    fx = golden('g7_configB.npz')
    cfg = O.make_cfg(**CFG_B)
    model, sd = H.make_model_and_sd(cfg, int(fx['wseed']), device=DEV)
    frames, m0 = H.clip_from_fixture(fx)
    out = (int(fx['out_h']), int(fx['out_w']))
    steps = oracle_trajectory('g7', O.Model(sd, cfg), frames, m0, out, fixture=fx)
    with H.arith(mode, model) as ar:
        rows = teacher_forced_clip(model, steps, frames, out, tight=TIGHT_B[mode])
    H.record_parity('teacher_forced_configB_g7[%s]' % mode, {'conv_launches_by_math': ar.summary(),
                                                              'plans_digest': model.book.digest(), 'frames': rows})


@pytest.mark.parametrize('n_obj', (1, 3), ids=['one_object', 'three_objects'])
def test_teacher_forced_config_b_other_object_counts(lib, n_obj):
    """BASELINE configs[1] at the other object counts DAVIS17-val carries (swem_evaluator.py:59-102 runs whatever init_masks
    holds; SURVEY 8d: N in {1, 2, 3}), in the arithmetic the product ships for them ('tuned': the plan file holds these shapes
    too): the same bars as the two-object clip, against the oracle run here (a 3-frame clip: the oracle is the CPU reference)."""
    from swem_amd import synth
    cfg = O.make_cfg(**CFG_B)
    model, sd = H.make_model_and_sd(cfg, 3, device=DEV)
    out = (480, 854)
    frames, m0 = synth.make_clip(t=3, h=480, w=864, n_obj=n_obj, out_hw=out, seed=60 + n_obj)
    steps = oracle_trajectory(('cfgB', n_obj), O.Model(sd, cfg), frames, m0, out, seed=9)
    with H.arith('tuned', model) as ar:
        rows = teacher_forced_clip(model, steps, frames, out, tight=TIGHT_B['tuned'])
    # the shipped file must really hold this object count's layer shapes: nothing fell to the untuned fallback but the stems
    assert ar.ran.get(7, 0) >= 0.95 * sum(ar.ran.values()), ar.ran
    H.record_parity('teacher_forced_configB_%dobj[tuned]' % n_obj, {'conv_launches_by_math': ar.summary(),
                                                                    'plans_digest': model.book.digest(), 'frames': rows})


@pytest.mark.parametrize('n_obj,h,w,bases,topl', [(1, 96, 160, 64, 64), (5, 112, 176, 64, 32), (3, 80, 144, 128, 64)],
                         ids=['one_object', 'five_objects_topl32', 'three_objects_k128'])
@pytest.mark.parametrize('mode', ('fp32', 'f16x3', 'bf16x3'))
def test_teacher_forced_edge_shapes(lib, n_obj, h, w, bases, topl, mode):
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
    steps = oracle_trajectory(('edge', n_obj, bases), O.Model(sd, cfg), frames, m0, (h, w), seed=3)
    with H.arith(mode, model) as ar:
        rows = teacher_forced_clip(model, steps, frames, (h, w))
    H.record_parity('teacher_forced_edge_%dobj_k%d[%s]' % (n_obj, bases, mode), {'conv_launches_by_math': ar.summary(),
                                                                                 'frames': rows})


@pytest.mark.parametrize('mode', ('tuned', 'fp32'))
def test_lockstep_lanes_against_the_reference_fixture_g7(lib, golden, mode):
    """The lock-step lanes (evaluator.LockstepPool, round 6: bench.py's default launch form) at config-B size against the
    REFERENCE's own index maps (fixture g7, recorded from lmm077/SWEM itself): four copies of the 4-frame clip run as two lanes of two
    sequences with one frame per replay, so frame 2 of every sequence comes out of the captured lock-step graph -- the key encoder
    over both sequences' frames in one pass, decoder and value encoder batched over their four objects -- with the shipped plans
    ('tuned') and on the fp32 kernels.  Every sequence meets the free-running bar of the plain loop on every frame, agrees with the
    plain loop of a replica to the same bar, and on the fp32 kernels equals it on the frame before the graph (frame 1)."""
    fx = golden('g7_configB.npz')
    cfg = O.make_cfg(**CFG_B)
    out = (int(fx['out_h']), int(fx['out_w']))
    frames, m0 = H.clip_from_fixture(fx)
    frames, m0 = frames.to(DEV), m0.to(DEV)
    models = [H.make_model_and_sd(cfg, int(fx['wseed']), device=DEV)[0] for _ in range(5)]
    for m in models[1:4]:
        m.book = models[0].book
    with torch.no_grad(), H.arith(mode, models[0], models[4]) as ar:
        torch.manual_seed(77)
        plain, _ = evaluator.evaluate_davis_seq(models[4], frames, [m0, None, None, None], out)
        plain = [p.clone() for p in plain]
        pool = evaluator.LockstepPool(models[:4], lockstep=2, lookahead=1, plans=None)
        got = pool.run([(frames, m0, out)] * 4, seeds=[77] * 4)
        torch.cuda.synchronize()
        assert all(isinstance(g_, evaluator.LockstepGraph) for g_ in pool.graphs)
    agree = []
    for preds in got:
        assert len(preds) == 3
        if mode == 'fp32':
            assert torch.equal(preds[0], plain[0])                 # frame 1: the same eager launches
        # ('tuned': which producers already write their consumers' fp16 planes -- and then leave the fp32 map out, so that a residual
        # is read from the pair -- depends on what the book's hints have seen; four interleaved sequences and one plain loop differ there)
        a_ = [float((p_.cpu().to(torch.uint8) == fx['pred%d' % i_]).float().mean()) for i_, p_ in enumerate(preds)]
        for i_, v_ in enumerate(a_):
            assert v_ >= min(0.9995, float(fx['agree64'][i_]) - 0.01), (i_, v_)
        agree.append(a_)
    # (against the plain loop of a replica the same free-running bar: the clip's EM amplifies a last-bit difference of frame 1 to
    # the fp32-vs-float64 level of the fixture by frames 2-3)
    vs_plain = [float((a == b).float().mean()) for preds in got for a, b in zip(preds, plain)]
    for j_, v_ in enumerate(vs_plain):
        assert v_ >= min(0.9995, float(fx['agree64'][j_ % 3]) - 0.01), (j_, v_)
    H.record_parity('lockstep_lanes_g7[%s]' % mode, {'conv_launches_by_math': ar.summary(), 'plans_digest': models[0].book.digest(),
                                                    'index_maps_agreement_with_reference_fixture_g7': agree,
                                                    'index_maps_agreement_with_the_plain_loop_min': min(vs_plain),
                                                    'note': 'two lanes of two sequences, one frame per replay: frame 2 of every sequence from '
                                                            'the lock-step graph'})


@pytest.mark.parametrize('mode', H.ARITH_MODES)
def test_config_e_long_video(lib, golden, mode):
    """BASELINE config E: >= 1000 frames at 480x864, the memory re-estimated on EVERY frame (sequential base merging),
    one sequence.  The state never grows (allocation constant from the captured frame on), stays finite, the labels stay
    alive, and the first three index maps are those of the 4-frame config-B clip (g7) evaluated by the plain loop -- in each
    conv arithmetic (helpers.arith)."""
    import time
    from swem_amd import synth
    fx = golden('g7_configB.npz')
    cfg = O.make_cfg(**CFG_B)
    n_obj, out = int(fx['n_obj']), (int(fx['out_h']), int(fx['out_w']))
    frames, m0 = synth.make_clip(t=8, h=int(fx['h']), w=int(fx['w']), n_obj=n_obj, out_hw=out, seed=int(fx['seed']))
    short, m0s = H.clip_from_fixture(fx)
    assert torch.equal(frames[:, :short.shape[1]], short) and torch.equal(m0, m0s)
    frames, m0 = frames.to(DEV), m0.to(DEV)
    model0, _ = H.make_model_and_sd(cfg, int(fx['wseed']), device=DEV)
    model, _ = H.make_model_and_sd(cfg, int(fx['wseed']), device=DEV)
    with torch.no_grad(), H.arith(mode, model0, model) as ar:
        torch.manual_seed(77)
        ref, _ = evaluator.evaluate_davis_seq(model0, frames[:, :4], [m0, None, None, None], out)
        ref = [p.clone() for p in ref]
        # ... and those are the REFERENCE's index maps of that clip (fixture g7, recorded from lmm077/SWEM itself), to the
        # free-running bar: the long run starts from a state that is parity-checked against the reference, not only against
        # another HIP path
        agree_ref = [float((p_.cpu().to(torch.uint8) == fx['pred%d' % i_]).float().mean()) for i_, p_ in enumerate(ref)]
        for i_, a_ in enumerate(agree_ref):
            assert a_ >= min(0.9995, float(fx['agree64'][i_]) - 0.01), (i_, a_)
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
    H.record_parity('config_e_long_video[%s]' % mode,
                    {'frames': total, 'frames_per_s_graph_replay': (total - 4) / dt, 'conv_launches_by_math': ar.summary(),
                     'plans_digest': model.book.digest(), 'allocated_bytes': mem1, 'labels_last_frame': labels,
                     'first_three_index_maps_equal_plain_loop': True,
                     'first_three_index_maps_agreement_with_reference_fixture_g7': agree_ref,
                     'note': 'plain (not software-pipelined) frame graph, one sequence, first replays included'})
