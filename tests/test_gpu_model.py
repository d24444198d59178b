"""GPU: the SWEM model (drop-in module API) stage by stage and end to end against the CPU oracle and the golden
clips generated from the reference.

Stage tests feed the HIP stage the ORACLE's inputs, so errors do not compound (tolerance 1e-4 relative to the
tensor's max, fp32 everywhere).  End-to-end clips are free running: the reference's own fp32-vs-fp64 noise floor
on these clips is 0.4-0.7 in the logits (stored in the fixtures, SURVEY.md section 7.2: low-mass bases are
rounding noise, and already frame 1 reads a memory built by several EM iterations), so the free-running bar is
max(1e-3, 2 x floor)  on the logits and an index-map agreement at least as good as the reference's own fp32-vs-fp64
agreement.  The 1e-3 / 0.9995 bars of the north star are asserted per frame with the memory TEACHER-FORCED from the oracle
(tests/test_gpu_parity.py); the numbers measured here are written to the round's parity report (helpers.record_parity)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import swem_oracle as O
from swem_amd import evaluator, ops
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def relmax(a, b):
    return float((a.float().cpu() - b).abs().max() / b.abs().max())


def logit_bound(ref, tol=1e-3):
    """tol + 4 ulp of the probability pushed through d(logit)/dp = 1/(p(1-p)).
    swem.py:111-115 computes logit = log(p/(1-p)) from an fp32 sigmoid: where p is within 1e-5 of 0 or 1 the
    reference's own logit moves by 0.01-0.3 per ulp of p, so a flat 1e-3 only applies where |logit| <~ 7."""
    p = torch.sigmoid(ref.cpu().double())
    return tol + 2.5e-7 / (p * (1 - p))


def logits_close(got, ref, tol=1e-3):
    return bool(((got.cpu().double() - ref.cpu().double()).abs() <= logit_bound(ref, tol)).all())


def probs_close(got, ref_prob, ref_logits, tol=1e-3):
    """softmax is 1/2-Lipschitz in the max-norm of the logits: |dprob| <= max_channel(logit bound) / 2."""
    bound = 0.5 * logit_bound(ref_logits, tol).max(dim=1, keepdim=True)[0]
    return bool(((got.cpu().double() - ref_prob.cpu().double()).abs() <= bound).all())


CFG_A = dict(BACKBONE='resnet18', NUM_BASES=64, NUM_EM_ITERS=4, SINGLE_OBJ=False)
CFG_A_SO = dict(BACKBONE='resnet18', NUM_BASES=64, NUM_EM_ITERS=4, SINGLE_OBJ=True)
CFG_B = dict(BACKBONE='resnet50', NUM_BASES=256, NUM_EM_ITERS=5, SINGLE_OBJ=False)


@pytest.mark.parametrize('kw,h,w,n_obj', [(CFG_A, 240, 432, 2), (CFG_A_SO, 240, 432, 1), (CFG_B, 480, 864, 2)],
                         ids=['configA', 'configA_single_obj', 'configB'])
def test_stages_vs_oracle(lib, kw, h, w, n_obj):
    _stages_vs_oracle(kw, h, w, n_obj)


def _stages_vs_oracle(kw, h, w, n_obj, book=None):
    cfg = O.make_cfg(**kw)
    model, sd = H.make_model_and_sd(cfg, wseed=11, device=DEV)
    if book is not None:
        model.book = book
    from swem_amd import synth
    out_hw = (h, w - 10)
    frames, m0 = synth.make_clip(t=2, h=h, w=w, n_obj=n_obj, out_hw=out_hw, seed=21)
    om = O.Model(sd, cfg)
    with torch.no_grad():
        # ---- encode_key (swem.py:39-43)
        oqk, oqv, os16, os8, os4 = om('encode_key', frames[:, 0])
        qk, qv, s16, s8, s4 = model('encode_key', frames[:, 0].to(DEV))
        assert qk.shape == oqk.shape and s4.shape == os4.shape
        for name, a, b in (('s4', s4, os4), ('s8', s8, os8), ('s16', s16, os16), ('qk16', qk, oqk), ('qv16', qv, oqv)):
            assert relmax(a, b) < 1e-4, 'encode_key %s rel err %.3g' % (name, relmax(a, b))
        # ---- encode_value (swem.py:45-62) on the oracle's s16
        mfull = F.interpolate(m0, size=(h, w), mode='nearest')
        omv = om('encode_value', frames[:, 0], mfull, os16)
        mv = model('encode_value', frames[:, 0].to(DEV), mfull.to(DEV), os16.to(DEV))
        assert mv.shape == omv.shape
        assert relmax(mv, omv) < 1e-4, 'encode_value rel err %.3g' % relmax(mv, omv)
        # ---- init + memorize (swem.py:64-86): same seeded random bases on both sides
        torch.manual_seed(5)
        om('init', oqk, omv, m0)
        torch.manual_seed(5)
        model('init', oqk.to(DEV), omv.to(DEV), m0.to(DEV))
        ob, hb = om.core.first.bases, model.swem_core.memories['first'].bases
        zr = ob['zita'].squeeze(-2).unsqueeze(-2)
        for name in ('kappa', 'nu'):
            err = ((hb[name].cpu() - ob[name]) * zr).abs().max() / (ob[name] * zr).abs().max()
            assert err < 5e-5, 'init %s mass-weighted rel err %.3g' % (name, err)
        assert relmax(hb["zita"], ob["zita"]) < 1e-4
        # ---- match (swem.py:88-90) with the ORACLE's bases injected into the HIP banks
        model.swem_core.memories['first'].bases = {k: t.to(DEV) for k, t in ob.items()}
        oqk1, oqv1, os16_1, os8_1, os4_1 = om('encode_key', frames[:, 1])
        octx, on = om('match', oqk1, oqv1)
        ctx, n = model('match', oqk1.to(DEV), oqv1.to(DEV))
        assert n == on == n_obj and ctx.shape == octx.shape
        assert relmax(ctx, octx) < 1e-4, 'match context rel err %.3g' % relmax(ctx, octx)
        # ---- decoder (networks.py:208-213) on the oracle's context.  A dozen chained convs with K up to 4608
        # accumulate fp32 rounding; the yardstick is the reference's own fp32 error against an fp64 run of the
        # same arithmetic: the HIP decoder must be within max(1e-3, 3x that) on the pre-sigmoid logit.
        s8e, s4e = os8_1.expand(on, -1, -1, -1), os4_1.expand(on, -1, -1, -1)
        o32 = O.decoder_logit(sd, octx, s8e, s4e)
        sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items() if k.startswith('decoder.')}
        o64 = O.decoder_logit(sd64, octx.double(), s8e.double(), s4e.double())
        from swem_amd.modules import to_pixel_major
        with ops.use_book(model.book):
            l4 = model.engine().decoder_logit(to_pixel_major(octx.to(DEV)), to_pixel_major(os8_1.to(DEV)),
                                              to_pixel_major(os4_1.to(DEV))).cpu()
        e_ref = float((o32.double() - o64).abs().max())
        e_hip = float((l4.double() - o64[:, 0]).abs().max())
        print('decoder logit: max|x| %.3g  reference fp32 err vs fp64 %.3g  HIP err vs fp64 %.3g'
              % (float(o64.abs().max()), e_ref, e_hip))
        assert e_hip < max(1e-3, 3 * e_ref), 'decoder logit err %.3g (reference fp32 floor %.3g)' % (e_hip, e_ref)
        # ---- segment (swem.py:92-108) end to end on the oracle's context
        tol = max(1e-3, 3 * e_ref)
        ologits, oprob = om('segment', on, octx, os8_1, os4_1, None, out_hw)
        logits, prob = model('segment', n, octx.to(DEV), os8_1.to(DEV), os4_1.to(DEV), None, out_hw)
        assert logits_close(logits, ologits, tol), 'logits abs err %.3g' % float((logits.cpu() - ologits).abs().max())
        assert probs_close(prob, oprob, ologits, tol)
        agree = float((prob.cpu().argmax(1) == oprob.argmax(1)).float().mean())
        assert agree > 0.9995, 'index-map agreement %.6f' % agree
        # valid_obj path (training call site, swem_trainer.py:79)
        valid = torch.ones(1, n_obj + 1)
        valid[0, -1] = 0
        ol2, _ = om('segment', on, octx, os8_1, os4_1, valid, out_hw)
        l2, _ = model('segment', n, octx.to(DEV), os8_1.to(DEV), os4_1.to(DEV), valid.to(DEV), out_hw)
        assert logits_close(l2, ol2, tol)
    return model


def test_batch_of_two_clips_vs_oracle(lib):
    """The trainer's call shape (swem_trainer.py:59-90): every mode with a BATCH of clips, B = 2 here, N = 2 objects each.
    The body of the reference's one_step loop -- encode_key, encode_value, init, match, segment(valid_obj), memorize, match
    on both banks -- against the oracle stage by stage (the oracle's inputs go into every HIP stage), clip-major layouts
    (B*N, ...) as the reference's flatten(0, 1)."""
    from swem_amd import synth
    cfg = O.make_cfg(**CFG_A)
    model, sd = H.make_model_and_sd(cfg, wseed=13, device=DEV)
    h, w, n_obj = 128, 192, 2
    clips = [synth.make_clip(t=3, h=h, w=w, n_obj=n_obj, seed=60 + b, all_masks=True) for b in range(2)]
    frames = torch.cat([c[0] for c in clips])                               # (2, 3, 3, h, w)
    m0 = torch.cat([c[1][0] for c in clips])                                # (2, N+1, h, w)
    valid = torch.tensor([[1., 1., 1.], [1., 1., 0.]])
    om = O.Model(sd, cfg)
    d = lambda t: t.to(DEV)
    with torch.no_grad():
        oqk, oqv, os16, os8, os4 = om('encode_key', frames[:, 0])
        qk, qv, s16, s8, s4 = model('encode_key', d(frames[:, 0]))
        for name, a, b in (('s4', s4, os4), ('s16', s16, os16), ('qk16', qk, oqk), ('qv16', qv, oqv)):
            assert a.shape == b.shape and relmax(a, b) < 1e-4, name
        omv = om('encode_value', frames[:, 0], m0, os16)
        mv = model('encode_value', d(frames[:, 0]), d(m0), d(os16))
        assert mv.shape == omv.shape == (2, n_obj, 512, h // 16, w // 16) and relmax(mv, omv) < 1e-4
        torch.manual_seed(5)
        om('init', oqk, omv, m0)
        torch.manual_seed(5)
        model('init', d(oqk), d(omv), d(m0))
        ob, hb = om.core.first.bases, model.swem_core.memories['first'].bases
        assert hb['kappa'].shape == ob['kappa'].shape == (2, n_obj, 2, 128, 64)
        zr = ob['zita'].squeeze(-2).unsqueeze(-2)
        for name in ('kappa', 'nu'):
            err = ((hb[name].cpu() - ob[name]) * zr).abs().max() / (ob[name] * zr).abs().max()
            assert err < 5e-5, 'init %s mass-weighted rel err %.3g' % (name, err)
        # frame 1: match on the first bank, segment with valid_obj, memorize -> the update bank
        model.swem_core.memories['first'].bases = {k: d(t) for k, t in ob.items()}
        oqk1, oqv1, os16_1, os8_1, os4_1 = om('encode_key', frames[:, 1])
        octx, on = om('match', oqk1, oqv1)
        ctx, n = model('match', d(oqk1), d(oqv1))
        assert n == on == n_obj and ctx.shape == octx.shape == (2 * n_obj, 512, h // 16, w // 16)
        assert relmax(ctx, octx) < 1e-4, 'match (one bank) context rel err %.3g' % relmax(ctx, octx)
        ologits, oprob = om('segment', on, octx, os8_1, os4_1, valid, (h, w))
        logits, prob = model('segment', n, d(octx), d(os8_1), d(os4_1), d(valid), (h, w))
        assert logits.shape == ologits.shape == (2, n_obj + 1, h, w)
        assert logits_close(logits, ologits, 1e-3), float((logits.cpu() - ologits).abs().max())
        assert float((prob.cpu().argmax(1) == oprob.argmax(1)).float().mean()) > 0.9995
        opred = oprob.argmax(1, keepdim=True)
        ohard = (opred.expand(-1, n_obj + 1, -1, -1) == torch.arange(n_obj + 1).view(1, -1, 1, 1)).long()
        omv1 = om('encode_value', frames[:, 1], oprob, os16_1)
        model('memorize', d(oqk1), d(omv1), d(ohard), d(oprob))
        om('memorize', oqk1, omv1, ohard, oprob)
        ou, hu = om.core.upd.bases, model.swem_core.memories['update'].bases
        assert hu['kappa'].shape == ou['kappa'].shape
        # frame 2: match on both banks (the oracle's)
        model.swem_core.memories['update'].bases = {k: d(t) for k, t in ou.items()}
        oqk2, oqv2, _, _, _ = om('encode_key', frames[:, 2])
        octx2, _ = om('match', oqk2, oqv2)
        ctx2, _ = model('match', d(oqk2), d(oqv2))
        assert relmax(ctx2, octx2) < 1e-4, 'match (two banks) context rel err %.3g' % relmax(ctx2, octx2)


def test_stages_vs_oracle_with_autotuner(lib):
    """The bench's warm-up: per-layer plans chosen by the on-device autotuner among fp32 MFMA, bf16x6 and bf16x3 (pre-split
    operands moved by LDS-DMA).  Same stage-by-stage bars as above; the book the tuner filled belongs to this test's model."""
    ops.AUTOTUNE = True
    ops.MATH_RAN = ran = {}
    try:
        model = _stages_vs_oracle(CFG_B, 480, 864, 2)
    finally:
        ops.AUTOTUNE = False
        ops.MATH_RAN = None
    hist = model.book.math_histogram()
    H.record_parity('stages_configB[autotuned]', {'plans_by_math': hist, 'plans_digest': model.book.digest(),
                                                  'conv_launches_by_math': {str(k): v for k, v in ran.items()}})
    assert hist['f16x3'] + hist['bf16x6'] > 0 and not hist['bf16x3'] and not ops.BOOK.conv, (hist, 'plans leaked into the default book')


@pytest.mark.parametrize('mode', ('f16x3', 'bf16x3', 'tuned'))
def test_stages_vs_oracle_config_b_in_bench_arithmetic(lib, mode):
    """Stage by stage at config B with f16x3 / bf16x3 FORCED on every layer / with the bench's committed plans: 1e-4 per stage."""
    book = ops.PlanBook()
    with H.arith(mode, book) as ar:
        _stages_vs_oracle(CFG_B, 480, 864, 2, book=book)
    H.record_parity('stages_configB[%s]' % mode, {'conv_launches_by_math': ar.summary()})


def _run_fixture(golden, name, kw, sub, mode='fp32'):
    fx = golden(name)
    cfg = O.make_cfg(**kw)
    model, sd = H.make_model_and_sd(cfg, int(fx['wseed']), device=DEV)
    frames, m0 = H.clip_from_fixture(fx)
    t = frames.shape[1]
    trace = []
    with torch.no_grad(), H.arith(mode, model, need_bf16x3=name.startswith('g7')):
        torch.manual_seed(77)
        preds, scores = evaluator.evaluate_davis_seq(model, frames.to(DEV), [m0.to(DEV)] + [None] * (t - 1),
                                                      (int(fx['out_h']), int(fx['out_w'])), trace)
    torch.cuda.synchronize()
    rows = []
    for i in range(t - 1):
        dl = float((trace[i]['logits'][:, :, ::sub, ::sub].cpu() - fx['logits%d' % i]).abs().max())
        agree = float((preds[i].cpu().to(torch.uint8) == fx['pred%d' % i]).float().mean())
        dctx = float((trace[i]['context'][:, ::8].cpu() - fx['ctx%d' % i]).abs().max())
        rows.append((dl, agree, dctx))
    return fx, rows, trace


@pytest.mark.parametrize('name,kw,sub,mode', [('g6_configA.npz', CFG_A_SO, 2, 'fp32'), ('g6_configA_mo.npz', CFG_A, 2, 'fp32'),
                                              ('g6_configA_mo.npz', CFG_A, 2, 'bf16x3'), ('g6_configA_mo.npz', CFG_A, 2, 'f16x3'),
                                              ('g7_configB.npz', CFG_B, 8, 'fp32'), ('g7_configB.npz', CFG_B, 8, 'bf16x3'),
                                              ('g7_configB.npz', CFG_B, 8, 'f16x3'), ('g7_configB.npz', CFG_B, 8, 'tuned')],
                         ids=['configA_single_object', 'configA_multi_object', 'configA_multi_object_bf16x3',
                              'configA_multi_object_f16x3', 'configB_480p_r50_k256', 'configB_480p_r50_k256_bf16x3',
                              'configB_480p_r50_k256_f16x3', 'configB_480p_r50_k256_tuned'])
def test_clip_vs_golden(lib, golden, name, kw, sub, mode):
    """Free-running clips against the reference's outputs (BASELINE configs[0] and configs[1]), the multi-object ones also
    with bf16x3 forced on every conv layer, config B also with the bench's committed plans (helpers.arith)."""
    fx, rows, trace = _run_fixture(golden, name, kw, sub, mode)
    floor, agree64 = fx['floor64'], fx['agree64']
    for i, (dl, agree, dctx) in enumerate(rows):
        print('%s frame %d: |dlogits| %.3g (reference fp32-vs-fp64 floor %.3g)  index agree %.6f (floor %.6f)  |dctx| %.3g'
              % (name, i + 1, dl, float(floor[i]), agree, float(agree64[i]), dctx))
    H.record_parity('free_running_%s[%s]' % (name.split('.')[0], mode), [
        {'frame': i + 1, 'dlogits_max': dl, 'reference_fp32_vs_fp64_floor': float(floor[i]), 'index_agreement': agree,
         'reference_fp32_vs_fp64_agreement': float(agree64[i]), 'dcontext_max': dctx} for i, (dl, agree, dctx) in enumerate(rows)])
    # before the memory is involved the 1e-4 bar holds
    assert relmax(trace[0]['qk16'], fx['qk16_0']) < 1e-4
    for i, (dl, agree, _) in enumerate(rows):
        assert dl < max(1e-3, 2 * float(floor[i])), 'frame %d logits %.3g vs floor %.3g' % (i + 1, dl, float(floor[i]))
        assert agree >= min(0.9995, float(agree64[i]) - 0.01), 'frame %d index agreement %.6f' % (i + 1, agree)


def test_fps_meter_and_sequences(lib):
    """basic_evaluator.py:171-176 semantics: every frame incl. frame 0 counts; synchronised wall time."""
    cfg = O.make_cfg(**CFG_A)
    model, _ = H.make_model_and_sd(cfg, wseed=3, device=DEV)
    from swem_amd import synth
    frames, m0 = synth.make_clip(t=3, h=128, w=192, n_obj=2, seed=2)
    res, meter = evaluator.run_sequences(model, [(frames.to(DEV), m0.to(DEV), (128, 192))] * 2)
    assert meter.frame_n == 6 and meter.total_time > 0 and len(res) == 2 and len(res[0]) == 2
    assert res[0][0].dtype == torch.int64 and res[0][0].shape == (1, 128, 192)


def test_frame_graph_matches_eager(lib):
    """The HIP-graph replay of the steady-state frame (evaluator.FrameGraph) is bit-identical to eager launches,
    including the recurrent memory state, over several frames."""
    from swem_amd import synth
    cfg = O.make_cfg(**CFG_A)
    frames, m0 = synth.make_clip(t=6, h=128, w=192, n_obj=2, seed=9)
    frames, m0 = frames.to(DEV), m0.to(DEV)

    def run(use_graph):
        model, _ = H.make_model_and_sd(cfg, wseed=4, device=DEV)
        with torch.no_grad():
            torch.manual_seed(11)
            mk16, _, s16, _, _ = model('encode_key', frames[:, 0])
            mv16 = model('encode_value', frames[:, 0], m0, s16)
            model('init', mk16, mv16, m0)
            preds = [evaluator.frame_step(model, frames[:, i], (128, 192)).clone() for i in (1, 2)]
            g = evaluator.FrameGraph(model, frames[:, 1].shape, (128, 192)).capture(frames[:, 1]) if use_graph else None
            for i in (3, 4, 5, 3, 4):
                p = g.run(frames[:, i]) if use_graph else evaluator.frame_step(model, frames[:, i], (128, 192))
                preds.append(p.clone())
            bases = {k: v.clone() for k, v in model.swem_core.memories['update'].bases.items()}
        torch.cuda.synchronize()
        return preds, bases

    pe, be = run(False)
    pg, bg = run(True)
    for a, b in zip(pe, pg):
        assert torch.equal(a, b)
    for k in be:
        assert torch.equal(be[k], bg[k]), k


def test_pipelined_frame_graph_matches_eager(lib):
    """evaluator.PipelinedFrameGraph (frame t-1's encode_value + memorize on a forked branch under frame t's encode_key) gives
    the index maps of the sequential loop bit for bit, and after flush() the same memory."""
    from swem_amd import synth
    cfg = O.make_cfg(**CFG_A)
    frames, m0 = synth.make_clip(t=6, h=128, w=192, n_obj=2, seed=9)
    frames, m0 = frames.to(DEV), m0.to(DEV)
    order = (3, 4, 5, 3, 4, 5, 2)

    def run(pipelined):
        model, _ = H.make_model_and_sd(cfg, wseed=4, device=DEV)
        with torch.no_grad():
            torch.manual_seed(11)
            mk16, _, s16, _, _ = model('encode_key', frames[:, 0])
            mv16 = model('encode_value', frames[:, 0], m0, s16)
            model('init', mk16, mv16, m0)
            preds = [evaluator.frame_step(model, frames[:, i], (128, 192)).clone() for i in (1, 2)]
            g = evaluator.PipelinedFrameGraph(model, frames[:, 1].shape, (128, 192)).capture(frames[:, 1]) if pipelined else None
            for i in order:
                p = g.run(frames[:, i]) if pipelined else evaluator.frame_step(model, frames[:, i], (128, 192))
                preds.append(p.clone())
            if pipelined:
                g.flush()
            bases = {k: v.clone() for k, v in model.swem_core.memories['update'].bases.items()}
        torch.cuda.synchronize()
        return preds, bases

    pe, be = run(False)
    pg, bg = run(True)
    for i, (a, b) in enumerate(zip(pe, pg)):
        assert torch.equal(a, b), 'frame %d' % i
    for k in be:
        assert torch.equal(be[k], bg[k]), k


def test_planes_only_outputs_leave_the_frames_unchanged(lib):
    """ops.PLANES_ONLY (conv1 / conv2 inside every block write only their bf16 planes once the next convolution reads planes,
    engine.py) against the same frames with every fp32 map written: index maps, probabilities and the memory after the last
    frame are BIT-IDENTICAL, and the switch really takes effect (planes-only tensors are produced from the second frame on)."""
    from swem_amd import synth
    cfg = O.make_cfg(**CFG_A)
    frames, m0 = synth.make_clip(t=6, h=128, w=192, n_obj=2, seed=12)
    frames, m0 = frames.to(DEV), m0.to(DEV)

    def run(planes_only):
        model, _ = H.make_model_and_sd(cfg, wseed=5, device=DEV)
        model.book.fallback = 0x30011                   # every layer on the pre-split kernel (bf16x3), 64x64 tile
        seen = []
        real = ops.conv2d

        def spy(*a, **kw):
            y = real(*a, **kw)
            seen.append(bool(y.__dict__.get('_swem_planes_only')))
            return y
        with torch.no_grad(), ops.flags(PLANES_ONLY=planes_only):
            ops.conv2d = spy
            try:
                torch.manual_seed(3)
                mk16, _, s16, _, _ = model('encode_key', frames[:, 0])
                model('init', mk16, model('encode_value', frames[:, 0], m0, s16), m0)
                preds = [evaluator.frame_step(model, frames[:, i], (128, 192)).clone() for i in range(1, 6)]
            finally:
                ops.conv2d = real
            bases = {kk: v.clone() for kk, v in model.swem_core.memories['update'].bases.items()}
        torch.cuda.synchronize()
        return preds, bases, sum(seen), len(seen)

    pa, ba, n_a, tot_a = run(True)
    pb, bb, n_b, tot_b = run(False)
    assert n_b == 0 and tot_a == tot_b and n_a > 0.25 * tot_a, (n_a, tot_a)    # (ResNet-18 encoders: one conv in three sits inside a block)
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)
    for kk in ba:
        assert torch.equal(ba[kk], bb[kk]), kk


@pytest.mark.parametrize('overlap', [True, False], ids=['keys_on_side_stream', 'one_stream'])
def test_lookahead_graph_matches_sequential_loop(lib, overlap):
    """evaluator.LookaheadGraph: k = 3 frames per replay, the key encoder of the NEXT three frames as one B = 3 pass next to
    (or behind) the current three frames' chains.  With plans that do not depend on the batch (PlanBook.fallback = a tile
    without K-split: every output element is one k-ordered MFMA chain whatever the grid) the index maps AND the memory after
    the last frame are those of the frame-by-frame loop BIT FOR BIT; the batched outputs carry the bf16 planes their consumers
    asked for (no split launch inside the captured chains).  The side-stream variant also runs the key half of every memorize
    beside the value encoder (em_overlap: SWEM.memorize_begin / memorize_end)."""
    from swem_amd import synth
    cfg = O.make_cfg(**CFG_A)
    k, t = 3, 12
    frames, m0 = synth.make_clip(t=t, h=128, w=192, n_obj=2, seed=9)
    frames, m0 = frames.to(DEV), m0.to(DEV)

    def run(lookahead, math):
        model, _ = H.make_model_and_sd(cfg, wseed=4, device=DEV)
        model.book.fallback = 0x111 | math << 16        # 64x64 tile, no K-split, fp32 MFMA or bf16x3
        with torch.no_grad():
            torch.manual_seed(11)
            mk16, _, s16, _, _ = model('encode_key', frames[:, 0])
            model('init', mk16, model('encode_value', frames[:, 0], m0, s16), m0)
            preds = [evaluator.frame_step(model, frames[:, i], (128, 192)).clone() for i in (1, 2)]
            if lookahead:
                g = evaluator.LookaheadGraph(model, frames[:, 1].shape, (128, 192), k, overlap=overlap,
                                             em_overlap=overlap).capture(frames[0, 3:3 + k])
                g.prime(frames[0, 3:3 + k])
                for i in range(3, t, k):
                    nxt = frames[0, i + k:i + 2 * k] if i + 2 * k <= t else None
                    preds += [p.clone() for p in g.run(nxt)]
                torch.cuda.synchronize()
                # the captured chains find the planes on the batched key-encoder outputs
                assert any('_swem_split' in ops.batch_item(x.__dict__['_swem_nhwc'], 1).__dict__ for x in g.keys[0]
                           if '_swem_nhwc' in x.__dict__) or math == 0
            else:
                preds += [evaluator.frame_step(model, frames[:, i], (128, 192)).clone() for i in range(3, t)]
            bases = {kk: v.clone() for kk, v in model.swem_core.memories['update'].bases.items()}
        torch.cuda.synchronize()
        return preds, bases

    for math in (0, 3):
        pe, be = run(False, math)
        pg, bg = run(True, math)
        assert len(pe) == len(pg) == t - 1
        for i, (a, b) in enumerate(zip(pe, pg)):
            assert torch.equal(a, b), 'math %d frame %d' % (math, i + 1)
        for kk in be:
            assert torch.equal(be[kk], bg[kk]), (math, kk)


def test_persistent_pack_is_kept_across_frames(lib):
    """SWEMCore keeps ONE packed copy of the banks: after the first two frames no frame re-packs a bank or allocates a new
    pack (memorize writes the new bank's packed form itself, matching reads it)."""
    from swem_amd import _lib, synth
    cfg = O.make_cfg(**CFG_A)
    model, _ = H.make_model_and_sd(cfg, 5, device=DEV)
    frames, m0 = synth.make_clip(t=6, h=128, w=192, n_obj=2, seed=9)
    calls, packs = [], []
    real_call, real_new = _lib.call, ops.new_pack

    def counting(name, *a):
        if name in ('swem_match_pack_bank_f32', 'swem_em_pack_bases_f32', 'swem_em_norm_bases_f32'):
            calls.append(name)
        return real_call(name, *a)

    def new_pack(*a):
        packs.append(1)
        return real_new(*a)
    _lib.call, ops.new_pack = counting, new_pack
    try:
        with torch.no_grad():
            torch.manual_seed(1)
            evaluator.evaluate_davis_seq(model, frames.to(DEV), [m0.to(DEV)] + [None] * 5, (128, 192))
    finally:
        _lib.call, ops.new_pack = real_call, real_new
    assert len(packs) == 1, packs
    # frame 0 packs the random prior once (memorize without a packed prior); nothing after it
    assert len(calls) <= 2, calls


@pytest.mark.parametrize('mode', ('fp32', 'f16x3', 'bf16x3'))
def test_ytvos_loop_and_tta_vs_golden(lib, golden, mode):
    """f1 rows: evaluate_ytvos_seq with an object that appears at frame 2 (exercises N_new > 0 in swem() and
    MemoryBank.add_new) and the multi-scale + flip TTA, against the reference's index maps."""
    from swem_amd import synth
    fx = golden('g8_ytvos_tta.npz')
    cfg = O.make_cfg(**CFG_A)
    model, _ = H.make_model_and_sd(cfg, int(fx['wseed']), device=DEV)
    frames, per_frame = synth.make_clip(t=5, h=240, w=432, n_obj=2, out_hw=(240, 432), seed=int(fx['seed']), all_masks=True)
    masks = [None if m is None else m.to(DEV) for m in H.ytvos_masks(per_frame, 2)]
    with torch.no_grad(), H.arith(mode, model):
        torch.manual_seed(78)
        preds = evaluator.evaluate_ytvos_seq(model, frames.to(DEV), masks, (240, 432))
    core = model.swem_core
    assert core.memories['first'].bases['kappa'].shape[1] == 2 and core.memories['update'].bases['kappa'].shape[1] == 2
    for i, p in enumerate(preds):
        agree = float((p.cpu().to(torch.uint8) == fx['pred%d' % i]).float().mean())
        print('ytvos frame %d: index agreement %.6f (reference fp32-vs-fp64 %.6f)' % (i + 1, agree, float(fx['agree64'][i])))
        assert agree >= min(0.9995, float(fx['agree64'][i]) - 0.01)
    assert int(preds[0].max()) <= 1 and int(preds[-1].max()) == 2
    rec = {'ytvos_index_agreement': [float((p.cpu().to(torch.uint8) == fx['pred%d' % i]).float().mean())
                                     for i, p in enumerate(preds)],
           'reference_fp32_vs_fp64_agreement': [float(v) for v in fx['agree64']]}
    with torch.no_grad(), H.arith(mode, model):
        tta = evaluator.evaluate_davis_seq_ms(H.SeededInit(model, 79), frames[:, :3].to(DEV),
                                              [per_frame[0].to(DEV), None, None], (240, 432), scales=(240, 288), is_flip=True)
    for i, p in enumerate(tta):
        agree = float((p.cpu().to(torch.uint8) == fx['tta%d' % i]).float().mean())
        print('tta frame %d: index agreement %.6f' % (i + 1, agree))
        rec.setdefault('tta_index_agreement', []).append(agree)
        # (free-running: four passes at two scales, each with its own chaotic memory, averaged; measured 0.9999 / 0.993 in
        # every arithmetic -- the second frame's pixels that differ sit where the averaged maps tie)
        assert agree >= 0.99, agree
    H.record_parity('free_running_g8_ytvos_tta[%s]' % mode, rec)


@pytest.mark.parametrize('n_obj,h,w,bases,topl', [(1, 96, 160, 64, 64), (5, 112, 176, 64, 32), (3, 80, 144, 128, 64)],
                         ids=['one_object', 'five_objects_topl32', 'three_objects_k128'])
def test_edge_shapes_free_running(lib, n_obj, h, w, bases, topl):
    """Edge cases of the frame loop: a single object, the reference's maximum of five objects, a top-l smaller than the
    bank, sizes whose 1/16 grid is not a multiple of the 32-pixel tile, an EMPTY initial mask for one object and an
    object that leaves the frame.  Free-running against the CPU oracle (index agreement; logits on frame 1)."""
    from swem_amd import synth
    cfg = O.make_cfg(BACKBONE='resnet18', NUM_BASES=bases, NUM_EM_ITERS=3, TOPL=topl)
    model, sd = H.make_model_and_sd(cfg, wseed=21 + n_obj, device=DEV)
    frames, m0 = synth.make_clip(t=3, h=h, w=w, n_obj=n_obj, seed=40 + n_obj)
    if n_obj >= 3:                       # object n_obj has no pixels at all in the first frame (empty mask)
        m0[:, 0] += m0[:, n_obj]
        m0[:, n_obj] = 0
    with torch.no_grad():
        torch.manual_seed(3)
        otr = []
        opreds, _ = O.evaluate_seq(O.Model(sd, cfg), frames, [m0, None, None], (h, w), otr)
        # the yardstick of a free-running clip: the reference arithmetic itself in float64 (same weights, same random bases)
        sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
        torch.manual_seed(3)
        init32 = O.random_init
        O.random_init = lambda size, valdim, dtype=torch.float32: tuple(t.double() for t in init32(size, valdim))
        try:
            opreds64, _ = O.evaluate_seq(O.Model(sd64, cfg), frames.double(), [m0.double(), None, None], (h, w))
        finally:
            O.random_init = init32
        torch.manual_seed(3)
        tr = []
        preds, scores = evaluator.evaluate_davis_seq(model, frames.to(DEV), [m0.to(DEV), None, None], (h, w), tr)
    assert scores[0].shape == (1, n_obj + 1, h, w) and torch.isfinite(scores[-1]).all()
    assert float((scores[0].sum(1) - 1).abs().max()) < 1e-5
    assert relmax(tr[0]['qk16'], otr[0]['qk16']) < 1e-4
    rec = []
    for i in range(2):
        agree = float((preds[i].cpu() == opreds[i]).float().mean())
        agree64 = float((opreds[i] == opreds64[i]).float().mean())
        rec.append({'frame': i + 1, 'index_agreement': agree, 'reference_fp32_vs_fp64_agreement': agree64})
        print('edge %d objects frame %d: index agreement %.4f (reference fp32 vs fp64: %.4f)' % (n_obj, i + 1, agree, agree64))
    H.record_parity('free_running_edge_%dobj_k%d' % (n_obj, bases), rec)
    # free-running, so only as good as the reference agrees with itself across precisions (the tight per-frame bars are in
    # tests/test_gpu_parity.py::test_teacher_forced_edge_shapes)
    for r in rec:
        assert r['index_agreement'] >= min(0.9995, 1 - 3 * (1 - r['reference_fp32_vs_fp64_agreement']) - 0.01), r
    bases_ = model.swem_core.memories['update'].bases
    assert bases_['kappa'].shape == (1, n_obj, 2, 128, bases) and torch.isfinite(bases_['nu']).all()


def test_memory_is_constant_size_over_a_long_clip(lib):
    """BASELINE config E in miniature: memorise every frame of a longer clip; the state never grows (two banks of
    K bases per object and class) and stays finite."""
    from swem_amd import synth
    cfg = O.make_cfg(**CFG_A)
    model, _ = H.make_model_and_sd(cfg, wseed=6, device=DEV)
    frames, m0 = synth.make_clip(t=6, h=96, w=160, n_obj=2, seed=77)
    frames, m0 = frames.to(DEV), m0.to(DEV)
    with torch.no_grad():
        mk16, _, s16, _, _ = model('encode_key', frames[:, 0])
        model('init', mk16, model('encode_value', frames[:, 0], m0, s16), m0)
        shapes = None
        for rep in range(25):
            evaluator.frame_step(model, frames[:, 1 + rep % 5], (96, 160))
            mem = model.swem_core.memories
            cur = {k: tuple(v.shape) for k, v in mem['update'].bases.items()}
            assert shapes is None or cur == shapes
            shapes = cur
            zsum = float(mem['update'].bases['zita'].sum())
        assert mem['first'].bases['kappa'].shape == mem['update'].bases['kappa'].shape == (1, 2, 2, 128, 64)
        assert all(torch.isfinite(v).all() for v in mem['update'].bases.values()) and zsum > 0


def test_output_staging_pack_and_png_writer(lib, tmp_path):
    """f4 (basic_evaluator.py:176-190): int64 maps -> uint8 on the device, async copy, palette PNGs."""
    import numpy as np
    from PIL import Image
    from swem_amd import io, ops
    g = torch.Generator().manual_seed(4)
    preds = [torch.randint(0, 4, (1, 37, 53), generator=g).to(DEV) for _ in range(3)]
    assert torch.equal(ops.pack_u8(preds[0]).cpu(), preds[0].cpu().to(torch.uint8))
    w = io.IndexMapWriter(str(tmp_path))
    w.submit('seq', preds)
    onehot = torch.nn.functional.one_hot(preds[0][0].cpu(), 4).permute(2, 0, 1)[None].float().to(DEV)
    w.save_first('seq', onehot)
    assert w.close() == 3
    for t in range(3):
        back = np.array(Image.open(str(tmp_path / 'seq' / ('%05d.png' % (t + 1)))))
        assert np.array_equal(back, preds[t][0].cpu().numpy().astype(np.uint8))
    assert np.array_equal(np.array(Image.open(str(tmp_path / 'seq' / '00000.png'))), preds[0][0].cpu().numpy())
    fr = torch.rand(1, 2, 3, 100, 180, generator=g).to(DEV)
    m = torch.zeros(1, 1, 3, 100, 180, device=DEV)
    inf, inm = io.stage_sequence(fr, m)
    ref = torch.nn.functional.interpolate(fr[0].cpu(), size=(480, 864), mode='bicubic', align_corners=False)
    assert inf.shape == (1, 2, 3, 480, 864) and float((inf[0].cpu() - ref).abs().max()) < 2e-5 and inm[1] is None


def test_module_level_vectors_vs_reference(lib, golden):
    """G4 / G5 on the HIP path: ResBlock (identity and 3x3 shortcut), UpsampleBlock, FeatureFusionBlock with CBAM, the GLU
    fusion layer, Decoder + decode/aggregate with valid_obj, against outputs recorded from the reference's own classes."""
    from swem_amd import engine as E, networks as Nw, weights
    from swem_amd.swem import SWEM
    fx = golden('g45_modules.npz')
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)
    back = lambda t: t.permute(0, 3, 1, 2).cpu()

    def load(mod, tag):
        sd = {k[len(tag) + 3:]: v for k, v in fx.items() if k.startswith(tag + '_m.')}
        mod.load_state_dict(sd, strict=True)
        return mod.to(DEV)
    for tag, (ci, co) in (('rb_same', (32, 32)), ('rb_down', (64, 32))):
        rb = E._ResBlock(load(Nw.ResBlock(ci, co), tag))
        assert relmax(back(rb([nhwc(fx[tag + '_x'])])), fx[tag + '_y']) < 2e-5
    ub = load(Nw.UpsampleBlock(32, 64, 32), 'up')
    sk = ops.conv2d([nhwc(fx['up_skip'])], ops.pack_conv(ub.skip_conv.weight, ub.skip_conv.bias))
    y = E._ResBlock(ub.out_conv)([ops.upsample_add(sk, nhwc(fx['up_low']))])
    assert relmax(back(y), fx['up_y']) < 2e-5
    fb = load(Nw.FeatureFusionBlock(64, 64), 'ffb')
    x = E._ResBlock(fb.block1)([nhwc(fx['ffb_x']), nhwc(fx['ffb_f16'])])
    att = fb.attention
    x = ops.cbam_residual(x, att.ChannelGate.mlp[1].weight.detach(), att.ChannelGate.mlp[1].bias.detach(),
                          att.ChannelGate.mlp[3].weight.detach(), att.ChannelGate.mlp[3].bias.detach(),
                          att.SpatialGate.spatial.conv.weight.detach(), att.SpatialGate.spatial.conv.bias.detach())
    assert relmax(back(E._ResBlock(fb.block2)([x])), fx['ffb_y']) < 5e-5
    glu = ops.pack_glu(fx['ffl_layer_f.weight'].to(DEV), fx['ffl_layer_f.bias'].to(DEV), fx['ffl_layer_a.weight'].to(DEV),
                       fx['ffl_layer_a.bias'].to(DEV))
    assert relmax(back(ops.conv2d([nhwc(fx['ffl_x'])], glu)), fx['ffl_y']) < 2e-5
    cfg = O.make_cfg(BACKBONE='resnet18', NUM_BASES=64)
    model = SWEM(cfg)
    full = weights.fill_state_dict(model.state_dict(), seed=int(fx['dec_wseed']), backbone='resnet18')
    full['decoder.pred.weight'] = full['decoder.pred.weight'] * float(fx['dec_pred_scale'])
    model.load_state_dict(full)
    model = model.eval().to(DEV)
    with torch.no_grad():
        lg, pr = model('segment', 2, fx['dec_ctx'].to(DEV), fx['dec_s8'].to(DEV), fx['dec_s4'].to(DEV),
                       fx['dec_valid'].to(DEV), (61, 90))
        lg2, _ = model('segment', 2, fx['dec_ctx'].to(DEV), fx['dec_s8'].to(DEV), fx['dec_s4'].to(DEV), None, (64, 96))
    assert logits_close(lg.cpu(), fx['dec_logits'])
    assert probs_close(pr.cpu(), fx['dec_prob'], fx['dec_logits'])
    assert logits_close(lg2.cpu(), fx['dec_logits_novalid'])


@pytest.mark.parametrize('lanes,lookahead', [(2, 0), (1, 0), (2, 2), (1, 2)],
                         ids=['two_lanes', 'one_lane_pipelined', 'two_lanes_lookahead2', 'one_lane_lookahead2'])
def test_sequence_pool_equals_sequential_evaluation(lib, lanes, lookahead):
    """Two sequences in flight per GPU (two streams, HIP-graph replay re-bound from sequence to sequence) -- or one lane, whose
    graph is the software-pipelined PipelinedFrameGraph -- give the same index maps as evaluating the sequences one after
    another with the plain loop; so do the look-ahead graphs (two frames per replay, the key encoder batched over them; the
    plans are pinned to a batch-invariant tile so that the comparison is bit for bit)."""
    cfg = O.make_cfg(**CFG_A)
    models = [H.make_model_and_sd(cfg, 5, DEV)[0] for _ in range(lanes)]
    models[0].book.fallback = 0x111
    seqs, seeds = [], [11, 12, 13, 14, 15]
    # same shape three times (the lane's graph is re-bound), then one object (re-captured), then another frame size
    for k, (t, hh, ww, n) in enumerate(((5, 240, 432, 2), (4, 240, 432, 2), (6, 240, 432, 2), (5, 240, 432, 1),
                                        (4, 192, 320, 2))):
        frames, m0 = synth_clip(t, hh, ww, n, 40 + k)
        seqs.append((frames.to(DEV), m0.to(DEV), (hh, ww)))
    ref = []
    for (frames, m0, out), sd_ in zip(seqs, seeds):
        torch.manual_seed(sd_)
        with torch.no_grad():
            preds, _ = evaluator.evaluate_davis_seq(models[0], frames, [m0] + [None] * (frames.shape[1] - 1), out)
        ref.append([p.clone() for p in preds])
    pool = evaluator.SequencePool(models, use_graph=True, lookahead=lookahead, plans=None)   # (the book as the reference runs used it)
    got = pool.run(seqs, seeds=seeds)
    torch.cuda.synchronize()
    assert pool.graphs[0] is not None
    if lookahead:
        assert isinstance(pool.graphs[0], evaluator.LookaheadGraph) and pool.graphs[0].overlap == (lanes == 1)
    else:
        assert isinstance(pool.graphs[0], evaluator.PipelinedFrameGraph) == (lanes == 1)
    for r, g_ in zip(ref, got):
        assert len(r) == len(g_)
        for a, b in zip(r, g_):
            assert torch.equal(a, b)
    # a second run() on the same pool starts its sequence indices at 0 again: the lanes' graphs (and, with one lane, the
    # pipelined graph's pending frame) must be re-bound to the new sequences, not replayed on the previous ones' banks
    got2 = pool.run(seqs[:3][::-1], seeds=seeds[:3][::-1])
    torch.cuda.synchronize()
    for r, g_ in zip(ref[:3][::-1], got2):
        assert len(r) == len(g_)
        for a, b in zip(r, g_):
            assert torch.equal(a, b)
    assert all(m.book is models[0].book for m in models)


@pytest.mark.parametrize('backbone', ['resnet18', 'resnet50'])
def test_shared_source_split_of_the_fusion_block(lib, backbone):
    """engine._SharedSourceSplit (round 6): the value encoder's ResBlock(cat[x, f16]) (networks.py:35-50, 113-129) with the clip's key
    feature split off its two 1280-channel convolutions -- conv_s(f16) once per CLIP, conv_x(x_n) per object with conv_s's result as
    the residual addend -- against the one-launch form on the same inputs: the same function up to fp32 summation order, for one
    clip (the addend broadcast over the objects) and for two clips in one batch (repeated per object), two and three objects, in the
    fp32 and the f16x3 arithmetic; one object keeps the one-launch form."""
    from swem_amd import synth
    cfg = O.make_cfg(BACKBONE=backbone, NUM_BASES=64)
    model, _ = H.make_model_and_sd(cfg, wseed=6, device=DEV)
    g = torch.Generator().manual_seed(3)
    for math in (0, 7):
        model.book.fallback = 0x111 | math << 16
        for B, N in ((1, 2), (2, 2), (1, 3), (1, 1)):
            frames = torch.rand(B, 3, 96, 160, generator=g).to(DEV)
            masks = torch.rand(B, N + 1, 96, 160, generator=g)
            masks = (masks / masks.sum(1, keepdim=True)).to(DEV)
            with torch.no_grad():
                _, _, s16, _, _ = model('encode_key', frames)
                seen = []
                real = ops.conv2d

                def spy(srcs, *a, **kw):
                    seen.append(len(srcs))
                    return real(srcs, *a, **kw)
                ops.conv2d = spy
                try:
                    a = model('encode_value', frames, masks, s16)
                finally:
                    ops.conv2d = real
                with ops.flags(SPLIT_SHARED_SOURCE=False):
                    b = model('encode_value', frames, masks, s16)
                # a key feature that does not come with the encoder's "ends in a ReLU" tag (a caller's own tensor: it may be
                # negative) takes the two-launch form of the split, input ReLU on conv1's half only
                neg = s16.clone() - 0.3
                a2 = model('encode_value', frames, masks, neg)
                with ops.flags(SPLIT_SHARED_SOURCE=False):
                    b2 = model('encode_value', frames, masks, neg)
            assert float((a2 - b2).abs().max() / b2.abs().max()) < 1e-5, (math, B, N)
            # (ResNet-18: the block maps 256 + 256 -> 512 channels with an IDENTITY shortcut -- the concatenated tensor itself is the
            # residual, networks.py:22-32 -- and keeps the one-launch form; ResNet-50: 256 + 1024 -> 512 with a downsample conv)
            has_split = model.engine().fuse1_split is not None
            assert has_split == (backbone == 'resnet50')
            split_ran = has_split and N > 1
            assert (2 in seen) == (not split_ran), (B, N, seen)   # the two-source (cat) launches are gone where objects share f16
            assert a.shape == b.shape == (B, N, 512, 6, 10)
            err = float((a - b).abs().max() / b.abs().max())
            assert err < (1e-5 if split_ran else 1e-12), (math, B, N, err)
            assert split_ran or torch.equal(a, b)


@pytest.mark.parametrize('forks,fused', [('none', False), (None, False), ('none', True), ('none', 'em')],
                         ids=['linear_graph', 'forked_graph', 'fusion_conv_batched', 'em_and_matching_batched'])
def test_lockstep_graph_matches_sequential_loops(lib, forks, fused):
    """evaluator.LockstepGraph (round 6): THREE sequences in lock step, k = 3 frames of each per replay -- one key-encoder pass over
    the 3 x 3 frames of the next group, decoder and value encoder batched over the objects of the three sequences, match and
    memorize per sequence (one after the other, or on forked streams inside the graph; the fusion conv of matching per sequence or,
    fuse_batched, once for all of them).  With plans that do not depend on the
    batch (a tile without K-split) every sequence's index maps AND its memory after the last frame are those of its own
    frame-by-frame loop BIT FOR BIT, in both arithmetics; a second set of sequences re-bound to the same graphs likewise.  (With the
    fusion conv batched too, its GLU launch has another row count than the per-sequence one and the pinned plan does not bind a GLU
    layer: fp32 summation order differs -- index maps agree on > 0.9995 of the pixels, memories to 1e-2 of their range.)"""
    from swem_amd import synth
    cfg = O.make_cfg(**CFG_A)
    k, t, S = 3, 12, 3
    clips = []
    for s_ in range(2 * S):
        frames, m0 = synth.make_clip(t=t, h=128, w=192, n_obj=2, seed=20 + s_)
        clips.append((frames.to(DEV), m0.to(DEV)))

    def start(model, frames, m0, seed):
        torch.manual_seed(seed)
        mk16, _, s16, _, _ = model('encode_key', frames[:, 0])
        model('init', mk16, model('encode_value', frames[:, 0], m0, s16), m0)
        return [evaluator.frame_step(model, frames[:, i], (128, 192)).clone() for i in (1, 2)]

    def bases_of(model):
        return {kk: v.clone() for kk, v in model.swem_core.memories['update'].bases.items()}

    for math in (0, 3):
        with torch.no_grad():
            ref = []
            model, _ = H.make_model_and_sd(cfg, wseed=4, device=DEV)
            model.book.fallback = 0x111 | math << 16
            for s_, (frames, m0) in enumerate(clips):
                preds = start(model, frames, m0, 30 + s_)
                preds += [evaluator.frame_step(model, frames[:, i], (128, 192)).clone() for i in range(3, t)]
                ref.append((preds, bases_of(model)))
            models = [H.make_model_and_sd(cfg, wseed=4, device=DEV)[0] for _ in range(S)]
            for m in models:
                m.book = models[0].book
            models[0].book.fallback = 0x111 | math << 16
            g = None
            for half in (0, 1):
                mine = clips[half * S:(half + 1) * S]
                preds = [start(m, f, m0, 30 + half * S + s_) for s_, (m, (f, m0)) in enumerate(zip(models, mine))]
                stack = lambda i: torch.stack([f[0, i:i + k] for f, _ in mine], dim=1)
                if g is None:
                    g = evaluator.LockstepGraph(models, mine[0][0][:, 1].shape, (128, 192), k, forks=forks, fuse_batched=fused is True,
                                                batched_em=fused == 'em').capture(stack(3))
                else:
                    assert g.rebind()
                g.prime(stack(3))
                for i in range(3, t, k):
                    out = g.run(stack(i + k) if i + 2 * k <= t else None)
                    for p_ in out:
                        assert p_.shape == (S, 128, 192)
                        for s_ in range(S):
                            preds[s_].append(p_[s_:s_ + 1].clone())
                torch.cuda.synchronize()
                for s_ in range(S):
                    rp, rb = ref[half * S + s_]
                    assert len(rp) == len(preds[s_]) == t - 1
                    got = bases_of(models[s_])
                    if fused:      # (the GLU conv of 3 x 2 objects runs the heuristic's tile for ITS row count: same products, other order)
                        for i, (a, b) in enumerate(zip(rp, preds[s_])):
                            assert float((a == b).float().mean()) > 0.9995, 'math %d sequence %d frame %d' % (math, half * S + s_, i + 1)
                        for kk in rb:
                            assert float((rb[kk] - got[kk]).abs().max()) <= 1e-2 * float(rb[kk].abs().max()), (math, half, s_, kk)
                        continue
                    for i, (a, b) in enumerate(zip(rp, preds[s_])):
                        assert torch.equal(a, b), 'math %d sequence %d frame %d' % (math, half * S + s_, i + 1)
                    for kk in rb:
                        assert torch.equal(rb[kk], got[kk]), (math, half, s_, kk)


@pytest.mark.parametrize('batched_em', [False, True], ids=['em_per_sequence', 'em_batched'])
def test_lockstep_pool_equals_sequential_evaluation(lib, batched_em):
    """evaluator.LockstepPool: two lanes of two sequences in lock step.  Eight sequences -- five of one shape (two lock-step groups
    + one left over), one with another object count, one of another frame size and length, one of a single frame -- come back in input order with the
    index maps of evaluating them one after another with the plain loop (batch-invariant plans: bit for bit); the left-overs ran
    on the inner SequencePool, and a second run() re-binds the lanes' graphs."""
    cfg = O.make_cfg(**CFG_A)
    models = [H.make_model_and_sd(cfg, 5, DEV)[0] for _ in range(4)]
    models[0].book.fallback = 0x111
    seqs, seeds = [], [11, 12, 13, 14, 15, 16, 17, 18]
    for k_, (t, hh, ww, n) in enumerate(((7, 240, 432, 2), (7, 240, 432, 2), (7, 240, 432, 1), (7, 240, 432, 2), (7, 240, 432, 2),
                                         (5, 192, 320, 2), (7, 240, 432, 2), (1, 240, 432, 2))):      # (the last: ONE frame, nothing to segment)
        frames, m0 = synth_clip(t, hh, ww, n, 60 + k_)
        seqs.append((frames.to(DEV), m0.to(DEV), (hh, ww)))
    ref = []
    for (frames, m0, out), sd_ in zip(seqs, seeds):
        torch.manual_seed(sd_)
        with torch.no_grad():
            preds, _ = evaluator.evaluate_davis_seq(models[0], frames, [m0] + [None] * (frames.shape[1] - 1), out)
        ref.append([p.clone() for p in preds])
    pool = evaluator.LockstepPool(models, lockstep=2, lookahead=2, plans=None, batched_em=batched_em)
    got = pool.run(seqs, seeds=seeds)
    torch.cuda.synchronize()
    assert all(isinstance(g_, evaluator.LockstepGraph) and g_.batched_em == batched_em for g_ in pool.graphs)
    assert any(g_ is not None for g_ in pool.rest.graphs)
    # (EM and matching batched over a lane: the fusion conv with them, and its GLU launch picks the tile for ITS row count -- same
    # products, other fp32 order: a few tie pixels may flip, and the sequences' EM carries them on)
    same = (lambda a, b: torch.equal(a, b)) if not batched_em else (lambda a, b: float((a == b).float().mean()) > 0.999)
    for r, g_ in zip(ref, got):
        assert len(r) == len(g_)
        for a, b in zip(r, g_):
            assert same(a, b)
    got2 = pool.run(seqs[:5][::-1], seeds=seeds[:5][::-1])
    torch.cuda.synchronize()
    for r, g_ in zip(ref[:5][::-1], got2):
        assert len(r) == len(g_)
        for a, b in zip(r, g_):
            assert same(a, b)
    if batched_em:           # ... and deterministic: the second run() of the same sequences (re-bound graphs) repeats the first
        for g1, g2 in zip(got[:5][::-1], got2):
            for a, b in zip(g1, g2):
                assert torch.equal(a, b)
    with pytest.raises(ValueError):
        evaluator.LockstepPool(models[:3], lockstep=2)


def test_range_fault_falls_back_to_the_full_range_arithmetic(lib):
    """VERDICT r04 item 1 / ADVICE r04: the shipped default arithmetic (f16x3) has the fp16 range, the reference's fp32
    inference (networks.py:22-32) has none.  A model whose activations leave it -- here: the key encoder's stem scaled by 3e4,
    the kind of thing a real checkpoint may do -- must never return a silently wrong mask: (a) the raw path raises
    SwemRangeError at the sequence boundary; (b) the evaluator loops and SequencePool move the model's book to the full-range
    arithmetic (bf16x6 / fp32 kernels), warn, re-run, and return EXACTLY what a model that ran full-range from the start
    returns -- index maps within the usual fp32-vs-fp32 agreement of the exact fp32-MFMA run."""
    import warnings
    from swem_amd import synth
    cfg = O.make_cfg(**CFG_A)

    def make(fallback):
        model, sd = H.make_model_and_sd(cfg, 5, device=DEV)
        sd = dict(sd)
        sd['key_encoder.conv1.weight'] = sd['key_encoder.conv1.weight'] * 3.0e4
        model.load_state_dict(sd, strict=True)
        model.book.fallback = fallback
        return model
    frames, m0 = synth.make_clip(t=5, h=128, w=192, n_obj=2, seed=9)
    frames, m0 = frames.to(DEV), m0.to(DEV)
    masks = [m0] + [None] * 4
    ops.check_faults()
    # (a) the raw stages: finite outputs (ReLU epilogues everywhere), a set fault word, a raise where the host looks
    model = make(ops.MODEL_FALLBACK)
    with torch.no_grad():
        qk, qv, s16, s8, s4 = model('encode_key', frames[:, 0])
    with pytest.raises(ops.SwemRangeError):
        ops.check_faults()
    # (b) the evaluator loop: warning + re-run in the full-range arithmetic
    seeded = H.SeededInit(model, 3)
    with torch.no_grad(), pytest.warns(RuntimeWarning, match='full-range arithmetic'):
        preds, scores = evaluator.evaluate_davis_seq(seeded, frames, masks, (128, 192))
    assert model.book.full_range and (model.book.fallback >> 16) & 7 == 1
    ref_model = make(ops.MODEL_FALLBACK)
    ref_model.book.to_full_range()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter('error')                      # (no fault, no warning on a book that is full-range already)
        rpreds, rscores = evaluator.evaluate_davis_seq(H.SeededInit(ref_model, 3), frames, masks, (128, 192))
        # ... and later sequences on the downgraded model run straight through
        preds2, _ = evaluator.evaluate_davis_seq(seeded, frames, masks, (128, 192))
    for a, b, c in zip(preds, rpreds, preds2):
        assert torch.equal(a, b) and torch.equal(a, c)
    for a, b in zip(scores, rscores):
        assert torch.equal(a, b)
    exact = make(0)                                          # every GEMM on the fp32 MFMA
    with torch.no_grad():
        epreds, _ = evaluator.evaluate_davis_seq(H.SeededInit(exact, 3), frames, masks, (128, 192))
    agree = min(float((a == b).float().mean()) for a, b in zip(preds, epreds))
    assert agree >= 0.995, agree
    # the YouTube-VOS loop and the pool take the same way out
    m2 = make(ops.MODEL_FALLBACK)
    with torch.no_grad(), pytest.warns(RuntimeWarning, match='full-range arithmetic'):
        yp = evaluator.evaluate_ytvos_seq(H.SeededInit(m2, 3), frames, masks, (128, 192))
    for a, b in zip(yp, rpreds):
        assert torch.equal(a, b)
    m3 = make(ops.MODEL_FALLBACK)
    pool = evaluator.SequencePool([m3], use_graph=True, lookahead=2, plans=None)
    with pytest.warns(RuntimeWarning, match='full-range arithmetic'):
        got = pool.run([(frames, m0, (128, 192))], seeds=[3])
    assert m3.book.full_range
    m3.swem_core.init_on_host = True
    agree = min(float((a == b).float().mean()) for a, b in zip(got[0], rpreds))
    assert agree >= 0.995, agree                             # (graph replay batches the key encoder: not bit for bit)
    # ... and the lock-step pool (round 6): two sequences in one lane fault inside the captured lock-step graph, the pool moves its
    # shared book to the full-range arithmetic, drops the graphs and runs the call's sequences again
    m4 = [make(ops.MODEL_FALLBACK) for _ in range(2)]
    lpool = evaluator.LockstepPool(m4, lockstep=2, use_graph=True, lookahead=1, plans=None)
    with pytest.warns(RuntimeWarning, match='full-range arithmetic'):
        lgot = lpool.run([(frames, m0, (128, 192))] * 2, seeds=[3, 3])
    assert m4[0].book.full_range and m4[1].book is m4[0].book and isinstance(lpool.graphs[0], evaluator.LockstepGraph)
    for seq in lgot:
        assert len(seq) == len(rpreds)
        agree = min(float((a == b).float().mean()) for a, b in zip(seq, rpreds))
        assert agree >= 0.995, agree


def test_range_fault_fallback_against_the_oracle(lib):
    """the one scenario in which the product changes its arithmetic BY ITSELF -- a range fault moves the book
    to the full-range kernels -- held to the ORACLE (the reference's fp32 arithmetic on the same weights), not only to another HIP
    run.  The model: the first block of the key encoder hands conv2 an activation 2^17 times larger (bn1's gamma and beta x 2^17,
    conv2's filters x 2^-17): in fp32 that is the SAME function, exactly (powers of two), and as well conditioned as the
    unscaled model -- but the activation (up to ~4e6) does not fit an fp16 pair.  The HIP model faults, falls back by itself, and
    is then compared frame by frame, teacher-forced from the oracle's memory, at the north star's bars (teacher_forced_clip:
    logits 1e-3 outside saturation, probabilities, index maps with every differing pixel a near-tie, memorize vs float64)."""
    import warnings
    from swem_amd import synth
    from tests.test_gpu_parity import oracle_trajectory, teacher_forced_clip
    cfg = O.make_cfg(**CFG_A)
    model, sd = H.make_model_and_sd(cfg, 5, device=DEV)
    sd = dict(sd)
    for k, f in (('key_encoder.res2.0.bn1.weight', 2.0 ** 17), ('key_encoder.res2.0.bn1.bias', 2.0 ** 17),
                 ('key_encoder.res2.0.conv2.weight', 2.0 ** -17)):
        sd[k] = sd[k] * f
    model.load_state_dict(sd, strict=True)
    model.book.fallback = ops.MODEL_FALLBACK
    frames, m0 = synth.make_clip(t=4, h=128, w=192, n_obj=2, seed=9)
    out = (128, 192)
    ops.check_faults()
    with torch.no_grad(), pytest.warns(RuntimeWarning, match='full-range arithmetic'):
        evaluator.evaluate_davis_seq(H.SeededInit(model, 3), frames.to(DEV), [m0.to(DEV), None, None, None], out)
    assert model.book.full_range                      # this model took the automatic way out; from here on: against the oracle
    steps = oracle_trajectory(('range_fault', 5), O.Model(sd, cfg), frames, m0, out, seed=3)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        rows = teacher_forced_clip(model, steps, frames, out)
    ops.check_faults()
    H.record_parity('range_fault_fallback_vs_oracle', rows)
    ran = {}
    with torch.no_grad(), ops.flags(MATH_RAN=ran):
        model('encode_key', frames[:, 1].to(DEV))
    assert not ran.get(7) and not ran.get(3), ran       # (no fp16 / 16-bit operand layer is left on the fallen-back book)


def synth_clip(t, h, w, n, seed):
    from swem_amd import synth
    return synth.make_clip(t=t, h=h, w=w, n_obj=n, out_hw=(h, w), seed=seed)


def test_bench_gpus_2_starts_two_ranks_by_itself(lib):
    """`python bench.py --gpus 2` WITHOUT torchrun must come back with n_gpus = 2 from a real two-rank process group (the
    parent starts the ranks as child processes before it touches the GPU).  Rehearsed on this one-GPU box: both ranks share
    device 0 and the counter reduction runs over gloo (SWEM_DIST_BACKEND); on an 8-GPU node the same launch uses RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    env = dict(os.environ, SWEM_DIST_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2',
                          '--seqs', '1', '--no-autotune', '--no-cpu-baseline', '--no-em'], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    # (the communicator of this rehearsal is gloo: the line must say so, and must NOT claim RCCL ranks)
    assert line['n_gpus'] == 2 and line['gloo_ranks'] == 2 and 'rccl_ranks' not in line and line['scaling'] == 'weak'
    assert line['config']['frames_per_step'] == 1 and line['value'] > 0


def test_bench_gpus_8_rehearsal_on_one_gpu(lib):
    """The driver's 8-GPU launch, rehearsed on this one-GPU box (no multi-GPU box was ever available to
    this build): `python bench.py --gpus 8` starts eight ranks itself, they wrap onto device 0 and reduce their counters over
    gloo (SWEM_DIST_BACKEND).  Asserted: eight ranks in the process group, ONE JSON line (rank 0's), value = the frames of all
    eight ranks / the max-over-ranks time, every rank's torch CPU pool sized to its share of the container's CPU quota, the
    stream probe usable while eight processes load the host, and no RCCL claim on a gloo run."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    env = dict(os.environ, SWEM_DIST_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '4', '--warmup', '2',
                          '--regions', '3', '--seqs', '2', '--lookahead', '2', '--no-autotune', '--no-cpu-baseline', '--no-em'],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'rank 0 alone prints the line (got %d)' % len(lines)
    line = json.loads(lines[0])
    assert line['n_gpus'] == 8 and line['gloo_ranks'] == 8 and 'rccl_ranks' not in line and line['scaling'] == 'weak'
    assert line['config']['frames_per_step'] == 2 and line['timed_regions'] == 3 and line['value'] > 0
    assert line['value_min'] <= line['value'] <= line['value_max']
    # eight ranks share the container: each rank's torch pool is at most an eighth of the quota and at least one thread
    # (torchrun itself starts its ranks with OMP_NUM_THREADS=1: then it is one)
    from swem_amd import dist as sdist
    import torch as _t
    before = _t.get_num_threads()
    share = sdist.respect_cpu_quota(8)
    _t.set_num_threads(before)
    assert 1 <= line['config']['torch_cpu_threads_per_rank'] <= max(1, share)
    # whole-job frames per region: 8 ranks x 2 sequences x 4 steps
    assert abs(line['value'] * line['ms_per_step'] * 1e-3 * line['steps'] - 8 * 2 * 4) < 0.02 * 64
