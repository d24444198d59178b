"""GPU: the EM / matching kernels through the C ABI against (a) the golden vectors generated from the reference
and (b) the CPU oracle on fresh seeded inputs, per step (parity is asserted per step,
on live bases and on mass-weighted quantities, because low-mass bases are ratios of rounding noise in the
reference itself).  Full-size (config B) cases use size-independent properties of the algorithm.

Tolerances: 1e-4 relative per step for anything behind an exp((s - max)/tau): the logits s = x.kn are fp32 dot
products of magnitude |x| ~ 11 whose summation order differs between MKL and the MFMA chain (abs error ~1e-6),
and 1/tau = 20 turns that into ~3e-5 relative on z (measured 3.1e-5); 1e-5/1e-6 for the steps without an exp
(M step, zita, l2norm); matching <= 1e-4 absolute."""
import pytest
import torch

from oracle import swem_oracle as O
from swem_amd import ops
from swem_amd.modules import SWEMCore
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def d(t):
    return t.to(DEV).contiguous()


def relmax(a, b):
    return float((a.cpu() - b).abs().max() / b.abs().max())


def test_norm_bases(lib):
    k = torch.randn(4, 128, 64)
    kn = ops.em_norm_bases(d(k)).cpu()
    kn = kn.permute(0, 2, 1, 3).reshape(kn.shape[0], kn.shape[2], -1)      # (NK, C/4, L, 4) -> (NK, L, C)
    assert relmax(kn, O.l2norm(k, 1).transpose(1, 2)) < 1e-6


def test_single_steps_vs_golden(lib, golden):
    """swe_step / swm_step / sww_step (modules.py:93-127) with the reference's own inputs and outputs."""
    g = golden('g1_steps.npz')
    core = SWEMCore(n_bases=64, valdim=128, n_iters=4, tau=g['tau'], topl=64)
    x = g['x']
    xf = x.flatten(2)[:, None, None]
    x_t = xf.transpose(-2, -1).contiguous()
    z = core.swe_step(d(x_t), d(g['kappa']), d(g['masks']))
    assert relmax(z, g['z']) < 1e-4, 'E step'
    w = core.sww_step(d(g['kappa']), d(x_t), d(g['masks']))
    assert relmax(w, g['weights']) < 1e-4, 'W step'
    kap, zita = core.swm_step(d(g['z']), d(xf), d(g['kappa_prev']), d(g['zita_prev']))
    assert relmax(zita, g['zita_m']) < 1e-6, 'M step zita'
    assert relmax(kap, g['kappa_m']) < 1e-5, 'M step kappa'
    # value update (modules.py:164-165) = the same M kernel with the value map as A
    P = x_t.shape[-2]
    zp = torch.zeros(2, ops.em_pad(P), 128)
    zp[:, :P] = g['z'].reshape(2, 2, P, 64).permute(0, 2, 1, 3).reshape(2, P, 128)     # z[n][p][cls*L + l]
    vp = g['v'].flatten(3)[0].transpose(1, 2).contiguous()                          # (N, P, V) pixel-major
    nu, _, _ = ops.em_mstep(d(vp), True, d(zp), d(g['nu_prev'].reshape(4, 128, 64)), d(g['zita_prev'].reshape(4, 64)), P)
    assert relmax(nu.view(1, 2, 2, 128, 64), g['nu']) < 1e-5, 'nu update'


def _check_bases(got, ref, zita_ref, tag, tol=1e-4):
    """live bases (zita >= 1e-3) element-wise, every base mass-weighted."""
    zr = zita_ref.squeeze(-2)                                  # (1,N,2,L)
    live = (zr >= 1e-3)
    for name in ('kappa', 'nu'):
        a, b = got[name].cpu(), ref[name]
        mass_err = ((a - b) * zr.unsqueeze(-2)).abs().max() / (b * zr.unsqueeze(-2)).abs().max()
        assert mass_err < tol, '%s %s mass-weighted rel err %.3g' % (tag, name, mass_err)
        lm = live.unsqueeze(-2).expand_as(b)
        live_err = (a - b)[lm].abs().max() / b[lm].abs().max()
        assert live_err < 5 * tol, '%s %s live-base rel err %.3g' % (tag, name, live_err)
    assert relmax(got['zita'], ref['zita']) < tol, tag + ' zita'
    return float(live.float().mean())


def test_memorize_two_frames_vs_golden(lib, golden):
    """SWEMCore.memorize (modules.py:183-193) over two frames from the reference's captured random init."""
    g = golden('g2_memorize.npz')
    core = SWEMCore(n_bases=64, valdim=128, n_iters=4, tau=0.05, topl=64)
    init = {'kappa': d(g['init_kappa']), 'nu': d(g['init_nu']), 'zita': d(g['init_zita'])}
    b0 = core.swem(d(g['x0']), d(g['v0']), d(g['m0']), init)
    frac = _check_bases(b0, {k: g[k + '0'] for k in ('kappa', 'nu', 'zita')}, g['zita0'], 'frame 0')
    # frame 1 from the REFERENCE's frame-0 bases (per-step parity: identical inputs)
    ref0 = {k: d(g[k + '0']) for k in ('kappa', 'nu', 'zita')}
    b1 = core.swem(d(g['x1']), d(g['v1']), d(g['m1']), ref0)
    _check_bases(b1, {k: g[k + '1'] for k in ('kappa', 'nu', 'zita')}, g['zita1'], 'frame 1')
    assert 0.05 < frac <= 1.0


def test_matching_vs_golden(lib, golden):
    """get_affinity + perm_inv_feat (modules.py:198-208,232-276) from the reference's bases, Lm = 64 and 128."""
    g, m = golden('g2_memorize.npz'), golden('g3_matching.npz')
    first = [d(g['kappa0'][0]), d(g['nu0'][0])]
    upd = [d(g['kappa1'][0]), d(g['nu1'][0])]
    qk = m['qk']
    xp = d(qk[0].flatten(1).t())                                      # (P, C)
    h, w = qk.shape[-2:]
    for tag, banks, Sref, memref in (('Lm=64', (first[0], first[1], None, None), m['S1'], m['mem1']),
                                     ('Lm=128', (first[0], first[1], upd[0], upd[1]), m['S2'], m['mem2'])):
        mem, S = ops.match(xp, *banks, 64, 0.05)
        mem = mem.reshape(2, h, w, -1).permute(0, 3, 1, 2).cpu()
        S = S.view(2, h, w, -1).permute(0, 3, 1, 2).cpu()
        e_mem = float((mem - memref[0]).abs().max())
        e_S = float((S - Sref).abs().max())
        assert e_mem < 1e-4 * max(1.0, float(memref.abs().max())), '%s mem_out abs err %.3g' % (tag, e_mem)
        assert e_S < 1e-4, '%s S abs err %.3g' % (tag, e_S)


def test_integration_md_ctypes_binding_runs_as_printed(lib, golden):
    """INTEGRATION.md section 2 shows the reference-side binding a maintainer would paste into methods/SWEM/modules.py
    (`matching_features`, in place of get_affinity + perm_inv_feat, modules.py:198-208,232-276).  The code block is taken from
    the document VERBATIM and executed -- only the library path is made absolute -- on a core holding the reference's banks
    (fixtures g2 / g3), with one bank and with two, against the reference's recorded S / mem_out and against the oracle."""
    import os
    import re
    import types
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    md = open(os.path.join(root, 'INTEGRATION.md')).read()
    sec = md[md.index('## 2. Binding the C ABI directly'):md.index('## 3.')]
    code = re.search(r'```python\n(.*?)```', sec, flags=re.S).group(1)
    assert 'def matching_features(core, qk)' in code and "C.CDLL('swem_amd/libswem_hip.so')" in code
    ns = {}
    exec(compile(code.replace("'swem_amd/libswem_hip.so'", repr(os.path.join(root, 'swem_amd', 'libswem_hip.so'))),
                 'INTEGRATION.md#2', 'exec'), ns)
    g, m = golden('g2_memorize.npz'), golden('g3_matching.npz')
    qk = d(m['qk'])                                                   # (1, C, h, w) as modules.py:278-283 passes it
    h, w = qk.shape[-2:]
    bank = lambda i: {'kappa': d(g['kappa%d' % i]), 'nu': d(g['nu%d' % i])}
    for tag, upd, Sref, memref in (('Lm=64', None, m['S1'], m['mem1']), ('Lm=128', bank(1), m['S2'], m['mem2'])):
        core = types.SimpleNamespace(memories={'first': types.SimpleNamespace(bases=bank(0)),
                                               'update': types.SimpleNamespace(bases=upd)}, topl=64, tau=0.05)
        S, mem = ns['matching_features'](core, qk)
        torch.cuda.synchronize()
        mem = mem.reshape(2, h, w, -1).permute(0, 3, 1, 2).cpu()
        S = S.view(2, h, w, -1).permute(0, 3, 1, 2).cpu()
        assert float((mem - memref[0]).abs().max()) < 1e-4 * max(1.0, float(memref.abs().max())), tag
        assert float((S - Sref).abs().max()) < 1e-4, tag
        # ... and the oracle's get_affinity on the same banks (oracle/swem_oracle.py, bit-identical to the reference)
        oc = O.Core(g['kappa0'].shape[-1], g['nu0'].shape[-2], 4, 0.05, 64)
        oc.first.update({'kappa': g['kappa0'], 'nu': g['nu0'], 'zita': g['zita0']})
        if upd is not None:
            oc.upd.update({'kappa': g['kappa1'], 'nu': g['nu1'], 'zita': g['zita1']})
        omem, _, oS, _ = oc.match_features(m['qk'], torch.zeros(1, g['nu0'].shape[-2], h, w))
        assert float((mem - omem).abs().max()) < 1e-4 * max(1.0, float(omem.abs().max())), tag
        assert float((S - oS).abs().max()) < 1e-4, tag


@pytest.mark.parametrize('L,T', [(64, 4), (128, 3), (256, 5)])
def test_memorize_and_match_vs_oracle(lib, L, T):
    """Fresh seeded inputs, oracle computed on the CPU at test time; covers every template instance (L = 64/128/256)."""
    g = torch.Generator().manual_seed(100 + L)
    h, w, C, V, N = 12, 20, 128, 128, 2
    x0, v0, m0 = H.em_inputs(h, w, C, V, N, g)
    x1, v1, m1 = H.em_inputs(h, w, C, V, N, g)
    torch.manual_seed(L)
    init = dict(zip(('kappa', 'nu', 'zita'), O.random_init((1, N, 2, C, L), V)))
    o0 = O.swem(x0, v0, m0, init, L, T, 0.05, V)
    o1 = O.swem(x1, v1, m1, o0, L, T, 0.05, V)
    core = SWEMCore(n_bases=L, valdim=V, n_iters=T, tau=0.05, topl=64)
    b0 = core.swem(d(x0), d(v0), d(m0), {k: d(t) for k, t in init.items()})
    _check_bases(b0, o0, o0['zita'], 'L=%d frame 0' % L)
    b1 = core.swem(d(x1), d(v1), d(m1), {k: d(t) for k, t in o0.items()})
    _check_bases(b1, o1, o1['zita'], 'L=%d frame 1' % L)
    # matching on the ORACLE's banks
    ocore = O.Core(L, V, T, 0.05, 64)
    ocore.first.update(o0)
    ocore.upd.update(o1)
    qx, _ = H.structured_keys(h * w, C, 6, g)
    qk = qx.t().reshape(1, C, h, w).contiguous()
    omem, _, oS, _ = ocore.match_features(qk, torch.zeros(1, V, h, w))
    mem, S = ops.match(d(qx), d(o0['kappa'][0]), d(o0['nu'][0]), d(o1['kappa'][0]), d(o1['nu'][0]), 64, 0.05)
    mem = mem.reshape(N, h, w, V).permute(0, 3, 1, 2).cpu()
    S = S.view(N, h, w, -1).permute(0, 3, 1, 2).cpu()
    assert float((mem - omem).abs().max()) < 1e-4 * max(1.0, float(omem.abs().max())), 'mem_out'
    assert float((S - oS).abs().max()) < 1e-4, 'S'


def test_config_b_size_properties(lib):
    """BASELINE config B sizes (P = 30*54, C = 128, V = 512, K = 256, T = 5, N = 3): properties that hold for any
    input.  (1) responsibility mass is conserved: sum_l (zita - zita_prev) = sum_p weights of the last iteration,
    where weights <= masks; (2) kappa and nu are convex combinations of the prior and the data;
    (3) S features lie in [0,1] with feat + complement = 1, non-negative exp-sums; (4) mem_out lies in the
    convex hull of the value bases; (5) the fused memorize equals the step-by-step composition on the device."""
    g = torch.Generator().manual_seed(7)
    h, w, C, V, N, L, T = 30, 54, 128, 512, 3, 256, 5
    P = h * w
    x, v, m = H.em_inputs(h, w, C, V, N, g)
    torch.manual_seed(1)
    kap0, nu0, z0 = O.random_init((1, N, 2, C, L), V)
    core = SWEMCore(n_bases=L, valdim=V, n_iters=T, tau=0.05, topl=64)
    init = {'kappa': d(kap0), 'nu': d(nu0), 'zita': d(z0)}
    b = core.swem(d(x), d(v), d(m), init)
    zita = b['zita'].cpu()
    gained = (zita - z0).sum(-1).flatten()                                  # per (n, class)
    msum = m.flatten(3).sum(-1).flatten()
    assert (gained <= msum * (1 + 1e-5) + 1e-3).all() and (gained >= 0).all()
    assert torch.isfinite(b['kappa']).all() and torch.isfinite(b['nu']).all()
    # (2) with zita_prev = 1e-6 the new bases are (almost) weighted means of the pixels: inside the data range
    kap = b['kappa'].cpu()
    xmin, xmax = x.flatten(2).min(-1)[0].view(1, 1, 1, C, 1), x.flatten(2).max(-1)[0].view(1, 1, 1, C, 1)
    live = (zita > 1e-3).expand_as(kap)
    assert ((kap >= xmin - 1e-3) & (kap <= xmax + 1e-3))[live].all()
    # (5) composition of the single-step kernels reproduces the fused driver bit for bit (same kernels, same order)
    xf = d(x.flatten(2)[:, None, None])
    x_t = xf.transpose(-2, -1).contiguous()
    mk = d(m.flatten(3).unsqueeze(-1))
    kappa, weights = init['kappa'], mk
    for it in range(T):
        z = core.swe_step(x_t, kappa, weights)
        kappa, zt = core.swm_step(z, xf, init['kappa'], init['zita'])
        if it < T - 1:
            weights = core.sww_step(kappa, x_t, mk)
    assert torch.equal(kappa, b['kappa']) and torch.equal(zt, b['zita'])
    # frame 2 + matching at Lm = 512
    x2, v2, m2 = H.em_inputs(h, w, C, V, N, g)
    b2 = core.swem(d(x2), d(v2), d(m2), b)
    qx, _ = H.structured_keys(P, C, 6, g)
    mem, S = ops.match(d(qx), b['kappa'][0], b['nu'][0], b2['kappa'][0], b2['nu'][0], 64, 0.05)
    S = S.cpu()
    assert torch.isfinite(S).all() and (S >= 0).all() and (S <= 1).all()
    assert float((S[..., :64] + S[..., 64:] - 1).abs().max()) < 1e-6
    allnu = torch.cat([b['nu'][0], b2['nu'][0]], -1).cpu()                  # (N,2,V,2L)
    lo = allnu.permute(0, 2, 1, 3).flatten(2).min(-1)[0]                    # (N,V)
    hi = allnu.permute(0, 2, 1, 3).flatten(2).max(-1)[0]
    mem = mem.cpu()                                                         # (N,P,V)
    tol = 1e-4 * float(allnu.abs().max())
    assert (mem >= lo[:, None] - tol).all() and (mem <= hi[:, None] + tol).all()


@pytest.mark.parametrize('L', [64, 256])
def test_clips_forms_equal_the_single_clip_calls(lib, L):
    """swem_memorize_packed_clips_f32 / swem_match_packed_clips_f32 (round 6: the objects of several clips in ONE call, a key map per
    clip -- modules.py:129-168 / 232-293 with B = clips; what a lock-step lane runs for its sequences): new bases, the pack they
    keep current (keys, values, fp16 value planes) and matching's outputs are BIT-IDENTICAL to one packed call per clip, over two
    frames with the second reading its prior's normalised keys from the pack; both readouts (fp32 and pre-split f16x3)."""
    g = torch.Generator().manual_seed(500 + L)
    h, w, C, V, N, T, S = 12, 20, 128, 128, 2, 3, 3
    P = h * w
    pm = lambda x: d(x[0].flatten(1).t())                                   # (P, C)
    pv = lambda v: d(v[0].flatten(2).transpose(1, 2))                        # (N, P, V)
    pk = lambda m: d(m[0].flatten(2))                                        # (N, 2, P)
    frames = [[H.em_inputs(h, w, C, V, N, g) for _ in range(S)] for _ in range(2)]
    torch.manual_seed(L)
    prior = []
    for _ in range(S):
        kap, nu, zita = [d(t[0]) for t in O.random_init((1, N, 2, C, L), V)]
        prior.append((kap, nu, zita[:, :, 0].contiguous()))
    book = ops.PlanBook(fallback=ops.MODEL_FALLBACK)
    qx = [d(H.structured_keys(P, C, 6, g)[0]) for _ in range(S)]
    with ops.use_book(book):
        # one call per clip
        packs = [ops.new_pack(N, C, V, L, DEV) for _ in range(S)]
        ref = []
        for s_ in range(S):
            x0, v0, m0 = frames[0][s_]
            x1, v1, m1 = frames[1][s_]
            b0 = ops.memorize(pm(x0), pv(v0), pk(m0), *prior[s_], T, 0.05, pack=packs[s_], prior_packed=False, bank=0)
            ops.pack_bank(b0[0], b0[1], packs[s_], 1)
            b1 = ops.memorize(pm(x1), pv(v1), pk(m1), *b0, T, 0.05, pack=packs[s_], prior_packed=True, bank=1)
            ref.append((b0, b1))
        # all clips in one call: the objects back to back, the packs as ONE pack
        cat = lambda ts: torch.cat(list(ts)).contiguous()
        pack_all = ops.new_pack(S * N, C, V, L, DEV)
        xs = [torch.stack([pm(frames[f][s_][0]) for s_ in range(S)]).contiguous() for f in (0, 1)]
        vs = [cat(pv(frames[f][s_][1]) for s_ in range(S)) for f in (0, 1)]
        ms = [cat(pk(frames[f][s_][2]) for s_ in range(S)) for f in (0, 1)]
        pr = [cat(prior[s_][i] for s_ in range(S)) for i in range(3)]
        a0 = ops.memorize(xs[0], vs[0], ms[0], *pr, T, 0.05, pack=pack_all, prior_packed=False, bank=0, clips=S)
        ops.pack_bank(a0[0], a0[1], pack_all, 1)
        a1 = ops.memorize(xs[1], vs[1], ms[1], *a0, T, 0.05, pack=pack_all, prior_packed=True, bank=1, clips=S)
        for s_ in range(S):
            for i in range(3):
                assert torch.equal(a0[i][s_ * N:(s_ + 1) * N], ref[s_][0][i]), (s_, i)
                assert torch.equal(a1[i][s_ * N:(s_ + 1) * N], ref[s_][1][i]), (s_, i)
            assert torch.equal(pack_all[0][2 * N * s_:2 * N * (s_ + 1)], packs[s_][0])
            assert torch.equal(pack_all[1][N * s_:N * (s_ + 1)], packs[s_][1])
            assert torch.equal(pack_all[2][N * s_:N * (s_ + 1)].view(torch.int16), packs[s_][2].view(torch.int16))
        assert pack_all[2].any()
        qall = torch.stack(qx).contiguous()
        for plan in (0, 2 | 2 << 4 | 1 << 8 | 3 << 16):
            try:
                if plan:
                    ops._MATCH_PLANS[(N, C, V, P, L, 2)] = plan
                    ops._MATCH_PLANS[(S * N, C, V, P, L, 2)] = plan
                mem_a, S_a = ops.match_packed(qall, pack_all, L, 64, 0.05, hw=(h, w), clips=S)
                for s_ in range(S):
                    mem_r, S_r = ops.match_packed(qx[s_], packs[s_], L, 64, 0.05, hw=(h, w))
                    assert torch.equal(mem_a[s_ * N:(s_ + 1) * N], mem_r) and torch.equal(S_a[s_ * N:(s_ + 1) * N], S_r), (plan, s_)
            finally:
                ops._MATCH_PLANS.pop((N, C, V, P, L, 2), None)
                ops._MATCH_PLANS.pop((S * N, C, V, P, L, 2), None)
    with pytest.raises(Exception):
        ops.memorize(xs[0][:2], vs[0], ms[0], *pr, T, 0.05, pack=pack_all, clips=S)


@pytest.mark.parametrize('L', [64, 256])
def test_packed_banks_equal_the_per_frame_packing(lib, L):
    """swem_memorize_packed_f32 / swem_match_packed_f32 (the persistent packed banks SWEMCore keeps: no per-frame `cat` +
    2 x `l2norm` of both banks, modules.py:282-283, 295-306) give BIT-IDENTICAL results to swem_memorize_f32 /
    swem_match_f32, which normalise and pack inside the call -- over two frames, with the second memorize reading its
    prior's normalised form from the pack the first one wrote."""
    g = torch.Generator().manual_seed(300 + L)
    h, w, C, V, N, T = 12, 20, 128, 128, 2, 3
    P = h * w
    x0, v0, m0 = H.em_inputs(h, w, C, V, N, g)
    x1, v1, m1 = H.em_inputs(h, w, C, V, N, g)
    x2, v2, m2 = H.em_inputs(h, w, C, V, N, g)
    pm = lambda x: d(x[0].flatten(1).t())                                   # (P, C)
    pv = lambda v: d(v[0].flatten(2).transpose(1, 2))                        # (N, P, V)
    pk = lambda m: d(m[0].flatten(2))                                        # (N, 2, P)
    torch.manual_seed(L)
    kap, nu, zita = [d(t[0]) for t in O.random_init((1, N, 2, C, L), V)]
    zita = zita[:, :, 0].contiguous()
    # frame 0 -> 'first' bank, frame 1 -> 'update' bank, frame 2 -> 'update' again (prior read from the pack)
    ref0 = ops.memorize(pm(x0), pv(v0), pk(m0), kap, nu, zita, T, 0.05)
    ref1 = ops.memorize(pm(x1), pv(v1), pk(m1), *ref0, T, 0.05)
    ref2 = ops.memorize(pm(x2), pv(v2), pk(m2), *ref1, T, 0.05)
    pack = ops.new_pack(N, C, V, L, DEV)
    # (the pack's fp16 value planes are kept only under a book whose readout may read them -- ops.value_planes_wanted: a
    # model's default book does, the free-standing default book of plain ops.* calls, exact fp32 kernels, does not)
    f16_book = ops.PlanBook(fallback=ops.MODEL_FALLBACK)
    with ops.use_book(f16_book):
        got0 = ops.memorize(pm(x0), pv(v0), pk(m0), kap, nu, zita, T, 0.05, pack=pack, prior_packed=False, bank=0)
        got1 = ops.memorize(pm(x1), pv(v1), pk(m1), *got0, T, 0.05, pack=pack, prior_packed=False, bank=1)
        got2 = ops.memorize(pm(x2), pv(v2), pk(m2), *got1, T, 0.05, pack=pack, prior_packed=True, bank=1)
    pack_nop = ops.new_pack(N, C, V, L, DEV)
    ops.memorize(pm(x0), pv(v0), pk(m0), kap, nu, zita, T, 0.05, pack=pack_nop, prior_packed=False, bank=0)
    assert not pack_nop[2].any() and torch.equal(pack_nop[1][:, :, :L], pack[1][:, :, :L])     # planes left out, values kept
    for a, b in zip(ref0 + ref1 + ref2, got0 + got1 + got2):
        assert torch.equal(a, b)
    qx, _ = H.structured_keys(P, C, 6, g)
    mem_r, S_r = ops.match(d(qx), ref0[0], ref0[1], ref2[0], ref2[1], 64, 0.05)
    mem_p, S_p = ops.match_packed(d(qx), pack, L, 64, 0.05)
    assert torch.equal(mem_r, mem_p) and torch.equal(S_r, S_p)
    # a pack rebuilt from the bases (SWEMCore.repack) is the pack memorize kept
    pack2 = ops.new_pack(N, C, V, L, DEV)
    with ops.use_book(f16_book):
        ops.pack_bank(ref0[0], ref0[1], pack2, 0)
        ops.pack_bank(ref2[0], ref2[1], pack2, 1)
    assert torch.equal(pack2[0], pack[0]) and torch.equal(pack2[1], pack[1])
    # ... including the values' fp16 pair (hi, mid) the pre-split readout GEMM reads: hi + mid = nu to 22 significant bits (or to
    # 2^-25 absolute where mid is subnormal)
    assert pack[2].dtype == torch.float16 and torch.equal(pack2[2].view(torch.int16), pack[2].view(torch.int16))
    mvq = pack[2].double()                                                  # (N, 2, 4L/8, V, 8)
    back = (mvq[:, 0] + mvq[:, 1]).permute(0, 2, 1, 3).reshape(N, V, 4 * L)     # [n][v][k]
    assert float((back - pack[1].double()).abs().max()) <= 2.0 ** -22 * float(pack[1].abs().max()) + 2.0 ** -25
    # readout on the pre-split planes (plan math field 3: three f16 products) against the fp32 readout
    key = (N, C, V, P, L, 2)
    try:
        ops._MATCH_PLANS[key] = 2 | 2 << 4 | 1 << 8 | 3 << 16
        mem_q, S_q = ops.match_packed(d(qx), pack, L, 64, 0.05)
        # ... and, as NHWC images, with their own bf16 planes once the fusion conv has asked for them (swem_match_packed_f32_planes):
        # same outputs, planes bit-identical to swem_split_bf16x3_f32 (mem_out's pixel axis is the PADDED one)
        mem_i, S_i = ops.match_packed(d(qx), pack, L, 64, 0.05, hw=(h, w))
        assert torch.equal(mem_i.flatten(1, 2), mem_q) and torch.equal(S_i.flatten(1, 2), S_q)
        assert '_swem_split' not in mem_i.__dict__ and '_swem_split' not in S_i.__dict__
        F16 = ops.PLANES_F16
        for want_m, want_s in ((2, 3), (F16, F16), (3, F16)):
            ops.SPLIT_HINTS[mem_i._swem_site] = {False: want_m}
            ops.SPLIT_HINTS[S_i._swem_site] = {False: want_s}
            mem_j, S_j = ops.match_packed(d(qx), pack, L, 64, 0.05, hw=(h, w))
            assert torch.equal(mem_j, mem_i) and torch.equal(S_j, S_i)
            Pm = mem_j.stride(0) // V
            full = torch.as_strided(mem_j, (N, Pm, V), (Pm * V, V, 1))
            for img, flat, npix, Cc, npl in ((mem_j, full, N * Pm, V, want_m), (S_j, S_j, N * P, 128, want_s)):
                planes, n_ = img.__dict__['_swem_split'][ops._pkey(False, npl)]
                assert n_ == npl and ops.presplit(img, False, npl) is planes
                if npl == F16:
                    sp = torch.empty((2, npix * Cc), dtype=torch.float16, device=DEV)
                    __import__('swem_amd')._lib.call('swem_split_f16x2_f32', ops._stream(), flat.data_ptr(), sp.data_ptr(), npix, Cc, 0, 0)
                    assert torch.equal(planes.view(torch.int16), sp.view(torch.int16))
                else:
                    sp = torch.empty((3, npix * Cc), dtype=torch.bfloat16, device=DEV)
                    __import__('swem_amd')._lib.call('swem_split_bf16x3_f32', ops._stream(), flat.data_ptr(), sp.data_ptr(), npix, Cc, 0)
                    assert torch.equal(planes[:npl].view(torch.int16), sp[:npl].view(torch.int16))
    finally:
        ops._MATCH_PLANS.pop(key, None)
        ops.SPLIT_HINTS.clear()
    assert torch.equal(S_q, S_p)
    err = float((mem_q - mem_p).abs().max()) / float(mem_p.abs().max())
    print('pre-split (f16x3) readout vs fp32 readout: rel %.3g' % err)
    assert 0 < err < 1e-6         # (round 3, bf16 planes: 3e-6)


def test_memorize_in_two_calls_equals_the_one_call_form(lib):
    """swem_memorize_packed_keys_f32 + swem_memorize_packed_values_f32 (the part of memorize that does not read the value map,
    run beside the value encoder by evaluator.frame_chain; then the value update) launch the same blocks on the same data as
    swem_memorize_packed_f32: bases and pack are bit-identical, also with the value update on another stream."""
    g = torch.Generator().manual_seed(31)
    N, C, V, P, L, T = 2, 128, 512, 405, 64, 4
    d = lambda t: t.to(DEV)
    x = d(torch.randn(P, C, generator=g))
    v = d(torch.randn(N, P, V, generator=g))
    m = d(torch.rand(N, 2, P, generator=g))
    kp = d(torch.nn.functional.normalize(torch.randn(N, 2, C, L, generator=g), dim=2))
    nup = d(torch.randn(N, 2, V, L, generator=g))
    zp = d(torch.rand(N, 2, L, generator=g) * 3 + 0.1)
    pack_a, pack_b = ops.new_pack(N, C, V, L, DEV), ops.new_pack(N, C, V, L, DEV)
    for pk in (pack_a, pack_b):
        ops.pack_bank(kp, nup, pk, 0)
        ops.pack_bank(kp, nup, pk, 1)
    ref = ops.memorize(x, v, m, kp, nup, zp, T, 0.05, pack=pack_a, prior_packed=True, bank=1)
    kappa, zita, z = ops.memorize_keys(x, m, kp, zp, T, 0.05, pack_b, prior_packed=True, bank=1)
    side = ops.new_stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        nu = ops.memorize_values(v, z, nup, zp, pack_b, bank=1)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(kappa, ref[0]) and torch.equal(nu, ref[1]) and torch.equal(zita, ref[2])
    for a, b in zip(pack_a, pack_b):
        assert torch.equal(a.view(torch.int16) if a.dtype == torch.float16 else a, b.view(torch.int16) if b.dtype == torch.float16 else b)


def test_round3_stream_k_readout_plan_still_runs(lib):
    """the pre-split readout now always runs f16x3, whose kernels have no stream-K form; a readout plan tuned or saved
    in round 3 with bf16x3 + stream-K (plan bits 24-27 == 1: the r03 tuner offered them) must run the plain grid of its tile, not
    fail at launch -- and give the fp32 readout's result to the f16x3 readout's 1e-6."""
    g = torch.Generator().manual_seed(7)
    h, w, C, V, N, L = 12, 20, 128, 128, 2, 64
    P = h * w
    kap = d(torch.nn.functional.normalize(torch.randn(N, 2, C, L, generator=g), dim=2))
    nu = d(torch.randn(N, 2, V, L, generator=g))
    qx, _ = H.structured_keys(P, C, 6, g)
    key = (N, C, V, P, L, 2)
    ref_mem, ref_S = ops.match(d(qx), kap, nu, kap, nu, 64, 0.05)          # default book: fp32 readout
    with ops.use_book(ops.PlanBook(fallback=ops.MODEL_FALLBACK)) as book:
        pack = ops.new_pack(N, C, V, L, DEV)
        ops.pack_bank(kap, nu, pack, 0)
        ops.pack_bank(kap, nu, pack, 1)
        for plan in (2 | 2 << 4 | 1 << 8 | 3 << 16 | 6 << 20 | 1 << 24,      # r03: 128x128 tile, 8 waves, bf16x3, stream-K
                     2 | 2 << 4 | 1 << 8 | 3 << 16 | 5 << 20 | 1 << 24):     # ... prefetched fragments + stream-K
            book.match[key] = plan
            mem, S = ops.match_packed(d(qx), pack, L, 64, 0.05)
            ops.check_faults()
            assert torch.equal(S, ref_S)
            assert float((mem - ref_mem).abs().max()) <= 2e-6 * float(ref_mem.abs().max())


def test_shipped_plans_are_keyed_by_device(lib):
    """SequencePool loads the shipped plan file only on the architecture the file names: here, on the gfx950 box,
    it loads; a copy that names another architecture does not."""
    import json
    import os
    import tempfile
    dev = torch.device(DEV)
    assert ops.device_arch(dev) == 'gfx950'
    b = ops.PlanBook(fallback=ops.MODEL_FALLBACK)
    assert b.load(ops.shipped_plans(), device=dev) is b and b.conv
    dd = json.load(open(ops.shipped_plans()))
    with tempfile.TemporaryDirectory() as tmp:
        other = os.path.join(tmp, 'other.json')
        json.dump(dict(dd, device='gfx942'), open(other, 'w'))
        b2 = ops.PlanBook(fallback=ops.MODEL_FALLBACK)
        assert b2.load(other, device=dev) is False and not b2.conv
