"""CPU: host-side logic of the drop-in boundary (state dict, bank policy, determinism, loud failure)."""
import json
import os

import pytest
import torch

from oracle import swem_oracle as O
from swem_amd import synth, weights
from swem_amd.modules import MemoryBank, SWEMCore
from swem_amd.swem import SWEM

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('tag,kw', [('resnet50_mo', dict(BACKBONE='resnet50')),
                                    ('resnet18_so', dict(BACKBONE='resnet18', SINGLE_OBJ=True, NUM_BASES=64))])
def test_state_dict_keys_match_reference(tag, kw):
    ref = json.load(open(os.path.join(GOLDEN, 'g0_state_dict_keys.json')))[tag]
    sd = SWEM(O.make_cfg(**kw)).state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == ref


def test_weights_and_clip_are_deterministic():
    m = SWEM(O.make_cfg(BACKBONE='resnet18', NUM_BASES=64))
    a = weights.fill_state_dict(m.state_dict(), seed=3, backbone='resnet18')
    b = weights.fill_state_dict(dict(reversed(list(m.state_dict().items()))), seed=3, backbone='resnet18')
    assert all(torch.equal(a[k], b[k]) for k in a)
    f1, m1 = synth.make_clip(t=2, h=64, w=96, n_obj=2, seed=5)
    f2, m2 = synth.make_clip(t=2, h=64, w=96, n_obj=2, seed=5)
    assert torch.equal(f1, f2) and torch.equal(m1, m2)
    assert m1.sum(1).eq(1).all() and f1.min() >= 0 and f1.max() <= 1


def _bases(n, tag):
    return {'kappa': torch.full((1, n, 2, 4, 8), tag), 'nu': torch.full((1, n, 2, 6, 8), tag),
            'zita': torch.full((1, n, 2, 1, 8), tag)}


def test_memory_bank_policy_matches_oracle():
    """modules.py:29-60,183-193: 'first' only appends unseen object ids, 'update' is replaced."""
    first, upd = MemoryBank('fixed'), MemoryBank('updated')
    ofirst, oupd = O.MemoryBank(True), O.MemoryBank(False)
    for step, (n, tag) in enumerate([(2, 1.0), (2, 2.0), (3, 3.0), (3, 4.0)]):
        b = _bases(n, tag)
        had = first.bases is not None
        first.update(b)
        ofirst.update(_bases(n, tag))
        if had:
            upd.update(b)
            oupd.update(_bases(n, tag))
        for k in b:
            assert torch.equal(first.bases[k], ofirst.bases[k])
            if had:
                assert torch.equal(upd.bases[k], oupd.bases[k])
    assert first.bases['kappa'].shape[1] == 3 and first.n_objs == 3
    assert first.bases['kappa'][0, 0, 0, 0, 0] == 1.0 and first.bases['kappa'][0, 2, 0, 0, 0] == 3.0
    with pytest.raises(AssertionError):
        MemoryBank('slow')


def test_core_attributes_and_get_mem():
    core = SWEMCore(n_bases=32, valdim=64, n_iters=3, tau=0.1, topl=64)
    assert (core.n_bases, core.n_iters, core.tau, core.topl, core.p_drop) == (32, 3, 0.1, 32, 0.0)
    assert core.fusion_layer.layer_f.weight.shape == (64, 2 * 64 + 2 * 32, 3, 3)
    assert set(core.memories) == {'first', 'update'}
    core.memories['first'].update(_bases(2, 1.0))
    core.memories['update'].update(_bases(2, 2.0))
    k, v = core.get_mem()
    assert k.shape == (1, 2, 2, 4, 16) and v.shape == (1, 2, 2, 6, 16)
    core.empty()
    assert core.memories['first'].bases is None and core.memories['update'].bases is None


def test_random_init_matches_reference_stream():
    """modules.py:170-178: same seed -> same bases as the oracle's restatement (host generator)."""
    core = SWEMCore(n_bases=16, valdim=8)
    core.init_on_host = True
    torch.manual_seed(9)
    k, nu, z = core.random_init(size=(1, 2, 2, 12, 16), dtype=torch.float32, device=torch.device('cpu'))
    torch.manual_seed(9)
    ok, onu, oz = O.random_init((1, 2, 2, 12, 16), 8)
    assert torch.equal(k, ok) and torch.equal(nu, onu) and torch.equal(z, oz)


def test_cpu_model_fails_loudly():
    m = SWEM(O.make_cfg(BACKBONE='resnet18', NUM_BASES=64)).eval()
    with pytest.raises(RuntimeError, match='HIP device only'):
        m('encode_key', torch.zeros(1, 3, 32, 32))
    with pytest.raises(NotImplementedError):
        m('nonsense')
    with pytest.raises(KeyError):
        SWEM(O.make_cfg(BACKBONE='resnet101'))


def test_checkpoint_so_to_mo_surgery(tmp_path):
    """basic_evaluator.py:104-124: a 4-channel (single-object) value-encoder stem loads into the 5-channel model."""
    from swem_amd import checkpoint
    so = SWEM(O.make_cfg(BACKBONE='resnet18', NUM_BASES=64, SINGLE_OBJ=True))
    mo = SWEM(O.make_cfg(BACKBONE='resnet18', NUM_BASES=64, SINGLE_OBJ=False))
    sd = weights.fill_state_dict(so.state_dict(), seed=8, backbone='resnet18')
    assert sd['value_encoder.conv1.weight'].shape[1] == 4
    path = str(tmp_path / 'SWEM.pth')
    torch.save(sd, path)
    res = checkpoint.load_model(mo, path, strict=True, cpu=True)
    assert not res.missing_keys and not res.unexpected_keys
    w = mo.state_dict()['value_encoder.conv1.weight']
    assert w.shape[1] == 5 and torch.equal(w[:, :4], sd['value_encoder.conv1.weight']) and w[:, 4].abs().sum() > 0
    checkpoint.load_model(so, sd)          # same-arity checkpoints load untouched
    assert torch.equal(so.state_dict()['key_comp.weight'], sd['key_comp.weight'])


def test_training_host_logic_matches_reference_semantics():
    """CPU-side pieces of the training step: bootstrap ratio schedule (bce_losses.py:44-48), MultiStepLR, the flat
    parameter buffer and the host-drawn random bases (modules.py:170-178)."""
    import math
    import torch
    from oracle import swem_oracle as O
    from swem_amd import losses, optim, train
    assert losses.this_p(5, 20, 70, 0.3) is None
    assert losses.this_p(45, 20, 70, 0.3) == pytest.approx(0.3 + 0.7 * 0.5)
    assert losses.this_p(71, 20, 70, 0.3) == 0.3

    class Opt:
        lr = 2e-5
    o = Opt()
    sch = optim.MultiStepLR(o, [3, 6], 0.1)
    ref_p = torch.nn.Parameter(torch.zeros(1))
    ropt = torch.optim.SGD([ref_p], lr=2e-5)
    rs = torch.optim.lr_scheduler.MultiStepLR(ropt, milestones=[3, 6], gamma=0.1)
    for it in range(8):
        assert o.lr == pytest.approx(ropt.param_groups[0]['lr'])
        assert o.lr == pytest.approx(O.multistep_lr(2e-5, [3, 6], 0.1, it))
        ropt.step()
        rs.step()
        sch.step()
    # parameters become views of one buffer (16-byte aligned slots), values kept
    ps = [torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2, 2))]
    before = [p.detach().clone() for p in ps]
    flat, table = optim.flatten_parameters(ps)
    assert flat.numel() == 16 + 8 + 8
    for p, b in zip(ps, before):
        off, n = table[id(p)]
        assert off % 4 == 0 and torch.equal(p.detach(), b) and p.data_ptr() == flat[off:].data_ptr()
    flat.mul_(2)
    assert torch.equal(ps[1].detach(), before[1] * 2)
    # one draw for the whole batch, like the reference: clip b gets slice b
    torch.manual_seed(3)
    kap = train.random_init_host(2, 2, 8, 64)
    torch.manual_seed(3)
    k, n, z = O.random_init((2, 2, 2, 8, 64), 4)
    assert torch.equal(kap, k) and float(z.max()) == pytest.approx(1e-6) and float(n.abs().max()) == 0.0


def test_nchw_views_find_their_nhwc_tensor_again():
    """SWEM.forward hands NHWC activations out as NCHW-shaped views (the reference's callers index them that way); when such a
    view comes back in, the ORIGINAL tensor object -- and with it whatever a producing kernel cached on it (bf16 planes, the
    site its consumers report to) -- is used again (ops.presplit drops cached planes whose tensor was modified in place since: tests/test_gpu_ops.py)."""
    from swem_amd.modules import as_nchw, to_pixel_major
    t = torch.randn(2, 5, 7, 8)                       # NHWC
    t.__dict__['_swem_split'] = {'marker': 1}
    v = as_nchw(t)
    assert v.shape == (2, 8, 5, 7) and v.data_ptr() == t.data_ptr()
    assert to_pixel_major(v) is t and to_pixel_major(v).__dict__['_swem_split'] == {'marker': 1}
    back = to_pixel_major(as_nchw(torch.randn(2, 5, 7, 8)).clone(memory_format=torch.channels_last))
    assert '_swem_split' not in back.__dict__         # a tensor from elsewhere: plain view, nothing attached
    other = torch.randn(2, 8, 5, 7).contiguous(memory_format=torch.channels_last)
    assert torch.equal(to_pixel_major(other), other.permute(0, 2, 3, 1))


def test_shipped_plan_file_is_well_formed():
    """swem_amd/plans/mi355x_480p_k256.json (what `python bench.py` loads by default, ops.PlanBook.load_shipped): every conv key
    is a layer signature (10 fields, or 13 with the (math, ...) tag of the fp32-level mode), every value a plan hint whose
    fields the C ABI accepts (include/swem_hip.h): tile in {64, 128}^2 or the 256-column tiles, math field 0..3, K-split <= 255; untagged entries never
    use the tuner forms that are off by default (prefetched fragments 5 / 7 / 15) except where the whole-frame check kept one."""
    from swem_amd import ops
    book = ops.PlanBook().load_shipped()
    assert len(book.conv) >= 200 and len(book.match) >= 16        # two arithmetics x object counts 1, 2, 3, 5
    untagged = {k: v for k, v in book.conv.items() if len(k) == 10}
    tagged = {k: v for k, v in book.conv.items() if len(k) == 13}
    assert len(untagged) + len(tagged) == len(book.conv) and all(k[10:] == ('math', 0, 1) for k in tagged)
    for k, v in book.conv.items():
        wm, wn, ns, math = v & 15, (v >> 4) & 15, (v >> 8) & 255, (v >> 16) & 7
        # (0 = the tuner found the library's own heuristic -- fp32 MFMA, its choice of tile -- fastest for that shape)
        # (tile 4 x 4 = the 256-column tiles of conv_t256_kernel, round 5: f16x3 with bits 20-23 = tile rows / 32 or 0 for 256;
        # bf16x6 -- the exact-split leg -- on 128-row tiles only)
        t256 = wm == 4 and wn == 4 and (v >> 24) == 0 and ((math == 7 and (v >> 20) & 15 in (0, 4, 5, 6, 7)) or (math == 1 and (v >> 20) & 15 == 4))
        assert v == 0 or ((wm in (1, 2) and wn in (1, 2) or t256) and 1 <= ns <= 255), (k, hex(v))
        assert math in ((0, 1) if len(k) == 13 else (0, 1, 7)), (k, hex(v))     # (7 = f16x3: math 3 + SWEM_PLAN_F16)
        cin, cout, kh, kw, stride, pad, flags, B, H, W = k[:10]
        assert cin > 0 and cout % 4 == 0 and kh == kw and stride in (1, 2) and B in (1, 2, 3, 4, 5, 8, 10, 12, 20, 40)    # 1-5 objects; key encoder batched over a look-ahead of 4, 8 or 10 frames; lock-step lanes (round 6): 4 sequences x 1, 2, 3, 5 objects, 4 x 10 frames
    hist, hist32 = book.math_histogram(), book.math_histogram(('math', 0, 1))
    # the default leg: f16x3 nearly everywhere, never a 16-bit or 8-bit operand mode; the exact-split leg: fp32 MFMA / bf16x6
    assert hist['f16x3'] >= 50 and hist['bf16'] == hist['bf16x3'] == 0 and sum(hist.values()) == len(untagged)
    assert hist32['bf16x6'] + hist32['fp32'] == len(tagged)
    assert sum(1 for v in untagged.values() if (v >> 20) & 15 in (5, 7, 15) and v & 0xff != 0x44) <= 1
    assert isinstance(book.digest(), str) and len(book.digest()) == 12


def test_plan_epochs_gate_planes_only_outputs(tmp_path):
    """ops.PlanBook.epoch(): the count a planes-only convolution output depends on (ops.conv2d(planes_only=True) leaves the fp32
    map out only while its consumer's request for planes carries the CURRENT count).  Every change that can send a consumer down
    another path moves it: a new or changed plan (not a re-assignment of the same value), removing one, loading a file, the
    fallback, clearing, a conv_math / flags block (on entry and on exit); reading does not."""
    from swem_amd import ops
    b = ops.PlanBook()
    seen = [b.epoch()]

    def moved():
        seen.append(b.epoch())
        return seen[-1] != seen[-2]
    key = (64, 64, 3, 3, 1, 1, 2, 1, 120, 216)
    b.conv[key] = 0x30011
    assert moved()
    b.conv[key] = 0x30011
    assert not moved()                          # the same plan again: nothing changed
    b.conv[key] = 0x630022
    assert moved()
    assert b.conv.get(key) == 0x630022 and key in b.conv and not moved()
    b.match[(2, 128, 512, 1620, 256, 2)] = 0x30111
    assert moved()
    b.conv.pop(key)
    assert moved()
    b.fallback = 0x111
    assert moved()
    path = str(tmp_path / 'p.json')
    b.save(path)
    assert not moved()
    b.load(path)
    assert moved()
    with ops.conv_math((3,)):
        assert moved()
    assert moved()
    with ops.flags(FUSE_SPLIT=False):
        assert moved()
    assert moved()
    b.clear()
    assert moved()
    other = ops.PlanBook()                      # books count on their own: another model's tuning does not disturb this one
    e = b.epoch()
    other.conv[key] = 1
    assert b.epoch() == e
    assert len(set(seen)) == len(seen) - 3      # three reads without a change in between


def test_plan_book_scoping_roundtrip_and_flags(tmp_path):
    """ops.PlanBook: what a model learns about its launches belongs to the model -- the current book is swapped for the
    duration of a block and restored, two models never see each other's plans unless they share a book, the tables of the
    default book are reachable under their old module names, save / load round-trips, the digest names the plans."""
    from swem_amd import ops
    a, b = ops.PlanBook(), ops.PlanBook()
    default = ops.BOOK
    with ops.use_book(a):
        ops.BOOK.conv[(256, 256, 3, 3, 1, 1, 2, 2, 120, 216)] = 0x630122
        ops.BOOK.match[(2, 128, 512, 1620, 256, 2)] = 0x130221
        ops.BOOK.hints[('conv', ('engine', 7), 2, 120, 216, 2)] = {True: 2}
        assert ops._CONV_PLANS is a.conv and ops.SPLIT_HINTS is a.hints
        with ops.use_book(b):
            assert not ops.BOOK.conv and ops.BOOK is b
        assert ops.BOOK is a
    assert ops.BOOK is default and not default.conv and not b.conv
    assert a.math_histogram() == {'fp32': 0, 'bf16x6': 0, 'bf16': 0, 'bf16x3': 1, 'f16x3': 0}
    path = str(tmp_path / 'plans.json')
    a.save(path)
    c = ops.PlanBook().load(path)
    assert c.conv == a.conv and c.match == a.match and c.digest() == a.digest() != b.digest()
    m1, m2 = SWEM(O.make_cfg(BACKBONE='resnet18', NUM_BASES=64)), SWEM(O.make_cfg(BACKBONE='resnet18', NUM_BASES=64))
    assert m1.book is not m2.book and isinstance(m1.book, ops.PlanBook)
    # module switches for a block
    assert ops.FUSE_SPLIT and not ops.TUNE_ROUND3_FORMS
    with ops.flags(FUSE_SPLIT=False, TUNE_ROUND3_FORMS=True):
        assert not ops.FUSE_SPLIT and ops.TUNE_ROUND3_FORMS
    assert ops.FUSE_SPLIT and not ops.TUNE_ROUND3_FORMS
    # layer names of the hints: stable under pack_keys, unique otherwise
    with ops.pack_keys('engine'):
        k1 = ops._next_pack_key()
    with ops.pack_keys('engine'):
        k2 = ops._next_pack_key()
    assert k1 == k2 == ('engine', 1) and ops._next_pack_key() != ops._next_pack_key()


def test_launch_ranks_timeout_takes_the_whole_job_down(tmp_path):
    """dist.launch_ranks runs the job in a session of its own: when the timeout fires, the rank processes (which would hold
    the GPUs) are signalled through the process group, not only the torchrun parent."""
    import subprocess
    import time
    from swem_amd import dist as sdist
    script = tmp_path / 'sleeper.py'
    script.write_text('import os, sys, time\n'
                      'open(sys.argv[1] + "." + os.environ["RANK"], "w").write(str(os.getpid()))\n'
                      'print("rank", os.environ["RANK"], flush=True)\n'
                      'time.sleep(120)\n')
    stem = str(tmp_path / 'pid')
    t0 = time.time()
    with pytest.raises(subprocess.TimeoutExpired):
        sdist.launch_ranks(2, [str(script), stem], timeout=20)
    assert time.time() - t0 < 60
    pids = [int(open('%s.%d' % (stem, r)).read()) for r in (0, 1)]
    time.sleep(1.0)
    for pid in pids:
        alive = os.path.exists('/proc/%d' % pid) and 'Z' not in open('/proc/%d/stat' % pid).read().split()[2]
        assert not alive, 'rank process %d survived the timeout' % pid


def test_cpu_pool_is_sized_by_the_container_quota(tmp_path, monkeypatch):
    """dist.respect_cpu_quota: torch's intra-op pool follows the cgroup quota (cpu.max) and the affinity mask, divided by the
    ranks that share the container; train.one_cpu_thread restores the count it found."""
    import builtins
    import torch
    from swem_amd import dist as sdist, train
    before = torch.get_num_threads()
    real_open = builtins.open
    quota = tmp_path / 'cpu.max'
    quota.write_text('400000 100000\n')          # 4 CPUs per 100 ms

    def fake_open(path, *a, **k):
        return real_open(str(quota) if path == '/sys/fs/cgroup/cpu.max' else path, *a, **k)
    try:
        monkeypatch.setattr(builtins, 'open', fake_open)
        torch.set_num_threads(max(before, 4))
        assert sdist.respect_cpu_quota() == min(4, torch.get_num_threads()) <= 4
        assert sdist.respect_cpu_quota(ranks=2) <= 2 and sdist.respect_cpu_quota(ranks=64) == 1
        quota.write_text('max 100000\n')
        torch.set_num_threads(3)
        assert sdist.respect_cpu_quota() <= 3        # no quota: never MORE threads than before
        monkeypatch.undo()
        torch.set_num_threads(before)
        with train.one_cpu_thread():
            assert torch.get_num_threads() == 1
        assert torch.get_num_threads() == before
    finally:
        torch.set_num_threads(before)


def test_full_range_book_policy(tmp_path):
    """The host side of the f16x3 range fault, no GPU needed: a book that leaves the fp16
    range converts its tuned f16x3 conv plans to bf16x6 on the same tile (kernel-variant / tail-split bits dropped: some f16x3
    variants have no three-plane form), drops matching's readout plans and every plane hint, falls back to bf16x6 for untuned
    shapes, stops keeping matching's fp16 value planes, and stays there through later plan loads; the shipped file is keyed by
    the architecture it was tuned on."""
    import json
    from swem_amd import ops
    book = ops.PlanBook(fallback=ops.MODEL_FALLBACK).load_shipped()
    n7 = book.math_histogram()['f16x3']
    assert n7 > 100 and book.match and not book.full_range
    with ops.use_book(book):
        assert ops.value_planes_wanted()
    some = next(k for k, v in book.conv.items() if (v >> 16) & 7 == 7 and (v >> 20) and v & 0xff != 0x44)      # a plan with variant bits
    t256 = next(k for k, v in book.conv.items() if v & 0xff == 0x44)                       # a 256-column tile plan (f16x3 only)
    t256_ns = book.conv[t256] & 0xff00
    tile = book.conv[some] & 0xffff
    book.hints[('conv', ('x',), 1, 2, 3, 0)] = {False: ops.PLANES_F16}
    e0 = book.epoch()
    assert book.to_full_range() == n7
    assert book.full_range and book.epoch() != e0 and not book.hints and not book.match
    assert book.math_histogram()['f16x3'] == 0 and book.conv[some] == (tile | 1 << 16) and (book.fallback >> 16) & 7 == 1
    assert book.conv[t256] == (t256_ns | 0x22 | 1 << 16)          # ... goes back to the 128x128 tile, K-split kept
    with ops.use_book(book):
        assert not ops.value_planes_wanted()
        assert ops._pack_planes((None, None, torch.zeros(1, dtype=torch.float16))) is None
    book.load_shipped()                                        # a later load cannot bring the fp16 range back
    assert book.math_histogram()['f16x3'] == 0 and not book.match and book.full_range
    # books on other arithmetics: the exact fp32 kernels never read the fp16 value planes; a tuned pre-split readout does
    exact = ops.PlanBook(fallback=0)
    with ops.use_book(exact):
        assert not ops.value_planes_wanted()
        exact.match[(2, 128, 512, 1620, 256, 2)] = 0x30111
        assert ops.value_planes_wanted()
        with ops.conv_math((0, 1)):                            # ... but not inside the exact-split block (its own tag)
            assert not ops.value_planes_wanted()
    with ops.use_book(ops.PlanBook(fallback=ops.MODEL_FALLBACK)), ops.conv_math((3,)):
        assert ops.value_planes_wanted()
    # the plan file names its device; a file tuned elsewhere is not taken as a default (device=...: only checked with a GPU)
    d = json.load(open(ops.shipped_plans()))
    assert d['device'] == 'gfx950'
    other = tmp_path / 'other.json'
    json.dump(dict(d, device='gfx942'), open(other, 'w'))
    b2 = ops.PlanBook()
    assert b2.load(str(other)) is b2 and b2.conv                # (no device given: loaded unconditionally)
    # fault-word bookkeeping without a device: nothing to read, nothing raised
    ops.check_faults()
    assert ops.FAULT_BITS.keys() == {1, 2, 4} and issubclass(ops.SwemRangeError, __import__('swem_amd')._lib.SwemHipError)


def test_round6_host_logic(tmp_path, monkeypatch):
    """Host-side pieces of round 6 that need no GPU: ONE f16x3 -> bf16x6 plan conversion for `to_full_range` and `load` (ADVICE r05:
    `load` used to keep the 256-column tile with math 1 and no variant, which the library rejects), no fp16 / 16-bit candidate for
    the tuner on a full-range book, the scheduler wound back over gated optimizer steps, the fault word's owners, per-unit header
    dependencies of the build, the one-rank process-group switch."""
    import json
    from swem_amd import build, dist as sdist, ops, optim
    # --- one conversion
    P = ops.PlanBook._plan_full_range
    assert P(0x770144) == 0x10122 and P(0x670422) == 0x10422 and P(0x30211) == 0x10211      # f16x3 t256 / 128x128+variant, bf16x3
    assert P(0x10422) == 0x10422 and P(0x8810122) == 0x8810122 and P(0) == 0 and P(0x22) == 0x22   # bf16x6 / fp32 stay
    shipped = json.load(open(ops.shipped_plans()))
    book = ops.PlanBook(fallback=ops.MODEL_FALLBACK)
    book.to_full_range()
    book.load(ops.shipped_plans())
    a = ops.PlanBook(fallback=ops.MODEL_FALLBACK).load_shipped()
    a.to_full_range()
    assert book.conv == a.conv                                   # load-after == load-then-convert, entry for entry
    assert not any((v >> 16) & 7 in (7, 3) for v in book.conv.values()) and not any(v & 0xff == 0x44 and (v >> 16) & 7 == 1 and
                                                                                   (v >> 20) & 15 != 4 for v in book.conv.values())
    assert len(book.conv) == len(shipped['conv'])
    # --- the tuner's candidates on a full-range book
    seen = []
    monkeypatch.setattr(ops, '_autotune_pick', lambda cands, timed, reps: seen.append(list(cands)) or cands[0])
    with ops.use_book(book):
        ops._autotune(lambda plan, fresh=False: None, 4096, 512, 144, False, modes=(0, 1, 7, 3))
    assert seen and not any((c >> 16) & 7 in (7, 3) for c in seen[0]) and any((c >> 16) & 7 == 1 for c in seen[0])
    with ops.use_book(ops.PlanBook(fallback=ops.MODEL_FALLBACK)):
        ops._autotune(lambda plan, fresh=False: None, 4096, 512, 144, False, modes=(0, 1, 7))
    assert any((c >> 16) & 7 == 7 for c in seen[1])
    # --- MultiStepLR.rewind: n scheduler steps taken for optimizer steps the device-side gate skipped
    class Opt:
        lr = 1.0
    o = Opt()
    sch = optim.MultiStepLR(o, [3, 6], 0.1)
    for _ in range(7):
        sch.step()
    assert sch.last_epoch == 7 and o.lr == pytest.approx(0.01)
    sch.rewind(3)
    assert sch.last_epoch == 4 and o.lr == pytest.approx(0.1)
    sch.rewind(0)
    assert sch.last_epoch == 4
    # --- owners of the fault word (no device: nothing to drain, nothing raised; registration is by weak reference)
    class Owner:
        got = 0

        def on_foreign_fault(self, bits):
            self.got |= bits
    ops.FAULT_OWNERS.clear()
    own = Owner()
    ops.register_fault_owner(own)
    assert len(ops.FAULT_OWNERS) == 1 and ops.FAULT_OWNERS[0]() is own
    ops.drain_faults('a test')                 # (no fault word exists on a CPU-only host)
    del own
    ops.register_fault_owner(Owner())          # dead references are dropped on the next registration
    assert len(ops.FAULT_OWNERS) == 1
    ops.FAULT_OWNERS.clear()
    txt = ops._fault_text(ops.FAULT_RANGE | ops.FAULT_KSPLIT)
    assert 'K-split' in txt and 'fp16 range' in txt and 'unknown' not in txt and 'unknown fault bits' in ops._fault_text(64)
    # --- build: a unit is rebuilt for ITS headers only (conv.hip does not include the training header)
    conv_h = {p.split('/')[-1] for p in build._headers_of(build.CSRC + '/conv.hip')}
    train_h = {p.split('/')[-1] for p in build._headers_of(build.CSRC + '/train.hip')}
    assert 'swem_hip.h' in conv_h and 'swem_hip_train.h' not in conv_h and 'swem_hip_train.h' in train_h
    # --- the one-rank process-group switch
    monkeypatch.delenv('SWEM_DIST_SINGLE_RANK', raising=False)
    assert not sdist.single_rank_group() and not sdist.active()
    monkeypatch.setenv('SWEM_DIST_SINGLE_RANK', '1')
    assert sdist.single_rank_group() and not sdist.active()     # (no process group initialised here)
    assert ops.graph_capture_kwargs() == {}


def test_lockstep_host_logic():
    """Round 6's lock-step lanes, the parts that need no GPU: the pool refuses a model count that does not make whole lanes (before it
    touches a device), the shipped plan file holds the layer shapes of bench.py's default lanes (four sequences x two objects per
    object layer, 4 x 10 frames per key-encoder pass) in both arithmetics, the bench's defaults are those lanes, and the C ABI
    declares the grouped skip-add the batched decoder uses."""
    import json
    import os
    import re
    import sys
    from swem_amd import evaluator, ops
    with pytest.raises(ValueError):
        evaluator.LockstepPool([object()] * 3, lockstep=2)
    with pytest.raises(ValueError):
        evaluator.LockstepPool([object()] * 4, lockstep=1)
    shipped = json.load(open(ops.shipped_plans()))
    keys = {tuple(k) for k, _ in shipped['conv']}
    for tag in ((), ('math', 0, 1)):
        assert (256, 256, 3, 3, 1, 1, 1, 8, 120, 216) + tag in keys         # decoder, 8 objects at 1/4 scale
        assert (1280, 512, 3, 3, 1, 1, 1, 8, 30, 54) + tag in keys          # value encoder's fusion block, 8 objects at 1/16 scale
        assert (256, 256, 3, 3, 1, 1, 0, 40, 120, 216) + tag in keys        # key encoder + decoder skip conv over 4 x 10 frames
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    src = open(os.path.join(root, 'bench.py')).read()
    assert re.search(r"add_argument\('--seqs', type=int, default=8", src) and re.search(r"add_argument\('--lockstep', type=int, default=4", src)
    hdr = open(os.path.join(root, 'include', 'swem_hip.h')).read()
    assert 'swem_upsample_add_grouped_nhwc_f32(' in hdr and 'swem_upsample_add_grouped_nhwc_f32_planes(' in hdr
