"""CPU: libswem_hip.so builds for gfx950, loads, and exports exactly what include/*.h declare."""
import ctypes
import os
import re

import pytest

from swem_amd import _lib

INCLUDE = os.path.join(os.path.dirname(__file__), '..', 'include')


def declared_symbols():
    src = ''.join(open(os.path.join(INCLUDE, h)).read() for h in sorted(os.listdir(INCLUDE)) if h.endswith('.h'))
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(swem_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_are_exported_and_bound(lib):
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), 'library does not export %s' % s
        assert s in _lib.SIGNATURES, 'ctypes binding lacks %s' % s
    assert sorted(_lib.SIGNATURES) == syms, 'binding lists symbols the header does not declare'


def test_version_and_error_channel(lib):
    assert lib.swem_version() == 2      # round 5: explicit fault-word arguments
    # argument validation happens before any HIP call, so it can be exercised without a GPU
    rc = lib.swem_conv2d_nhwc_f32(None, None, 4, 0, None, 0, 0, None, 0, 0, 1, 8, 8, None, 0, None, None, None, 0, None,
                                  32, 3, 3, 1, 1, 0, 0, None, 0)
    assert rc == -4 and b'null pointer' in lib.swem_last_error()
    rc = lib.swem_match_f32(None, 1, 1, 1, None, None, 1, 1, 1, 128, 512, 100, 100, 64, ctypes.c_float(0.05), 0, None, 0)
    assert rc == -1 and b'bases per class' in lib.swem_last_error()
    with pytest.raises(_lib.SwemHipError):
        _lib.call('swem_em_ew_f32', None, 1, 1, None, None, None, None, 1, 100, 10, 64, 0.05, 1, 0)


def test_size_queries(lib):
    assert lib.swem_em_pad(1620) == 1664 and lib.swem_em_pad(1664) == 1664
    assert lib.swem_match_pad(1620) == 1664
    assert lib.swem_match_workspace(2, 128, 512, 1620, 256, 2, 0) >= 2 * 1024 * (128 + 512 + 1664) * 4
    assert lib.swem_memorize_workspace(2, 128, 512, 1620, 256) > 0
    # a 1/16-scale 3x3 conv (13 x 4 tiles of 128x128) is split over K; a full-resolution one is not
    assert lib.swem_conv2d_workspace(1, 30, 54, 1024, 512, 3, 3, 1, 1, 0, 0) > 0
    assert lib.swem_conv2d_workspace(2, 120, 216, 256, 256, 3, 3, 1, 1, 0, 0) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.SwemHipError, match='no CPU fallback'):
        _lib.load()


@pytest.mark.skipif(not os.environ.get('SWEM_SLOW_TESTS'), reason='the FULL ISA check of conv.hip takes ~5 minutes: nightly (SWEM_SLOW_TESTS=1)')
def test_lds_dma_kernels_are_the_only_m0_users_every_instantiation():
    """The check below on EVERY instantiation of conv_igemm_bf3s_kernel (stream-K, one- and three-plane kernels, the four-stage and
    prefetched-fragment variants): the default run compiles a fifth of them (-DSWEM_ISA_SUBSET) to stay within a minute, so an
    M0 regression confined to the others would pass it.  Run with SWEM_SLOW_TESTS=1."""
    _check_m0(subset=False, at_least=30)


def test_lds_dma_kernels_are_the_only_m0_users():
    """conv.hip issues buffer_load ... lds from inline asm and writes M0 itself, without saving it (csrc/conv.hip, dma16).
    That is only sound while hipcc keeps nothing of its own in M0 inside those kernels: check the generated ISA.
    (-DSWEM_ISA_SUBSET: every block tile, the two-plane kernels of both operand formats, the eight-wave and 16-k-block forms --
    a fifth of the instantiations, so that the check compiles in about a minute; the rest: the slow test above.)"""
    _check_m0(subset=True, at_least=6)


def _check_m0(subset, at_least):
    import subprocess
    src = os.path.join(os.path.dirname(__file__), '..', 'swem_amd', 'csrc', 'conv.hip')
    asm = ''
    for unit in ([], ['-DSWEM_CONV_T256_ONLY']):        # (conv.hip is two translation units: swem_amd/build.py)
        asm += subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S',
                               *(['-DSWEM_ISA_SUBSET'] if subset else []), *unit, src, '-o', '-'], check=True,
                              capture_output=True, text=True, timeout=1800).stdout
    checked = 0
    t256 = 0
    for m in re.finditer(r'^(_ZN\S*(?:conv_igemm_bf3s_kernel|conv_t256_kernel)\S*):', asm, flags=re.M):
        name = m.group(1)
        body = asm[m.end():asm.index('s_endpgm', m.end())]
        uses = [ln.strip() for ln in body.splitlines() if re.search(r'\bm0\b', ln) and not ln.strip().startswith(';')]
        assert uses, name
        assert all(re.fullmatch(r's_mov_b32 m0, s\d+', u) for u in uses), (name, uses[:5])
        assert body.count('offen lds') == len(uses)            # one M0 write per transfer, nothing else
        checked += 1
        t256 += 'conv_t256_kernel' in name
    assert checked >= at_least and t256 == 6          # (conv_t256_kernel: the 256-row form, four tile heights, the bf16x6 form)


def test_integration_md_binding_matches_the_signature_table(lib):
    """INTEGRATION.md section 2's ctypes stub (the reference-side binding of matching) is executed on the GPU by
    tests/test_gpu_em.py::test_integration_md_ctypes_binding_runs_as_printed; here, without a GPU: the block compiles, loads the
    library, and declares the argument types swem_amd/_lib.py holds for the same entry points."""
    root = os.path.join(os.path.dirname(__file__), '..')
    md = open(os.path.join(root, 'INTEGRATION.md')).read()
    sec = md[md.index('## 2. Binding the C ABI directly'):md.index('## 3.')]
    code = re.search(r'```python\n(.*?)```', sec, flags=re.S).group(1)
    ns = {}
    exec(compile(code.replace("'swem_amd/libswem_hip.so'", repr(os.path.join(root, 'swem_amd', 'libswem_hip.so'))),
                 'INTEGRATION.md#2', 'exec'), ns)
    for name in ('swem_match_workspace', 'swem_match_f32'):
        fn = getattr(ns['lib'], name)
        res, args = _lib.SIGNATURES[name]
        assert fn.restype is res and list(fn.argtypes) == list(args), name
    assert callable(ns['matching_features'])
