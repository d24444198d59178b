"""Builds libswem_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libswem_hip.so')
SOURCES = ['api.hip', 'conv.hip', 'bneck.hip', 'pointwise.hip', 'em.hip', 'match.hip', 'train.hip', 'train_conv.hip']
# (source, extra flags, object): conv.hip a second time for conv_t256_kernel alone (-DSWEM_CONV_T256_ONLY: csrc/conv.hip)
EXTRA_UNITS = [('conv.hip', ['-DSWEM_CONV_T256_ONLY'], 'conv_t256.o')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers_of(path, seen=None):
    """The repo headers a source includes, transitively (`#include "..."` lines only): a unit is rebuilt when ONE OF ITS OWN
    headers is newer than its object -- conv.hip (five minutes) does not include the training header."""
    import re
    seen = set() if seen is None else seen
    try:
        text = open(path).read()
    except OSError:
        return seen
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        h = os.path.normpath(os.path.join(os.path.dirname(path), inc))
        if h not in seen and os.path.exists(h):
            seen.add(h)
            _headers_of(h, seen)
    return seen


def build(force=False, verbose=False):
    from concurrent.futures import ThreadPoolExecutor
    objs, todo = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + sorted(_headers_of(s))):
            todo.append([HIPCC] + FLAGS + ['-c', s, '-o', o])
        objs.append(o)
    for src, extra, obj in EXTRA_UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, obj)
        if force or _stale(o, [s] + sorted(_headers_of(s))):
            todo.append([HIPCC] + FLAGS + extra + ['-c', s, '-o', o])
        objs.append(o)

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    # the translation units are independent: compile the stale ones side by side (conv.hip alone takes ~5 minutes; the
    # container has 8 CPUs -- SWEM_BUILD_JOBS overrides)
    jobs = max(1, min(len(todo), int(os.environ.get('SWEM_BUILD_JOBS', '0')) or (os.cpu_count() or 1)))
    if todo:
        with ThreadPoolExecutor(jobs) as ex:
            list(ex.map(run, todo))
    if force or _stale(LIB, objs):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
    print(LIB)
