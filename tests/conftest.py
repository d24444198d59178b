import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'leaves_fault: the test provokes an asynchronous fault on purpose and does not collect it')
    # The CPU oracle runs beside the GPU tests.  torch sizes its intra-op pool by the HOST's logical CPUs (256 on the GPU boxes of
    # this pool) while the container may use 16 of them per 100 ms (cgroup cpu.max): a 128-thread pool burns that budget in
    # milliseconds and the kernel then stalls the whole process for the rest of every period (training row).
    # One thread per CPU of the quota is all the oracle can use anyway.
    from swem_amd import dist as sdist
    sdist.respect_cpu_quota()


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    import torch

    def load(name):
        with np.load(os.path.join(GOLDEN, name)) as z:
            return {k: ([str(x) for x in z[k].tolist()] if z[k].dtype.kind in 'US' and z[k].ndim else
                        torch.from_numpy(z[k]) if z[k].ndim else z[k].item()) for k in z.files}
    return load


@pytest.fixture(scope='session')
def lib():
    """Build (if stale) and load libswem_hip.so."""
    from swem_amd import build, _lib
    build.build()
    return _lib.load()


@pytest.fixture(autouse=True)
def _fresh_plans(request):
    """Every test starts from an empty default PlanBook, the tuner off and no forced conv math: the arithmetic a test runs is
    what the test itself selects (tests/helpers.py::arith), never what an earlier test left behind.  (Models own their books,
    ops.PlanBook; this covers the free-standing ops.* calls.)  And it must END with a clean fault word: a test that provokes an
    asynchronous fault collects it (ops.check_faults / the evaluator / the trainer) -- one that leaves the word set fails here
    (silently clearing it hid faults nobody looked at), unless it is marked `leaves_fault` and says why."""
    from swem_amd import ops
    ops.reset_plans()
    ops.AUTOTUNE = False
    ops.MATH_RAN = None
    ops.FAULT_OWNERS.clear()
    assert ops._PLAN_TAG == (), 'a conv_math context leaked out of a test'
    yield
    ops.AUTOTUNE = False
    ops.MATH_RAN = None
    ops.FAULT_OWNERS.clear()
    left = 0
    for t in ops._fault.values():
        w = int(t.item())
        if w:
            left |= w
            t.zero_()
    if left and request.node.get_closest_marker('leaves_fault') is None:
        pytest.fail('the test left the device fault word set (%#x: %s) without collecting it' % (left, ops._fault_text(left)))
