#!/usr/bin/env python
"""VERDICT r05 item 9, the one untried lever for EM + matching of ONE sequence: the N objects of a frame are independent problems
(modules.py:129-168 couples nothing across objects), so run them as N single-object chains on N streams -- one object's M step
under another's E / W step -- instead of one chain whose every launch carries all objects.  Probe before building: tools/em_bench.py's
`concurrent` (memorize + match from HIP graphs, one per stream) with 1 stream x 2 objects against 2 streams x 1 object (and 3 / 3).
   python tools/em_object_split_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.argv = sys.argv[:1]
import em_bench  # noqa: E402

if __name__ == '__main__':
    for n in (2, 3):
        print('--- %d objects: one chain with all objects, then one single-object chain per object on its own stream' % n)
        em_bench.concurrent(1, objects=n)
        em_bench.concurrent(n, objects=1)
