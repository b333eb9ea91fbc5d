#!/usr/bin/env python3
"""Is the pop order of the reference's heap among EQUAL priorities (distributeBitsRDO / siftDown, bitallocation.js:203-341) a
simple rule?  If it were, the greedy allocation would have a closed form (sort by priority, then by the rule) and could run
wave-cooperatively inside the analysis kernel.  A priority-sorted greedy with four tie rules against the heap's word lengths
on noise-like scale-factor vectors: none reproduces more than a fifth of the units.  Test infrastructure
(uses the Python restatement of the heap in tests/test_alloc_bound_cpu.py and the oracle's tables)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_alloc_bound_cpu import distribute, SPECS, DDF, DB, f32   # noqa: E402
import oracle_lib as O                                              # noqa: E402


def sorted_greedy(n, rem, bsf, sfi, key):
    wl = [0] * n
    live = {b: f32(bsf[sfi[b]] * DDF[0] / DB[0]) for b in range(n) if sfi[b]}
    seq, stamp = 0, {b: 0 for b in live}
    while rem > 0 and live:
        b = max(live, key=lambda b: (live[b], key(b, stamp[b])))
        cur = wl[b]
        cost = DB[cur] * SPECS[b]
        if cost > rem or cost <= 0:
            del live[b]
            continue
        rem -= cost
        wl[b] = cur + 1
        if cur + 1 < 15 and DB[cur + 1] > 0:
            live[b] = f32(bsf[sfi[b]] * DDF[cur + 1] / DB[cur + 1])
            seq += 1
            stamp[b] = seq
        else:
            del live[b]
    return wl


def main():
    bsf = [O.h2d(x) for x in O.golden_tables()['scale_factors_f64']]
    rng = np.random.RandomState(1)
    rules = {'BFU index ascending': lambda b, s: -b, 'BFU index descending': lambda b, s: b,
             'oldest entry first': lambda b, s: -s, 'newest entry first': lambda b, s: s}
    hits = {k: 0 for k in rules}
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    for _ in range(n):
        sfi = list(np.clip(rng.randint(25, 45) + rng.randint(-3, 4, size=52), 1, 63))
        ref = distribute(52, 1136, bsf, sfi)
        for k, f in rules.items():
            hits[k] += sorted_greedy(52, 1136, bsf, sfi, f) == ref
    for k, v in hits.items():
        print('%-24s reproduces the heap on %3d of %d units' % (k, v, n))


if __name__ == '__main__':
    main()
