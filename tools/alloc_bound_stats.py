#!/usr/bin/env python3
"""tools/alloc_bound_stats.py (GPU) -- how sharp k_alloc_bound is and how many heaps per unit the two-round scheme runs, per family of
scale-factor vectors (the families of tests/test_gpu_alloc_bound.py), from c1_alloc_bounds_device."""
import sys, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests')]
import oracle_lib as O, carta1_amd as c1
from test_gpu_alloc_bound import _families
ctx = c1.Context(0)
for bias in (0.5, 1.0, 2.0):
    rng = np.random.RandomState(1)
    units = 120000
    side = np.zeros((units, 64), np.uint8); side[:, :52] = _families(rng, units)
    d_side = torch.from_numpy(side).cuda(); d_out = torch.zeros((units, 16), dtype=torch.float64, device='cuda')
    opts = c1.EncoderOptions({'allocationBias': bias, 'fixedBlockModes': [0, 0, 0]}, biased_table=O.biased_table(bias))
    ctx.alloc_bounds_device(d_side.data_ptr(), units, d_out.data_ptr(), opts); ctx.synchronize()
    out = d_out.cpu().numpy(); tot, lb = out[:, :8], out[:, 8:15]
    pos = tot[:, :7] > 0
    gap = (tot[:, :7] - lb)[pos] / tot[:, :7][pos]
    # heaps per unit under the scheme: 52 always; cheap bound; then LB rounds
    per = units // 10
    names = ['random', 'flat', 'falling', 'partials', 'sparse', 'lowpass', 'equal', 'quiet', 'loud', 'rising']
    print('bias', bias, 'gap median %.4f p99 %.4f max %.4f' % (np.median(gap), np.percentile(gap, 99), gap.max()))
    for i, nm in enumerate(names):
        sl = slice(i * per, (i + 1) * per if i < 9 else units)
        T, L = tot[sl], lb[sl]
        best = T[:, 7].copy(); heaps = np.ones(len(T)); brute = np.zeros(len(T))
        alive = L <= best[:, None]
        # first round: argmin LB among alive
        Lm = np.where(alive, L, np.inf); cs = np.argmin(Lm, axis=1); has = alive.any(axis=1)
        t1 = T[np.arange(len(T)), cs]; best = np.where(has & (t1 < best), t1, best); heaps += has
        alive2 = alive & (L <= best[:, None]); alive2[np.arange(len(T)), cs] = False
        heaps += alive2.sum(axis=1)
        print('   %-9s heaps/unit %.2f' % (nm, heaps.mean()))
