#!/usr/bin/env python3
"""Randomised differential test of the HIP path against the CPU oracle (test infrastructure: needs oracle/ built).
Random signals (the patchwork generator of tests/test_gpu_parity.py), lengths, channel counts, block-mode options,
biases, thresholds and halo splits; units and decoded PCM must be bit-identical.  Usage: fuzz_parity.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib as O          # noqa: E402
import carta1_amd as c1         # noqa: E402
from test_gpu_parity import _patchwork   # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = c1.Context(0)
    t0 = time.time()
    for k in range(cases):
        frames = int(rng.randint(3, 1500))
        nch = int(rng.randint(1, 3))
        chs = [_patchwork(frames, int(rng.randint(1, 1 << 30))) for _ in range(nch)]
        kind = rng.randint(0, 4)
        opts = {}
        if kind == 0:
            opts['fixedBlockModes'] = [0, 0, 0]
        elif kind == 1:
            opts['fixedBlockModes'] = [int(rng.choice([0, 2])), int(rng.choice([0, 2])), int(rng.choice([0, 3]))]
        else:
            opts['transientThresholdLow'] = float(rng.choice([0.1, 0.3, 0.7, 1.0, 1.5]))
        bias = float(rng.choice([0.5, 1.0, 2.0]))
        opts['allocationBias'] = bias
        want, _ = O.encode_stream(chs, fixed_modes=opts.get('fixedBlockModes'), bias=bias, threshold=opts.get('transientThresholdLow', 1.0))
        eo = c1.EncoderOptions(opts, biased_table=O.biased_table(bias))
        ctx.set_speculation(int(rng.choice([0, 1, 1, 2])))      # exact only / material-local / always speculate
        got = ctx.encode(chs, eo)
        assert np.array_equal(got, want), ('units', k, frames, nch, opts)
        # a random split point: the tail encoded from its halo
        cut = int(rng.randint(1, frames))
        h = min(2, cut)
        tail = ctx.encode([c[(cut - h) * 512:] for c in chs], eo, halo_frames=h)
        assert np.array_equal(tail, want.reshape(frames, nch, 212)[cut:].reshape(-1, 212)), ('halo', k, cut, opts)
        if rng.randint(0, 4) == 0:                              # units no encoder wrote: a few flipped bytes
            want = want.copy()
            for _ in range(8):
                want[rng.randint(0, want.shape[0]), rng.randint(0, 212)] ^= np.uint8(rng.randint(1, 256))
        pcm_want, _ = O.decode_stream(want, nch)
        pcm = ctx.decode(want, nch)
        for c in range(nch):
            assert np.array_equal(pcm[c].view(np.uint32), pcm_want[c].view(np.uint32)), ('pcm', k, opts)
        if (k + 1) % 20 == 0:
            print('%d cases ok (%.0f s)' % (k + 1, time.time() - t0), flush=True)
    print('ALL %d CASES BIT-IDENTICAL' % cases)


if __name__ == '__main__':
    main()
