"""GPU: a bounded cut of tools/fuzz_parity.py under the driver -- random signals, lengths, channel counts, block-mode
options, biases, thresholds, speculation modes and halo splits; units and decoded PCM bit-identical to the oracle.
Also the transient detector's stage taps against the reference's own magnitudes and decisions."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from test_gpu_parity import _patchwork

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def ctx():
    import carta1_amd as c1
    c = c1.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_fuzz_against_the_oracle(ctx, seed):
    import carta1_amd as c1
    rng = np.random.RandomState(seed)
    for k in range(50):
        frames = int(rng.randint(3, 420))
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
        ctx.set_speculation(int(rng.choice([0, 1, 2, 2])))
        want, _ = O.encode_stream(chs, fixed_modes=opts.get('fixedBlockModes'), bias=bias, threshold=opts.get('transientThresholdLow', 1.0))
        eo = c1.EncoderOptions(opts, biased_table=O.biased_table(bias))
        got = ctx.encode(chs, eo)
        assert np.array_equal(got, want), ('units', seed, k, frames, nch, opts)
        cut = int(rng.randint(1, frames))                      # the tail encoded from its halo
        h = min(2, cut)
        tail = ctx.encode([c[(cut - h) * 512:] for c in chs], eo, halo_frames=h)
        assert np.array_equal(tail, want.reshape(frames, nch, 212)[cut:].reshape(-1, 212)), ('halo', seed, k, cut, opts)
        pcm_want, _ = O.decode_stream(want, nch)
        pcm = ctx.decode(want, nch)
        for c in range(nch):
            assert np.array_equal(pcm[c].view(np.uint32), pcm_want[c].view(np.uint32)), ('pcm', seed, k, opts)
    ctx.set_speculation(1)


def test_detector_taps_against_reference_magnitudes_and_decisions(ctx):
    """performFFT's magnitudes bit for bit, and detectTransient's decision in every band at the seven thresholds the
    reference was run with (tests/golden/stages_pinkT3_*, written by the reference itself)."""
    import torch
    import carta1_amd as c1
    p = O.gen_pinkT(3, 8 * 512)
    ref_mags = np.fromfile(os.path.join(G, 'stages_pinkT3_mags.f32.bin'), dtype=np.float32).reshape(8, 256)
    ref = json.load(open(os.path.join(G, 'stages_pinkT3_transient.json')))
    d_pcm = torch.from_numpy(p).cuda()
    mags = torch.zeros(8 * 256, dtype=torch.float32, device='cuda')
    modes = torch.zeros(8, dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize()
    for ti, thr in enumerate(ref['thresholds']):
        ctx.detect_stages_device([d_pcm.data_ptr()], 8, mags.data_ptr(), modes.data_ptr(),
                                 c1.EncoderOptions({'transientThresholdLow': thr}))
        ctx.synchronize()
        assert np.array_equal(mags.cpu().numpy().reshape(8, 256).view(np.uint32), ref_mags.view(np.uint32))
        m = modes.cpu().numpy().astype(int)
        got = [[int((m[f] >> (2 * b)) & 3) != 0 for b in range(3)] for f in range(8)]
        want = [[bool(x) for x in ref['decisions_frame_thr_band'][f][ti]] for f in range(8)]
        assert got == want, (thr, got, want)
        # and the mode values themselves: 2, 2, 3 (encoder.js:143)
        for f in range(8):
            for b in range(3):
                assert ((m[f] >> (2 * b)) & 3) in (0, 3 if b == 2 else 2)


@pytest.mark.gpu
@pytest.mark.parametrize('fn,name', list(enumerate(O.LIBM_FUNCTIONS)))
def test_engine_libm_on_the_device_matches_v8(ctx, fn, name):
    """Math.log / exp / log1p / log10 inside the detector's kernels against V8's own results (tests/golden/libm_v8_*.bin,
    tests/golden/gen/gen_libm.mjs): bit for bit on every vector, NaN for NaN.  The functions are not correctly rounded
    (glibc differs from V8 on 1-7 % of these arguments), so this pins the algorithm, not just its accuracy."""
    import torch
    io = np.fromfile(os.path.join(G, 'libm_v8_%s.bin' % name), dtype='<f8').reshape(-1, 2)
    x = torch.from_numpy(np.ascontiguousarray(io[:, 0])).cuda()
    y = torch.zeros_like(x)
    torch.cuda.synchronize()
    ctx.libm_device(fn, x.data_ptr(), y.data_ptr(), x.numel())
    ctx.synchronize()
    got, want = y.cpu().numpy(), np.ascontiguousarray(io[:, 1])
    same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), (name, io[~same][:4], got[~same][:4])
    # and the oracle's restatement on a denser set of detector-like arguments (Float32 magnitudes above 1e-10)
    if name == 'log':
        rng = np.random.default_rng(5)
        m = np.exp(rng.uniform(np.log(1.0000001e-10), np.log(1e6), 1 << 20)).astype(np.float32).astype(np.float64)
        x = torch.from_numpy(m).cuda()
        y = torch.zeros_like(x)
        torch.cuda.synchronize()
        ctx.libm_device(0, x.data_ptr(), y.data_ptr(), x.numel())
        ctx.synchronize()
        assert np.array_equal(y.cpu().numpy().view(np.uint64), O.libm('log', m).view(np.uint64))
