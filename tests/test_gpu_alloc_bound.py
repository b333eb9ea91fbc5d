"""GPU: the lower bounds the bit allocation prunes candidates with (k_alloc_bound) against brute force.
c1_alloc_bounds_device runs the greedy heap of allocateBits (bitallocation.js:74-142) for all eight candidate BFU
counts of every unit and returns the totals next to the bounds and next to what the pruned production path chose:
  * a bound above its candidate's total would let the library skip a candidate that might win -> never;
  * the production choice must be the reference's: smallest total, first (smallest count) on ties (:116-129)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    import carta1_amd as c1
    c = c1.Context(0)
    yield c
    c.close()


def _families(rng, units):
    """scale-factor index vectors: what transforms of real signals give, and what they never would"""
    per = units // 10
    out = []
    out.append(rng.randint(0, 64, size=(per, 52)))                                        # anything
    out.append(np.clip(rng.randint(20, 50, size=(per, 1)) + rng.randint(-2, 3, size=(per, 52)), 0, 63))   # flat spectra (noise)
    tilt = -rng.uniform(0.2, 1.5, size=(per, 1)) * np.arange(52)[None, :]
    out.append(np.clip(rng.randint(40, 63, size=(per, 1)) + tilt + rng.randint(-3, 4, size=(per, 52)), 0, 63).astype(int))   # falling spectra
    peaks = np.full((per, 52), 3) + rng.randint(0, 4, size=(per, 52))
    for k in range(per):
        for b in rng.choice(52, size=rng.randint(1, 6), replace=False):
            peaks[k, b] = rng.randint(35, 64)
    out.append(peaks)                                                                     # a few partials over a floor
    sparse = rng.randint(1, 64, size=(per, 52)) * (rng.uniform(size=(per, 52)) < rng.uniform(0.05, 0.9, size=(per, 1)))
    out.append(sparse)                                                                    # many silent BFUs
    low = rng.randint(0, 64, size=(per, 52))
    for k in range(per):
        low[k, rng.randint(8, 52):] = 0
    out.append(low)                                                                       # silent above some BFU
    out.append(np.repeat(rng.randint(0, 64, size=(per, 1)), 52, axis=1))                  # all equal (ties everywhere)
    out.append(np.clip(rng.randint(0, 4, size=(per, 52)), 0, 63))                         # nearly silent
    out.append(np.clip(60 + rng.randint(0, 4, size=(per, 52)), 0, 63))                    # clipping level
    rest = units - 9 * per
    up = np.clip(rng.randint(0, 20, size=(rest, 1)) + (rng.uniform(0.3, 1.2, size=(rest, 1)) * np.arange(52)[None, :]).astype(int), 0, 63)
    out.append(up)                                                                        # rising spectra: the upper BFUs matter most
    return np.concatenate(out).astype(np.uint8)


@pytest.mark.parametrize('bias', [0.5, 1.0, 2.0, 1.37])
def test_bounds_hold_and_the_pruned_choice_is_the_brute_force_minimum(ctx, bias):
    import torch
    import carta1_amd as c1
    rng = np.random.RandomState(int(bias * 100))
    units = 120000
    side = np.zeros((units, 64), np.uint8)
    side[:, :52] = _families(rng, units)
    d_side = torch.from_numpy(side).cuda()
    d_out = torch.zeros((units, 16), dtype=torch.float64, device='cuda')
    opts = c1.EncoderOptions({'allocationBias': bias, 'fixedBlockModes': [0, 0, 0]}, biased_table=O.biased_table(bias))
    ctx.alloc_bounds_device(d_side.data_ptr(), units, d_out.data_ptr(), opts)
    ctx.synchronize()
    out = d_out.cpu().numpy()
    tot, lb, choice = out[:, :8], out[:, 8:15], out[:, 15]
    assert np.isfinite(tot).all() and (tot >= 0).all()
    bad = np.argwhere(lb > tot[:, :7])
    assert bad.size == 0, ('bound above total', bias, bad[:5], lb[bad[0][0]], tot[bad[0][0]])
    brute = np.argmin(tot, axis=1)                       # first minimum = smallest BFU count on ties
    wrong = np.nonzero(choice != brute)[0]
    assert wrong.size == 0, ('choice', bias, wrong[:5], choice[wrong[:5]], brute[wrong[:5]], tot[wrong[0]])
    # how sharp: the bound decides nothing unless it is close to the totals
    pos = tot[:, :7] > 0
    gap = np.where(pos, (tot[:, :7] - lb) / np.where(pos, tot[:, :7], 1), 0)
    assert np.median(gap[pos]) < 0.05, np.median(gap[pos])


def test_bounds_on_transformed_signals(ctx):
    """the same through the real analysis: white noise, pink noise with bursts, stationary partials, and near silence"""
    import torch
    import carta1_amd as c1
    frames = 4096
    t = np.arange(frames * 512, dtype=np.float64)
    tone = sum(a * np.sin(2 * np.pi * f * t / 44100 + p) for a, f, p in ((0.3, 220, 0), (0.2, 440, 1), (0.1, 660, 2), (0.05, 1320, .5), (0.02, 3300, .1), (0.01, 7040, .3)))
    harm = sum((0.4 / (k + 1)) * np.sin(2 * np.pi * 110 * (k + 1) * t / 44100 + k) for k in range(40))
    sigs = [O.gen_white(5, frames * 512), O.gen_pinkT(6, frames * 512), tone.astype(np.float32), harm.astype(np.float32),
            (1e-4 * tone).astype(np.float32), (harm * 0.5 + 0.01 * O.gen_white(7, frames * 512)).astype(np.float32)]
    for bias in (1.0, 2.0):
        opts = c1.EncoderOptions({'allocationBias': bias, 'fixedBlockModes': [0, 0, 0]}, biased_table=O.biased_table(bias))
        for x in sigs:
            d_pcm = torch.from_numpy(np.ascontiguousarray(x)).cuda()
            d_coefs = torch.zeros(frames * 512, dtype=torch.float32, device='cuda')
            d_side = torch.zeros(frames * 64, dtype=torch.uint8, device='cuda')
            d_alloc = torch.zeros(frames * 32, dtype=torch.uint8, device='cuda')
            ctx.encode_stages_device([d_pcm.data_ptr()], frames, 0, d_coefs.data_ptr(), d_side.data_ptr(), d_alloc.data_ptr(), opts)
            d_out = torch.zeros((frames, 16), dtype=torch.float64, device='cuda')
            ctx.alloc_bounds_device(d_side.data_ptr(), frames, d_out.data_ptr(), opts)
            ctx.synchronize()
            out = d_out.cpu().numpy()
            tot, lb, choice = out[:, :8], out[:, 8:15], out[:, 15]
            assert (lb <= tot[:, :7]).all()
            assert np.array_equal(choice, np.argmin(tot, axis=1))
            amount = (d_alloc.cpu().numpy().reshape(frames, 32).view(np.uint32)[:, 7] >> 28) & 7
            assert np.array_equal(amount, np.argmin(tot, axis=1))        # what encode itself chose
