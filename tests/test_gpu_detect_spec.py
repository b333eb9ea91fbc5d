"""GPU: the speculative transient detector (k_detect_features<SPEC>, k_detect_decide<SPEC>, k_detect_recheck;
carta1_amd/csrc/c1_detect_bound.h) against the exact detector and the oracle.
 1. the device's binary32 log2 meets the error model the interval assumes, for every normal positive binary32 number;
 2. the interval [lo, hi] the binary32 path derives contains the reference's transient score (transient.js:197-226) for
    every unit and band of twenty signal classes, and is narrow enough to decide nearly all of them;
 3. the block modes after the exact recheck are the exact detector's, with the threshold anywhere -- also in the middle
    of the score distribution, where many units are left open;
 4. encoding with the speculative detector is bit-identical to encoding with the exact one and to the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from test_spec_bound import signals

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    import carta1_amd as c1
    c = c1.Context(0)
    yield c
    c.close()


def more_signals():
    n = 96 * 512
    t = np.arange(n)
    rng = np.random.default_rng(11)
    yield 'silence', np.zeros(n, dtype=np.float32)
    x = O.gen_white(5, n)
    x[20 * 512:40 * 512] = 0
    x[60 * 512:61 * 512] = 0
    yield 'noise_with_silent_gaps', x
    yield 'castanets', (O.gen_pinkT(9, n) * (0.05 + (t % 4096 < 300))).astype(np.float32)
    yield 'fade_in', (O.gen_white(6, n) * np.linspace(0, 1, n) ** 4).astype(np.float32)
    yield 'denormal_noise', (rng.standard_normal(n) * 1e-38).astype(np.float32)
    yield 'below_validity', (rng.standard_normal(n) * 3e-11).astype(np.float32)      # magnitudes around the 1e-10 validity threshold
    yield 'lowpassed', np.convolve(O.gen_white(7, n), np.ones(24) / 24, mode='same').astype(np.float32)
    yield 'tone_on_off', (0.5 * np.sin(2 * np.pi * 3000 * t / 44100) * ((t // 2048) % 2)).astype(np.float32)


ALL = list(signals()) + list(more_signals())


def scores(ctx, pcm, speculative, threshold=1.0, halo=0):
    import torch
    import carta1_amd as c1
    other = O.gen_pinkT(4, len(pcm))
    frames = len(pcm) // 512 - halo
    dev = [torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in (pcm, other)]
    units = frames * 2
    sc = torch.zeros(units * 6, dtype=torch.float64, device='cuda')
    modes = torch.zeros(units, dtype=torch.uint8, device='cuda')
    opened = torch.zeros(1, dtype=torch.int32, device='cuda')
    opts = c1.EncoderOptions({'transientThresholdLow': threshold})
    ptrs = [d.data_ptr() + halo * 2048 for d in dev]
    torch.cuda.synchronize()
    ctx.detect_scores_device(ptrs, frames, sc.data_ptr(), modes.data_ptr(), opened.data_ptr(), opts, halo_frames=halo, speculative=speculative)
    ctx.synchronize()
    return sc.cpu().numpy().reshape(units, 3, 2), modes.cpu().numpy(), int(opened.item())


def test_device_log2_meets_the_error_model(ctx):
    """|v_log_f32(x) - log2 x| <= 2 u |log2 x| + 2^-22 for every normal positive binary32 x (c1_detect_bound.h)"""
    worst_rel = worst_abs = 0.0
    first, end, step = 0x00800000, 0x7f800000, 1 << 28
    while first < end:
        rel, ab = ctx.log2f_error(first, min(step, end - first))
        worst_rel, worst_abs = max(worst_rel, rel), max(worst_abs, ab)
        first += step
    assert worst_rel <= 2.0, worst_rel            # in units of 2^-24, where |log2 x| >= 2^-6
    assert worst_abs <= 2.0 ** -22, worst_abs     # elsewhere (x near 1)


WORST = {}


@pytest.mark.parametrize('name,pcm', ALL, ids=[s[0] for s in ALL])
def test_magnitudes_lie_within_their_bound(ctx, name, pcm):
    """|| binary32 magnitudes - the exact kernel's (= the reference's Float32) magnitudes ||_2 <= Delta, per band and frame:
    the K of c1_detect_bound.h at work.  The worst ratio seen is printed (DESIGN.md 3c quotes it)."""
    import torch
    import carta1_amd as c1
    other = O.gen_white(8, len(pcm))
    frames = len(pcm) // 512
    dev = [torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in (pcm, other)]
    ptrs = [d.data_ptr() for d in dev]
    units = frames * 2
    exact = torch.zeros(units * 256, dtype=torch.float32, device='cuda')
    spec = torch.zeros(units * 256, dtype=torch.float32, device='cuda')
    bounds = torch.zeros(units * 3, dtype=torch.float32, device='cuda')
    torch.cuda.synchronize()
    ctx.detect_stages_device(ptrs, frames, exact.data_ptr(), 0, c1.EncoderOptions({}))
    ctx.detect_spec_mags_device(ptrs, frames, spec.data_ptr(), bounds.data_ptr())
    ctx.synchronize()
    e = exact.cpu().numpy().reshape(units, 256).astype(np.float64)
    g = spec.cpu().numpy().reshape(units, 256).astype(np.float64)
    b = bounds.cpu().numpy().reshape(units, 3).astype(np.float64)
    worst = 0.0
    for k, (lo, hi) in enumerate(((0, 64), (64, 128), (128, 256))):
        dist = np.sqrt(np.sum((g[:, lo:hi] - e[:, lo:hi]) ** 2, axis=1))
        fin = np.isfinite(dist) & np.isfinite(b[:, k])
        assert (dist[fin] <= b[fin, k]).all(), (name, k, np.argwhere(fin & (dist > b[:, k]))[:4])
        assert (~np.isfinite(b[~np.isfinite(dist), k])).all()          # a non-finite spectrum never carries a finite bound
        pos = fin & (b[:, k] > 0)
        if pos.any():
            worst = max(worst, float(np.max(dist[pos] / b[pos, k])))
        assert (dist[fin & (b[:, k] == 0)] == 0).all()                   # Delta = 0 only for an all-zero band, whose magnitudes are exact
    WORST[name] = worst
    print('worst ||error|| / Delta for %s: %.4f' % (name, worst))
    assert worst < 0.5, (name, worst)                                    # the bound is a worst case over rounding directions; a sanity margin


@pytest.mark.parametrize('name,pcm', ALL, ids=[s[0] for s in ALL])
def test_interval_contains_the_reference_score(ctx, name, pcm):
    exact, modes_exact, _ = scores(ctx, pcm, False)
    spec, modes_spec, opened = scores(ctx, pcm, True)
    s = exact[:, :, 0]
    lo, hi = spec[:, :, 0], spec[:, :, 1]
    known = np.isfinite(s)
    assert ((lo <= s) & (s <= hi))[known].all(), (name, np.argwhere(~((lo <= s) & (s <= hi)) & known)[:4])
    assert np.array_equal(modes_spec, modes_exact)
    width = (hi - lo)[np.isfinite(hi - lo) & (np.abs(hi) < 1e299)]
    if name in ('white', 'pink_bursts', 'castanets', 'lowpassed', 'sign_noise'):
        assert opened <= 0.02 * len(modes_exact), (name, opened)
        assert np.median(width) < 0.02, (name, float(np.median(width)))


@pytest.mark.parametrize('name', ['white', 'pink_bursts', 'castanets', 'tone_on_off', 'noise_with_silent_gaps'])
def test_threshold_inside_the_score_distribution(ctx, name):
    pcm = dict(ALL)[name]
    exact, _, _ = scores(ctx, pcm, False)
    s = exact[:, :, 0]
    for q in (0.2, 0.5, 0.9):
        thr = float(np.quantile(s[np.isfinite(s)], q, method='nearest'))      # the score of some unit and band
        if not 0.011 < thr < 1.99:                # the option's range (options.js:101-105)
            continue
        for t in (thr, np.nextafter(thr, 0), np.nextafter(thr, 9)):
            _, modes_exact, _ = scores(ctx, pcm, False, threshold=t)
            _, modes_spec, opened = scores(ctx, pcm, True, threshold=t)
            assert np.array_equal(modes_spec, modes_exact), (name, q, t)
            assert opened > 0                     # a score equal or next to the threshold is never decided speculatively


@pytest.mark.parametrize('thr', [0.01, 0.05, 2.0])
def test_threshold_at_the_ends_of_its_range(ctx, thr):
    """options.js:101-105 allows 0.01 .. 2: nearly everything / nearly nothing is a transient"""
    for name in ('pink_bursts', 'castanets', 'sine_1k', 'silence', 'tiny'):
        pcm = dict(ALL)[name]
        _, modes_exact, _ = scores(ctx, pcm, False, threshold=thr)
        _, modes_spec, _ = scores(ctx, pcm, True, threshold=thr)
        assert np.array_equal(modes_spec, modes_exact), (name, thr)


def test_halo_frames_and_stream_start(ctx):
    pcm = dict(ALL)['castanets']
    for halo in (0, 1, 2):
        exact, modes_exact, _ = scores(ctx, pcm, False, halo=halo)
        spec, modes_spec, _ = scores(ctx, pcm, True, halo=halo)
        s, lo, hi = exact[:, :, 0], spec[:, :, 0], spec[:, :, 1]
        assert ((lo <= s) & (s <= hi)).all()
        assert np.array_equal(modes_spec, modes_exact)


@pytest.mark.parametrize('name,pcm', ALL, ids=[s[0] for s in ALL])
def test_encode_with_the_speculative_detector_is_bit_identical(ctx, name, pcm):
    import carta1_amd as c1
    other = O.gen_pinkT(4, len(pcm))
    opts = c1.EncoderOptions({})
    ctx.set_speculation(0)
    exact = ctx.encode([pcm, other], opts).copy()
    ctx.set_speculation(2)
    ctx.speculation_stats(reset=True)
    spec = ctx.encode([pcm, other], opts).copy()
    units, rechecked = ctx.detection_stats()
    ctx.set_speculation(1)
    assert units == exact.shape[0] and rechecked <= units
    assert np.array_equal(exact, spec), np.nonzero((exact != spec).any(axis=1))[0][:8]
    if name in ('white', 'pink_bursts', 'sine_1k', 'impulses', 'silence', 'castanets'):
        want, _ = O.encode_stream([pcm[:32 * 512], other[:32 * 512]])
        assert np.array_equal(spec[:64], want)


def test_non_finite_samples_go_to_the_exact_detector(ctx):
    import carta1_amd as c1
    n = 64 * 512
    for bad in (np.inf, -np.inf, np.nan):
        x = O.gen_white(3, n)
        x[10 * 512 + 77] = bad
        y = O.gen_pinkT(4, n)
        opts = c1.EncoderOptions({})
        ctx.set_speculation(0)
        exact = ctx.encode([x, y], opts).copy()
        ctx.set_speculation(2)
        ctx.speculation_stats(reset=True)
        spec = ctx.encode([x, y], opts).copy()
        units, rechecked = ctx.detection_stats()
        ctx.set_speculation(1)
        assert np.array_equal(exact, spec)
        assert rechecked >= 2                      # the frame holding it and the next one, at least
