"""GPU: the speculative binary32 path (c1_k_spec.hip, k_pack<.., SPEC>) -- DESIGN.md 3b.
 1. the kernel computes what its CPU model computes (tests/model/spec_model.c);
 2. its coefficients lie within its own bound of the reference's (oracle);
 3. encoding with speculation forced on is bit-identical to the exact kernels and to the oracle, whatever fraction of
    the units had to be redone, and the redo fraction is what DESIGN.md reports."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import spec_model_lib as M
from test_spec_bound import BAND, signals

pytestmark = pytest.mark.gpu
LONG = {'fixedBlockModes': [0, 0, 0]}


@pytest.fixture(scope='module')
def ctx():
    import carta1_amd as c1
    c = c1.Context(0)
    yield c
    c.close()


def spec_stages(ctx, chans, modes=(0, 0, 0)):
    import torch
    import carta1_amd as c1
    frames = len(chans[0]) // 512
    dev = [torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in chans]
    n = frames * len(chans)
    coefs = torch.zeros(n * 512, dtype=torch.float32, device='cuda')
    eps = torch.zeros(n * 4, dtype=torch.float32, device='cuda')
    side = torch.zeros(n * 64, dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize()
    ctx.spec_stages_device([d.data_ptr() for d in dev], frames, coefs.data_ptr(), eps.data_ptr(), side.data_ptr(),
                           c1.EncoderOptions({'fixedBlockModes': list(modes)}))
    ctx.synchronize()
    return (coefs.cpu().numpy().reshape(frames, len(chans), 512), eps.cpu().numpy().reshape(frames, len(chans), 4),
            side.cpu().numpy().reshape(frames, len(chans), 64))


def run_length(frames, channels):
    """c1k_pick_run (c1_internal.h): 64-frame runs, shorter ones for small batches (latency of a streaming push)"""
    import os
    if int(os.environ.get('C1_RUN_FRAMES', '0')) > 0:
        return int(os.environ['C1_RUN_FRAMES'])
    units = frames * channels
    return 64 if units >= 64 * 2048 else max(4, -(-units // 2048))


@pytest.mark.parametrize('modes', [(0, 0, 0), (2, 2, 3)], ids=['long', 'short'])
@pytest.mark.parametrize('name,pcm', list(signals()), ids=[s[0] for s in signals()])
def test_kernel_equals_its_model_and_stays_within_its_bound(ctx, name, pcm, modes):
    other = O.gen_white(9, len(pcm))
    short = modes != (0, 0, 0)
    co, eps, side = spec_stages(ctx, [pcm, other], modes)
    mco, meps, _ = M.run(pcm, short)
    # frames are processed in runs of 64 with one warm-up frame: the first frame of a later run has seen one frame of
    # history where the model has seen the whole stream, which is the same for these feed-forward filters
    assert np.array_equal(co[:, 0], mco), np.argwhere(co[:, 0] != mco)[:4]
    # the bound is the model's too, except in the first frame of a run (run lengths depend on the batch size): there the
    # energy of the previous frame's first-stage low band comes from the warm-up frame, whose first 23 outputs saw an
    # empty delay line (they are outside the reach of the unit, so the bound holds with either value)
    ok = np.isfinite(meps)
    close = np.isclose(eps[:, 0, :3], meps, rtol=2e-6, atol=0) | ~ok
    run = run_length(co.shape[0], 2)
    inside = np.arange(co.shape[0]) % run != 0
    assert close.all(axis=1)[inside].all()
    assert np.allclose(eps[:, 0, :3][ok], meps[ok], rtol=0.2, atol=0)
    ref = M.reference_coefs(pcm, modes)
    err = np.abs(co[:, 0].astype(np.float64) - ref.astype(np.float64))
    assert (err <= eps[:, 0, :3][:, BAND]).all()
    assert np.array_equal(co[:, 1], M.run(other, short)[0])
    assert (side[:, :, 52] == (modes[0] | modes[1] << 2 | modes[2] << 4)).all()


def encode_both_ways(ctx, chans, opts):
    ctx.set_speculation(0)
    exact = ctx.encode(chans, opts).copy()
    ctx.set_speculation(2)
    ctx.speculation_stats(reset=True)
    spec = ctx.encode(chans, opts).copy()
    units, redone = ctx.speculation_stats()
    ctx.set_speculation(1)
    return exact, spec, units, redone


@pytest.mark.parametrize('modes', [(0, 0, 0), (2, 2, 3)], ids=['long', 'short'])
@pytest.mark.parametrize('name,pcm', list(signals()), ids=[s[0] for s in signals()])
def test_speculative_encode_is_bit_identical(ctx, name, pcm, modes):
    import carta1_amd as c1
    other = O.gen_pinkT(5, len(pcm))
    opts = c1.EncoderOptions({'fixedBlockModes': list(modes)})
    exact, spec, units, redone = encode_both_ways(ctx, [pcm, other], opts)
    assert units == exact.shape[0]
    assert np.array_equal(exact, spec), np.nonzero((exact != spec).any(axis=1))[0][:8]
    want, _ = O.encode_stream([pcm, other], fixed_modes=modes)
    assert np.array_equal(spec, want)


@pytest.mark.parametrize('modes', [(0, 0, 0), (2, 2, 3)], ids=['long', 'short'])
@pytest.mark.parametrize('bias', [0.5, 1.0, 2.0])
def test_white_noise_redo_fraction_and_identity(ctx, bias, modes):
    import carta1_amd as c1
    n = 4096 * 512
    chans = [O.gen_white(1, n), O.gen_white(2, n)]
    opts = c1.EncoderOptions({'fixedBlockModes': list(modes), 'allocationBias': bias}, biased_table=O.biased_table(bias))
    exact, spec, units, redone = encode_both_ways(ctx, chans, opts)
    assert np.array_equal(exact, spec)
    assert units == 8192 and 0 < redone < 0.12 * units, (units, redone)      # ~5 % on white noise
    want, _ = O.encode_stream([c[:512 * 512] for c in chans], fixed_modes=modes, bias=bias)
    assert np.array_equal(spec[:1024], want)


def tone(frames, f=1000.0, amp=0.5, f2=None):
    t = np.arange(frames * 512)
    x = amp * np.sin(2 * np.pi * f * t / 44100)
    if f2:
        x = x + 0.2 * amp * np.sin(2 * np.pi * f2 * t / 44100)
    return x.astype(np.float32)


def test_material_local_mode_hands_tonal_runs_to_the_exact_kernels(ctx):
    """default mode: the speculative kernel's predictor rejects tonal material run by run"""
    import carta1_amd as c1
    x = tone(256)
    opts = c1.EncoderOptions(LONG)
    ctx.set_speculation(0)
    exact = ctx.encode([x], opts).copy()
    ctx.set_speculation(1)
    ctx.speculation_stats(reset=True)
    a = ctx.encode([x], opts).copy()
    u, r = ctx.speculation_stats()
    d = ctx.speculation_deferred()
    assert d >= 256 - 16 and u + d == 256, (u, r, d)                      # every run left at its first check (the onset
                                                                          # from silence may pass the first one)
    assert np.array_equal(a, exact)
    want, _ = O.encode_stream([x], fixed_modes=(0, 0, 0))
    assert np.array_equal(a, want)
    # forced speculation on the same stream: nearly every unit fails the guard band and is redone
    ctx.set_speculation(2)
    ctx.speculation_stats(reset=True)
    b = ctx.encode([x], opts).copy()
    u, r = ctx.speculation_stats()
    assert u == 256 and r > 0.5 * u and ctx.speculation_deferred() == 0
    assert np.array_equal(b, exact)
    ctx.set_speculation(1)


@pytest.mark.parametrize('modes', [(0, 0, 0), (2, 2, 3)], ids=['long', 'short'])
def test_material_local_mode_decides_per_run_and_carries_nothing_over(ctx, modes):
    """two channels alternate noise and tones in opposite phase, 128 frames each: the tonal runs -- and only they -- go
    to the exact kernels, whatever the other channel or the previous call held"""
    import carta1_amd as c1
    seg = 128
    w1, w2 = O.gen_white(11, seg * 512), O.gen_pinkT(12, seg * 512)
    t1, t2 = tone(seg, 440.0, 0.4, 3520.0), tone(seg, 1234.5, 0.7)
    ch0 = np.concatenate([w1, t1, w2, t2])
    ch1 = np.concatenate([t2, w2, t1, w1])
    opts = c1.EncoderOptions({'fixedBlockModes': list(modes)})
    ctx.set_speculation(1)
    ctx.speculation_stats(reset=True)
    got = ctx.encode([ch0, ch1], opts).copy()
    u, r = ctx.speculation_stats()
    d = ctx.speculation_deferred()
    want, _ = O.encode_stream([ch0, ch1], fixed_modes=modes)
    assert np.array_equal(got, want), np.nonzero((got != want).any(axis=1))[0][:8]
    # segment boundaries are run boundaries here.  The four tonal segments are deferred -- a tone's onset frame is
    # broadband, so a segment may be left at its second check (16 frames late), and the first run of a noise segment
    # right behind a tone may still be handed over (its first unit holds the tone's windowed tail)
    assert u + d == 8 * seg and 4 * seg - 4 * 16 <= d <= 4 * seg + 4 * 64, (d, u, r)
    assert r < 0.35 * u, (u, r)                                  # of the units that WERE speculated few are redone
    uq, rq = ctx.quantization_stats()
    assert uq >= d                                               # the deferred runs are quantized in binary32 behind a bound of zero
    # nothing is carried from call to call: noise right after tones defers nothing, tones after noise (nearly) everything
    ctx.speculation_stats(reset=True)
    ctx.encode([t1], opts)
    d1 = ctx.speculation_deferred()
    assert seg - 16 <= d1 <= seg
    ctx.encode([w1], opts)
    assert ctx.speculation_deferred() == d1
    ctx.encode([t2], opts)
    assert seg - 16 <= ctx.speculation_deferred() - d1 <= seg


def test_material_change_inside_a_run(ctx):
    """material that turns tonal (or stops being tonal) in the middle of a 64-frame run: the run is handed over at the next
    16-frame check and stays with the exact kernels until it ends; bytes never change"""
    import carta1_amd as c1
    x = np.concatenate([O.gen_white(21, 100 * 512), tone(100, 2000.0, 0.6), O.gen_white(22, 100 * 512)])
    opts = c1.EncoderOptions(LONG)
    ctx.set_speculation(1)
    ctx.speculation_stats(reset=True)
    got = ctx.encode([x], opts).copy()
    d = ctx.speculation_deferred()
    want, _ = O.encode_stream([x], fixed_modes=(0, 0, 0))
    assert np.array_equal(got, want)
    # tones occupy frames 100..199.  A run is left at its first check (its first frame, then every 16th) that looks at a
    # tonal frame, and stays with the exact kernels to its end although the tone may have stopped: with 64-frame runs,
    # run 64..127 at frame 112, run 128..191 at 128, run 192..255 at 192; the frames at the two edges may go either way
    run = run_length(300, 1)

    def handed_over(first, last):
        total = 0
        for start in range(0, 300, run):
            end = min(start + run, 300)
            hit = [c for c in range(start, end, 16) if first <= c <= last]
            if hit:
                total += end - hit[0]
        return total
    assert handed_over(102, 197) <= d <= handed_over(99, 200), (d, run)


def test_halo_and_unaligned_runs(ctx):
    """slices of a stream with their halo frames, lengths that are not multiples of the 64-frame run"""
    import carta1_amd as c1
    n = 300 * 512
    chans = [O.gen_white(3, n), O.gen_pinkT(6, n)]
    opts = c1.EncoderOptions(LONG)
    ctx.set_speculation(2)
    whole = ctx.encode(chans, opts).copy()
    for f0, f1, halo in ((1, 66, 1), (63, 131, 2), (129, 300, 1), (0, 1, 0)):
        part = ctx.encode([c[(f0 - halo) * 512:f1 * 512] for c in chans], opts, halo_frames=halo)
        assert np.array_equal(part, whole[2 * f0:2 * f1]), (f0, f1, halo)
    ctx.set_speculation(1)


@pytest.mark.parametrize('frames,channels', [(2049, 1), (6143, 1), (6145, 2), (20481, 1), (33000, 2), (65537, 1)])
def test_run_lengths_of_mid_sized_batches(ctx, frames, channels):
    """c1k_pick_run spreads a batch of fewer than 131 072 units over up to 2 048 waves: run lengths 4..64 that do not
    divide the batch, in every path that walks runs (speculative, material-local with a tonal stretch across run ends,
    exact, detection; the decoder).  Bytes equal across the modes and equal the oracle's on the first 1 500 frames; the
    decoded PCM equals the oracle's there."""
    import carta1_amd as c1
    n = frames * 512
    chans = [np.concatenate([O.gen_white(41 + c, (frames // 3) * 512), tone(frames // 3, 1500.0 + 300 * c, 0.5),
                             O.gen_pinkT(43 + c, n - 2 * (frames // 3) * 512)]) for c in range(channels)]
    assert 4 < run_length(frames, channels) <= 64 or frames * channels < 4 * 2048
    head = 1500
    for label, o, fm in (('long', LONG, (0, 0, 0)), ('short', {'fixedBlockModes': [2, 2, 3]}, (2, 2, 3)), ('mixed', {'fixedBlockModes': [0, 2, 0]}, (0, 2, 0)), ('detect', {}, None)):
        opts = c1.EncoderOptions(o)
        got = {}
        for mode in (0, 1, 2):
            ctx.set_speculation(mode)
            got[mode] = ctx.encode(chans, opts).copy()
        ctx.set_speculation(1)
        assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2]), label
        want, _ = O.encode_stream([c[:head * 512] for c in chans], fixed_modes=fm)
        assert np.array_equal(got[0][:head * channels], want), label
    pcm = ctx.decode(got[0], channels)
    want_pcm, _ = O.decode_stream(want, channels)
    for c in range(channels):
        assert np.array_equal(pcm[c][:(head - 1) * 512], want_pcm[c][:(head - 1) * 512])
    # the decoder's runs: a slice that starts inside the batch, from the unit before it
    part = ctx.decode(got[0][(777 - 1) * channels:], channels, halo_units=1)
    for c in range(channels):
        assert np.array_equal(part[c], pcm[c][777 * 512:])


def test_tail_overlap_mode_is_bit_identical_and_keeps_stream_order():
    """C1_OVERLAP=1: the exact redo of a chunk runs on a second stream beside the next chunk's / call's analysis.  Several
    device encodes enqueued back to back without synchronising -- into separate buffers, and into ONE buffer reused call
    after call (the later call's bytes must stand) -- then one synchronise: every result equals the in-line mode's."""
    import os
    import torch
    import carta1_amd as c1
    frames = 5000
    srcs = [[O.gen_white(31, frames * 512), O.gen_pinkT(32, frames * 512)],
            [O.gen_pinkT(33, frames * 512), O.gen_white(34, frames * 512)],
            [np.concatenate([O.gen_white(35, 2500 * 512), tone(2500, 700.0, 0.5)]), O.gen_white(36, frames * 512)]]
    opts = c1.EncoderOptions(LONG)
    plain = c1.Context(0)
    want = [plain.encode(s, opts).copy() for s in srcs]
    plain.close()
    os.environ['C1_OVERLAP'] = '1'
    os.environ['C1_CHUNK_FRAMES'] = '1024'                    # several chunks per call: the tails overlap inside a call too
    try:
        ctx = c1.Context(0)
    finally:
        del os.environ['C1_OVERLAP'], os.environ['C1_CHUNK_FRAMES']
    dev = [[torch.from_numpy(ch).cuda() for ch in s] for s in srcs]
    outs = [torch.zeros(frames * 2 * 212, dtype=torch.uint8, device='cuda') for _ in srcs]
    shared = torch.zeros(frames * 2 * 212, dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize()
    for rep in range(3):
        for d, o in zip(dev, outs):
            ctx.encode_device([t.data_ptr() for t in d], frames, o.data_ptr(), opts)
        for d in dev:                                          # the same output buffer, call after call: the last one wins
            ctx.encode_device([t.data_ptr() for t in d], frames, shared.data_ptr(), opts)
    ctx.synchronize()
    for o, w in zip(outs, want):
        assert np.array_equal(o.cpu().numpy().reshape(-1, 212), w)
    assert np.array_equal(shared.cpu().numpy().reshape(-1, 212), want[-1])
    # the host-resident entry points join the tail themselves
    assert np.array_equal(ctx.encode(srcs[0], opts), want[0])
    u, r = ctx.speculation_stats()
    assert u > 0 and r > 0
    ctx.close()


def test_device_encode_only_enqueues():
    """c1_encode_device never waits for the device (round 2's adaptive mode read counters back on entry): three 262144-frame
    encodes, each some 2 ms of kernels, are enqueued in a fraction of the time the device needs for one of them -- in the
    default mode, on tonal material too (where round 2 looked at the previous call's redo fraction), and the results are
    those of calls made one by one"""
    import time
    import torch
    import carta1_amd as c1
    frames = 262144
    ctx = c1.Context(0)
    opts = c1.EncoderOptions(LONG)
    pcm = [torch.empty(frames * 512, dtype=torch.float32, device='cuda') for _ in range(2)]
    torch.cuda.synchronize()
    ctx.generate_device(c1.SIGNAL_MIXED, 5, frames, pcm[0].data_ptr())
    ctx.generate_device(c1.SIGNAL_PARTIALS, 6, frames, pcm[1].data_ptr())
    outs = [torch.zeros(frames * 2 * 212, dtype=torch.uint8, device='cuda') for _ in range(3)]
    ptrs = [p.data_ptr() for p in pcm]
    ctx.encode_device(ptrs, frames, outs[0].data_ptr(), opts)      # first call: workspace allocation, options upload
    ctx.synchronize()
    want = outs[0].clone()
    t0 = time.perf_counter()
    ctx.encode_device(ptrs, frames, outs[0].data_ptr(), opts)
    ctx.synchronize()
    one = time.perf_counter() - t0
    t0 = time.perf_counter()
    for o in outs:
        ctx.encode_device(ptrs, frames, o.data_ptr(), opts)
    enqueue = time.perf_counter() - t0
    ctx.synchronize()
    total = time.perf_counter() - t0
    assert enqueue < 0.5 * one, (enqueue, one, total)               # three calls enqueued in less than half of one call's run time
    assert total > 2.0 * one                                       # ... and the device did need its time for them
    for o in outs:
        assert torch.equal(o, want)
    ctx.close()
