"""GPU: BASELINE configs[3]'s synthetic corpus (C1_SIGNAL_MIXED: 512-frame segments cycling white noise, pink noise with
bursts, stationary partials, quiet white noise) and its tonal segment kind alone (C1_SIGNAL_PARTIALS), generated on the
device as bench.py generates them.  The PCM is copied back, so the oracle runs on the very samples the device holds:
 * units bit-identical to the oracle over windows that span all four segment kinds, with the exact kernels only, with the
   material-local speculation of the default mode and with speculation forced;
 * the default mode keeps the noise segments on the speculative kernels and hands the tonal ones to the exact kernels;
 * at the size of one chunk: default mode == exact kernels byte for byte, a window deep inside == the oracle from its halo."""
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


def generate(ctx, signal, seeds, frames):
    import torch
    pcm = [torch.empty(frames * 512, dtype=torch.float32, device='cuda') for _ in seeds]
    torch.cuda.synchronize()
    for p, seed in zip(pcm, seeds):
        ctx.generate_device(signal, seed, frames, p.data_ptr())
    ctx.synchronize()
    return pcm


def encode_device(ctx, pcm, frames, opts, first=0, halo=0):
    import torch
    units = torch.empty(frames * len(pcm) * 212, dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize()
    ctx.encode_device([p.data_ptr() + first * 2048 for p in pcm], frames, units.data_ptr(), opts, halo_frames=halo)
    ctx.synchronize()
    return units


@pytest.mark.parametrize('modes', [(0, 0, 0), (2, 2, 3)], ids=['long', 'short'])
def test_mixed_corpus_window_over_all_segment_kinds_equals_the_oracle(ctx, modes):
    import carta1_amd as c1
    frames = 4096                                              # two cycles of the four segment kinds, from a segment boundary
    pcm = generate(ctx, c1.SIGNAL_MIXED, (5, 6), frames)
    host = [p.cpu().numpy() for p in pcm]
    want, _ = O.encode_stream(host, fixed_modes=modes)
    opts = c1.EncoderOptions({'fixedBlockModes': list(modes)})
    got = {}
    for mode in (0, 1, 2):
        ctx.set_speculation(mode)
        ctx.speculation_stats(reset=True)
        got[mode] = encode_device(ctx, pcm, frames, opts).cpu().numpy().reshape(-1, 212)
        u, r = ctx.speculation_stats()
        d = ctx.speculation_deferred()
        assert np.array_equal(got[mode], want), (mode, np.nonzero((got[mode] != want).any(axis=1))[0][:8])
        if mode == 0:
            assert (u, r, d) == (0, 0, 0)
        if mode == 1:
            # of 8 segments x 2 channels the 4 tonal ones (2 x 512 frames x 2 channels) go to the exact kernels, give or
            # take the runs at their borders; what stays speculative is noise, of which pink + bursts redoes the most
            assert u + d == 2 * frames and 2048 - 4 * 16 <= d <= 2048 + 4 * 64, (u, r, d)
            assert r < 0.15 * u, (u, r)
        if mode == 2:
            assert u == 2 * frames and d == 0 and r > 0.25 * u, (u, r, d)      # forced: the tonal quarter is redone unit by unit
    ctx.set_speculation(1)


def test_partials_corpus_equals_the_oracle_and_is_left_to_the_exact_kernels(ctx):
    import carta1_amd as c1
    frames = 1536                                              # three segments: three different sets of partials
    pcm = generate(ctx, c1.SIGNAL_PARTIALS, (7, 8), frames)
    host = [p.cpu().numpy() for p in pcm]
    want, _ = O.encode_stream(host, fixed_modes=(0, 0, 0))
    opts = c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]})
    for mode in (0, 1, 2):
        ctx.set_speculation(mode)
        ctx.speculation_stats(reset=True)
        got = encode_device(ctx, pcm, frames, opts).cpu().numpy().reshape(-1, 212)
        assert np.array_equal(got, want), mode
        if mode == 1:
            assert ctx.speculation_deferred() >= 0.95 * 2 * frames
    ctx.set_speculation(1)
    # and through detection (the exact analysis with the speculative detector)
    want_d, _ = O.encode_stream([h[:256 * 512] for h in host])
    got_d = encode_device(ctx, [p[:256 * 512] for p in pcm], 256, c1.EncoderOptions()).cpu().numpy().reshape(-1, 212)
    assert np.array_equal(got_d, want_d)


def test_mixed_corpus_at_chunk_size(ctx):
    """1 M stereo frames of the corpus (one chunk of the library): default mode == exact kernels byte for byte; a window deep
    inside the batch that crosses a tonal -> noise border equals the oracle run from its one-frame halo; a slice encoded on
    its own from its halo equals the same frames of the whole batch (what frame-batch sharding relies on)."""
    import torch
    import carta1_amd as c1
    frames = 1 << 20
    pcm = generate(ctx, c1.SIGNAL_MIXED, (5, 6), frames)
    opts = c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]})
    ctx.set_speculation(0)
    exact = encode_device(ctx, pcm, frames, opts)
    ctx.set_speculation(1)
    ctx.speculation_stats(reset=True)
    local = encode_device(ctx, pcm, frames, opts)
    u, r = ctx.speculation_stats()
    d = ctx.speculation_deferred()
    assert torch.equal(exact, local)
    assert u + d == 2 * frames and 0.22 * 2 * frames < d < 0.30 * 2 * frames, (u, r, d)   # a quarter of the corpus is tonal
    assert r < 0.12 * u, (u, r)
    a = 700 * 512 + 3 * 512 - 40                               # 40 frames before the border partials -> quiet white of cycle 175
    host = [p[(a - 1) * 512:(a + 200) * 512].cpu().numpy() for p in pcm]
    st = (O.EncState * 2)()
    w, _ = O.encode_stream(host, fixed_modes=(0, 0, 0), states=st)
    got = local[a * 424:(a + 200) * 424].cpu().numpy().reshape(-1, 212)
    assert np.array_equal(got, w[2:])
    part = encode_device(ctx, pcm, 3000, opts, first=a, halo=1)
    assert torch.equal(part, local[a * 424:(a + 3000) * 424])
    del exact, local, pcm
