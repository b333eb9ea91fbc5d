"""GPU parity: the HIP path (through the C ABI) against the reference's golden vectors and against the
CPU oracle on the same seeded inputs.  Bit-exact on units (integer/byte work) and on decoded PCM."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
INDEX = json.load(open(os.path.join(G, 'kat_index.json')))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope='module')
def ctx():
    import carta1_amd as c1
    c = c1.Context(0)
    yield c
    c.close()


def make_input(case, frames):
    n = frames * 512
    if case['signal'] == 'white':
        return [O.gen_white(1, n), O.gen_white(2, n)]
    return [O.gen_pinkT(3, n), O.gen_pinkT(4, n)]


def options_of(case):
    import carta1_amd as c1
    o = dict(case['options'])
    bias = o.get('allocationBias', 1.0)
    return c1.EncoderOptions(o, biased_table=O.biased_table(bias))


def first_diff(a, b):
    bad = np.nonzero((a != b).any(axis=1))[0]
    return bad[:8]


def test_config1_known_answer(ctx):
    import carta1_amd as c1
    k = json.load(open(os.path.join(G, 'config1_sine1k.json')))
    i = np.arange(512)
    pcm = np.sin(2 * np.pi * 1000 * i / 44100).astype(np.float32)
    units = ctx.encode([pcm], c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]}))
    assert units[0].tobytes().hex() == k['unit_hex']
    f = c1.codec.deserialize_frame(units[0])
    assert f['nBfu'] == 44 and f['wordLengthIndices'] == k['wordLengthIndices']
    assert f['scaleFactorIndices'] == k['scaleFactorIndices']
    assert f['quantizedCoefficients'] == k['quantizedCoefficients']
    pcm_out = ctx.decode(units, 1)
    assert sha(pcm_out[0]) == k['decoded_first_frame_sha256']


@pytest.mark.parametrize('name', sorted(INDEX))
def test_kat64_against_reference_fixtures(ctx, name):
    case = INDEX[name]
    chs = make_input(case, 64)
    units = ctx.encode(chs, options_of(case))
    ref_units = np.fromfile(os.path.join(G, 'kat64_%s.units.bin' % name), dtype=np.uint8).reshape(-1, 212)
    assert first_diff(units, ref_units).size == 0, 'units differ at %s' % first_diff(units, ref_units)
    assert sha(units) == case['units_sha256']
    pcm = ctx.decode(ref_units, 2)
    head = np.fromfile(os.path.join(G, 'kat64_%s.pcm8.bin' % name), dtype=np.float32)
    assert np.array_equal(pcm[0][:4096].view(np.uint32), head[:4096].view(np.uint32))
    assert np.array_equal(pcm[1][:4096].view(np.uint32), head[4096:].view(np.uint32))
    assert sha(pcm[0]) == case['decoded_planar_L_sha256']
    assert sha(pcm[1]) == case['decoded_planar_R_sha256']


@pytest.mark.parametrize('name', sorted(INDEX))
def test_long2048_against_reference_hashes(ctx, name):
    case = INDEX[name]
    lg = case['long2048']
    chs = make_input(case, lg['frames'])
    units = ctx.encode(chs, options_of(case))
    modes = np.fromfile(os.path.join(G, 'long2048_%s.modes.bin' % name), dtype=np.uint8).astype(int)
    hdr = units[:, 0].astype(int)
    got = (2 - (hdr >> 6 & 3)) | ((2 - (hdr >> 4 & 3)) << 2) | ((3 - (hdr >> 2 & 3)) << 4)
    assert np.array_equal(got, modes), 'block modes differ at units %s' % np.nonzero(got != modes)[0][:8]
    assert sha(units) == lg['units_sha256']
    pcm = ctx.decode(units, 2)
    assert sha(pcm[0]) == lg['decoded_planar_L_sha256']
    assert sha(pcm[1]) == lg['decoded_planar_R_sha256']


def test_stage_taps_against_reference(ctx):
    import torch
    import carta1_amd as c1
    pcm = O.gen_white(1, 3 * 512)
    ref_bands = np.fromfile(os.path.join(G, 'stages_white1_bands.f32.bin'), dtype=np.float32)
    d_pcm = torch.from_numpy(pcm).cuda()
    for tag, modes in (('m000', [0, 0, 0]), ('m223', [2, 2, 3])):
        ref_coefs = np.fromfile(os.path.join(G, 'stages_white1_%s_coefs.f32.bin' % tag), dtype=np.float32)
        ref_alloc = json.load(open(os.path.join(G, 'stages_white1_%s_alloc.json' % tag)))
        bands = torch.zeros(3 * 512, dtype=torch.float32, device='cuda')
        coefs = torch.zeros(3 * 512, dtype=torch.float32, device='cuda')
        side = torch.zeros(3 * 64, dtype=torch.uint8, device='cuda')
        alloc = torch.zeros(3 * 32, dtype=torch.uint8, device='cuda')
        torch.cuda.synchronize()   # the context runs on its own stream
        ctx.encode_stages_device([d_pcm.data_ptr()], 3, bands.data_ptr(), coefs.data_ptr(), side.data_ptr(),
                                 alloc.data_ptr(), c1.EncoderOptions({'fixedBlockModes': modes}))
        ctx.synchronize()
        assert np.array_equal(bands.cpu().numpy().view(np.uint32), ref_bands.view(np.uint32))
        assert np.array_equal(coefs.cpu().numpy().view(np.uint32), ref_coefs.view(np.uint32))
        side_h = side.cpu().numpy().reshape(3, 64)
        alloc_h = alloc.cpu().numpy().reshape(3, 32)
        for f in range(3):
            assert list(side_h[f, :52]) == ref_alloc[f]['sfi']
            words = alloc_h[f].view(np.uint64)
            n = c1.codec.BFU_AMOUNTS[int(words[3] >> np.uint64(60)) & 7]
            wl = [int((int(words[b >> 4]) >> ((b & 15) * 4)) & 15) for b in range(n)]
            assert n == ref_alloc[f]['nBfu'] and wl == ref_alloc[f]['wl']


def _tonal(frames, seed):
    # mixture of stationary sines + slow AM: the input class where float32 arithmetic flips 1 % of frames
    rng = np.random.RandomState(seed)
    t = np.arange(frames * 512, dtype=np.float64)
    x = np.zeros_like(t)
    for _ in range(6):
        f, a, ph = rng.uniform(50, 18000), rng.uniform(0.02, 0.3), rng.uniform(0, 6.28)
        x += a * np.sin(2 * np.pi * f * t / 44100 + ph) * (1 + 0.3 * np.sin(2 * np.pi * rng.uniform(0.1, 3) * t / 44100))
    return x.astype(np.float32)


@pytest.mark.parametrize('modes,bias', [([0, 0, 0], 1.0), ([2, 2, 3], 1.0), ([0, 2, 0], 2.0), (None, 1.0), ([2, 0, 3], 0.5)])
def test_tonal_stereo_against_oracle(ctx, modes, bias):
    import carta1_amd as c1
    frames = 3000
    chs = [_tonal(frames, 5), _tonal(frames, 6)]
    opt = {'allocationBias': bias}
    if modes:
        opt['fixedBlockModes'] = modes
    want, _ = O.encode_stream(chs, fixed_modes=modes, bias=bias, threshold=1.0)
    got = ctx.encode(chs, c1.EncoderOptions(opt, biased_table=O.biased_table(bias)))
    assert first_diff(got, want).size == 0, 'units differ at %s' % first_diff(got, want)
    pcm_want, _ = O.decode_stream(want, 2)
    pcm_got = ctx.decode(want, 2)
    for c in range(2):
        assert np.array_equal(pcm_got[c].view(np.uint32), pcm_want[c].view(np.uint32))


def test_halo_and_chunk_invariance(ctx):
    """Encoding a slice of a stream with its PCM halo gives the same units as encoding the whole stream:
    what frame-batch sharding across GPUs relies on."""
    import carta1_amd as c1
    frames = 400
    for modes, halo in (([0, 0, 0], 1), ([2, 2, 3], 1), (None, 2)):
        chs = [O.gen_pinkT(3, frames * 512), O.gen_pinkT(4, frames * 512)]
        opt = c1.EncoderOptions({'fixedBlockModes': modes} if modes else {})
        whole = ctx.encode(chs, opt).reshape(frames, 2, 212)
        for a, b in ((0, 57), (57, 58), (58, 333), (333, 400)):
            h = min(halo, a)
            part = ctx.encode([c[(a - h) * 512:b * 512] for c in chs], opt, halo_frames=h).reshape(b - a, 2, 212)
            assert np.array_equal(part, whole[a:b]), (modes, a, b)
    # decode: one unit of history is enough
    units = ctx.encode(chs, c1.EncoderOptions()).reshape(frames, 2, 212)
    full = ctx.decode(units.reshape(-1, 212), 2)
    for a, b in ((0, 33), (33, 34), (34, 400)):
        h = min(1, a)
        part = ctx.decode(units[a - h:b].reshape(-1, 212), 2, halo_units=h)
        for c in range(2):
            assert np.array_equal(part[c].view(np.uint32), full[c][a * 512:b * 512].view(np.uint32)), (a, b)


def test_streams_continue_bit_for_bit(ctx):
    import carta1_amd as c1
    frames = 70
    x = O.gen_pinkT(9, frames * 512)
    opt = c1.EncoderOptions()
    whole = ctx.encode([x], opt)
    s = c1.EncoderStream(ctx, 1, opt)
    parts = []
    pos = 0
    for n in (1, 1, 1, 5, 17, 1, 44):
        parts.append(s.push([x[pos * 512:(pos + n) * 512]]))
        pos += n
    s.close()
    assert np.array_equal(np.concatenate(parts), whole)
    full = ctx.decode(whole, 1)[0]
    d = c1.DecoderStream(ctx, 1)
    outs, pos = [], 0
    for n in (1, 2, 1, 30, 36):
        outs.append(d.push(whole[pos:pos + n])[0])
        pos += n
    d.close()
    assert np.array_equal(np.concatenate(outs).view(np.uint32), full.view(np.uint32))


def test_decode_arbitrary_bytes_matches_oracle(ctx):
    rng = np.random.RandomState(7)
    units = rng.randint(0, 256, size=(600, 212)).astype(np.uint8)
    want, _ = O.decode_stream(units, 2)
    got = ctx.decode(units, 2)
    for c in range(2):
        assert np.array_equal(got[c].view(np.uint32), want[c].view(np.uint32))


def test_decode_without_the_step_form_of_dequantize():
    """the reciprocal + two-FMA form of dequantize (quantization.js:65-78) that the per-BFU step form shadows with the
    reference's tables: same PCM, bit for bit, on arbitrary units and on an encoded stream"""
    import carta1_amd as c1
    os.environ['C1_NO_DQ_STEP'] = '1'
    try:
        c = c1.Context(0)                       # tables are built when the context is created
    finally:
        del os.environ['C1_NO_DQ_STEP']
    try:
        rng = np.random.RandomState(78)
        units = rng.randint(0, 256, size=(300 * 2, 212)).astype(np.uint8)
        want, _ = O.decode_stream(units, 2)
        got = c.decode(units, 2)
        for ch in range(2):
            assert np.array_equal(got[ch].view(np.uint32), want[ch].view(np.uint32))
        chs = [O.gen_pinkT(3, 200 * 512), O.gen_white(2, 200 * 512)]
        enc, _ = O.encode_stream(chs)
        want, _ = O.decode_stream(enc, 2)
        got = c.decode(enc, 2)
        for ch in range(2):
            assert np.array_equal(got[ch].view(np.uint32), want[ch].view(np.uint32))
    finally:
        c.close()


def test_edge_cases_against_reference(ctx):
    import carta1_amd as c1
    e = json.load(open(os.path.join(G, 'aea_edge_cases.json')))['cases']
    units = c1.encode_pcm([O.gen_white(11, 700), O.gen_white(12, 700)], c1.EncoderOptions(), ctx)
    assert units.tobytes().hex() == e['stereo700']['units_hex']
    pcm = ctx.decode(units, 2)
    assert [sha(pcm[0]), sha(pcm[1])] == e['stereo700']['decoded_sha256']
    units = c1.encode_pcm([O.gen_white(13, 1300), O.gen_white(14, 600)], c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]}), ctx)
    assert sha(units) == e['ragged_1300_600']['units_sha256']
    units = c1.encode_pcm([np.zeros(1024, np.float32)], c1.EncoderOptions(), ctx)
    assert units.tobytes().hex() == e['silence_mono_2frames']['units_hex']
    i = np.arange(2048)
    loud = (4.0 * np.sin((2 * np.pi * 440 * i) / 44100)).astype(np.float32)
    units = c1.encode_pcm([loud], c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]}), ctx)
    assert sha(units) == e['loud_sine_mono']['units_sha256']
    assert sha(ctx.decode(units, 1)[0]) == e['loud_sine_mono']['decoded_sha256']
    tiny = (O.gen_white(15, 1024).astype(np.float64) * 1e-36).astype(np.float32)
    units = c1.encode_pcm([tiny], c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]}), ctx)
    assert units.tobytes().hex() == e['tiny_mono']['units_hex']
    quiet = (O.gen_white(15, 1024).astype(np.float64) * 1e-6).astype(np.float32)
    units = c1.encode_pcm([quiet], c1.EncoderOptions(), ctx)
    assert sha(units) == e['quiet_mono']['units_sha256']
    assert sha(ctx.decode(units, 1)[0]) == e['quiet_mono']['decoded_sha256']
    # empty input
    assert c1.encode_pcm([np.zeros(0, np.float32)], c1.EncoderOptions(), ctx).shape == (0, 212)
    # huge and non-finite samples: the ToInt32 wrap path; must match the oracle exactly
    wild = O.gen_white(21, 4 * 512).copy()
    wild[100] = 3e9
    wild[700] = -1e30
    wild[1300] = 65504.0
    want, _ = O.encode_stream([wild], fixed_modes=(0, 0, 0))
    got = ctx.encode([wild], c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]}))
    assert np.array_equal(got, want)
    # the same through the exact paths, whose packing quantizes in binary32 behind a guard band and must hand
    # everything it cannot certify (and anything not finite) to the binary64 kernel
    ctx.set_speculation(2)
    before = ctx.quantization_stats()
    for opts, fm in (({'fixedBlockModes': [0, 2, 0]}, (0, 2, 0)), ({}, None), ({'transientThresholdLow': 0.1}, None)):
        want, _ = O.encode_stream([wild], fixed_modes=fm, threshold=opts.get('transientThresholdLow', 1.0))
        got = ctx.encode([wild], c1.EncoderOptions(opts))
        assert np.array_equal(got, want), opts
    after = ctx.quantization_stats()
    ctx.set_speculation(1)
    assert after[0] - before[0] == 3 * 4 and after[1] > before[1]            # 4 units per call; the wild ones were packed twice


def test_aea_round_trip_api(ctx):
    import carta1_amd as c1
    l, r = O.gen_white(31, 700), O.gen_white(32, 700)
    img = c1.encode_aea_pcm([l, r], {'title': 'x'}, ctx)
    assert len(img) == 2048 + 4 * 212
    info = c1.parse_aea_header(img[:2048])
    assert info == {'title': 'x', 'frameCount': 4, 'channelCount': 2}
    out = c1.decode_aea_pcm(img, ctx)
    assert len(out) == 2 and len(out[0]) == 1024 and len(out[1]) == 1024   # processor.test.js:95-108
    with pytest.raises(TypeError):
        c1.encode_aea_pcm([l.astype(np.float64)], {}, ctx)
    with pytest.raises(TypeError):
        c1.decode_aea_pcm(12345, ctx)


def test_device_generator_matches_host_generator(ctx):
    import torch
    import carta1_amd as c1
    frames = 1100
    buf = torch.zeros(frames * 512, dtype=torch.float32, device='cuda')
    torch.cuda.synchronize()
    ctx.generate_device(c1.SIGNAL_WHITE, 1, frames, buf.data_ptr())
    assert np.array_equal(buf.cpu().numpy().view(np.uint32), O.gen_white(1, frames * 512).view(np.uint32))
    ctx.generate_device(c1.SIGNAL_PINK_BURSTS, 3, frames, buf.data_ptr())
    got = buf.cpu().numpy()
    want = O.gen_pinkT(3, 512 * 512)
    assert np.array_equal(got[:512 * 512].view(np.uint32), want.view(np.uint32))   # segment 0 == BASELINE generator
    assert np.isfinite(got).all() and got[512 * 512:].std() > 0.01


# ---- formats either side of the path: WAV PCM <-> planar float32 on the device --------------

def test_wav16_output_matches_reference_blob_and_oracle(ctx):
    import torch
    w = json.load(open(os.path.join(G, 'wav16.json')))
    L = np.frombuffer(bytes.fromhex(w['left_f32']), dtype='<f4').copy()
    R = np.frombuffer(bytes.fromhex(w['right_f32']), dtype='<f4').copy()
    dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.zeros(len(L), dtype=torch.int16, device='cuda')
    ctx.pcm_to_int16_device([dl.data_ptr()], len(L), out.data_ptr())
    ctx.synchronize()
    assert out.cpu().numpy().astype('<i2').tobytes().hex() == w['mono_i16']
    out2 = torch.zeros(2 * len(L), dtype=torch.int16, device='cuda')
    ctx.pcm_to_int16_device([dl.data_ptr(), dr.data_ptr()], len(L), out2.data_ptr())
    ctx.synchronize()
    assert out2.cpu().numpy().astype('<i2').tobytes().hex() == w['stereo_i16']
    # every float32 bit pattern class at scale: 4M random bit patterns (NaNs, infinities, denormals included)
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 1 << 32, size=1 << 22, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    dx = torch.from_numpy(x.copy()).cuda()
    o = torch.zeros(len(x), dtype=torch.int16, device='cuda')
    ctx.pcm_to_int16_device([dx.data_ptr()], len(x), o.data_ptr())
    ctx.synchronize()
    assert np.array_equal(o.cpu().numpy(), O.pcm_to_int16([x]))


@pytest.mark.parametrize('bits,ch', [(16, 1), (16, 2), (24, 1), (24, 2), (32, 1), (32, 2)])
def test_wav_ingest_matches_oracle(ctx, bits, ch):
    import torch
    rng = np.random.default_rng(bits * 3 + ch)
    n = 100003                                           # odd length: no alignment assumptions
    raw = rng.integers(0, 256, size=n * ch * (bits // 8), dtype=np.uint8)
    raw[:bits // 8] = [0] * (bits // 8 - 1) + [0x80]     # most negative sample -> exactly -1
    want = O.pcm_from_int(raw, bits, ch)
    d = torch.from_numpy(raw).cuda()
    outs = [torch.zeros(n, dtype=torch.float32, device='cuda') for _ in range(ch)]
    ctx.pcm_from_int_device(d.data_ptr(), bits, ch, n, [o.data_ptr() for o in outs])
    ctx.synchronize()
    for c in range(ch):
        assert np.array_equal(outs[c].cpu().numpy().view(np.uint32), want[c].view(np.uint32))
    assert want[0][0] == -1.0
    # unaligned input / output pointers take the scalar kernel: same answer
    bps = bits // 8
    d1 = torch.zeros(len(raw) + 1, dtype=torch.uint8, device='cuda')
    d1[1:] = d
    outs1 = [torch.zeros(n + 1, dtype=torch.float32, device='cuda') for _ in range(ch)]
    ctx.pcm_from_int_device(d1.data_ptr() + 1, bits, ch, n, [o.data_ptr() + 4 for o in outs1])
    ctx.synchronize()
    for c in range(ch):
        assert np.array_equal(outs1[c][1:].cpu().numpy().view(np.uint32), want[c].view(np.uint32))
        assert outs1[c][0].item() == 0.0


def test_wav16_file_to_aea_to_wav16_device_chain(ctx):
    """int16 WAV body -> planar f32 -> units -> f32 -> int16, all on the device, against the oracle chain."""
    import torch
    import carta1_amd as c1
    n = 64 * 512
    pcm16 = (np.stack([O.gen_white(21, n), O.gen_white(22, n)], axis=1) * 20000).astype('<i2')
    raw = torch.from_numpy(pcm16.reshape(-1).view(np.uint8).copy()).cuda()
    chans = [torch.zeros(n, dtype=torch.float32, device='cuda') for _ in range(2)]
    ctx.pcm_from_int_device(raw.data_ptr(), 16, 2, n, [c.data_ptr() for c in chans])
    units = torch.zeros(64 * 2 * 212, dtype=torch.uint8, device='cuda')
    ctx.encode_device([c.data_ptr() for c in chans], 64, units.data_ptr(), c1.EncoderOptions())
    back = [torch.zeros(n, dtype=torch.float32, device='cuda') for _ in range(2)]
    ctx.decode_device(units.data_ptr(), 2, 64, [b.data_ptr() for b in back])
    out = torch.zeros(2 * n, dtype=torch.int16, device='cuda')
    ctx.pcm_to_int16_device([b.data_ptr() for b in back], n, out.data_ptr())
    ctx.synchronize()
    f = O.pcm_from_int(pcm16.reshape(-1).view(np.uint8), 16, 2)
    u, _ = O.encode_stream(f)
    p, _ = O.decode_stream(u, 2)
    assert np.array_equal(units.cpu().numpy().reshape(-1, 212), u)
    assert np.array_equal(out.cpu().numpy(), O.pcm_to_int16(p))


def test_signed_zero_and_non_finite_pcm_take_the_reference_arithmetic(ctx):
    """The kernels use binary32 shortcuts for unit twiddles only when that is exact; -0, infinities and NaN in the
    PCM must fall back to the reference's binary64 expressions.  Units against the oracle, fixed and detected modes."""
    import carta1_amd as c1
    n = 48 * 512
    rng = np.random.default_rng(77)
    pcm = O.gen_white(31, n).copy()
    pcm[rng.integers(0, n, 4000)] = -0.0
    pcm[5 * 512:7 * 512] = 0.0                       # two silent frames: zeros of both signs inside the transforms
    pcm[6 * 512 + 17] = -0.0
    pcm[20 * 512 + 3] = np.inf
    pcm[21 * 512 + 100] = -np.inf
    pcm[30 * 512 + 7] = np.nan
    pcm[40 * 512:41 * 512] *= 1e-38                  # denormal range
    for opts in ({'fixedBlockModes': [0, 0, 0]}, {}, {'fixedBlockModes': [2, 2, 3]}):
        want, _ = O.encode_stream([pcm], fixed_modes=opts.get('fixedBlockModes'))
        got = ctx.encode([pcm], c1.EncoderOptions(opts))
        bad = first_diff(got, want)
        assert len(bad) == 0, (opts, bad)


@pytest.mark.parametrize('modes', [[0, 0, 3], [2, 2, 0], [2, 0, 0], [0, 2, 3]])
def test_every_long_short_combination(ctx, modes):
    """Each band is long or short on its own: the mixed radix-4 cores of the encoder and the decoder."""
    import carta1_amd as c1
    chs = [O.gen_white(41, 300 * 512), O.gen_pinkT(42, 300 * 512)]
    want, _ = O.encode_stream(chs, fixed_modes=modes)
    got = ctx.encode(chs, c1.EncoderOptions({'fixedBlockModes': modes}))
    assert first_diff(got, want).size == 0, first_diff(got, want)
    pcm_want, _ = O.decode_stream(want, 2)
    pcm_got = ctx.decode(want, 2)
    for c in range(2):
        assert np.array_equal(pcm_got[c].view(np.uint32), pcm_want[c].view(np.uint32))


def test_internal_chunking_is_invisible():
    """The library cuts a batch into chunks of C1_CHUNK_FRAMES; a context with tiny chunks must produce the same
    bytes (the detection pipeline carries frame -1 of every chunk in its workspace)."""
    import carta1_amd as c1
    chs = [O.gen_pinkT(7, 500 * 512), O.gen_pinkT(8, 500 * 512)]
    old = os.environ.get('C1_CHUNK_FRAMES')
    os.environ['C1_CHUNK_FRAMES'] = '96'
    try:
        small = c1.Context(0)
    finally:
        if old is None:
            del os.environ['C1_CHUNK_FRAMES']
        else:
            os.environ['C1_CHUNK_FRAMES'] = old
    big = c1.Context(0)
    try:
        for opts in ({}, {'fixedBlockModes': [0, 0, 0]}, {'fixedBlockModes': [2, 2, 3]}, {'transientThresholdLow': 0.3}):
            a = small.encode(chs, c1.EncoderOptions(opts))
            b = big.encode(chs, c1.EncoderOptions(opts))
            assert np.array_equal(a, b), opts
        want, _ = O.encode_stream(chs)
        assert np.array_equal(small.encode(chs, c1.EncoderOptions()), want)
    finally:
        small.close()
        big.close()


def _patchwork(frames, seed):
    """Signal made of short segments of very different character and level: exercises block switching, every
    scale-factor range, silence / denormal levels and clipping inside one stream."""
    rng = np.random.RandomState(seed)
    out = np.zeros(frames * 512, dtype=np.float32)
    pos = 0
    t = np.arange(frames * 512, dtype=np.float64)
    while pos < len(out):
        n = int(rng.randint(64, 6000))
        kind = rng.randint(0, 6)
        gain = 10.0 ** rng.uniform(-6, 0.6)
        seg = slice(pos, min(pos + n, len(out)))
        m = seg.stop - seg.start
        if kind == 0:
            x = rng.uniform(-1, 1, m)
        elif kind == 1:
            x = np.sin(2 * np.pi * rng.uniform(20, 20000) * t[seg] / 44100 + rng.uniform(0, 6.28))
        elif kind == 2:
            x = np.cumsum(rng.uniform(-1, 1, m)) * 0.05
        elif kind == 3:
            x = np.zeros(m)
            x[rng.randint(0, m, max(1, m // 200))] = rng.uniform(-1, 1, max(1, m // 200)) * 8
        elif kind == 4:
            x = np.zeros(m)
        else:
            x = np.sign(np.sin(2 * np.pi * rng.uniform(50, 4000) * t[seg] / 44100))
        out[seg] = (gain * x).astype(np.float32)
        pos += n
    return out


@pytest.mark.parametrize('opts', [{}, {'transientThresholdLow': 0.3}, {'fixedBlockModes': [2, 2, 3], 'allocationBias': 2.0},
                                  {'fixedBlockModes': [0, 0, 0], 'allocationBias': 0.5}])
def test_patchwork_stream_against_oracle(ctx, opts):
    import carta1_amd as c1
    frames = 6000
    chs = [_patchwork(frames, 101), _patchwork(frames, 202)]
    bias = opts.get('allocationBias', 1.0)
    want, _ = O.encode_stream(chs, fixed_modes=opts.get('fixedBlockModes'), bias=bias,
                              threshold=opts.get('transientThresholdLow', 1.0))
    got = ctx.encode(chs, c1.EncoderOptions(opts, biased_table=O.biased_table(bias)))
    assert first_diff(got, want).size == 0, 'units differ at %s' % first_diff(got, want)
    modes = want[:, 0] >> 2                      # block-size-mode bits of the header byte
    if 'fixedBlockModes' not in opts:
        assert len(np.unique(modes)) > 2         # the stream really switches block modes
    pcm_want, _ = O.decode_stream(want, 2)
    pcm_got = ctx.decode(want, 2)
    for c in range(2):
        assert np.array_equal(pcm_got[c].view(np.uint32), pcm_want[c].view(np.uint32))


def test_streamed_pinned_host_path_equals_plain_batch(ctx):
    """c1_encode_batch / c1_decode_batch with every host buffer in page-locked memory stream the batch in overlapping
    chunks of 32768 frames; the bytes must not depend on that (chunk seams carry the PCM halo / the previous unit)."""
    import carta1_amd as c1
    frames = 32768 * 2 + 777
    chs = [O.gen_pinkT(51, (frames + 2) * 512), O.gen_pinkT(52, (frames + 2) * 512)]
    for opts, halo in (({}, 2), ({'fixedBlockModes': [0, 0, 0]}, 1), ({'fixedBlockModes': [2, 2, 3]}, 0)):
        src = [c[(2 - halo) * 512:] for c in chs]
        want = ctx.encode(src, c1.EncoderOptions(opts), halo_frames=halo)
        pin = [c1.pinned_empty(len(s), np.float32) for s in src]
        for a, b in zip(pin, src):
            a[:] = b
        out = c1.pinned_empty((frames * 2, 212), np.uint8)
        got = ctx.encode(pin, c1.EncoderOptions(opts), halo_frames=halo, out=out)
        assert np.array_equal(got, want), opts
    # decode with one unit of history in front
    units = want
    ref = ctx.decode(units, 2, halo_units=1)
    pu = c1.pinned_empty(units.shape, np.uint8)
    pu[:] = units
    po = [c1.pinned_empty((frames - 1) * 512, np.float32) for _ in range(2)]
    got = ctx.decode(pu, 2, halo_units=1, out=po)
    for c in range(2):
        assert np.array_equal(got[c].view(np.uint32), ref[c].view(np.uint32))


@pytest.mark.parametrize('bits,ch', [(16, 2), (24, 1), (32, 2)])
def test_wav_body_to_units_and_back_in_one_call(ctx, bits, ch):
    """c1_encode_wav_batch / c1_decode_wav16_batch == the oracle chain WavReader conversion -> zero padded frames ->
    encode, and decode -> 16-bit samples; more than one streaming chunk, ragged length, plain and page-locked input."""
    import carta1_amd as c1
    samples = 32768 * 512 + 5000 + 333                       # two chunks and a partial last frame
    rng = np.random.default_rng(bits)
    bps = bits // 8
    raw = rng.integers(0, 256, size=samples * ch * bps, dtype=np.uint8)
    if bits > 16:                                            # keep the level sane: zero the low bytes' influence
        raw.reshape(-1, bps)[:, -1] = rng.integers(-40, 40, size=samples * ch).astype(np.int8).view(np.uint8)
    chans = O.pcm_from_int(raw, bits, ch)
    want, _ = O.encode_stream([O.pad_frames(c) for c in chans])
    got = ctx.encode_wav(raw, bits, ch)
    assert np.array_equal(got, want), first_diff(got, want)
    praw = c1.pinned_empty(raw.shape, np.uint8)
    praw[:] = raw
    assert np.array_equal(ctx.encode_wav(praw, bits, ch), want)
    pcm_want, _ = O.decode_stream(want, ch)
    back = ctx.decode_wav16(want, ch)
    assert back.shape == (len(pcm_want[0]), ch)
    assert np.array_equal(back.reshape(-1), O.pcm_to_int16(pcm_want))


def test_two_contexts_in_two_threads(ctx):
    """Distinct contexts are independent (own stream, own workspace): two host threads encoding and decoding at the
    same time get the bytes a single thread gets."""
    import threading
    import carta1_amd as c1
    data = [[O.gen_pinkT(61 + 2 * k, 700 * 512), O.gen_white(62 + 2 * k, 700 * 512)] for k in range(2)]
    opts = [c1.EncoderOptions(), c1.EncoderOptions({'fixedBlockModes': [2, 2, 3]})]
    want = [ctx.encode(data[k], opts[k]) for k in range(2)]
    want_pcm = [ctx.decode(want[k], 2) for k in range(2)]
    got, got_pcm, errors = [None, None], [None, None], []

    def work(k):
        try:
            c = c1.Context(0)
            for _ in range(5):
                got[k] = c.encode(data[k], opts[k])
                got_pcm[k] = c.decode(got[k], 2)
            c.close()
        except Exception as e:      # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(2):
        assert np.array_equal(got[k], want[k])
        for c in range(2):
            assert np.array_equal(got_pcm[k][c].view(np.uint32), want_pcm[k][c].view(np.uint32))


def test_sharded_encode_and_decode_equal_one_device(ctx):
    """encode_sharded / decode_sharded: contiguous frame ranges with their halo, one host thread and one context per
    shard (here all on device 0), outputs concatenated -- the bytes of a single-device call."""
    import carta1_amd as c1
    frames = 999
    chs = [O.gen_pinkT(71, frames * 512), O.gen_pinkT(72, frames * 512)]
    for opts in ({}, {'fixedBlockModes': [0, 0, 0]}, {'fixedBlockModes': [2, 2, 3]}):
        want = ctx.encode(chs, c1.EncoderOptions(opts))
        got = c1.encode_sharded(chs, c1.EncoderOptions(opts), devices=(0, 0, 0))
        assert np.array_equal(got, want), opts
    pcm_want = ctx.decode(want, 2)
    pcm_got = c1.decode_sharded(want, 2, devices=(0, 0, 0, 0))
    for c in range(2):
        assert np.array_equal(pcm_got[c].view(np.uint32), pcm_want[c].view(np.uint32))


def test_multi_device_entry_points_equal_one_device(ctx):
    """c1_encode_batch_multi / c1_decode_batch_multi with the same device listed three times (the box has one GPU):
    contiguous shards from their halo, bit-identical to the single-context calls, for detection and fixed modes"""
    import carta1_amd as c1
    n = 1000 * 512
    chans = [O.gen_pinkT(3, n), O.gen_white(2, n)]
    for opts in (c1.EncoderOptions(), c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]}), c1.EncoderOptions({'fixedBlockModes': [2, 0, 3]})):
        one = ctx.encode(chans, opts)
        multi = c1.encode_multi(chans, opts, devices=(0, 0, 0))
        assert np.array_equal(one, multi)
        a = ctx.decode(one, 2)
        b = c1.decode_multi(one, 2, devices=(0, 0, 0))
        for c in range(2):
            assert np.array_equal(a[c].view(np.uint32), b[c].view(np.uint32))
    mono = c1.encode_multi([chans[0][:5 * 512]], c1.EncoderOptions(), devices=(0, 0, 0, 0, 0, 0, 0, 0))   # more shards than frames
    assert np.array_equal(mono, ctx.encode([chans[0][:5 * 512]], c1.EncoderOptions()))


def test_multi_entry_points_follow_the_installed_tables_and_survive_concurrent_calls(ctx):
    """the pooled contexts of the *_multi entry points are leased per call and retired when the tables change: after
    c1_set_tables(custom) encode_multi equals a context created afresh (not the context the pool made under the default
    tables), back again after c1_set_tables(NULL); and two threads sharing devices in opposite order get their own
    results (each call holds its contexts until it returns)"""
    import ctypes as C
    import threading
    import carta1_amd as c1
    from carta1_amd import capi
    n = 600 * 512
    chans = [O.gen_pinkT(13, n), O.gen_white(14, n)]
    opts = c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]})
    base = c1.encode_multi(chans, opts, devices=(0, 0))
    assert np.array_equal(base, ctx.encode(chans, opts))
    t = capi.default_tables()
    t.window_short[7] *= 1.01                       # any table that changes the output will do
    try:
        capi.check(capi.load().c1_set_tables(C.byref(t)))
        fresh = c1.Context(0)
        want = fresh.encode(chans, opts).copy()
        fresh.close()
        assert not np.array_equal(want, base)
        assert np.array_equal(c1.encode_multi(chans, opts, devices=(0, 0)), want)
    finally:
        capi.check(capi.load().c1_set_tables(None))
    assert np.array_equal(c1.encode_multi(chans, opts, devices=(0, 0)), base)
    # concurrent calls
    other = [O.gen_white(15, n), O.gen_pinkT(16, n)]
    want_b = ctx.encode(other, c1.EncoderOptions()).copy()
    out = {}

    def run(name, ch, o, dev):
        for _ in range(3):
            out[name] = c1.encode_multi(ch, o, devices=dev)
    th = [threading.Thread(target=run, args=('a', chans, opts, (0, 0, 0))), threading.Thread(target=run, args=('b', other, c1.EncoderOptions(), (0, 0)))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert np.array_equal(out['a'], base) and np.array_equal(out['b'], want_b)
