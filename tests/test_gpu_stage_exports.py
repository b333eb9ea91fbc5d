"""GPU: the single-stage functions the reference exports next to encode()/decode() (codec/index.js:30-35,42) through the
C ABI -- c1_quantize, c1_dequantize, c1_fft, c1_qmf_analysis_batch, c1_mdct_batch -- against outputs of the reference
itself (tests/golden/quantize.json, tests/golden/stage_exports.json made by tests/golden/gen/gen_stage_exports.mjs) and,
batched, against the oracle.  The JavaScript wrappers with the reference's call shapes are checked by js/selftest.mjs --gpu."""
import json
import os
import struct

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def ctx():
    import carta1_amd as c1
    c = c1.Context(0)
    yield c
    c.close()


def xorshift_white(seed, n, amp):
    s = seed
    out = np.empty(n, dtype=np.float32)
    for i in range(n):
        s ^= (s << 13) & 0xffffffff
        s ^= s >> 17
        s ^= (s << 5) & 0xffffffff
        out[i] = np.float32((s / 4294967296.0 * 2 - 1) * amp)
    return out


def le_hex(a):
    return np.ascontiguousarray(a).tobytes().hex()


def test_quantize_and_dequantize_equal_the_reference(ctx):
    vec = json.load(open(os.path.join(G, 'quantize.json')))
    be = lambda h: struct.unpack('>f', bytes.fromhex(h))[0]
    for v in vec:
        x = np.array([be(h) for h in v['x']], dtype=np.float32)
        q = ctx.quantize(x, v['sfi'], v['bits'])
        assert q.tolist() == v['q'], v
        d = ctx.dequantize(np.array(v['q'], dtype=np.int32), v['sfi'], v['bits'])
        want = np.array([be(h) for h in v['d']], dtype=np.float32)
        assert np.array_equal(d.view(np.uint32), want.view(np.uint32)), v
    sx = json.load(open(os.path.join(G, 'stage_exports.json')))
    for v in sx['quantize']:
        q = ctx.quantize(xorshift_white(v['seed'], v['n'], v['amp']), v['sfi'], v['bits'])
        assert q.tolist() == v['q'] and le_hex(ctx.dequantize(q, v['sfi'], v['bits'])) == v['d'], v
    # a whole unit's worth at once, specials included, against the oracle
    x = np.concatenate([O.gen_white(3, 500) * np.float32(3.0), np.array([np.inf, -np.inf, np.nan, 3e9, -1e30, 0.0, -0.0], dtype=np.float32)])
    import ctypes as C
    for sfi, bits in ((33, 7), (63, 16), (2, 2)):
        want = np.zeros(x.size, dtype=np.int32)
        O.lib().c1o_quantize_bfu(x.ctypes.data_as(C.POINTER(C.c_float)), x.size, sfi, bits, want.ctypes.data_as(C.POINTER(C.c_int)))
        assert np.array_equal(ctx.quantize(x, sfi, bits), want), (sfi, bits)


def test_fft_equals_the_reference(ctx):
    sx = json.load(open(os.path.join(G, 'stage_exports.json')))
    t = O.golden_tables()
    for v in sx['fft']:
        n = v['n']
        re, im = xorshift_white(v['seed_real'], n, v['amp']), xorshift_white(v['seed_imag'], n, v['amp'])
        w = []
        stride = 2
        while stride <= n:
            if str(stride) in t['fft_w_f64']:
                w += [O.h2d(h) for h in t['fft_w_f64'][str(stride)]]            # V8's own (cos, sin) for strides the reference's tables hold
            else:
                w += [np.cos(-2 * np.pi / stride), np.sin(-2 * np.pi / stride)]  # beyond them (n > 256): libm, equal to V8 here or the test fails
            stride *= 2
        ctx.fft(re, im, np.array(w))
        assert le_hex(re) == v['real'] and le_hex(im) == v['imag'], n


def test_qmf_and_mdct_stages_equal_the_reference(ctx):
    sx = json.load(open(os.path.join(G, 'stage_exports.json')))
    for run in sx['stages']:
        pcm = xorshift_white(run['seed'], 4 * 512, 0.5)
        bands = ctx.qmf_analysis(pcm)                                  # the whole stream at once
        for f in range(4):
            assert le_hex(bands[f]) == ''.join(run['frames'][f]['bands_raw']), (run['modes'], f)
            again = ctx.qmf_analysis(pcm[max(0, f - 1) * 512:(f + 1) * 512], halo_frames=min(f, 1))   # frame by frame from one frame of history
            assert np.array_equal(again[0], bands[f])
        modes = np.tile(np.array(run['modes'], dtype=np.int32), (4, 1))
        co, bw = ctx.mdct(bands, modes)
        for f in range(4):
            assert le_hex(co[f]) == run['frames'][f]['coefficients'], (run['modes'], f)
            assert le_hex(bw[f]) == ''.join(run['frames'][f]['bands_after']), (run['modes'], f)
            if f:
                c1f, _ = ctx.mdct(bands[f - 1:f + 1], modes[:1], halo_frames=1)    # one frame from its predecessor's bands
                assert np.array_equal(c1f[0], co[f])


def test_stage_batches_equal_the_oracle(ctx):
    """300 frames, block modes changing from frame to frame"""
    import ctypes as C
    n = 300
    pcm = O.gen_pinkT(7, n * 512)
    rng = np.random.default_rng(3)
    modes = (rng.integers(0, 2, (n, 3)) * np.array([2, 2, 3])).astype(np.int32)
    bands = ctx.qmf_analysis(pcm)
    co, _ = ctx.mdct(bands, modes)
    st = O.EncState()
    fp = C.POINTER(C.c_float)
    for f in range(n):
        b = np.zeros(512, dtype=np.float32)
        O.lib().c1o_qmf_analysis_frame(C.byref(st), pcm[512 * f:].ctypes.data_as(fp), b.ctypes.data_as(fp))
        assert np.array_equal(bands[f].view(np.uint32), b.view(np.uint32)), f
        c = np.zeros(512, dtype=np.float32)
        O.lib().c1o_mdct_frame(C.byref(st), b.ctypes.data_as(fp), (C.c_int * 3)(*modes[f]), c.ctypes.data_as(fp))
        assert np.array_equal(co[f].view(np.uint32), c.view(np.uint32)), f
