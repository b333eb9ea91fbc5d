"""CPU-only: the C-ABI library loads and exports every symbol include/carta1_hip.h declares, fails
loudly without a GPU (no CPU fallback), the generated tables are in sync with the reference fixtures,
and the host-side format code (AEA header, sound-unit fields, options) behaves like the reference's."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='module')
def lib():
    from carta1_amd import build, capi
    build.build_library()
    return capi.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from carta1_amd import capi
    header = open(os.path.join(ROOT, 'include', 'carta1_hip.h')).read()
    declared = set(re.findall(r'^\s*(?:int|const char \*)\s*\*?\s*(c1_[a-z0-9_]+)\s*\(', header, re.M))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), 'library does not export ' + name
    assert declared == set(capi.SIGNATURES), declared ^ set(capi.SIGNATURES)
    assert lib.c1_abi_version() == 3


def test_struct_layouts_match_header():
    from carta1_amd import capi
    assert C.sizeof(capi.Tables) == (64 + 32 + 32 + 128 + 256 + 32 + 128 + 256 + 16 + 1) * 8
    assert C.sizeof(capi.EncodeOptions) == 64 * 8 + 8 + 16


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason='checks the no-device behaviour')
def test_product_path_fails_loudly_without_a_device(lib):
    import carta1_amd as c1
    with pytest.raises(c1.Carta1Error) as e:
        c1.Context(0)
    assert e.value.code == 2      # C1_ERR_NO_DEVICE
    with pytest.raises(c1.Carta1Error):
        c1.encode_pcm([np.zeros(512, np.float32)])


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, 'carta1_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp', '.js', '.cc', '.inc')) or f == 'Makefile':
                text = open(os.path.join(d, f), errors='replace').read()
                assert 'oracle' not in text.lower(), os.path.join(d, f)


def test_generated_tables_are_in_sync_with_reference_fixture():
    assert subprocess.call([sys.executable, os.path.join(ROOT, 'tools', 'gen_tables.py'), '--check']) == 0


def test_default_tables_equal_reference_values(lib):
    from carta1_amd import capi
    import oracle_lib as O
    t = capi.default_tables()
    g = O.golden_tables()
    assert [t.scale_factors[i] for i in range(64)] == [O.h2d(h) for h in g['scale_factors_f64']]
    assert [t.window_short[i] for i in range(32)] == [O.h2d(h) for h in g['window_short_f64']]
    for name in ('fwd64', 'fwd256', 'fwd512', 'inv64', 'inv256', 'inv512'):
        arr = getattr(t, 'mdct_' + name)
        assert list(arr) == [O.h2d(h) for h in g['mdct_sincos_f64'][name]]
    for k, s in enumerate((2, 4, 8, 16, 32, 64, 128, 256)):
        assert [t.fft_w[k][0], t.fft_w[k][1]] == [O.h2d(h) for h in g['fft_w_f64'][str(s)]]
    assert t.log1p_10 == O.h2d(g['log1p_10_f64'])
    # derived constants the kernels hard-code: powers of two are exact in the reference too
    assert [O.h2d(h) for h in g['inv_power_of_two_f64']] == [2.0 ** -b for b in range(17)]
    ddf = [O.h2d(h) for h in g['distortion_delta_factors_f64']]
    assert ddf[0] == 1.75 and ddf[1:] == [2.0 ** -(w + 2) for w in range(1, 15)]
    assert g['word_length_delta_bits'] == [2] + [1] * 14
    assert g['word_length_bits'] == [0] + list(range(2, 17))


def test_default_options_are_the_reference_defaults(lib):
    import carta1_amd as c1
    o = c1.EncoderOptions().to_c()
    assert o.transient_threshold == 1.0 and list(o.fixed_block_modes) == [-1, -1, -1]
    assert list(o.biased_scale_factors) == [2.0 ** (i / 3.0 - 21) if i % 3 == 0 else o.biased_scale_factors[i] for i in range(64)]
    with pytest.raises(ValueError, match='Value for allocationBias must be between 0.0 and 5.0, got 7'):
        c1.EncoderOptions({'allocationBias': 7})
    with pytest.raises(ValueError):
        c1.EncoderOptions({'transientThresholdLow': 0})
    o = c1.EncoderOptions({'fixedBlockModes': [2, 0, 3], 'allocationBias': 2.0, 'unknownKey': 1}).to_c()
    assert list(o.fixed_block_modes) == [2, 0, 3]
    import oracle_lib as O
    assert list(o.biased_scale_factors) == list(O.biased_table(2.0))     # libm pow == V8 pow for bias 2
    o = c1.EncoderOptions({'allocationBias': 0.5}).to_c()
    assert list(o.biased_scale_factors) == list(O.biased_table(0.5))


def test_aea_header_matches_reference_bytes():
    import carta1_amd as c1
    e = json.load(open(os.path.join(G, 'aea_edge_cases.json')))
    h = c1.aea_header('encoded by carta1', 4, 2)
    assert len(h) == e['header_len'] == 2048
    assert h[:272].hex() == e['header_hex_first_272']
    assert c1.parse_aea_header(h) == {'title': 'encoded by carta1', 'frameCount': 4, 'channelCount': 2}
    # the C entry point (host only, no device needed) writes the same bytes
    from carta1_amd import capi
    buf = (C.c_uint8 * 2048)()
    assert capi.load().c1_aea_header(b'encoded by carta1', 4, 2, buf) == 0
    assert bytes(buf) == h
    with pytest.raises(ValueError, match='Invalid AEA file'):
        c1.parse_aea_header(bytes(2048))
    with pytest.raises(ValueError, match='Header must be 2048 bytes'):
        c1.parse_aea_header(bytes(10))


def test_sound_unit_fields_round_trip_and_match_reference():
    from carta1_amd import codec
    import oracle_lib as O
    k = json.load(open(os.path.join(G, 'config1_sine1k.json')))
    f = codec.deserialize_frame(bytes.fromhex(k['unit_hex']))
    assert f['nBfu'] == k['nBfu'] and f['blockModes'] == k['blockModes']
    assert f['wordLengthIndices'] == k['wordLengthIndices'] and f['scaleFactorIndices'] == k['scaleFactorIndices']
    assert f['quantizedCoefficients'] == k['quantizedCoefficients']
    assert codec.serialize_frame(f).hex() == k['unit_hex']
    units = np.fromfile(os.path.join(G, 'kat64_pinkT_detect.units.bin'), dtype=np.uint8).reshape(-1, 212)
    for u in units[:40]:
        f = codec.deserialize_frame(u)
        assert codec.serialize_frame(f) == u.tobytes()
        o = O.unpack_unit(u)
        assert o.nbfu == f['nBfu'] and list(o.modes) == f['blockModes']
        assert list(o.wl)[:o.nbfu] == f['wordLengthIndices'] and list(o.sfi)[:o.nbfu] == f['scaleFactorIndices']
        assert list(o.q)[:sum(codec.SPECS_PER_BFU[:o.nbfu])] == [v for b in f['quantizedCoefficients'] for v in b]
    with pytest.raises(ValueError, match='Frame must be 212 bytes'):
        codec.deserialize_frame(b'123')


def test_table_dependent_shortcuts_are_verified_for_the_default_tables():
    """The kernels read findScaleFactor off the binary32 bit pattern and dequantize with a reciprocal + two FMAs
    only when the host has checked those forms over their whole input domain for the installed tables."""
    from carta1_amd import capi
    lib = capi.load()
    a, b = C.c_int(-1), C.c_int(-1)
    assert lib.c1_table_fast_paths(C.byref(a), C.byref(b)) == 0
    assert (a.value, b.value) == (1, 2)
    t = capi.Tables()
    assert lib.c1_get_default_tables(C.byref(t)) == 0
    t.scale_factors[10] *= 1.0000001            # no longer 2^(i/3 - 21): the bit-pattern form must be refused
    try:
        assert lib.c1_set_tables(C.byref(t)) == 0
        assert lib.c1_table_fast_paths(C.byref(a), C.byref(b)) == 0
        assert a.value == 0
    finally:
        lib.c1_set_tables(None)
    assert lib.c1_table_fast_paths(C.byref(a), C.byref(b)) == 0 and (a.value, b.value) == (1, 2)
    os.environ['C1_NO_DQ_STEP'] = '1'
    try:
        assert lib.c1_table_fast_paths(C.byref(a), C.byref(b)) == 0 and (a.value, b.value) == (1, 1)
    finally:
        del os.environ['C1_NO_DQ_STEP']


def test_page_locked_allocation_needs_a_device():
    """c1_host_alloc is HIP host memory: without a device it fails loudly like every other entry point."""
    from carta1_amd import capi
    lib = capi.load()
    n = C.c_int(0)
    if lib.c1_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip('a HIP device is present')
    p = C.c_void_p()
    assert lib.c1_host_alloc(1 << 20, C.byref(p)) != 0 and not p.value
    assert b'no HIP device' in lib.c1_last_error()
    assert lib.c1_host_free(None) == 0


def test_shard_plan_tiles_the_batch():
    from carta1_amd import shard_plan
    for frames, shards, hist in ((100, 8, 2), (7, 8, 1), (1048576, 8, 2), (5, 1, 2), (0, 4, 1)):
        plan = shard_plan(frames, shards, hist)
        assert plan[0][0] == 0 and plan[-1][1] == frames
        assert all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
        sizes = [b - a for a, b, _ in plan]
        assert max(sizes) - min(sizes) <= 1
        assert all(h == min(hist, a) for a, _, h in plan)
