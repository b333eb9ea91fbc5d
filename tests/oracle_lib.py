"""ctypes binding of the CPU oracle (oracle/atrac1_oracle.c).

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Builds oracle/_build/libatrac1_oracle.so with
oracle/Makefile when it is missing or stale.
"""
import ctypes as C
import json
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, 'oracle')
SO = os.path.join(ODIR, '_build', 'libatrac1_oracle.so')


class EncState(C.Structure):
    _fields_ = [('qmf_low', C.c_float * 46), ('qmf_mid', C.c_float * 46), ('qmf_high', C.c_float * 39),
                ('overlap', (C.c_float * 32) * 3), ('prev_mag', C.c_float * 256)]


class DecState(C.Structure):
    _fields_ = [('qmf_low', C.c_float * 46), ('qmf_mid', C.c_float * 46), ('qmf_high', C.c_float * 39),
                ('tail', (C.c_float * 16) * 3)]


class Options(C.Structure):
    _fields_ = [('fixed_modes', C.c_int * 3), ('threshold', C.c_double), ('biased_sf', C.c_double * 64)]


class Fields(C.Structure):
    _fields_ = [('nbfu', C.c_int), ('modes', C.c_int * 3), ('wl', C.c_int * 52), ('sfi', C.c_int * 52),
                ('q', C.c_int * 512)]


def build():
    srcs = [os.path.join(ODIR, f) for f in ('atrac1_oracle.c', 'atrac1_oracle.h', 'c1o_tables.inc', 'Makefile')]
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
        subprocess.check_call(['make', '-s', '-C', ODIR])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int)
        _lib.c1o_default_biased_sf.argtypes = [C.c_double, C.POINTER(C.c_double)]
        _lib.c1o_scale_factors.restype = C.POINTER(C.c_double)
        _lib.c1o_qmf_analysis_frame.argtypes = [C.POINTER(EncState), fp, fp]
        _lib.c1o_block_modes.argtypes = [C.POINTER(EncState), fp, C.POINTER(Options), ip]
        _lib.c1o_transient_mags.argtypes = [fp, fp]
        _lib.c1o_libm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_long]
        _lib.c1o_libm.restype = None
        _lib.c1o_detect_transient.argtypes = [fp, fp, C.c_int, C.c_double]
        _lib.c1o_detect_transient.restype = C.c_int
        _lib.c1o_transient_score.argtypes = [fp, fp, C.c_int]
        _lib.c1o_transient_score.restype = C.c_double
        _lib.c1o_mdct_frame.argtypes = [C.POINTER(EncState), fp, ip, fp]
        _lib.c1o_find_scale_factor.argtypes = [fp, C.c_int]
        _lib.c1o_find_scale_factor.restype = C.c_int
        _lib.c1o_allocate.argtypes = [fp, ip, C.POINTER(C.c_double), ip, ip, ip]
        _lib.c1o_quantize_bfu.argtypes = [fp, C.c_int, C.c_int, C.c_int, ip]
        _lib.c1o_dequantize_bfu.argtypes = [ip, C.c_int, C.c_int, C.c_int, fp]
        _lib.c1o_encode_frame.argtypes = [C.POINTER(EncState), fp, C.POINTER(Options), C.POINTER(Fields)]
        _lib.c1o_pack_unit.argtypes = [C.POINTER(Fields), C.POINTER(C.c_uint8)]
        _lib.c1o_unpack_unit.argtypes = [C.POINTER(C.c_uint8), C.POINTER(Fields)]
        _lib.c1o_decode_frame.argtypes = [C.POINTER(DecState), C.POINTER(Fields), fp]
        _lib.c1o_encode_stream.argtypes = [C.POINTER(fp), C.c_int, C.c_long, C.POINTER(Options),
                                           C.POINTER(EncState), C.POINTER(C.c_uint8)]
        _lib.c1o_decode_stream.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_long, C.POINTER(DecState),
                                           C.POINTER(fp)]
        _lib.c1o_gen_white.argtypes = [C.c_uint32, C.c_long, fp]
        _lib.c1o_gen_pinkT.argtypes = [C.c_uint32, C.c_long, fp]
        _lib.c1o_pcm_from_int.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_long, C.POINTER(fp)]
        _lib.c1o_pcm_to_int16.argtypes = [C.POINTER(fp), C.c_int, C.c_long, C.POINTER(C.c_int16)]
        _lib.c1o_aea_header.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint8)]
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


_tables = None


def golden_tables():
    global _tables
    if _tables is None:
        _tables = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'tables.json')))
    return _tables


def h2d(h):
    return struct.unpack('>d', bytes.fromhex(h))[0]


def h2f(h):
    return struct.unpack('>f', bytes.fromhex(h))[0]


def biased_table(bias):
    """pow(SCALE_FACTORS[i], bias) as V8 computed it when the golden vectors were made; falls
    back to libm pow for a bias the fixtures do not hold (parity with V8 then unpinned)."""
    t = golden_tables()['biased_scale_factors_f64']
    for k, v in t.items():
        if float(k) == float(bias):
            return np.array([h2d(x) for x in v], dtype=np.float64)
    out = np.zeros(64, dtype=np.float64)
    lib().c1o_default_biased_sf(float(bias), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def make_options(fixed_modes=None, bias=1.0, threshold=1.0):
    o = Options()
    fm = fixed_modes if fixed_modes is not None else (-1, -1, -1)
    for i in range(3):
        o.fixed_modes[i] = int(fm[i])
    o.threshold = float(threshold)
    b = biased_table(bias)
    for i in range(64):
        o.biased_sf[i] = b[i]
    return o


def pcm_from_int(raw, bits, channels):
    """raw: uint8 array of interleaved little-endian integer PCM.  Returns planar float32 arrays."""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    n = len(raw) // (channels * (bits // 8))
    outs = [np.empty(n, dtype=np.float32) for _ in range(channels)]
    ptrs = (C.POINTER(C.c_float) * channels)(*[_fp(o) for o in outs])
    lib().c1o_pcm_from_int(_u8(raw), bits, channels, n, ptrs)
    return outs


def pcm_to_int16(channels):
    chans = [np.ascontiguousarray(c, dtype=np.float32) for c in channels]
    n = len(chans[0])
    out = np.empty(n * len(chans), dtype=np.int16)
    ptrs = (C.POINTER(C.c_float) * len(chans))(*[_fp(c) for c in chans])
    lib().c1o_pcm_to_int16(ptrs, len(chans), n, out.ctypes.data_as(C.POINTER(C.c_int16)))
    return out


def aea_header(title, frame_count, channels):
    out = np.zeros(2048, dtype=np.uint8)
    lib().c1o_aea_header(title.encode('utf-8'), frame_count, channels, _u8(out))
    return out.tobytes()


def gen_white(seed, n):
    out = np.empty(n, dtype=np.float32)
    lib().c1o_gen_white(seed, n, _fp(out))
    return out


def gen_pinkT(seed, n):
    out = np.empty(n, dtype=np.float32)
    lib().c1o_gen_pinkT(seed, n, _fp(out))
    return out


def encode_stream(channels, fixed_modes=None, bias=1.0, threshold=1.0, states=None):
    """channels: list of float32 arrays of equal length (a multiple of 512).  Returns units
    uint8 [frames*nch, 212] interleaved L,R and the final states."""
    nch = len(channels)
    chans = [np.ascontiguousarray(c, dtype=np.float32) for c in channels]
    frames = len(chans[0]) // 512
    assert all(len(c) == frames * 512 for c in chans)
    o = make_options(fixed_modes, bias, threshold)
    st = states if states is not None else (EncState * nch)()
    units = np.zeros((frames * nch, 212), dtype=np.uint8)
    ptrs = (C.POINTER(C.c_float) * nch)(*[_fp(c) for c in chans])
    lib().c1o_encode_stream(ptrs, nch, frames, C.byref(o), st, _u8(units))
    return units, st


def decode_stream(units, nch, states=None):
    units = np.ascontiguousarray(units, dtype=np.uint8).reshape(-1, 212)
    frames = units.shape[0] // nch
    st = states if states is not None else (DecState * nch)()
    outs = [np.zeros(frames * 512, dtype=np.float32) for _ in range(nch)]
    ptrs = (C.POINTER(C.c_float) * nch)(*[_fp(c) for c in outs])
    lib().c1o_decode_stream(_u8(units), nch, frames, st, ptrs)
    return outs, st


def unpack_unit(unit):
    f = Fields()
    u = np.ascontiguousarray(unit, dtype=np.uint8)
    lib().c1o_unpack_unit(_u8(u), C.byref(f))
    return f


def pad_frames(x):
    n = (len(x) + 511) // 512 * 512
    out = np.zeros(n, dtype=np.float32)
    out[:len(x)] = x
    return out


LIBM_FUNCTIONS = ('log', 'exp', 'log1p', 'log10')


def libm(name, x):
    """Math.log / exp / log1p / log10 as the reference's engine evaluates them (oracle/c1o_fdlibm.h)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().c1o_libm(LIBM_FUNCTIONS.index(name), x.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)), x.size)
    return out
