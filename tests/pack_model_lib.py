"""ctypes binding of tests/model/pack_model.c: the CPU model of the decisions the product's speculative path takes on
binary32 coefficients (scale-factor guard of k_analysis_spec, quantizer + guard band of k_pack<.., SPEC>).  Test
infrastructure; built on demand with gcc into oracle/_build/."""
import ctypes as C
import os
import subprocess

import numpy as np

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'model', 'pack_model.c')
SO = os.path.join(ROOT, 'oracle', '_build', 'libpack_model.so')

SPECS = np.array([8] * 4 + [4] * 4 + [8] * 4 + [6] * 12 + [7] * 4 + [9] * 4 + [10] * 4 + [12] * 8 + [20] * 8, dtype=np.int32)
FIRST = np.concatenate([[0], np.cumsum(SPECS)]).astype(np.int32)          # BFU-major slot of every BFU's first coefficient
START_LONG = FIRST[:52].copy()                                            # BFU_START_LONG == the slot order (constants.js:38-44)
START_SHORT = np.array([0, 32, 64, 96, 8, 40, 72, 104, 12, 44, 76, 108, 20, 52, 84, 116, 26, 58, 90, 122, 128, 160, 192, 224,
                        134, 166, 198, 230, 141, 173, 205, 237, 150, 182, 214, 246, 256, 288, 320, 352, 384, 416, 448, 480,
                        268, 300, 332, 364, 396, 428, 460, 492], dtype=np.int32)   # BFU_START_SHORT (constants.js:46-52)
BAND_OF_BFU = np.array([0] * 20 + [1] * 16 + [2] * 16, dtype=np.int32)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO) or os.path.getmtime(SRC) > os.path.getmtime(SO):
            os.makedirs(os.path.dirname(SO), exist_ok=True)
            subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-std=c11', '-ffp-contract=off', '-fno-fast-math',
                                   '-o', SO, SRC, '-lm'])
        L = C.CDLL(SO)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.pack_model_sf.argtypes = [fp, fp, C.c_uint32, C.c_uint32, ip, ip]
        L.pack_model_quantize.argtypes = [fp, fp, ip, ip, C.c_int, fp, ip, fp, ip]
        _lib = L
    return _lib


def scale_factors():
    return np.array([O.h2d(x) for x in O.golden_tables()['scale_factors_f64']], dtype=np.float64)


def sf_fraction_patterns():
    """(m1, m2): the 23 fraction bits of the largest binary32 values <= 2^(1/3) and 2^(2/3) times a power of two, as
    c1_api.hip (build_device_tables) derives them from the installed SCALE_FACTORS."""
    sf = scale_factors()
    out = []
    for i in (61, 62):
        f = np.float32(sf[i])
        if float(f) > sf[i]:
            f = np.nextafter(f, np.float32(0))
        out.append(int(f.view(np.uint32)) & 0x7fffff)
    return tuple(out)


def norm32_table():
    """fl32(quantRange(wl) / SCALE_FACTORS[sfi]), index sfi * 16 + wl (quantization.js:42-44)"""
    sf = scale_factors()
    t = np.zeros(64 * 16, dtype=np.float64)
    for s in range(64):
        for w in range(1, 16):
            t[s * 16 + w] = ((1 << w) - 1) / sf[s]
    return t.astype(np.float32)


def to_slots(coefs, modes=(0, 0, 0)):
    """coefficient order of quantizationStage (low128 | mid128 | high256) -> BFU-major slot order for the block modes"""
    coefs = np.asarray(coefs, dtype=np.float32)
    out = np.empty(512, dtype=np.float32)
    for b in range(52):
        st = (START_LONG if modes[BAND_OF_BFU[b]] == 0 else START_SHORT)[b]
        out[FIRST[b]:FIRST[b + 1]] = coefs[st:st + SPECS[b]]
    return out


def sf_guard(slots, eps, per_bfu=False):
    """-> (sfi[52] as the kernel stores them, unstable flag of the unit[, which BFUs are open])"""
    m1, m2 = sf_fraction_patterns()
    slots = np.ascontiguousarray(slots, dtype=np.float32)
    eps = np.ascontiguousarray(eps, dtype=np.float32)
    sfi = np.zeros(52, dtype=np.int32)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    opened = np.zeros(52, dtype=np.int32)
    un = lib().pack_model_sf(slots.ctypes.data_as(fp), eps.ctypes.data_as(fp), m1, m2, sfi.ctypes.data_as(ip), opened.ctypes.data_as(ip))
    if per_bfu:
        return sfi, bool(un), opened.astype(bool)
    return sfi, bool(un)


_norm32 = None


def quantize(slots, eps, sfi, wl, nbfu, per_slot=False):
    """-> (mantissas[512] in slot order, doubtful flag, worst |fract - 1/2| + et[, which mantissas are doubtful])"""
    global _norm32
    if _norm32 is None:
        _norm32 = norm32_table()
    slots = np.ascontiguousarray(slots, dtype=np.float32)
    eps = np.ascontiguousarray(eps, dtype=np.float32)
    sfi = np.ascontiguousarray(sfi, dtype=np.int32)
    wl = np.ascontiguousarray(wl, dtype=np.int32)
    q = np.zeros(512, dtype=np.int32)
    worst = np.zeros(1, dtype=np.float32)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    doubt = np.zeros(512, dtype=np.int32)
    d = lib().pack_model_quantize(slots.ctypes.data_as(fp), eps.ctypes.data_as(fp), sfi.ctypes.data_as(ip), wl.ctypes.data_as(ip),
                                  int(nbfu), _norm32.ctypes.data_as(fp), q.ctypes.data_as(ip), worst.ctypes.data_as(fp), doubt.ctypes.data_as(ip))
    if per_slot:
        return q, bool(d), float(worst[0]), doubt.astype(bool)
    return q, bool(d), float(worst[0])


def allocate(coefs, modes=(0, 0, 0), bias=1.0):
    """the reference's allocateBits through the oracle: -> (nbfu, wl[52], sfi[52])"""
    coefs = np.ascontiguousarray(coefs, dtype=np.float32)
    m = (C.c_int * 3)(*modes)
    b = O.biased_table(bias)
    n = C.c_int(0)
    wl = np.zeros(52, dtype=np.int32)
    sfi = np.zeros(52, dtype=np.int32)
    ip = C.POINTER(C.c_int)
    O.lib().c1o_allocate(coefs.ctypes.data_as(C.POINTER(C.c_float)), m, b.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n),
                         wl.ctypes.data_as(ip), sfi.ctypes.data_as(ip))
    return n.value, wl, sfi


def reference_quantize(coefs, modes, nbfu, wl, sfi):
    """the reference's mantissas (quantization.js:34-56 through the oracle) in BFU-major slot order"""
    coefs = np.ascontiguousarray(coefs, dtype=np.float32)
    q = np.zeros(512, dtype=np.int32)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    for b in range(52):
        n = int(SPECS[b])
        w = int(wl[b]) if b < nbfu else 0
        bits = 0 if w == 0 else w + 1
        st = int((START_LONG if modes[BAND_OF_BFU[b]] == 0 else START_SHORT)[b])
        x = np.ascontiguousarray(coefs[st:st + n])
        out = np.zeros(n, dtype=np.int32)
        O.lib().c1o_quantize_bfu(x.ctypes.data_as(fp), n, int(sfi[b]), bits, out.ctypes.data_as(ip))
        q[FIRST[b]:FIRST[b] + n] = out
    return q
