"""ctypes binding of tests/model/spec_model.c: the CPU model of the product's speculative binary32 analysis kernel
(carta1_amd/csrc/c1_k_spec.hip).  Test infrastructure; built on demand with gcc into oracle/_build/."""
import ctypes as C
import os
import subprocess

import numpy as np

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'model', 'spec_model.c')
SO = os.path.join(ROOT, 'oracle', '_build', 'libspec_model.so')

# the constants of the bound as carta1_amd/csrc/c1_api.hip (build_spec_tables) has them
GH, GQ = 1.4160, 4.80
KA_POST, KA_PRE, KA_ROUND_A, KA_ROUND4, KA_ROUND2, THETA = 4.83, 6.25, 4.0, 7.0, 5.0, 1.01

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO) or os.path.getmtime(SRC) > os.path.getmtime(SO):
            os.makedirs(os.path.dirname(SO), exist_ok=True)
            subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-std=c11', '-ffp-contract=off', '-fno-fast-math',
                                   '-o', SO, SRC, '-lm'])
        L = C.CDLL(SO)
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.spec_model_init.argtypes = [fp, dp, dp, dp, dp, dp, fp]
        L.spec_model_stream.argtypes = [fp, C.c_long, fp, fp, fp, C.c_int]
        t = O.golden_tables()
        d = lambda hs: np.array([O.h2d(x) for x in hs], dtype=np.float64)
        even = np.array([O.h2f(x) for x in t['qmf_even_f32']], dtype=np.float32)
        win = d(t['window_short_f64'])
        f64, f256, f512 = (d(t['mdct_sincos_f64'][k]) for k in ('fwd64', 'fwd256', 'fwd512'))
        fw = np.array([[O.h2d(a) for a in t['fft_w_f64'][str(1 << (s + 1))]] for s in range(8)], dtype=np.float64).ravel()
        coef = bound_coefficients(f256, f512, f64)
        L.spec_model_init(even.ctypes.data_as(fp), win.ctypes.data_as(dp), f64.ctypes.data_as(dp), f256.ctypes.data_as(dp),
                          f512.ctypes.data_as(dp), fw.ctypes.data_as(dp), coef.ctypes.data_as(fp))
        _lib = L
    return _lib


def _round_up_f32(x):
    f = np.float32(x)
    if float(f) < x:
        f = np.nextafter(f, np.float32(np.inf))
    return f


def bound_coefficients(f256, f512, f64):
    """cz[3], cw[3], cl[3], eabs, then the triples for short blocks, exactly as build_spec_tables (c1_api.hip) computes them."""
    u = 2.0 ** -24 * THETA
    s256 = max(f256[2 * i] ** 2 + f256[2 * i + 1] ** 2 for i in range(64))
    s512 = max(f512[2 * i] ** 2 + f512[2 * i + 1] ** 2 for i in range(128))
    ka64 = KA_POST + KA_ROUND_A + 2 * KA_ROUND4 + KA_PRE
    ka128 = ka64 + KA_ROUND2
    s64 = max(f64[2 * i] ** 2 + f64[2 * i + 1] ** 2 for i in range(16))
    ka16 = KA_POST + KA_ROUND_A + KA_ROUND4 + KA_PRE
    out = np.zeros(19, dtype=np.float32)
    for b in range(3):
        n, sg2 = (128, s512) if b == 2 else (64, s256)
        out[b] = _round_up_f32(u * (ka128 if b == 2 else ka64) * np.sqrt(sg2 * n))
        gb = u * sg2 * np.sqrt(2 * n)
        if b == 2:
            out[3 + b], out[6 + b] = _round_up_f32(gb * (5 * GH + GQ)), 0.0
        else:
            out[3 + b], out[6 + b] = _round_up_f32(gb * GH * GQ), _round_up_f32(gb * (7 * GH + GQ))
    out[9] = 2.0 ** -70
    for b in range(3):
        out[10 + b] = _round_up_f32(u * ka16 * np.sqrt(s64 * 16))
        gs = u * s64 * np.sqrt(2.0 * 16)
        if b == 2:
            out[13 + b], out[16 + b] = _round_up_f32(gs * (5 * GH + GQ)), 0.0
        else:
            out[13 + b], out[16 + b] = _round_up_f32(gs * GH * GQ), _round_up_f32(gs * (7 * GH + GQ))
    return out


def run(pcm, all_short=False):
    """pcm: float32 mono stream (multiple of 512).  Returns coefs [frames,512], eps [frames,3], bands [frames,512].
    all_short: every band in short blocks (any non-zero fixed block modes) instead of [0,0,0]."""
    pcm = np.ascontiguousarray(pcm, dtype=np.float32)
    frames = len(pcm) // 512
    fp = C.POINTER(C.c_float)
    co = np.zeros((frames, 512), dtype=np.float32)
    ep = np.zeros((frames, 3), dtype=np.float32)
    bd = np.zeros((frames, 512), dtype=np.float32)
    lib().spec_model_stream(pcm.ctypes.data_as(fp), frames, co.ctypes.data_as(fp), ep.ctypes.data_as(fp), bd.ctypes.data_as(fp), 1 if all_short else 0)
    return co, ep, bd


def reference_coefs(pcm, modes=(0, 0, 0)):
    """The reference's coefficients for the given fixed block modes (through the oracle's stage entry points)."""
    pcm = np.ascontiguousarray(pcm, dtype=np.float32)
    L = O.lib()
    frames = len(pcm) // 512
    fp = C.POINTER(C.c_float)
    st = O.EncState()
    co = np.zeros((frames, 512), dtype=np.float32)
    modes = (C.c_int * 3)(*modes)
    b = np.zeros(512, dtype=np.float32)
    for f in range(frames):
        L.c1o_qmf_analysis_frame(C.byref(st), pcm[512 * f:].ctypes.data_as(fp), b.ctypes.data_as(fp))
        L.c1o_mdct_frame(C.byref(st), b.ctypes.data_as(fp), modes, co[f].ctypes.data_as(fp))
    return co
