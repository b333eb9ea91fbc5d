"""CPU: the interval the speculative transient detector derives for the reference's transient score
(carta1_amd/csrc/c1_detect_bound.h, evaluated on the device by k_detect_decide<SPEC>), without a GPU.
The header is compiled with gcc (tests/model/detect_bound.c).  Magnitude spectra within the bound Delta of the oracle's
-- the binary64 DFT rounded to binary32, and the oracle's own magnitudes pushed by 0.95 Delta in the directions that
move each feature most -- go through the ten sums and the interval; the oracle's score (c1o_transient_score,
transient.js:63-226) must lie inside every time, and for honest magnitudes the interval must be narrow enough to decide."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from test_spec_bound import signals

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'model', 'detect_bound.c')
HDR = os.path.join(ROOT, 'carta1_amd', 'csrc', 'c1_detect_bound.h')
SO = os.path.join(ROOT, 'oracle', '_build', 'libdetect_bound.so')
fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
_lib = None


def model():
    global _lib
    if _lib is None:
        if not os.path.exists(SO) or max(os.path.getmtime(SRC), os.path.getmtime(HDR)) > os.path.getmtime(SO):
            os.makedirs(os.path.dirname(SO), exist_ok=True)
            subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-std=c11', '-ffp-contract=off', '-fno-fast-math', '-o', SO, SRC, '-lm'])
        L = C.CDLL(SO)
        L.detm_record.argtypes = [fp, fp, fp, fp]
        L.detm_interval.argtypes = [fp, fp, C.c_int, C.c_double, dp, dp]
        L.detm_interval.restype = C.c_int
        L.detm_constant.argtypes = [C.c_int]
        L.detm_constant.restype = C.c_double
        _lib = L
    return _lib


def oracle_bands_and_mags(pcm):
    L = O.lib()
    frames = len(pcm) // 512
    bands = np.zeros((frames, 512), np.float32)
    mags = np.zeros((frames, 256), np.float32)
    state = O.EncState()                                # zero = a fresh BufferPool
    for f in range(frames):
        L.c1o_qmf_analysis_frame(C.byref(state), O._fp(np.ascontiguousarray(pcm[f * 512:(f + 1) * 512])), O._fp(bands[f]))
        L.c1o_transient_mags(O._fp(bands[f]), O._fp(mags[f]))
    return bands, mags


def delta_of(bands_f):
    """Delta per band as the device forms it (tfft_spec): K u theta sqrt(n) ||x|| + eabs, 0 for an all-zero band"""
    M = model()
    k128, k256, theta, eabs = (M.detm_constant(i) for i in range(4))
    out = np.zeros(3, np.float32)
    for b, (o, n, k) in enumerate(((0, 128, k128), (128, 128, k128), (256, 256, k256))):
        x = bands_f[o:o + n].astype(np.float64)
        if np.any(x != 0):
            out[b] = np.float32((k * 2.0 ** -24 * theta * np.sqrt(n) * np.sqrt(np.sum(x * x)) + eabs) * 1.00001)
    return out


def dft_mags(bands_f):
    out = np.zeros(256, np.float32)
    for o, n, m in ((0, 128, 0), (128, 128, 64), (256, 256, 128)):
        z = np.fft.fft(bands_f[o:o + n].astype(np.float64))
        out[m:m + n // 2] = np.abs(z[:n // 2]).astype(np.float32)
    return out


def pushed(mags, delta, rng, how):
    """the oracle's magnitudes moved by 0.95 Delta (l2, per band) in a chosen direction, kept >= 0"""
    out = mags.astype(np.float64).copy()
    for b, (m, n) in enumerate(((0, 64), (64, 64), (128, 128))):
        c = out[m:m + n]
        if how == 'random':
            d = rng.standard_normal(n)
        elif how == 'up':
            d = np.ones(n)
        elif how == 'down':
            d = -np.ones(n)
        elif how == 'tilt':                               # energy from the lower half to the upper one
            d = np.concatenate([-np.ones(n // 2), np.ones(n // 2)])
        elif how == 'smallest':                           # everything on the smallest bin: the flatness moves most
            d = np.zeros(n)
            d[np.argmin(c)] = -1.0
        else:                                             # 'log': proportional to 1 / c
            d = -1.0 / np.maximum(c, 1e-30)
        nrm = np.sqrt(np.sum(d * d))
        if nrm > 0:
            c += 0.95 * float(delta[b]) * d / nrm
        out[m:m + n] = np.maximum(c, 0.0)
    return out.astype(np.float32)


CASES = [s for s in signals() if s[0] in ('white', 'pink_bursts', 'sine_1k', 'two_tones_loud', 'impulses', 'dc', 'tiny', 'wide_dynamic')]


@pytest.mark.parametrize('name,pcm', CASES, ids=[s[0] for s in CASES])
def test_interval_contains_the_oracle_score(name, pcm):
    M, L = model(), O.lib()
    log1p10 = float(np.log1p(10.0))
    bands, mags = oracle_bands_and_mags(pcm[:40 * 512])
    rng = np.random.default_rng(5)
    frames = len(bands)
    deltas = [delta_of(bands[f]) for f in range(frames)]
    widths, undecided = [], 0
    for how in ('dft', 'random', 'up', 'down', 'tilt', 'smallest', 'log'):
        spec = [dft_mags(bands[f]) if how == 'dft' else pushed(mags[f], deltas[f], rng, how) for f in range(frames)]
        recs = np.zeros((frames, 40), np.float32)
        zero = np.zeros(256, np.float32)
        for f in range(frames):
            M.detm_record(O._fp(spec[f]), O._fp(spec[f - 1] if f else zero), O._fp(deltas[f]), O._fp(recs[f]))
        for f in range(frames):
            for b, (m, n) in enumerate(((0, 64), (64, 64), (128, 128))):
                prev_m = mags[f - 1] if f else zero
                score = L.c1o_transient_score(O._fp(mags[f][m:m + n].copy()), O._fp(prev_m[m:m + n].copy()), n)
                lo, hi = C.c_double(), C.c_double()
                ok = M.detm_interval(O._fp(recs[f]), O._fp(recs[f - 1]) if f else None, b, log1p10, C.byref(lo), C.byref(hi))
                if ok:
                    assert lo.value <= score <= hi.value, (name, how, f, b, lo.value, score, hi.value)
                    if how == 'dft':
                        widths.append(hi.value - lo.value)
                elif how == 'dft':
                    undecided += 1
    if name in ('white', 'pink_bursts', 'wide_dynamic'):
        assert undecided <= 0.05 * frames * 3 and np.median(widths) < 0.02, (name, undecided, float(np.median(widths)))


def test_specials_are_never_decided():
    M = model()
    rec = np.zeros(40, np.float32)
    lo, hi = C.c_double(), C.c_double()
    mags = np.abs(np.random.default_rng(1).standard_normal(256)).astype(np.float32)
    for bad in (np.nan, np.inf):
        for where in ('mag', 'delta'):
            m, d = mags.copy(), np.full(3, 1e-6, np.float32)
            if where == 'mag':
                m[5] = bad; m[70] = bad; m[200] = bad
            else:
                d[:] = bad
            M.detm_record(O._fp(m), O._fp(mags), O._fp(d), O._fp(rec))
            for b in range(3):
                assert M.detm_interval(O._fp(rec), None, b, 2.3978952727983707, C.byref(lo), C.byref(hi)) == 0
    # magnitudes around the validity threshold 1e-10 of the flatness: which bins count is not certain
    m = np.full(256, 1.2e-10, np.float32)
    M.detm_record(O._fp(m), O._fp(m), O._fp(np.full(3, 5e-11, np.float32)), O._fp(rec))
    assert M.detm_interval(O._fp(rec), None, 0, 2.3978952727983707, C.byref(lo), C.byref(hi)) == 0
    # digital silence is decided: every feature is exactly zero
    z = np.zeros(256, np.float32)
    M.detm_record(O._fp(z), O._fp(z), O._fp(np.zeros(3, np.float32)), O._fp(rec))
    assert M.detm_interval(O._fp(rec), O._fp(rec), 0, 2.3978952727983707, C.byref(lo), C.byref(hi)) == 1
    assert abs(lo.value) < 1e-5 and abs(hi.value) < 1e-5
