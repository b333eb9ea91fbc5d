"""The error bound of the speculative binary32 analysis, checked on the CPU: the model of the kernel
(tests/model/spec_model.c, operation for operation what carta1_amd/csrc/c1_k_spec.hip issues) against the
reference's coefficients from the oracle.  The bound is a worst-case bound: on every signal below, including ones
built to line roundings up, the observed error stays far inside it."""
import os
import re

import numpy as np
import pytest

import oracle_lib as O
import spec_model_lib as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAND = np.repeat([0, 1, 2], [128, 128, 256])


def signals():
    n = 96 * 512
    t = np.arange(n)
    rng = np.random.default_rng(7)
    yield 'white', O.gen_white(1, n)
    yield 'pink_bursts', O.gen_pinkT(3, n)
    yield 'sine_1k', (0.5 * np.sin(2 * np.pi * 1000 * t / 44100)).astype(np.float32)
    yield 'two_tones_loud', (0.9 * np.sin(2 * np.pi * 61.3 * t / 44100) + 0.09 * np.sin(2 * np.pi * 15000.7 * t / 44100)).astype(np.float32)
    yield 'impulses', (rng.random(n) < 0.004).astype(np.float32) * rng.choice([-1.0, 1.0], n).astype(np.float32)
    yield 'square_nyquist', np.where(t % 2 == 0, 0.75, -0.75).astype(np.float32)
    yield 'dc', np.full(n, 0.999, dtype=np.float32)
    yield 'sign_noise', rng.choice([-1.0, 1.0], n).astype(np.float32)          # every sample at full scale
    yield 'tiny', (rng.standard_normal(n) * 1e-6).astype(np.float32)          # around the smallest scale factor
    yield 'huge', (rng.standard_normal(n) * 1e6).astype(np.float32)
    yield 'wide_dynamic', (rng.standard_normal(n) * np.exp(rng.uniform(-18, 2, n))).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32)
    x[::7] = np.float32(1 + 2 ** -23)                                         # mantissas that round at every product
    yield 'rounding_bait', x


@pytest.mark.parametrize('all_short', [False, True], ids=['long', 'short'])
@pytest.mark.parametrize('name,pcm', list(signals()), ids=[s[0] for s in signals()])
def test_bound_dominates_the_observed_error(name, pcm, all_short):
    co, eps, _ = M.run(pcm, all_short)
    ref = M.reference_coefs(pcm, (2, 2, 3) if all_short else (0, 0, 0))
    err = np.abs(co.astype(np.float64) - ref.astype(np.float64))
    ratio = err / eps[:, BAND].astype(np.float64)
    assert np.isfinite(eps).all()
    assert ratio.max() < 0.25, (name, float(ratio.max()))     # worst case seen: ~0.05; the bound is a proof, this is a sanity margin


def test_silence_and_specials_never_pass_unnoticed():
    n = 8 * 512
    co, eps, _ = M.run(np.zeros(n, dtype=np.float32))
    assert (co == 0).all() and (eps > 0).all() and (eps < 1e-20).all()
    x = O.gen_white(1, n)
    x[1000] = np.inf
    _, eps, _ = M.run(x)
    assert not np.isfinite(eps[1]).any() and not np.isfinite(eps[2]).any()    # the frame holding it and the next one
    x = O.gen_white(1, n)
    x[1000] = np.nan
    _, eps, _ = M.run(x)
    assert np.isnan(eps[1]).all() and np.isnan(eps[2]).all()
    x = O.gen_white(1, n) * np.float32(1e30)                                  # energies overflow before anything else does
    _, eps, _ = M.run(x)
    assert not np.isfinite(eps[1:]).any()


def test_product_constants_match_the_ones_tested_here():
    src = open(os.path.join(ROOT, 'carta1_amd', 'csrc', 'c1_api.hip')).read()
    val = lambda name: float(re.search(name + r'\s*=\s*([0-9.]+)', src).group(1))
    assert val('kSpecGH') == M.GH and val('kSpecGQ') == M.GQ
    assert val('kSpecKAPost') == M.KA_POST and val('kSpecKAPre') == M.KA_PRE
    assert val('kSpecKARoundA') == M.KA_ROUND_A and val('kSpecKARound4') == M.KA_ROUND4 and val('kSpecKARound2') == M.KA_ROUND2
    assert val('kSpecTheta') == M.THETA


def test_qmf_gain_constants_cover_the_prototype():
    """gH and gQ of the bound depend only on the QMF taps: recompute them (tools/spec_constants.py) and require the
    compiled-in values to be no smaller."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('spec_constants', os.path.join(ROOT, 'tools', 'spec_constants.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gh, gq = mod.qmf_gains()
    assert gh <= M.GH and gq <= M.GQ, (gh, gq)
    assert M.GH < 1.01 * gh and M.GQ < 1.02 * gq          # and not wastefully larger
