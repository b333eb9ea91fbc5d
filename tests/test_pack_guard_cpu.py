"""The guards of the speculative path under attack, on the CPU.

The speculative path may hand a sound unit on without the exact kernels only when every decision taken on its binary32
coefficients F -- which integer |x| norm + 0.5 truncates to (quantization.js:43-53), which scale-factor interval a BFU's
maximum falls in (bitallocation.js:290-299) -- is the decision the reference takes on ITS coefficients R, for every R
with |F - R| <= eps.  tests/model/pack_model.c restates the two guards operation for operation (the GPU twin,
tests/test_gpu_pack_guard.py, checks kernel == model on the same inputs).  Here the coefficients are moved by up to
0.95 eps in the direction that flips a truncation or a scale-factor index, with bounds from realistic to absurdly loose:
 * soundness: every unit whose mantissas or indices differ from the reference's is flagged;
 * the contract behind it: an accepted unit has the reference's mantissas for R = F - eps and R = F + eps as well;
 * no vacuous guard: a coefficient placed just outside the guard band of a boundary is accepted, just inside is not."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import pack_model_lib as P
import spec_model_lib as M

LONG = (0, 0, 0)


def material():
    n = 48 * 512
    t = np.arange(n)
    rng = np.random.default_rng(11)
    yield 'white', O.gen_white(1, n)
    yield 'pink_bursts', O.gen_pinkT(3, n)
    yield 'tones', (0.5 * np.sin(2 * np.pi * 997 * t / 44100) + 0.05 * np.sin(2 * np.pi * 7919 * t / 44100)).astype(np.float32)
    yield 'wide_dynamic', (rng.standard_normal(n) * np.exp(rng.uniform(-12, 0, n))).astype(np.float32)
    yield 'tiny', (rng.standard_normal(n) * 3e-6).astype(np.float32)


def reference_units(pcm, modes=LONG):
    """reference coefficients, allocation and mantissas (slot order) per frame, through the oracle"""
    ref = M.reference_coefs(pcm, modes)
    out = []
    for f in range(ref.shape[0]):
        n, wl, sfi = P.allocate(ref[f], modes)
        out.append((ref[f], n, wl, sfi, P.reference_quantize(ref[f], modes, n, wl, sfi)))
    return out


def adversarial(slots_ref, eps, sfi, wl, nbfu, strength, rng):
    """F = R + d, |d| <= strength * eps_band, each coefficient pushed across the nearest truncation boundary of the
    reference's quantizer when that is within reach, else to the far end of its interval in a random direction"""
    sf = P.scale_factors()
    F = slots_ref.astype(np.float64).copy()
    for b in range(52):
        w = int(wl[b]) if b < nbfu else 0
        e = float(eps[P.BAND_OF_BFU[b]]) * strength
        lo, hi = P.FIRST[b], P.FIRST[b + 1]
        x = F[lo:hi]
        if w == 0 or sfi[b] == 0:
            x += rng.choice([-1.0, 1.0], hi - lo) * e
            continue
        norm = ((1 << w) - 1) / sf[sfi[b]]
        a = np.abs(x) * norm + 0.5
        fr = a - np.floor(a)
        reach = e * norm
        down = fr <= reach                      # |x| - e truncates one lower
        up = (1.0 - fr) <= reach                # |x| + e truncates one higher
        step = np.where(down, -1.0, np.where(up, 1.0, rng.choice([-1.0, 1.0], hi - lo)))
        x += np.sign(x + (x == 0)) * step * e
    return F.astype(np.float32)


@pytest.mark.parametrize('scale', [1.0, 30.0, 1000.0], ids=['bound', 'bound_x30', 'bound_x1000'])
@pytest.mark.parametrize('name,pcm', list(material()), ids=[m[0] for m in material()])
def test_every_unit_whose_mantissas_would_change_is_flagged(name, pcm, scale):
    """the attack: bounds as the kernel computes them (and 30 x, 1000 x looser: flips become common), coefficients moved
    0.95 of the bound towards the nearest boundary"""
    _, eps_all, _ = M.run(pcm)
    rng = np.random.default_rng(5)
    changed = flagged_changed = accepted = 0
    for f, (ref, n, wl, sfi, q_ref) in enumerate(reference_units(pcm)):
        if f == 0:
            continue
        eps = (eps_all[f] * scale).astype(np.float32)
        slots = P.to_slots(ref)
        F = adversarial(slots, eps, sfi, wl, n, 0.95, rng)
        assert (np.abs(F.astype(np.float64) - slots) <= eps[P.BAND_OF_BFU][np.repeat(np.arange(52), P.SPECS)]).all()      # the attack stays inside the bound
        q, doubtful, _, doubt = P.quantize(F, eps, sfi, wl, n, per_slot=True)
        differs = q != q_ref
        changed += int(differs.sum())
        flagged_changed += int((differs & doubt).sum())
        accepted += not doubtful
        # mantissa by mantissa (with loose bounds some mantissa of nearly every unit is doubtful, which would hide a hole
        # in the guard of another); the kernel flags the unit when any is
        assert not (differs & ~doubt).any(), (name, f, np.nonzero(differs & ~doubt)[0][:4])
        assert doubtful == bool(doubt.any())
    assert flagged_changed == changed
    if scale >= 30.0 and name in ('white', 'pink_bursts', 'tones', 'wide_dynamic'):
        assert changed > 0, 'the attack flipped nothing: it does not test the guard'


@pytest.mark.parametrize('name,pcm', list(material()), ids=[m[0] for m in material()])
def test_accepted_units_are_right_for_every_reference_within_the_bound(name, pcm):
    """the contract: not flagged => the reference's quantizer gives these very mantissas for R = F - eps and R = F + eps
    (the truncation is monotone in |x|, so for everything in between as well)"""
    co, eps_all, _ = M.run(pcm)
    checked = 0
    for f in range(1, co.shape[0]):
        for scale in (1.0, 100.0):
            eps = (eps_all[f] * scale).astype(np.float32)
            slots = P.to_slots(co[f])
            sfi, unstable = P.sf_guard(slots, eps)
            n, wl, _ = P.allocate(co[f], LONG)
            q, doubtful, _ = P.quantize(slots, eps, sfi, wl, n)
            if doubtful or unstable:
                continue
            checked += 1
            e = eps[P.BAND_OF_BFU][np.repeat(np.arange(52), P.SPECS)].astype(np.float64)
            for sign in (-1.0, 1.0):
                # both ends of the interval, as doubles (the reference's coefficients are binary32, any of them lies between)
                R = np.abs(slots.astype(np.float64)) + sign * e
                R = np.copysign(np.maximum(R, 0.0), slots)
                qr = quantize_double(R, n, wl, sfi)
                assert np.array_equal(qr, q), (name, f, scale, sign, np.nonzero(qr != q)[0][:4])
    assert checked > 0 or name == 'tones'


def quantize_double(x, nbfu, wl, sfi):
    """quantization.js:34-56 on doubles (the reference's arithmetic is binary64): trunc(x norm +- 0.5), clamped"""
    sf = P.scale_factors()
    q = np.zeros(512, dtype=np.int32)
    for b in range(52):
        w = int(wl[b]) if b < nbfu else 0
        if w == 0 or sfi[b] == 0:
            continue
        rng_ = (1 << w) - 1
        norm = rng_ / sf[sfi[b]]
        lo, hi = P.FIRST[b], P.FIRST[b + 1]
        v = x[lo:hi] * norm
        y = np.trunc(v + np.where(v >= 0, 0.5, -0.5))
        q[lo:hi] = np.clip(y, -rng_, rng_).astype(np.int32)
    return q


def test_guard_band_is_neither_vacuous_nor_blind():
    """one coefficient walked across a truncation boundary in steps of a fraction of the guard band: it is accepted (with the
    reference's mantissa on its side) outside the band, flagged inside, and never accepted with the wrong mantissa"""
    sf = P.scale_factors()
    sfi = np.zeros(52, dtype=np.int32)
    wl = np.zeros(52, dtype=np.int32)
    b, s, w = 10, 40, 7                     # BFU 10 (band 0), 8 bits
    sfi[b], wl[b] = s, w
    norm = ((1 << w) - 1) / sf[s]
    eps = np.array([sf[s] * 2.0 ** -13, 1e-30, 1e-30], dtype=np.float32)
    band = float(eps[0]) * norm             # the bound in units of the quantizer's steps (~0.03)
    k = 37
    seen = {'accepted_below': 0, 'flagged': 0, 'accepted_above': 0}
    for off in np.linspace(-3 * band, 3 * band, 241):
        x = np.float32((k + 0.5 + off) / norm)          # |x| norm + 0.5 = k + 1 + off
        slots = np.zeros(512, dtype=np.float32)
        slots[P.FIRST[b] + 3] = -x
        q, doubtful, _ = P.quantize(slots, eps, sfi, wl, 20)
        got = -q[P.FIRST[b] + 3]
        a = float(x) * norm + 0.5
        if doubtful:
            seen['flagged'] += 1
            assert abs(a - (k + 1)) < 1.6 * band + 1e-3, off          # flagged only near the boundary: the guard is not vacuous
        else:
            assert got == int(np.floor(a)), off
            assert abs(a - (k + 1)) > 0.999 * band, off               # and never accepted inside the bound's reach
            seen['accepted_below' if got == k else 'accepted_above'] += 1
    assert min(seen.values()) > 20, seen


@pytest.mark.parametrize('name,pcm', list(material()), ids=[m[0] for m in material()])
def test_scale_factor_guard_under_attack(name, pcm):
    """BFU maxima pushed 0.95 eps towards the nearest scale-factor boundary: an index that differs from the reference's
    (findScaleFactor on the reference's coefficients) is never handed on unflagged; and an unflagged unit has the
    reference's indices at both ends of the bound"""
    _, eps_all, _ = M.run(pcm)
    ref = M.reference_coefs(pcm, LONG)
    sf = P.scale_factors()
    wrong = caught = 0
    for scale in (1.0, 300.0):
        for f in range(1, ref.shape[0]):
            eps = (eps_all[f] * scale).astype(np.float32)
            slots = P.to_slots(ref[f]).astype(np.float64)
            want = np.zeros(52, dtype=np.int32)
            F = slots.copy()
            for b in range(52):
                lo, hi = P.FIRST[b], P.FIRST[b + 1]
                x = np.ascontiguousarray(slots[lo:hi].astype(np.float32))
                want[b] = O.lib().c1o_find_scale_factor(x.ctypes.data_as(C.POINTER(C.c_float)), hi - lo)
                m = np.abs(slots[lo:hi]).max()
                if m == 0:
                    continue
                e = 0.95 * float(eps[P.BAND_OF_BFU[b]])
                # nearest boundary of the table to the maximum: move every coefficient of the BFU that way
                below = sf[want[b] - 1] if want[b] > 0 else 0.0
                toward = -1.0 if (m - below) < (sf[want[b]] - m) else 1.0
                F[lo:hi] = np.sign(slots[lo:hi]) * np.maximum(np.abs(slots[lo:hi]) + toward * e, 0.0)
            got, unstable, opened = P.sf_guard(F.astype(np.float32), eps, per_bfu=True)
            differs = got != want
            wrong += int(differs.sum())
            caught += int((differs & opened).sum())
            # BFU by BFU (with loose bounds some BFU of nearly every unit is open, which would hide a hole in the guard of another)
            assert not (differs & ~opened).any(), (name, f, scale, np.nonzero(differs & ~opened)[0][:4])
            assert unstable == bool(opened.any())
    assert caught == wrong
    if name in ('white', 'pink_bursts', 'wide_dynamic'):
        assert wrong > 0, 'the attack moved no index: it does not test the guard'
