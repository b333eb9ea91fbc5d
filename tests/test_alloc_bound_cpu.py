"""CPU: the mathematics of the allocation's second lower bound (k_alloc_bound) without a GPU.
allocateBits (bitallocation.js:74-142) runs a greedy heap for eight candidate BFU counts and keeps the smallest total
distortion.  The library skips a candidate when a Lagrangian lower bound on its total is strictly above a total that
was really computed.  Here the reference's heap and its total are restated in Python (the C oracle only returns the
winner), the bound is formed exactly as the kernel forms it -- explicit inner minimum at clamp(floor(log2(a / lambda)),
2, 16) against the Float32 zero-bit term, multiplier by bisection on the relaxed spend, 2^-44 slack -- and checked:
bound <= total for every candidate, for multipliers good and bad; winner of the pruned two-round scheme == winner of
all eight heaps, which in turn == the oracle's c1o_allocate."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_lib as O

SPECS = [8, 8, 8, 8, 4, 4, 4, 4, 8, 8, 8, 8, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 7, 7, 7, 7, 9, 9, 9, 9, 10, 10, 10, 10,
         12, 12, 12, 12, 12, 12, 12, 12, 20, 20, 20, 20, 20, 20, 20, 20]
START = [sum(SPECS[:b]) for b in range(52)]
AMOUNTS = [20, 28, 32, 36, 40, 44, 48, 52]
WLB = [0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
DB = [WLB[i + 1] - WLB[i] for i in range(15)]
IP2 = [2.0 ** -b for b in range(17)]
DDF = [2.0 - IP2[2]] + [IP2[WLB[i]] - IP2[WLB[i + 1]] for i in range(1, 15)]
f32 = np.float32


def _sift(hi, hp, i, n):                                  # siftDown, bitallocation.js:314-341
    iv, pv = hi[i], hp[i]
    while True:
        l, r, m, mp = 2 * i + 1, 2 * i + 2, i, pv
        if l < n and hp[l] > mp:
            m, mp = l, hp[l]
        if r < n and hp[r] > mp:
            m = r
        if m == i:
            break
        hi[i], hp[i], i = hi[m], hp[m], m
    hi[i], hp[i] = iv, pv


def distribute(n, rem, bsf, sfi):                         # distributeBitsRDO, :203-281
    wl, hi, hp = [0] * n, [], []
    for b in range(n):
        if sfi[b]:
            hi.append(b)
            hp.append(f32(bsf[sfi[b]] * DDF[0] / DB[0]))
    hs = len(hi)
    for i in range((hs >> 1) - 1, -1, -1):
        _sift(hi, hp, i, hs)
    def pop():
        nonlocal hs
        hi[0], hp[0] = hi[hs - 1], hp[hs - 1]
        hs -= 1
        if hs > 0:
            _sift(hi, hp, 0, hs)
    while rem > 0 and hs > 0:
        b = hi[0]
        cur = wl[b]
        cost = DB[cur] * SPECS[b]
        if cost > rem or cost <= 0:
            pop()
            continue
        rem -= cost
        nxt = cur + 1
        wl[b] = nxt
        if nxt < 15 and DB[nxt] > 0:
            hp[0] = f32(bsf[sfi[b]] * DDF[nxt] / DB[nxt])
            _sift(hi, hp, 0, hs)
        else:
            pop()
    return wl


def total(n, wl, sfi, bsf, z):                            # calculateTotalDistortion, :157-190
    t = 0.0
    for i in range(n):
        bits = WLB[wl[i]]
        if bits == 0:
            t += float(z[i])
        elif sfi[i]:
            t += bsf[sfi[i]] * IP2[bits] * SPECS[i]
    for i in range(n, 52):
        t += float(z[i])
    return t


def relaxed_bits(y):                                      # as the kernel's search: what the relaxed problem gives a BFU
    return 0 if y <= 0.19264507 else (2 if y < 3 else min(int(math.floor(y)), 16))


def bound(n, lam, sfi, bsf, z):                           # as k_alloc_bound forms it
    B = 212 * 8 - 40 - 10 * n
    P = 0.0
    for b in range(52):
        if not sfi[b]:
            continue
        zb = float(z[b])
        g = zb
        if b < n:
            a = bsf[sfi[b]]
            e = min(max(math.frexp(a / lam)[1] - 1, 2), 16)            # floor(log2(a / lambda)) clamped
            h = a * IP2[e] * SPECS[b] + lam * SPECS[b] * e
            g = min(h, zb)
        P += g
    M = lam * B
    return (P - M) - 2.0 ** -44 * (P + M)


def multiplier(n, sfi, bsf, iters=10):
    B = 212 * 8 - 40 - 10 * n
    la = {b: math.log2(bsf[sfi[b]]) for b in range(n) if sfi[b]}
    if not la:
        return 1.0
    lo, hi = min(la.values()) - 17.0, max(la.values())
    for _ in range(iters):
        x = 0.5 * (lo + hi)
        used = sum(SPECS[b] * relaxed_bits(v - x) for b, v in la.items())
        lo, hi = (x, hi) if used > B else (lo, x)
    return 2.0 ** hi


def families(rng, per):
    out = [rng.randint(0, 64, size=52) for _ in range(per)]
    out += [np.clip(rng.randint(20, 50) + rng.randint(-2, 3, size=52), 0, 63) for _ in range(per)]
    out += [np.clip(rng.randint(40, 63) - rng.uniform(0.2, 1.5) * np.arange(52) + rng.randint(-3, 4, size=52), 0, 63).astype(int) for _ in range(per)]
    for _ in range(per):
        v = 3 + rng.randint(0, 4, size=52)
        v[rng.choice(52, size=rng.randint(1, 6), replace=False)] = rng.randint(35, 64)
        out.append(v)
    for _ in range(per):
        v = rng.randint(0, 64, size=52)
        v[rng.randint(8, 52):] = 0
        out.append(v)
    out += [np.full(52, rng.randint(0, 64)) for _ in range(max(1, per // 2))]
    out += [np.zeros(52, int), np.full(52, 63), np.full(52, 1)]
    return out


@pytest.mark.parametrize('bias', [1.0, 2.0, 0.5])
def test_bound_never_exceeds_the_total_and_pruning_keeps_the_winner(bias):
    L = O.lib()
    L.c1o_scale_factors.restype = C.POINTER(C.c_double)
    SF = np.ctypeslib.as_array(L.c1o_scale_factors(), shape=(64,)).copy()
    bsf = np.array(O.biased_table(bias), dtype=np.float64)
    rng = np.random.RandomState(int(100 * bias))
    heaps_all = heaps_pruned = 0
    for sfi in families(rng, 6):
        sfi = [int(s) for s in sfi]
        z = [f32(bsf[s] * 2.0 * SPECS[b]) if s else f32(0) for b, s in enumerate(sfi)]
        tot = []
        for n in AMOUNTS:
            tot.append(total(n, distribute(n, 212 * 8 - 40 - 10 * n, bsf, sfi), sfi, bsf, z))
        win = min(range(8), key=lambda c: (tot[c], c))
        # the oracle's allocateBits agrees with the restated heap (it works from coefficients: give it one per BFU)
        coefs = np.zeros(512, np.float32)
        for b, s_ in enumerate(sfi):
            if s_:
                coefs[START[b]] = np.float32(SF[s_] * 0.95)
        nb, wl_o, sf_o = C.c_int(0), (C.c_int * 52)(), (C.c_int * 52)()
        L.c1o_allocate(O._fp(coefs), (C.c_int * 3)(0, 0, 0), bsf.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nb), wl_o, sf_o)
        assert list(sf_o) == sfi
        assert nb.value == AMOUNTS[win], (sfi, tot)
        # bounds: valid for any multiplier, sharp for the searched one
        lbs = []
        for c, n in enumerate(AMOUNTS):
            lam = multiplier(n, sfi, bsf)
            for k in (1e-6, 0.3, 1.0, 2.5, 1e6):
                assert bound(n, lam * k, sfi, bsf, z) <= tot[c], (c, k, sfi)
            lbs.append(bound(n, lam, sfi, bsf, z))
        # the two-round scheme of c1k_launch_allocate (52 BFUs first)
        best, evaluated = tot[7], {7}
        alive = [c for c in range(7) if not lbs[c] > best]
        if alive:
            c0 = min(alive, key=lambda c: lbs[c])
            evaluated.add(c0)
            best = min(best, tot[c0])
            evaluated |= {c for c in alive if not lbs[c] > best}
        assert min(evaluated, key=lambda c: (tot[c], c)) == win
        heaps_all += 8
        heaps_pruned += len(evaluated)
    assert heaps_pruned < 0.45 * heaps_all
