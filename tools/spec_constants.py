#!/usr/bin/env python3
"""Constants of the speculative path's error bound that depend only on the QMF prototype.

gH  bound on the l2 gain of one branch of the two-band QMF analysis (x -> low or x -> high, decimated):
    ||low||_2 = ||E * x_odd + O * x_even||_2 <= sqrt(max|E(w)|^2 + max|O(w)|^2) ||x||_2, E/O the polyphase components.
gQ  accumulated rounding of the binary32 convolution as c1_k_spec.hip orders it, in units of u * ||input||_2:
    every fused multiply-add rounds its partial sum; the partial sums of a chain, seen over all outputs, are the input
    filtered by the partial filter, so their l2 norm is at most (max |partial response|) * ||input phase||.  Summed
    over the roundings of both chains, their sum and the centre tap, for the even and the odd branch:
    gQ = sqrt(gE^2 + gO^2).

The maxima over frequency are taken on a grid of 2^16 points; the derivative of a response is bounded by
sum |h_j| j, so the grid maximum is increased by that bound times half the grid step (a rigorous upper bound).
"""
import json
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def taps():
    t = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'tables.json')))
    return np.array([struct.unpack('>f', bytes.fromhex(h))[0] for h in t['qmf_even_f32']], dtype=np.float64)


GRID = 1 << 16
_W = np.linspace(0.0, np.pi, GRID + 1)


def gain(E, idx):
    """rigorous upper bound of max_w |sum_{j in idx} E[j] e^{-i w j}|"""
    idx = np.array(idx)
    resp = np.abs(np.exp(-1j * np.outer(_W, idx)) @ E[idx]).max()
    lipschitz = float(np.sum(np.abs(E[idx]) * idx))
    return resp + lipschitz * (np.pi / GRID) / 2


def chain_gain(E):
    """sum of the gains of every rounded partial result of one 24-tap sum as the kernel orders it:
    chain A taps 0..11 ascending, chain B taps 23..13 descending, their sum, then the centre tap 12"""
    A = list(range(12))
    B = list(range(23, 12, -1))
    g = sum(gain(E, A[:k]) for k in range(1, 13)) + sum(gain(E, B[:k]) for k in range(1, 12))
    return g + gain(E, A + B) + gain(E, A + B + [12])


def qmf_gains():
    E = taps()
    O = E[::-1].copy()
    gh = float(np.sqrt(gain(E, list(range(24))) ** 2 + gain(O, list(range(24))) ** 2))
    # the odd branch runs the same chains over the mirrored taps: same partial filters up to a delay
    ge = chain_gain(E)
    gq = float(np.sqrt(2.0) * ge)
    return gh, gq


if __name__ == '__main__':
    gh, gq = qmf_gains()
    print('gH = %.5f   gQ = %.5f   (compiled in: c1_api.hip kSpecGH, kSpecGQ)' % (gh, gq))
