#!/usr/bin/env python3
"""tools/spec_predictor_sim.py -- how well does the run-level predictor of c1_k_spec.hip ("is this material worth
speculating on?") foresee the fraction of units the guards of the speculative path flag?  CPU only: the model of the
binary32 analysis (tests/model/spec_model.c), the model of the guards (tests/model/pack_model.c), the oracle's
allocation.  Prints, per signal class, the measured flag rate next to the predicted number of doubtful decisions per
unit P (the kernel defers a run to the exact kernels when P exceeds its threshold)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib as O            # noqa: E402
import pack_model_lib as P        # noqa: E402
import spec_model_lib as M        # noqa: E402


def xorshift(seed, n):
    s = np.uint32(seed)
    out = np.empty(n, dtype=np.float64)
    s = int(seed)
    for i in range(n):
        s ^= (s << 13) & 0xffffffff
        s ^= s >> 17
        s ^= (s << 5) & 0xffffffff
        out[i] = s / 4294967296.0 * 2 - 1
    return out


def partials(frames, seg_seed):
    t = np.arange(frames * 512, dtype=np.float64)
    seg = (seg_seed * 2654435761 + 7) & 0xffffffff
    f0 = 55.0 * 2 ** ((seg % 61) / 12.0)
    f1 = f0 * (2.0 + ((seg >> 8) % 5))
    f2 = 3000.0 + ((seg >> 16) % 9000)
    w = lambda f: 2 * np.pi * f / 44100.0
    v = (0.45 * np.sin(w(f0) * t) + 0.12 * np.sin(w(f1) * t + 1.0) + 0.02 * np.sin(w(f2) * t + 2.0)) * (1.0 + 0.3 * np.sin(w(0.7) * t))
    return v.astype(np.float32)


def music(frames, seed):
    """harmonics over a noise floor"""
    rng = np.random.default_rng(seed)
    t = np.arange(frames * 512, dtype=np.float64)
    v = 0.002 * rng.standard_normal(frames * 512)
    f0 = 220.0
    for k in range(1, 12):
        v += 0.3 / k * np.sin(2 * np.pi * f0 * k / 44100.0 * t + k)
    return v.astype(np.float32)


def predictor(sfi, eps, bias=1.0, budget=1136.0, iters=3):
    """the kernel's estimate: water level L of the greedy allocation (Newton from the left on the convex spend function),
    then P = expected doubtful mantissas + expected open scale-factor indices"""
    size = P.SPECS.astype(np.float64)
    act = sfi > 0
    l = sfi / 3.0 - 21.0                     # log2 SCALE_FACTORS[sfi]
    lb = bias * l                            # log2 of the biased table
    eb = eps[P.BAND_OF_BFU].astype(np.float64)
    if not act.any():
        return 0.0
    L = (np.sum(size[act] * lb[act]) - budget) / np.sum(size[act])
    for _ in range(iters):
        bits = np.clip(lb - L, 0.0, 16.0) * act
        live = act & (lb - L > 0) & (lb - L < 16)
        spend = np.sum(size * bits)
        n = np.sum(size[live])
        if n <= 0:
            break
        L += (spend - budget) / n
    bits = np.clip(lb - L, 0.0, 16.0) * act
    coded = act & (bits >= 1.0)
    p_m = np.sum(size[coded] * eb[coded] * 2.0 ** (bits[coded] - l[coded]))        # 2 et per coefficient, et = eps 2^(bits-1) / SF
    p_s = np.sum(np.minimum(1.0, 9.7 * eb[act] * 2.0 ** (-l[act])))
    return p_m + p_s


def evaluate(name, pcm, bias=1.0):
    co, ep, _ = M.run(pcm)
    frames = co.shape[0]
    flags, preds, sfo, dbt = [], [], [], []
    for f in range(1, frames):
        slots = P.to_slots(co[f])
        sfi, unstable = P.sf_guard(slots, ep[f])
        n, wl, _ = P.allocate(co[f], (0, 0, 0), bias)
        _, doubtful, _ = P.quantize(slots, ep[f], sfi, wl, n)
        flags.append(unstable or doubtful)
        sfo.append(unstable)
        dbt.append(doubtful)
        preds.append(predictor(sfi, ep[f], bias))
    flags, preds = np.array(flags), np.array(preds)
    print('%-22s flagged %5.1f %% (sf %5.1f %%, mantissa %5.1f %%)   P: median %8.3f  p10 %8.3f  p90 %8.3f   1-exp(-P) mean %5.1f %%' % (
        name, 100 * flags.mean(), 100 * np.mean(sfo), 100 * np.mean(dbt), np.median(preds), np.percentile(preds, 10), np.percentile(preds, 90),
        100 * np.mean(1 - np.exp(-preds))))
    return flags, preds


if __name__ == '__main__':
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    n = F * 512
    evaluate('white 0.5', (xorshift(1, n) * 0.5).astype(np.float32))
    evaluate('white 0.5 bias 2', (xorshift(1, n) * 0.5).astype(np.float32), 2.0)
    evaluate('white 0.5 bias 0.5', (xorshift(1, n) * 0.5).astype(np.float32), 0.5)
    evaluate('quiet white 0.003', (xorshift(9, n) * 0.003).astype(np.float32))
    evaluate('pink + bursts', O.gen_pinkT(3, n))
    for s in (0, 1, 2):
        evaluate('partials seg %d' % s, partials(F, s))
    evaluate('music', music(F, 5))
    evaluate('sine 1 kHz', (0.8 * np.sin(2 * np.pi * 1000 / 44100.0 * np.arange(n))).astype(np.float32))
