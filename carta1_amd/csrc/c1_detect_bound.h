// c1_detect_bound.h -- the speculative transient detector's interval for the reference's transient score
// (DESIGN.md 3c).  Shared by the device (k_detect_decide<SPEC> in c1_k_detect.hip) and by the CPU tests
// (tests/model/detect_bound.c compiles this header with gcc and checks the interval against the reference's score).
//
// The reference (analysis/transient.js:44-226) scores a band from the Float32 magnitude spectra c (this frame) and p
// (previous frame) of its transient FFT.  The speculative detector has binary32 magnitudes c~, p~ of the same exact band
// samples and a bound Delta on the l2 norm of (c~ - c), per band and frame:
//     Delta = K u theta sqrt(n) ||x|| + eabs        (n FFT points, x the band samples; K below)
// and reduces, per band and frame, the sums listed in C1DetSums.  From those of the frame and of its predecessor this
// header forms [lo, hi] with  lo <= score_reference <= hi  whenever `ok`; the caller takes the decision
// `score > threshold` only when the whole interval lies on one side and sends every other unit to the exact kernels.
// Every step is monotone interval arithmetic; the inequalities used are quoted where they are applied.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define C1_HD __host__ __device__ inline
#else
#define C1_HD static inline
#endif

// K: rounding errors of both FFTs in units of u ||spectrum|| (u = 2^-24), stage by stage; every radix-2 stage doubles
// the squared norm, so an error of relative size c u injected after stage s is still c u ||spectrum|| at the end.
//   reference: one Float32 store per stage                                   1 per stage
//   ours: stages 1, 2 (real adds; -i rotations are exact)                    1 per stage
//         a stage with products: product (2 roundings, fused second), rounded table value: |dt| <= 3 u |o| on the
//         odd half (norm <= ||stage output|| / sqrt 2, reaching both outputs: x sqrt 2), final add 1:     4 per stage
//   magnitudes: reference 1 (store); ours 3.1 (square, fused sum, 1-ulp square root)
// 128 points (7 stages): 7 + 2 + 5 x 4 + 4.1 = 33.1;  256 points (8 stages): 8 + 2 + 6 x 4 + 4.1 = 38.1
#define C1_DET_K128 33.1
#define C1_DET_K256 38.1
#define C1_DET_THETA 1.01            /* second-order terms, (1+u)^k, twiddles within 1e-12 of roots of unity, rounded norms */
#define C1_DET_EABS 8.673617379884035e-19   /* 2^-60: flushed subnormal squares under the square root (<= 2^-63) and below */
#define C1_DET_ROW_FLOATS 10
#define C1_DET_ROWS 4                /* band 0 | band 1 | band 2 lanes 32..47 | band 2 lanes 48..63 */

// The interval itself is evaluated in binary32 with outward slack (one lane per sound unit evaluates six of these per
// call; in binary64 that was a third of the decision kernel).  Every binary32 operation used here -- add, multiply,
// divide and square root (correctly rounded), expf, log2f, log1pf (within 3 ulp = 3.6e-7 relative by the OpenCL bounds
// the device library is built to, 1 ulp in the host's libm) -- returns its result within 2^-21 relative; C1_UP / C1_DN
// move a freshly computed upper / lower endpoint outwards by 2^-20 relative and 1e-37 absolute (underflow), which
// covers that and their own two roundings.  All formulas are monotone in every endpoint they read, so lower endpoints
// stay below and upper endpoints above what exact arithmetic would give.
#define C1_UP(x) ((x) + fabsf(x) * 9.5367431640625e-07f + 1e-37f)
#define C1_DN(x) ((x) - fabsf(x) * 9.5367431640625e-07f - 1e-37f)
#define C1_FIN(x) ((x) < 3.0e38f && (x) > -3.0e38f)     /* finite and not NaN */

// sums of one band of one frame (all over the band's bins k; "valid" = magnitude certainly > 1e-10)
typedef struct {
  float flux;    // sum max(c~_k - p~_k, 0)
  float elo;     // sum c~_k^2, lower half of the bins
  float ehi;     // upper half
  float slog;    // sum_valid log2 c~_k
  float sabs;    // sum_valid |log2 c~_k|
  float slin;    // sum_valid c~_k
  float sinv2;   // sum_valid 1 / (c~_k - Delta)^2
  float nv;      // number of valid bins
  float bad;     // bins whose validity is not certain (neither c~_k - Delta > 1e-10 nor c~_k + Delta <= 1e-10), NaNs included
  float delta;   // Delta
} C1DetSums;

C1_HD C1DetSums c1_det_sums(const float *rec, int band) {
  C1DetSums s;
  const float *r = rec + C1_DET_ROW_FLOATS * (band == 2 ? 2 : band);
  s.flux = r[0]; s.elo = r[1]; s.ehi = r[2]; s.slog = r[3]; s.sabs = r[4];
  s.slin = r[5]; s.sinv2 = r[6]; s.nv = r[7]; s.bad = r[8]; s.delta = r[9];
  if (band == 2) {
    const float *q = r + C1_DET_ROW_FLOATS;
    s.flux += q[0]; s.elo += q[1]; s.ehi += q[2]; s.slog += q[3]; s.sabs += q[4];
    s.slin += q[5]; s.sinv2 += q[6]; s.nv += q[7]; s.bad += q[8];
    if (!(q[9] == r[9])) s.bad += 1.0f;      // both rows carry the band's Delta
  }
  return s;
}
C1_HD C1DetSums c1_det_zero_sums(void) {
  C1DetSums s;
  s.flux = s.elo = s.ehi = s.slog = s.sabs = s.slin = s.sinv2 = s.nv = s.bad = s.delta = 0.0f;
  return s;
}

// what a frame contributes on its own: intervals for flatness, high-frequency ratio, energy, and for the norm of c
typedef struct {
  float flat_lo, flat_hi, hf_lo, hf_hi, e_lo, e_hi, r_lo, r_hi;
  int zero;      // every band sample is +-0: the reference's magnitudes are exactly 0
  int ok;
} C1DetOwn;

#define C1_DET_SUM 1.9073486328125e-06f    /* 2^-19 = 32 u: a rounded term plus a binary32 sum of <= 128 terms along a path of <= 10 additions */

C1_HD C1DetOwn c1_det_own(const C1DetSums s) {
  C1DetOwn o;
  o.flat_lo = o.flat_hi = o.hf_lo = o.hf_hi = o.e_lo = o.e_hi = o.r_lo = o.r_hi = 0.0f;
  o.zero = 0;
  const float e = s.elo + s.ehi;
  o.ok = (s.delta >= 0.0f) && C1_FIN(s.delta) && (s.bad == 0.0f) && C1_FIN(e) && C1_FIN(s.slin) && C1_FIN(s.sinv2) &&
         C1_FIN(s.sabs) && C1_FIN(s.flux) && C1_FIN(s.slog);
  if (!o.ok) return o;
  if (s.delta == 0.0f && e == 0.0f) { o.zero = 1; return o; }        // flat = 0 (no valid bin), hf = 0 (total 0), energy 0
  // squares of magnitudes below 2^-63 lose bits to underflow in the binary32 sums (at most 128 x 2^-126 in all): with
  // e >= 2^-90 that is below 2^-29 e; quieter bands are left to the exact kernels
  if (!(e >= 8.077935669463161e-28f)) { o.ok = 0; return o; }
  const float rt = sqrtf(e);
  // | ||c~|| - ||c|| | <= ||c~ - c|| <= Delta
  o.r_lo = C1_DN(C1_DN(rt * (1.0f - C1_DET_SUM)) - s.delta);
  if (o.r_lo < 0.0f) o.r_lo = 0.0f;
  o.r_hi = C1_UP(C1_UP(rt * (1.0f + C1_DET_SUM)) + s.delta);
  o.e_lo = C1_DN(o.r_lo * o.r_lo);
  if (o.e_lo < 0.0f) o.e_lo = 0.0f;
  o.e_hi = C1_UP(o.r_hi * o.r_hi);
  // calculateSpectralFlatness (:120-141).  Which bins count is certain (bad == 0).
  if (s.nv > 0.0f) {
    // |ln c_k - ln c~_k| <= |c_k - c~_k| / min(c_k, c~_k) <= |d_k| / (c~_k - Delta); Cauchy-Schwarz over the bins;
    // + the device's log2 (|error| <= 2 u |log2 c| + 2^-22 for every binary32 c, checked exhaustively on the device:
    // tests/test_gpu_detect_spec.py) and the rounded sum of the terms (32 u sum |term|)
    const float ln2 = 0.6931472f;
    const float el = C1_UP(C1_UP(C1_UP(C1_UP(s.delta * sqrtf(s.sinv2)) * 1.001f) + 1.5e-06f * s.sabs) / s.nv) + 1.7e-07f;
    if (!(el < 0.5f)) { o.ok = 0; return o; }
    const float ml = ln2 * s.slog / s.nv;                // within |ml| 2^-21 of the exact mean; expf adds 3 ulp: (|ml| + 2) 2^-20 covers both
    const float gm = expf(ml), ge = (fabsf(ml) + 2.0f) * 9.5367431640625e-07f;
    const float gm_lo = C1_DN(gm * C1_DN(1.0f - el - ge));             // exp(-x) >= 1 - x
    const float gm_hi = C1_UP(gm * C1_UP(1.0f + el + el * el + ge));   // exp(x) <= 1 + x + x^2 on [0, 1]
    // |sum c~_k - sum c_k| <= sqrt(nv) Delta
    const float sq = C1_UP(sqrtf(s.nv) * s.delta);
    const float am_lo = C1_DN(C1_DN(C1_DN(s.slin * (1.0f - C1_DET_SUM)) - sq) / s.nv);
    const float am_hi = C1_UP(C1_UP(C1_UP(s.slin * (1.0f + C1_DET_SUM)) + sq) / s.nv);
    if (!(am_lo > 1.0001e-10f)) { o.ok = 0; return o; }                // `arithmeticMean > EPSILON ? ... : 0` must be certain
    o.flat_lo = C1_DN(gm_lo / am_hi);
    if (o.flat_lo < 0.0f) o.flat_lo = 0.0f;
    o.flat_hi = C1_UP(gm_hi / am_lo);
  }
  // calculateHighFrequencyRatio (:149-164): hf = h^2 / (l^2 + h^2) = sin^2 phi for the norms (l, h) of the two halves.
  // (l, h) lies within Delta of (l~, h~); the angle between the two vectors is at most asin(Delta / r~) <= (pi/2) Delta / r~,
  // and |d sin^2 phi / d phi| <= 1.
  if (!(o.r_lo > 0.0f)) { o.ok = 0; return o; }                        // `totalEnergy > 0`, `sqrt(energy) || 1e-6` must be certain
  {
    const float hf = s.ehi / e, eh = C1_UP(C1_UP(1.5708f * s.delta) / rt) + 4.0e-06f;   // + the rounded sums of the two halves (64 u)
    o.hf_lo = hf - eh < 0.0f ? 0.0f : C1_DN(hf - eh);
    o.hf_hi = hf + eh > 1.0f ? 1.0f : C1_UP(hf + eh);
  }
  return o;
}

// transient score of one band (calculateSpectralFeatures + calculateTransientScore, :63-226) as an interval.
// `prev` = the previous frame's sums and own features, or the zero state of a fresh BufferPool (have_prev == 0).
C1_HD int c1_det_score(const C1DetSums cur, const C1DetOwn oc, int have_prev, const C1DetSums prev, const C1DetOwn op,
                       int bins, double log1p10, double *lo, double *hi) {
  *lo = -1e300; *hi = 1e300;
  if (!oc.ok || (have_prev && !op.ok)) return 0;
  const float dp = have_prev ? prev.delta : 0.0f;
  // calculateSpectralFlux (:92-112): |max(a, 0) - max(b, 0)| <= |a - b|; sum_k |d_k| <= sqrt(bins) ||d||
  float fl_lo, fl_hi;
  if (oc.zero) fl_lo = fl_hi = 0.0f;                                 // sum max(0 - p_k, 0) = 0, divided by 1e-6
  else {
    const float slack = C1_UP((bins == 128 ? 11.31371f : 8.0f) * C1_UP(cur.delta + dp));
    float a = C1_DN(C1_DN(cur.flux * (1.0f - C1_DET_SUM)) - slack);
    if (a < 0.0f) a = 0.0f;
    const float b = C1_UP(C1_UP(cur.flux * (1.0f + C1_DET_SUM)) + slack);
    fl_lo = C1_DN(a / oc.r_hi);
    if (fl_lo < 0.0f) fl_lo = 0.0f;
    fl_hi = C1_UP(b / oc.r_lo);
  }
  const float pf_lo = have_prev ? op.flat_lo : 0.0f, pf_hi = have_prev ? op.flat_hi : 0.0f;
  const float ph_lo = have_prev ? op.hf_lo : 0.0f, ph_hi = have_prev ? op.hf_hi : 0.0f;
  const float pe_lo = have_prev ? op.e_lo : 0.0f, pe_hi = have_prev ? op.e_hi : 0.0f;
  // |a - b| over two intervals
  float fd_lo = oc.flat_lo - pf_hi > pf_lo - oc.flat_hi ? oc.flat_lo - pf_hi : pf_lo - oc.flat_hi;
  fd_lo = fd_lo <= 0.0f ? 0.0f : C1_DN(fd_lo);
  if (fd_lo < 0.0f) fd_lo = 0.0f;
  const float fd_hi = C1_UP(oc.flat_hi - pf_lo > pf_hi - oc.flat_lo ? oc.flat_hi - pf_lo : pf_hi - oc.flat_lo);
  float hd_lo = oc.hf_lo - ph_hi > ph_lo - oc.hf_hi ? oc.hf_lo - ph_hi : ph_lo - oc.hf_hi;
  hd_lo = hd_lo <= 0.0f ? 0.0f : C1_DN(hd_lo);
  if (hd_lo < 0.0f) hd_lo = 0.0f;
  const float hd_hi = C1_UP(oc.hf_hi - ph_lo > ph_hi - oc.hf_lo ? oc.hf_hi - ph_lo : ph_hi - oc.hf_lo);
  // calculateEnergyChange (:172-189): increasing in the current energy, decreasing in the previous one;
  // 10 log10 x = 3.0103 log2 x, log2f within 2^-22 (1 + |log2 x|)
  const float ce_lo = oc.e_lo > 0.9999999e-10f ? oc.e_lo : 0.9999999e-10f, ce_hi = oc.e_hi > 1.0000001e-10f ? oc.e_hi : 1.0000001e-10f;
  const float qe_lo = pe_lo > 0.9999999e-10f ? pe_lo : 0.9999999e-10f, qe_hi = pe_hi > 1.0000001e-10f ? pe_hi : 1.0000001e-10f;
  float db_lo = C1_DN(3.0102999f * log2f(C1_DN(ce_lo / qe_hi))) - 3.0e-06f, db_hi = C1_UP(3.0103002f * log2f(C1_UP(ce_hi / qe_lo))) + 3.0e-06f;
  if (!(db_lo > 0.0f)) db_lo = 0.0f;                                  // also -inf (a quotient that underflowed)
  if (db_hi < 0.0f) db_hi = 0.0f;
  float ec_lo = C1_DN(db_lo / 30.0f), ec_hi = C1_UP(db_hi / 30.0f);
  if (ec_lo > 1.0f) ec_lo = 1.0f;
  if (ec_lo < 0.0f) ec_lo = 0.0f;
  if (!(ec_hi < 1.0f)) ec_hi = 1.0f;                                  // also +inf
  const float l10 = (float)log1p10;
  float hc_lo = C1_DN(log1pf(C1_DN(hd_lo * 10.0f)) / C1_UP(l10)) - 1.0e-06f, hc_hi = C1_UP(log1pf(C1_UP(hd_hi * 10.0f)) / C1_DN(l10)) + 1.0e-06f;
  if (hc_lo < 0.0f) hc_lo = 0.0f;
  float sf_lo = C1_DN(sqrtf(fd_lo)), sf_hi = C1_UP(sqrtf(fd_hi));
  if (sf_lo < 0.0f) sf_lo = 0.0f;
  const float s_lo = C1_DN(C1_DN(C1_DN(fl_lo + sf_lo) + C1_DN(hc_lo + ec_lo)) * 0.25f) - 2.0e-06f;
  const float s_hi = C1_UP(C1_UP(C1_UP(fl_hi + sf_hi) + C1_UP(hc_hi + ec_hi)) * 0.25f) + 2.0e-06f;   // + the reference's own binary64 roundings
  *lo = s_lo;
  *hi = s_hi;
  return C1_FIN(s_lo) && C1_FIN(s_hi);                                // a NaN or an overflow anywhere: not certain
}
