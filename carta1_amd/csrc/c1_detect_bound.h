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

// sums of one band of one frame (all over the band's bins k; "valid" = magnitude certainly > 1e-10)
typedef struct {
  double flux;    // sum max(c~_k - p~_k, 0)
  double elo;     // sum c~_k^2, lower half of the bins
  double ehi;     // upper half
  double slog;    // sum_valid log2 c~_k
  double sabs;    // sum_valid |log2 c~_k|
  double slin;    // sum_valid c~_k
  double sinv2;   // sum_valid 1 / (c~_k - Delta)^2
  double nv;      // number of valid bins
  double bad;     // bins whose validity is not certain (neither c~_k - Delta > 1e-10 nor c~_k + Delta <= 1e-10), NaNs included
  double delta;   // Delta
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
    if (!(q[9] == r[9])) s.bad += 1.0;      // both rows carry the band's Delta
  }
  return s;
}
C1_HD C1DetSums c1_det_zero_sums(void) {
  C1DetSums s;
  s.flux = s.elo = s.ehi = s.slog = s.sabs = s.slin = s.sinv2 = s.nv = s.bad = s.delta = 0.0;
  return s;
}

// what a frame contributes on its own: intervals for flatness, high-frequency ratio, energy, and for the norm of c
typedef struct {
  double flat_lo, flat_hi, hf_lo, hf_hi, e_lo, e_hi, r_lo, r_hi;
  int zero;      // every band sample is +-0: the reference's magnitudes are exactly 0
  int ok;
} C1DetOwn;

#define C1_DET_SUM 9.5367431640625e-07     /* 2^-20 = 16 u: a rounded term plus a binary32 sum of <= 128 terms along a path of <= 9 additions */
#define C1_DET_TINY 1e-12                 /* the reference's own binary64 roundings and libm (<= a few 2^-53 each) */

C1_HD C1DetOwn c1_det_own(const C1DetSums s) {
  C1DetOwn o;
  o.flat_lo = o.flat_hi = o.hf_lo = o.hf_hi = o.e_lo = o.e_hi = o.r_lo = o.r_hi = 0.0;
  o.zero = 0;
  o.ok = (s.delta >= 0.0) && (s.delta < 1e300) && (s.bad == 0.0) && (s.elo + s.ehi < 1e300) && (s.slin < 1e300) &&
         (s.sinv2 < 1e300) && (s.sabs < 1e300) && (s.flux < 1e300);
  if (!o.ok) return o;
  const double e = s.elo + s.ehi, rt = sqrt(e);
  if (s.delta == 0.0 && e == 0.0) { o.zero = 1; return o; }          // flat = 0 (no valid bin), hf = 0 (total 0), energy 0
  // squares of magnitudes below 2^-63 lose bits to underflow in the binary32 sums (at most 128 x 2^-126 in all): with
  // e >= 2^-90 that is below 2^-29 e; quieter bands are left to the exact kernels
  if (!(e >= 8.077935669463161e-28)) { o.ok = 0; return o; }
  // | ||c~|| - ||c|| | <= ||c~ - c|| <= Delta
  o.r_lo = rt * (1.0 - C1_DET_SUM) - s.delta;
  if (o.r_lo < 0.0) o.r_lo = 0.0;
  o.r_hi = rt * (1.0 + C1_DET_SUM) + s.delta;
  o.e_lo = o.r_lo * o.r_lo * (1.0 - C1_DET_TINY);
  o.e_hi = o.r_hi * o.r_hi * (1.0 + C1_DET_TINY);
  // calculateSpectralFlatness (:120-141).  Which bins count is certain (bad == 0).
  if (s.nv > 0.0) {
    // |ln c_k - ln c~_k| <= |c_k - c~_k| / min(c_k, c~_k) <= |d_k| / (c~_k - Delta); Cauchy-Schwarz over the bins
    const double ln2 = 0.6931471805599453;
    // + the device's log2 (|error| <= 2 u |log2 c| + 2^-22 for every binary32 c, checked exhaustively on the device:
    // tests/test_gpu_detect_spec.py) and the rounded sum of the terms (16 u sum |term|)
    const double el = (s.delta * sqrt(s.sinv2) * 1.001 + ln2 * 20.0 * 5.9604644775390625e-08 * s.sabs) / s.nv + ln2 * 2.384185791015625e-07;
    const double ml = ln2 * s.slog / s.nv;
    const double gm_lo = exp(ml - el) * (1.0 - C1_DET_TINY), gm_hi = exp(ml + el) * (1.0 + C1_DET_TINY);
    // |sum c~_k - sum c_k| <= sqrt(nv) Delta
    const double am_lo = (s.slin * (1.0 - C1_DET_SUM) - sqrt(s.nv) * s.delta) / s.nv;
    const double am_hi = (s.slin * (1.0 + C1_DET_SUM) + sqrt(s.nv) * s.delta) / s.nv;
    if (!(am_lo > 1.0001e-10)) { o.ok = 0; return o; }               // `arithmeticMean > EPSILON ? ... : 0` must be certain
    o.flat_lo = gm_lo / am_hi * (1.0 - C1_DET_TINY);
    o.flat_hi = gm_hi / am_lo * (1.0 + C1_DET_TINY);
  }
  // calculateHighFrequencyRatio (:149-164): hf = h^2 / (l^2 + h^2) = sin^2 phi for the norms (l, h) of the two halves.
  // (l, h) lies within Delta of (l~, h~); the angle between the two vectors is at most asin(Delta / r~) <= (pi/2) Delta / r~,
  // and |d sin^2 phi / d phi| <= 1.
  if (!(o.r_lo > 0.0)) { o.ok = 0; return o; }                       // `totalEnergy > 0`, `sqrt(energy) || 1e-6` must be certain
  {
    const double hf = s.ehi / e, eh = 1.5708 * s.delta / rt * 1.00001 + 40.0 * 5.9604644775390625e-08;
    o.hf_lo = hf - eh < 0.0 ? 0.0 : hf - eh;
    o.hf_hi = hf + eh > 1.0 ? 1.0 : hf + eh;
  }
  return o;
}

// transient score of one band (calculateSpectralFeatures + calculateTransientScore, :63-226) as an interval.
// `prev` = the previous frame's sums and own features, or the zero state of a fresh BufferPool (have_prev == 0).
C1_HD int c1_det_score(const C1DetSums cur, const C1DetOwn oc, int have_prev, const C1DetSums prev, const C1DetOwn op,
                       int bins, double log1p10, double *lo, double *hi) {
  *lo = -1e300; *hi = 1e300;
  if (!oc.ok || (have_prev && !op.ok)) return 0;
  const double dp = have_prev ? prev.delta : 0.0;
  // calculateSpectralFlux (:92-112): |max(a, 0) - max(b, 0)| <= |a - b|; sum_k |d_k| <= sqrt(bins) ||d||
  double fl_lo, fl_hi;
  if (oc.zero) fl_lo = fl_hi = 0.0;                                  // sum max(0 - p_k, 0) = 0, divided by 1e-6
  else {
    const double slack = sqrt((double)bins) * (cur.delta + dp);
    double a = cur.flux * (1.0 - C1_DET_SUM) - slack;
    if (a < 0.0) a = 0.0;
    const double b = cur.flux * (1.0 + C1_DET_SUM) + slack;
    fl_lo = a / oc.r_hi * (1.0 - C1_DET_TINY);
    fl_hi = b / oc.r_lo * (1.0 + C1_DET_TINY);
  }
  const double pf_lo = have_prev ? op.flat_lo : 0.0, pf_hi = have_prev ? op.flat_hi : 0.0;
  const double ph_lo = have_prev ? op.hf_lo : 0.0, ph_hi = have_prev ? op.hf_hi : 0.0;
  const double pe_lo = have_prev ? op.e_lo : 0.0, pe_hi = have_prev ? op.e_hi : 0.0;
  // |a - b| over two intervals
  double fd_lo = oc.flat_lo - pf_hi > pf_lo - oc.flat_hi ? oc.flat_lo - pf_hi : pf_lo - oc.flat_hi;
  if (fd_lo < 0.0) fd_lo = 0.0;
  const double fd_hi = oc.flat_hi - pf_lo > pf_hi - oc.flat_lo ? oc.flat_hi - pf_lo : pf_hi - oc.flat_lo;
  double hd_lo = oc.hf_lo - ph_hi > ph_lo - oc.hf_hi ? oc.hf_lo - ph_hi : ph_lo - oc.hf_hi;
  if (hd_lo < 0.0) hd_lo = 0.0;
  const double hd_hi = oc.hf_hi - ph_lo > ph_hi - oc.hf_lo ? oc.hf_hi - ph_lo : ph_hi - oc.hf_lo;
  // calculateEnergyChange (:172-189): increasing in the current energy, decreasing in the previous one
  const double ce_lo = oc.e_lo > 1e-10 ? oc.e_lo : 1e-10, ce_hi = oc.e_hi > 1e-10 ? oc.e_hi : 1e-10;
  const double qe_lo = pe_lo > 1e-10 ? pe_lo : 1e-10, qe_hi = pe_hi > 1e-10 ? pe_hi : 1e-10;
  double db_lo = 10.0 * log10(ce_lo / qe_hi), db_hi = 10.0 * log10(ce_hi / qe_lo);
  if (db_lo < 0.0) db_lo = 0.0;
  if (db_hi < 0.0) db_hi = 0.0;
  const double ec_lo = db_lo / 30.0 < 1.0 ? db_lo / 30.0 : 1.0, ec_hi = db_hi / 30.0 < 1.0 ? db_hi / 30.0 : 1.0;
  const double s_lo = (fl_lo + sqrt(fd_lo) + log1p(hd_lo * 10.0) / log1p10 + ec_lo) / 4.0;
  const double s_hi = (fl_hi + sqrt(fd_hi) + log1p(hd_hi * 10.0) / log1p10 + ec_hi) / 4.0;
  *lo = s_lo * (1.0 - 4.0 * C1_DET_TINY) - 4.0 * C1_DET_TINY;
  *hi = s_hi * (1.0 + 4.0 * C1_DET_TINY) + 4.0 * C1_DET_TINY;
  return (*lo == *lo) && (*hi == *hi);                               // a NaN anywhere: not certain
}
