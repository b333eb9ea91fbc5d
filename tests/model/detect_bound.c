/* tests/model/detect_bound.c -- CPU harness around carta1_amd/csrc/c1_detect_bound.h, the header the device's
 * k_detect_decide<SPEC> evaluates: the ten sums per 16-lane row as k_detect_features<SPEC> (spec_sums) reduces them, from
 * given binary32 magnitudes, and the interval for the reference's transient score.  Test infrastructure
 * (tests/test_detect_bound_cpu.py); built on demand with gcc. */
#include <math.h>
#include <string.h>

#include "../../carta1_amd/csrc/c1_detect_bound.h"

/* bins of row r: band 0 (64 bins) | band 1 (64) | band 2 lanes 32..47: g + 32 i, g < 16 | band 2 lanes 48..63: g >= 16 */
static int row_bins(int row, int idx[64]) {
  int n = 0;
  if (row < 2) { for (int k = 0; k < 64; k++) idx[n++] = 64 * row + k; return n; }
  for (int g = 16 * (row - 2); g < 16 * (row - 1); g++)
    for (int i = 0; i < 4; i++) idx[n++] = 128 + g + 32 * i;
  return n;
}

/* mags, pmags: 256 binary32 magnitudes (64 | 64 | 128); delta[3]: the bands' bounds; rec: 40 floats.  Sums are taken
 * in binary64 and rounded once: inside the error model for any summation order. */
void detm_record(const float *mags, const float *pmags, const float *delta, float *rec) {
  for (int row = 0; row < 4; row++) {
    const int band = row < 2 ? row : 2, base = band == 0 ? 0 : (band == 1 ? 64 : 128), bins = band == 2 ? 128 : 64;
    const float d = delta[band];
    const float t_valid = fmaf(d, 1.000001f, 1.0001e-10f), t_not = fmaf(d, -1.000001f, 0.9999e-10f);
    int idx[64];
    const int n = row_bins(row, idx);
    double flux = 0, elo = 0, ehi = 0, slog = 0, sabs = 0, slin = 0, sinv2 = 0, nv = 0, bad = 0;
    for (int k = 0; k < n; k++) {
      const float cm = mags[idx[k]], pm = pmags[idx[k]];
      const float df = cm - pm;
      flux += df > 0 ? df : 0;
      if (idx[k] - base < bins / 2) elo += (double)(cm * cm); else ehi += (double)(cm * cm);
      const int valid = cm > t_valid, sure = valid || cm < t_not;
      if (valid) {
        const float lg = log2f(cm), inv = 1.0f / (cm - d);
        slog += lg; sabs += fabsf(lg); slin += cm; sinv2 += (double)(inv * inv); nv += 1;
      }
      if (!sure) bad += 1;
    }
    float *r = rec + C1_DET_ROW_FLOATS * row;
    r[0] = (float)flux; r[1] = (float)elo; r[2] = (float)ehi; r[3] = (float)slog; r[4] = (float)sabs;
    r[5] = (float)slin; r[6] = (float)sinv2; r[7] = (float)nv; r[8] = (float)bad; r[9] = d;
  }
}

/* prev may be null: the zero state of a fresh BufferPool */
int detm_interval(const float *cur, const float *prev, int band, double log1p10, double *lo, double *hi) {
  const C1DetSums sc = c1_det_sums(cur, band);
  const C1DetSums sp = prev ? c1_det_sums(prev, band) : c1_det_zero_sums();
  const C1DetOwn oc = c1_det_own(sc);
  const C1DetOwn op = prev ? c1_det_own(sp) : oc;
  return c1_det_score(sc, oc, prev != 0, sp, op, band == 2 ? 128 : 64, log1p10, lo, hi);
}

double detm_constant(int which) {
  return which == 0 ? C1_DET_K128 : which == 1 ? C1_DET_K256 : which == 2 ? C1_DET_THETA : C1_DET_EABS;
}
