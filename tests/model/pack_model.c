/*
 * pack_model.c -- CPU model of the decisions the product's speculative path takes on binary32 coefficients:
 *   (1) the scale-factor guard at the end of k_analysis_spec (carta1_amd/csrc/c1_k_spec.hip: "the bound, coefficients
 *       out, scale-factor indices with their guard"), and
 *   (2) the quantizer with its guard band in k_pack<.., SPEC> (carta1_amd/csrc/c1_k_pack.hip).
 * Restated operation for operation in binary32 (fmaf = v_fma_f32, one rounding per other operation; build with
 * -ffp-contract=off), so that the guards can be attacked without a GPU: tests/test_pack_guard_cpu.py moves coefficients
 * by up to the bound in the directions that flip a truncation or a scale-factor index and checks that every unit whose
 * bytes would change is flagged.  TEST INFRASTRUCTURE: the model of the builder's own kernels, not of the reference
 * (that is oracle/atrac1_oracle.c: quantization.js:34-56, bitallocation.js:290-299).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static const unsigned char SPECS[52] = {8, 8, 8, 8, 4, 4, 4, 4, 8, 8, 8, 8, 6, 6, 6, 6, 6, 6, 6, 6,
                                        6, 6, 6, 6, 7, 7, 7, 7, 9, 9, 9, 9, 10, 10, 10, 10,
                                        12, 12, 12, 12, 12, 12, 12, 12, 20, 20, 20, 20, 20, 20, 20, 20};

static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* scale_factor_index_fast (c1_device.h): 3 (e + 21) + [frac > 0] + [frac > m1] + [frac > m2], clamped */
static int sf_index(float maxabs, uint32_t m1, uint32_t m2) {
  const uint32_t u = f2u(maxabs);
  const int e = (int)(u >> 23) - 127;
  const uint32_t frac = u & 0x7fffffu;
  int r = 3 * (e + 21) + (frac > 0u) + (frac > m1) + (frac > m2);
  if (r > 63) r = 63;
  return e < -21 ? 0 : r;
}

/* The scale-factor guard: coefficients in BFU-major slot order (BFU b = slots first[b] .. first[b] + SPECS[b]),
 * eps[3] the per-band bounds.  Writes the accepted index (that of the low end, as the kernel stores it) and returns 1
 * when any BFU's index is not certain within the bound.
 * open_out (optional): per BFU, 1 when that BFU's index is open. */
int pack_model_sf(const float *slots, const float eps[3], uint32_t m1, uint32_t m2, int sfi_out[52], int *open_out) {
  int unstable = 0, first = 0;
  for (int b = 0; b < 52; b++) {
    float mx = 0.0f;
    for (int j = 0; j < SPECS[b]; j++) mx = fmaxf(mx, fabsf(slots[first + j]));
    first += SPECS[b];
    const float e = b >= 36 ? eps[2] : (b >= 20 ? eps[1] : eps[0]);
    const float lo = fmaxf((mx - e) * 0.99999976f, 0.0f), hi = (mx + e) * 1.00000024f;
    const int s_lo = sf_index(lo, m1, m2), s_hi = sf_index(hi, m1, m2);
    sfi_out[b] = s_lo;
    const int open = !(s_lo == s_hi && e < INFINITY);
    if (open_out) open_out[b] = open;
    if (open) unstable = 1;
  }
  return unstable;
}

/* v_cvt_i32_f32: truncation towards zero, saturating; NaN -> 0 */
static int32_t cvt_i32(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.0f) return INT32_MAX;
  if (x <= -2147483648.0f) return INT32_MIN;
  return (int32_t)x;
}

/* The quantizer of k_pack<.., SPEC>.  slots: coefficients in BFU-major slot order; sfi/wl: the unit's allocation
 * (wl index 0..15, BFUs >= nbfu coded with 0 bits); norm32[64 * 16] = fl32(quantRange(wl) / SCALE_FACTORS[sfi]).
 * q_out: mantissas in slot order.  Returns 1 when any mantissa is doubtful (the unit goes to the redo list).
 * worst_out (optional): the largest |fract(a) - 1/2| + et seen.  doubt_out (optional): per slot, 1 when that mantissa is
 * doubtful by itself (the kernel only keeps the unit's maximum). */
int pack_model_quantize(const float *slots, const float eps[3], const int sfi[52], const int wl[52], int nbfu,
                        const float *norm32, int q_out[512], float *worst_out, int *doubt_out) {
  uint32_t worst = 0u;
  int first = 0;
  for (int b = 0; b < 52; b++) {
    const int w = b < nbfu ? wl[b] : 0;
    const int bits = w == 0 ? 0 : w + 1;
    const float eb = b >= 36 ? eps[2] : (b >= 20 ? eps[1] : eps[0]);
    const float nf = (sfi[b] != 0 && bits != 0) ? norm32[sfi[b] * 16 + w] : 0.0f;
    const float g = eb * nf;
    const float guard = fmaf(g, 9.5367431640625e-07f, g) + 2.384185791015625e-07f;
    const int range = (1 << (bits > 0 ? bits - 1 : 0)) - 1;
    for (int j = 0; j < SPECS[b]; j++) {
      const float x = slots[first + j];
      const float a = fmaf(fabsf(x), nf, 0.5f);
      float d = a - floorf(a);                                 /* v_fract_f32 */
      if (d >= 1.0f) d = 0.99999994f;
      const float et = fmaf(a, 2.384185791015625e-07f, guard);
      const float t = fabsf(d - 0.5f) + et;
      const uint32_t tu = f2u(t);
      if (tu > worst) worst = tu;
      if (doubt_out) doubt_out[first + j] = !(tu < 0x3EFFFFFCu);
      /* the clamp before the conversion: trunc(min(a, range + 1/2)) = min(trunc(a), range) for every finite a >= 1/2;
       * v_min_f32 returns the other operand for a NaN */
      const float rh = (float)range + 0.5f;
      const float ac = (a != a) ? rh : fminf(a, rh);
      q_out[first + j] = cvt_i32(copysignf(ac, x));
    }
    first += SPECS[b];
  }
  if (worst_out) memcpy(worst_out, &worst, 4);
  return !(worst < 0x3EFFFFFCu);
}
