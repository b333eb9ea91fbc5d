/* c1o_fdlibm.h -- TEST INFRASTRUCTURE (part of the CPU oracle, see atrac1_oracle.h).
 *
 * Math.log / Math.exp / Math.log1p / Math.log10 as the reference's JavaScript engine evaluates them.  The reference
 * calls them in the transient score (codec/analysis/transient.js:129, :137, :185, :211); V8 implements them in
 * src/base/ieee754.cc (a third-party dependency of the reference's runtime, absent from /root/reference) as ports of
 * the published fdlibm algorithms e_log.c, e_exp.c, s_log1p.c and e_log10.c.  This file restates those algorithms.
 * They are not correctly rounded, so a different libm (glibc, OCML) returns a neighbouring double on ~1 % (log) to
 * ~7 % (exp) of the arguments; pinned bit for bit against V8 7.8 (Node 12.22 in this image) on 4 x 10 032 arguments
 * by tests/golden/libm_v8_*.bin (generator: tests/golden/gen/gen_libm.mjs, test: tests/test_oracle_golden.py).
 * Compile with -ffp-contract=off: V8's x64 build evaluates every expression below without fusing.
 */
#ifndef C1O_FDLIBM_H
#define C1O_FDLIBM_H
#include <stdint.h>
#include <string.h>
static inline uint64_t c1o_fd_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double c1o_fd_from(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
static inline int32_t c1o_fd_hi(double x) { return (int32_t)(c1o_fd_bits(x) >> 32); }
static inline uint32_t c1o_fd_lo(double x) { return (uint32_t)c1o_fd_bits(x); }
static inline double c1o_fd_set_hi(double x, int32_t hi) { return c1o_fd_from(((uint64_t)(uint32_t)hi << 32) | c1o_fd_lo(x)); }

static const double c1o_fd_ln2_hi = 6.93147180369123816490e-01, c1o_fd_ln2_lo = 1.90821492927058770002e-10, c1o_fd_two54 = 1.80143985094819840000e+16;

static double c1o_fd_log(double x) {
  static const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                      Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                      Lg7 = 1.479819860511658591e-01;
  const double zero = 0.0;
  double hfsq, f, s, z, R, w, t1, t2, dk;
  int32_t k = 0, hx = c1o_fd_hi(x), i, j;
  uint32_t lx = c1o_fd_lo(x);
  if (hx < 0x00100000) {
    if (((hx & 0x7fffffff) | lx) == 0) return -c1o_fd_two54 / zero;
    if (hx < 0) return (x - x) / zero;
    k -= 54; x *= c1o_fd_two54; hx = c1o_fd_hi(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  i = (hx + 0x95f64) & 0x100000;
  x = c1o_fd_set_hi(x, hx | (i ^ 0x3ff00000));
  k += (i >> 20);
  f = x - 1.0;
  if ((0x000fffff & (2 + hx)) < 3) {
    if (f == zero) { if (k == 0) return zero; dk = (double)k; return dk * c1o_fd_ln2_hi + dk * c1o_fd_ln2_lo; }
    R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    dk = (double)k;
    return dk * c1o_fd_ln2_hi - ((R - dk * c1o_fd_ln2_lo) - f);
  }
  s = f / (2.0 + f);
  dk = (double)k;
  z = s * s;
  i = hx - 0x6147a;
  w = z * z;
  j = 0x6b851 - hx;
  t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  R = t2 + t1;
  if (i > 0) {
    hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * c1o_fd_ln2_hi - ((hfsq - (s * (hfsq + R) + dk * c1o_fd_ln2_lo)) - f);
  }
  if (k == 0) return f - s * (f - R);
  return dk * c1o_fd_ln2_hi - ((s * (f - R) - dk * c1o_fd_ln2_lo) - f);
}

static double c1o_fd_exp(double x) {
  static const double halF[2] = {0.5, -0.5}, o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
                      ln2HI[2] = {6.93147180369123816490e-01, -6.93147180369123816490e-01},
                      ln2LO[2] = {1.90821492927058770002e-10, -1.90821492927058770002e-10}, invln2 = 1.44269504088896338700e+00,
                      P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                      P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08, E = 2.718281828459045;
  const double one = 1.0, huge = 1.0e+300, twom1000 = 9.33263618503218878990e-302, two1023 = 8.988465674311579539e307;
  double y, hi = 0.0, lo = 0.0, c, t, twopk;
  int32_t k = 0, xsb;
  uint32_t hx = (uint32_t)c1o_fd_hi(x);
  xsb = (hx >> 31) & 1;
  hx &= 0x7fffffff;
  if (hx >= 0x40862E42) {
    if (hx >= 0x7ff00000) {
      uint32_t lx = c1o_fd_lo(x);
      if (((hx & 0xfffff) | lx) != 0) return x + x;
      return (xsb == 0) ? x : 0.0;
    }
    if (x > o_threshold) return huge * huge;
    if (x < u_threshold) return twom1000 * twom1000;
  }
  if (hx > 0x3fd62e42) {
    if (hx < 0x3FF0A2B2) {
      if (x == 1.0) return E;
      hi = x - ln2HI[xsb]; lo = ln2LO[xsb]; k = 1 - xsb - xsb;
    } else {
      k = (int32_t)(invln2 * x + halF[xsb]);
      t = k;
      hi = x - t * ln2HI[0];
      lo = t * ln2LO[0];
    }
    x = hi - lo;
  } else if (hx < 0x3e300000) {
    if (huge + x > one) return one + x;
  } else {
    k = 0;
  }
  t = x * x;
  if (k >= -1021) twopk = c1o_fd_from((uint64_t)(uint32_t)(0x3ff00000 + (k << 20)) << 32);
  else twopk = c1o_fd_from((uint64_t)(uint32_t)(0x3ff00000 + ((k + 1000) << 20)) << 32);
  c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return one - ((x * c) / (c - 2.0) - x);
  y = one - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k >= -1021) {
    if (k == 1024) return y * 2.0 * two1023;
    return y * twopk;
  }
  return y * twopk * twom1000;
}

static double c1o_fd_log1p(double x) {
  static const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                      Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                      Lp7 = 1.479819860511658591e-01;
  const double zero = 0.0;
  double hfsq, f = 0, c = 0, s, z, R, u;
  int32_t k, hx, hu = 0, ax;
  hx = c1o_fd_hi(x);
  ax = hx & 0x7fffffff;
  k = 1;
  if (hx < 0x3FDA827A) {
    if (ax >= 0x3ff00000) {
      if (x == -1.0) return -c1o_fd_two54 / zero;
      return (x - x) / (x - x);
    }
    if (ax < 0x3e200000) {
      if (c1o_fd_two54 + x > zero && ax < 0x3c900000) return x;
      return x - x * x * 0.5;
    }
    if (hx > 0 || hx <= ((int32_t)0xbfd2bec4)) { k = 0; f = x; hu = 1; }
  }
  if (hx >= 0x7ff00000) return x + x;
  if (k != 0) {
    if (hx < 0x43400000) {
      u = 1.0 + x;
      hu = c1o_fd_hi(u);
      k = (hu >> 20) - 1023;
      c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
      c /= u;
    } else {
      u = x;
      hu = c1o_fd_hi(u);
      k = (hu >> 20) - 1023;
      c = 0;
    }
    hu &= 0x000fffff;
    if (hu < 0x6a09e) {
      u = c1o_fd_set_hi(u, hu | 0x3ff00000);
    } else {
      k += 1;
      u = c1o_fd_set_hi(u, hu | 0x3fe00000);
      hu = (0x00100000 - hu) >> 2;
    }
    f = u - 1.0;
  }
  hfsq = 0.5 * f * f;
  if (hu == 0) {
    if (f == zero) {
      if (k == 0) return zero;
      c += k * c1o_fd_ln2_lo;
      return k * c1o_fd_ln2_hi + c;
    }
    R = hfsq * (1.0 - 0.66666666666666666 * f);
    if (k == 0) return f - R;
    return k * c1o_fd_ln2_hi - ((R - (k * c1o_fd_ln2_lo + c)) - f);
  }
  s = f / (2.0 + f);
  z = s * s;
  R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  return k * c1o_fd_ln2_hi - ((hfsq - (s * (hfsq + R) + (k * c1o_fd_ln2_lo + c))) - f);
}

static double c1o_fd_log10(double x) {
  static const double ivln10 = 4.34294481903251816668e-01, log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
  const double zero = 0.0;
  double y, z;
  int32_t i, k = 0, hx = c1o_fd_hi(x);
  uint32_t lx = c1o_fd_lo(x);
  if (hx < 0x00100000) {
    if (((hx & 0x7fffffff) | lx) == 0) return -c1o_fd_two54 / zero;
    if (hx < 0) return (x - x) / zero;
    k -= 54; x *= c1o_fd_two54; hx = c1o_fd_hi(x); lx = c1o_fd_lo(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  if (hx == 0x3ff00000 && lx == 0) return zero;
  k += (hx >> 20) - 1023;
  i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
  hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
  y = (double)(k + i);
  x = c1o_fd_from(((uint64_t)(uint32_t)hx << 32) | lx);
  z = y * log10_2lo + ivln10 * c1o_fd_log(x);
  return z + y * log10_2hi;
}

#endif
