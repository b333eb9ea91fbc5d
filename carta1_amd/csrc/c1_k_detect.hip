// c1_k_detect.hip -- transient detection pipeline: features (runs), decisions (per unit), MDCT from the stored bands (per unit)
#include "c1_device.h"
#include "c1_detect_bound.h"

namespace {

// =====================================================================================================
// transient detection as its own pipeline: blockSelectorStage (encoder.js:111-152, analysis/transient.js)
// =====================================================================================================
//   k_detect_features  one wave per run of frames of one channel (QMF state and the previous frame's magnitudes
//                      are sequential): QMF analysis, the 128|128|256-point transient FFT, per-bin feature terms and
//                      the reference's 18 sequential sums.  Writes the raw band samples (2 KB) and the sums (160 B)
//                      of every frame to the workspace.
//   k_detect_decide    one lane per sound unit: the scalar feature arithmetic (exp, log10, log1p, sqrt ...; a few
//                      hundred fp64 instructions that kept 3 of 64 lanes busy inside the frame loop) -> block modes.
//   k_mdct_bands       one wave per unit, units independent: windowing + MDCT + scale factors from the stored band
//                      samples of the frame and the 32-sample tails of the previous one.
// Workspace slots are indexed (frame + 1) * channels + channel: slot row 0 is frame -1 (the PCM halo), which the
// decision and the overlap of frame 0 need; before the stream start everything is the zero state (buffers.js:30-59).
constexpr int kFeatureDoubles = kFeatureWsDoubles;   // 18 sums: band-major x {flux, energy, log, linear, low, high}; then nv[3] as int32

struct alignas(16) DetectLds {
  double d1[46];
  double d2[46];
  alignas(16) float hbuf[296];
  alignas(16) float band[512];
  union alignas(16) {
    struct { alignas(16) double w1[698]; } q1;
    struct { alignas(16) double w2[454]; } q2;
    struct { alignas(16) float2 z[576]; } t;      // transient FFT points, 1 pad slot per 8
    struct { alignas(16) double term[4][256]; } tt;
  } u;
};
// the speculative detector reduces its sums in registers: without the per-bin terms a wave needs 9.4 KB, 4 waves per SIMD
struct alignas(16) DetectSpecLds {
  double d1[46];
  double d2[46];
  alignas(16) float hbuf[296];
  alignas(16) float band[512];
  union alignas(16) {
    struct { alignas(16) double w1[698]; } q1;
    struct { alignas(16) double w2[454]; } q2;
    struct { alignas(16) float2 z[576]; } t;
  } u;
};
static_assert(sizeof(DetectSpecLds) <= 10240, "speculative detector: 16 waves per CU");

__device__ __forceinline__ int tslot(int pos) { return pos + (pos >> 3); }

// =====================================================================================================
// Math.log / exp / log1p / log10 as the reference's engine evaluates them (transient.js:129, :137, :185, :211).
// V8 (src/base/ieee754.cc) ports the published fdlibm algorithms; they are not correctly rounded, so another libm
// returns a neighbouring double for 1-7 % of the arguments.  These restate the same algorithms operation for
// operation (contraction is off in this file); tests/test_gpu_parity.py checks them bit for bit against V8's results
// (tests/golden/libm_v8_*.bin) through c1_libm_device.
// =====================================================================================================
__device__ __forceinline__ double js_with_hi(double x, int hi) { return __hiloint2double(hi, __double2loint(x)); }
constexpr double kLn2Hi = 6.93147180369123816490e-01, kLn2Lo = 1.90821492927058770002e-10, kTwo54 = 1.80143985094819840000e+16;

// e_log.c.  The four return expressions of the main path are one: with dk = 0 the k != 0 forms reduce to the k == 0 ones
// exactly (0 - (a - f) == f - a, P + 0 == P), and the two mantissa ranges differ in two operands.
__device__ __forceinline__ double js_log(double x) {
  constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                   Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                   Lg7 = 1.479819860511658591e-01;
  int hx = __double2hiint(x), k = 0;
  if (hx < 0x00100000) {
    if (((hx & 0x7fffffff) | __double2loint(x)) == 0) return -__builtin_huge_val();
    if (hx < 0) return __builtin_nan("");
    k = -54; x *= kTwo54; hx = __double2hiint(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  const int i = (hx + 0x95f64) & 0x100000;
  x = js_with_hi(x, hx | (i ^ 0x3ff00000));
  k += (i >> 20);
  const double f = x - 1.0, dk = (double)k;
  const double hi = dk * kLn2Hi, lo = dk * kLn2Lo;
  if ((0x000fffff & (2 + hx)) < 3) {
    if (f == 0.0) return hi + lo;
    const double R = f * f * (0.5 - 0.33333333333333333 * f);
    return hi - ((R - lo) - f);
  }
  const double s = f / (2.0 + f), z = s * s, w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const bool mid = ((hx - 0x6147a) | (0x6b851 - hx)) > 0;
  const double hfsq = 0.5 * f * f;
  const double P = s * ((mid ? hfsq : f) + (mid ? R : -R));
  const double Q = mid ? hfsq - (P + lo) : P - lo;
  return hi - (Q - f);
}

// e_exp.c
__device__ double js_exp(double x) {
  constexpr double o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02, invln2 = 1.44269504088896338700e+00,
                   P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                   P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08, E = 2.718281828459045,
                   huge = 1.0e+300, twom1000 = 9.33263618503218878990e-302, two1023 = 8.988465674311579539e307;
  double hi = 0.0, lo = 0.0;
  int k = 0;
  uint32_t hx = (uint32_t)__double2hiint(x);
  const int xsb = (int)(hx >> 31);
  hx &= 0x7fffffffu;
  if (hx >= 0x40862E42u) {
    if (hx >= 0x7ff00000u) {
      if (((hx & 0xfffffu) | (uint32_t)__double2loint(x)) != 0) return x + x;
      return xsb == 0 ? x : 0.0;
    }
    if (x > o_threshold) return huge * huge;
    if (x < u_threshold) return twom1000 * twom1000;
  }
  if (hx > 0x3fd62e42u) {
    if (hx < 0x3FF0A2B2u) {
      if (x == 1.0) return E;
      hi = x - (xsb ? -kLn2Hi : kLn2Hi); lo = xsb ? -kLn2Lo : kLn2Lo; k = 1 - xsb - xsb;
    } else {
      k = (int)(invln2 * x + (xsb ? -0.5 : 0.5));
      const double t = (double)k;
      hi = x - t * kLn2Hi;
      lo = t * kLn2Lo;
    }
    x = hi - lo;
  } else if (hx < 0x3e300000u) {
    if (huge + x > 1.0) return 1.0 + x;
  }
  const double t = x * x;
  const double twopk = __hiloint2double(0x3ff00000 + ((k >= -1021 ? k : k + 1000) << 20), 0);
  const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
  const double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k >= -1021) {
    if (k == 1024) return y * 2.0 * two1023;
    return y * twopk;
  }
  return y * twopk * twom1000;
}

// s_log1p.c
__device__ double js_log1p(double x) {
  constexpr double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                   Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                   Lp7 = 1.479819860511658591e-01;
  double f = 0.0, c = 0.0, u;
  int hu = 0, k = 1;
  const int hx = __double2hiint(x), ax = hx & 0x7fffffff;
  if (hx < 0x3FDA827A) {
    if (ax >= 0x3ff00000) {
      if (x == -1.0) return -__builtin_huge_val();
      return __builtin_nan("");
    }
    if (ax < 0x3e200000) {
      if (kTwo54 + x > 0.0 && ax < 0x3c900000) return x;
      return x - x * x * 0.5;
    }
    if (hx > 0 || hx <= (int)0xbfd2bec4) { k = 0; f = x; hu = 1; }
  }
  if (hx >= 0x7ff00000) return x + x;
  if (k != 0) {
    if (hx < 0x43400000) {
      u = 1.0 + x;
      hu = __double2hiint(u);
      k = (hu >> 20) - 1023;
      c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
      c /= u;
    } else {
      u = x;
      hu = __double2hiint(u);
      k = (hu >> 20) - 1023;
      c = 0.0;
    }
    hu &= 0x000fffff;
    if (hu < 0x6a09e) {
      u = js_with_hi(u, hu | 0x3ff00000);
    } else {
      k += 1;
      u = js_with_hi(u, hu | 0x3fe00000);
      hu = (0x00100000 - hu) >> 2;
    }
    f = u - 1.0;
  }
  const double hfsq = 0.5 * f * f, dk = (double)k;
  if (hu == 0) {
    if (f == 0.0) {
      if (k == 0) return 0.0;
      c += dk * kLn2Lo;
      return dk * kLn2Hi + c;
    }
    const double R = hfsq * (1.0 - 0.66666666666666666 * f);
    if (k == 0) return f - R;
    return dk * kLn2Hi - ((R - (dk * kLn2Lo + c)) - f);
  }
  const double s = f / (2.0 + f), z = s * s;
  const double R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  return dk * kLn2Hi - ((hfsq - (s * (hfsq + R) + (dk * kLn2Lo + c))) - f);
}

// e_log10.c as V8 carries it (log of the normalised argument, then the exponent in two pieces)
__device__ double js_log10(double x) {
  constexpr double ivln10 = 4.34294481903251816668e-01, log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
  int hx = __double2hiint(x), k = 0;
  uint32_t lx = (uint32_t)__double2loint(x);
  if (hx < 0x00100000) {
    if (((hx & 0x7fffffff) | lx) == 0) return -__builtin_huge_val();
    if (hx < 0) return __builtin_nan("");
    k = -54; x *= kTwo54; hx = __double2hiint(x); lx = (uint32_t)__double2loint(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  if (hx == 0x3ff00000 && lx == 0) return 0.0;
  k += (hx >> 20) - 1023;
  const int i = (int)(((uint32_t)k & 0x80000000u) >> 31);
  hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
  const double y = (double)(k + i);
  x = __hiloint2double(hx, (int)lx);
  const double z = y * log10_2lo + ivln10 * js_log(x);
  return z + y * log10_2hi;
}

__global__ void k_libm_tap(int fn, const double *__restrict__ in, double *__restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = in[i];
  out[i] = fn == 0 ? js_log(x) : (fn == 1 ? js_exp(x) : (fn == 2 ? js_log1p(x) : js_log10(x)));
}

// ---- lane-only geometry of the transient FFT: lanes 0..15 band 0 (128 points), 16..31 band 1, 32..63 band 2 (256);
// eight points per lane and round
struct TGeom {
  int band, g, S;          // S = N/8: sample stride of round A, point stride of round C
  int src;                 // band sample of the lane's first round-A input (bit-reversed group)
  int za, zb, zc, zc_stride;
  int twb, twc, twc_stride, twd;             // byte offsets into fft_tw (binary64 pairs)
  int twb32, twc32, twc32_stride, twd32;     // the same entries of tw32 (binary32 pairs)
  int mag;                 // magnitude index of the lane's first bin; the next bins are S further each
};
__device__ __forceinline__ TGeom tfft_geometry(int lane0) {
  TGeom G;
  G.band = lane0 < 16 ? 0 : (lane0 < 32 ? 1 : 2);
  G.g = lane0 - (G.band == 0 ? 0 : (G.band == 1 ? 16 : 32));
  G.S = G.band == 2 ? 32 : 16;
  G.src = (G.band == 0 ? 0 : (G.band == 1 ? 128 : 256)) + bitrev(G.g, G.band == 2 ? 5 : 4);
  const int pbase = G.band == 0 ? 0 : (G.band == 1 ? 128 : 256);
  G.za = tslot(pbase + 8 * G.g);
  G.zb = tslot(pbase + 64 * (G.g >> 3) + (G.g & 7));
  G.zc = tslot(pbase + G.g);
  G.zc_stride = G.S + G.S / 8;
  const int eb = 7 + (G.g & 7), ec = (G.band == 2 ? 127 : 63) + G.g, ed = 63 + (G.g & 31);
  G.twb = (int)offsetof(C1DevTables, fft_tw) + 16 * eb;
  G.twc = (int)offsetof(C1DevTables, fft_tw) + 16 * ec;
  G.twc_stride = 16 * G.S;
  G.twd = (int)offsetof(C1DevTables, fft_tw) + 16 * ed;
  G.twb32 = (int)offsetof(C1DevTables, tw32) + 8 * eb;
  G.twc32 = (int)offsetof(C1DevTables, tw32) + 8 * ec;
  G.twc32_stride = 8 * G.S;
  G.twd32 = (int)offsetof(C1DevTables, tw32) + 8 * ed;
  G.mag = (G.band == 0 ? 0 : (G.band == 1 ? 64 : 128)) + G.g;
  return G;
}

// Round A of the transient FFT (performFFT, transient.js:17-35): real input, stages h = 1, 2, 4 on the points at
// bit-reversed positions 8g..8g+7.  Seven of the twelve butterflies have the twiddle (1, 0); when every sample is
// finite, not -0 and small enough not to overflow they are exact as Float32 adds (see r2_unit_ok), and the
// imaginary parts they touch are +0 throughout.
__device__ __forceinline__ void tfft_round_a(float2 (&x)[8], TablesPtr T) {
  const double2 w0 = make_double2(T->fft_tw[0][0], T->fft_tw[0][1]), w1 = make_double2(T->fft_tw[1][0], T->fft_tw[1][1]);
  const double2 w2 = make_double2(T->fft_tw[2][0], T->fft_tw[2][1]), w3 = make_double2(T->fft_tw[3][0], T->fft_tw[3][1]);
  const double2 w4 = make_double2(T->fft_tw[4][0], T->fft_tw[4][1]), w5 = make_double2(T->fft_tw[5][0], T->fft_tw[5][1]);
  const double2 w6 = make_double2(T->fft_tw[6][0], T->fft_tw[6][1]);
  uint32_t big = 0;
  bool neg_zero = false;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t u = __float_as_uint(x[j].x);
    big = max(big, u & 0x7fffffffu);
    neg_zero |= (u == 0x80000000u);
  }
  const bool exact = big < 0x7b800000u && !neg_zero;      // |x| < 2^120 (also excludes inf and NaN)
  if (__all(exact)) {
    // stage 1: all unit; stage 2: (0,2) (4,6) unit; stage 3: (0,4) unit.  Real parts only where the imaginary is +0.
    float a0 = x[0].x + x[1].x, a1 = x[0].x - x[1].x, a2 = x[2].x + x[3].x, a3 = x[2].x - x[3].x;
    float a4 = x[4].x + x[5].x, a5 = x[4].x - x[5].x, a6 = x[6].x + x[7].x, a7 = x[6].x - x[7].x;
    x[0] = make_float2(a0 + a2, 0.0f); x[2] = make_float2(a0 - a2, 0.0f);
    x[4] = make_float2(a4 + a6, 0.0f); x[6] = make_float2(a4 - a6, 0.0f);
    x[1] = make_float2(a1, 0.0f); x[3] = make_float2(a3, 0.0f); x[5] = make_float2(a5, 0.0f); x[7] = make_float2(a7, 0.0f);
    r2_butterfly(x[1], x[3], w2); r2_butterfly(x[5], x[7], w2);
    const float b0 = x[0].x + x[4].x, b4 = x[0].x - x[4].x;
    x[0].x = b0; x[4].x = b4;
  } else {
    r2_butterfly(x[0], x[1], w0); r2_butterfly(x[2], x[3], w0); r2_butterfly(x[4], x[5], w0); r2_butterfly(x[6], x[7], w0);
    r2_butterfly(x[0], x[2], w1); r2_butterfly(x[1], x[3], w2); r2_butterfly(x[4], x[6], w1); r2_butterfly(x[5], x[7], w2);
    r2_butterfly(x[0], x[4], w3);
  }
  r2_butterfly(x[1], x[5], w4); r2_butterfly(x[2], x[6], w5); r2_butterfly(x[3], x[7], w6);
}
// e-output only of a butterfly: the last stage feeds the positive-frequency half (transient.js:29-32)
__device__ __forceinline__ float2 r2_butterfly_e(const float2 e, const float2 o, const double2 w) {
  const double er = e.x, ei = e.y, orr = o.x, oi = o.y;
  const double xr = orr * w.x - oi * w.y;
  const double xi = orr * w.y + oi * w.x;
  return make_float2(f32(er + xr), f32(ei + xi));
}

// performFFT (transient.js:17-35) of the three bands in `band`, exactly as the reference rounds it, in radix-8 rounds
// through `z`; mg = the Float32 magnitudes of the lane's four bins.  Ends with a fence: z may be reused.
__device__ __forceinline__ void tfft_exact(const float *band, float2 *z, const TGeom &G, TablesPtr T, TablesRsrc RT, float (&mg)[4]) {
  float2 x[8];
  {
    const float *src = band + G.src;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int jr = ((j & 1) << 2) | (j & 2) | (j >> 2);          // bitrev3
      x[j] = make_float2(src[jr * G.S], 0.0f);
    }
  }
  // twiddles of round B are requested before round A computes, those of round C before round B
  const double2 w8 = table_pair(RT, G.twb), w16a = table_pair(RT, G.twb + 128), w16b = table_pair(RT, G.twb + 256);
  const double2 w32a = table_pair(RT, G.twb + 384), w32b = table_pair(RT, G.twb + 512);
  const double2 w32c = table_pair(RT, G.twb + 640), w32d = table_pair(RT, G.twb + 768);
  __builtin_amdgcn_s_setprio(0);
  tfft_round_a(x, T);
  {
    float4 *dst = reinterpret_cast<float4 *>(z + G.za);
#pragma unroll
    for (int j = 0; j < 4; j++) dst[j] = make_float4(x[2 * j].x, x[2 * j].y, x[2 * j + 1].x, x[2 * j + 1].y);
  }
  wave_fence();
  {
    float2 *p = z + G.zb;                                    // stages 8, 16, 32 on the points p + 8j
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = p[9 * j];
    r2_butterfly(x[0], x[1], w8); r2_butterfly(x[2], x[3], w8); r2_butterfly(x[4], x[5], w8); r2_butterfly(x[6], x[7], w8);
    r2_butterfly(x[0], x[2], w16a); r2_butterfly(x[1], x[3], w16b); r2_butterfly(x[4], x[6], w16a); r2_butterfly(x[5], x[7], w16b);
    r2_butterfly(x[0], x[4], w32a); r2_butterfly(x[1], x[5], w32b); r2_butterfly(x[2], x[6], w32c); r2_butterfly(x[3], x[7], w32d);
#pragma unroll
    for (int j = 0; j < 8; j++) p[9 * j] = x[j];
  }
  const double2 wDa = table_pair(RT, G.twd), wDb = table_pair(RT, G.twd + 512);
  const double2 wC0 = table_pair(RT, G.twc), wC1 = table_pair(RT, G.twc + G.twc_stride);
  const double2 wC2 = table_pair(RT, G.twc + 2 * G.twc_stride), wC3 = table_pair(RT, G.twc + 3 * G.twc_stride);
  wave_fence();
  {
    // points g + S*t, t = 0..7.  Band 2 first runs stage 64 on them; then stage N/2 (64 for the 128-point
    // transforms, 128 for the 256-point one) pairs (t, t+4) and only its e-outputs, the bins g + S*t, are needed
    const float2 *p = z + G.zc;
#pragma unroll
    for (int t = 0; t < 8; t++) x[t] = p[t * G.zc_stride];
    if (G.band == 2) {
      r2_butterfly(x[0], x[2], wDa); r2_butterfly(x[1], x[3], wDb); r2_butterfly(x[4], x[6], wDa); r2_butterfly(x[5], x[7], wDb);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float2 e = r2_butterfly_e(x[i], x[i + 4], i == 0 ? wC0 : (i == 1 ? wC1 : (i == 2 ? wC2 : wC3)));
      const double r = e.x, im = e.y;
      mg[i] = f32(sqrt(r * r + im * im));
    }
  }
  wave_fence();                                           // the per-bin terms reuse the memory of the points
}

// feature terms per bin, then the reference's 18 sequential sums (transient.js:92-189) -> feat[0..18), nv[3] as int32 behind
__device__ __forceinline__ void exact_sums(double (*term)[256], const TGeom &G, int lane, const float (&mg)[4], const float (&pmag)[4],
                                           double *feat) {
  bool valid[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int g = G.mag + i * G.S;
    const double cm = (double)mg[i], pm = (double)pmag[i];
    const double diff = cm - pm;
    valid[i] = cm > 1e-10;
    term[0][g] = diff > 0 ? diff : 0.0;            // spectral flux terms (transient.js:96-106)
    term[1][g] = cm * cm;                          // energy terms (exact product)
    term[2][g] = valid[i] ? js_log(cm) : 0.0;      // flatness terms (transient.js:126-133)
    term[3][g] = valid[i] ? cm : 0.0;
  }
  int nv_all = 0;
  {
    uint64_t m = 0;
    int n0 = 0, n1 = 0, n2 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      m = __ballot(valid[i]);
      n0 += __popc((uint32_t)m & 0xffffu); n1 += __popc((uint32_t)m >> 16); n2 += __popcll(m >> 32);
    }
    nv_all = lane == 0 ? n0 : (lane == 1 ? n1 : n2);
  }
  wave_fence();
  if (lane < 18) {
    // 18 lanes each own one running sum (3 bands x {flux, energy, log, linear, low, high}), index ascending
    const int b = lane / 6, kind = lane - 6 * b;
    const int n = b == 2 ? 128 : 64, g0 = b == 0 ? 0 : (b == 1 ? 64 : 128);
    const int which = kind == 0 ? 0 : (kind == 2 ? 2 : (kind == 3 ? 3 : 1));
    const int start = g0 + (kind == 5 ? n / 2 : 0);
    const int len = kind >= 4 ? n / 2 : n;
    const double2 *arr = reinterpret_cast<const double2 *>(term[which] + start);
    double acc = 0.0;
#pragma unroll
    for (int blk = 0; blk < 4; blk++) {
      if (32 * blk < len) {
#pragma unroll
        for (int i = 0; i < 16; i++) { const double2 v = arr[16 * blk + i]; acc += v.x; acc += v.y; }
      }
    }
    feat[lane] = acc;
  }
  if (lane < 3) reinterpret_cast<int *>(feat + 18)[lane] = nv_all;
}

// ---------------------------------------------------------------------------------------------------------------------
// The speculative detector: the same transient FFT in binary32 on packed (re, im) pairs, any rounding
// order (c1_detect_bound.h counts its roundings), magnitudes, and per band the sums c1_detect_bound.h turns into an
// interval for the reference's score.  Row r of the wave (16 lanes) reduces its own ten values with DPP.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bf32(v2f &e, v2f &o, const v2f w) {
  const v2f t = cmul32(o, w);
  o = e - t;
  e = e + t;
}
__device__ __forceinline__ uint32_t row_allreduce_umax(uint32_t x) {
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, false));
  return x;
}
__device__ __forceinline__ void tfft_spec(const float *band, float2 *zf, const TGeom &G, TablesPtr T, TablesRsrc RT, float (&mg)[4],
                                          float &delta) {
  v2f *z = reinterpret_cast<v2f *>(zf);
  float xr[8];
  {
    const float *src = band + G.src;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int jr = ((j & 1) << 2) | (j & 2) | (j >> 2);          // bitrev3
      xr[j] = src[jr * G.S];
    }
  }
  const v2f w8 = table_f2(RT, G.twb32), w16a = table_f2(RT, G.twb32 + 64), w16b = table_f2(RT, G.twb32 + 128);
  const v2f w32a = table_f2(RT, G.twb32 + 192), w32b = table_f2(RT, G.twb32 + 256);
  const v2f w32c = table_f2(RT, G.twb32 + 320), w32d = table_f2(RT, G.twb32 + 384);
  // ---- Delta = K u theta sqrt(n) ||x|| + eabs from the band's own samples (c1_detect_bound.h) ----
  {
    float ss = 0.0f;
    uint32_t am = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { ss = __builtin_fmaf(xr[j], xr[j], ss); am = max(am, __float_as_uint(xr[j]) & 0x7fffffffu); }
    ss = row_allreduce(ss);
    am = row_allreduce_umax(am);                               // as a bit pattern: NaN and infinity come out on top
    const float ss_o = __shfl_xor(ss, 16);
    const uint32_t am_o = (uint32_t)__shfl_xor((int)am, 16);
    if (G.band == 2) { ss += ss_o; am = max(am, am_o); }
    const float amax = __uint_as_float(am);
    // squares of samples below 2^-63 underflow: below 2^-40 take sqrt(n) max|x| instead of the rounded sum
    const float w_up = amax < 9.094947e-13f ? 16.0f * amax : __builtin_sqrtf(ss) * 1.00001f;
    // three scalar reads and a select, not a lane-varying load (a cache round trip waited for on the spot); the asm keeps the
    // compiler from folding the select back into one
    float ck0 = T->det_ck[0], ck1 = T->det_ck[1], ck2 = T->det_ck[2];
    asm volatile("" : "+s"(ck0), "+s"(ck1), "+s"(ck2));
    const float ck = G.band == 0 ? ck0 : (G.band == 1 ? ck1 : ck2);
    delta = am == 0u ? 0.0f : __builtin_fmaf(ck, w_up, T->det_eabs) * 1.000001f;
  }
  // ---- round A: stages 1, 2, 4 of a real sequence; the rotations by -i are exact ----
  v2f c[8];
  {
    const float a0 = xr[0] + xr[1], a1 = xr[0] - xr[1], a2 = xr[2] + xr[3], a3 = xr[2] - xr[3];
    const float a4 = xr[4] + xr[5], a5 = xr[4] - xr[5], a6 = xr[6] + xr[7], a7 = xr[6] - xr[7];
    const float b0 = a0 + a2, b2 = a0 - a2, b4 = a4 + a6, b6 = a4 - a6;
    constexpr float s = 0.70710678118654752f;
    const float p = s * (a5 - a7), q = s * (a5 + a7);          // w4 b5 = (p, -q), w6 b7 = (-p, -q)
    c[0] = V2(b0 + b4, 0.0f); c[4] = V2(b0 - b4, 0.0f);
    c[2] = V2(b2, -b6); c[6] = V2(b2, b6);
    c[1] = V2(a1 + p, -a3 - q); c[5] = V2(a1 - p, q - a3);
    c[3] = V2(a1 - p, a3 - q); c[7] = V2(a1 + p, a3 + q);
  }
  {
    float4 *dst = reinterpret_cast<float4 *>(z + G.za);
#pragma unroll
    for (int j = 0; j < 4; j++) dst[j] = make_float4(c[2 * j].x, c[2 * j].y, c[2 * j + 1].x, c[2 * j + 1].y);
  }
  wave_fence();
  {
    v2f *p = z + G.zb;                                       // stages 8, 16, 32 on the points p + 8j
#pragma unroll
    for (int j = 0; j < 8; j++) c[j] = p[9 * j];
    bf32(c[0], c[1], w8); bf32(c[2], c[3], w8); bf32(c[4], c[5], w8); bf32(c[6], c[7], w8);
    bf32(c[0], c[2], w16a); bf32(c[1], c[3], w16b); bf32(c[4], c[6], w16a); bf32(c[5], c[7], w16b);
    bf32(c[0], c[4], w32a); bf32(c[1], c[5], w32b); bf32(c[2], c[6], w32c); bf32(c[3], c[7], w32d);
#pragma unroll
    for (int j = 0; j < 8; j++) p[9 * j] = c[j];
  }
  const v2f wDa = table_f2(RT, G.twd32), wDb = table_f2(RT, G.twd32 + 256);
  const v2f wC0 = table_f2(RT, G.twc32), wC1 = table_f2(RT, G.twc32 + G.twc32_stride);
  const v2f wC2 = table_f2(RT, G.twc32 + 2 * G.twc32_stride), wC3 = table_f2(RT, G.twc32 + 3 * G.twc32_stride);
  wave_fence();
  {
    const v2f *p = z + G.zc;
#pragma unroll
    for (int t = 0; t < 8; t++) c[t] = p[t * G.zc_stride];
    if (G.band == 2) { bf32(c[0], c[2], wDa); bf32(c[1], c[3], wDb); bf32(c[4], c[6], wDa); bf32(c[5], c[7], wDb); }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const v2f e = c[i] + cmul32(c[i + 4], i == 0 ? wC0 : (i == 1 ? wC1 : (i == 2 ? wC2 : wC3)));
      mg[i] = __builtin_sqrtf(__builtin_fmaf(e.y, e.y, e.x * e.x));
    }
  }
  wave_fence();
}

// the ten sums of c1_detect_bound.h for the lane's row; lane 16 r writes row r of the frame's record
__device__ __forceinline__ void spec_sums(const TGeom &G, int lane, const float (&mg)[4], const float (&pmag)[4], float delta, float *rec) {
  // certainly valid: c - Delta > 1e-10; certainly not: c + Delta <= 1e-10 (margins cover the roundings of these lines)
  const float t_valid = __builtin_fmaf(delta, 1.000001f, 1.0001e-10f);
  const float t_not = __builtin_fmaf(delta, -1.000001f, 0.9999e-10f);
  float flux = 0.0f, elo = 0.0f, ehi = 0.0f, slog = 0.0f, sabs = 0.0f, slin = 0.0f, sinv2 = 0.0f, nv = 0.0f, bad = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float cm = mg[i];
    flux += fmaxf(cm - pmag[i], 0.0f);
    if (i < 2) elo = __builtin_fmaf(cm, cm, elo); else ehi = __builtin_fmaf(cm, cm, ehi);
    const bool valid = cm > t_valid, sure = valid || cm < t_not;
    const float lg = __builtin_amdgcn_logf(cm);                 // v_log_f32: log2, 1 ulp (checked exhaustively, tests/test_gpu_detect_spec.py)
    const float inv = __builtin_amdgcn_rcpf(cm - delta);
    slog += valid ? lg : 0.0f;
    sabs += valid ? fabsf(lg) : 0.0f;
    slin += valid ? cm : 0.0f;
    sinv2 = valid ? __builtin_fmaf(inv, inv, sinv2) : sinv2;
    nv += valid ? 1.0f : 0.0f;
    bad += sure ? 0.0f : 1.0f;
  }
  flux = row_allreduce(flux); elo = row_allreduce(elo); ehi = row_allreduce(ehi);
  slog = row_allreduce(slog); sabs = row_allreduce(sabs); slin = row_allreduce(slin);
  sinv2 = row_allreduce(sinv2); nv = row_allreduce(nv); bad = row_allreduce(bad);
  if ((lane & 15) == 0) {
    float2 *dst = reinterpret_cast<float2 *>(rec + C1_DET_ROW_FLOATS * (lane >> 4));
    dst[0] = make_float2(flux, elo); dst[1] = make_float2(ehi, slog); dst[2] = make_float2(sabs, slin);
    dst[3] = make_float2(sinv2, nv); dst[4] = make_float2(bad, delta);
  }
}

template <bool SPEC>
__global__ __launch_bounds__(C1_WAVE, SPEC ? 4 : 3) void k_detect_features(C1EncodeLaunch L, float *bands_ws, double *feat_ws) {
  __shared__ typename std::conditional<SPEC, DetectSpecLds, DetectLds>::type S;
  const int lane0 = threadIdx.x;
  int lane = lane0;
  const int ch = blockIdx.x % L.channels;
  const int64_t f0 = (int64_t)(blockIdx.x / L.channels) * L.run_frames;
  const float *__restrict__ pcm = L.pcm[ch];
  for (int i = lane; i < 46; i += 64) { S.d1[i] = 0.0; S.d2[i] = 0.0; }
  for (int i = lane; i < 296; i += 64) S.hbuf[i] = 0.0f;
  float pmag[4] = {0.0f, 0.0f, 0.0f, 0.0f};      // magnitudes of the previous frame at this lane's four bins
  const TGeom G = tfft_geometry(lane0);
  const TablesRsrc RT = tables_rsrc(L.tables);
  wave_fence();

  const int64_t f_end = (f0 + L.run_frames < L.frames) ? f0 + L.run_frames : L.frames;
  int64_t f_first = f0 - 2;                                  // frame -2 rebuilds the QMF delay lines, frame -1 the magnitudes
  if (f_first < -(int64_t)L.halo_frames) f_first = -(int64_t)L.halo_frames;
  if (f_first > f0) f_first = f0;
  typedef float v4f __attribute__((ext_vector_type(4)));   // whole 16-byte register groups (as eight scalars the asm cost eight moves a frame)
  v4f pre_a, pre_b;
  {
    const v4f *p4 = reinterpret_cast<const v4f *>(pcm + f_first * 512);
    pre_a = p4[lane0]; pre_b = p4[64 + lane0];
    // delivered before the loop: a load still pending at the loop's entry makes the compiler wait inside the loop, every frame
    asm volatile("" : "+v"(pre_a), "+v"(pre_b));
  }
  for (int64_t f = f_first; f < f_end; ++f) {
    // frame -1 of the whole batch is emitted too (slot row 0): frame 0 needs its features and its band tails
    const bool emit = (f >= f0) || (f0 == 0 && f == -1);
    const bool qmf_only = (f == f0 - 2);
    TablesPtr T = tables_for_this_frame(L.tables);
    lane = lane_for_this_frame(lane0);

    // ---------------- qmfAnalysisStage (encoder.js:57-96) ----------------
    {
      const v4f a = pre_a, b = pre_b;
      double *w1 = S.u.q1.w1;
      if (lane < 46) w1[pidx<3>(lane)] = S.d1[lane];
      const int e0 = 46 + 4 * lane;
      *reinterpret_cast<double2 *>(&w1[pidx<3>(e0)]) = make_double2((double)a.x, (double)a.y);
      *reinterpret_cast<double2 *>(&w1[pidx<3>(e0 + 2)]) = make_double2((double)a.z, (double)a.w);
      *reinterpret_cast<double2 *>(&w1[pidx<3>(e0 + 256)]) = make_double2((double)b.x, (double)b.y);
      *reinterpret_cast<double2 *>(&w1[pidx<3>(e0 + 258)]) = make_double2((double)b.z, (double)b.w);
    }
    wave_fence();
    {
      // the next frame's PCM (the last frame asks for itself again: under a condition the loaded values are copied into the
      // loop-carried registers behind the load, i.e. waited for on the spot)
      const v4f *p4 = reinterpret_cast<const v4f *>(pcm + ((f + 1 < f_end) ? f + 1 : f) * 512);
      pre_a = p4[lane]; pre_b = p4[64 + lane];
    }
    {
      double ev[4], od[4];
      __builtin_amdgcn_s_setprio(3);   // wave priorities as in k_analysis_fast: QMF cores 3, transient FFT 0, the rest 1
      if (own_block()) qmf_analysis_core<4, 3>(S.u.q1.w1, lane, T, ev, od); else { for (int d = 0; d < 4; d++) { ev[d] = S.u.q1.w1[lane + d]; od[d] = 1.0; } }
      double *w2 = S.u.q2.w2;
      if (lane < 46) { w2[pidx<2>(lane)] = S.d2[lane]; S.d1[lane] = S.u.q1.w1[pidx<3>(512 + lane)]; }
      float lo[4];
#pragma unroll
      for (int d = 0; d < 4; d++) {
        lo[d] = f32(ev[d] + od[d]);
        S.hbuf[39 + 4 * lane + d] = f32(ev[d] - od[d]);
      }
      *reinterpret_cast<double2 *>(&w2[pidx<2>(46 + 4 * lane)]) = make_double2((double)lo[0], (double)lo[1]);
      *reinterpret_cast<double2 *>(&w2[pidx<2>(48 + 4 * lane)]) = make_double2((double)lo[2], (double)lo[3]);
    }
    wave_fence();
    {
      double ev[2], od[2];
      if (own_block()) qmf_analysis_core<2, 2>(S.u.q2.w2, lane, T, ev, od); else { for (int d = 0; d < 2; d++) { ev[d] = S.u.q2.w2[lane + d]; od[d] = 1.0; } }
      __builtin_amdgcn_s_setprio(1);
      *reinterpret_cast<float2 *>(&S.band[2 * lane]) = make_float2(f32(ev[0] + od[0]), f32(ev[1] + od[1]));
      *reinterpret_cast<float2 *>(&S.band[128 + 2 * lane]) = make_float2(f32(ev[0] - od[0]), f32(ev[1] - od[1]));
      *reinterpret_cast<float4 *>(&S.band[256 + 4 * lane]) = *reinterpret_cast<const float4 *>(&S.hbuf[4 * lane]);
      if (lane < 46) S.d2[lane] = S.u.q2.w2[pidx<2>(256 + lane)];
    }
    wave_fence();
    {
      float keep = 0.0f;
      if (lane < 39) keep = S.hbuf[256 + lane];
      wave_fence();
      if (lane < 39) S.hbuf[lane] = keep;
    }
    // The next frame's PCM, requested before the first QMF stage, is taken delivery of here -- at a point every path to the top
    // of the loop passes, and before any store of this frame is issued: a load still pending on ONE path makes the compiler wait
    // at the top of the loop on ALL of them, and there the wait would sit right behind the frame's stores (loads and stores
    // share one in-order counter on this part, vmcnt)
    asm volatile("" : "+v"(pre_a), "+v"(pre_b));
    if (qmf_only) { wave_fence(); continue; }
    const int64_t slot = (f + 1) * L.channels + ch;
    // Every store of the frame (bands, feature record) comes LAST: a wait for a load issued behind a store -- the table
    // values of the transient FFT -- would be a wait for that store to reach memory.
    auto store_bands = [&]() {
      if (emit) {
        float4 *dst = reinterpret_cast<float4 *>(bands_ws + (slot << 9));
        const float4 *src = reinterpret_cast<const float4 *>(S.band);
        dst[lane] = src[lane];
        dst[64 + lane] = src[64 + lane];
      }
    };

    float mg[4];
    if constexpr (SPEC) {
      float delta;
      __builtin_amdgcn_s_setprio(0);
      tfft_spec(S.band, S.u.t.z, G, T, RT, mg, delta);
      if (L.mags && f >= f0) {                                   // test tap: the binary32 magnitudes and their bound (3 floats per unit)
#pragma unroll
        for (int i = 0; i < 4; i++) L.mags[((f * L.channels + ch) << 8) + G.mag + i * G.S] = mg[i];
        if (L.mag_bounds && G.g == 0) L.mag_bounds[(f * L.channels + ch) * 3 + G.band] = delta;
      }
      store_bands();
      if (emit) spec_sums(G, lane, mg, pmag, delta, reinterpret_cast<float *>(feat_ws + slot * kFeatureDoubles));
    } else {
      store_bands();
      tfft_exact(S.band, S.u.t.z, G, T, RT, mg);
      if (L.mags && f >= f0) {                                   // stage tap: performFFT's magnitudes (transient.js:17-35)
#pragma unroll
        for (int i = 0; i < 4; i++) L.mags[((f * L.channels + ch) << 8) + G.mag + i * G.S] = mg[i];
      }
      if constexpr (!SPEC) { if (emit) exact_sums(S.u.tt.term, G, lane, mg, pmag, feat_ws + slot * kFeatureDoubles); }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) pmag[i] = mg[i];
    __builtin_amdgcn_s_setprio(1);
    wave_fence();
  }
}

// features of one band of one frame from its sums (transient.js:88-189); `flux` needs the previous magnitudes and
// is only meaningful for the current frame
struct BandFeatures { double flux, flat, hf, energy; };
__device__ __forceinline__ BandFeatures band_features(const double *s, int nv) {
  BandFeatures r;
  const double s_flux = s[0], s_e = s[1], s_log = s[2], s_lin = s[3], s_lo = s[4], s_hi = s[5];
  double norm = sqrt(s_e);
  if (!(norm != 0.0)) norm = 1e-6;                       // `Math.sqrt(e) || 1e-6`
  r.flux = s_flux / norm;
  r.flat = 0.0;                                          // calculateSpectralFlatness :120-141
  if (nv > 0) {
    const double gm = js_exp(s_log / (double)nv), am = s_lin / (double)nv;
    r.flat = am > 1e-10 ? gm / am : 0.0;
  }
  const double tot = s_lo + s_hi;                        // calculateHighFrequencyRatio :149-164
  r.hf = tot > 0 ? s_hi / tot : 0.0;
  r.energy = s_e;
  return r;
}

// block mode of band b of one sound unit from the feature sums of its frame (`cur`: 18 sums, nv[3] behind) and of the
// previous one (`prev`, or null for the zero state of a fresh BufferPool) (encoder.js:137-143)
__device__ __forceinline__ int detect_band_mode(const double *cur, const double *prev, int b, double log1p10, double threshold,
                                                double *score_out) {
  const BandFeatures c = band_features(cur + 6 * b, reinterpret_cast<const int *>(cur + 18)[b]);
  double prev_flat = 0.0, prev_hf = 0.0, prev_e = 0.0;
  if (prev) {
    const BandFeatures p = band_features(prev + 6 * b, reinterpret_cast<const int *>(prev + 18)[b]);
    prev_flat = p.flat; prev_hf = p.hf; prev_e = p.energy;
  }
  const double ce = c.energy > 1e-10 ? c.energy : 1e-10;     // calculateEnergyChange :172-189
  const double pe = prev_e > 1e-10 ? prev_e : 1e-10;
  const double db = 10.0 * js_log10(ce / pe);
  const double e_change = db > 0 ? db : 0.0;
  const double flat_c = sqrt(fabs(c.flat - prev_flat));       // calculateTransientScore :197-226
  const double hf_c = js_log1p(fabs(c.hf - prev_hf) * 10.0) / log1p10;
  const double e_c = e_change / 30.0 < 1.0 ? e_change / 30.0 : 1.0;
  const double score = (c.flux + flat_c + hf_c + e_c) / 4.0;
  if (score_out) *score_out = score;
  return (score > threshold) ? (b + 1 > 2 ? b + 1 : 2) : 0;   // encoder.js:143
}

// two work lists for the MDCT stage: all-long units and units with a short band (lists[0], lists[1] = counts, then
// `units` entries each), and behind them the units the speculative detector could not decide (lists[2]).  One atomic per
// 256-thread block and list: the three counters take about 5 ns per atomic whoever issues it, and one per wave (94 k
// for 2 M units) was 0.56 of the speculative decision kernel's 0.80 ms.
__device__ __forceinline__ void append_by_mode(bool live, int kind, int64_t unit, int64_t units, uint32_t *__restrict__ lists) {
  __shared__ uint32_t wave_count[3][4], wave_base[3][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t below = (1ull << lane) - 1ull;
  uint64_t m[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    m[k] = __ballot(live && kind == k);
    if (lane == 0) wave_count[k][wave] = (uint32_t)__popcll(m[k]);
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    const uint32_t c0 = wave_count[k][0], c1 = wave_count[k][1], c2 = wave_count[k][2], c3 = wave_count[k][3];
    const uint32_t total = c0 + c1 + c2 + c3;
    const uint32_t base = total ? atomicAdd(&lists[k], total) : 0u;
    wave_base[k][0] = base; wave_base[k][1] = base + c0; wave_base[k][2] = base + c0 + c1; wave_base[k][3] = base + c0 + c1 + c2;
  }
  __syncthreads();
  const uint64_t mine = kind == 0 ? m[0] : (kind == 1 ? m[1] : m[2]);
  if (live) lists[4 + (int64_t)kind * units + wave_base[kind][wave] + __popcll(mine & below)] = (uint32_t)unit;
}

// SPEC = false: the reference's decision from its 18 sums.  SPEC = true: the interval of c1_detect_bound.h from the
// binary32 sums; a unit with a band whose interval contains the threshold goes to the third list (k_detect_recheck).
// `tap` (tests): per unit and band {lo, hi} (SPEC) or {score, score}.
constexpr int kDecideRecFloats = C1_DET_ROWS * C1_DET_ROW_FLOATS + 1;   // a record in LDS: 40 floats + 1 of padding (conflict-free lane stride)
template <bool SPEC>
__global__ __launch_bounds__(256) void k_detect_decide(const double *__restrict__ feat_ws, int channels, int64_t frames,
                                                        int halo_frames, const C1DevTables *tables, const C1DevEncOpts *opts,
                                                        uint8_t *__restrict__ modes, uint32_t *__restrict__ lists, double *__restrict__ tap) {
  const int64_t unit0 = (int64_t)blockIdx.x * blockDim.x, unit = unit0 + threadIdx.x;
  const int64_t units = frames * channels;
  const bool live = unit < units;
  int mode_byte = 0;
  bool certain = true;
  // SPEC: the records of the block's 256 units and of the `channels` slots before them are one contiguous piece of the
  // workspace; a lane reading its own 160 bytes from there touches a cache line per load, so the block copies the piece
  // into LDS with coalesced 16-byte reads first (the kernel was bound by exactly that: 0.8 -> 0.1 ms per 2 M units)
  __shared__ float recs[SPEC ? (256 + C1_MAX_CHANNELS) * kDecideRecFloats : 1];
  if constexpr (SPEC) {
    const int64_t first = unit0, last = (unit0 + 256 < units ? unit0 + 256 : units) + channels;   // slots [first, last)
    const float4 *src = reinterpret_cast<const float4 *>(feat_ws + first * kFeatureDoubles);
    const int n4 = (int)(last - first) * (C1_DET_ROWS * C1_DET_ROW_FLOATS / 4);
    for (int i = threadIdx.x; i < n4; i += 256) {
      const float4 v = src[i];
      float *dst = recs + (i / 10) * kDecideRecFloats + 4 * (i % 10);
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
  }
  if (live) {
    const int64_t f = unit / channels;
    const bool have_prev = (f - 1 >= -(int64_t)halo_frames);      // else the zero state of a fresh BufferPool
    const double log1p10 = tables->log1p10, threshold = opts->threshold;
    if constexpr (SPEC) {
      const float *prev = recs + threadIdx.x * kDecideRecFloats;   // slot unit = frame f - 1
      const float *cur = prev + channels * kDecideRecFloats;       // slot unit + channels = frame f
      for (int b = 0; b < 3; b++) {
        const C1DetSums sc = c1_det_sums(cur, b);
        const C1DetSums sp = have_prev ? c1_det_sums(prev, b) : c1_det_zero_sums();
        const C1DetOwn oc = c1_det_own(sc);
        C1DetOwn op = oc;
        if (have_prev) op = c1_det_own(sp);
        double lo, hi;
        const int ok = c1_det_score(sc, oc, have_prev ? 1 : 0, sp, op, b == 2 ? 128 : 64, log1p10, &lo, &hi);
        if (tap) { tap[(unit * 3 + b) * 2] = lo; tap[(unit * 3 + b) * 2 + 1] = hi; }
        if (ok && lo > threshold) mode_byte |= (b + 1 > 2 ? b + 1 : 2) << (2 * b);
        else if (!(ok && hi < threshold)) certain = false;
      }
    } else {
      const double *cur = feat_ws + (unit + channels) * kFeatureDoubles;
      const double *prev = have_prev ? cur - (int64_t)channels * kFeatureDoubles : nullptr;
      for (int b = 0; b < 3; b++) {
        double score;
        mode_byte |= detect_band_mode(cur, prev, b, log1p10, threshold, &score) << (2 * b);
        if (tap) { tap[(unit * 3 + b) * 2] = score; tap[(unit * 3 + b) * 2 + 1] = score; }
      }
    }
  }
  if (live && certain) modes[unit] = (uint8_t)mode_byte;
  append_by_mode(live, !certain ? 2 : (mode_byte == 0 ? 0 : 1), unit, units, lists);
}

// The units the speculative detector left open: the reference's own arithmetic on the stored (exact) band samples of
// the frame and of its predecessor -- two exact transient FFTs, the 18 sequential sums of each, the scalar features --
// then the unit joins the MDCT list its block modes call for.  One wave per listed unit.
struct alignas(16) RecheckLds {
  alignas(16) float band[512];
  union alignas(16) {
    struct { alignas(16) float2 z[576]; } t;
    struct { alignas(16) double term[4][256]; } tt;
  } u;
  alignas(16) double feat_c[kFeatureDoubles];
  alignas(16) double feat_p[kFeatureDoubles];
  int mode[4];
};
__global__ __launch_bounds__(C1_WAVE, 3) void k_detect_recheck(C1EncodeLaunch L, const float *__restrict__ bands_ws,
                                                               uint8_t *__restrict__ modes, uint32_t *__restrict__ lists) {
  __shared__ RecheckLds S;
  const int lane = threadIdx.x;
  const int64_t units = L.frames * L.channels;
  const uint32_t count = lists[2];
  const uint32_t *__restrict__ list = lists + 4 + 2 * units;
  const TGeom G = tfft_geometry(lane);
  const TablesRsrc RT = tables_rsrc(L.tables);
  TablesPtr T = tables_for_this_frame(L.tables);
  for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
    const int64_t unit = list[i], slot = unit + L.channels;
    const bool have_prev = (unit / L.channels - 1 >= -(int64_t)L.halo_frames);
    float mg[4], pmag[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (have_prev) {
      const float4 *p4 = reinterpret_cast<const float4 *>(bands_ws + ((slot - L.channels) << 9));
      reinterpret_cast<float4 *>(S.band)[lane] = p4[lane];
      reinterpret_cast<float4 *>(S.band)[64 + lane] = p4[64 + lane];
      wave_fence();
      tfft_exact(S.band, S.u.t.z, G, T, RT, mg);
      exact_sums(S.u.tt.term, G, lane, mg, pmag, S.feat_p);    // its flux sum is not used
#pragma unroll
      for (int k = 0; k < 4; k++) pmag[k] = mg[k];
      wave_fence();
    }
    {
      const float4 *p4 = reinterpret_cast<const float4 *>(bands_ws + (slot << 9));
      reinterpret_cast<float4 *>(S.band)[lane] = p4[lane];
      reinterpret_cast<float4 *>(S.band)[64 + lane] = p4[64 + lane];
      wave_fence();
      tfft_exact(S.band, S.u.t.z, G, T, RT, mg);
      exact_sums(S.u.tt.term, G, lane, mg, pmag, S.feat_c);
      wave_fence();
    }
    if (lane < 3) S.mode[lane] = detect_band_mode(S.feat_c, have_prev ? S.feat_p : nullptr, lane, T->log1p10, L.opts->threshold, nullptr);
    wave_fence();
    if (lane == 0) {
      const int mode_byte = S.mode[0] | (S.mode[1] << 2) | (S.mode[2] << 4);
      modes[unit] = (uint8_t)mode_byte;
      const int k = mode_byte == 0 ? 0 : 1;
      lists[4 + (int64_t)k * units + atomicAdd(&lists[k], 1u)] = (uint32_t)unit;
    }
    wave_fence();
  }
}


struct alignas(16) MdctLds {
  alignas(16) float band[512];
  alignas(16) float ovl[96];
  alignas(4) uint8_t sfi[64];
  union alignas(16) {
    struct { alignas(16) float in0[256]; alignas(16) float in1[256]; alignas(16) float in2[512]; } i;   // long-block inputs
    struct { alignas(16) float in[kStageFloats]; } g;                                                    // staging of frames with short blocks
    struct { alignas(16) float coef[512]; } c;
  } a;
  union alignas(16) {
    float2 z[320];
  } zz;
};

// the all-long instantiation's image (no staging area of frames with short blocks) and what its waves share: the tables the
// long core reads with lane-varying indices (LdsTab, c1_device.h), WINDOW_SHORT, and lane-only geometry that is the same for
// every wave -- the end-of-transform values of mdct_long_r4 (r4_late_word) and the scale-factor scan's word (as k_analysis_fast)
struct alignas(16) MdctLdsLong {
  alignas(16) float band[512];
  alignas(16) float ovl[96];
  alignas(4) uint8_t sfi[64];
  union alignas(16) {
    struct { alignas(16) float in0[256]; alignas(16) float in1[256]; alignas(16) float in2[512]; } i;
    struct { alignas(16) float coef[512]; } c;
  } a;
  union alignas(16) {
    float2 z[320];
  } zz;
};
struct alignas(16) MdctShared {
  alignas(16) char tabs[kLdsTabBytes];
  alignas(16) double win[32];
  uint32_t late[4][64];
  uint32_t sfw[64];
};
struct alignas(16) MdctNoShared { alignas(16) char tabs[16]; alignas(16) double win[4]; uint32_t late[1][1]; uint32_t sfw[1]; };

// mdctStage + scale factors of one sound unit from the stored band samples; units are independent.  Two
// instantiations work through the two lists k_detect_decide wrote: LONG (all three bands long, the common case;
// lean enough for 4 waves per SIMD) and mixed (at least one short band).
// LONG: eight waves per workgroup, each with its own unit and LDS image; they share one copy of the (cos, sin) tables the long
// core reads with lane-varying indices (LdsTab, c1_device.h): sixteen such reads a unit, which through the cache were three
// round trips in the middle of every unit's chain, on the counter its loads and stores share (2.44 -> 2.29 ms per 2 M units
// of config 3).  With the window and the lane-only geometry shared as well the kernel needs 114 registers and 9.2 KB of LDS
// per wave: 4 waves per SIMD (two workgroups per CU) -- worth another 1 %: at 8 GB per launch and 3.6 TB/s the kernel is
// closer to the memory system's rate for equal reads and writes (4.85 TB/s for a copy) than to any limit of its own.
constexpr int kMdctWavesLong = 8;
template <bool LONG>
__global__ __launch_bounds__(C1_WAVE * (LONG ? kMdctWavesLong : 1), LONG ? 4 : 3) void k_mdct_bands(C1EncodeLaunch L, const float *__restrict__ bands_ws,
                                                                         const uint8_t *__restrict__ modes,
                                                                         const uint32_t *__restrict__ lists) {
  constexpr int kW = LONG ? kMdctWavesLong : 1;
  using Lds = typename std::conditional<LONG, MdctLdsLong, MdctLds>::type;
  __shared__ Lds Sw[kW];
  __shared__ typename std::conditional<LONG, MdctShared, MdctNoShared>::type SH;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  Lds &S = Sw[wave];
  const int lane0 = threadIdx.x & 63;
  int lane = lane0;
  R4Geometry G4 = r4_geometry(lane0);                // LONG
  if constexpr (LONG) {
    if (wave == 0) {
      const SfLong SFL = sf_long_geometry(lane0);
#pragma unroll
      for (int j = 0; j < 4; j++) SH.late[j][lane0] = r4_late_word(G4, j);
      SH.sfw[lane0] = (uint32_t)SFL.src | ((uint32_t)SFL.b << 13) | (SFL.wide ? 1u << 19 : 0u) | (SFL.store ? 1u << 20 : 0u);
      if (lane0 < 32) SH.win[lane0] = C1_TABLES(L.tables)->window[lane0];
    }
    const char *src = reinterpret_cast<const char *>(L.tables);
    for (int k = 16 * (int)threadIdx.x; k < kLdsTabBytes; k += 16 * C1_WAVE * kW)
      *reinterpret_cast<uint4 *>(SH.tabs + k) = *reinterpret_cast<const uint4 *>(src + lds_tab_source(k));
    __syncthreads();                                   // the only time the waves of a workgroup meet
#pragma unroll
    for (int j = 0; j < 4; j++) G4.cx[j] = G4.cy[j] = G4.post_tab[j] = 0;   // read back from SH.late at the end of every transform
    G4 = r4_geometry_lds(G4);                          // table offsets -> offsets into SH.tabs
  }
  const LdsTab LT{SH.tabs};
  const int64_t units = L.frames * L.channels;
  const uint32_t count = lists[LONG ? 0 : 1];
  const uint32_t *__restrict__ list = lists + 4 + (LONG ? 0 : units);
  const int my_size = lane0 < 52 ? kSpecs[lane0] : 0, my_long = lane0 < 52 ? kStartLong[lane0] : 0, my_short = lane0 < 52 ? kStartShort[lane0] : 0;
  const TablesRsrc RT = tables_rsrc(L.tables);
  // tails of the previous frame: lane < 24 loads four samples of band lane / 8
  const int tail_band = lane0 >> 3, tail_k = 4 * (lane0 & 7);
  const int tail_src = (tail_band == 0 ? 96 : (tail_band == 1 ? 224 : 480)) + tail_k;
  // window values the lane needs every unit: fixed per lane -- kept in registers by the mixed instantiation, read from the
  // shared copy where they are used by the all-long one (ten registers towards its 128)
  double wt0 = 0.0, wt1 = 0.0, wt2 = 0.0, wt3 = 0.0;
  if constexpr (!LONG) {
    wt0 = C1_TABLES(L.tables)->window[tail_k & 31]; wt1 = C1_TABLES(L.tables)->window[(tail_k + 1) & 31];
    wt2 = C1_TABLES(L.tables)->window[(tail_k + 2) & 31]; wt3 = C1_TABLES(L.tables)->window[(tail_k + 3) & 31];
  }
  uint32_t i = blockIdx.x * kW + (uint32_t)wave;
  const uint32_t stride = gridDim.x * kW;
  if (i >= count) return;
  typedef float v4f __attribute__((ext_vector_type(4)));   // whole 16-byte register groups for the delivery asm below
  v4f pre_a, pre_b, pre_t = {0.0f, 0.0f, 0.0f, 0.0f};
  int pre_mode = 0;
  bool pre_have = false;                                     // wave-uniform: the fetched unit has a previous frame in the workspace
  auto fetch = [&](int64_t u) {
    const int64_t slot = u + L.channels;
    const v4f *p4 = reinterpret_cast<const v4f *>(bands_ws + (slot << 9));
    pre_a = p4[lane0]; pre_b = p4[64 + lane0];
    pre_have = ((L.channels == 2 ? u >> 1 : u) - 1 >= -(int64_t)L.halo_frames);
    // the tails are loaded unconditionally (from the unit's own slot where there is no previous frame; every lane: tail_src
    // stays inside the slot) and masked when they are used: a load under a condition is waited for where it is issued
    pre_t = *reinterpret_cast<const v4f *>(bands_ws + ((pre_have ? slot - L.channels : slot) << 9) + tail_src);
    if (!LONG) pre_mode = modes[u];
  };
  // The samples of the NEXT unit are requested while this one is transformed, unconditionally (past the end of the list the
  // last unit is fetched again: under a condition the loaded values are copied into the loop-carried registers right behind
  // the loads, i.e. waited for on the spot), and taken delivery of before this unit's stores are issued: loads and stores
  // share one in-order counter on this part (vmcnt), so a wait for a load behind a store is a wait for the store as well.
  auto listed = [&](uint32_t k) -> int64_t { return (int64_t)list[k < count ? k : count - 1]; };
  auto deliver = [&]() {
    asm volatile("" : "+v"(pre_a), "+v"(pre_b), "+v"(pre_t), "+v"(pre_mode));
  };
  int64_t unit = listed(i);
  int64_t unit_next = listed(i + stride);
  fetch(unit);
  deliver();
  for (; i < count; i += stride) {
    TablesPtr T = tables_for_this_frame(L.tables);
    lane = lane_for_this_frame(lane0);
    const v4f zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const v4f a = pre_a, b = pre_b, t = (pre_have && lane < 24) ? pre_t : zero4;
    const int mode_byte = LONG ? 0 : __builtin_amdgcn_readfirstlane(pre_mode);
    const int64_t unit_now = unit;
    unit = unit_next;
    fetch(unit);
    unit_next = listed(i + 2 * stride);
    reinterpret_cast<v4f *>(S.band)[lane] = a;
    reinterpret_cast<v4f *>(S.band)[64 + lane] = b;
    if (lane < 24) {
      // mdctOverlap of the previous frame (applyTailWindowing, encoder.js:309-316): W[k] * tail sample
      float4 o;
      if constexpr (LONG) {
        const double2 w01 = *reinterpret_cast<const double2 *>(&SH.win[(4 * (lane & 7)) & 31]), w23 = *reinterpret_cast<const double2 *>(&SH.win[((4 * (lane & 7)) & 31) + 2]);
        o.x = f32(w01.x * (double)t.x); o.y = f32(w01.y * (double)t.y);
        o.z = f32(w23.x * (double)t.z); o.w = f32(w23.y * (double)t.w);
      } else {
        o.x = f32(wt0 * (double)t.x); o.y = f32(wt1 * (double)t.y);
        o.z = f32(wt2 * (double)t.z); o.w = f32(wt3 * (double)t.w);
      }
      reinterpret_cast<float4 *>(S.ovl)[lane] = o;
    }
    wave_fence();
    const FrameModes M{mode_byte & 3, (mode_byte >> 2) & 3, (mode_byte >> 4) & 3};
    float *coef = S.a.c.coef;
    if constexpr (LONG) {
      // ---------------- long blocks (encoder.js:228-258) ----------------
      float *in0 = S.a.i.in0, *in1 = S.a.i.in1, *in2 = S.a.i.in2;
      const float *band_ = S.band;
      if (lane < 32) {
        const double w_hi = SH.win[31 - lane];
        const double x0 = band_[96 + lane], x1 = band_[128 + 96 + lane], x2 = band_[256 + 224 + lane];
        in0[48 + lane] = S.ovl[lane]; in1[48 + lane] = S.ovl[32 + lane]; in2[112 + lane] = S.ovl[64 + lane];
        in0[80 + 96 + lane] = f32(x0 * w_hi);
        in1[80 + 96 + lane] = f32(x1 * w_hi);
        in2[144 + 224 + lane] = f32(x2 * w_hi);
      }
      {
        const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (lane < 48) {
          float *inb = lane < 24 ? in0 : in1;
          const int q = lane < 24 ? lane : lane - 24;                 // 24 float4 per band: [0,48) and [208,256)
          *reinterpret_cast<float4 *>(&inb[q < 12 ? 4 * q : 208 + 4 * (q - 12)]) = zero4;
        }
        if (lane < 56) *reinterpret_cast<float4 *>(&in2[lane < 28 ? 4 * lane : 400 + 4 * (lane - 28)]) = zero4;   // [0,112), [400,512)
        if (lane < 48) {
          *reinterpret_cast<float2 *>(&in0[80 + 2 * lane]) = *reinterpret_cast<const float2 *>(&band_[2 * lane]);
          *reinterpret_cast<float2 *>(&in1[80 + 2 * lane]) = *reinterpret_cast<const float2 *>(&band_[128 + 2 * lane]);
        }
        if (lane < 56) *reinterpret_cast<float4 *>(&in2[144 + 4 * lane]) = *reinterpret_cast<const float4 *>(&band_[256 + 4 * lane]);
      }
      wave_fence();
      mdct_long_r4_t(in0, S.zz.z, coef, G4, T, LT, r4_early_t(G4, LT), &SH.late[0][0] + lane);
      wave_fence();
    } else {
      const MixGeometry GM = mix_geometry(lane, M);
      mix_stage(S.band, S.ovl, S.a.g.in, M, lane, RT);
      wave_fence();
      mdct_mixed_r4(S.a.g.in, S.zz.z, coef, GM, M.m0 == 0 || M.m1 == 0 || M.m2 == 0, M.m2 == 0, T, RT);
      wave_fence();
    }
    // ---------------- coefficients out + scale-factor indices (bitallocation.js:80-90) ----------------
    deliver();
    {
      float4 *dst = reinterpret_cast<float4 *>(L.coefs + (unit_now << 9));
      const float4 *src = reinterpret_cast<const float4 *>(coef);
      dst[lane] = src[lane];
      dst[64 + lane] = src[64 + lane];
    }
    if constexpr (LONG) {
      {
        // the scan of sf_long from three 16-byte reads (sf_scan_long_groups), its geometry read back from the shared word
        const uint32_t sw = SH.sfw[lane];
        const float4 *grp = reinterpret_cast<const float4 *>(coef + (sw & 0x1fcu));
        float mx = sf_scan_long_groups(grp[0], grp[1], grp[2]);
        const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0xB1, 0xf, 0xf, false));
        mx = fmaxf(mx, ((sw >> 19) & 1u) ? other : 0.0f);
        const int sfi = T->sf_fast ? scale_factor_index_fast(mx, T->sf_m1, T->sf_m2) : scale_factor_index(mx, T);
        if ((sw >> 20) & 1u) S.sfi[(sw >> 13) & 63u] = (uint8_t)sfi;
      }
      if (lane >= 60) reinterpret_cast<uint32_t *>(S.sfi)[13 + (lane - 60) % 3] = 0;   // modes byte (all long) and padding
    } else {
      if (lane < 52) {
        const int start = M.mode_of_band(band_of_bfu(lane)) == 0 ? my_long : my_short;
        const int n = my_size;
        float mx = 0.0f;
        for (int j = 0; j < n; j++) mx = fmaxf(mx, fabsf(coef[start + j]));
        S.sfi[lane] = (uint8_t)(T->sf_fast ? scale_factor_index_fast(mx, T->sf_m1, T->sf_m2) : scale_factor_index(mx, T));
      } else {
        S.sfi[lane] = lane == 52 ? (uint8_t)mode_byte : 0;
      }
    }
    wave_fence();
    if (lane < 16) reinterpret_cast<uint32_t *>(L.side + unit_now * kSideBytes)[lane] = reinterpret_cast<const uint32_t *>(S.sfi)[lane];
    wave_fence();
  }
}

}  // namespace

void c1k_launch_mdct_bands(const C1EncodeLaunch &L, const float *bands_ws, const uint8_t *modes_ws, const uint32_t *lists_ws, hipStream_t stream) {
  const int64_t units = L.frames * L.channels;
  const dim3 grid((unsigned)std::min<int64_t>(units, 256 * 48)), block(C1_WAVE);
  const dim3 lgrid((unsigned)std::min<int64_t>((units + kMdctWavesLong - 1) / kMdctWavesLong, 256 * 48 / kMdctWavesLong)), lblock(C1_WAVE * kMdctWavesLong);
  hipLaunchKernelGGL((k_mdct_bands<true>), lgrid, lblock, 0, stream, L, bands_ws, modes_ws, lists_ws);
  hipLaunchKernelGGL((k_mdct_bands<false>), grid, block, 0, stream, L, bands_ws, modes_ws, lists_ws);
}

void c1k_launch_detect(const C1EncodeLaunch &L0, float *bands_ws, double *feat_ws, uint8_t *modes_ws, uint32_t *lists_ws,
                       bool speculative, double *score_tap, hipStream_t stream) {
  static const int slots = c1k_wave_slots(k_detect_features<false>);
  C1EncodeLaunch L = L0;
  L.run_frames = c1k_pick_run(L.frames, L.channels, slots);
  const int64_t runs = (L.frames + L.run_frames - 1) / L.run_frames, units = L.frames * L.channels;
  (void)hipMemsetAsync(lists_ws, 0, 4 * sizeof(uint32_t), stream);
  const dim3 fgrid((unsigned)(runs * L.channels)), dgrid((unsigned)((units + 255) / 256));
  if (speculative) {
    // binary32 transient FFT and sums, the decision where the score's interval allows it, the reference's arithmetic
    // for the units left open; lists_ws[2] = how many those were
    hipLaunchKernelGGL((k_detect_features<true>), fgrid, dim3(C1_WAVE), 0, stream, L, bands_ws, feat_ws);
    hipLaunchKernelGGL((k_detect_decide<true>), dgrid, dim3(256), 0, stream, feat_ws, L.channels, L.frames, L.halo_frames,
                       L.tables, L.opts, modes_ws, lists_ws, score_tap);
    hipLaunchKernelGGL(k_detect_recheck, dim3((unsigned)std::min<int64_t>(units, 256 * 12)), dim3(C1_WAVE), 0, stream, L, bands_ws,
                       modes_ws, lists_ws);
  } else {
    hipLaunchKernelGGL((k_detect_features<false>), fgrid, dim3(C1_WAVE), 0, stream, L, bands_ws, feat_ws);
    hipLaunchKernelGGL((k_detect_decide<false>), dgrid, dim3(256), 0, stream, feat_ws, L.channels, L.frames, L.halo_frames,
                       L.tables, L.opts, modes_ws, lists_ws, score_tap);
  }
  if (!L.coefs) return;                                      // decisions only (score taps)
  // both list kernels size their grids for the whole batch and stop at the device-side count
  const dim3 grid((unsigned)std::min<int64_t>(units, 256 * 48)), block(C1_WAVE);
  const dim3 lgrid((unsigned)std::min<int64_t>((units + kMdctWavesLong - 1) / kMdctWavesLong, 256 * 48 / kMdctWavesLong)), lblock(C1_WAVE * kMdctWavesLong);
  hipLaunchKernelGGL((k_mdct_bands<true>), lgrid, lblock, 0, stream, L, bands_ws, modes_ws, lists_ws);
  hipLaunchKernelGGL((k_mdct_bands<false>), grid, block, 0, stream, L, bands_ws, modes_ws, lists_ws);
}

// accuracy of v_log_f32 over a range of binary32 bit patterns (normal, positive): out[0] = max |r - log2 x| / |log2 x| in
// units of 2^-24 where |log2 x| >= 2^-6, out[1] = max |r - log2 x| elsewhere (both as the bit patterns of non-negative doubles)
__global__ __launch_bounds__(256) void k_log2f_error(uint32_t first, uint64_t count, unsigned long long *out) {
  double rel = 0.0, absolute = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float(first + (uint32_t)i);
    const double exact = log2((double)x), got = (double)__builtin_amdgcn_logf(x);
    const double err = fabs(got - exact);
    if (fabs(exact) >= 0.015625) rel = fmax(rel, err / fabs(exact) * 16777216.0); else absolute = fmax(absolute, err);
  }
  atomicMax(&out[0], (unsigned long long)__double_as_longlong(rel));
  atomicMax(&out[1], (unsigned long long)__double_as_longlong(absolute));
}

void c1k_launch_log2f_error(uint32_t first, uint64_t count, unsigned long long *out, hipStream_t stream) {
  (void)hipMemsetAsync(out, 0, 2 * sizeof(unsigned long long), stream);
  if (count == 0) return;
  hipLaunchKernelGGL(k_log2f_error, dim3(256 * 32), dim3(256), 0, stream, first, count, out);
}

// test tap: the speculative detector's first kernel alone, writing its binary32 magnitudes (L.mags) and bounds (L.mag_bounds)
void c1k_launch_detect_spec_tap(const C1EncodeLaunch &L0, float *bands_ws, double *feat_ws, hipStream_t stream) {
  static const int slots = c1k_wave_slots(k_detect_features<true>);
  C1EncodeLaunch L = L0;
  L.run_frames = c1k_pick_run(L.frames, L.channels, slots);
  const int64_t runs = (L.frames + L.run_frames - 1) / L.run_frames;
  hipLaunchKernelGGL((k_detect_features<true>), dim3((unsigned)(runs * L.channels)), dim3(C1_WAVE), 0, stream, L, bands_ws, feat_ws);
}

void c1k_launch_libm(int fn, const double *in, double *out, int64_t n, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_libm_tap, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, fn, in, out, n);
}
