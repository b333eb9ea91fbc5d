// c1_api.hip -- host side of libcarta1_hip.so: the C ABI of include/carta1_hip.h.
// Contexts, table upload, workspace, chunking of large batches, stateful streams, profiling events.
// Compiled with -ffp-contract=off: the twiddle recurrence and the normalisation table below are part
// of the reference's numerics (codec/transforms/fft.js:62-64, codec/coding/quantization.js:42-44).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "c1_internal.h"
#include "c1_detect_bound.h"

#pragma clang fp contract(off)

namespace {

#include "c1_default_tables.inc"

thread_local std::string g_error;
int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}
#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(C1_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

std::mutex g_tables_mutex;
bool g_tables_custom = false;
std::atomic<uint64_t> g_tables_gen{0};        // bumped by every c1_set_tables(): pooled contexts made before it are retired
c1_tables g_tables;

void default_tables(c1_tables *t) {
  memcpy(t->scale_factors, C1D_SCALE_FACTORS, sizeof t->scale_factors);
  memcpy(t->window_short, C1D_WINDOW_SHORT, sizeof t->window_short);
  memcpy(t->mdct_fwd64, C1D_MDCT_FWD64, sizeof t->mdct_fwd64);
  memcpy(t->mdct_fwd256, C1D_MDCT_FWD256, sizeof t->mdct_fwd256);
  memcpy(t->mdct_fwd512, C1D_MDCT_FWD512, sizeof t->mdct_fwd512);
  memcpy(t->mdct_inv64, C1D_MDCT_INV64, sizeof t->mdct_inv64);
  memcpy(t->mdct_inv256, C1D_MDCT_INV256, sizeof t->mdct_inv256);
  memcpy(t->mdct_inv512, C1D_MDCT_INV512, sizeof t->mdct_inv512);
  memcpy(t->fft_w, C1D_FFT_W, sizeof t->fft_w);
  t->log1p_10 = C1D_LOG1P_10;
}

// QMF prototype (codec/core/constants.js:74-107): `new Float32Array([...])` rounds the decimal
// literals decimal -> double -> float; the 48-tap window is 2 * prototype, mirrored; EVEN = window[2j].
void qmf_even_taps(double out[24]) {
  static const double proto[24] = {
      -0.00001461907, -0.00009205479, -0.000056157569, 0.00030117269, 0.0002422519, -0.00085293897,
      -0.0005205574,  0.0020340169,   0.00078333891,   -0.0042153862, -0.00075614988, 0.0078402944,
      -0.000061169922, -0.01344162,   0.0024626821,    0.021736089,   -0.007801671,  -0.034090221,
      0.01880949,     0.054326009,    -0.043596379,    -0.099384367,  0.13207909,    0.46424159};
  float window[48];
  for (int i = 0; i < 24; i++) {
    const float c = (float)proto[i];
    window[i] = (float)((double)c * 2.0);
    window[47 - i] = window[i];
  }
  for (int j = 0; j < 24; j++) out[j] = (double)window[2 * j];
}

// ---- speculative binary32 path (c1_k_spec.hip): rounded tables and the coefficients of its error bound -------------
// DESIGN.md 3b derives   eps_b = u (1 + theta) [ KA_b sigma sqrt(n_b) Z_b + sigma^2 sqrt(2 n_b) (D_b + 3 T_b) ] + eabs
// with u = 2^-24, Z_b the measured norm of the band's pre-twiddled points, W / L the measured norms of the PCM and of
// the first-stage low band over the two frames a unit depends on, and
//   D_2 = (2 gH + gQ) W,  T_2 = gH W;   D_0 = D_1 = gH (2 L + gQ W) + (2 gH + gQ) L,  T_0 = T_1 = gH L.
// gH bounds the l2 gain of one QMF branch (sqrt(|E|^2 + |O|^2) of the polyphase responses), gQ the accumulated
// rounding of the two-chain binary32 convolution in units of u * |input|; both depend only on the QMF prototype, which
// is compiled in (tools/spec_constants.py recomputes them from the taps; tests/test_spec_bound.py checks these are >=).
constexpr double kSpecGH = 1.4160, kSpecGQ = 4.80;
// KA = post-twiddle (1 + 2 sqrt 2) + 1  |  FFT rounds  |  pre-twiddle (1 + 3 sqrt 2) + 1      (reference's share + ours)
constexpr double kSpecKAPost = 4.83, kSpecKAPre = 6.25;
constexpr double kSpecKARoundA = 4.0, kSpecKARound4 = 7.0, kSpecKARound2 = 5.0;
constexpr double kSpecTheta = 1.01;
float round_up_f32(double x) {
  float f = (float)x;
  if ((double)f < x) f = std::nextafterf(f, INFINITY);
  return f;
}
void build_spec_tables(const c1_tables &t, C1DevTables *d) {
  for (int j = 0; j < 24; j++) d->tap32[j] = (float)d->tap_e[j];
  for (int j = 0; j < 24; j++) { d->tap_pair[j][0] = d->tap32[23 - j]; d->tap_pair[j][1] = d->tap32[j]; }
  d->tap_pair[24][0] = 0.0f; d->tap_pair[24][1] = d->tap32[11];
  d->tap_pair[25][0] = d->tap32[11]; d->tap_pair[25][1] = 0.0f;
  bool ok = d->sf_fast != 0;
  for (int i = 0; i < 32; i++) {
    d->win32[i] = (float)t.window_short[i];
    if (!(t.window_short[i] >= 0.0 && t.window_short[i] <= 1.0)) ok = false;
  }
  for (int i = 0; i < 16; i++) { d->pre32_64[i][0] = (float)t.mdct_fwd64[2 * i]; d->pre32_64[i][1] = (float)t.mdct_fwd64[2 * i + 1]; }
  for (int i = 0; i < 64; i++) { d->pre32_256[i][0] = (float)t.mdct_fwd256[2 * i]; d->pre32_256[i][1] = (float)t.mdct_fwd256[2 * i + 1]; }
  for (int i = 0; i < 128; i++) { d->pre32_512[i][0] = (float)t.mdct_fwd512[2 * i]; d->pre32_512[i][1] = (float)t.mdct_fwd512[2 * i + 1]; }
  // every (cos, sin) pair of an MDCT table has the same modulus sigma = sqrt(scale / N) (mdct.js:27-36)
  auto sigma2 = [&](const double *tab, int pairs, double want) {
    double mx = 0;
    for (int i = 0; i < pairs; i++) {
      const double m = tab[2 * i] * tab[2 * i] + tab[2 * i + 1] * tab[2 * i + 1];
      if (!(std::fabs(m - want) <= 1e-9 * want)) ok = false;
      mx = std::max(mx, m);
    }
    return mx;
  };
  const double s64 = sigma2(t.mdct_fwd64, 16, 0.5 / 64), s256 = sigma2(t.mdct_fwd256, 64, 0.5 / 256), s512 = sigma2(t.mdct_fwd512, 128, 1.0 / 512);
  // the FFT twiddles must be the unit-modulus roots the radix-4 regrouping assumes (to 1e-12; the bound's theta absorbs that)
  const double (*tw)[2] = d->fft_tw;
  for (int h = 1; h <= 128; h <<= 1)
    for (int k = 0; k < h; k++) {
      const double ang = -M_PI * (double)k / (double)h;
      if (std::fabs(tw[h - 1 + k][0] - std::cos(ang)) > 1e-12 || std::fabs(tw[h - 1 + k][1] - std::sin(ang)) > 1e-12) ok = false;
    }
  for (int r = 0; r < 2; r++) {
    const int h = r == 0 ? 4 : 16;
    for (int k = 0; k < h; k++) {
      const double *a = tw[h - 1 + k], *b = tw[2 * h - 1 + k];
      float (*dst)[2] = r == 0 ? d->r4b[k] : d->r4c[k];
      dst[0][0] = (float)a[0]; dst[0][1] = (float)a[1];
      dst[1][0] = (float)b[0]; dst[1][1] = (float)b[1];
      dst[2][0] = (float)(a[0] * b[0] - a[1] * b[1]);
      dst[2][1] = (float)(a[0] * b[1] + a[1] * b[0]);
    }
  }
  for (int k = 0; k < 64; k++) { d->r2d[k][0] = (float)tw[63 + k][0]; d->r2d[k][1] = (float)tw[63 + k][1]; }
  for (int i = 0; i < 64 * 16; i++) d->norm32[i] = (float)d->norm[i];
  for (int i = 0; i < 16; i++) { d->inv32_64[i][0] = (float)t.mdct_inv64[2 * i]; d->inv32_64[i][1] = (float)t.mdct_inv64[2 * i + 1]; }
  for (int i = 0; i < 64; i++) { d->inv32_256[i][0] = (float)t.mdct_inv256[2 * i]; d->inv32_256[i][1] = (float)t.mdct_inv256[2 * i + 1]; }
  for (int i = 0; i < 128; i++) { d->inv32_512[i][0] = (float)t.mdct_inv512[2 * i]; d->inv32_512[i][1] = (float)t.mdct_inv512[2 * i + 1]; }
  for (int i = 0; i < 256; i++) { d->tw32[i][0] = (float)tw[i][0]; d->tw32[i][1] = (float)tw[i][1]; }
  const double u = std::ldexp(1.0, -24) * kSpecTheta;
  const double ka64 = kSpecKAPost + kSpecKARoundA + 2 * kSpecKARound4 + kSpecKAPre;
  const double ka128 = ka64 + kSpecKARound2;
  const double ka16 = kSpecKAPost + kSpecKARoundA + kSpecKARound4 + kSpecKAPre;
  for (int b = 0; b < 3; b++) {
    const double n = b == 2 ? 128 : 64, sg2 = b == 2 ? s512 : s256;
    d->spec_cz[b] = round_up_f32(u * (b == 2 ? ka128 : ka64) * std::sqrt(sg2 * n));
    const double gb = u * sg2 * std::sqrt(2 * n);
    if (b == 2) { d->spec_cw[b] = round_up_f32(gb * (5 * kSpecGH + kSpecGQ)); d->spec_cl[b] = 0.0f; }
    else { d->spec_cw[b] = round_up_f32(gb * kSpecGH * kSpecGQ); d->spec_cl[b] = round_up_f32(gb * (7 * kSpecGH + kSpecGQ)); }
    d->spec_cz_short[b] = round_up_f32(u * ka16 * std::sqrt(s64 * 16));
    const double gs = u * s64 * std::sqrt(2.0 * 16);
    if (b == 2) { d->spec_cw_short[b] = round_up_f32(gs * (5 * kSpecGH + kSpecGQ)); d->spec_cl_short[b] = 0.0f; }
    else { d->spec_cw_short[b] = round_up_f32(gs * kSpecGH * kSpecGQ); d->spec_cl_short[b] = round_up_f32(gs * (7 * kSpecGH + kSpecGQ)); }
  }
  d->spec_cz[3] = d->spec_cw[3] = d->spec_cl[3] = d->spec_cz_short[3] = d->spec_cw_short[3] = d->spec_cl_short[3] = 0.0f;
  d->spec_eabs = (float)std::ldexp(1.0, -70);
  // speculative transient detector (c1_detect_bound.h): Delta_b = K u theta sqrt(n) ||band samples|| + eabs
  for (int b = 0; b < 3; b++)
    d->det_ck[b] = round_up_f32(std::ldexp(1.0, -24) * C1_DET_THETA * (b == 2 ? C1_DET_K256 * 16.0 : C1_DET_K128 * std::sqrt(128.0)));
  d->det_ck[3] = 0.0f;
  d->det_eabs = round_up_f32(C1_DET_EABS);
  d->spec_ok = ok ? 1 : 0;
}

void build_device_tables(const c1_tables &t, C1DevTables *d) {
  memset(d, 0, sizeof *d);
  qmf_even_taps(d->tap_e);
  for (int j = 0; j < 24; j++) d->tap_o[j] = d->tap_e[23 - j];
  memcpy(d->window, t.window_short, sizeof d->window);
  memcpy(d->mdct_fwd64, t.mdct_fwd64, sizeof d->mdct_fwd64);
  memcpy(d->mdct_fwd256, t.mdct_fwd256, sizeof d->mdct_fwd256);
  memcpy(d->mdct_fwd512, t.mdct_fwd512, sizeof d->mdct_fwd512);
  memcpy(d->mdct_inv64, t.mdct_inv64, sizeof d->mdct_inv64);
  memcpy(d->mdct_inv256, t.mdct_inv256, sizeof d->mdct_inv256);
  memcpy(d->mdct_inv512, t.mdct_inv512, sizeof d->mdct_inv512);
  // twiddle recurrence of FFT.fft (fft.js:44-64): starts at (1,0), advanced by a complex multiply in
  // double, unfused; it does not depend on the data, so it is tabulated once per stride
  int stage = 0;
  for (int h = 1; h <= 128; h <<= 1, stage++) {
    const double wr = t.fft_w[stage][0], wi = t.fft_w[stage][1];
    double tr = 1.0, ti = 0.0;
    for (int k = 0; k < h; k++) {
      d->fft_tw[h - 1 + k][0] = tr;
      d->fft_tw[h - 1 + k][1] = ti;
      const double nr = tr * wr - ti * wi;
      ti = tr * wi + ti * wr;
      tr = nr;
    }
  }
  memcpy(d->scale_factors, t.scale_factors, sizeof d->scale_factors);
  for (int s = 0; s < 64; s++)
    for (int wl = 0; wl < 16; wl++) {
      const int bits = wl == 0 ? 0 : wl + 1;
      const int range = bits ? (1 << (bits - 1)) - 1 : 0;
      d->norm[s * 16 + wl] = (double)range / t.scale_factors[s];  // quantization.js:42-44
    }
  d->log1p10 = t.log1p_10;
  // Float32 thresholds of findScaleFactor: for a binary32 m, m > SCALE_FACTORS[i] <=> m > floor_f32(SF[i]).
  // The reference's table is 2^(i/3-21): every octave has the same two fraction patterns; verify, else
  // the kernels fall back to comparing against the double table.
  auto floor_f32_bits = [](double x) -> uint32_t {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, 0.0f);
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
  };
  d->sf_fast = 1;
  d->sf_m1 = floor_f32_bits(t.scale_factors[1]) & 0x7fffffu;
  d->sf_m2 = floor_f32_bits(t.scale_factors[2]) & 0x7fffffu;
  for (int i = 0; i < 64; i++) {
    const uint32_t u = floor_f32_bits(t.scale_factors[i]);
    const int e = (int)(u >> 23) - 127, want_e = i / 3 - 21;
    const uint32_t frac = u & 0x7fffffu, want = i % 3 == 0 ? 0u : (i % 3 == 1 ? d->sf_m1 : d->sf_m2);
    const bool exact_pow2 = i % 3 != 0 || (double)std::ldexp(1.0f, want_e) == t.scale_factors[i];
    if (e != want_e || frac != want || !exact_pow2) d->sf_fast = 0;
  }
  // dequantize (quantization.js:65-78): Float32((q * SF) / range).  One correctly rounded reciprocal, a multiply
  // and two FMAs give the correctly rounded quotient (Markstein); rather than rely on the theorem's side
  // conditions, compare against the division for every input the decoder can meet with this table.
  d->dq_fast = 1;
  d->dq_step = 1;
  for (int wl = 1; wl < 16 && d->dq_fast; wl++) {
    const int bits = wl + 1;
    const double range = (double)((1 << (bits - 1)) - 1), y = 1.0 / range;
    d->inv_range[wl] = y;
    for (int s = 1; s < 64 && d->dq_fast; s++) {
      const double sf = t.scale_factors[s];
      for (int q = -(1 << (bits - 1)); q < (1 << (bits - 1)); q++) {
        const double a = (double)q * sf;
        const double q0 = a * y, r = std::fma(-q0, range, a), fast = std::fma(r, y, q0), exact = a / range;
        if (!(fast == exact) || std::signbit(fast) != std::signbit(exact)) { d->dq_fast = 0; break; }
        // after the store to the Float32 array even the plain product with one rounded step per BFU is the same value
        const float f_exact = (float)exact, f_step = (float)((double)q * (sf * y));
        if (!(f_step == f_exact) || std::signbit(f_step) != std::signbit(f_exact)) d->dq_step = 0;
      }
    }
  }
  if (!d->dq_fast || getenv("C1_NO_DQ_STEP")) d->dq_step = 0;      // the variable: tests of the reciprocal form, which the step form shadows
  build_spec_tables(t, d);
}

// rank table of the Float32 heap priorities (bitallocation.js:226-231, 267-269)
// C1DevEncOpts = the fields of this call (threshold, block modes: fill_call_fields, run on every call) + what derives from
// the biased scale-factor table alone (the table itself, its log2 line, the rank tables and their integer form:
// build_table_fields).  The search for an integer form costs ~30 M host operations when there is none, so the last few
// table-derived parts are kept; nothing that depends on another option may live in that cache.
int fill_call_fields(const c1_encode_options &o, C1DevEncOpts *d) {
  if (std::isnan(o.transient_threshold)) return fail(C1_ERR_ARG, "transient_threshold is NaN");
  d->threshold = o.transient_threshold;
  const bool detect = o.fixed_block_modes[0] < 0;
  for (int b = 0; b < 3; b++) {
    const int m = o.fixed_block_modes[b];
    if (detect) { d->modes[b] = -1; continue; }
    if (m < 0 || m > (b == 2 ? 3 : 2))
      return fail(C1_ERR_ARG, "fixed_block_modes[%d] = %d is outside 0..%d", b, m, b == 2 ? 3 : 2);
    d->modes[b] = m;
  }
  static const int no_tonal = getenv("C1_ALLOC_NO_TONAL") ? atoi(getenv("C1_ALLOC_NO_TONAL")) : 0;
  d->alloc_no_tonal = no_tonal;                          // experiments: 1 = always run the 52-BFU candidate first
  return C1_OK;
}

int build_table_fields(const c1_encode_options &o, C1DevEncOpts *d);

int build_encode_opts(const c1_encode_options &o, C1DevEncOpts *d) {
  struct Entry { double biased[64]; C1DevEncOpts table_part; };
  static std::mutex mu;
  static std::vector<Entry> cache;
  bool hit = false;
  {
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry &e : cache)
      if (memcmp(e.biased, o.biased_scale_factors, sizeof e.biased) == 0) { *d = e.table_part; hit = true; break; }
  }
  if (!hit) {
    const int rc = build_table_fields(o, d);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(mu);
    if (cache.size() >= 8) cache.erase(cache.begin());
    Entry e;
    memcpy(e.biased, o.biased_scale_factors, sizeof e.biased);
    e.table_part = *d;
    cache.push_back(e);
  }
  return fill_call_fields(o, d);
}

int build_table_fields(const c1_encode_options &o, C1DevEncOpts *d) {
  memset(d, 0, sizeof *d);
  for (int i = 0; i < 64; i++) {
    if (!std::isfinite(o.biased_scale_factors[i]) || o.biased_scale_factors[i] < 0)
      return fail(C1_ERR_ARG, "biased_scale_factors[%d] is not a finite non-negative number", i);
    d->biased[i] = o.biased_scale_factors[i];
  }
  {
    // log2 of the biased table as a line in the index (exact for pow(2^(s/3-21), bias)); see C1DevEncOpts
    const double l1 = std::log2(d->biased[1]), l63 = std::log2(d->biased[63]);
    const double slope = (l63 - l1) / 62.0;
    d->la_slope = (std::isfinite(slope) && std::isfinite(l1)) ? (float)slope : 0.0f;
    d->la_off = (std::isfinite(slope) && std::isfinite(l1)) ? (float)(l1 - slope) : 0.0f;
  }
  float pri[64 * 15];
  std::vector<float> uniq;
  for (int s = 1; s < 64; s++)
    for (int wl = 0; wl < 15; wl++) {
      const int b0 = wl == 0 ? 0 : wl + 1, b1 = wl + 2;
      const double ddf = wl == 0 ? 2.0 - std::ldexp(1.0, -b1) : std::ldexp(1.0, -b0) - std::ldexp(1.0, -b1);
      const double dbits = (double)(b1 - b0);
      pri[s * 15 + wl] = (float)(d->biased[s] * ddf / dbits);
      uniq.push_back(pri[s * 15 + wl]);
    }
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  for (int s = 1; s < 64; s++)
    for (int wl = 0; wl < 15; wl++) {
      const auto it = std::lower_bound(uniq.begin(), uniq.end(), pri[s * 15 + wl]);
      d->rank[s * 16 + wl] = (uint16_t)(1 + (it - uniq.begin()));
    }
  // Is the order of the ranks the order of an integer form?  (bias 1: priority = 2^(s/3-21) * {0.875 | 2^-(wl+2)}
  // -> 2s-1 for wl = 0 and 2s - 6wl - 12 for wl >= 1.)  Search small coefficients; ties must match too.
  d->rank_affine = 0;
  {
    std::vector<int> order;                       // (s, wl) pairs sorted by rank
    for (int s = 1; s < 64; s++)
      for (int wl = 0; wl < 15; wl++) order.push_back(s * 16 + wl);
    std::sort(order.begin(), order.end(), [&](int x, int y) { return d->rank[x] < d->rank[y]; });
    for (int A = 1; A <= 12 && !d->rank_affine; A++)
      for (int B = 1; B <= 36 && !d->rank_affine; B++)
        for (int C = -2 * B; C <= 0 && !d->rank_affine; C++) {
          auto key = [&](int idx) { const int s = idx >> 4, wl = idx & 15; return wl == 0 ? A * s + C : A * s - B * wl - B; };
          bool ok = true;
          int lo = key(order[0]), hi = lo;
          for (size_t i = 1; i < order.size() && ok; i++) {
            const int k0 = key(order[i - 1]), k1 = key(order[i]);
            const bool req = d->rank[order[i]] == d->rank[order[i - 1]];
            if (req ? (k1 != k0) : !(k1 > k0)) ok = false;
            lo = std::min(lo, k1); hi = std::max(hi, k1);
          }
          if (ok && hi - lo + 1 < 1023) {
            d->rank_affine = 1; d->rank_a = A; d->rank_b = B; d->rank_c = C; d->rank_off = 1 - lo;
          }
        }
  }
  return C1_OK;
}

constexpr int kTotals = 8;                             // running totals of a context (c1_ctx::d_spec_totals)
constexpr int64_t kSpecMinUnits = 64;                  // default mode: calls below this many sound units use the exact kernels only
constexpr int kListHead = 8;                           // uint32 counters in front of the speculative path's lists
constexpr int64_t kMaxChunkFrames = (int64_t)1 << 27;   // x 2 channels = 2^28 units per chunk < 2^29

struct Timing {
  hipEvent_t start, stop;
  int kind;
};
enum { K_ANALYSIS = 0, K_ALLOCATE, K_PACK, K_DECODE, K_REDO, K_KINDS };
const char *const kKindNames[K_KINDS] = {"analysis", "allocate", "pack", "decode", "redo"};

}  // namespace

struct c1_ctx {
  // Every public entry point that touches the context holds this for its whole duration: calls on one context are
  // serialised on the host (workspace, staging buffers, option cache and timings are per context), as the GPU side
  // already is by the context's stream.  Recursive because entry points call each other (batch -> device, ...).
  std::recursive_mutex mu;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  C1DevTables *d_tables = nullptr;
  C1DevEncOpts *d_opts = nullptr;
  c1_encode_options last_opts;
  bool have_opts = false;
  // workspace: two chunk-sized sets, so chunk i+1's analysis overlaps chunk i's allocation and packing
  int64_t ws_units = 0;
  float *d_coefs[2] = {nullptr, nullptr};
  uint8_t *d_side[2] = {nullptr, nullptr};
  uint8_t *d_alloc[2] = {nullptr, nullptr};
  uint8_t *d_cand[2] = {nullptr, nullptr};
  uint32_t *d_work[2] = {nullptr, nullptr};      // [0] = count, list from [4]
  // speculative binary32 path: per-unit error bounds, redo list ([0] = count, list from [4]), running totals
  float *d_eps[2] = {nullptr, nullptr};
  uint32_t *d_redo[2] = {nullptr, nullptr};
  // running totals on the device: [0] units that stayed with the speculative analysis, [1] units among them redone exactly,
  // [2] units of exact coefficients quantized in binary32, [3] units among them packed again, [4] units the speculative
  // detector decided, [5] units among them rechecked, [6] units the speculative analysis handed to the exact kernels.
  // Statistics only (c1_ctx_*_stats): no decision of the encode path reads them.
  unsigned long long *d_spec_totals = nullptr;
  int spec_mode = 1;                             // 0 exact only, 1 material-local (default), 2 always speculate
  float spec_defer = 1.0f;                       // mode 1: predicted open decisions per unit past which a run goes to the exact kernels
  bool decode_binary32 = false;                  // c1_ctx_set_decode_precision: opt-in binary32 decoder
  bool spec_tables_ok = false;
  // transient-detection workspace (allocated on first use): band samples, feature sums, block modes
  int64_t det_units = 0;
  float *d_bands[2] = {nullptr, nullptr};
  double *d_feat[2] = {nullptr, nullptr};
  uint8_t *d_modes[2] = {nullptr, nullptr};
  uint32_t *d_lists[2] = {nullptr, nullptr};
  // Tail overlap: the exact redo of a speculative chunk -- short lists, latency-bound launches -- runs on
  // s_tail while the next chunk's (or, on a context that owns its stream, the next call's) main kernels run on the
  // context's stream; the two chunks work on different halves of the workspace.  ev_main[p] / ev_tail[p]: main part /
  // tail of the chunk that last used half p.  tail_pending: a tail is in flight that the context's stream has not
  // been made to wait for yet (join_tail); every entry point but the device encode joins before it does anything.
  bool overlap = false;
  hipStream_t s_tail = nullptr;
  hipEvent_t ev_main[2] = {nullptr, nullptr}, ev_tail[2] = {nullptr, nullptr};
  bool tail_used[2] = {false, false};
  bool tail_pending = false;
  int ws_next = 0;
  int64_t chunk_frames = 0;
  bool pipeline = true;
  hipStream_t s_ana = nullptr, s_rest = nullptr;  // internal streams of the two pipeline halves
  hipEvent_t ev_in = nullptr, ev_ana[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr}, ev_end[2] = {nullptr, nullptr};
  // profiling
  int timing_depth = 0;                          // > 0 inside a multi-chunk host call: the per-chunk device calls keep the timings
  bool profiling = false;
  std::vector<Timing> timings;
  std::vector<hipEvent_t> event_pool;
  double ms[K_KINDS] = {0, 0, 0, 0, 0};
  int launches[K_KINDS] = {0, 0, 0, 0, 0};
  // scratch for host-resident calls
  void *d_io = nullptr;
  size_t d_io_bytes = 0;
  // streamed host path (pinned buffers): copy streams, two chunk-sized staging sets and their events
  hipStream_t s_up = nullptr, s_down = nullptr;
  void *d_ring = nullptr;
  size_t d_ring_bytes = 0;
  hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_run[2] = {nullptr, nullptr}, ev_down[2] = {nullptr, nullptr};
};

namespace {

#define CTX_GUARD(c)                                   \
  std::unique_lock<std::recursive_mutex> ctx_guard_;   \
  if (c) ctx_guard_ = std::unique_lock<std::recursive_mutex>((c)->mu)

// the context's stream waits for every tail in flight: from here on it sees the finished results of all earlier calls
int join_tail(c1_ctx *ctx) {
  if (!ctx->tail_pending) return C1_OK;
  for (int p = 0; p < 2; p++)
    if (ctx->tail_used[p]) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_tail[p], 0));
  ctx->tail_pending = false;
  return C1_OK;
}

int ctx_bind(c1_ctx *ctx, bool join = true) {
  if (!ctx) return fail(C1_ERR_ARG, "context is NULL");
  HIP_TRY(hipSetDevice(ctx->device));
  if (join) return join_tail(ctx);
  return C1_OK;
}

void free_workspace(c1_ctx *ctx) {
  for (int p = 0; p < 2; p++) {
    if (ctx->d_coefs[p]) (void)hipFree(ctx->d_coefs[p]);
    if (ctx->d_side[p]) (void)hipFree(ctx->d_side[p]);
    if (ctx->d_alloc[p]) (void)hipFree(ctx->d_alloc[p]);
    if (ctx->d_cand[p]) (void)hipFree(ctx->d_cand[p]);
    if (ctx->d_work[p]) (void)hipFree(ctx->d_work[p]);
    if (ctx->d_eps[p]) (void)hipFree(ctx->d_eps[p]);
    if (ctx->d_redo[p]) (void)hipFree(ctx->d_redo[p]);
    ctx->d_coefs[p] = nullptr; ctx->d_side[p] = nullptr; ctx->d_alloc[p] = nullptr; ctx->d_cand[p] = nullptr; ctx->d_work[p] = nullptr;
    ctx->d_eps[p] = nullptr; ctx->d_redo[p] = nullptr;
  }
  ctx->ws_units = 0;
  for (int p = 0; p < 2; p++) {
    if (ctx->d_bands[p]) (void)hipFree(ctx->d_bands[p]);
    if (ctx->d_feat[p]) (void)hipFree(ctx->d_feat[p]);
    if (ctx->d_modes[p]) (void)hipFree(ctx->d_modes[p]);
    if (ctx->d_lists[p]) (void)hipFree(ctx->d_lists[p]);
    ctx->d_bands[p] = nullptr; ctx->d_feat[p] = nullptr; ctx->d_modes[p] = nullptr; ctx->d_lists[p] = nullptr;
  }
  ctx->det_units = 0;
}

int ensure_detect_workspace(c1_ctx *ctx, int64_t units) {
  if (units <= ctx->det_units) return C1_OK;
  HIP_TRY(hipDeviceSynchronize());
  for (int p = 0; p < (ctx->pipeline ? 2 : 1); p++) {
    if (ctx->d_bands[p]) (void)hipFree(ctx->d_bands[p]);
    if (ctx->d_feat[p]) (void)hipFree(ctx->d_feat[p]);
    if (ctx->d_modes[p]) (void)hipFree(ctx->d_modes[p]);
    if (ctx->d_lists[p]) (void)hipFree(ctx->d_lists[p]);
    ctx->d_bands[p] = nullptr; ctx->d_feat[p] = nullptr; ctx->d_modes[p] = nullptr; ctx->d_lists[p] = nullptr;
    // one extra row of slots in front: frame -1 of the batch (c1_internal.h)
    HIP_TRY(hipMalloc(&ctx->d_bands[p], (size_t)(units + C1_MAX_CHANNELS) * 512 * sizeof(float)));
    HIP_TRY(hipMalloc(&ctx->d_feat[p], (size_t)(units + C1_MAX_CHANNELS) * kFeatureWsDoubles * sizeof(double)));
    HIP_TRY(hipMalloc(&ctx->d_modes[p], (size_t)units));
    HIP_TRY(hipMalloc(&ctx->d_lists[p], ((size_t)units * 3 + 4) * sizeof(uint32_t)));
  }
  ctx->det_units = units;
  return C1_OK;
}

int ensure_workspace(c1_ctx *ctx, int64_t units) {
  if (units <= ctx->ws_units) return C1_OK;
  HIP_TRY(hipDeviceSynchronize());
  free_workspace(ctx);
  ctx->tail_used[0] = ctx->tail_used[1] = false;
  ctx->tail_pending = false;
  for (int p = 0; p < ((ctx->pipeline || ctx->overlap) ? 2 : 1); p++) {
    HIP_TRY(hipMalloc(&ctx->d_coefs[p], (size_t)units * 512 * sizeof(float)));
    HIP_TRY(hipMalloc(&ctx->d_side[p], (size_t)units * kSideBytes));
    HIP_TRY(hipMalloc(&ctx->d_alloc[p], (size_t)units * kAllocBytes));
    HIP_TRY(hipMalloc(&ctx->d_cand[p], (size_t)units * kCandidateBytes));
    HIP_TRY(hipMalloc(&ctx->d_work[p], ((size_t)units * 8 + 4) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&ctx->d_eps[p], (size_t)units * kEpsFloats * sizeof(float)));
    HIP_TRY(hipMalloc(&ctx->d_redo[p], ((size_t)units * 4 + kListHead + 64) * sizeof(uint32_t)));   // counts, then four lists (bind_lists) + slack for the masks of tiny batches
  }
  ctx->ws_units = units;
  return C1_OK;
}

// Frames per chunk of an encode call.  Every kernel of the chain ends in a tail of draining workgroups and the allocation's
// later rounds are bound by the latency of one heap run whatever their list's length, so a batch is best kept in ONE chunk
// (BASELINE configs[3]'s share of 12.5 M stereo frames: 130.7 M frames/s in chunks of 1 M, 138.9 M in chunks of 4 M, 144.7 M
// in one) -- the workspace costs 2.6 KB per unit (4.8 KB with transient detection) and this part has 288 GB.  The configured
// chunk (C1_CHUNK_FRAMES, default 2^24 frames) is cut down to what the device can hold right now next to the caller's
// buffers; a workspace that is already large enough is used as it is.
constexpr size_t kWsBytesPerUnit = 512 * sizeof(float) + kSideBytes + kAllocBytes + kCandidateBytes + 8 * sizeof(uint32_t) + kEpsFloats * sizeof(float) + 4 * sizeof(uint32_t);
constexpr size_t kDetectWsBytesPerUnit = 512 * sizeof(float) + kFeatureWsDoubles * sizeof(double) + 1 + 3 * sizeof(uint32_t);
int64_t chunk_for_call(c1_ctx *ctx, int64_t frames, int channels, bool detect) {
  const int64_t want = std::min(frames, ctx->chunk_frames);
  if (want * channels <= ctx->ws_units && (!detect || want * channels <= ctx->det_units)) return want;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return std::min<int64_t>(want, 1048576); }
  const size_t sets = (ctx->pipeline || ctx->overlap) ? 2 : 1, det_sets = ctx->pipeline ? 2 : 1;
  size_t avail = free_b + (size_t)ctx->ws_units * kWsBytesPerUnit * sets;          // the present workspace is freed before it grows
  size_t per_unit = kWsBytesPerUnit * sets;
  if (detect) { avail += (size_t)ctx->det_units * kDetectWsBytesPerUnit * det_sets; per_unit += kDetectWsBytesPerUnit * det_sets; }
  const int64_t fit = (int64_t)((double)avail * 0.9 / (double)per_unit) / channels;
  return std::max<int64_t>(16, std::min(want, fit));
}

// the lists of the speculative path in workspace half p: counts at [0, kListHead), then the redo, reallocation,
// re-analysis and deferred-run lists, ws_units entries each
void bind_lists(c1_ctx *ctx, int p, C1EncodeLaunch *L) {
  uint32_t *base = ctx->d_redo[p];
  const size_t u = (size_t)ctx->ws_units;
  L->redo_count = base;
  L->realloc_count = base + 1;
  L->reana_count = base + 2;
  L->redo_list = base + kListHead;
  L->realloc_list = base + kListHead + u;
  L->reana_list = base + kListHead + 2 * u;
}
void bind_defer(c1_ctx *ctx, int p, C1EncodeLaunch *L) {
  L->defer_list = ctx->d_redo[p] + kListHead + 3 * (size_t)ctx->ws_units;   // one slot per run and channel (<= units)
}

int ensure_io(c1_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->d_io_bytes) return C1_OK;
  if (ctx->d_io) hipFree(ctx->d_io);
  ctx->d_io = nullptr; ctx->d_io_bytes = 0;
  HIP_TRY(hipMalloc(&ctx->d_io, bytes));
  ctx->d_io_bytes = bytes;
  return C1_OK;
}

int upload_opts(c1_ctx *ctx, const c1_encode_options *opts) {
  if (!opts) return fail(C1_ERR_ARG, "options are NULL");
  if (ctx->have_opts && memcmp(&ctx->last_opts, opts, sizeof *opts) == 0) return C1_OK;
  C1DevEncOpts h;
  const int rc = build_encode_opts(*opts, &h);
  if (rc) return rc;
  // the previous options may still be in use by kernels queued on the stream (or on the tail stream)
  int jr = join_tail(ctx);
  if (jr) return jr;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipMemcpy(ctx->d_opts, &h, sizeof h, hipMemcpyHostToDevice));
  ctx->last_opts = *opts;
  ctx->have_opts = true;
  return C1_OK;
}

hipEvent_t take_event(c1_ctx *ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  // timing marks only: nobody reads memory behind them, so no system-scope release (a cache write-back per record, ~5 us of
  // idle stream each; tools/prof_cost.py)
  hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
  return e;
}
struct ScopedTiming {
  c1_ctx *ctx;
  Timing t;
  bool on;
  hipStream_t stream;
  ScopedTiming(c1_ctx *c, int kind, hipStream_t s = nullptr) : ctx(c), on(c->profiling), stream(s ? s : c->stream) {
    if (!on) return;
    t.kind = kind;
    t.start = take_event(ctx);
    t.stop = take_event(ctx);
    (void)hipEventRecord(t.start, stream);
  }
  ~ScopedTiming() {
    if (!on) return;
    (void)hipEventRecord(t.stop, stream);
    ctx->timings.push_back(t);
  }
};
void reset_timings(c1_ctx *ctx) {
  for (auto &t : ctx->timings) { ctx->event_pool.push_back(t.start); ctx->event_pool.push_back(t.stop); }
  ctx->timings.clear();
  for (int k = 0; k < K_KINDS; k++) { ctx->ms[k] = 0; ctx->launches[k] = 0; }
}
int collect_timings(c1_ctx *ctx) {
  if (ctx->timings.empty()) return C1_OK;
  int jr = join_tail(ctx);
  if (jr) return jr;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (auto &t : ctx->timings) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, t.start, t.stop));
    ctx->ms[t.kind] += ms;
    ctx->launches[t.kind]++;
    ctx->event_pool.push_back(t.start);
    ctx->event_pool.push_back(t.stop);
  }
  ctx->timings.clear();
  return C1_OK;
}

int check_channels(int channels) {
  if (channels != 1 && channels != 2) return fail(C1_ERR_ARG, "channels must be 1 or 2, got %d", channels);
  return C1_OK;
}

// lazy: the call may return with its last tail still unjoined (c1_encode_device on a context that owns its stream: nobody
// else can enqueue on that stream, and every other entry point joins first)
int encode_device_impl(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                       const c1_encode_options *opts, uint8_t *units, float *bands, float *coefs_tap,
                       uint8_t *side_tap, uint8_t *alloc_tap, bool lazy = false) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx, false);
  if (rc) return rc;
  if ((rc = check_channels(channels))) return rc;
  if (frames < 0) return fail(C1_ERR_ARG, "frames must be >= 0");
  if (halo_frames < 0 || halo_frames > 2) return fail(C1_ERR_ARG, "halo_frames must be 0, 1 or 2");
  if (!pcm) return fail(C1_ERR_ARG, "pcm is NULL");
  for (int c = 0; c < channels; c++) {
    if (!pcm[c] && frames > 0) return fail(C1_ERR_ARG, "pcm[%d] is NULL", c);
    if ((uintptr_t)pcm[c] & 15) return fail(C1_ERR_ARG, "pcm[%d] must be 16-byte aligned on the device", c);
  }
  if ((rc = upload_opts(ctx, opts))) return rc;
  if (ctx->profiling && ctx->timing_depth == 0) reset_timings(ctx);
  if (frames == 0) return C1_OK;
  const bool detect = opts->fixed_block_modes[0] < 0;
  const bool taps = coefs_tap || side_tap || alloc_tap;
  const int64_t chunk = taps ? ctx->chunk_frames : chunk_for_call(ctx, frames, channels, detect);
  if ((rc = ensure_workspace(ctx, (taps ? frames : std::min(frames, chunk)) * channels))) return rc;
  if (detect && (rc = ensure_detect_workspace(ctx, (taps ? frames : std::min(frames, chunk)) * channels))) return rc;
  if (taps && (!coefs_tap || !side_tap || !alloc_tap)) return fail(C1_ERR_ARG, "coefs, side and alloc taps must be given together");
  if (taps && frames > kMaxChunkFrames) return fail(C1_ERR_ARG, "stage taps are not chunked: at most %lld frames per call", (long long)kMaxChunkFrames);
  // Two-stage software pipeline over chunks: the analysis of chunk i+1 (fp64-VALU bound) runs on one
  // stream while allocation + packing of chunk i (latency bound) run on another, each chunk on its own
  // half of the workspace.  Everything is ordered after the caller's stream and joined back into it.
  const bool all_long_modes = !detect && opts->fixed_block_modes[0] == 0 && opts->fixed_block_modes[1] == 0 &&
                              opts->fixed_block_modes[2] == 0 && !getenv("C1_NO_FAST_LONG");
  const bool all_short_modes = !detect && opts->fixed_block_modes[0] != 0 && opts->fixed_block_modes[1] != 0 &&
                               opts->fixed_block_modes[2] != 0;
  bool speculate = (all_long_modes || all_short_modes) && !taps && units && ctx->spec_tables_ok && ctx->spec_mode != 0;
  bool quantize32 = !taps && units && ctx->spec_tables_ok && ctx->spec_mode != 0;   // exact coefficients, binary32 quantization with the guard (below)
  static const bool det_spec_env_off = getenv("C1_DETECT_SPEC") && atoi(getenv("C1_DETECT_SPEC")) == 0;   // experiments: exact detector, the rest as usual
  bool detect_spec = detect && !taps && ctx->spec_tables_ok && ctx->spec_mode != 0 && !det_spec_env_off;   // binary32 transient detector with a score interval
  // A call of a few frames (a frame closure, a short streaming push) is bound by the number of launches behind it, and
  // every speculative shortcut adds some (the redo chain, the recheck, the second packing pass): in the default mode such
  // calls take the exact kernels (one mono frame: 159 against 185 us, tools/latency_probe.py).  A rule on the size of
  // this call alone; mode 2 still speculates on anything.
  if (ctx->spec_mode == 1 && frames * channels < kSpecMinUnits) speculate = quantize32 = detect_spec = false;
  // Both shortcuts of the exact paths (binary32 quantization of exact coefficients, binary32 transient detector) are
  // taken whenever speculation is on; nothing is carried from call to call.  What they hand back to the exact arithmetic
  // is listed unit by unit inside the call: 0.07 % (noise) to 4 % (stationary partials) of the units packed again, 0.02 to
  // 0.3 % rechecked (tools/adaptive_probe.py) -- round 2's per-context switches at 10 % and 20 % never fired on any
  // material tried and made throughput depend on what a context had encoded before.  c1_ctx_set_speculation(ctx, 0)
  // turns every shortcut off for a stream that is known to defeat them; the bytes are the same either way.
  const bool overlap = speculate && ctx->overlap && ctx->s_tail != nullptr;
  if (!overlap && (rc = join_tail(ctx))) return rc;       // every other path works on the context's stream alone
  const bool piped = ctx->pipeline && !taps && frames > chunk && !speculate;
  hipStream_t sA = piped ? ctx->s_ana : ctx->stream, sB = piped ? ctx->s_rest : ctx->stream;
  if (piped) {
    HIP_TRY(hipEventRecord(ctx->ev_in, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(sA, ctx->ev_in, 0));
    HIP_TRY(hipStreamWaitEvent(sB, ctx->ev_in, 0));
  }
  int64_t index = 0;
  for (int64_t f0 = 0, n = 0; f0 < frames; f0 += n, ++index) {
    n = taps ? frames : std::min(chunk, frames - f0);
    int p = piped ? (int)(index & 1) : 0;
    if (overlap) { p = ctx->ws_next; ctx->ws_next ^= 1; }
    C1EncodeLaunch L;
    memset(&L, 0, sizeof L);
    for (int c = 0; c < channels; c++) L.pcm[c] = pcm[c] + f0 * 512;
    L.channels = channels;
    L.frames = n;
    L.halo_frames = (int)std::min<int64_t>(2, f0 + halo_frames);
    L.tables = ctx->d_tables;
    L.opts = ctx->d_opts;
    L.coefs = taps ? coefs_tap : ctx->d_coefs[p];
    L.side = taps ? side_tap : ctx->d_side[p];
    L.alloc = taps ? alloc_tap : ctx->d_alloc[p];
    L.cand = ctx->d_cand[p];
    L.work_count = ctx->d_work[p];
    L.work_list = ctx->d_work[p] + 4;
    L.sel_list = ctx->d_work[p] + 4 + (size_t)ctx->ws_units * 7;
    L.bands = bands ? bands + f0 * channels * 512 : nullptr;
    L.units = units ? units + f0 * channels * C1_UNIT_BYTES : nullptr;
    const bool all_long = all_long_modes;
    if (piped && index >= 2) HIP_TRY(hipStreamWaitEvent(sA, ctx->ev_free[p], 0));   // workspace half p is free again
    if (speculate) {
      // Speculative pass in binary32 (c1_k_spec.hip): coefficients with a proven error bound; allocation works on the
      // scale-factor indices; the packing kernel accepts a unit only when every decision is certain within the bound
      // and lists the others.  Then the exact kernels redo the listed units in place.
      // Material-local: every 16 frames of a run the speculative kernel estimates from the scale-factor indices and its
      // bound how many decisions of a unit will stay open; past spec_defer it hands the rest of the run to the exact
      // kernels (run list), whose units are then quantized in binary32 behind a bound of zero like the exact paths'.
      L.eps = ctx->d_eps[p];
      bind_lists(ctx, p, &L);
      if (ctx->spec_mode == 1) { bind_defer(ctx, p, &L); L.spec_defer = ctx->spec_defer; }
      // masks of the units with an open scale factor: behind the deferred runs' slots (one per run and channel, at most a
      // quarter of the units: a run is at least 4 frames), two words per run; only with runs of at most 64 frames
      if (c1k_pick_run(n, channels, 0) <= 64 && !getenv("C1_NO_SF_PREPASS"))
        L.open_masks = reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(ctx->d_redo[p] + kListHead + 3 * (size_t)ctx->ws_units + (size_t)ctx->ws_units / 4 + 4) + 7) & ~(uintptr_t)7);
      // this half of the workspace is free once the tail of the chunk that used it last is through
      if (overlap && ctx->tail_used[p]) HIP_TRY(hipStreamWaitEvent(sA, ctx->ev_tail[p], 0));
      HIP_TRY(hipMemsetAsync(ctx->d_redo[p], 0, kListHead * sizeof(uint32_t), sA));
      {
        ScopedTiming t(ctx, K_ANALYSIS, sA);
        c1k_launch_analysis_spec(L, all_short_modes, sA);
        if (L.defer_list) {
          C1EncodeLaunch D = L;
          uint32_t *dense = ctx->d_redo[p] + kListHead + 2 * (size_t)ctx->ws_units;   // the re-analysis list's space: that list is filled
          c1k_launch_defer_compact(L, dense, ctx->d_redo[p] + 3, sA);                 // by the packing kernel, after this pass is done
          D.unit_list = dense;
          D.unit_count = ctx->d_redo[p] + 3;
          D.list_runs = 1;
          D.defer_list = nullptr;
          if (all_long_modes) c1k_launch_analysis_long(D, sA); else c1k_launch_analysis(D, false, sA);
        }
      }
      if (L.open_masks) {
        ScopedTiming t(ctx, K_REDO, sA);                       // exact work on uncertain units: timed with the redo, wherever it runs
        {
          // the units whose scale-factor guard stayed open (1.8 % of white noise) are re-analysed exactly HERE, in front of the
          // allocation: it then sees the reference's indices the first time, and the redo behind the packing pass has no
          // allocation chain of its own (five launches, one of them a lone heap run long: 0.13 ms per 2 M units)
          C1EncodeLaunch X = L;
          uint32_t *open_list = ctx->d_redo[p] + kListHead + 2 * (size_t)ctx->ws_units;   // the re-analysis list's space again (the
          c1k_launch_open_compact(L, open_list, ctx->d_redo[p] + 5, sA);                  // deferred runs' list above is consumed)
          X.unit_list = open_list;
          X.unit_count = ctx->d_redo[p] + 5;
          X.list_runs = 0;
          X.list_zero_eps = 1;
          X.defer_list = nullptr;
          X.open_masks = nullptr;
          if (all_long_modes) c1k_launch_analysis_long(X, sA); else c1k_launch_analysis(X, false, sA);
        }
      }
      { ScopedTiming t(ctx, K_ALLOCATE, sA); c1k_launch_allocate(L, sA); }
      // The previous chunk's tail repacks units of ITS output range; when a caller reuses one output buffer call after
      // call that range is this chunk's: the tail's stores must have landed before this chunk packs (in practice it
      // finished long ago: it has had the whole analysis and allocation of this chunk to run beside)
      if (overlap && ctx->tail_used[p ^ 1]) HIP_TRY(hipStreamWaitEvent(sA, ctx->ev_tail[p ^ 1], 0));
      { ScopedTiming t(ctx, K_PACK, sA); c1k_launch_pack_spec(L, all_long_modes, sA); }
      hipStream_t sT = sA;
      if (overlap) {
        sT = ctx->s_tail;
        HIP_TRY(hipEventRecord(ctx->ev_main[p], sA));
        HIP_TRY(hipStreamWaitEvent(sT, ctx->ev_main[p], 0));
      }
      {
        ScopedTiming t(ctx, K_REDO, sT);
        hipStream_t sA = sT;                                   // the exact redo of the listed units: on the tail stream
        C1EncodeLaunch R = L;
        R.defer_list = nullptr;
        R.unit_list = L.reana_list;
        R.unit_count = L.reana_count;
        if (all_long_modes) c1k_launch_analysis_long(R, sA); else c1k_launch_analysis(R, false, sA);
        if (!L.open_masks) {                                   // with the pre-pass above no unit reaches the packing pass with an open scale factor
          C1EncodeLaunch A = R;
          A.unit_list = L.realloc_list;
          A.unit_count = L.realloc_count;
          c1k_launch_allocate(A, sA);
        }
        R.unit_list = L.redo_list;
        R.unit_count = L.redo_count;
        c1k_launch_pack(R, all_long_modes, sA);
        c1k_launch_spec_totals(ctx->d_spec_totals, (uint64_t)(n * channels), ctx->d_redo[p], 0, sA);
      }
      if (overlap) {
        HIP_TRY(hipEventRecord(ctx->ev_tail[p], sT));
        ctx->tail_used[p] = true;
        ctx->tail_pending = true;
      }
      continue;
    }
    {
      ScopedTiming t(ctx, K_ANALYSIS, sA);
      if (all_long) c1k_launch_analysis_long(L, sA);
      else if (detect) {
        c1k_launch_detect(L, ctx->d_bands[p], ctx->d_feat[p], ctx->d_modes[p], ctx->d_lists[p], detect_spec, nullptr, sA);
        if (detect_spec) c1k_launch_spec_totals(ctx->d_spec_totals, (uint64_t)(n * channels), ctx->d_lists[p] + 2, 2, sA);
        if (L.bands) HIP_TRY(hipMemcpyAsync(L.bands, ctx->d_bands[p] + (size_t)channels * 512, (size_t)n * channels * 512 * sizeof(float),
                                            hipMemcpyDeviceToDevice, sA));
      } else c1k_launch_analysis(L, false, sA);
    }
    if (piped) {
      HIP_TRY(hipEventRecord(ctx->ev_ana[p], sA));
      HIP_TRY(hipStreamWaitEvent(sB, ctx->ev_ana[p], 0));
    }
    { ScopedTiming t(ctx, K_ALLOCATE, sB); c1k_launch_allocate(L, sB); }
    if (L.units && quantize32) {
      // The coefficients are the reference's, and still the quantization need not be done in binary64: the packing
      // kernel of the speculative path with a bound of zero forms |x| norm + 0.5 in binary32 and accepts a mantissa only
      // when no value within the roundings of that (norm32 against norm, the fused operation, the reference's own
      // two) truncates differently; the few units it lists (and anything not finite) are packed again by the exact kernel.
      ScopedTiming t(ctx, K_PACK, sB);
      L.eps = ctx->d_eps[p];
      bind_lists(ctx, p, &L);                                  // the reallocation and re-analysis lists stay empty: with bounds of zero no
                                                               // scale-factor index is open and every coefficient is the exact kernels'
      HIP_TRY(hipMemsetAsync(L.eps, 0, (size_t)n * channels * kEpsFloats * sizeof(float), sB));
      HIP_TRY(hipMemsetAsync(ctx->d_redo[p], 0, kListHead * sizeof(uint32_t), sB));
      c1k_launch_pack_spec(L, all_long, sB);
      C1EncodeLaunch R = L;
      R.unit_list = L.redo_list;
      R.unit_count = L.redo_count;
      c1k_launch_pack(R, all_long, sB);
      c1k_launch_spec_totals(ctx->d_spec_totals, (uint64_t)(n * channels), L.redo_count, 1, sB);
    } else if (L.units) { ScopedTiming t(ctx, K_PACK, sB); c1k_launch_pack(L, all_long, sB); }
    if (piped) HIP_TRY(hipEventRecord(ctx->ev_free[p], sB));
  }
  if (piped) {
    HIP_TRY(hipEventRecord(ctx->ev_end[0], sA));
    HIP_TRY(hipEventRecord(ctx->ev_end[1], sB));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_end[0], 0));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_end[1], 0));
  }
  if (overlap && !lazy && (rc = join_tail(ctx))) return rc;
  HIP_TRY(hipGetLastError());
  return C1_OK;
}

// xorshift32 as a GF(2) linear map: column form of T^n, used to jump the generator
struct XsMatrix {
  uint32_t col[32];
  uint32_t apply(uint32_t s) const {
    uint32_t r = 0;
    for (int b = 0; b < 32; b++) if ((s >> b) & 1u) r ^= col[b];
    return r;
  }
};
uint32_t xs_step(uint32_t s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
XsMatrix xs_power(uint64_t n) {
  XsMatrix result, base;
  for (int b = 0; b < 32; b++) { result.col[b] = 1u << b; base.col[b] = xs_step(1u << b); }
  while (n) {
    if (n & 1) { XsMatrix r; for (int b = 0; b < 32; b++) r.col[b] = base.apply(result.col[b]); result = r; }
    XsMatrix sq;
    for (int b = 0; b < 32; b++) sq.col[b] = base.apply(base.col[b]);
    base = sq;
    n >>= 1;
  }
  return result;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
namespace {
struct DeviceScratch {       // a few small device buffers for one call, freed on every path
  std::vector<void *> ptrs;
  ~DeviceScratch() { for (void *p : ptrs) (void)hipFree(p); }
  template <class T> int alloc(T **out, size_t count) {
    void *p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return fail(C1_ERR_HIP, "device allocation of %zu bytes failed", count * sizeof(T));
    ptrs.push_back(p);
    *out = static_cast<T *>(p);
    return C1_OK;
  }
};
}  // namespace

extern "C" {

int c1_abi_version(void) { return C1_ABI_VERSION; }
const char *c1_last_error(void) { return g_error.c_str(); }

int c1_device_count(int *count) {
  if (!count) return fail(C1_ERR_ARG, "count is NULL");
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(C1_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return C1_OK;
}

int c1_get_default_tables(c1_tables *out) {
  if (!out) return fail(C1_ERR_ARG, "out is NULL");
  default_tables(out);
  return C1_OK;
}

int c1_set_tables(const c1_tables *tables) {
  std::lock_guard<std::mutex> lock(g_tables_mutex);
  if (!tables) { g_tables_custom = false; g_tables_gen++; return C1_OK; }
  const double *p = reinterpret_cast<const double *>(tables);
  for (size_t i = 0; i < sizeof(c1_tables) / sizeof(double); i++)
    if (!std::isfinite(p[i])) return fail(C1_ERR_ARG, "table entry %zu is not finite", i);
  g_tables = *tables;
  g_tables_custom = true;
  g_tables_gen++;
  return C1_OK;
}

int c1_table_fast_paths(int *scale_factor_bits, int *dequant_reciprocal) {
  c1_tables t;
  {
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    if (g_tables_custom) t = g_tables; else default_tables(&t);
  }
  std::unique_ptr<C1DevTables> d(new C1DevTables);
  build_device_tables(t, d.get());
  if (scale_factor_bits) *scale_factor_bits = d->sf_fast;
  if (dequant_reciprocal) *dequant_reciprocal = d->dq_fast + d->dq_step;
  return C1_OK;
}

int c1_default_encode_options(c1_encode_options *out) {
  if (!out) return fail(C1_ERR_ARG, "out is NULL");
  memset(out, 0, sizeof *out);
  c1_tables t;
  {
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    if (g_tables_custom) t = g_tables; else default_tables(&t);
  }
  memcpy(out->biased_scale_factors, t.scale_factors, sizeof out->biased_scale_factors);
  out->transient_threshold = 1.0;
  out->fixed_block_modes[0] = out->fixed_block_modes[1] = out->fixed_block_modes[2] = -1;
  return C1_OK;
}

int c1_ctx_create(int device, void *hip_stream, c1_ctx **out) {
  if (!out) return fail(C1_ERR_ARG, "out is NULL");
  *out = nullptr;
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    return fail(C1_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device < 0 || device >= n) return fail(C1_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
  HIP_TRY(hipSetDevice(device));
  c1_ctx *ctx = new c1_ctx();
  ctx->device = device;
  if (hip_stream) ctx->stream = (hipStream_t)hip_stream;
  else {
    const hipError_t se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete ctx; return fail(C1_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(se)); }
    ctx->own_stream = true;
  }
  c1_tables t;
  {
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    if (g_tables_custom) t = g_tables; else default_tables(&t);
  }
  C1DevTables *h = new C1DevTables();
  build_device_tables(t, h);
  hipError_t me = hipMalloc(&ctx->d_tables, sizeof(C1DevTables));
  if (me == hipSuccess) me = hipMalloc(&ctx->d_opts, sizeof(C1DevEncOpts));
  if (me == hipSuccess) me = hipMemcpy(ctx->d_tables, h, sizeof *h, hipMemcpyHostToDevice);
  if (me == hipSuccess) me = hipMalloc(&ctx->d_spec_totals, kTotals * sizeof(unsigned long long));
  if (me == hipSuccess) me = hipMemset(ctx->d_spec_totals, 0, kTotals * sizeof(unsigned long long));
  ctx->spec_tables_ok = h->spec_ok != 0;
  {
    const char *sp = getenv("C1_SPEC");       // 0 exact only, 1 material-local (default), 2 always speculate
    ctx->spec_mode = sp ? atoi(sp) : 1;
    if (ctx->spec_mode < 0 || ctx->spec_mode > 2) ctx->spec_mode = 1;
    const char *sd = getenv("C1_SPEC_DEFER");   // experiments: the predictor's threshold (open decisions per unit)
    if (sd && atof(sd) > 0) ctx->spec_defer = (float)atof(sd);
  }
  delete h;
  if (me != hipSuccess) { c1_ctx_destroy(ctx); return fail(C1_ERR_HIP, "table upload: %s", hipGetErrorString(me)); }
  {
    hipError_t pe = hipStreamCreateWithFlags(&ctx->s_ana, hipStreamNonBlocking);
    if (pe == hipSuccess) pe = hipStreamCreateWithFlags(&ctx->s_rest, hipStreamNonBlocking);
    if (pe == hipSuccess) pe = hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming);
    for (int p = 0; p < 2 && pe == hipSuccess; p++) {
      pe = hipEventCreateWithFlags(&ctx->ev_ana[p], hipEventDisableTiming);
      if (pe == hipSuccess) pe = hipEventCreateWithFlags(&ctx->ev_free[p], hipEventDisableTiming);
      if (pe == hipSuccess) pe = hipEventCreateWithFlags(&ctx->ev_end[p], hipEventDisableTiming);
    }
    if (pe != hipSuccess) { c1_ctx_destroy(ctx); return fail(C1_ERR_HIP, "pipeline streams: %s", hipGetErrorString(pe)); }
    if (pe == hipSuccess) pe = hipStreamCreateWithFlags(&ctx->s_tail, hipStreamNonBlocking);
    for (int p = 0; p < 2 && pe == hipSuccess; p++) {
      pe = hipEventCreateWithFlags(&ctx->ev_main[p], hipEventDisableTiming);
      if (pe == hipSuccess) pe = hipEventCreateWithFlags(&ctx->ev_tail[p], hipEventDisableTiming);
    }
    if (pe != hipSuccess) { c1_ctx_destroy(ctx); return fail(C1_ERR_HIP, "tail stream: %s", hipGetErrorString(pe)); }
    // C1_OVERLAP=1: the exact redo of a chunk on the tail stream, beside the next chunk's (or call's) analysis.  Off by default:
    // measured +1.3 % (white noise) to +2.7 % (mixed corpus) -- the redo is mostly real work that the analysis beside it pays
    // for -- at the price of per-kernel times that include each other
    const char *ov = getenv("C1_OVERLAP");
    ctx->overlap = ov ? atoi(ov) != 0 : false;
    const char *pl = getenv("C1_PIPELINE");
    ctx->pipeline = pl ? atoi(pl) != 0 : false;   // measured: no gain while one kernel's grid already owns every CU's LDS
  }
  const char *env = getenv("C1_CHUNK_FRAMES");
  ctx->chunk_frames = env ? atoll(env) : (int64_t)1 << 24;   // chunk_for_call() cuts it down to what fits
  if (ctx->chunk_frames < 16) ctx->chunk_frames = 16;
  // the allocation work list packs (unit << 3 | candidate) into 32 bits and unit lists are 32-bit: a chunk holds
  // fewer than 2^29 units, with room to spare
  if (ctx->chunk_frames > kMaxChunkFrames) ctx->chunk_frames = kMaxChunkFrames;
  *out = ctx;
  return C1_OK;
}

int c1_ctx_destroy(c1_ctx *ctx) {
  if (!ctx) return C1_OK;
  hipSetDevice(ctx->device);
  if (ctx->s_tail) hipStreamSynchronize(ctx->s_tail);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  for (auto &t : ctx->timings) { hipEventDestroy(t.start); hipEventDestroy(t.stop); }
  for (auto e : ctx->event_pool) hipEventDestroy(e);
  if (ctx->d_tables) hipFree(ctx->d_tables);
  if (ctx->d_opts) hipFree(ctx->d_opts);
  if (ctx->d_spec_totals) hipFree(ctx->d_spec_totals);
  (void)hipDeviceSynchronize();
  free_workspace(ctx);
  if (ctx->s_tail) (void)hipStreamDestroy(ctx->s_tail);
  for (int p = 0; p < 2; p++) {
    if (ctx->ev_main[p]) (void)hipEventDestroy(ctx->ev_main[p]);
    if (ctx->ev_tail[p]) (void)hipEventDestroy(ctx->ev_tail[p]);
  }
  if (ctx->s_ana) (void)hipStreamDestroy(ctx->s_ana);
  if (ctx->s_rest) (void)hipStreamDestroy(ctx->s_rest);
  if (ctx->ev_in) (void)hipEventDestroy(ctx->ev_in);
  for (int p = 0; p < 2; p++) {
    if (ctx->ev_ana[p]) (void)hipEventDestroy(ctx->ev_ana[p]);
    if (ctx->ev_free[p]) (void)hipEventDestroy(ctx->ev_free[p]);
    if (ctx->ev_end[p]) (void)hipEventDestroy(ctx->ev_end[p]);
  }
  if (ctx->d_io) hipFree(ctx->d_io);
  if (ctx->d_ring) (void)hipFree(ctx->d_ring);
  if (ctx->s_up) (void)hipStreamDestroy(ctx->s_up);
  if (ctx->s_down) (void)hipStreamDestroy(ctx->s_down);
  for (int p = 0; p < 2; p++) {
    if (ctx->ev_up[p]) (void)hipEventDestroy(ctx->ev_up[p]);
    if (ctx->ev_run[p]) (void)hipEventDestroy(ctx->ev_run[p]);
    if (ctx->ev_down[p]) (void)hipEventDestroy(ctx->ev_down[p]);
  }
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return C1_OK;
}

int c1_ctx_synchronize(c1_ctx *ctx) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return C1_OK;
}

int c1_ctx_set_profiling(c1_ctx *ctx, int enabled) {
  if (!ctx) return fail(C1_ERR_ARG, "context is NULL");
  CTX_GUARD(ctx);
  ctx->profiling = enabled != 0;
  return C1_OK;
}

int c1_ctx_set_speculation(c1_ctx *ctx, int mode) {
  if (!ctx) return fail(C1_ERR_ARG, "context is NULL");
  if (mode < 0 || mode > 2) return fail(C1_ERR_ARG, "speculation mode must be 0, 1 or 2, got %d", mode);
  CTX_GUARD(ctx);
  ctx->spec_mode = mode;
  return C1_OK;
}

int c1_ctx_set_decode_precision(c1_ctx *ctx, int binary32) {
  if (!ctx) return fail(C1_ERR_ARG, "context is NULL");
  CTX_GUARD(ctx);
  ctx->decode_binary32 = binary32 != 0;
  return C1_OK;
}

int c1_ctx_speculation_stats(c1_ctx *ctx, uint64_t *units, uint64_t *redone, int reset) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  unsigned long long tot[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(tot, ctx->d_spec_totals, sizeof tot, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (units) *units = tot[0];
  if (redone) *redone = tot[1];
  if (reset) {
    HIP_TRY(hipMemsetAsync(ctx->d_spec_totals, 0, kTotals * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  return C1_OK;
}

int c1_ctx_speculation_deferred(c1_ctx *ctx, uint64_t *units) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  unsigned long long tot[kTotals];
  HIP_TRY(hipMemcpyAsync(tot, ctx->d_spec_totals, sizeof tot, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (units) *units = tot[6];
  return C1_OK;
}

int c1_ctx_quantization_stats(c1_ctx *ctx, uint64_t *units, uint64_t *repacked) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  unsigned long long tot[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(tot, ctx->d_spec_totals, sizeof tot, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (units) *units = tot[2];
  if (repacked) *repacked = tot[3];
  return C1_OK;
}

int c1_ctx_detection_stats(c1_ctx *ctx, uint64_t *units, uint64_t *rechecked) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  unsigned long long tot[6] = {0, 0, 0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(tot, ctx->d_spec_totals, sizeof tot, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (units) *units = tot[4];
  if (rechecked) *rechecked = tot[5];
  return C1_OK;
}

int c1_ctx_kernel_ms(c1_ctx *ctx, const char *name, double *ms, int *launches) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (!name || !ms) return fail(C1_ERR_ARG, "name or ms is NULL");
  if ((rc = collect_timings(ctx))) return rc;
  double total = 0;
  int count = 0;
  for (int k = 0; k < K_KINDS; k++) {
    if (!strcmp(name, kKindNames[k])) {
      *ms = ctx->ms[k];
      if (launches) *launches = ctx->launches[k];
      return C1_OK;
    }
    total += ctx->ms[k];
    count += ctx->launches[k];
  }
  if (!strcmp(name, "total")) {
    *ms = total;
    if (launches) *launches = count;
    return C1_OK;
  }
  return fail(C1_ERR_ARG, "unknown kernel name '%s'", name);
}

int c1_encode_device(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                     const c1_encode_options *opts, uint8_t *units) {
  if (!units && frames > 0) return fail(C1_ERR_ARG, "units is NULL");
  // On a context that owns its stream the last chunk's exact redo may still be running on the tail stream when the call
  // returns: it overlaps the next call's analysis.  c1_ctx_synchronize and every other entry point wait for it.
  return encode_device_impl(ctx, pcm, channels, frames, halo_frames, opts, units, nullptr, nullptr, nullptr, nullptr, ctx && ctx->own_stream);
}
// for the host-resident entry points below: ordered like any other work on the context's stream
static int encode_device_joined(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                                const c1_encode_options *opts, uint8_t *units) {
  if (!units && frames > 0) return fail(C1_ERR_ARG, "units is NULL");
  return encode_device_impl(ctx, pcm, channels, frames, halo_frames, opts, units, nullptr, nullptr, nullptr, nullptr, false);
}

int c1_encode_stages_device(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                            const c1_encode_options *opts, float *bands, float *coefs, uint8_t *side,
                            uint8_t *alloc) {
  return encode_device_impl(ctx, pcm, channels, frames, halo_frames, opts, nullptr, bands, coefs, side, alloc);
}

int c1_detect_stages_device(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                            const c1_encode_options *opts, float *mags, uint8_t *modes) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if ((rc = check_channels(channels))) return rc;
  if (frames < 0 || halo_frames < 0 || halo_frames > 2) return fail(C1_ERR_ARG, "bad frames / halo_frames");
  if (!pcm || !opts) return fail(C1_ERR_ARG, "NULL argument");
  if (opts->fixed_block_modes[0] >= 0) return fail(C1_ERR_ARG, "the detector taps need transient detection (fixed_block_modes -1)");
  if (frames > kMaxChunkFrames) return fail(C1_ERR_ARG, "stage taps are not chunked: at most %lld frames per call", (long long)kMaxChunkFrames);
  for (int c = 0; c < channels; c++)
    if (!pcm[c] || ((uintptr_t)pcm[c] & 15)) return fail(C1_ERR_ARG, "pcm[%d] must be a 16-byte aligned device pointer", c);
  if ((rc = upload_opts(ctx, opts))) return rc;
  if (frames == 0) return C1_OK;
  const int64_t units = frames * channels;
  if ((rc = ensure_workspace(ctx, units))) return rc;
  if ((rc = ensure_detect_workspace(ctx, units))) return rc;
  C1EncodeLaunch L;
  memset(&L, 0, sizeof L);
  for (int c = 0; c < channels; c++) L.pcm[c] = pcm[c];
  L.channels = channels; L.frames = frames; L.halo_frames = halo_frames;
  L.tables = ctx->d_tables; L.opts = ctx->d_opts;
  L.coefs = ctx->d_coefs[0]; L.side = ctx->d_side[0]; L.alloc = ctx->d_alloc[0]; L.cand = ctx->d_cand[0];
  L.work_count = ctx->d_work[0]; L.work_list = ctx->d_work[0] + 4; L.sel_list = ctx->d_work[0] + 4 + (size_t)ctx->ws_units * 7;
  L.mags = mags;
  c1k_launch_detect(L, ctx->d_bands[0], ctx->d_feat[0], ctx->d_modes[0], ctx->d_lists[0], false, nullptr, ctx->stream);
  if (modes) HIP_TRY(hipMemcpyAsync(modes, ctx->d_modes[0], (size_t)units, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(hipGetLastError());
  return C1_OK;
}

int c1_detect_scores_device(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                            const c1_encode_options *opts, int speculative, double *scores, uint8_t *modes, uint32_t *open_units) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if ((rc = check_channels(channels))) return rc;
  if (frames < 0 || halo_frames < 0 || halo_frames > 2) return fail(C1_ERR_ARG, "bad frames / halo_frames");
  if (!pcm || !opts) return fail(C1_ERR_ARG, "NULL argument");
  if (opts->fixed_block_modes[0] >= 0) return fail(C1_ERR_ARG, "the detector taps need transient detection (fixed_block_modes -1)");
  if (frames > kMaxChunkFrames) return fail(C1_ERR_ARG, "stage taps are not chunked: at most %lld frames per call", (long long)kMaxChunkFrames);
  if (speculative && !ctx->spec_tables_ok) return fail(C1_ERR_ARG, "the installed tables do not admit the speculative paths");
  for (int c = 0; c < channels; c++)
    if (!pcm[c] || ((uintptr_t)pcm[c] & 15)) return fail(C1_ERR_ARG, "pcm[%d] must be a 16-byte aligned device pointer", c);
  if ((rc = upload_opts(ctx, opts))) return rc;
  if (frames == 0) return C1_OK;
  const int64_t units = frames * channels;
  if ((rc = ensure_detect_workspace(ctx, units))) return rc;
  C1EncodeLaunch L;
  memset(&L, 0, sizeof L);
  for (int c = 0; c < channels; c++) L.pcm[c] = pcm[c];
  L.channels = channels; L.frames = frames; L.halo_frames = halo_frames;
  L.tables = ctx->d_tables; L.opts = ctx->d_opts;
  c1k_launch_detect(L, ctx->d_bands[0], ctx->d_feat[0], ctx->d_modes[0], ctx->d_lists[0], speculative != 0, scores, ctx->stream);
  if (modes) HIP_TRY(hipMemcpyAsync(modes, ctx->d_modes[0], (size_t)units, hipMemcpyDeviceToDevice, ctx->stream));
  if (open_units) HIP_TRY(hipMemcpyAsync(open_units, ctx->d_lists[0] + 2, sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(hipGetLastError());
  return C1_OK;
}

int c1_detect_spec_mags_device(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                               float *mags, float *bounds) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if ((rc = check_channels(channels))) return rc;
  if (frames < 0 || halo_frames < 0 || halo_frames > 2) return fail(C1_ERR_ARG, "bad frames / halo_frames");
  if (!pcm || !mags || !bounds) return fail(C1_ERR_ARG, "NULL argument");
  if (frames > kMaxChunkFrames) return fail(C1_ERR_ARG, "stage taps are not chunked: at most %lld frames per call", (long long)kMaxChunkFrames);
  if (!ctx->spec_tables_ok) return fail(C1_ERR_ARG, "the installed tables do not admit the speculative paths");
  for (int c = 0; c < channels; c++)
    if (!pcm[c] || ((uintptr_t)pcm[c] & 15)) return fail(C1_ERR_ARG, "pcm[%d] must be a 16-byte aligned device pointer", c);
  if (frames == 0) return C1_OK;
  if ((rc = ensure_detect_workspace(ctx, frames * channels))) return rc;
  C1EncodeLaunch L;
  memset(&L, 0, sizeof L);
  for (int c = 0; c < channels; c++) L.pcm[c] = pcm[c];
  L.channels = channels; L.frames = frames; L.halo_frames = halo_frames;
  L.tables = ctx->d_tables; L.opts = ctx->d_opts;
  L.mags = mags; L.mag_bounds = bounds;
  c1k_launch_detect_spec_tap(L, ctx->d_bands[0], ctx->d_feat[0], ctx->stream);
  HIP_TRY(hipGetLastError());
  return C1_OK;
}

int c1_log2f_error_device(c1_ctx *ctx, uint32_t first_bits, uint64_t count, double *out_host) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (!out_host) return fail(C1_ERR_ARG, "out is NULL");
  if (first_bits < 0x00800000u || (uint64_t)first_bits + count > 0x7f800000ull) return fail(C1_ERR_ARG, "range must stay within the normal positive binary32 numbers");
  if ((rc = ensure_io(ctx, 16))) return rc;
  c1k_launch_log2f_error(first_bits, count, reinterpret_cast<unsigned long long *>(ctx->d_io), ctx->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out_host, ctx->d_io, 16, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return C1_OK;
}

int c1_alloc_bounds_device(c1_ctx *ctx, const uint8_t *side, int64_t units, const c1_encode_options *opts, double *out) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (units < 0 || units > ((int64_t)1 << 24) || (units > 0 && (!side || !out))) return fail(C1_ERR_ARG, "bad units / NULL argument");
  if ((rc = upload_opts(ctx, opts))) return rc;
  if (units == 0) return C1_OK;
  if ((rc = ensure_workspace(ctx, units * 2))) return rc;        // the tap lists all eight candidates of every unit: 9 list entries per unit
  C1EncodeLaunch L;
  memset(&L, 0, sizeof L);
  L.channels = 1;
  L.frames = units;
  L.tables = ctx->d_tables;
  L.opts = ctx->d_opts;
  L.side = const_cast<uint8_t *>(side);
  L.alloc = ctx->d_alloc[0];
  L.cand = ctx->d_cand[0];
  L.work_count = ctx->d_work[0];
  L.work_list = ctx->d_work[0] + 4;
  L.sel_list = ctx->d_work[0] + 4 + (size_t)ctx->ws_units * 7;
  c1k_launch_alloc_tap(L, out, ctx->stream);
  HIP_TRY(hipGetLastError());
  return C1_OK;
}

int c1_libm_device(c1_ctx *ctx, int fn, const double *in, double *out, int64_t n) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (fn < 0 || fn > 3 || n < 0 || (n > 0 && (!in || !out))) return fail(C1_ERR_ARG, "bad fn / n / NULL argument");
  c1k_launch_libm(fn, in, out, n, ctx->stream);
  HIP_TRY(hipGetLastError());
  return C1_OK;
}

int c1_spec_stages_device(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                          const c1_encode_options *opts, float *coefs, float *eps, uint8_t *side) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if ((rc = check_channels(channels))) return rc;
  if (frames < 0 || halo_frames < 0 || halo_frames > 2) return fail(C1_ERR_ARG, "bad frames / halo_frames");
  if (!pcm || !opts || !coefs || !eps || !side) return fail(C1_ERR_ARG, "NULL argument");
  const bool sp_long = opts->fixed_block_modes[0] == 0 && opts->fixed_block_modes[1] == 0 && opts->fixed_block_modes[2] == 0;
  const bool sp_short = opts->fixed_block_modes[0] > 0 && opts->fixed_block_modes[1] > 0 && opts->fixed_block_modes[2] > 0;
  if (!sp_long && !sp_short)
    return fail(C1_ERR_ARG, "the speculative analysis covers fixed block modes with all bands long or all bands short");
  if (!ctx->spec_tables_ok) return fail(C1_ERR_STATE, "the installed tables fail the checks the error bound relies on");
  for (int c = 0; c < channels; c++)
    if (!pcm[c] || ((uintptr_t)pcm[c] & 15)) return fail(C1_ERR_ARG, "pcm[%d] must be a 16-byte aligned device pointer", c);
  if ((rc = upload_opts(ctx, opts))) return rc;
  if (frames == 0) return C1_OK;
  C1EncodeLaunch L;
  memset(&L, 0, sizeof L);
  for (int c = 0; c < channels; c++) L.pcm[c] = pcm[c];
  L.channels = channels; L.frames = frames; L.halo_frames = halo_frames;
  L.tables = ctx->d_tables; L.opts = ctx->d_opts;
  L.coefs = coefs; L.eps = eps; L.side = side;
  c1k_launch_analysis_spec(L, sp_short, ctx->stream);
  HIP_TRY(hipGetLastError());
  return C1_OK;
}

// ---- the single-stage functions of the reference's export surface (codec/index.js:30-35,42), host-resident ----------

int c1_quantize(c1_ctx *ctx, const float *coefficients, int n, int scale_factor_index, int bits_per_sample, int32_t *out) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (n < 0 || (n > 0 && (!coefficients || !out))) return fail(C1_ERR_ARG, "quantize: bad arguments");
  if (scale_factor_index < 0 || scale_factor_index > 63) return fail(C1_ERR_ARG, "quantize: scaleFactorIndex %d outside SCALE_FACTORS", scale_factor_index);
  if (bits_per_sample < 0 || bits_per_sample > 32) return fail(C1_ERR_ARG, "quantize: bitsPerSample %d", bits_per_sample);
  if (n == 0) return C1_OK;
  DeviceScratch ds;
  float *dx; int32_t *dq;
  if ((rc = ds.alloc(&dx, (size_t)n)) || (rc = ds.alloc(&dq, (size_t)n))) return rc;
  HIP_TRY(hipMemcpyAsync(dx, coefficients, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  c1k_launch_quantize_one(ctx->d_tables, dx, n, scale_factor_index, bits_per_sample, dq, ctx->stream);
  HIP_TRY(hipMemcpyAsync(out, dq, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return C1_OK;
}

int c1_dequantize(c1_ctx *ctx, const int32_t *quantized, int n, int scale_factor_index, int bits_per_sample, float *out) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (n < 0 || (n > 0 && (!quantized || !out))) return fail(C1_ERR_ARG, "dequantize: bad arguments");
  if (scale_factor_index < 0 || scale_factor_index > 63) return fail(C1_ERR_ARG, "dequantize: scaleFactorIndex %d outside SCALE_FACTORS", scale_factor_index);
  if (bits_per_sample < 0 || bits_per_sample > 32) return fail(C1_ERR_ARG, "dequantize: bitsPerSample %d", bits_per_sample);
  if (n == 0) return C1_OK;
  DeviceScratch ds;
  int32_t *dq; float *dx;
  if ((rc = ds.alloc(&dq, (size_t)n)) || (rc = ds.alloc(&dx, (size_t)n))) return rc;
  HIP_TRY(hipMemcpyAsync(dq, quantized, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  c1k_launch_dequantize_one(ctx->d_tables, dq, n, scale_factor_index, bits_per_sample, dx, ctx->stream);
  HIP_TRY(hipMemcpyAsync(out, dx, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return C1_OK;
}

int c1_fft(c1_ctx *ctx, float *real, float *imag, int n, const double *w) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (n < 1 || (n & (n - 1)) || n > (1 << 22)) return fail(C1_ERR_ARG, "fft: size must be a power of two <= 2^22, got %d", n);
  if (!real || !imag || (n > 1 && !w)) return fail(C1_ERR_ARG, "fft: NULL argument");
  if (n == 1) return C1_OK;                                  // fft.js:16
  int stages = 0;
  while ((1 << stages) < n) stages++;
  DeviceScratch ds;
  float *dr, *di; double *dw, *dtw;
  if ((rc = ds.alloc(&dr, (size_t)n)) || (rc = ds.alloc(&di, (size_t)n)) || (rc = ds.alloc(&dw, (size_t)2 * stages)) || (rc = ds.alloc(&dtw, (size_t)n))) return rc;
  HIP_TRY(hipMemcpyAsync(dr, real, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(di, imag, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(dw, w, (size_t)2 * stages * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  c1k_launch_fft_reference(dr, di, n, dw, dtw, ctx->stream);
  HIP_TRY(hipMemcpyAsync(real, dr, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(imag, di, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return C1_OK;
}

int c1_qmf_analysis_batch(c1_ctx *ctx, const float *pcm, int64_t frames, int halo_frames, float *bands) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (frames < 0 || halo_frames < 0 || halo_frames > 2) return fail(C1_ERR_ARG, "qmf analysis: bad frames / halo_frames");
  if (frames == 0) return C1_OK;
  if (!pcm || !bands) return fail(C1_ERR_ARG, "qmf analysis: NULL argument");
  if (frames > (1 << 20)) return fail(C1_ERR_ARG, "qmf analysis: at most 2^20 frames per call");
  DeviceScratch ds;
  float *dp, *db, *dc; uint8_t *dside, *dalloc;
  const size_t total = (size_t)(frames + halo_frames) * 512;
  if ((rc = ds.alloc(&dp, total)) || (rc = ds.alloc(&db, (size_t)frames * 512)) || (rc = ds.alloc(&dc, (size_t)frames * 512)) ||
      (rc = ds.alloc(&dside, (size_t)frames * kSideBytes)) || (rc = ds.alloc(&dalloc, (size_t)frames * kAllocBytes))) return rc;
  HIP_TRY(hipMemcpyAsync(dp, pcm, total * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  c1_encode_options o;
  c1_default_encode_options(&o);
  o.fixed_block_modes[0] = o.fixed_block_modes[1] = o.fixed_block_modes[2] = 0;   // the bands do not depend on the block modes
  const float *chan[1] = {dp + (size_t)halo_frames * 512};
  if ((rc = encode_device_impl(ctx, chan, 1, frames, halo_frames, &o, nullptr, db, dc, dside, dalloc))) return rc;
  HIP_TRY(hipMemcpyAsync(bands, db, (size_t)frames * 512 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return C1_OK;
}

int c1_mdct_batch(c1_ctx *ctx, const float *bands, int64_t frames, int halo_frames, const int32_t *block_modes, float *coefs,
                  float *bands_windowed) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (frames < 0 || halo_frames < 0 || halo_frames > 1) return fail(C1_ERR_ARG, "mdct: halo_frames must be 0 or 1");
  if (frames == 0) return C1_OK;
  if (!bands || !block_modes || !coefs) return fail(C1_ERR_ARG, "mdct: NULL argument");
  if (frames > (1 << 20)) return fail(C1_ERR_ARG, "mdct: at most 2^20 frames per call");
  // any non-zero mode is a short band (encoder.js:196); the unit header's values are 2, 2, 3
  std::vector<uint8_t> modes((size_t)frames);
  std::vector<uint32_t> lists(4 + 2 * (size_t)frames, 0u);
  for (int64_t f = 0; f < frames; f++) {
    const int m0 = block_modes[3 * f] ? 2 : 0, m1 = block_modes[3 * f + 1] ? 2 : 0, m2 = block_modes[3 * f + 2] ? 3 : 0;
    modes[(size_t)f] = (uint8_t)(m0 | (m1 << 2) | (m2 << 4));
    if (modes[(size_t)f] == 0) lists[4 + lists[0]++] = (uint32_t)f;
    else lists[4 + (size_t)frames + lists[1]++] = (uint32_t)f;
  }
  DeviceScratch ds;
  float *db, *dc, *dw = nullptr; uint8_t *dm, *dside; uint32_t *dl;
  if ((rc = ds.alloc(&db, (size_t)(frames + 1) * 512)) || (rc = ds.alloc(&dc, (size_t)frames * 512)) || (rc = ds.alloc(&dm, (size_t)frames)) ||
      (rc = ds.alloc(&dside, (size_t)frames * kSideBytes)) || (rc = ds.alloc(&dl, lists.size()))) return rc;
  if (bands_windowed && (rc = ds.alloc(&dw, (size_t)frames * 512))) return rc;
  // slot row 0 = frame -1: the halo frame, or nothing (a fresh BufferPool's zero overlap, buffers.js:44-48)
  if (!halo_frames) HIP_TRY(hipMemsetAsync(db, 0, 512 * sizeof(float), ctx->stream));
  HIP_TRY(hipMemcpyAsync(db + (halo_frames ? 0 : 512), bands, (size_t)(frames + halo_frames) * 512 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(dm, modes.data(), modes.size(), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(dl, lists.data(), lists.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  C1EncodeLaunch L;
  memset(&L, 0, sizeof L);
  L.channels = 1; L.frames = frames; L.halo_frames = halo_frames;
  L.tables = ctx->d_tables; L.opts = ctx->d_opts;
  L.coefs = dc; L.side = dside;
  c1k_launch_mdct_bands(L, db, dm, dl, ctx->stream);
  if (dw) c1k_launch_window_bands(db + 512, dm, frames, ctx->d_tables, dw, ctx->stream);
  HIP_TRY(hipMemcpyAsync(coefs, dc, (size_t)frames * 512 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (dw) HIP_TRY(hipMemcpyAsync(bands_windowed, dw, (size_t)frames * 512 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));       // also: the host vectors above outlive the copies
  HIP_TRY(hipGetLastError());
  return C1_OK;
}

int c1_pack_spec_tap_device(c1_ctx *ctx, const float *coefs, const float *eps, const uint8_t *side, const uint8_t *alloc,
                            int64_t units, int all_long, uint8_t *units_out, uint32_t *lists) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (units < 0 || units > kMaxChunkFrames) return fail(C1_ERR_ARG, "bad unit count");
  if (!coefs || !eps || !side || !alloc || !units_out || !lists) return fail(C1_ERR_ARG, "NULL argument");
  if (((uintptr_t)coefs | (uintptr_t)eps) & 15) return fail(C1_ERR_ARG, "coefs and eps must be 16-byte aligned device pointers");
  if (!ctx->spec_tables_ok) return fail(C1_ERR_STATE, "the installed tables fail the checks the error bound relies on");
  HIP_TRY(hipMemsetAsync(lists, 0, kListHead * sizeof(uint32_t), ctx->stream));
  if (units == 0) return C1_OK;
  C1EncodeLaunch L;
  memset(&L, 0, sizeof L);
  L.channels = 1; L.frames = units;
  L.tables = ctx->d_tables; L.opts = ctx->d_opts;
  L.coefs = const_cast<float *>(coefs); L.eps = const_cast<float *>(eps);
  L.side = const_cast<uint8_t *>(side); L.alloc = const_cast<uint8_t *>(alloc);
  L.units = units_out;
  L.redo_count = lists; L.realloc_count = lists + 1; L.reana_count = lists + 2;
  L.redo_list = lists + kListHead; L.realloc_list = lists + kListHead + units; L.reana_list = lists + kListHead + 2 * units;
  c1k_launch_pack_spec(L, all_long != 0, ctx->stream);
  HIP_TRY(hipGetLastError());
  return C1_OK;
}

// ---- streamed host path --------------------------------------------------------------------------------------
namespace {
constexpr int64_t kStreamChunkFrames = 32768;   // frames per channel per chunk of the streamed host path

int ensure_ring(c1_ctx *ctx, size_t bytes) {
  if (!ctx->s_up) {
    HIP_TRY(hipStreamCreateWithFlags(&ctx->s_up, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->s_down, hipStreamNonBlocking));
    for (int p = 0; p < 2; p++) {
      HIP_TRY(hipEventCreateWithFlags(&ctx->ev_up[p], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&ctx->ev_run[p], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&ctx->ev_down[p], hipEventDisableTiming));
    }
  }
  if (bytes <= ctx->d_ring_bytes) return C1_OK;
  HIP_TRY(hipDeviceSynchronize());
  if (ctx->d_ring) (void)hipFree(ctx->d_ring);
  ctx->d_ring = nullptr; ctx->d_ring_bytes = 0;
  HIP_TRY(hipMalloc(&ctx->d_ring, bytes));
  ctx->d_ring_bytes = bytes;
  return C1_OK;
}

// Leaves no copy in flight into or out of the caller's host memory, whichever way the function returns.  While it
// lives, the per-chunk device calls add to the context's kernel timings instead of restarting them.
struct StreamDrain {
  c1_ctx *ctx;
  explicit StreamDrain(c1_ctx *c) : ctx(c) {
    if (ctx->profiling && ctx->timing_depth == 0) reset_timings(ctx);
    ctx->timing_depth++;
  }
  ~StreamDrain() {
    ctx->timing_depth--;
    if (ctx->s_up) (void)hipStreamSynchronize(ctx->s_up);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->s_down) (void)hipStreamSynchronize(ctx->s_down);
  }
};

// upload of chunk i+1 | kernels of chunk i | download of chunk i-1, two staging sets.  The download of chunk i-1 is
// queued after the upload and the kernels of chunk i, so a download into pageable memory (which blocks the host
// until it is done) still leaves the other two stages of the next chunk in flight.
int encode_batch_streamed(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                          const c1_encode_options *opts, uint8_t *units) {
  const int64_t chunk = kStreamChunkFrames;
  const size_t in_bytes = (size_t)(chunk + 2) * 512 * sizeof(float);            // per channel, with the 2-frame halo
  const size_t out_bytes = ((size_t)chunk * channels * C1_UNIT_BYTES + 255) & ~(size_t)255;
  const size_t set_bytes = in_bytes * channels + out_bytes;
  int rc = ensure_ring(ctx, 2 * set_bytes);
  if (rc) return rc;
  StreamDrain drain(ctx);
  auto download = [&](int64_t index) -> int {
    const int p = (int)(index & 1);
    const int64_t f0 = index * chunk, n = std::min(chunk, frames - f0);
    const uint8_t *d_units = reinterpret_cast<const uint8_t *>((char *)ctx->d_ring + (size_t)p * set_bytes + in_bytes * channels);
    HIP_TRY(hipStreamWaitEvent(ctx->s_down, ctx->ev_run[p], 0));
    HIP_TRY(hipMemcpyAsync(units + (size_t)f0 * channels * C1_UNIT_BYTES, d_units, (size_t)n * channels * C1_UNIT_BYTES,
                           hipMemcpyDeviceToHost, ctx->s_down));
    HIP_TRY(hipEventRecord(ctx->ev_down[p], ctx->s_down));
    return C1_OK;
  };
  int64_t index = 0;
  for (int64_t f0 = 0; f0 < frames; f0 += chunk, ++index) {
    const int p = (int)(index & 1);
    const int64_t n = std::min(chunk, frames - f0);
    const int h = (int)std::min<int64_t>(2, f0 + halo_frames);
    char *set = (char *)ctx->d_ring + (size_t)p * set_bytes;
    if (index >= 2) {                                           // staging set p is free again
      HIP_TRY(hipStreamWaitEvent(ctx->s_up, ctx->ev_run[p], 0));
      HIP_TRY(hipStreamWaitEvent(ctx->s_up, ctx->ev_down[p], 0));
      HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_down[p], 0));
    }
    const float *dptr[C1_MAX_CHANNELS] = {nullptr, nullptr};
    for (int c = 0; c < channels; c++) {
      float *d = reinterpret_cast<float *>(set + in_bytes * c);
      HIP_TRY(hipMemcpyAsync(d, pcm[c] + (f0 - h) * 512, (size_t)(n + h) * 512 * sizeof(float), hipMemcpyHostToDevice, ctx->s_up));
      dptr[c] = d + (size_t)h * 512;
    }
    HIP_TRY(hipEventRecord(ctx->ev_up[p], ctx->s_up));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_up[p], 0));
    uint8_t *d_units = reinterpret_cast<uint8_t *>(set + in_bytes * channels);
    if ((rc = encode_device_joined(ctx, dptr, channels, n, h, opts, d_units))) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_run[p], ctx->stream));
    if (index >= 1 && (rc = download(index - 1))) return rc;
  }
  if ((rc = download(index - 1))) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->s_down));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return C1_OK;
}

int decode_batch_streamed(c1_ctx *ctx, const uint8_t *units, int channels, int64_t frames, int halo_units, float *const *pcm) {
  const int64_t chunk = kStreamChunkFrames;
  const size_t in_bytes = ((size_t)(chunk + 1) * channels * C1_UNIT_BYTES + 255) & ~(size_t)255;
  const size_t out_bytes = (size_t)chunk * 512 * sizeof(float);                // per channel
  const size_t set_bytes = in_bytes + out_bytes * channels;
  int rc = ensure_ring(ctx, 2 * set_bytes);
  if (rc) return rc;
  StreamDrain drain(ctx);
  auto download = [&](int64_t index) -> int {
    const int p = (int)(index & 1);
    const int64_t f0 = index * chunk, n = std::min(chunk, frames - f0);
    char *set = (char *)ctx->d_ring + (size_t)p * set_bytes;
    HIP_TRY(hipStreamWaitEvent(ctx->s_down, ctx->ev_run[p], 0));
    for (int c = 0; c < channels; c++)
      HIP_TRY(hipMemcpyAsync(pcm[c] + f0 * 512, set + in_bytes + out_bytes * c, (size_t)n * 512 * sizeof(float), hipMemcpyDeviceToHost, ctx->s_down));
    HIP_TRY(hipEventRecord(ctx->ev_down[p], ctx->s_down));
    return C1_OK;
  };
  int64_t index = 0;
  for (int64_t f0 = 0; f0 < frames; f0 += chunk, ++index) {
    const int p = (int)(index & 1);
    const int64_t n = std::min(chunk, frames - f0);
    const int h = (int)std::min<int64_t>(1, f0 + halo_units);
    char *set = (char *)ctx->d_ring + (size_t)p * set_bytes;
    if (index >= 2) {
      HIP_TRY(hipStreamWaitEvent(ctx->s_up, ctx->ev_run[p], 0));
      HIP_TRY(hipStreamWaitEvent(ctx->s_up, ctx->ev_down[p], 0));
      HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_down[p], 0));
    }
    const size_t hb = (size_t)h * channels * C1_UNIT_BYTES;
    HIP_TRY(hipMemcpyAsync(set, units + (size_t)f0 * channels * C1_UNIT_BYTES - hb, (size_t)n * channels * C1_UNIT_BYTES + hb,
                           hipMemcpyHostToDevice, ctx->s_up));
    HIP_TRY(hipEventRecord(ctx->ev_up[p], ctx->s_up));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_up[p], 0));
    float *dptr[C1_MAX_CHANNELS] = {nullptr, nullptr};
    for (int c = 0; c < channels; c++) dptr[c] = reinterpret_cast<float *>(set + in_bytes + out_bytes * c);
    if ((rc = c1_decode_device(ctx, (const uint8_t *)set + hb, channels, n, h, dptr))) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_run[p], ctx->stream));
    if (index >= 1 && (rc = download(index - 1))) return rc;
  }
  if ((rc = download(index - 1))) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->s_down));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return C1_OK;
}
}  // namespace

int c1_host_alloc(size_t bytes, void **out) {
  if (!out) return fail(C1_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (bytes == 0) return C1_OK;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); return fail(C1_ERR_NO_DEVICE, "no HIP device: page-locked memory needs the HIP runtime"); }
  HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return C1_OK;
}

int c1_host_free(void *p) {
  if (!p) return C1_OK;
  HIP_TRY(hipHostFree(p));
  return C1_OK;
}

int c1_encode_batch(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                    const c1_encode_options *opts, uint8_t *units) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if ((rc = check_channels(channels))) return rc;
  if (frames < 0 || halo_frames < 0 || halo_frames > 2) return fail(C1_ERR_ARG, "bad frames / halo_frames");
  if (frames == 0) return C1_OK;
  if (!pcm || !units) return fail(C1_ERR_ARG, "pcm or units is NULL");
  if (frames > 2 * kStreamChunkFrames) {
    // large batches are streamed in chunks (upload | kernels | download on three streams), from page-locked memory at the
    // link's rate (12.9 M stereo frames/s), from pageable memory nearly so (11.8 M: the runtime stages those copies, but the
    // chunks keep the stages of the pipeline busy; copy - compute - copy in one piece managed 8.1 M)
    for (int c = 0; c < channels; c++) if (!pcm[c]) return fail(C1_ERR_ARG, "pcm[%d] is NULL", c);
    return encode_batch_streamed(ctx, pcm, channels, frames, halo_frames, opts, units);
  }
  const size_t ch_bytes = (size_t)(frames + halo_frames) * 512 * sizeof(float);
  const size_t unit_bytes = (size_t)frames * channels * C1_UNIT_BYTES;
  const size_t unit_off = (ch_bytes * channels + 255) & ~(size_t)255;
  if ((rc = ensure_io(ctx, unit_off + unit_bytes))) return rc;
  const float *dptr[C1_MAX_CHANNELS] = {nullptr, nullptr};
  for (int c = 0; c < channels; c++) {
    if (!pcm[c]) return fail(C1_ERR_ARG, "pcm[%d] is NULL", c);
    float *d = reinterpret_cast<float *>((char *)ctx->d_io + ch_bytes * c);
    HIP_TRY(hipMemcpyAsync(d, pcm[c] - (size_t)halo_frames * 512, ch_bytes, hipMemcpyHostToDevice, ctx->stream));
    dptr[c] = d + (size_t)halo_frames * 512;
  }
  uint8_t *d_units = (uint8_t *)ctx->d_io + unit_off;
  if ((rc = encode_device_joined(ctx, dptr, channels, frames, halo_frames, opts, d_units))) return rc;
  HIP_TRY(hipMemcpyAsync(units, d_units, unit_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return C1_OK;
}

int c1_decode_device(c1_ctx *ctx, const uint8_t *units, int channels, int64_t frames, int halo_units,
                     float *const *pcm) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if ((rc = check_channels(channels))) return rc;
  if (frames < 0) return fail(C1_ERR_ARG, "frames must be >= 0");
  if (halo_units < 0 || halo_units > 1) return fail(C1_ERR_ARG, "halo_units must be 0 or 1");
  if (ctx->profiling && ctx->timing_depth == 0) reset_timings(ctx);
  if (frames == 0) return C1_OK;
  if (!units || !pcm) return fail(C1_ERR_ARG, "units or pcm is NULL");
  if ((uintptr_t)units & 3) return fail(C1_ERR_ARG, "units must be 4-byte aligned on the device");
  C1DecodeLaunch L;
  memset(&L, 0, sizeof L);
  L.units = units;
  L.channels = channels;
  L.frames = frames;
  L.halo_units = halo_units;
  L.tables = ctx->d_tables;
  for (int c = 0; c < channels; c++) {
    if (!pcm[c]) return fail(C1_ERR_ARG, "pcm[%d] is NULL", c);
    if ((uintptr_t)pcm[c] & 15) return fail(C1_ERR_ARG, "pcm[%d] must be 16-byte aligned on the device", c);
    L.pcm[c] = pcm[c];
  }
  { ScopedTiming t(ctx, K_DECODE); c1k_launch_decode(L, ctx->decode_binary32, ctx->stream); }
  HIP_TRY(hipGetLastError());
  return C1_OK;
}

int c1_decode_batch(c1_ctx *ctx, const uint8_t *units, int channels, int64_t frames, int halo_units,
                    float *const *pcm) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if ((rc = check_channels(channels))) return rc;
  if (frames < 0 || halo_units < 0 || halo_units > 1) return fail(C1_ERR_ARG, "bad frames / halo_units");
  if (frames == 0) return C1_OK;
  if (!units || !pcm) return fail(C1_ERR_ARG, "units or pcm is NULL");
  if (frames > 2 * kStreamChunkFrames) {
    for (int c = 0; c < channels; c++) if (!pcm[c]) return fail(C1_ERR_ARG, "pcm[%d] is NULL", c);
    return decode_batch_streamed(ctx, units, channels, frames, halo_units, pcm);
  }
  const size_t halo_bytes = (size_t)halo_units * channels * C1_UNIT_BYTES;   // multiple of 4
  const size_t unit_bytes = (size_t)frames * channels * C1_UNIT_BYTES + halo_bytes;
  const size_t pcm_off = (unit_bytes + 255) & ~(size_t)255;
  const size_t ch_bytes = (size_t)frames * 512 * sizeof(float);
  if ((rc = ensure_io(ctx, pcm_off + ch_bytes * channels))) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->d_io, units - halo_bytes, unit_bytes, hipMemcpyHostToDevice, ctx->stream));
  float *dptr[C1_MAX_CHANNELS] = {nullptr, nullptr};
  for (int c = 0; c < channels; c++) dptr[c] = reinterpret_cast<float *>((char *)ctx->d_io + pcm_off + ch_bytes * c);
  if ((rc = c1_decode_device(ctx, (const uint8_t *)ctx->d_io + halo_bytes, channels, frames, halo_units, dptr))) return rc;
  for (int c = 0; c < channels; c++) {
    if (!pcm[c]) return fail(C1_ERR_ARG, "pcm[%d] is NULL", c);
    HIP_TRY(hipMemcpyAsync(pcm[c], dptr[c], ch_bytes, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return C1_OK;
}

// ---- one batch over several devices ------------------------------------------------------------------------------------
namespace {
// Contexts for the *_multi entry points: one per shard for the duration of a call, checked out of a process-wide pool and
// given back at its end.  A context is never shared between two calls in flight nor destroyed while one uses it (a second
// call that needs the same device meanwhile gets another context); a context made before the last c1_set_tables() is
// retired when it comes back, so the *_multi entry points follow the installed tables like a context created afresh.
struct ShardPool {
  struct Entry { int device; uint64_t tables_gen; c1_ctx *ctx; bool busy; };
  std::mutex mu;
  std::vector<Entry> entries;
  c1_ctx *checkout(int device, int *rc) {
    const uint64_t gen = g_tables_gen.load();
    std::vector<c1_ctx *> stale;
    c1_ctx *found = nullptr;
    {
      std::lock_guard<std::mutex> lock(mu);
      for (size_t i = 0; i < entries.size();) {
        Entry &e = entries[i];
        if (!e.busy && e.tables_gen != gen) { stale.push_back(e.ctx); entries.erase(entries.begin() + (long)i); continue; }
        if (!found && !e.busy && e.device == device) { e.busy = true; found = e.ctx; }
        ++i;
      }
    }
    for (c1_ctx *c : stale) c1_ctx_destroy(c);
    *rc = C1_OK;
    if (found) return found;
    c1_ctx *c = nullptr;
    *rc = c1_ctx_create(device, nullptr, &c);
    if (*rc) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    entries.push_back({device, gen, c, true});
    return c;
  }
  void give_back(c1_ctx *c) {
    std::lock_guard<std::mutex> lock(mu);
    for (Entry &e : entries) if (e.ctx == c) e.busy = false;
  }
};
ShardPool g_shards;
struct ShardLease {           // the contexts of one *_multi call
  std::vector<c1_ctx *> ctxs;
  ~ShardLease() { for (c1_ctx *c : ctxs) if (c) g_shards.give_back(c); }
  int take(const int *devices, int shards) {
    for (int s = 0; s < shards; s++) {
      int rc = C1_OK;
      c1_ctx *c = g_shards.checkout(devices[s], &rc);
      if (rc) return rc;                               // c1_last_error() of this thread holds the reason
      ctxs.push_back(c);
    }
    return C1_OK;
  }
};

// contiguous ranges whose sizes differ by at most one frame
void shard_plan(int64_t frames, int shards, std::vector<std::pair<int64_t, int64_t>> *plan) {
  const int64_t base = frames / shards, extra = frames % shards;
  int64_t at = 0;
  for (int r = 0; r < shards; r++) {
    const int64_t n = base + (r < extra ? 1 : 0);
    plan->push_back({at, at + n});
    at += n;
  }
}
}  // namespace

int c1_encode_batch_multi(const int *devices, int n_devices, const float *const *pcm, int channels, int64_t frames,
                          int halo_frames, const c1_encode_options *opts, uint8_t *units) {
  int rc = check_channels(channels);
  if (rc) return rc;
  if (!devices || n_devices < 1 || n_devices > 64) return fail(C1_ERR_ARG, "devices: 1..64 entries");
  if (frames < 0 || halo_frames < 0 || halo_frames > 2) return fail(C1_ERR_ARG, "bad frames / halo_frames");
  if (frames == 0) return C1_OK;
  if (!pcm || !units || !opts) return fail(C1_ERR_ARG, "NULL argument");
  for (int c = 0; c < channels; c++) if (!pcm[c]) return fail(C1_ERR_ARG, "pcm[%d] is NULL", c);
  const int shards = (int)std::min<int64_t>(n_devices, frames);
  std::vector<std::pair<int64_t, int64_t>> plan;
  shard_plan(frames, shards, &plan);
  ShardLease lease;
  if ((rc = lease.take(devices, shards))) return rc;
  const std::vector<c1_ctx *> &ctxs = lease.ctxs;
  std::vector<int> rcs(shards, C1_OK);
  std::vector<std::string> errs(shards);
  std::vector<std::thread> threads;
  for (int s = 0; s < shards; s++)
    threads.emplace_back([&, s] {
      const int64_t a = plan[s].first, b = plan[s].second;
      const int h = (int)std::min<int64_t>(2, a + halo_frames);        // frames of real PCM directly in front of the range
      const float *ptrs[C1_MAX_CHANNELS] = {nullptr, nullptr};
      for (int c = 0; c < channels; c++) ptrs[c] = pcm[c] + a * 512;
      rcs[s] = c1_encode_batch(ctxs[s], ptrs, channels, b - a, h, opts, units + (size_t)a * channels * C1_UNIT_BYTES);
      if (rcs[s]) errs[s] = c1_last_error();
    });
  for (auto &t : threads) t.join();
  for (int s = 0; s < shards; s++)
    if (rcs[s]) return fail(rcs[s], "shard %d on device %d: %s", s, devices[s], errs[s].c_str());
  return C1_OK;
}

int c1_decode_batch_multi(const int *devices, int n_devices, const uint8_t *units, int channels, int64_t frames,
                          int halo_units, float *const *pcm) {
  int rc = check_channels(channels);
  if (rc) return rc;
  if (!devices || n_devices < 1 || n_devices > 64) return fail(C1_ERR_ARG, "devices: 1..64 entries");
  if (frames < 0 || halo_units < 0 || halo_units > 1) return fail(C1_ERR_ARG, "bad frames / halo_units");
  if (frames == 0) return C1_OK;
  if (!pcm || !units) return fail(C1_ERR_ARG, "NULL argument");
  for (int c = 0; c < channels; c++) if (!pcm[c]) return fail(C1_ERR_ARG, "pcm[%d] is NULL", c);
  const int shards = (int)std::min<int64_t>(n_devices, frames);
  std::vector<std::pair<int64_t, int64_t>> plan;
  shard_plan(frames, shards, &plan);
  ShardLease lease;
  if ((rc = lease.take(devices, shards))) return rc;
  const std::vector<c1_ctx *> &ctxs = lease.ctxs;
  std::vector<int> rcs(shards, C1_OK);
  std::vector<std::string> errs(shards);
  std::vector<std::thread> threads;
  for (int s = 0; s < shards; s++)
    threads.emplace_back([&, s] {
      const int64_t a = plan[s].first, b = plan[s].second;
      const int h = a > 0 ? 1 : halo_units;
      float *ptrs[C1_MAX_CHANNELS] = {nullptr, nullptr};
      for (int c = 0; c < channels; c++) ptrs[c] = pcm[c] + a * 512;
      rcs[s] = c1_decode_batch(ctxs[s], units + (size_t)a * channels * C1_UNIT_BYTES, channels, b - a, h, ptrs);
      if (rcs[s]) errs[s] = c1_last_error();
    });
  for (auto &t : threads) t.join();
  for (int s = 0; s < shards; s++)
    if (rcs[s]) return fail(rcs[s], "shard %d on device %d: %s", s, devices[s], errs[s].c_str());
  return C1_OK;
}

// ---- WAV body <-> units in one host call (streamed) -----------------------------------------------------------------
int c1_encode_wav_batch(c1_ctx *ctx, const void *interleaved, int bits, int channels, int64_t samples_per_channel,
                        const c1_encode_options *opts, uint8_t *units) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if ((rc = check_channels(channels))) return rc;
  if (bits != 16 && bits != 24 && bits != 32) return fail(C1_ERR_ARG, "bits must be 16, 24 or 32, got %d", bits);
  if (samples_per_channel < 0) return fail(C1_ERR_ARG, "samples_per_channel must be >= 0");
  if (samples_per_channel == 0) return C1_OK;
  if (!interleaved || !units) return fail(C1_ERR_ARG, "interleaved or units is NULL");
  const int64_t frames = (samples_per_channel + 511) / 512, chunk = kStreamChunkFrames;
  const size_t bps = (size_t)bits / 8, frame_raw = 512 * (size_t)channels * bps;
  const size_t raw_bytes = ((size_t)(chunk + 2) * frame_raw + 255) & ~(size_t)255;
  const size_t pcm_bytes = (size_t)(chunk + 2) * 512 * sizeof(float);          // per channel
  const size_t out_bytes = ((size_t)chunk * channels * C1_UNIT_BYTES + 255) & ~(size_t)255;
  const size_t set_bytes = raw_bytes + pcm_bytes * channels + out_bytes;
  if ((rc = ensure_ring(ctx, 2 * set_bytes))) return rc;
  StreamDrain drain(ctx);
  const uint8_t *src = static_cast<const uint8_t *>(interleaved);
  auto download = [&](int64_t index) -> int {
    const int p = (int)(index & 1);
    const int64_t f0 = index * chunk, n = std::min(chunk, frames - f0);
    const uint8_t *d_units = reinterpret_cast<const uint8_t *>((char *)ctx->d_ring + (size_t)p * set_bytes + raw_bytes + pcm_bytes * channels);
    HIP_TRY(hipStreamWaitEvent(ctx->s_down, ctx->ev_run[p], 0));
    HIP_TRY(hipMemcpyAsync(units + (size_t)f0 * channels * C1_UNIT_BYTES, d_units, (size_t)n * channels * C1_UNIT_BYTES,
                           hipMemcpyDeviceToHost, ctx->s_down));
    HIP_TRY(hipEventRecord(ctx->ev_down[p], ctx->s_down));
    return C1_OK;
  };
  int64_t index = 0;
  for (int64_t f0 = 0; f0 < frames; f0 += chunk, ++index) {
    const int p = (int)(index & 1);
    const int64_t n = std::min(chunk, frames - f0);
    const int h = (int)std::min<int64_t>(2, f0);                 // frames of PCM history in front of the chunk
    char *set = (char *)ctx->d_ring + (size_t)p * set_bytes;
    if (index >= 2) {
      HIP_TRY(hipStreamWaitEvent(ctx->s_up, ctx->ev_run[p], 0));
      HIP_TRY(hipStreamWaitEvent(ctx->s_up, ctx->ev_down[p], 0));
      HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_down[p], 0));
    }
    const int64_t s0 = (f0 - h) * 512;                                         // first sample (per channel) of the upload
    const int64_t have = std::min<int64_t>((n + h) * 512, samples_per_channel - s0);
    HIP_TRY(hipMemcpyAsync(set, src + (size_t)s0 * channels * bps, (size_t)have * channels * bps, hipMemcpyHostToDevice, ctx->s_up));
    HIP_TRY(hipEventRecord(ctx->ev_up[p], ctx->s_up));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_up[p], 0));
    float *dch[C1_MAX_CHANNELS] = {nullptr, nullptr};
    const float *dptr[C1_MAX_CHANNELS] = {nullptr, nullptr};
    for (int c = 0; c < channels; c++) {
      dch[c] = reinterpret_cast<float *>(set + raw_bytes + pcm_bytes * c);
      dptr[c] = dch[c] + (size_t)h * 512;
      if (have < (n + h) * 512)                                                // zero padding of the last, partial frame
        HIP_TRY(hipMemsetAsync(dch[c] + have, 0, (size_t)((n + h) * 512 - have) * sizeof(float), ctx->stream));
    }
    c1k_launch_pcm_from_int(set, bits, channels, have, dch, ctx->stream);
    if ((rc = encode_device_joined(ctx, dptr, channels, n, h, opts, reinterpret_cast<uint8_t *>(set + raw_bytes + pcm_bytes * channels)))) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_run[p], ctx->stream));
    if (index >= 1 && (rc = download(index - 1))) return rc;
  }
  if ((rc = download(index - 1))) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->s_down));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return C1_OK;
}

int c1_decode_wav16_batch(c1_ctx *ctx, const uint8_t *units, int channels, int64_t frames, int16_t *interleaved) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if ((rc = check_channels(channels))) return rc;
  if (frames < 0) return fail(C1_ERR_ARG, "frames must be >= 0");
  if (frames == 0) return C1_OK;
  if (!units || !interleaved) return fail(C1_ERR_ARG, "units or interleaved is NULL");
  const int64_t chunk = kStreamChunkFrames;
  const size_t in_bytes = ((size_t)(chunk + 1) * channels * C1_UNIT_BYTES + 255) & ~(size_t)255;
  const size_t pcm_bytes = (size_t)chunk * 512 * sizeof(float);                // per channel
  const size_t out_bytes = (size_t)chunk * 512 * channels * sizeof(int16_t);
  const size_t set_bytes = in_bytes + pcm_bytes * channels + out_bytes;
  if ((rc = ensure_ring(ctx, 2 * set_bytes))) return rc;
  StreamDrain drain(ctx);
  auto download = [&](int64_t index) -> int {
    const int p = (int)(index & 1);
    const int64_t f0 = index * chunk, n = std::min(chunk, frames - f0);
    char *set = (char *)ctx->d_ring + (size_t)p * set_bytes;
    HIP_TRY(hipStreamWaitEvent(ctx->s_down, ctx->ev_run[p], 0));
    HIP_TRY(hipMemcpyAsync(interleaved + (size_t)f0 * 512 * channels, set + in_bytes + pcm_bytes * channels,
                           (size_t)n * 512 * channels * sizeof(int16_t), hipMemcpyDeviceToHost, ctx->s_down));
    HIP_TRY(hipEventRecord(ctx->ev_down[p], ctx->s_down));
    return C1_OK;
  };
  int64_t index = 0;
  for (int64_t f0 = 0; f0 < frames; f0 += chunk, ++index) {
    const int p = (int)(index & 1);
    const int64_t n = std::min(chunk, frames - f0);
    const int h = (int)std::min<int64_t>(1, f0);
    char *set = (char *)ctx->d_ring + (size_t)p * set_bytes;
    if (index >= 2) {
      HIP_TRY(hipStreamWaitEvent(ctx->s_up, ctx->ev_run[p], 0));
      HIP_TRY(hipStreamWaitEvent(ctx->s_up, ctx->ev_down[p], 0));
      HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_down[p], 0));
    }
    const size_t hb = (size_t)h * channels * C1_UNIT_BYTES;
    HIP_TRY(hipMemcpyAsync(set, units + (size_t)f0 * channels * C1_UNIT_BYTES - hb, (size_t)n * channels * C1_UNIT_BYTES + hb,
                           hipMemcpyHostToDevice, ctx->s_up));
    HIP_TRY(hipEventRecord(ctx->ev_up[p], ctx->s_up));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_up[p], 0));
    float *dptr[C1_MAX_CHANNELS] = {nullptr, nullptr};
    for (int c = 0; c < channels; c++) dptr[c] = reinterpret_cast<float *>(set + in_bytes + pcm_bytes * c);
    if ((rc = c1_decode_device(ctx, (const uint8_t *)set + hb, channels, n, h, dptr))) return rc;
    c1k_launch_pcm_to_int16(dptr, channels, n * 512, reinterpret_cast<int16_t *>(set + in_bytes + pcm_bytes * channels), ctx->stream);
    HIP_TRY(hipEventRecord(ctx->ev_run[p], ctx->stream));
    if (index >= 1 && (rc = download(index - 1))) return rc;
  }
  if ((rc = download(index - 1))) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->s_down));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipGetLastError());
  return C1_OK;
}

// ---- stateful streams -------------------------------------------------------------------------------
struct c1_enc_stream {
  c1_ctx *ctx;
  int channels;
  c1_encode_options opts;
  float *d_hist;       // channels * 2 frames: the last 1024 PCM samples of each channel
  float *d_buf = nullptr;
  uint8_t *d_units = nullptr;
  int64_t cap_frames = 0;
};

int c1_enc_stream_create(c1_ctx *ctx, int channels, const c1_encode_options *opts, c1_enc_stream **out) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (!out || !opts) return fail(C1_ERR_ARG, "out or opts is NULL");
  if ((rc = check_channels(channels))) return rc;
  C1DevEncOpts probe;
  if ((rc = build_encode_opts(*opts, &probe))) return rc;
  c1_enc_stream *s = new c1_enc_stream();
  s->ctx = ctx; s->channels = channels; s->opts = *opts; s->d_hist = nullptr;
  const size_t hb = (size_t)channels * 1024 * sizeof(float);
  hipError_t e = hipMalloc(&s->d_hist, hb);
  if (e == hipSuccess) e = hipMemsetAsync(s->d_hist, 0, hb, ctx->stream);   // zero history == stream start
  if (e != hipSuccess) { delete s; return fail(C1_ERR_HIP, "enc stream alloc: %s", hipGetErrorString(e)); }
  *out = s;
  return C1_OK;
}

int c1_enc_stream_push(c1_enc_stream *s, const float *const *pcm, int64_t frames, uint8_t *units) {
  if (!s) return fail(C1_ERR_ARG, "stream is NULL");
  c1_ctx *ctx = s->ctx;
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (frames < 0) return fail(C1_ERR_ARG, "frames must be >= 0");
  if (frames == 0) return C1_OK;
  if (!pcm || !units) return fail(C1_ERR_ARG, "pcm or units is NULL");
  if (frames > s->cap_frames) {
    if (s->d_buf) { hipFree(s->d_buf); hipFree(s->d_units); }
    s->d_buf = nullptr; s->d_units = nullptr; s->cap_frames = 0;
    HIP_TRY(hipMalloc(&s->d_buf, (size_t)s->channels * (frames + 2) * 512 * sizeof(float)));
    HIP_TRY(hipMalloc(&s->d_units, (size_t)s->channels * frames * C1_UNIT_BYTES));
    s->cap_frames = frames;
  }
  const size_t stride = (size_t)(s->cap_frames + 2) * 512;
  const float *dptr[C1_MAX_CHANNELS] = {nullptr, nullptr};
  for (int c = 0; c < s->channels; c++) {
    if (!pcm[c]) return fail(C1_ERR_ARG, "pcm[%d] is NULL", c);
    float *d = s->d_buf + stride * c;
    HIP_TRY(hipMemcpyAsync(d, s->d_hist + 1024 * c, 1024 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d + 1024, pcm[c], (size_t)frames * 512 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    dptr[c] = d + 1024;
  }
  if ((rc = encode_device_joined(ctx, dptr, s->channels, frames, 2, &s->opts, s->d_units))) return rc;
  for (int c = 0; c < s->channels; c++)   // new history = the last two frames of [history | pushed]
    HIP_TRY(hipMemcpyAsync(s->d_hist + 1024 * c, s->d_buf + stride * c + (size_t)frames * 512, 1024 * sizeof(float),
                           hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(units, s->d_units, (size_t)s->channels * frames * C1_UNIT_BYTES, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return C1_OK;
}

int c1_enc_stream_destroy(c1_enc_stream *s) {
  if (!s) return C1_OK;
  hipSetDevice(s->ctx->device);
  hipStreamSynchronize(s->ctx->stream);
  if (s->d_hist) hipFree(s->d_hist);
  if (s->d_buf) hipFree(s->d_buf);
  if (s->d_units) hipFree(s->d_units);
  delete s;
  return C1_OK;
}

struct c1_dec_stream {
  c1_ctx *ctx;
  int channels;
  bool have_prev = false;
  uint8_t *d_prev = nullptr;   // channels units of the previous frame
  uint8_t *d_units = nullptr;
  float *d_pcm = nullptr;
  int64_t cap_frames = 0;
};

int c1_dec_stream_create(c1_ctx *ctx, int channels, c1_dec_stream **out) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (!out) return fail(C1_ERR_ARG, "out is NULL");
  if ((rc = check_channels(channels))) return rc;
  c1_dec_stream *s = new c1_dec_stream();
  s->ctx = ctx; s->channels = channels;
  const hipError_t e = hipMalloc(&s->d_prev, (size_t)channels * C1_UNIT_BYTES);
  if (e != hipSuccess) { delete s; return fail(C1_ERR_HIP, "dec stream alloc: %s", hipGetErrorString(e)); }
  *out = s;
  return C1_OK;
}

int c1_dec_stream_push(c1_dec_stream *s, const uint8_t *units, int64_t frames, float *const *pcm) {
  if (!s) return fail(C1_ERR_ARG, "stream is NULL");
  c1_ctx *ctx = s->ctx;
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (frames < 0) return fail(C1_ERR_ARG, "frames must be >= 0");
  if (frames == 0) return C1_OK;
  if (!units || !pcm) return fail(C1_ERR_ARG, "units or pcm is NULL");
  const size_t ub = (size_t)s->channels * C1_UNIT_BYTES;
  if (frames > s->cap_frames) {
    if (s->d_units) { hipFree(s->d_units); hipFree(s->d_pcm); }
    s->d_units = nullptr; s->d_pcm = nullptr; s->cap_frames = 0;
    HIP_TRY(hipMalloc(&s->d_units, ub * (frames + 1)));
    HIP_TRY(hipMalloc(&s->d_pcm, (size_t)s->channels * frames * 512 * sizeof(float)));
    s->cap_frames = frames;
  }
  if (s->have_prev) HIP_TRY(hipMemcpyAsync(s->d_units, s->d_prev, ub, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(s->d_units + ub, units, ub * frames, hipMemcpyHostToDevice, ctx->stream));
  float *dptr[C1_MAX_CHANNELS] = {nullptr, nullptr};
  for (int c = 0; c < s->channels; c++) dptr[c] = s->d_pcm + (size_t)c * s->cap_frames * 512;
  if ((rc = c1_decode_device(ctx, s->d_units + ub, s->channels, frames, s->have_prev ? 1 : 0, dptr))) return rc;
  HIP_TRY(hipMemcpyAsync(s->d_prev, s->d_units + ub * frames, ub, hipMemcpyDeviceToDevice, ctx->stream));
  s->have_prev = true;
  for (int c = 0; c < s->channels; c++) {
    if (!pcm[c]) return fail(C1_ERR_ARG, "pcm[%d] is NULL", c);
    HIP_TRY(hipMemcpyAsync(pcm[c], dptr[c], (size_t)frames * 512 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return C1_OK;
}

int c1_dec_stream_destroy(c1_dec_stream *s) {
  if (!s) return C1_OK;
  hipSetDevice(s->ctx->device);
  hipStreamSynchronize(s->ctx->stream);
  if (s->d_prev) hipFree(s->d_prev);
  if (s->d_units) hipFree(s->d_units);
  if (s->d_pcm) hipFree(s->d_pcm);
  delete s;
  return C1_OK;
}

// ---- formats either side of the path ---------------------------------------------------------------------
int c1_pcm_from_int_device(c1_ctx *ctx, const void *interleaved, int bits, int channels, int64_t samples_per_channel,
                           float *const *pcm) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if ((rc = check_channels(channels))) return rc;
  if (bits != 16 && bits != 24 && bits != 32) return fail(C1_ERR_ARG, "bits must be 16, 24 or 32, got %d", bits);
  if (samples_per_channel < 0) return fail(C1_ERR_ARG, "samples_per_channel must be >= 0");
  if (samples_per_channel == 0) return C1_OK;
  if (!interleaved || !pcm || !pcm[0] || (channels == 2 && !pcm[1])) return fail(C1_ERR_ARG, "NULL buffer");
  c1k_launch_pcm_from_int(interleaved, bits, channels, samples_per_channel, pcm, ctx->stream);
  HIP_TRY(hipGetLastError());
  return C1_OK;
}

int c1_pcm_to_int16_device(c1_ctx *ctx, const float *const *pcm, int channels, int64_t samples_per_channel,
                           int16_t *interleaved) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if ((rc = check_channels(channels))) return rc;
  if (samples_per_channel < 0) return fail(C1_ERR_ARG, "samples_per_channel must be >= 0");
  if (samples_per_channel == 0) return C1_OK;
  if (!interleaved || !pcm || !pcm[0] || (channels == 2 && !pcm[1])) return fail(C1_ERR_ARG, "NULL buffer");
  if (channels == 2 && ((uintptr_t)interleaved & 3)) return fail(C1_ERR_ARG, "stereo int16 output must be 4-byte aligned");
  c1k_launch_pcm_to_int16(pcm, channels, samples_per_channel, interleaved, ctx->stream);
  HIP_TRY(hipGetLastError());
  return C1_OK;
}

int c1_aea_header(const char *title, uint32_t unit_count, int channels, uint8_t out[2048]) {
  if (!out) return fail(C1_ERR_ARG, "out is NULL");
  memset(out, 0, 2048);
  out[0] = 0x00; out[1] = 0x08; out[2] = 0x00; out[3] = 0x00;          // AEA_MAGIC
  if (title) {
    size_t n = strlen(title);
    if (n > 255) n = 255;                                               // AEA_TITLE_SIZE - 1
    memcpy(out + 4, title, n);
  }
  out[260] = (uint8_t)unit_count; out[261] = (uint8_t)(unit_count >> 8);
  out[262] = (uint8_t)(unit_count >> 16); out[263] = (uint8_t)(unit_count >> 24);
  out[264] = (uint8_t)channels;
  return C1_OK;
}

// ---- synthetic input ----------------------------------------------------------------------------------
int c1_generate_device(c1_ctx *ctx, int signal, uint32_t seed, int64_t frames, float *pcm) {
  CTX_GUARD(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (frames < 0) return fail(C1_ERR_ARG, "frames must be >= 0");
  if (frames == 0) return C1_OK;
  if (!pcm || ((uintptr_t)pcm & 15)) return fail(C1_ERR_ARG, "pcm must be a 16-byte aligned device pointer");
  if (seed == 0) return fail(C1_ERR_ARG, "xorshift32 seed must be non-zero");
  if (signal != C1_SIGNAL_WHITE && signal != C1_SIGNAL_PINK_BURSTS && signal != C1_SIGNAL_MIXED && signal != C1_SIGNAL_PARTIALS) return fail(C1_ERR_ARG, "unknown signal %d", signal);
  std::vector<uint32_t> white_states, pink_states;
  if (signal == C1_SIGNAL_WHITE || signal == C1_SIGNAL_MIXED) {
    // one draw per sample: state before frame f = T^(512 f) seed
    const XsMatrix jump = xs_power(512);
    white_states.resize((size_t)frames);
    uint32_t s = seed;
    for (int64_t f = 0; f < frames; f++) { white_states[(size_t)f] = s; s = jump.apply(s); }
  }
  if (signal == C1_SIGNAL_PINK_BURSTS || signal == C1_SIGNAL_MIXED) {
    // 512-frame segments; per 8 frames the generator draws 8*512 + 256 values, so the PRNG state at the
    // start of every segment is the exact continuation; the integrator restarts from 0 there
    const int64_t segs = (frames + 511) / 512;
    const XsMatrix jump = xs_power(64ull * (8 * 512 + 256));
    pink_states.resize((size_t)segs);
    uint32_t s = seed;
    for (int64_t k = 0; k < segs; k++) { pink_states[(size_t)k] = s; s = jump.apply(s); }
  }
  uint32_t *d_states = nullptr;
  const size_t nw = white_states.size(), np = pink_states.size();
  HIP_TRY(hipMalloc(&d_states, (nw + np + 1) * sizeof(uint32_t)));
  hipError_t e = hipSuccess;
  if (nw) e = hipMemcpy(d_states, white_states.data(), nw * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess && np) e = hipMemcpy(d_states + nw, pink_states.data(), np * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    if (signal == C1_SIGNAL_WHITE) c1k_launch_generate_white(d_states, frames, pcm, 15, 0.5, ctx->stream);
    else if (signal == C1_SIGNAL_PINK_BURSTS) c1k_launch_generate_pink(d_states + nw, frames, pcm, 15, ctx->stream);
    else if (signal == C1_SIGNAL_PARTIALS) c1k_launch_generate_sines(frames, pcm, 15, seed, ctx->stream);
    else {
      // mixed corpus (BASELINE configs[3]): 512-frame segments cycling white noise / pink noise with bursts /
      // stationary partials / quiet white noise
      c1k_launch_generate_white(d_states, frames, pcm, 1, 0.5, ctx->stream);
      c1k_launch_generate_pink(d_states + nw, frames, pcm, 2, ctx->stream);
      c1k_launch_generate_sines(frames, pcm, 4, seed, ctx->stream);
      c1k_launch_generate_white(d_states, frames, pcm, 8, 0.02, ctx->stream);
    }
    e = hipStreamSynchronize(ctx->stream);
  }
  hipFree(d_states);
  if (e != hipSuccess) return fail(C1_ERR_HIP, "generate: %s", hipGetErrorString(e));
  return C1_OK;
}

}  // extern "C"
