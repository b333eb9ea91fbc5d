/*
 * atrac1_oracle.c -- CPU restatement of the reference's ATRAC1 hot path (see the
 * header for status: TEST INFRASTRUCTURE ONLY, parity PINNED by tests/golden/).
 *
 * Citations are file:line in aynik/carta1 v1.1.10 (/root/reference in the build
 * container).  F32() marks every place where the reference stores into a
 * Float32Array; everything else is double arithmetic, unfused.
 */
#include "atrac1_oracle.h"

#include <math.h>
#include <string.h>

#include "c1o_tables.inc"
#include "c1o_fdlibm.h"

#define F32(x) ((float)(x))

/* ---- tables: codec/core/constants.js ---------------------------------------------- */

/* SPECS_PER_BFU, BFU_START_LONG, BFU_START_SHORT, BFU_AMOUNTS: constants.js:29-52 */
static const int SPECS[52] = {8, 8, 8, 8, 4,  4,  4,  4,  8,  8,  8,  8,  6,  6,  6,  6,  6,  6,
                              6, 6, 6, 6, 6,  6,  7,  7,  7,  7,  9,  9,  9,  9,  10, 10, 10, 10,
                              12, 12, 12, 12, 12, 12, 12, 12, 20, 20, 20, 20, 20, 20, 20, 20};
static const int START_LONG[52] = {0,   8,   16,  24,  32,  36,  40,  44,  48,  56,  64,  72,  80,
                                   86,  92,  98,  104, 110, 116, 122, 128, 134, 140, 146, 152, 159,
                                   166, 173, 180, 189, 198, 207, 216, 226, 236, 246, 256, 268, 280,
                                   292, 304, 316, 328, 340, 352, 372, 392, 412, 432, 452, 472, 492};
static const int START_SHORT[52] = {0,   32,  64,  96,  8,   40,  72,  104, 12,  44,  76,  108, 20,
                                    52,  84,  116, 26,  58,  90,  122, 128, 160, 192, 224, 134, 166,
                                    198, 230, 141, 173, 205, 237, 150, 182, 214, 246, 256, 288, 320,
                                    352, 384, 416, 448, 480, 268, 300, 332, 364, 396, 428, 460, 492};
static const int BFU_AMOUNTS[8] = {20, 28, 32, 36, 40, 44, 48, 52};
/* WORD_LENGTH_BITS: constants.js:141-143 */
static const int WL_BITS[16] = {0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

/* QMF prototype, constants.js:74-107.  The decimal literals are rounded the way
 * `new Float32Array([...])` rounds them: decimal -> double -> float. */
static float QMF_EVEN[24], QMF_ODD[24];
static int tables_ready;
static void init_tables(void) {
  static const double proto[24] = {
      -0.00001461907, -0.00009205479, -0.000056157569, 0.00030117269, 0.0002422519, -0.00085293897,
      -0.0005205574,  0.0020340169,   0.00078333891,   -0.0042153862, -0.00075614988, 0.0078402944,
      -0.000061169922, -0.01344162,   0.0024626821,    0.021736089,   -0.007801671,  -0.034090221,
      0.01880949,     0.054326009,    -0.043596379,    -0.099384367,  0.13207909,    0.46424159};
  float window[48];
  if (tables_ready) return;
  for (int i = 0; i < 24; i++) {
    float c = F32(proto[i]);
    window[i] = F32((double)c * 2.0);
    window[47 - i] = F32((double)c * 2.0);
  }
  for (int i = 0; i < 24; i++) {
    QMF_EVEN[i] = window[2 * i];
    QMF_ODD[i] = window[2 * i + 1];
  }
  tables_ready = 1;
}

const double *c1o_scale_factors(void) { return C1O_SCALE_FACTORS; }

/* buildBiasedScaleFactorTable: codec/coding/bitallocation.js:46-61 */
void c1o_default_biased_sf(double bias, double out[64]) {
  for (int i = 0; i < 64; i++) out[i] = (bias == 1.0) ? C1O_SCALE_FACTORS[i] : pow(C1O_SCALE_FACTORS[i], bias);
}

void c1o_enc_state_init(c1o_enc_state *s) { memset(s, 0, sizeof *s); }
void c1o_dec_state_init(c1o_dec_state *s) { memset(s, 0, sizeof *s); }

/* ---- QMF: codec/transforms/qmf.js --------------------------------------------------- */

/* qmfAnalysis, qmf.js:19-50.  n = input length; delay (46) is replaced by the new delay. */
static void qmf_analysis(const float *in, int n, float *delay, float *low, float *high) {
  float work[46 + 512];
  memcpy(work, delay, 46 * sizeof(float));
  memcpy(work + 46, in, (size_t)n * sizeof(float));
  for (int i = 0; i < n / 2; i++) {
    double even = 0, odd = 0;
    for (int j = 0; j < 24; j++) { /* qmf.js:39-42: sequential +=, j ascending */
      even += (double)work[2 * i + 47 - 2 * j] * (double)QMF_EVEN[j];
      odd += (double)work[2 * i + 46 - 2 * j] * (double)QMF_ODD[j];
    }
    low[i] = F32(even + odd);
    high[i] = F32(even - odd);
  }
  memcpy(delay, work + n, 46 * sizeof(float)); /* qmf.js:48 */
}

/* qmfSynthesis, qmf.js:60-105.  n = sub-band length; out has 2n samples. */
static void qmf_synthesis(const float *low, const float *high, int n, float *delay, float *out) {
  float work[46 + 512];
  memcpy(work, delay, 46 * sizeof(float));
  for (int i = 0; i < n; i++) { /* qmf.js:78-84 */
    double l = low[i], h = high[i];
    work[46 + 2 * i] = F32(0.5 * (l + h));
    work[46 + 2 * i + 1] = F32(0.5 * (l - h));
  }
  for (int i = 0; i < n; i++) { /* qmf.js:89-102 */
    double s0 = 0, s1 = 0;
    for (int j = 0; j < 24; j++) {
      s0 += (double)work[2 * i + 2 * j] * (double)QMF_EVEN[j];
      s1 += (double)work[2 * i + 2 * j + 1] * (double)QMF_ODD[j];
    }
    out[2 * i] = F32(s1);
    out[2 * i + 1] = F32(s0);
  }
  memcpy(delay, work + 2 * n, 46 * sizeof(float));
}

/* qmfAnalysisStage, codec/pipeline/encoder.js:57-96.  bands = low128 | mid128 | high256 */
void c1o_qmf_analysis_frame(c1o_enc_state *s, const float pcm[512], float bands[512]) {
  float low256[256], high256[256], delayed[39 + 256];
  init_tables();
  qmf_analysis(pcm, 512, s->qmf_low, low256, high256);
  qmf_analysis(low256, 256, s->qmf_mid, bands, bands + 128);
  memcpy(delayed, s->qmf_high, 39 * sizeof(float)); /* encoder.js:84-90 */
  memcpy(delayed + 39, high256, 256 * sizeof(float));
  memcpy(bands + 256, delayed, 256 * sizeof(float));
  memcpy(s->qmf_high, delayed + 256, 39 * sizeof(float));
}

/* ---- FFT: codec/transforms/fft.js:14-68 ------------------------------------------- */

static void fft_inplace(float *re, float *im, int size) {
  int bits = 0;
  while ((1 << bits) < size) bits++;
  for (int i = 0; i < size; i++) { /* fft.js:21-32 */
    int r = 0, t = i;
    for (int b = 0; b < bits; b++) {
      r = (r << 1) | (t & 1);
      t >>= 1;
    }
    if (r > i) {
      float x = re[i]; re[i] = re[r]; re[r] = x;
      x = im[i]; im[i] = im[r]; im[r] = x;
    }
  }
  int stage = 0;
  for (int stride = 2; stride <= size; stride <<= 1, stage++) { /* fft.js:35-66 */
    int half = stride >> 1;
    double wr = C1O_FFT_W[2 * stage], wi = C1O_FFT_W[2 * stage + 1]; /* cos/sin(-2*pi/stride) */
    for (int start = 0; start < size; start += stride) {
      double tr = 1, ti = 0;
      for (int k = 0; k < half; k++) {
        int e = start + k, o = e + half;
        double er = re[e], ei = im[e], orr = re[o], oi = im[o];
        double xr = orr * tr - oi * ti;
        double xi = orr * ti + oi * tr;
        re[e] = F32(er + xr);
        im[e] = F32(ei + xi);
        re[o] = F32(er - xr);
        im[o] = F32(ei - xi);
        double nr = tr * wr - ti * wi; /* fft.js:62-64 */
        ti = tr * wi + ti * wr;
        tr = nr;
      }
    }
  }
}

/* ---- MDCT / IMDCT: codec/transforms/mdct.js ---------------------------------------- */

static const double *mdct_table(int size, int inverse) {
  if (size == 64) return inverse ? C1O_MDCT_INV64 : C1O_MDCT_FWD64;
  if (size == 256) return inverse ? C1O_MDCT_INV256 : C1O_MDCT_FWD256;
  return inverse ? C1O_MDCT_INV512 : C1O_MDCT_FWD512;
}

/* MDCT.transform, mdct.js:54-122: size samples in, size/2 coefficients out */
static void mdct_forward(const float *in, int size, float *out) {
  const double *tab = mdct_table(size, 0);
  int n2 = size >> 1, n4 = size >> 2, n34 = 3 * n4, nfft = n2 >> 1;
  float re[128], im[128];
  for (int i = 0; i < n4; i += 2) { /* mdct.js:76-89 */
    double r = (double)in[n34 - 1 - i] + (double)in[n34 + i];
    double m = (double)in[n4 + i] - (double)in[n4 - 1 - i];
    double c = tab[i], s = tab[i + 1];
    re[i >> 1] = F32(r * c + m * s);
    im[i >> 1] = F32(m * c - r * s);
  }
  for (int i = n4; i < n2; i += 2) { /* mdct.js:91-105 */
    double r = (double)in[n34 - 1 - i] - (double)in[i - n4];
    double m = (double)in[n4 + i] + (double)in[5 * n4 - 1 - i];
    double c = tab[i], s = tab[i + 1];
    re[i >> 1] = F32(r * c + m * s);
    im[i >> 1] = F32(m * c - r * s);
  }
  fft_inplace(re, im, nfft);
  for (int i = 0; i < nfft; i++) { /* mdct.js:110-119 */
    double c = tab[2 * i], s = tab[2 * i + 1], r = re[i], m = im[i];
    out[2 * i] = F32(-r * c - m * s);
    out[n2 - 1 - 2 * i] = F32(-r * s + m * c);
  }
}

/* IMDCT.transform, mdct.js:139-211: size/2 coefficients in, size samples out */
static void mdct_inverse(const float *in, int size, float *out) {
  const double *tab = mdct_table(size, 1);
  int n2 = size >> 1, n4 = size >> 2, n34 = 3 * n4, nfft = n2 >> 1;
  float re[128], im[128];
  for (int i = 0; i < nfft; i++) { /* mdct.js:161-170 */
    double r = -(double)in[2 * i], m = -(double)in[n2 - 1 - 2 * i];
    double c = tab[2 * i], s = tab[2 * i + 1];
    re[i] = F32(m * s + r * c);
    im[i] = F32(m * c - r * s);
  }
  fft_inplace(re, im, nfft);
  for (int i = 0; i < nfft / 2; i++) { /* mdct.js:177-190 */
    int i2 = 2 * i;
    double c = tab[i2], s = tab[i2 + 1], r = re[i], m = im[i];
    double r1 = r * c + m * s, i1 = r * s - m * c;
    out[n34 - 1 - i2] = F32(r1);
    out[n34 + i2] = F32(r1);
    out[n4 + i2] = F32(i1);
    out[n4 - 1 - i2] = F32(-i1);
  }
  for (int i = nfft / 2; i < nfft; i++) { /* mdct.js:192-208 */
    int idx = (i - nfft / 2) * 2 + n4, i2 = 2 * i;
    double c = tab[i2], s = tab[i2 + 1], r = re[i], m = im[i];
    double r1 = r * c + m * s, i1 = r * s - m * c;
    out[n34 - 1 - idx] = F32(r1);
    out[idx - n4] = F32(-r1);
    out[n4 + idx] = F32(i1);
    out[5 * n4 - 1 - idx] = F32(i1);
  }
}

/* ---- transient detection: codec/analysis/transient.js ------------------------------- */

/* performFFT, transient.js:17-35: n band samples -> n/2 magnitudes */
static void fft_magnitudes(const float *x, int n, float *mag) {
  float re[256], im[256];
  memcpy(re, x, (size_t)n * sizeof(float));
  memset(im, 0, (size_t)n * sizeof(float));
  fft_inplace(re, im, n);
  for (int i = 0; i < n / 2; i++) mag[i] = F32(sqrt((double)re[i] * re[i] + (double)im[i] * im[i]));
}

/* test tap: the engine's Math.log (0), exp (1), log1p (2), log10 (3) on an array (c1o_fdlibm.h) */
void c1o_libm(int fn, const double *in, double *out, long n) {
  for (long i = 0; i < n; i++)
    out[i] = fn == 0 ? c1o_fd_log(in[i]) : (fn == 1 ? c1o_fd_exp(in[i]) : (fn == 2 ? c1o_fd_log1p(in[i]) : c1o_fd_log10(in[i])));
}

void c1o_transient_mags(const float bands[512], float mags[256]) {
  fft_magnitudes(bands, 128, mags);
  fft_magnitudes(bands + 128, 128, mags + 64);
  fft_magnitudes(bands + 256, 256, mags + 128);
}

/* calculateSpectralFlatness, transient.js:120-141 */
static double flatness(const float *c, int n) {
  double sum_log = 0, sum_lin = 0;
  int valid = 0;
  for (int i = 0; i < n; i++) {
    double m = fabs((double)c[i]);
    if (m > 1e-10) {
      sum_log += c1o_fd_log(m); /* Math.log as V8 evaluates it: c1o_fdlibm.h */
      sum_lin += m;
      valid++;
    }
  }
  if (valid == 0) return 0;
  double gm = c1o_fd_exp(sum_log / valid), am = sum_lin / valid;
  return am > 1e-10 ? gm / am : 0;
}

/* calculateHighFrequencyRatio, transient.js:149-164 */
static double hf_ratio(const float *c, int n) {
  double lo = 0, hi = 0;
  for (int i = 0; i < n / 2; i++) lo += (double)c[i] * c[i];
  for (int i = n / 2; i < n; i++) hi += (double)c[i] * c[i];
  double tot = lo + hi;
  return tot > 0 ? hi / tot : 0;
}

/* calculateSpectralFeatures + calculateTransientScore, transient.js:63-226 */
double c1o_transient_score(const float *cur, const float *prev, int n) {
  /* calculateSpectralFlux :92-112 */
  double flux = 0, cur_e = 0;
  for (int i = 0; i < n; i++) {
    double cm = fabs((double)cur[i]), pm = fabs((double)prev[i]);
    double d = cm - pm;
    if (d > 0) flux += d;
    cur_e += cm * cm;
  }
  double norm = sqrt(cur_e);
  if (!(norm != 0)) norm = 1e-6; /* `Math.sqrt(e) || 1e-6`: 0 and NaN are falsy */
  flux = flux / norm;
  double flat_change = fabs(flatness(cur, n) - flatness(prev, n));
  double hf_change = fabs(hf_ratio(cur, n) - hf_ratio(prev, n));
  /* calculateEnergyChange :172-189 */
  double ce = 0, pe = 0;
  for (int i = 0; i < n; i++) {
    ce += (double)cur[i] * cur[i];
    pe += (double)prev[i] * prev[i];
  }
  ce = ce > 1e-10 ? ce : 1e-10; /* Math.max(e, 1e-10) */
  pe = pe > 1e-10 ? pe : 1e-10;
  double db = 10 * c1o_fd_log10(ce / pe);
  double e_change = db > 0 ? db : 0;
  /* calculateTransientScore :197-226 */
  double flat_c = sqrt(flat_change);
  double hf_c = c1o_fd_log1p(hf_change * 10) / C1O_LOG1P_10;
  double e_c = e_change / 30 < 1 ? e_change / 30 : 1;
  return (flux + flat_c + hf_c + e_c) / 4;
}

/* detectTransient, transient.js:44-55 */
int c1o_detect_transient(const float *cur, const float *prev, int n, double threshold) {
  return c1o_transient_score(cur, prev, n) > threshold;
}

/* blockSelectorStage, encoder.js:111-152 */
void c1o_block_modes(c1o_enc_state *s, const float bands[512], const c1o_options *o, int modes[3]) {
  if (o->fixed_modes[0] >= 0) {
    modes[0] = o->fixed_modes[0];
    modes[1] = o->fixed_modes[1];
    modes[2] = o->fixed_modes[2];
    return;
  }
  static const int off[3] = {0, 128, 256}, len[3] = {128, 128, 256}, moff[3] = {0, 64, 128};
  float mags[256];
  c1o_transient_mags(bands, mags);
  for (int b = 0; b < 3; b++) {
    (void)off;
    int t = c1o_detect_transient(mags + moff[b], s->prev_mag + moff[b], len[b] / 2, o->threshold);
    modes[b] = t * (b + 1 > 2 ? b + 1 : 2); /* encoder.js:143 */
  }
  memcpy(s->prev_mag, mags, sizeof mags); /* encoder.js:142 */
}

/* ---- mdctStage: encoder.js:170-349 -------------------------------------------------- */

/* applyTailWindowing, encoder.js:309-316 */
static void tail_window(float *samples, float *overlap, int block) {
  int t0 = block - 32;
  for (int i = 0; i < 32; i++) {
    double v = samples[t0 + i];
    overlap[i] = F32(C1O_WINDOW_SHORT[i] * v);
    samples[t0 + i] = F32(v * C1O_WINDOW_SHORT[31 - i]);
  }
}

static void reverse_into(const float *in, int n, float *out) { /* utils.js:42-48 */
  for (int i = 0; i < n; i++) out[i] = in[n - 1 - i];
}

void c1o_mdct_frame(c1o_enc_state *s, float bands[512], const int modes[3], float coefs[512]) {
  static const int off[3] = {0, 128, 256}, len[3] = {128, 128, 256}, wstart[3] = {48, 48, 112};
  for (int b = 0; b < 3; b++) {
    float *x = bands + off[b], *ov = s->overlap[b], *dst = coefs + off[b];
    float spec[256];
    if (modes[b] == 0) { /* transformLongBlock, encoder.js:228-258 */
      int size = b == 2 ? 512 : 256;
      float in[512];
      memset(in, 0, sizeof in);
      memcpy(in + wstart[b], ov, 32 * sizeof(float));
      tail_window(x, ov, len[b]);
      memcpy(in + wstart[b] + 32, x, (size_t)len[b] * sizeof(float));
      mdct_forward(in, size, spec);
      if (b > 0) reverse_into(spec, len[b], dst);
      else memcpy(dst, spec, (size_t)len[b] * sizeof(float));
    } else { /* transformShortBlocks, encoder.js:269-307 */
      int blocks = len[b] / 32;
      for (int k = 0; k < blocks; k++) {
        float in[64];
        memcpy(in, ov, 32 * sizeof(float));
        tail_window(x + 32 * k, ov, 32);
        memcpy(in + 32, x + 32 * k, 32 * sizeof(float));
        mdct_forward(in, 64, spec);
        if (b > 0) reverse_into(spec, 32, dst + 32 * k);
        else memcpy(dst + 32 * k, spec, 32 * sizeof(float));
      }
    }
  }
}

/* ---- bit allocation: codec/coding/bitallocation.js ---------------------------------- */

/* findScaleFactor, bitallocation.js:290-299.  ceil(3*(log2(m)+21)) clamped to [0,63]
 * is the smallest i with m <= SCALE_FACTORS[i] (= 2^(i/3-21)); pinned at all 64
 * boundaries +-4 ulp by tests/golden/find_scale_factor.json. */
int c1o_find_scale_factor(const float *x, int n) {
  double m = 0;
  for (int i = 0; i < n; i++) {
    double a = fabs((double)x[i]);
    if (a > m) m = a;
  }
  if (m == 0) return 0;
  int i = 0;
  while (i < 63 && m > C1O_SCALE_FACTORS[i]) i++;
  return i;
}

/* siftDown, bitallocation.js:314-341 */
static void sift_down(int *hidx, float *hpri, int start, int size) {
  int i = start, iv = hidx[i];
  float pv = hpri[i];
  for (;;) {
    int l = 2 * i + 1, r = l + 1, mi = i;
    float mp = pv;
    if (l < size && hpri[l] > mp) { mi = l; mp = hpri[l]; }
    if (r < size && hpri[r] > mp) mi = r;
    if (mi == i) break;
    hidx[i] = hidx[mi];
    hpri[i] = hpri[mi];
    i = mi;
  }
  hidx[i] = iv;
  hpri[i] = pv;
}

/* DISTORTION_DELTA_FACTORS / WORD_LENGTH_DELTA_BITS, constants.js:163-179 (exact in binary) */
static double ddf(int wl) {
  if (wl == 0) return 2.0 - ldexp(1.0, -WL_BITS[1]);
  return ldexp(1.0, -WL_BITS[wl]) - ldexp(1.0, -WL_BITS[wl + 1]);
}
static int dbits(int wl) { return WL_BITS[wl + 1] - WL_BITS[wl]; }

/* distributeBitsRDO, bitallocation.js:203-281 */
static void distribute_bits(int n, int remaining, const double *bsf, const int *sfi, int *wl) {
  int hidx[52], hsize = 0;
  float hpri[52];
  memset(wl, 0, 52 * sizeof(int));
  for (int b = 0; b < n; b++) {
    if (sfi[b] == 0) continue;
    hidx[hsize] = b;
    hpri[hsize] = F32(bsf[sfi[b]] * ddf(0) / dbits(0));
    hsize++;
  }
  if (hsize == 0) return;
  for (int i = (hsize >> 1) - 1; i >= 0; i--) sift_down(hidx, hpri, i, hsize);
  while (remaining > 0 && hsize > 0) {
    int b = hidx[0], cur = wl[b];
    int cost = dbits(cur) * SPECS[b];
    if (cost > remaining || cost <= 0) { /* :251-258 */
      hidx[0] = hidx[hsize - 1];
      hpri[0] = hpri[hsize - 1];
      hsize--;
      if (hsize > 0) sift_down(hidx, hpri, 0, hsize);
      continue;
    }
    remaining -= cost;
    int nxt = cur + 1;
    wl[b] = nxt;
    if (nxt < 15) { /* :265-270; WORD_LENGTH_DELTA_BITS[nxt] > 0 always for nxt < 15 */
      hpri[0] = F32(bsf[sfi[b]] * ddf(nxt) / dbits(nxt));
      sift_down(hidx, hpri, 0, hsize);
    } else {
      hidx[0] = hidx[hsize - 1];
      hpri[0] = hpri[hsize - 1];
      hsize--;
      if (hsize > 0) sift_down(hidx, hpri, 0, hsize);
    }
  }
}

static const int *bfu_starts(const int modes[3], int b) {
  int band = b >= 36 ? 2 : b >= 20 ? 1 : 0; /* BFU_BAND_BOUNDARIES, constants.js:37 */
  return modes[band] == 0 ? START_LONG : START_SHORT;
}

/* groupIntoBFUs (quantization.js:106-149) + allocateBits (bitallocation.js:74-142) +
 * calculateTotalDistortion (:157-190) */
void c1o_allocate(const float coefs[512], const int modes[3], const double bsf[64], int *nbfu,
                  int wl_out[52], int sfi[52]) {
  float zero_bit[52];
  for (int b = 0; b < 52; b++) {
    sfi[b] = c1o_find_scale_factor(coefs + bfu_starts(modes, b)[b], SPECS[b]);
    zero_bit[b] = sfi[b] > 0 ? F32(bsf[sfi[b]] * 2.0 * SPECS[b]) : 0.0f;
  }
  double best = INFINITY;
  int best_n = -1, wl[52];
  for (int c = 0; c < 8; c++) {
    int n = BFU_AMOUNTS[c];
    int avail = 212 * 8 - 40 - n * 10; /* FRAME_BITS - FRAME_OVERHEAD_BITS - n*BITS_PER_BFU_METADATA */
    distribute_bits(n, avail, bsf, sfi, wl);
    double total = 0;
    for (int b = 0; b < n; b++) {
      int bits = WL_BITS[wl[b]];
      if (bits == 0) { total += (double)zero_bit[b]; continue; }
      if (sfi[b] == 0) continue;
      total += bsf[sfi[b]] * ldexp(1.0, -bits) * SPECS[b];
    }
    for (int b = n; b < 52; b++) total += (double)zero_bit[b];
    if (total < best) {
      best = total;
      best_n = n;
      memcpy(wl_out, wl, sizeof wl);
    }
  }
  if (best_n < 0) { /* :132-139: every candidate's distortion NaN */
    best_n = BFU_AMOUNTS[0];
    memset(wl_out, 0, 52 * sizeof(int));
    memset(sfi, 0, 52 * sizeof(int));
  }
  *nbfu = best_n;
}

/* ---- quantization: codec/coding/quantization.js ------------------------------------ */

/* ECMAScript ToInt32 (what `| 0` does): truncate, then wrap modulo 2^32 */
static int32_t to_int32(double x) {
  if (!isfinite(x)) return 0;
  double t = trunc(x);
  if (t >= -2147483648.0 && t <= 2147483647.0) return (int32_t)t;
  double m = fmod(t, 4294967296.0);
  if (m < 0) m += 4294967296.0;
  return (int32_t)(uint32_t)m;
}

/* quantize, quantization.js:34-56 */
void c1o_quantize_bfu(const float *x, int n, int sfi, int bits, int *out) {
  if (bits == 0 || sfi == 0) {
    memset(out, 0, (size_t)n * sizeof(int));
    return;
  }
  int range = (1 << (bits - 1)) - 1;
  double norm = (double)range / C1O_SCALE_FACTORS[sfi];
  for (int i = 0; i < n; i++) {
    double v = (double)x[i] * norm;
    int32_t y = to_int32(v + (v >= 0 ? 0.5 : -0.5));
    out[i] = y > range ? range : y < -range ? -range : y;
  }
}

/* dequantize, quantization.js:65-78 */
void c1o_dequantize_bfu(const int *q, int n, int sfi, int bits, float *out) {
  if (bits == 0 || sfi == 0) {
    memset(out, 0, (size_t)n * sizeof(float));
    return;
  }
  int range = (1 << (bits - 1)) - 1;
  for (int i = 0; i < n; i++) out[i] = F32(((double)q[i] * C1O_SCALE_FACTORS[sfi]) / (double)range);
}

/* ---- encode() closure: encoder.js:438-450 (+ quantizationStage :365-418) ------------ */

void c1o_encode_frame(c1o_enc_state *s, const float pcm[512], const c1o_options *o, c1o_fields *out) {
  float bands[512], coefs[512];
  memset(out, 0, sizeof *out);
  c1o_qmf_analysis_frame(s, pcm, bands);
  c1o_block_modes(s, bands, o, out->modes);
  c1o_mdct_frame(s, bands, out->modes, coefs);
  c1o_allocate(coefs, out->modes, o->biased_sf, &out->nbfu, out->wl, out->sfi);
  int pos = 0;
  for (int b = 0; b < out->nbfu; b++) {
    c1o_quantize_bfu(coefs + bfu_starts(out->modes, b)[b], SPECS[b], out->sfi[b], WL_BITS[out->wl[b]], out->q + pos);
    pos += SPECS[b];
  }
  for (int b = out->nbfu; b < 52; b++) { /* the closure returns only the first nBfu entries */
    out->wl[b] = 0;
    out->sfi[b] = 0;
  }
}

/* ---- sound unit: codec/io/serialization.js + bitstream.js --------------------------- */

static void put_bits(uint8_t *buf, int *pos, uint32_t v, int n) { /* bitstream.js:15-40, MSB first */
  for (int k = n - 1; k >= 0; k--, (*pos)++)
    if ((v >> k) & 1u) buf[*pos >> 3] |= (uint8_t)(0x80u >> (*pos & 7));
}
static uint32_t get_bits(const uint8_t *buf, int *pos, int n) { /* bitstream.js:49-70 */
  uint32_t v = 0;
  for (int k = 0; k < n; k++) {
    int p = *pos + k;
    if ((p >> 3) >= C1O_UNIT_BYTES) break; /* the reference stops at the end of the buffer and
                                              returns the bits read so far, unshifted */
    v = (v << 1) | ((buf[p >> 3] >> (7 - (p & 7))) & 1u);
  }
  *pos += n;
  return v;
}

/* serializeFrame, serialization.js:41-98 */
void c1o_pack_unit(const c1o_fields *f, uint8_t unit[212]) {
  memset(unit, 0, C1O_UNIT_BYTES);
  int amount = 0;
  while (amount < 8 && BFU_AMOUNTS[amount] != f->nbfu) amount++;
  uint32_t header = ((uint32_t)(2 - f->modes[0]) << 14) | ((uint32_t)(2 - f->modes[1]) << 12) |
                    ((uint32_t)(3 - f->modes[2]) << 10) | ((uint32_t)amount << 5);
  int pos = 0;
  put_bits(unit, &pos, header & 0xffffu, 16);
  for (int b = 0; b < f->nbfu; b++) put_bits(unit, &pos, (uint32_t)f->wl[b], 4);
  for (int b = 0; b < f->nbfu; b++) put_bits(unit, &pos, (uint32_t)f->sfi[b], 6);
  int q = 0;
  for (int b = 0; b < f->nbfu; b++) {
    int bits = WL_BITS[f->wl[b]];
    for (int i = 0; i < SPECS[b]; i++, q++)
      if (bits > 0) put_bits(unit, &pos, (uint32_t)f->q[q] & ((1u << bits) - 1u), bits);
  }
  unit[209] = unit[210] = unit[211] = 0; /* :93-95 */
}

/* deserializeFrame, serialization.js:111-176 */
void c1o_unpack_unit(const uint8_t unit[212], c1o_fields *f) {
  memset(f, 0, sizeof *f);
  int pos = 0;
  uint32_t header = get_bits(unit, &pos, 16);
  f->modes[0] = 2 - (int)((header >> 14) & 3);
  f->modes[1] = 2 - (int)((header >> 12) & 3);
  f->modes[2] = 3 - (int)((header >> 10) & 3);
  f->nbfu = BFU_AMOUNTS[(header >> 5) & 7];
  for (int b = 0; b < f->nbfu; b++) f->wl[b] = (int)get_bits(unit, &pos, 4);
  for (int b = 0; b < f->nbfu; b++) f->sfi[b] = (int)get_bits(unit, &pos, 6);
  int q = 0;
  for (int b = 0; b < f->nbfu; b++) {
    int bits = WL_BITS[f->wl[b]];
    for (int i = 0; i < SPECS[b]; i++, q++) {
      if (bits > 0) {
        uint32_t v = get_bits(unit, &pos, bits);
        f->q[q] = v >= (1u << (bits - 1)) ? (int)v - (1 << bits) : (int)v; /* bitstream.js:78-82 */
      }
    }
  }
}

/* ---- decode() closure: codec/pipeline/decoder.js ------------------------------------ */

/* overlapAdd, mdct.js:230-245, with size 16 and the 32-entry sine window */
static void overlap_add16(const float *prev, const float *curr, float *out) {
  for (int i = 0; i < 16; i++) {
    double w1 = C1O_WINDOW_SHORT[i], w2 = C1O_WINDOW_SHORT[31 - i];
    double p = prev[i], c = curr[15 - i];
    out[i] = F32(p * w2 - c * w1);
    out[31 - i] = F32(p * w1 + c * w2);
  }
}

void c1o_decode_frame(c1o_dec_state *s, const c1o_fields *f, float pcm[512]) {
  static const int off[3] = {0, 128, 256}, len[3] = {128, 128, 256};
  float coefs[512], bands[512];
  init_tables();
  memset(coefs, 0, sizeof coefs);
  /* dequantizationStage, decoder.js:52-98 */
  int q = 0;
  for (int b = 0; b < f->nbfu; b++) {
    int bits = WL_BITS[f->wl[b]];
    if (bits > 0) c1o_dequantize_bfu(f->q + q, SPECS[b], f->sfi[b], bits, coefs + bfu_starts(f->modes, b)[b]);
    q += SPECS[b];
  }
  /* imdctStage, decoder.js:116-330 */
  for (int b = 0; b < 3; b++) {
    int S = len[b];
    float *out = bands + off[b], *tail = s->tail[b];
    float spec[256], inv[512], mid[256];
    if (f->modes[b] == 0) { /* inverseLongBlock :175-233 */
      int size = b == 2 ? 512 : 256;
      if (b > 0) reverse_into(coefs + off[b], S, spec);
      else memcpy(spec, coefs + off[b], (size_t)S * sizeof(float));
      mdct_inverse(spec, size, inv);
      memcpy(mid, inv + size / 4, (size_t)S * sizeof(float));
      overlap_add16(tail, mid, out);
      memcpy(out + 32, mid + 16, (size_t)(S - 32) * sizeof(float));
      memcpy(tail, mid + S - 16, 16 * sizeof(float));
    } else { /* inverseShortBlocks :244-306 */
      float prev[16];
      memcpy(prev, tail, sizeof prev);
      for (int k = 0; k < S / 32; k++) {
        if (b > 0) reverse_into(coefs + off[b] + 32 * k, 32, spec);
        else memcpy(spec, coefs + off[b] + 32 * k, 32 * sizeof(float));
        mdct_inverse(spec, 64, inv);
        memcpy(mid + 32 * k, inv + 16, 32 * sizeof(float));
        overlap_add16(prev, mid + 32 * k, out + 32 * k);
        memcpy(prev, mid + 32 * k + 16, sizeof prev);
      }
      memcpy(tail, mid + S - 16, 16 * sizeof(float));
    }
  }
  /* qmfSynthesisStage, decoder.js:349-389 */
  float delayed[39 + 256], high[256], low256[256];
  memcpy(delayed, s->qmf_high, 39 * sizeof(float));
  memcpy(delayed + 39, bands + 256, 256 * sizeof(float));
  memcpy(high, delayed, 256 * sizeof(float));
  memcpy(s->qmf_high, delayed + 256, 39 * sizeof(float));
  qmf_synthesis(bands, bands + 128, 128, s->qmf_mid, low256);
  qmf_synthesis(low256, high, 256, s->qmf_low, pcm);
}

/* ---- streams: processor.js:119-136 and :193-237 ------------------------------------- */

void c1o_encode_stream(const float *const *pcm, int channels, long frames, const c1o_options *o,
                       c1o_enc_state *states, uint8_t *units) {
  c1o_fields f;
  for (long n = 0; n < frames; n++)
    for (int c = 0; c < channels; c++) {
      c1o_encode_frame(&states[c], pcm[c] + n * 512, o, &f);
      c1o_pack_unit(&f, units + (n * channels + c) * C1O_UNIT_BYTES);
    }
}

void c1o_decode_stream(const uint8_t *units, int channels, long frames, c1o_dec_state *states,
                       float *const *pcm) {
  c1o_fields f;
  for (long n = 0; n < frames; n++)
    for (int c = 0; c < channels; c++) {
      c1o_unpack_unit(units + (n * channels + c) * C1O_UNIT_BYTES, &f);
      c1o_decode_frame(&states[c], &f, pcm[c] + n * 512);
    }
}

/* ---- synthetic signals --------------------------------------- */

static double xorshift_u(uint32_t *s) { /* s^=s<<13; s^=s>>>17; s^=s<<5; u = s/2^32*2-1 */
  uint32_t x = *s;
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  *s = x;
  return ((double)x / 4294967296.0) * 2.0 - 1.0;
}

void c1o_gen_white(uint32_t seed, long n, float *out) {
  uint32_t s = seed;
  for (long i = 0; i < n; i++) out[i] = F32(xorshift_u(&s) * 0.5);
}

void c1o_gen_pinkT(uint32_t seed, long n, float *out) {
  uint32_t s = seed;
  double p = 0;
  for (long i = 0; i < n; i++) {
    double u = xorshift_u(&s);
    p = 0.98 * p + 0.05 * u;
    double v = p;
    if ((i >> 9) % 8 == 5 && (i % 512) >= 256) v += 0.8 * xorshift_u(&s);
    out[i] = F32(v);
  }
}

/* ---- formats either side of the path ------------------------------------------------------- */

/* bin/cli.js:367-404: DataView.getInt16/getInt32 little endian; 24-bit assembled from three bytes with the
 * sign fix `if (sample & 0x800000) sample |= ~0xffffff`; result / 2^(bits-1) stored into a Float32Array */
void c1o_pcm_from_int(const uint8_t *src, int bits, int channels, long samples, float *const *pcm) {
  const int bps = bits / 8;
  for (long i = 0; i < samples; i++)
    for (int c = 0; c < channels; c++) {
      const uint8_t *p = src + (i * channels + c) * bps;
      double v;
      if (bits == 16) v = (double)(int16_t)(uint16_t)(p[0] | (p[1] << 8)) / 32768.0;
      else if (bits == 24) {
        int32_t s = p[0] | (p[1] << 8) | (p[2] << 16);
        if (s & 0x800000) s |= ~0xffffff;
        v = (double)s / 8388608.0;
      } else {
        v = (double)(int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)) / 2147483648.0;
      }
      pcm[c][i] = (float)v;
    }
}

/* processor.js:381-389: Math.max(-1, Math.min(1, x)) (NaN propagates), then x<0 ? x*0x8000 : x*0x7fff,
 * DataView.setInt16 = ToInt16 (NaN -> 0, truncation toward zero; the clamped value is always in range) */
static int16_t wav_sample(float x) {
  double s = (double)x;
  if (isnan(s)) return 0;
  s = s > 1.0 ? 1.0 : s;
  s = s < -1.0 ? -1.0 : s;
  const double v = s < 0 ? s * 32768.0 : s * 32767.0;
  return (int16_t)(int32_t)v;
}
void c1o_pcm_to_int16(const float *const *pcm, int channels, long samples, int16_t *out) {
  for (long i = 0; i < samples; i++)
    for (int c = 0; c < channels; c++) out[i * channels + c] = wav_sample(pcm[c][i]);
}

void c1o_aea_header(const char *title, uint32_t frame_count, int channels, uint8_t out[2048]) {
  memset(out, 0, 2048);
  out[1] = 0x08;                                        /* AEA_MAGIC 00 08 00 00, constants.js:11 */
  size_t n = title ? strlen(title) : 0;
  if (n > 255) n = 255;                                 /* AEA_TITLE_SIZE - 1 */
  if (n) memcpy(out + 4, title, n);
  for (int k = 0; k < 4; k++) out[260 + k] = (uint8_t)(frame_count >> (8 * k));
  out[264] = (uint8_t)channels;
}
