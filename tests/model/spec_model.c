/*
 * spec_model.c -- CPU model of the product's SPECULATIVE binary32 analysis kernel (k_analysis_spec,
 * carta1_amd/csrc/c1_k_spec.hip), operation for operation.  TEST INFRASTRUCTURE: it is neither the
 * reference's algorithm (that is oracle/atrac1_oracle.c) nor part of the product; tests use it to
 *   - check on the CPU that the error bound the kernel attaches to its coefficients really dominates
 *     |binary32 coefficient - reference coefficient| (tests/test_spec_bound.py, no GPU needed), and
 *   - check on the GPU that the kernel computes what this model computes.
 * Every operation is an IEEE binary32 add/mul or a fused multiply-add (fmaf), exactly the VALU
 * instructions the kernel issues, in the same order.  Build: gcc -O2 -ffp-contract=off.
 *
 * The algorithm: two-stage QMF with each 24-tap sum split into two chains that run
 * from the small outer taps towards the centre, radix-4 FFT rounds with three twiddle products per
 * butterfly, fused pre/post twiddles; next to the values it measures three energies per frame
 * (PCM, first-stage low band, pre-twiddled MDCT points of each band) from which the bound is built.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y; } f2;

static float E32[24];                 /* QMF_EVEN (binary32 already, constants.js:94-107) */
static float W32[32];                 /* fl32(WINDOW_SHORT) */
static f2 PRE256[64], PRE512[128], PRE64[16];   /* fl32 of the MDCT sin/cos tables, (cos, sin) per point */
static f2 TWA[256], TWB[256], TWAB[256];        /* radix-4 twiddles by R's table index of wa: [h-1+k] */
static f2 TW2[256];                             /* radix-2 twiddles fl32(fft_tw[h-1+k]) */

/* coefficients of the bound, as build_spec_tables() in carta1_amd/csrc/c1_api.hip computes them:
 * eps_b = cz_b Z_b + cw_b W + cl_b L + eabs */
static float CZ[3], CW[3], CL[3], EABS, CZS[3], CWS[3], CLS[3];

void spec_model_init(const float *even_taps, const double *window, const double *fwd64, const double *fwd256,
                     const double *fwd512, const double *fft_w /* 8 x (cos,sin) */, const float *coef /* cz[3] cw[3] cl[3] eabs, then the same three triples for short blocks */) {
  memcpy(E32, even_taps, sizeof E32);
  for (int i = 0; i < 32; i++) W32[i] = (float)window[i];
  for (int i = 0; i < 16; i++) { PRE64[i].x = (float)fwd64[2 * i]; PRE64[i].y = (float)fwd64[2 * i + 1]; }
  for (int i = 0; i < 64; i++) { PRE256[i].x = (float)fwd256[2 * i]; PRE256[i].y = (float)fwd256[2 * i + 1]; }
  for (int i = 0; i < 128; i++) { PRE512[i].x = (float)fwd512[2 * i]; PRE512[i].y = (float)fwd512[2 * i + 1]; }
  /* the reference's twiddle recurrence (fft.js:44-64), in double, unfused */
  static double tw[256][2];
  int stage = 0;
  for (int h = 1; h <= 128; h <<= 1, stage++) {
    const double wr = fft_w[2 * stage], wi = fft_w[2 * stage + 1];
    double tr = 1.0, ti = 0.0;
    for (int k = 0; k < h; k++) {
      tw[h - 1 + k][0] = tr; tw[h - 1 + k][1] = ti;
      const double nr = tr * wr - ti * wi;
      ti = tr * wi + ti * wr;
      tr = nr;
    }
  }
  for (int i = 0; i < 255; i++) { TW2[i].x = (float)tw[i][0]; TW2[i].y = (float)tw[i][1]; }
  /* radix-4 round over stages h, 2h: wa = tw[h-1+k], wb = tw[2h-1+k], w3 = wa*wb (k < h) */
  for (int h = 4; h <= 16; h <<= 2)
    for (int k = 0; k < h; k++) {
      const double *a = tw[h - 1 + k], *b = tw[2 * h - 1 + k];
      TWA[h - 1 + k].x = (float)a[0]; TWA[h - 1 + k].y = (float)a[1];
      TWB[h - 1 + k].x = (float)b[0]; TWB[h - 1 + k].y = (float)b[1];
      TWAB[h - 1 + k].x = (float)(a[0] * b[0] - a[1] * b[1]);
      TWAB[h - 1 + k].y = (float)(a[0] * b[1] + a[1] * b[0]);
    }
  for (int b = 0; b < 3; b++) { CZ[b] = coef[b]; CW[b] = coef[3 + b]; CL[b] = coef[6 + b]; }
  EABS = coef[9];
  for (int b = 0; b < 3; b++) { CZS[b] = coef[10 + b]; CWS[b] = coef[13 + b]; CLS[b] = coef[16 + b]; }
}

typedef struct {
  float d1[46], d2[46], hb[39];
  float ov[3][32];
  float p_prev, q_prev;       /* PCM / first-stage-low energies of the previous frame */
} spec_state;

void spec_state_init(spec_state *s) { memset(s, 0, sizeof *s); }

/* sum over a wave the way the kernel does: per 16-lane row quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror,
 * row_mirror (every lane of a row ends with the same value), then (row0 + row1) + (row2 + row3) */
static float row_sum(const float *v) {
  float q[4];
  for (int k = 0; k < 4; k++) q[k] = (v[4 * k] + v[4 * k + 1]) + (v[4 * k + 2] + v[4 * k + 3]);
  return (q[0] + q[1]) + (q[2] + q[3]);
}
static float wave_sum(const float *v) { return (row_sum(v) + row_sum(v + 16)) + (row_sum(v + 32) + row_sum(v + 48)); }

/* one decimating QMF output pair.  w = &work[2 i]: even uses w[47 - 2 j] * E[j], odd uses w[46 - 2 j] * E[23 - j] */
static void qmf_pair(const float *w, float *low, float *high) {
  /* even: chain A over j = 0..11, chain B over j = 23..13, centre tap j = 12 last */
  float a = E32[0] * w[47];
  for (int j = 1; j <= 11; j++) a = fmaf(E32[j], w[47 - 2 * j], a);
  float b = E32[23] * w[1];
  for (int j = 22; j >= 13; j--) b = fmaf(E32[j], w[47 - 2 * j], b);
  float ev = fmaf(E32[12], w[23], a + b);
  /* odd: tap index m = 23 - j, sample w[46 - 2 j] = w[2 m]: chain A over m = 0..11, B over m = 23..13, centre m = 12 */
  float c = E32[0] * w[0];
  for (int m = 1; m <= 11; m++) c = fmaf(E32[m], w[2 * m], c);
  float d = E32[23] * w[46];
  for (int m = 22; m >= 13; m--) d = fmaf(E32[m], w[2 * m], d);
  float od = fmaf(E32[12], w[24], c + d);
  *low = ev + od;
  *high = ev - od;
}

static f2 cmul(f2 x, f2 w) {   /* (x.x + i x.y)(w.x + i w.y): two products rounded, two fused */
  f2 r;
  r.x = fmaf(x.x, w.x, -(x.y * w.y));
  r.y = fmaf(x.x, w.y, x.y * w.x);
  return r;
}
static f2 cadd(f2 a, f2 b) { f2 r = {a.x + b.x, a.y + b.y}; return r; }
static f2 csub(f2 a, f2 b) { f2 r = {a.x - b.x, a.y - b.y}; return r; }

static int bitrev(int k, int bits) { int r = 0; for (int b = 0; b < bits; b++) { r = (r << 1) | (k & 1); k >>= 1; } return r; }

/* long-block MDCT of one band in binary32; in: N samples (zero padded long-block input); out: N/2 coefficients
 * (not reversed); returns the energy of the pre-twiddled points, summed per lane the way the kernel sums it:
 * lane_energy[g] for g < n4/4 lanes (4 points per lane) */
static void mdct_long_f32(const float *in, int N, float *out, float *lane_energy) {
  const int n4 = N / 4, n2 = N / 2, n34 = 3 * n4, nfft = n4, lanes = nfft / 4;
  const f2 *pre = N == 512 ? PRE512 : PRE256;
  const int bits = N == 512 ? 7 : 6;
  f2 z[128];
  /* round A: lane g owns positions 4g..4g+3 = points k = r + q*bitrev2(j), r = bitrev(g) */
  for (int g = 0; g < lanes; g++) {
    const int r = bitrev(g, bits - 2), q = nfft / 4;
    f2 x[4];
    float enx = 0, eny = 0;
    for (int j = 0; j < 4; j++) {
      const int jp = ((j & 1) << 1) | (j >> 1);
      const int k = r + q * jp, i = 2 * k;
      float rr, mm;
      if (i < n4) { rr = in[n34 - 1 - i] + in[n34 + i]; mm = in[n4 + i] - in[n4 - 1 - i]; }
      else { rr = in[n34 - 1 - i] - in[i - n4]; mm = in[n4 + i] + in[5 * n4 - 1 - i]; }
      /* the kernel takes 0 for the operands that are structural zero padding instead of reading them: same values */
      const f2 t = pre[k];
      x[j].x = fmaf(rr, t.x, mm * t.y);
      x[j].y = fmaf(mm, t.x, -(rr * t.y));
      enx = j == 0 ? x[j].x * x[j].x : fmaf(x[j].x, x[j].x, enx);
      eny = j == 0 ? x[j].y * x[j].y : fmaf(x[j].y, x[j].y, eny);
    }
    lane_energy[g] = enx + eny;
    /* stages 1, 2 without products */
    const f2 t0 = cadd(x[0], x[1]), t1 = csub(x[0], x[1]), t2 = cadd(x[2], x[3]), t3 = csub(x[2], x[3]);
    f2 y1 = {t1.x + t3.y, t1.y - t3.x}, y3 = {t1.x - t3.y, t1.y + t3.x};
    z[4 * g] = cadd(t0, t2); z[4 * g + 1] = y1; z[4 * g + 2] = csub(t0, t2); z[4 * g + 3] = y3;
  }
  /* rounds B (stages 4, 8) and C (16, 32): radix-4 with wa, wb, wa*wb */
  for (int h = 4; h <= 16; h <<= 2) {
    for (int base = 0; base < nfft; base += 4 * h)
      for (int k = 0; k < h; k++) {
        f2 *p = z + base + k;
        const f2 x0 = p[0], y1 = cmul(p[h], TWA[h - 1 + k]), y2 = cmul(p[2 * h], TWB[h - 1 + k]), y3 = cmul(p[3 * h], TWAB[h - 1 + k]);
        const f2 t0 = cadd(x0, y1), t1 = csub(x0, y1), t2 = cadd(y2, y3), t3 = csub(y2, y3);
        p[0] = cadd(t0, t2);
        p[2 * h] = csub(t0, t2);
        p[h].x = t1.x + t3.y; p[h].y = t1.y - t3.x;
        p[3 * h].x = t1.x - t3.y; p[3 * h].y = t1.y + t3.x;
      }
  }
  if (nfft == 128)   /* round D: stage 64 */
    for (int k = 0; k < 64; k++) {
      const f2 y = cmul(z[k + 64], TW2[63 + k]), e = z[k];
      z[k] = cadd(e, y); z[k + 64] = csub(e, y);
    }
  for (int i = 0; i < nfft; i++) {
    const f2 t = pre[i];
    out[2 * i] = -fmaf(z[i].x, t.x, z[i].y * t.y);
    out[n2 - 1 - 2 * i] = fmaf(z[i].y, t.x, -(z[i].x * t.y));
  }
}

/* one 64-sample MDCT (16-point transform) in binary32: in64 = the block's input, out = 32 coefficients (not reversed);
 * lane_energy[0..3] = the partial energies of the four lanes that share the block */
static void mdct_short_f32(const float *in, float *out, float *lane_energy) {
  const int n4 = 16, n34 = 48, n2 = 32, nfft = 16;
  f2 z[16];
  for (int g = 0; g < 4; g++) {
    const int r = bitrev(g, 2), q = 4;
    f2 x[4];
    float enx = 0, eny = 0;
    for (int j = 0; j < 4; j++) {
      const int jp = ((j & 1) << 1) | (j >> 1);
      const int k = r + q * jp, i = 2 * k;
      float rr, mm;
      if (i < n4) { rr = in[n34 - 1 - i] + in[n34 + i]; mm = in[n4 + i] - in[n4 - 1 - i]; }
      else { rr = in[n34 - 1 - i] - in[i - n4]; mm = in[n4 + i] + in[5 * n4 - 1 - i]; }
      const f2 t = PRE64[k];
      x[j].x = fmaf(rr, t.x, mm * t.y);
      x[j].y = fmaf(mm, t.x, -(rr * t.y));
      enx = j == 0 ? x[j].x * x[j].x : fmaf(x[j].x, x[j].x, enx);
      eny = j == 0 ? x[j].y * x[j].y : fmaf(x[j].y, x[j].y, eny);
    }
    lane_energy[g] = enx + eny;
    const f2 t0 = cadd(x[0], x[1]), t1 = csub(x[0], x[1]), t2 = cadd(x[2], x[3]), t3 = csub(x[2], x[3]);
    f2 y1 = {t1.x + t3.y, t1.y - t3.x}, y3 = {t1.x - t3.y, t1.y + t3.x};
    z[4 * g] = cadd(t0, t2); z[4 * g + 1] = y1; z[4 * g + 2] = csub(t0, t2); z[4 * g + 3] = y3;
  }
  for (int k = 0; k < 4; k++) {
    f2 *p = z + k;
    const f2 x0 = p[0], y1 = cmul(p[4], TWA[3 + k]), y2 = cmul(p[8], TWB[3 + k]), y3 = cmul(p[12], TWAB[3 + k]);
    const f2 t0 = cadd(x0, y1), t1 = csub(x0, y1), t2 = cadd(y2, y3), t3 = csub(y2, y3);
    p[0] = cadd(t0, t2);
    p[8] = csub(t0, t2);
    p[4].x = t1.x + t3.y; p[4].y = t1.y - t3.x;
    p[12].x = t1.x - t3.y; p[12].y = t1.y + t3.x;
  }
  for (int i = 0; i < nfft; i++) {
    const f2 t = PRE64[i];
    out[2 * i] = -fmaf(z[i].x, t.x, z[i].y * t.y);
    out[n2 - 1 - 2 * i] = fmaf(z[i].y, t.x, -(z[i].x * t.y));
  }
}

/* one frame: pcm[512] -> coefs[512] (bands 1, 2 reversed as the reference stores them), eps[3].
 * all_short = 0: fixed block modes [0,0,0]; 1: every band in short blocks */
void spec_model_frame(spec_state *s, const float *pcm, float *coefs, float *eps, float *bands_out, int all_short) {
  float w1[46 + 512], low1[256], high1[256], w2[46 + 256], band[512];
  float lane[64];
  memcpy(w1, s->d1, sizeof s->d1);
  memcpy(w1 + 46, pcm, 512 * sizeof(float));
  for (int l = 0; l < 64; l++) {
    const float *a = pcm + 4 * l, *b = pcm + 256 + 4 * l;
    /* packed: two running sums, (x, y) halves of the 2-vectors the kernel squares */
    float px = a[0] * a[0], py = a[1] * a[1];
    px = fmaf(a[2], a[2], px); py = fmaf(a[3], a[3], py);
    px = fmaf(b[0], b[0], px); py = fmaf(b[1], b[1], py);
    px = fmaf(b[2], b[2], px); py = fmaf(b[3], b[3], py);
    lane[l] = px + py;
  }
  const float P = wave_sum(lane);
  for (int i = 0; i < 256; i++) qmf_pair(w1 + 2 * i, &low1[i], &high1[i]);
  memcpy(s->d1, w1 + 512, sizeof s->d1);
  for (int l = 0; l < 64; l++) {
    const float *a = low1 + 4 * l;
    float px = a[0] * a[0], py = a[1] * a[1];
    px = fmaf(a[2], a[2], px); py = fmaf(a[3], a[3], py);
    lane[l] = px + py;
  }
  const float Q = wave_sum(lane);
  memcpy(w2, s->d2, sizeof s->d2);
  memcpy(w2 + 46, low1, sizeof low1);
  for (int i = 0; i < 128; i++) qmf_pair(w2 + 2 * i, &band[i], &band[128 + i]);
  memcpy(s->d2, w2 + 256, sizeof s->d2);
  memcpy(band + 256, s->hb, sizeof s->hb);
  memcpy(band + 256 + 39, high1, (256 - 39) * sizeof(float));
  memcpy(s->hb, high1 + 217, sizeof s->hb);
  if (bands_out) memcpy(bands_out, band, sizeof band);
  const float W = sqrtf(P + s->p_prev), L = sqrtf(Q + s->q_prev);
  s->p_prev = P; s->q_prev = Q;
  float zen[3];
  for (int b = 0; b < 3; b++) {
    const int len = b == 2 ? 256 : 128, N = 2 * len, ws = b == 2 ? 112 : 48, off = b == 0 ? 0 : (b == 1 ? 128 : 256);
    if (!all_short) {
      float in[512], spec[256], le[32];
      memset(in, 0, sizeof in);
      memcpy(in + ws, s->ov[b], 32 * sizeof(float));
      memcpy(in + ws + 32, band + off, (size_t)len * sizeof(float));
      for (int i = 0; i < 32; i++) {
        const float v = band[off + len - 32 + i];
        s->ov[b][i] = W32[i] * v;
        in[ws + 32 + len - 32 + i] = v * W32[31 - i];
      }
      mdct_long_f32(in, N, spec, le);
      float rows[32];
      memset(rows, 0, sizeof rows);
      memcpy(rows, le, (size_t)(N / 16) * sizeof(float));
      zen[b] = b == 2 ? row_sum(rows) + row_sum(rows + 16) : row_sum(rows);
      if (b == 0) memcpy(coefs, spec, 128 * sizeof(float));
      else for (int i = 0; i < len; i++) coefs[off + i] = spec[len - 1 - i];
    } else {
      /* E[s] = W[s & 31] x[s] behind the previous frame's overlap, H[s] = x[s] W[31 - (s & 31)] (encoder.js:269-307) */
      float E[32 + 256], H[256];
      memcpy(E, s->ov[b], 32 * sizeof(float));
      for (int i = 0; i < len; i++) { E[32 + i] = W32[i & 31] * band[off + i]; H[i] = band[off + i] * W32[31 - (i & 31)]; }
      memcpy(s->ov[b], E + len, 32 * sizeof(float));
      float zmax = 0.0f;
      for (int q = 0; q < len / 32; q++) {
        float in[64], spec[32], le[4];
        memcpy(in, E + 32 * q, 32 * sizeof(float));
        memcpy(in + 32, H + 32 * q, 32 * sizeof(float));
        mdct_short_f32(in, spec, le);
        const float blk = (le[0] + le[1]) + (le[2] + le[3]);
        zmax = fmaxf(zmax, blk);
        for (int i = 0; i < 32; i++) coefs[off + 32 * q + i] = b == 0 ? spec[i] : spec[31 - i];
      }
      zen[b] = zmax;
    }
  }
  /* the bound; the kernel takes the square roots with v_sqrt_f32 (1 ulp), which the bound's theta covers */
  for (int b = 0; b < 3; b++)
    eps[b] = all_short ? fmaf(CZS[b], sqrtf(zen[b]), fmaf(CWS[b], W, fmaf(CLS[b], L, EABS)))
                       : fmaf(CZ[b], sqrtf(zen[b]), fmaf(CW[b], W, fmaf(CL[b], L, EABS)));
}

void spec_model_stream(const float *pcm, long frames, float *coefs, float *eps, float *bands, int all_short) {
  spec_state s;
  spec_state_init(&s);
  for (long f = 0; f < frames; f++)
    spec_model_frame(&s, pcm + 512 * f, coefs + 512 * f, eps + 3 * f, bands ? bands + 512 * f : 0, all_short);
}
