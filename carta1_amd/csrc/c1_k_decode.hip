// c1_k_decode.hip -- deserializeFrame + the decode() closure (decoder.js:408-411): one wave per run of units
// k_decode<double>: the reference's arithmetic (binary64 operations, binary32 at every typed-array store): decoded PCM
//                   bit-identical to the reference.
// k_decode<float>:  the same computation in binary32 (opt-in, c1_ctx_set_decode_precision): no conversions, half the
//                   LDS for the synthesis windows; the PCM differs from the reference's by rounding noise (RMS ~1e-8 at
//                   full scale, against the 1e-5 the task allows; tests/test_gpu_decode32.py).
#include "c1_device.h"

namespace {

template <typename R> struct Pair2;
template <> struct Pair2<double> { typedef double2 type; };
template <> struct Pair2<float> { typedef float2 type; };
// (cos, sin) table pair at entry `index` of the double table at `base64` / its binary32 twin at `base32`
template <typename R>
__device__ __forceinline__ typename Pair2<R>::type table_pair_r(TablesRsrc RT, int offset) {
  if constexpr (std::is_same<R, double>::value) return table_pair(RT, offset);
  else {
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(RT, offset, 0, 0);
    float2 d;
    __builtin_memcpy(&d, &v, sizeof d);
    return d;
  }
}
// one radix-2 butterfly of fft.js:46-60 in the arithmetic R (R = double: r2_butterfly of c1_device.h)
template <typename R>
__device__ __forceinline__ void r2_bf(float2 &e, float2 &o, const typename Pair2<R>::type w) {
  const R er = e.x, ei = e.y, orr = o.x, oi = o.y;
  const R xr = orr * w.x - oi * w.y;
  const R xi = orr * w.y + oi * w.x;
  e = make_float2((float)(er + xr), (float)(ei + xi));
  o = make_float2((float)(er - xr), (float)(ei - xi));
}

// =====================================================================================================
// k_decode : deserializeFrame + decode() closure (decoder.js:408-411)
// =====================================================================================================
template <typename R>
struct alignas(16) DecodeLds {
  R d1[46];             // stage-1 synthesis delay (qmfDelays.lowBand)
  R d2[46];             // stage-2 synthesis delay (qmfDelays.midBand)
  float dhi[39];        // high-band delay
  float tail[48];       // last 16 IMDCT samples per band (imdctOverlap tails, decoder.js:227-230)
  uint32_t words[56];   // the unit as big-endian words
  uint32_t desc[52];    // per BFU: bits(5) | sfi(6) << 5 | mantissa bit offset << 11 (may exceed the unit for arbitrary bytes)
  R sf_tab[64];         // SCALE_FACTORS and RN(1/range): lane-varying lookups, kept in LDS (a global load per
  R inv_tab[16];        // coefficient would cost a cache round trip each)
  R step[52];           // per BFU of the unit: SF * RN(1 / range), 0 for a silent BFU (dq_step; the binary32 decoder always)
  R wtab[32];           // WINDOW_SHORT (the overlap-add's lane-varying lookups: a global load each, waited for on the spot)
  uint32_t late[8][64]; // all-long frames: imdct_r4's end-of-transform values per lane (see there)
  int16_t dshort[52];   // BFU_START_SHORT[b] - (first slot of b): where a short band's coefficients go, relative to slot order
  union alignas(16) {
    float coef[512];    // dequantized coefficients: dead once the IMDCT pre-twiddle has read them
    float band[512];    // reconstructed bands: born at the overlap-add
  } cb;
  union alignas(16) {
    // IMDCT: points (4 pad per 16), then the outputs in the same memory: the post-twiddle writes `mid` from registers after the
    // last round has read its points (one wave: LDS operations execute in issue order)
    struct { union alignas(16) { float2 z[320]; float mid[512]; } zz; } m;
    struct { alignas(16) R w2[454]; } q2;                        // stage-2 synthesis work buffer (padded 2 per 4)
    struct { alignas(16) R w1[698]; } q1;                        // stage-1 synthesis work buffer (padded 2 per 8), after w2 is consumed
  } u;
};

// qmf_synthesis_core of c1_device.h in the arithmetic R (same window layout, element type R)
template <typename R, int D, int S>
__device__ __forceinline__ void qmf_synth_r(const R *w, int lane, TablesPtr T, R (&s0)[D], R (&s1)[D]) {
  if constexpr (std::is_same<R, double>::value) qmf_synthesis_core<D, S>(w, lane, T, s0, s1);
  else {
    typedef typename Pair2<R>::type pair;
#pragma unroll
    for (int d = 0; d < D; d++) s0[d] = s1[d] = 0.0f;
#pragma unroll
    for (int u = 0; u <= 22 + D; ++u) {
      const pair x = *reinterpret_cast<const pair *>(w + (2 * D + 2) * lane + (2 * u + 2 * ((2 * u) >> S)));
#pragma unroll
      for (int d = 0; d < D; d++) {
        const int j = u - d;
        if (j >= 0 && j < 24) {
          s0[d] = __builtin_fmaf(x.x, T->tap32[j], s0[d]);
          s1[d] = __builtin_fmaf(x.y, T->tap32[23 - j], s1[d]);
        }
      }
    }
  }
}

__device__ __forceinline__ uint32_t get_bits_be(const uint32_t *words, int pos, int nbits) {
  // unpackBits (bitstream.js:49-70): stops at the end of the 212-byte buffer and returns what it has
  const int avail = C1_UNIT_BYTES * 8 - pos;
  if (avail <= 0 || nbits == 0) return 0u;
  const int nb = nbits < avail ? nbits : avail;
  const int w = pos >> 5, o = pos & 31;
  const uint64_t two = ((uint64_t)words[w] << 32) | (uint64_t)words[w + 1];
  return (uint32_t)((two >> (64 - o - nb)) & ((1ull << nb) - 1ull));
}

// ---- inverse MDCT in radix-4 rounds: the decoder's mirror of mdct_long_r4 / mdct_mixed_r4 -----------------------
// Lanes 0..15 band 0, 16..31 band 1, 32..63 band 2, four points per lane; short bands stop after round B.
// Pre-twiddle of point i reads coefficients 2i and n2-1-2i (mdct.js:161-170; bands 1,2 arrive spectrally
// reversed, decoder.js:183-186); the post-twiddle keeps the middle half the decoder uses (decoder.js:191-199).
struct IMixGeometry {
  int ja[4], jb[4], pre_tab[4];
  int za, zb, zc, zd, twb, twc, twd;
  int post_tab[4], ox[4], oy[4];
  bool is_long, band2;
};
template <typename R>
__device__ __forceinline__ IMixGeometry imix_geometry(int lane, const FrameModes &M) {
  constexpr bool F32 = std::is_same<R, float>::value;
  constexpr int kPair = F32 ? 8 : 16;                          // bytes per (cos, sin) / twiddle pair
  IMixGeometry G;
  const int band = lane < 16 ? 0 : (lane < 32 ? 1 : 2);
  const int g = lane - (band == 0 ? 0 : (band == 1 ? 16 : 32));
  const bool lng = M.mode_of_band(band) == 0;
  const int nfft = lng ? (band == 2 ? 128 : 64) : 16, q4 = nfft / 4, n2 = 2 * nfft;
  const int r = lng ? bitrev(g, band == 2 ? 5 : 4) : bitrev(g & 3, 2);
  const int blk = lng ? 0 : (g >> 2);
  const int obase = (band == 0 ? 0 : (band == 1 ? 128 : 256)) + 32 * blk;       // coefficients in, samples out
  const int tab_base = F32 ? (lng ? (band == 2 ? (int)offsetof(C1DevTables, inv32_512) : (int)offsetof(C1DevTables, inv32_256))
                                  : (int)offsetof(C1DevTables, inv32_64))
                           : (lng ? (band == 2 ? (int)offsetof(C1DevTables, mdct_inv512) : (int)offsetof(C1DevTables, mdct_inv256))
                                  : (int)offsetof(C1DevTables, mdct_inv64));
  const int tw_base = F32 ? (int)offsetof(C1DevTables, tw32) : (int)offsetof(C1DevTables, fft_tw);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int jp = ((j & 1) << 1) | (j >> 1);
    const int i = r + q4 * jp;                              // position 4g+j holds point bitrev(4g+j)
    const int j0 = 2 * i, j1 = n2 - 1 - 2 * i;
    G.ja[j] = obase + (band > 0 ? n2 - 1 - j0 : j0);
    G.jb[j] = obase + (band > 0 ? n2 - 1 - j1 : j1);
    G.pre_tab[j] = tab_base + kPair * i;
  }
  const int pbase = band == 0 ? 0 : (band == 1 ? 64 : 128);
  G.za = zslot(pbase + 4 * g);
  G.zb = zslot(pbase + 16 * (g >> 2) + (g & 3));
  G.twb = tw_base + kPair * (3 + (g & 3));
  G.zc = zslot(pbase + 64 * (g >> 4) + (g & 15));
  G.twc = tw_base + kPair * (15 + (g & 15));
  G.zd = zslot(128 + (g & 31));
  G.twd = tw_base + kPair * (63 + (g & 31));
  G.is_long = lng;
  G.band2 = band == 2;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i = lng ? (band == 2 ? g + (j == 1 ? 64 : (j == 2 ? 32 : (j == 3 ? 96 : 0))) : g + 16 * j) : (g & 3) + 4 * j;
    const int idx = (i < nfft / 2) ? 2 * i : (2 * (i - nfft / 2) + nfft);
    G.post_tab[j] = tab_base + kPair * i;
    G.ox[j] = obase + n2 - 1 - idx;
    G.oy[j] = obase + idx;
  }
  return G;
}

// coef: 512 dequantized coefficients; z: 320 slots; mid: 512 outputs.  any_long / band2_long are wave-uniform.
// late: when not null, the values of the END of the transform (where the outputs go, the post-twiddle pairs) of an
// all-long frame, one word per lane and value: ox | oy << 16 for j = 0..3 at late[64 j + lane], post_tab at late[256 + 64 j + lane].
// Carried in registers through the frame loop they push five other values to scratch, and a scratch reload is a
// vector-memory load: it waits on the counter the unit prefetch and the PCM stores share (see k_decode).
template <typename RT_>
__device__ __forceinline__ void imdct_r4(const float *coef, float2 *z, float *mid, const IMixGeometry &G, bool any_long,
                                         bool band2_long, TablesPtr T, TablesRsrc R, const uint32_t *late = nullptr) {
  typedef RT_ real;
  typedef typename Pair2<real>::type pair;
  constexpr bool F32 = std::is_same<real, float>::value;
  constexpr int kPair = F32 ? 8 : 16;
  float2 x[4];
  // Lane-varying table values are cache round trips: every round asks for the values of the NEXT round before it starts
  // computing (as the encoder's cores do), so the loads are in flight during the arithmetic and the LDS exchange
  const pair bwa = table_pair_r<real>(R, G.twb), bwb = table_pair_r<real>(R, G.twb + 4 * kPair), bwc = table_pair_r<real>(R, G.twb + 8 * kPair);
  {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const real r = -(real)coef[G.ja[j]], mm = -(real)coef[G.jb[j]];
      const pair t = table_pair_r<real>(R, G.pre_tab[j]);
      x[j] = make_float2((float)(mm * t.y + r * t.x), (float)(mm * t.x - r * t.y));
    }
    pair w0, w1, w2;
    w0.x = (real)T->fft_tw[0][0]; w0.y = (real)T->fft_tw[0][1];
    w1.x = (real)T->fft_tw[1][0]; w1.y = (real)T->fft_tw[1][1];
    w2.x = (real)T->fft_tw[2][0]; w2.y = (real)T->fft_tw[2][1];
    if (F32 || __all(r2_unit_ok(x[0], x[1]) && r2_unit_ok(x[2], x[3]))) { r2_butterfly_unit(x[0], x[1]); r2_butterfly_unit(x[2], x[3]); }
    else { r2_bf<real>(x[0], x[1], w0); r2_bf<real>(x[2], x[3], w0); }
    if (F32 || __all(r2_unit_ok(x[0], x[2]))) r2_butterfly_unit(x[0], x[2]);
    else r2_bf<real>(x[0], x[2], w1);
    r2_bf<real>(x[1], x[3], w2);
    float4 *dst = reinterpret_cast<float4 *>(z + G.za);
    dst[0] = make_float4(x[0].x, x[0].y, x[1].x, x[1].y);
    dst[1] = make_float4(x[2].x, x[2].y, x[3].x, x[3].y);
  }
  // round C's twiddles (long bands), or the post-twiddle pairs of a frame that ends after round B
  pair n0, n1, n2, n3;
  if (any_long) { n0 = table_pair_r<real>(R, G.twc); n1 = table_pair_r<real>(R, G.twc + 16 * kPair); n2 = table_pair_r<real>(R, G.twc + 32 * kPair); n3 = n0; }
  else { n0 = table_pair_r<real>(R, G.post_tab[0]); n1 = table_pair_r<real>(R, G.post_tab[1]); n2 = table_pair_r<real>(R, G.post_tab[2]); n3 = table_pair_r<real>(R, G.post_tab[3]); }
  wave_fence();
  {
    float2 *p = z + G.zb;
    x[0] = p[0]; x[1] = p[4]; x[2] = p[8]; x[3] = p[12];
    r2_bf<real>(x[0], x[1], bwa); r2_bf<real>(x[2], x[3], bwa);
    r2_bf<real>(x[0], x[2], bwb); r2_bf<real>(x[1], x[3], bwc);
    if (G.is_long) { p[0] = x[0]; p[4] = x[1]; p[8] = x[2]; p[12] = x[3]; }
  }
  pair t0, t1, t2, t3;                                       // the post-twiddle pairs
  if (any_long) {
    // round D's twiddles (band 2), in flight during round C
    pair dwa, dwb;
    if (band2_long) { dwa = table_pair_r<real>(R, G.twd); dwb = table_pair_r<real>(R, G.twd + 32 * kPair); }
    wave_fence();
    if (G.is_long) {
      float2 *p = z + G.zc;
      x[0] = p[0]; x[1] = p[20]; x[2] = p[40]; x[3] = p[60];
      r2_bf<real>(x[0], x[1], n0); r2_bf<real>(x[2], x[3], n0);
      r2_bf<real>(x[0], x[2], n1); r2_bf<real>(x[1], x[3], n2);
      if (G.band2) { p[0] = x[0]; p[20] = x[1]; p[40] = x[2]; p[60] = x[3]; }
    }
    if (late) {
      t0 = table_pair_r<real>(R, (int)late[256]); t1 = table_pair_r<real>(R, (int)late[320]);
      t2 = table_pair_r<real>(R, (int)late[384]); t3 = table_pair_r<real>(R, (int)late[448]);
    } else {
      t0 = table_pair_r<real>(R, G.post_tab[0]); t1 = table_pair_r<real>(R, G.post_tab[1]);
      t2 = table_pair_r<real>(R, G.post_tab[2]); t3 = table_pair_r<real>(R, G.post_tab[3]);
    }
    if (band2_long) {
      wave_fence();
      if (G.band2) {
        const float2 *p = z + G.zd;
        x[0] = p[0]; x[1] = p[80]; x[2] = p[40]; x[3] = p[120];
        r2_bf<real>(x[0], x[1], dwa); r2_bf<real>(x[2], x[3], dwb);
      }
    }
  } else { t0 = n0; t1 = n1; t2 = n2; t3 = n3; }
  wave_fence();                                          // `mid` is the memory of `z`: every point has been read by now
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const pair t = j == 0 ? t0 : (j == 1 ? t1 : (j == 2 ? t2 : t3));
    const real rr = x[j].x, ii = x[j].y;
    const uint32_t oo = late ? late[64 * j] : ((uint32_t)G.ox[j] | ((uint32_t)G.oy[j] << 16));
    mid[oo & 0xffffu] = (float)(rr * t.x + ii * t.y);       // mdct.js:177-208
    mid[oo >> 16] = (float)(rr * t.y - ii * t.x);
  }
}

template <typename R>
__global__ __launch_bounds__(C1_WAVE, (std::is_same<R, float>::value ? 4 : 3)) void k_decode(C1DecodeLaunch L) {
  typedef R real;
  typedef typename Pair2<R>::type pair;
  constexpr bool F32 = std::is_same<R, float>::value;
  __shared__ DecodeLds<R> S;
  const int lane0 = threadIdx.x;
  int lane = lane0;
  const int ch = blockIdx.x % L.channels;
  const int64_t f0 = (int64_t)(blockIdx.x / L.channels) * L.run_frames;
  float *__restrict__ pcm = L.pcm[ch];

  for (int i = lane; i < 46; i += 64) { S.d1[i] = 0; S.d2[i] = 0; }
  for (int i = lane; i < 39; i += 64) S.dhi[i] = 0.0f;
  for (int i = lane; i < 48; i += 64) S.tail[i] = 0.0f;
  if (lane < 3) S.words[53 + lane] = 0u;
  S.sf_tab[lane] = (real)C1_TABLES(L.tables)->scale_factors[lane];
  if (lane < 16) S.inv_tab[lane] = (real)C1_TABLES(L.tables)->inv_range[lane];
  if (lane < 32) S.wtab[lane] = F32 ? (real)C1_TABLES(L.tables)->win32[lane] : (real)C1_TABLES(L.tables)->window[lane];
  // lane-only geometry, computed once per wave
  // A lane dequantizes the eight CONSECUTIVE slots 8 lane .. 8 lane + 7 (BFU-major order = bit-stream order = coefficient
  // order of a long band): one running bit cursor, one 64-bit window read per mantissa, two 16-byte stores.  They lie in at
  // most three BFUs b0, b0 + 1, b0 + 2: dq_geo = b0 | index of slot 8 lane inside b0 << 6 | mask of the slots past the first
  // boundary << 11 | mask of the slots past the second << 19 (| SPECS_PER_BFU[lane] << 27, for the bit offsets).
  uint32_t dq_geo;
  {
    const int s0 = 8 * lane0, b0 = bfu_of_slot(s0);
    uint32_t m1 = 0, m2 = 0;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int k = bfu_of_slot(s0 + m) - b0;
      m1 |= (k >= 1 ? 1u : 0u) << m;
      m2 |= (k >= 2 ? 1u : 0u) << m;
    }
    dq_geo = (uint32_t)b0 | ((uint32_t)(s0 - kBfuFirst[b0]) << 6) | (m1 << 11) | (m2 << 19) | ((uint32_t)(lane0 < 52 ? kSpecs[lane0] : 0) << 27);
  }
  if (lane0 < 52) S.dshort[lane0] = (int16_t)((int)kStartShort[lane0] - (int)kBfuFirst[lane0]);
  IMixGeometry IGL = imix_geometry<R>(lane0, FrameModes{0, 0, 0});   // all-long frames
#pragma unroll
  for (int j = 0; j < 4; j++) {
    S.late[j][lane0] = (uint32_t)IGL.ox[j] | ((uint32_t)IGL.oy[j] << 16);
    S.late[4 + j][lane0] = (uint32_t)IGL.post_tab[j];
    IGL.ox[j] = IGL.oy[j] = IGL.post_tab[j] = 0;              // not carried through the loop
  }
  const TablesRsrc RT = tables_rsrc(L.tables);
  wave_fence();

  const int64_t f_end = (f0 + L.run_frames < L.frames) ? f0 + L.run_frames : L.frames;
  int64_t f_first = f0 - 1;                                  // the unit before the run rebuilds the state
  if (f_first < -(int64_t)L.halo_units) f_first = f0;
  // the unit's 53 dwords are requested one unit ahead (a lane's dword) and taken delivery of before the PCM stores of the
  // unit in between are issued: loads and stores share one counter on this part, so a wait for a load behind a store is
  // a wait for the store to reach memory (c1_k_spec.hip)
  auto unit_word = [&](int64_t fr) -> uint32_t {
    return reinterpret_cast<const uint32_t *>(L.units + (fr * L.channels + ch) * C1_UNIT_BYTES)[lane0 < 53 ? lane0 : 0];
  };
  uint32_t next_word = unit_word(f_first);
  for (int64_t f = f_first; f < f_end; ++f) {
    const bool emit = f >= f0;
    TablesPtr T = tables_for_this_frame(L.tables);
    lane = lane_for_this_frame(lane0);

    // ---------------- deserializeFrame (serialization.js:111-176) ----------------
    if (lane < 53) S.words[lane] = __builtin_bswap32(next_word);
    next_word = unit_word(f + 1 < f_end ? f + 1 : f);
    wave_fence();
    const uint32_t header = S.words[0] >> 16;
    const int m0 = 2 - (int)((header >> 14) & 3), m1 = 2 - (int)((header >> 12) & 3), m2 = 3 - (int)((header >> 10) & 3);
    const int n = bfu_amount((header >> 5) & 7);
    int wl = 0, sfi = 0;
    if (lane < n) {
      wl = (int)get_bits_be(S.words, 16 + 4 * lane, 4);
      sfi = (int)get_bits_be(S.words, 16 + 4 * n + 6 * lane, 6);
    }
    const int mybits = wl_bits(wl) * (int)(dq_geo >> 27);            // SPECS_PER_BFU[lane] rides in the geometry word
    const int scan = wave_inclusive_scan(mybits);
    if (lane < 52) {
      S.desc[lane] = (uint32_t)wl_bits(wl) | ((uint32_t)sfi << 5) | ((uint32_t)(16 + 10 * n + scan - mybits) << 11);
      // SF * RN(1 / range) (see dq_step); a BFU with scale factor 0 dequantizes to zeros (quantization.js:66-68)
      S.step[lane] = sfi != 0 ? S.sf_tab[sfi] * S.inv_tab[wl_bits(wl) > 0 ? wl_bits(wl) - 1 : 0] : (real)0;
    }
    wave_fence();
    // ---------------- dequantizationStage (decoder.js:52-98) ----------------
    const bool all_long = (m0 | m1 | m2) == 0;
    {
      const int b0 = (int)(dq_geo & 63u), b1 = b0 + 1 < 52 ? b0 + 1 : 51, b2 = b0 + 2 < 52 ? b0 + 2 : 51;
      const uint32_t in1 = (dq_geo >> 11) & 255u, in2 = (dq_geo >> 19) & 255u;
      const uint32_t d0 = S.desc[b0], d1 = S.desc[b1], d2 = S.desc[b2];
      const real st0 = S.step[b0], st1 = S.step[b1], st2 = S.step[b2];
      const int nb0 = (int)(d0 & 31u), nb1 = (int)(d1 & 31u), nb2 = (int)(d2 & 31u);
      // the slow formulations are only needed where the tables fail the host's check (binary64) or the unit's mantissas run
      // past its 212 bytes (bytes no encoder wrote: unpackBits then returns what is left, bitstream.js:49-70)
      const int last_bits = (int)(S.desc[51] >> 11) + (int)(S.desc[51] & 31u) * 20;
      const bool plain = last_bits <= C1_UNIT_BYTES * 8 && (F32 || T->dq_step != 0);
      int pos = (int)(d0 >> 11) + (int)((dq_geo >> 6) & 31u) * nb0;      // bit position of the lane's first mantissa
      float v[8];
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const bool p1 = (in1 >> m) & 1u, p2 = (in2 >> m) & 1u;
        const int bits = p2 ? nb2 : (p1 ? nb1 : nb0);
        const real st = p2 ? st2 : (p1 ? st1 : st0);
        if (plain) {
          const int w = pos >> 5, o = pos & 31;
          const uint64_t two = ((uint64_t)S.words[w] << 32) | (uint64_t)S.words[w + 1];
          // the mantissa as a signed bit field: bits o .. o + bits of the 64-bit window (bitstream.js:78-82)
          const int64_t field = (int64_t)(two << o) >> ((64 - bits) & 63);
          const int32_t q = bits != 0 ? (int32_t)field : 0;
          v[m] = (float)((real)q * st);                           // == Float32((q * SF) / range) for every input (checked on the host)
        } else {
          const uint32_t dsc = p2 ? d2 : (p1 ? d1 : d0);
          const int sf = (int)((dsc >> 5) & 63u);
          float r = 0.0f;
          if (bits != 0) {
            const uint32_t raw = get_bits_be(S.words, pos, bits);
            const int32_t q = raw >= (1u << (bits - 1)) ? (int32_t)raw - (1 << bits) : (int32_t)raw;
            const int32_t range = (1 << (bits - 1)) - 1;
            if (sf != 0) {                                          // quantization.js:65-78
              if constexpr (F32) r = (float)q * (float)st;
              else {
                const double a = (double)q * S.sf_tab[sf];
                if (T->dq_step) r = f32((double)q * (double)st);
                else if (T->dq_fast) {
                  const double y = S.inv_tab[bits - 1], q0 = a * y;
                  r = f32(__builtin_fma(__builtin_fma(-q0, (double)range, a), y, q0));              // == a / range (checked on the host)
                } else r = f32(a / (double)range);
              }
            }
          }
          v[m] = r;
        }
        pos += bits;
      }
      if (all_long) {
        float4 *dst = reinterpret_cast<float4 *>(S.cb.coef + 8 * lane);
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        // a short band's BFUs are interleaved over its blocks (BFU_START_SHORT, constants.js:46-52)
        const int e0 = (b0 >= 36 ? m2 : (b0 >= 20 ? m1 : m0)) != 0 ? (int)S.dshort[b0] : 0;
        const int e1 = (b1 >= 36 ? m2 : (b1 >= 20 ? m1 : m0)) != 0 ? (int)S.dshort[b1] : 0;
        const int e2 = (b2 >= 36 ? m2 : (b2 >= 20 ? m1 : m0)) != 0 ? (int)S.dshort[b2] : 0;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const bool p1 = (in1 >> m) & 1u, p2 = (in2 >> m) & 1u;
          S.cb.coef[8 * lane + m + (p2 ? e2 : (p1 ? e1 : e0))] = v[m];
        }
      }
    }
    wave_fence();

    // ---------------- imdctStage (decoder.js:116-330) ----------------
    float *mid = S.u.m.zz.mid;
    if (all_long) {
      imdct_r4<R>(S.cb.coef, S.u.m.zz.z, mid, IGL, true, true, T, RT, &S.late[0][0] + lane);
      wave_fence();
      // overlap-add of the first 32 samples of every band (mdct.js:230-245 via decoder.js:203-232) ...
      if (lane < 32) {
        const bool lo = lane < 16;
        const int i = lo ? lane : 31 - lane;
        const real wa = S.wtab[i], wb = S.wtab[31 - i];   // w1 = W[i], w2 = W[31-i]
#pragma unroll
        for (int b = 0; b < 3; b++) {
          const int off = b == 0 ? 0 : (b == 1 ? 128 : 256);
          const real pv = S.tail[16 * b + i], cv = mid[off + 15 - i];
          S.cb.band[off + lane] = lo ? (float)(pv * wb - cv * wa) : (float)(pv * wa + cv * wb);
        }
      }
      // ... the rest of the band is invBuf[16 .. S-16) (decoder.js:215-221)
      if (lane < 48) {
        const int off = lane < 24 ? 0 : 128, q4 = lane < 24 ? lane : lane - 24;
        *reinterpret_cast<float4 *>(&S.cb.band[off + 32 + 4 * q4]) = *reinterpret_cast<const float4 *>(&mid[off + 16 + 4 * q4]);
      }
      if (lane < 56) *reinterpret_cast<float4 *>(&S.cb.band[256 + 32 + 4 * lane]) = *reinterpret_cast<const float4 *>(&mid[256 + 16 + 4 * lane]);
    } else {
    FrameModes M{m0, m1, m2};
    const IMixGeometry IG = imix_geometry<R>(lane, M);
    imdct_r4<R>(S.cb.coef, S.u.m.zz.z, mid, IG, m0 == 0 || m1 == 0 || m2 == 0, m2 == 0, T, RT);
    wave_fence();
    // overlap-add (mdct.js:230-245 via decoder.js:203-232 long / :262-300 short)
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int g = lane + 64 * m;
      const int b = g < 128 ? 0 : (g < 256 ? 1 : 2);
      const int off = b == 0 ? 0 : (b == 1 ? 128 : 256);
      const int l = g - off;
      const bool lng = M.mode_of_band(b) == 0;
      const int q = lng ? 0 : (l >> 5);            // block
      const int k = lng ? l : (l & 31);            // position inside the block's output
      float v;
      if (k < 32) {
        const float *prev = (q == 0) ? (S.tail + 16 * b) : (mid + off + 32 * (q - 1) + 16);
        const float *curr = mid + off + 32 * q;
        if (k < 16) {
          const real w1 = S.wtab[k], w2 = S.wtab[31 - k];
          v = (float)((real)prev[k] * w2 - (real)curr[15 - k] * w1);
        } else {
          const int i = 31 - k;
          const real w1 = S.wtab[i], w2 = S.wtab[31 - i];
          v = (float)((real)prev[i] * w1 + (real)curr[15 - i] * w2);
        }
      } else {
        v = mid[off + k - 16];                     // long block only: invBuf[16 .. S-16)
      }
      S.cb.band[g] = v;
    }
    }
    wave_fence();
    if (lane < 48) {
      const int b = lane >> 4, k = lane & 15;
      const int off = b == 0 ? 0 : (b == 1 ? 128 : 256), Sb = b == 2 ? 256 : 128;
      S.tail[lane] = mid[off + Sb - 16 + k];
    }
    wave_fence();

    // ---------------- qmfSynthesisStage (decoder.js:349-389) ----------------
    real *w2 = S.u.q2.w2, *w1 = S.u.q1.w1;
    // high band delay compensation (:360-366): delayed high sample j = j < 39 ? previous tail : band2[j-39]
    float hi4[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int j = 4 * lane + t;
      hi4[t] = j < 39 ? S.dhi[j] : S.cb.band[256 + j - 39];
    }
    {
      float keep = 0.0f;
      if (lane < 39) keep = S.cb.band[256 + 217 + lane];
      // stage 2: low + mid -> 256 samples (qmf.js:78-84 interleave)
      if (lane < 46) w2[pidx<2>(lane)] = S.d2[lane];
#pragma unroll
      for (int d = 0; d < 2; d++) {
        const int i = 2 * lane + d;
        const real l = S.cb.band[i], h = S.cb.band[128 + i];
        pair v2; v2.x = (real)(float)((real)0.5 * (l + h)); v2.y = (real)(float)((real)0.5 * (l - h));
        *reinterpret_cast<pair *>(&w2[pidx<2>(46 + 2 * i)]) = v2;
      }
      wave_fence();
      if (lane < 39) S.dhi[lane] = keep;
    }
    {
      real s0[2], s1[2];
      qmf_synth_r<R, 2, 2>(w2, lane, T, s0, s1);
      if (lane < 46) S.d2[lane] = w2[pidx<2>(256 + lane)];
      wave_fence();                                    // w1 reuses the memory of w2 from here on
      if (lane < 46) w1[pidx<3>(lane)] = S.d1[lane];
      // stage 1 input: (stage-2 output, delayed high); stage-2 output pair of i: out[2i] = s1, out[2i+1] = s0
#pragma unroll
      for (int d = 0; d < 2; d++) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const int sidx = 4 * lane + 2 * d + t;          // sample index in the 256-sample low band
          const real l = (real)(float)(t == 0 ? s1[d] : s0[d]);
          const real h = hi4[2 * d + t];
          pair v2; v2.x = (real)(float)((real)0.5 * (l + h)); v2.y = (real)(float)((real)0.5 * (l - h));
          *reinterpret_cast<pair *>(&w1[pidx<3>(46 + 2 * sidx)]) = v2;
        }
      }
    }
    wave_fence();
    {
      real s0[4], s1[4];
      qmf_synth_r<R, 4, 3>(w1, lane, T, s0, s1);
      if (lane < 46) S.d1[lane] = w1[pidx<3>(512 + lane)];
      asm volatile("" : "+v"(next_word));                 // the next unit has arrived: nothing waits on a load behind the stores below
      if (emit) {
        float4 *dst = reinterpret_cast<float4 *>(pcm + f * 512 + 8 * lane);
        dst[0] = make_float4((float)s1[0], (float)s0[0], (float)s1[1], (float)s0[1]);
        dst[1] = make_float4((float)s1[2], (float)s0[2], (float)s1[3], (float)s0[3]);
      }
    }
    wave_fence();
  }
}

}  // namespace

void c1k_launch_decode(const C1DecodeLaunch &L0, bool binary32, hipStream_t stream) {
  static const int slots = c1k_wave_slots(k_decode<double>);
  C1DecodeLaunch L = L0;
  L.run_frames = c1k_pick_run(L.frames, L.channels, slots);
  const int64_t runs = (L.frames + L.run_frames - 1) / L.run_frames;
  const dim3 grid((unsigned)(runs * L.channels)), block(C1_WAVE);
  if (binary32) hipLaunchKernelGGL((k_decode<float>), grid, block, 0, stream, L);
  else hipLaunchKernelGGL((k_decode<double>), grid, block, 0, stream, L);
}
