// c1_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ATRAC1 hot path.
//
// Numeric model (the reference is ECMAScript): every operation is an IEEE-754
// double operation, no a*b+c fusion, and every Float32Array store rounds to binary32.  This file
// is compiled with -ffp-contract=off; the ONLY fused operations are the QMF convolution terms,
// where both factors are binary32 values so the double product is exact and fma(a,b,acc) equals
// round(a*b)+acc bit for bit.  No MFMA: nothing here is a dense contraction.
//
// Work decomposition (one wavefront == one 64-thread workgroup, so __syncthreads() is a wave fence)
//   k_analysis  one wave walks kRunFrames consecutive frames of one channel, carrying the QMF delay
//               lines / MDCT overlap / transient history in LDS exactly like the reference's
//               BufferPool (codec/core/buffers.js), after a 1-2 frame warm-up from the PCM halo.
//               -> MDCT coefficients, scale-factor indices, block modes
//   k_allocate  one LANE per (frame, candidate BFU count): the 8 greedy heaps of
//               allocateBits (codec/coding/bitallocation.js:74-142) run side by side, 8 frames per wave
//   k_pack      one wave per sound unit: quantize (quantization.js:34-56) + MSB-first bit packing
//               (serialization.js:41-98) into the 212-byte unit
//   k_decode    one wave walks kRunFrames consecutive units of one channel:
//               unpack, dequantize, IMDCT + overlap-add, QMF synthesis
#include "c1_internal.h"

#include <algorithm>
#include <type_traits>

#pragma clang fp contract(off)

#define C1_WAVE 64

// The tables are written once per context before any kernel runs and never by a kernel: view them
// through the constant address space so wave-uniform reads become scalar loads (SGPR operands).
typedef const __attribute__((address_space(4))) C1DevTables *TablesPtr;
#define C1_TABLES(p) ((TablesPtr)(p))
// Re-derive the table pointer through an opaque asm once per frame: table reads then cannot be
// hoisted out of the frame loop (hundreds of loop-invariant twiddles would spill the register file).
// Same trick for the lane id: every LDS index of the (fully unrolled) frame body is a function of it,
// and hoisting those out of the frame loop costs more registers than recomputing them.
__device__ __forceinline__ int lane_for_this_frame(int lane) {
  asm volatile("" : "+v"(lane));
  return lane;
}
// Always true, but not to the compiler: `if (own_block()) core(); else cheap();` keeps a long unrolled core in a
// basic block of its own.  Merged into the surrounding block, the scheduler hoists the core's LDS reads over
// the code before it and the kernel spills (k_analysis_fast: 110 VGPRs and no scratch with, 128 + 49 spills without).
__device__ __forceinline__ bool own_block() {
  int one = 1;
  asm volatile("" : "+s"(one));
  return one != 0;
}
__device__ __forceinline__ TablesPtr tables_for_this_frame(const C1DevTables *p) {
  unsigned long long v = (unsigned long long)p;
  asm volatile("" : "+s"(v));
  return (TablesPtr)v;
}

namespace {

// ---- format tables: codec/core/constants.js:29-52, :141-143 -----------------------------------
__constant__ const uint8_t kSpecs[52] = {8, 8, 8, 8, 4,  4,  4,  4,  8,  8,  8,  8,  6,  6,  6,  6,  6,  6,
                                   6, 6, 6, 6, 6,  6,  7,  7,  7,  7,  9,  9,  9,  9,  10, 10, 10, 10,
                                   12, 12, 12, 12, 12, 12, 12, 12, 20, 20, 20, 20, 20, 20, 20, 20};
__constant__ const uint16_t kStartLong[52] = {0,   8,   16,  24,  32,  36,  40,  44,  48,  56,  64,  72,  80,
                                        86,  92,  98,  104, 110, 116, 122, 128, 134, 140, 146, 152, 159,
                                        166, 173, 180, 189, 198, 207, 216, 226, 236, 246, 256, 268, 280,
                                        292, 304, 316, 328, 340, 352, 372, 392, 412, 432, 452, 472, 492};
__constant__ const uint16_t kStartShort[52] = {0,   32,  64,  96,  8,   40,  72,  104, 12,  44,  76,  108, 20,
                                         52,  84,  116, 26,  58,  90,  122, 128, 160, 192, 224, 134, 166,
                                         198, 230, 141, 173, 205, 237, 150, 182, 214, 246, 256, 288, 320,
                                         352, 384, 416, 448, 480, 268, 300, 332, 364, 396, 428, 460, 492};
// first coefficient slot (BFU-major order) of each BFU = prefix sum of kSpecs
__constant__ const uint16_t kBfuFirst[53] = {0,   8,   16,  24,  32,  36,  40,  44,  48,  56,  64,  72,  80,  86,
                                       92,  98,  104, 110, 116, 122, 128, 134, 140, 146, 152, 159, 166, 173,
                                       180, 189, 198, 207, 216, 226, 236, 246, 256, 268, 280, 292, 304, 316,
                                       328, 340, 352, 372, 392, 412, 432, 452, 472, 492, 512};
// BFU_AMOUNTS {20, 28, 32, 36, 40, 44, 48, 52} (constants.js) as arithmetic: a lane-varying lookup would be a global load, and waiting for it (vmcnt is in order)
// also waits for every prefetch issued before it
__device__ __forceinline__ int bfu_amount(int index) { return index == 0 ? 20 : 24 + 4 * index; }

__device__ __forceinline__ int wl_bits(int wl) { return wl == 0 ? 0 : wl + 1; }  // WORD_LENGTH_BITS
__device__ __forceinline__ int band_of_bfu(int b) { return b >= 36 ? 2 : (b >= 20 ? 1 : 0); }
// BFU that owns coefficient slot p (BFU-major order); sizes are piecewise constant
__device__ __forceinline__ int bfu_of_slot(int p) {
  if (p < 32) return p >> 3;
  if (p < 48) return 4 + ((p - 32) >> 2);
  if (p < 80) return 8 + ((p - 48) >> 3);
  if (p < 152) return 12 + (p - 80) / 6;
  if (p < 180) return 24 + (p - 152) / 7;
  if (p < 216) return 28 + (p - 180) / 9;
  if (p < 256) return 32 + (p - 216) / 10;
  if (p < 352) return 36 + (p - 256) / 12;
  return 44 + (p - 352) / 20;
}

__device__ __forceinline__ float f32(double x) { return (float)x; }  // Float32Array store

// index of double element e in a QMF work buffer: 2 pad doubles after every 2^S, so that the 16-byte
// window reads of a wave whose lanes are 64 bytes (4 outputs per lane, S = 3) or 32 bytes (2 outputs
// per lane, S = 2) apart are bank-conflict free (tools/lds_model.py); the generic kernels use S = 5
template <int S = 5>
__device__ __forceinline__ int pidx(int e) { return e + ((e >> S) << 1); }

// ---- QMF convolution core ------------------------------------------------------------------------
// Analysis (qmf.js:33-47): output i needs work[2i .. 2i+47]:
//   even = sum_j work[2i+47-2j]*EVEN[j],  odd = sum_j work[2i+46-2j]*ODD[j],  j ascending.
// A lane owns D consecutive outputs i = D*lane+d, i.e. the 46+2D doubles from 2*D*lane, read
// as 16-byte (even,odd) pairs u = 22+D .. 0; pair u feeds tap j = d+23-u of output d, so walking
// u downwards adds the terms of every sum in the reference's order.
// When 2D == 2^S (4 outputs per lane with S = 3, 2 with S = 2) the padded index of a lane's window is
// affine in the lane: pidx<S>(2D*lane + 2u) = (2D+2)*lane + 2u + 2*((2u) >> S), so every read is
// "lane base + compile-time offset" and costs no address arithmetic.
template <int D, int S>
__device__ __forceinline__ const double2 *qmf_window(const double *w, int lane, int u) {
  if constexpr (2 * D == (1 << S)) return reinterpret_cast<const double2 *>(w + (2 * D + 2) * lane + (2 * u + 2 * ((2 * u) >> S)));
  else return reinterpret_cast<const double2 *>(&w[pidx<S>(2 * D * lane + 2 * u)]);
}
template <int D, int S = 5>
__device__ __forceinline__ void qmf_analysis_core(const double *w, int lane, TablesPtr T,
                                                  double (&even)[D], double (&odd)[D]) {
#pragma unroll
  for (int d = 0; d < D; d++) even[d] = odd[d] = 0.0;
#pragma unroll
  for (int u = 22 + D; u >= 0; --u) {
    const double2 x = *qmf_window<D, S>(w, lane, u);
#pragma unroll
    for (int d = 0; d < D; d++) {
      const int j = d + 23 - u;
      if (j >= 0 && j < 24) {
        odd[d] = __builtin_fma(x.x, T->tap_e[23 - j], odd[d]);    // exact product: both factors are binary32
        even[d] = __builtin_fma(x.y, T->tap_e[j], even[d]);
      }
    }
  }
}
// Synthesis (qmf.js:89-102): out[2i+1] = sum_j work[2i+2j]*EVEN[j], out[2i] = sum_j work[2i+2j+1]*ODD[j].
// Same window; pair u feeds tap j = u-d, walking u upwards.
template <int D, int S = 5>
__device__ __forceinline__ void qmf_synthesis_core(const double *w, int lane, TablesPtr T,
                                                   double (&s0)[D], double (&s1)[D]) {
#pragma unroll
  for (int d = 0; d < D; d++) s0[d] = s1[d] = 0.0;
#pragma unroll
  for (int u = 0; u <= 22 + D; ++u) {
    const double2 x = *qmf_window<D, S>(w, lane, u);
#pragma unroll
    for (int d = 0; d < D; d++) {
      const int j = u - d;
      if (j >= 0 && j < 24) {
        s0[d] = __builtin_fma(x.x, T->tap_e[j], s0[d]);
        s1[d] = __builtin_fma(x.y, T->tap_e[23 - j], s1[d]);
      }
    }
  }
}

__device__ __forceinline__ int bitrev(int k, int log2n) { return (int)(__brev((unsigned)k) >> (32 - log2n)); }

// findScaleFactor on binary32 bit patterns: with SF[3q] = 2^(q-21) and the two in-between fraction
// patterns m1 < m2 shared by every octave, the index is 3(e+21) + [frac > 0] + [frac > m1] + [frac > m2]
__device__ __forceinline__ int scale_factor_index_fast(float maxabs, uint32_t m1, uint32_t m2) {
  const uint32_t u = __float_as_uint(maxabs);
  const int e = (int)(u >> 23) - 127;
  const uint32_t frac = u & 0x7fffffu;
  int r = 3 * (e + 21) + (frac > 0u ? 1 : 0) + (frac > m1 ? 1 : 0) + (frac > m2 ? 1 : 0);
  r = r > 63 ? 63 : r;
  return (e < -21) ? 0 : r;     // also zero, denormals and anything below 2^-21
}

// smallest i with m <= SCALE_FACTORS[i], clamped to [0,63]  == findScaleFactor, bitallocation.js:290-299
__device__ __forceinline__ int scale_factor_index(float maxabs, TablesPtr T) {
  if (!(maxabs > 0.0f)) return 0;
  const double m = (double)maxabs;
  if (m > 1.0) return 63;                        // SCALE_FACTORS[63] = 2^0
  int e = (int)((__float_as_uint(maxabs) >> 23) & 0xff) - 127;  // floor(log2 m) for normal m
  if (e < -21) return 0;                         // also covers denormals (field 0 -> e = -127)
  int i = 3 * (e + 21);                          // SCALE_FACTORS[i] = 2^e <= m
  // m in [2^e, 2^(e+1)): answer is i, i+1, i+2 or i+3
  int r = i;
  if (m > T->scale_factors[i]) r = i + 1;
  if (i + 1 <= 63 && m > T->scale_factors[i + 1 > 63 ? 63 : i + 1]) r = i + 2;
  if (i + 2 <= 63 && m > T->scale_factors[i + 2 > 63 ? 63 : i + 2]) r = i + 3;
  return r > 63 ? 63 : r;
}

// =====================================================================================================
// k_analysis
// =====================================================================================================
// mode-dependent geometry of the 256 complex FFT points of one frame: band 0 -> [0,64), band 1 ->
// [64,128), band 2 -> [128,256); a long band is one transform, a short band is 16-point blocks.
struct FrameModes {
  int m0, m1, m2;
  __device__ __forceinline__ int mode_of_band(int b) const { return b == 0 ? m0 : (b == 1 ? m1 : m2); }
  __device__ __forceinline__ int fft_size_at(int p) const {
    const int b = p < 64 ? 0 : (p < 128 ? 1 : 2);
    return mode_of_band(b) != 0 ? 16 : (b == 2 ? 128 : 64);
  }
};

// scale factors of an all-long frame (bitallocation.js:80-90): lanes 0..43 take BFUs 0..43 (<= 12 coefficients),
// lanes 44..59 take one half (10 coefficients) of BFUs 44..51 each; 12 clamped reads per lane, then the halves are
// combined.  The lane's slice is fixed, so it is looked up once per wave (a lookup inside the frame loop is a
// global load whose wait also waits for the frame's stores).
struct SfLong { int cnt, src, b; bool wide, store; };
__device__ __forceinline__ SfLong sf_long_geometry(int lane) {
  SfLong g;
  g.wide = lane >= 44;
  g.b = g.wide ? 44 + ((lane - 44) >> 1) : lane;
  const int half = g.wide ? (lane & 1) : 0;
  g.cnt = lane < 60 ? (g.wide ? 10 : (int)kSpecs[lane < 44 ? lane : 0]) : 1;
  g.src = kStartLong[lane < 60 ? g.b : 0] + 10 * half;
  g.store = lane < 60 && (!g.wide || half == 0);
  return g;
}
__device__ __forceinline__ void sf_long(const float *coef, uint8_t *sfi_out, const SfLong &g, TablesPtr T) {
  const float *src = coef + g.src;
  float mx = 0.0f;
#pragma unroll
  for (int j = 0; j < 12; j++) mx = fmaxf(mx, fabsf(src[j < g.cnt ? j : g.cnt - 1]));
  mx = fmaxf(mx, g.wide ? __shfl_xor(mx, 1) : 0.0f);
  const int sfi = T->sf_fast ? scale_factor_index_fast(mx, T->sf_m1, T->sf_m2) : scale_factor_index(mx, T);
  if (g.store) sfi_out[g.b] = (uint8_t)sfi;
}

// =====================================================================================================
// k_analysis_fast : QMF -> block selection -> MDCT -> scale factors, one wave per run of frames
// =====================================================================================================
// Fixed block modes only (transient detection has its own pipeline further down).  template <ALL_LONG>:
//   <true>    modes [0,0,0]: the long-block MDCT core and nothing else
//   <false>   any other fixed modes: staging + the mixed long/short core
// Every vector instruction costs the same 4 cycles here, so the long-block core is shaped to minimise
// their count: MDCT inputs in LDS buffers, FFT on interleaved (re,im) pairs with lane-only geometry
// computed once per wave, bank-conflict-free layouts (tools/lds_model.py), no per-element mode logic.
struct alignas(16) LongLds {
  double d1[46];                 // stage-1 QMF delay line
  double d2[46];                 // stage-2 QMF delay line
  alignas(16) float hbuf[296];   // delayed high band: [0,39) tail of the previous frame, [39,295) this frame
  alignas(4) uint8_t sfi[64];
  // scratch with disjoint lifetimes inside one frame (10 KiB per wave in total: 16 waves per CU)
  union alignas(16) {
    struct { alignas(16) double w1[698]; } q1;     // stage-1 QMF work buffer, padded 2 per 8
    struct { alignas(16) double w2[454]; } q2;     // stage-2 QMF work buffer (after stage 1 has read w1), padded 2 per 4
    struct {
      union alignas(16) {
        struct { alignas(16) float in0[256]; alignas(16) float in1[256]; alignas(16) float in2[512]; } i;   // MDCT inputs
        struct { alignas(16) float coef[512]; } c;                                                           // coefficients (after the pre-twiddle)
      } a;
      union alignas(16) { float band[512]; } zz;   // before the pre-twiddle: low128 | mid128 | high256, raw
    } m;
    // FFT points (4 pad slots per 16).  They start 512 bytes before the end of the MDCT inputs: round A has read
    // every input before it writes its first point (one wave, LDS operations in issue order), and the overlap
    // keeps a wave at 8 128 bytes, i.e. 20 waves per CU
    struct { alignas(16) float skip_[896]; float2 z[320]; } zp;
  } u;
};

struct alignas(16) MixedLds {
  double d1[46];
  double d2[46];
  alignas(16) float band[512];
  alignas(16) float hbuf[296];
  alignas(4) uint8_t sfi[64];
  alignas(16) float ovl[96];     // mdctOverlap, 3 x 32 (the all-long kernel keeps it in registers)
  union alignas(16) {
    struct { alignas(16) double w1[698]; } q1;
    struct { alignas(16) double w2[454]; } q2;
    struct {
      union alignas(16) {
        struct { alignas(16) float in[1120]; } g;      // staging (kStageFloats)
        struct { alignas(16) float coef[512]; } c;
      } a;
      union alignas(16) { float2 z[320]; } zz;
    } m;
  } u;
};

// ---- long-block MDCT core, radix-4 rounds -------------------------------------------------------------
// The three long transforms of a frame (64, 64 and 128 complex points) run side by side: lanes 0..15 own
// band 0, 16..31 band 1, 32..63 band 2, four points per lane.  The reference's radix-2 stages (fft.js:41-66)
// are executed two at a time in registers -- same operations, same Float32 rounding after every stage, same
// tabulated twiddles -- so a frame makes 3 (4 for the 128-point transform) trips through LDS instead of 9:
//   round A  pre-twiddle (mdct.js:76-105) of the points at bit-reversed positions 4g..4g+3, stages h = 1, 2
//   round B  stages h = 4, 8       (positions p + 4j inside one 16-block)
//   round C  stages h = 16, 32     (positions p + 16j)
//   round D  stage  h = 64         (band 2 only: positions m, m + 64)
//   post-twiddle (mdct.js:110-119) + spectrum reversal straight from the registers of the last round.
// A point at position p of band b lives in slot zslot(base_b + p): 4 pad slots after every 16 keep every
// exchange "lane base + immediate offset" and free of bank conflicts (tools/lds_model.py).
__device__ __forceinline__ int zslot(int pos) { return pos + 4 * (pos >> 4); }

struct R4Geometry {
  int ia[4], ic[4], ib0, id0, ib3, id3;   // float indices into in0|in1|in2 of the pre-twiddle operands
  int pre_tab[4];                          // byte offset (from the tables) of (cos,sin) of point k_j
  int za, zb, zc, zd;                      // first slot of the lane's points in rounds A..D
  int twb, twc, twd;                       // byte offset of the lane's first twiddle in rounds B..D
  int post_tab[4];
  int cx[4], cy[4];                        // coefficient index of the two outputs of each final point
  bool band2;
};

__device__ __forceinline__ R4Geometry r4_geometry(int lane) {
  R4Geometry G;
  const int band = lane < 16 ? 0 : (lane < 32 ? 1 : 2);
  const int g = lane - (band == 0 ? 0 : (band == 1 ? 16 : 32));
  const int n4 = band == 2 ? 128 : 64, q = n4 / 4;
  const int r = bitrev(g, band == 2 ? 5 : 4);
  const int in_base = band == 0 ? 0 : (band == 1 ? 256 : 512);
  const int tab_base = band == 2 ? (int)offsetof(C1DevTables, mdct_fwd512) : (int)offsetof(C1DevTables, mdct_fwd256);
  const int tw_base = (int)offsetof(C1DevTables, fft_tw);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int jp = ((j & 1) << 1) | (j >> 1);       // position 4g+j holds point k = r + q * bitrev2(j)
    const int k = r + q * jp, i = 2 * k;
    G.ia[j] = in_base + 3 * n4 - 1 - i;
    G.ic[j] = in_base + n4 + i;
    G.pre_tab[j] = tab_base + 16 * k;
  }
  G.ib0 = in_base + 3 * n4 + 2 * r;                 // j = 0: first half (i < N/4)
  G.id0 = in_base + n4 - 1 - 2 * r;
  const int i3 = 2 * (r + 3 * q);                   // j = 3: second half
  G.ib3 = in_base + i3 - n4;
  G.id3 = in_base + 5 * n4 - 1 - i3;
  const int pbase = band == 0 ? 0 : (band == 1 ? 64 : 128);
  G.za = zslot(pbase + 4 * g);
  G.zb = zslot(pbase + 16 * (g >> 2) + (g & 3));
  G.twb = tw_base + 16 * (3 + (g & 3));
  G.zc = zslot(pbase + 64 * (g >> 4) + (g & 15));
  G.twc = tw_base + 16 * (15 + (g & 15));
  G.band2 = band == 2;
  G.zd = zslot(128 + (g & 31));
  G.twd = tw_base + 16 * (63 + (g & 31));
  const int cbase = band == 0 ? 0 : (band == 1 ? 128 : 256), n2 = 2 * n4;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    // final points: bands 0/1 hold g + 16j after round C; band 2 holds g, g+64, g+32, g+96 after round D
    const int i = band == 2 ? g + (j == 1 ? 64 : (j == 2 ? 32 : (j == 3 ? 96 : 0))) : g + 16 * j;
    G.post_tab[j] = tab_base + 16 * i;
    const int e0 = cbase + 2 * i, e1 = cbase + n2 - 1 - 2 * i;
    G.cx[j] = band == 0 ? e0 : e1;                  // bands 1 and 2 are stored reversed (utils.js:42-48)
    G.cy[j] = band == 0 ? e1 : e0;
  }
  return G;
}

// lane-varying table reads go through a buffer resource: 32-bit byte offsets (one VGPR per address; the
// 64-bit form costs two plus an add) and hardware bounds checking against the table size
typedef __amdgpu_buffer_rsrc_t TablesRsrc;
__device__ __forceinline__ TablesRsrc tables_rsrc(const C1DevTables *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<C1DevTables *>(p), 0, (int)sizeof(C1DevTables), 0x00020000);
}
__device__ __forceinline__ double2 table_pair(TablesRsrc R, int byte_offset) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(R, byte_offset, 0, 0);
  double2 d;
  __builtin_memcpy(&d, &v, sizeof d);
  return d;
}
// one radix-2 butterfly of fft.js:46-60 on Float32 points held in registers
__device__ __forceinline__ void r2_butterfly(float2 &e, float2 &o, const double2 w) {
  const double er = e.x, ei = e.y, orr = o.x, oi = o.y;
  const double xr = orr * w.x - oi * w.y;
  const double xi = orr * w.y + oi * w.x;
  e = make_float2(f32(er + xr), f32(ei + xi));
  o = make_float2(f32(er - xr), f32(ei - xi));
}

// The same butterfly when the twiddle is exactly (1, 0) (k = 0 of every stage, fft.js:44-45).  Then
// xr = or*1 - oi*0 == or and xi = or*0 + oi*1 == oi whenever or, oi are finite and not -0 (only then can the
// signed-zero products change the sum), and Float32(er + or) computed in binary64 equals the binary32 sum
// (53 >= 2*24 + 2: the double rounding is innocuous).  r2_unit_ok is that precondition; callers take the
// general butterfly when any lane fails it, so the result is the reference's in every case.
__device__ __forceinline__ bool r2_unit_ok(const float2 e, const float2 o) {
  constexpr int kFinite = 0x1F8, kFiniteNotNegZero = 0x1D8;   // v_cmp_class masks
  return __builtin_amdgcn_classf(e.x, kFinite) && __builtin_amdgcn_classf(e.y, kFinite) &&
         __builtin_amdgcn_classf(o.x, kFiniteNotNegZero) && __builtin_amdgcn_classf(o.y, kFiniteNotNegZero);
}
__device__ __forceinline__ void r2_butterfly_unit(float2 &e, float2 &o) {
  const float2 a = e, b = o;
  e = make_float2(a.x + b.x, a.y + b.y);
  o = make_float2(a.x - b.x, a.y - b.y);
}

// pre-twiddle pairs of round A, requested by the caller ahead of the core (before the second QMF stage)
struct R4Early { double2 t0, t1, t2, t3; };
__device__ __forceinline__ R4Early r4_early(const R4Geometry &G, TablesRsrc R) {
  R4Early e;
  e.t0 = table_pair(R, G.pre_tab[0]); e.t1 = table_pair(R, G.pre_tab[1]);
  e.t2 = table_pair(R, G.pre_tab[2]); e.t3 = table_pair(R, G.pre_tab[3]);
  return e;
}
// in: 1024 floats (in0 | in1 | in2, zero padded long-block inputs); z: 320 slots; coef: 512 floats (may share
// memory with `in`: the inputs are dead once round A has read them)
__device__ __forceinline__ void mdct_long_r4(const float *in, float2 *z, float *coef, const R4Geometry &G, TablesPtr T, TablesRsrc R, const R4Early &E) {
  float2 x[4];
  // the lane-varying table values of the frame are requested up front: the loads are in flight while round A
  // reads its inputs, instead of one cache round trip in front of every round
  const double2 t0 = E.t0, t1 = E.t1, t2 = E.t2, t3 = E.t3;
  const double2 wBa = table_pair(R, G.twb), wBb = table_pair(R, G.twb + 64), wBc = table_pair(R, G.twb + 128);
  // ---- round A: pre-twiddle + stages 1, 2 ----
  {
    const float a0 = in[G.ia[0]], c0 = in[G.ic[0]], b0 = in[G.ib0], d0 = in[G.id0];
    const float a1 = in[G.ia[1]], c1 = in[G.ic[1]];
    const float a2 = in[G.ia[2]], c2 = in[G.ic[2]];
    const float a3 = in[G.ia[3]], c3 = in[G.ic[3]], b3 = in[G.ib3], d3 = in[G.id3];
    // the long-block input is zero outside [N/4 - 16 .. 3N/4 + 16): for the points of positions 4g+1 and 4g+2
    // the operands b and d are those zeros for every lane, and x - (+0) == x, so only "+ 0.0" remains
    const double r0 = (double)a0 + (double)b0, m0 = (double)c0 - (double)d0;      // first half:  r = a + b, m = c - d
    const double r1 = (double)a1, m1 = (double)c1 + 0.0;                          // second half: r = a - b, m = c + d
    const double r2 = (double)a2 + 0.0, m2 = (double)c2;
    const double r3 = (double)a3 - (double)b3, m3 = (double)c3 + (double)d3;
    x[0] = make_float2(f32(r0 * t0.x + m0 * t0.y), f32(m0 * t0.x - r0 * t0.y));
    x[1] = make_float2(f32(r1 * t1.x + m1 * t1.y), f32(m1 * t1.x - r1 * t1.y));
    x[2] = make_float2(f32(r2 * t2.x + m2 * t2.y), f32(m2 * t2.x - r2 * t2.y));
    x[3] = make_float2(f32(r3 * t3.x + m3 * t3.y), f32(m3 * t3.x - r3 * t3.y));
    const double2 w0 = make_double2(T->fft_tw[0][0], T->fft_tw[0][1]);
    const double2 w1 = make_double2(T->fft_tw[1][0], T->fft_tw[1][1]);
    const double2 w2 = make_double2(T->fft_tw[2][0], T->fft_tw[2][1]);
    // stages 1 and 2: three of the four butterflies have the twiddle (1, 0) -> Float32 adds when that is exact
    if (__all(r2_unit_ok(x[0], x[1]) && r2_unit_ok(x[2], x[3]))) { r2_butterfly_unit(x[0], x[1]); r2_butterfly_unit(x[2], x[3]); }
    else { r2_butterfly(x[0], x[1], w0); r2_butterfly(x[2], x[3], w0); }
    if (__all(r2_unit_ok(x[0], x[2]))) r2_butterfly_unit(x[0], x[2]);
    else r2_butterfly(x[0], x[2], w1);
    r2_butterfly(x[1], x[3], w2);
    float4 *dst = reinterpret_cast<float4 *>(z + G.za);
    dst[0] = make_float4(x[0].x, x[0].y, x[1].x, x[1].y);
    dst[1] = make_float4(x[2].x, x[2].y, x[3].x, x[3].y);
  }
  // (the twiddles of rounds C and D take the registers the pre-twiddle pairs just left)
  const double2 wCa = table_pair(R, G.twc), wCb = table_pair(R, G.twc + 256), wCc = table_pair(R, G.twc + 512);
  const double2 wDa = table_pair(R, G.twd), wDb = table_pair(R, G.twd + 512);
  __syncthreads();
  // ---- round B: stages 4, 8 ----
  {
    float2 *p = z + G.zb;
    const double2 wa = wBa, wb = wBb, wc = wBc;
    x[0] = p[0]; x[1] = p[4]; x[2] = p[8]; x[3] = p[12];
    r2_butterfly(x[0], x[1], wa); r2_butterfly(x[2], x[3], wa);
    r2_butterfly(x[0], x[2], wb); r2_butterfly(x[1], x[3], wc);
    p[0] = x[0]; p[4] = x[1]; p[8] = x[2]; p[12] = x[3];
  }
  const double2 p0 = table_pair(R, G.post_tab[0]), p1 = table_pair(R, G.post_tab[1]);
  const double2 p2 = table_pair(R, G.post_tab[2]), p3 = table_pair(R, G.post_tab[3]);
  __syncthreads();
  // ---- round C: stages 16, 32 ----
  {
    float2 *p = z + G.zc;
    const double2 wa = wCa, wb = wCb, wc = wCc;
    x[0] = p[0]; x[1] = p[20]; x[2] = p[40]; x[3] = p[60];
    r2_butterfly(x[0], x[1], wa); r2_butterfly(x[2], x[3], wa);
    r2_butterfly(x[0], x[2], wb); r2_butterfly(x[1], x[3], wc);
    if (G.band2) { p[0] = x[0]; p[20] = x[1]; p[40] = x[2]; p[60] = x[3]; }
  }
  __syncthreads();
  // ---- round D: stage 64 of the 128-point transform ----
  if (G.band2) {
    const float2 *p = z + G.zd;
    const double2 wa = wDa, wb = wDb;
    x[0] = p[0]; x[1] = p[80]; x[2] = p[40]; x[3] = p[120];
    r2_butterfly(x[0], x[1], wa); r2_butterfly(x[2], x[3], wb);
  }
  // ---- post-twiddle + spectrum reversal ----
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const double2 t = j == 0 ? p0 : (j == 1 ? p1 : (j == 2 ? p2 : p3));
    const double rr = x[j].x, ii = x[j].y;
    coef[G.cx[j]] = f32(-rr * t.x - ii * t.y);
    coef[G.cy[j]] = f32(-rr * t.y + ii * t.x);
  }
}


// ---- MDCT core for frames with short blocks, radix-4 rounds ------------------------------------------------------
// Same lane ownership as mdct_long_r4 (lanes 0..15 band 0, 16..31 band 1, 32..63 band 2, four points per lane).
// A short band is 4 (8 for band 2) blocks of 32 samples, each a 64-sample MDCT = a 16-point transform = exactly
// rounds A and B; a long band of the same frame goes on through rounds C (and D).  Inputs come from a staging
// buffer of three regions R_b (floats 0, 288, 576):
//   long band   zero | overlap | samples, last 32 windowed | zero            (encoder.js:228-258)
//   short band  E = overlap(32) | W[s & 31] * x[s]      then   H = x[s] * W[31 - (s & 31)]
//               so that block q reads its first half at E[32q + i] and its second at H[32q + i]  (encoder.js:269-307)
constexpr int kStageFloats = 1120;
__device__ __forceinline__ int stage_region(int band) { return band == 0 ? 0 : (band == 1 ? 288 : 576); }

struct MixGeometry {
  int ia[4], ib[4], ic[4], id[4];
  int pre_tab[4];
  int za, zb, zc, zd, twb, twc, twd;
  int post_tab[4], cx[4], cy[4];
  bool is_long, band2;
};

__device__ __forceinline__ MixGeometry mix_geometry(int lane, const FrameModes &M) {
  MixGeometry G;
  const int band = lane < 16 ? 0 : (lane < 32 ? 1 : 2);
  const int g = lane - (band == 0 ? 0 : (band == 1 ? 16 : 32));
  const bool lng = M.mode_of_band(band) == 0;
  const int R = stage_region(band), Sb = band == 2 ? 256 : 128;
  const int n4 = lng ? (band == 2 ? 128 : 64) : 16, q4 = n4 / 4;
  const int r = lng ? bitrev(g, band == 2 ? 5 : 4) : bitrev(g & 3, 2);
  const int blk = g >> 2;                                  // short: block of the band
  const int tab_base = lng ? (band == 2 ? (int)offsetof(C1DevTables, mdct_fwd512) : (int)offsetof(C1DevTables, mdct_fwd256))
                           : (int)offsetof(C1DevTables, mdct_fwd64);
  const int tw_base = (int)offsetof(C1DevTables, fft_tw);
  // operand idx of a 4*n4-sample input -> float index in the staging buffer
  auto at = [&](int idx) { return lng ? R + idx : (idx < 32 ? R + 32 * blk + idx : R + 32 + Sb + 32 * blk + (idx - 32)); };
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int jp = ((j & 1) << 1) | (j >> 1);
    const int k = r + q4 * jp, i = 2 * k;
    const bool hi = (j & 1) != 0;                          // k >= n4/2 for positions 4g+1 and 4g+3
    G.ia[j] = at(3 * n4 - 1 - i);
    G.ic[j] = at(n4 + i);
    G.ib[j] = at(hi ? i - n4 : 3 * n4 + i);
    G.id[j] = at(hi ? 5 * n4 - 1 - i : n4 - 1 - i);
    G.pre_tab[j] = tab_base + 16 * k;
  }
  const int pbase = band == 0 ? 0 : (band == 1 ? 64 : 128);
  G.za = zslot(pbase + 4 * g);
  G.zb = zslot(pbase + 16 * (g >> 2) + (g & 3));
  G.twb = tw_base + 16 * (3 + (g & 3));
  G.zc = zslot(pbase + 64 * (g >> 4) + (g & 15));
  G.twc = tw_base + 16 * (15 + (g & 15));
  G.zd = zslot(128 + (g & 31));
  G.twd = tw_base + 16 * (63 + (g & 31));
  G.is_long = lng;
  G.band2 = band == 2;
  const int cbase = band == 0 ? 0 : (band == 1 ? 128 : 256);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    int i, e0, e1;
    if (lng) {
      i = band == 2 ? g + (j == 1 ? 64 : (j == 2 ? 32 : (j == 3 ? 96 : 0))) : g + 16 * j;
      e0 = cbase + 2 * i; e1 = cbase + 2 * n4 - 1 - 2 * i;
    } else {
      i = (g & 3) + 4 * j;                                 // points of block blk after round B
      e0 = cbase + 32 * blk + 2 * i; e1 = cbase + 32 * blk + 31 - 2 * i;
    }
    G.post_tab[j] = tab_base + 16 * i;
    G.cx[j] = band == 0 ? e0 : e1;
    G.cy[j] = band == 0 ? e1 : e0;
  }
  return G;
}

// staging buffer of one frame from its raw bands and the previous frame's windowed tails (ovl: 3 x 32)
__device__ __forceinline__ void mix_stage(const float *band_, const float *ovl_, float *stage, const FrameModes &M, int lane,
                                          TablesRsrc RT) {
  // window values of the lane's four samples (same residue mod 32 in both passes): W[4(l&7)+d] and W[31-4(l&7)-d]
  const int wofs = (int)offsetof(C1DevTables, window) + 8 * 4 * (lane & 7);
  const double2 wl01 = table_pair(RT, wofs), wl23 = table_pair(RT, wofs + 16);
  const int hofs = (int)offsetof(C1DevTables, window) + 8 * (28 - 4 * (lane & 7));
  const double2 wh32 = table_pair(RT, hofs), wh10 = table_pair(RT, hofs + 16);    // W[28-4m .. 31-4m]
  const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int b = pass == 0 ? (lane >> 5) : 2;
    const int s = pass == 0 ? 4 * (lane & 31) : 4 * lane;
    const int Sb = b == 2 ? 256 : 128, R = stage_region(b), ws = b == 2 ? 112 : 48;
    const bool lng = M.mode_of_band(b) == 0;
    const float4 v = *reinterpret_cast<const float4 *>(band_ + (b == 0 ? 0 : (b == 1 ? 128 : 256)) + s);
    float4 lo, hi;
    lo.x = f32(wl01.x * (double)v.x); lo.y = f32(wl01.y * (double)v.y); lo.z = f32(wl23.x * (double)v.z); lo.w = f32(wl23.y * (double)v.w);
    hi.x = f32((double)v.x * wh10.y); hi.y = f32((double)v.y * wh10.x); hi.z = f32((double)v.z * wh32.y); hi.w = f32((double)v.w * wh32.x);
    if (lng) {
      *reinterpret_cast<float4 *>(stage + R + ws + 32 + s) = (s >= Sb - 32) ? hi : v;
    } else {
      *reinterpret_cast<float4 *>(stage + R + 32 + s) = lo;
      *reinterpret_cast<float4 *>(stage + R + 32 + Sb + s) = hi;
    }
    // overlap of the previous frame, and the zero regions of a long band
    const int l8 = pass == 0 ? (lane & 31) : lane;
    if (l8 < 8) *reinterpret_cast<float4 *>(stage + R + (lng ? ws : 0) + 4 * l8) = *reinterpret_cast<const float4 *>(ovl_ + 32 * b + 4 * l8);
    if (lng) {
      const int nz = ws / 4;                                 // float4 per zero region: [0, ws) and [ws + 32 + Sb, 2 ws + 32 + Sb)
      const int l = l8 - 8;
      if (l >= 0 && l < 2 * nz) *reinterpret_cast<float4 *>(stage + R + (l < nz ? 4 * l : ws + 32 + Sb + 4 * (l - nz))) = zero4;
    }
  }
}

// stage: staging buffer; z: 320 slots; coef: 512 floats (may share memory with `stage`).  any_long / band2_long are
// wave-uniform.  Ends without a fence after the coefficient writes.
__device__ __forceinline__ void mdct_mixed_r4(const float *stage, float2 *z, float *coef, const MixGeometry &G, bool any_long,
                                              bool band2_long, TablesPtr T, TablesRsrc R) {
  float2 x[4];
  // table values are requested one round ahead of their use (see mdct_long_r4)
  const double2 pt0 = table_pair(R, G.pre_tab[0]), pt1 = table_pair(R, G.pre_tab[1]);
  const double2 pt2 = table_pair(R, G.pre_tab[2]), pt3 = table_pair(R, G.pre_tab[3]);
  const double2 wBa = table_pair(R, G.twb), wBb = table_pair(R, G.twb + 64), wBc = table_pair(R, G.twb + 128);
  {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const double a = stage[G.ia[j]], b = stage[G.ib[j]], c = stage[G.ic[j]], d = stage[G.id[j]];
      const double2 t = j == 0 ? pt0 : (j == 1 ? pt1 : (j == 2 ? pt2 : pt3));
      const double r = (j & 1) ? a - b : a + b;             // mdct.js:84-99
      const double m = (j & 1) ? c + d : c - d;
      x[j] = make_float2(f32(r * t.x + m * t.y), f32(m * t.x - r * t.y));
    }
    const double2 w0 = make_double2(T->fft_tw[0][0], T->fft_tw[0][1]);
    const double2 w1 = make_double2(T->fft_tw[1][0], T->fft_tw[1][1]);
    const double2 w2 = make_double2(T->fft_tw[2][0], T->fft_tw[2][1]);
    if (__all(r2_unit_ok(x[0], x[1]) && r2_unit_ok(x[2], x[3]))) { r2_butterfly_unit(x[0], x[1]); r2_butterfly_unit(x[2], x[3]); }
    else { r2_butterfly(x[0], x[1], w0); r2_butterfly(x[2], x[3], w0); }
    if (__all(r2_unit_ok(x[0], x[2]))) r2_butterfly_unit(x[0], x[2]);
    else r2_butterfly(x[0], x[2], w1);
    r2_butterfly(x[1], x[3], w2);
    float4 *dst = reinterpret_cast<float4 *>(z + G.za);
    dst[0] = make_float4(x[0].x, x[0].y, x[1].x, x[1].y);
    dst[1] = make_float4(x[2].x, x[2].y, x[3].x, x[3].y);
  }
  const double2 p0 = table_pair(R, G.post_tab[0]), p1 = table_pair(R, G.post_tab[1]);
  const double2 p2 = table_pair(R, G.post_tab[2]), p3 = table_pair(R, G.post_tab[3]);
  __syncthreads();
  {
    float2 *p = z + G.zb;
    const double2 wa = wBa, wb = wBb, wc = wBc;
    x[0] = p[0]; x[1] = p[4]; x[2] = p[8]; x[3] = p[12];
    r2_butterfly(x[0], x[1], wa); r2_butterfly(x[2], x[3], wa);
    r2_butterfly(x[0], x[2], wb); r2_butterfly(x[1], x[3], wc);
    if (G.is_long) { p[0] = x[0]; p[4] = x[1]; p[8] = x[2]; p[12] = x[3]; }
  }
  if (any_long) {
    __syncthreads();
    if (G.is_long) {
      float2 *p = z + G.zc;
      const double2 wa = table_pair(R, G.twc), wb = table_pair(R, G.twc + 256), wc = table_pair(R, G.twc + 512);
      x[0] = p[0]; x[1] = p[20]; x[2] = p[40]; x[3] = p[60];
      r2_butterfly(x[0], x[1], wa); r2_butterfly(x[2], x[3], wa);
      r2_butterfly(x[0], x[2], wb); r2_butterfly(x[1], x[3], wc);
      if (G.band2) { p[0] = x[0]; p[20] = x[1]; p[40] = x[2]; p[60] = x[3]; }
    }
    if (band2_long) {
      __syncthreads();
      if (G.band2) {
        const float2 *p = z + G.zd;
        const double2 wa = table_pair(R, G.twd), wb = table_pair(R, G.twd + 512);
        x[0] = p[0]; x[1] = p[80]; x[2] = p[40]; x[3] = p[120];
        r2_butterfly(x[0], x[1], wa); r2_butterfly(x[2], x[3], wb);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const double2 t = j == 0 ? p0 : (j == 1 ? p1 : (j == 2 ? p2 : p3));
    const double rr = x[j].x, ii = x[j].y;
    coef[G.cx[j]] = f32(-rr * t.x - ii * t.y);
    coef[G.cy[j]] = f32(-rr * t.y + ii * t.x);
  }
}

template <bool ALL_LONG>
__global__ __launch_bounds__(C1_WAVE, ALL_LONG ? 4 : 3) void k_analysis_fast(C1EncodeLaunch L) {
  using Lds = typename std::conditional<ALL_LONG, LongLds, MixedLds>::type;
  __shared__ Lds S;
  float *band_;                                        // low128 | mid128 | high256 of the current frame, raw
  if constexpr (ALL_LONG) band_ = S.u.m.zz.band; else band_ = S.band;
  const C1DevEncOpts *O = L.opts;
  const int lane0 = threadIdx.x;
  int lane = lane0;
  const int ch = blockIdx.x % L.channels;
  const int64_t f0 = (int64_t)(blockIdx.x / L.channels) * kRunFramesLong;
  const float *__restrict__ pcm = L.pcm[ch];

  for (int i = lane; i < 46; i += 64) { S.d1[i] = 0.0; S.d2[i] = 0.0; }
  for (int i = lane; i < 296; i += 64) S.hbuf[i] = 0.0f;
  if (lane < 12) S.sfi[52 + lane] = 0;   // modes byte (all long) and padding of the side record
  if constexpr (!ALL_LONG) {
    for (int i = lane; i < 96; i += 64) S.ovl[i] = 0.0f;
  }
  float ov0 = 0.0f, ov1 = 0.0f, ov2 = 0.0f;     // lanes 0..31: mdctOverlap of the three bands, carried in registers
  // lane-only geometry of the long-block MDCT core, computed once (everything else is re-derived per frame)
  const R4Geometry G4 = r4_geometry(lane0);
  const SfLong SFL = sf_long_geometry(lane0);
  const int my_size = lane0 < 52 ? kSpecs[lane0] : 0, my_long = lane0 < 52 ? kStartLong[lane0] : 0, my_short = lane0 < 52 ? kStartShort[lane0] : 0;
  const MixGeometry GM = mix_geometry(lane0, FrameModes{O->modes[0], O->modes[1], O->modes[2]});   // used when !ALL_LONG
  const TablesRsrc RT = tables_rsrc(L.tables);
  __syncthreads();

  const int64_t f_end = (f0 + kRunFramesLong < L.frames) ? f0 + kRunFramesLong : L.frames;
  constexpr int kWarm = 1;                         // one frame of history rebuilds the state
  int64_t f_first = f0 - kWarm;
  if (f_first < -(int64_t)L.halo_frames) f_first = -(int64_t)L.halo_frames;   // before the stream start the zero state stays
  if (f_first > f0) f_first = f0;
  // the PCM of the next frame is fetched while the current one is processed (two 16-byte loads per lane)
  float4 pre_a, pre_b;
  {
    const float4 *p4 = reinterpret_cast<const float4 *>(pcm + f_first * 512);
    pre_a = p4[lane0]; pre_b = p4[64 + lane0];
  }
  for (int64_t f = f_first; f < f_end; ++f) {
    const bool emit = (f >= f0);
    TablesPtr T = tables_for_this_frame(L.tables);
    lane = lane_for_this_frame(lane0);

    // ---------------- qmfAnalysisStage (encoder.js:57-96) ----------------
    {
      const float4 a = pre_a, b = pre_b;
      if (f + 1 < f_end) {
        const float4 *p4 = reinterpret_cast<const float4 *>(pcm + (f + 1) * 512);
        pre_a = p4[lane]; pre_b = p4[64 + lane];
      }
      double *w1 = S.u.q1.w1;
      if (lane < 46) w1[pidx<3>(lane)] = S.d1[lane];
      const int e0 = 46 + 4 * lane;
      *reinterpret_cast<double2 *>(&w1[pidx<3>(e0)]) = make_double2((double)a.x, (double)a.y);
      *reinterpret_cast<double2 *>(&w1[pidx<3>(e0 + 2)]) = make_double2((double)a.z, (double)a.w);
      *reinterpret_cast<double2 *>(&w1[pidx<3>(e0 + 256)]) = make_double2((double)b.x, (double)b.y);
      *reinterpret_cast<double2 *>(&w1[pidx<3>(e0 + 258)]) = make_double2((double)b.z, (double)b.w);
    }
    __syncthreads();
    {
      double ev[4], od[4];
      if (own_block()) qmf_analysis_core<4, 3>(S.u.q1.w1, lane, T, ev, od); else { for (int d = 0; d < 4; d++) { ev[d] = S.u.q1.w1[lane + d]; od[d] = 1.0; } }
      double *w2 = S.u.q2.w2;
      if (lane < 46) { w2[pidx<2>(lane)] = S.d2[lane]; S.d1[lane] = S.u.q1.w1[pidx<3>(512 + lane)]; }
      float lo[4];
#pragma unroll
      for (int d = 0; d < 4; d++) {
        lo[d] = f32(ev[d] + od[d]);                       // qmf.js:44-45
        S.hbuf[39 + 4 * lane + d] = f32(ev[d] - od[d]);   // high band enters behind its 39-sample delay
      }
      *reinterpret_cast<double2 *>(&w2[pidx<2>(46 + 4 * lane)]) = make_double2((double)lo[0], (double)lo[1]);
      *reinterpret_cast<double2 *>(&w2[pidx<2>(48 + 4 * lane)]) = make_double2((double)lo[2], (double)lo[3]);
    }
    __syncthreads();
    R4Early EARLY;
    if constexpr (ALL_LONG) EARLY = r4_early(G4, RT);          // in flight during the second QMF stage
    {
      double ev[2], od[2];
      if (own_block()) qmf_analysis_core<2, 2>(S.u.q2.w2, lane, T, ev, od); else { for (int d = 0; d < 2; d++) { ev[d] = S.u.q2.w2[lane + d]; od[d] = 1.0; } }
      *reinterpret_cast<float2 *>(&band_[2 * lane]) = make_float2(f32(ev[0] + od[0]), f32(ev[1] + od[1]));
      *reinterpret_cast<float2 *>(&band_[128 + 2 * lane]) = make_float2(f32(ev[0] - od[0]), f32(ev[1] - od[1]));
      *reinterpret_cast<float4 *>(&band_[256 + 4 * lane]) = *reinterpret_cast<const float4 *>(&S.hbuf[4 * lane]);
      if (lane < 46) S.d2[lane] = S.u.q2.w2[pidx<2>(256 + lane)];
    }
    __syncthreads();
    {
      float keep = 0.0f;
      if (lane < 39) keep = S.hbuf[256 + lane];
      __syncthreads();
      if (lane < 39) S.hbuf[lane] = keep;
    }
    if (emit && L.bands) {
      float4 *dst = reinterpret_cast<float4 *>(L.bands + ((f * L.channels + ch) << 9));
      const float4 *src = reinterpret_cast<const float4 *>(band_);
      dst[lane] = src[lane];
      dst[64 + lane] = src[64 + lane];
    }

    if constexpr (ALL_LONG) {
      // ---------------- mdctStage, long blocks (encoder.js:228-258, 309-316) ----------------
      // tail of every band: windowed copy into this frame's MDCT input, overlap for the next frame
      float *in0 = S.u.m.a.i.in0, *in1 = S.u.m.a.i.in1, *in2 = S.u.m.a.i.in2;
      float nov0 = 0.0f, nov1 = 0.0f, nov2 = 0.0f;
      if (lane < 32) {
        const double w_lo = T->window[lane], w_hi = T->window[31 - lane];
        const double x0 = band_[96 + lane], x1 = band_[128 + 96 + lane], x2 = band_[256 + 224 + lane];
        nov0 = f32(w_lo * x0); nov1 = f32(w_lo * x1); nov2 = f32(w_lo * x2);
        if (emit) {
          in0[48 + lane] = ov0; in1[48 + lane] = ov1; in2[112 + lane] = ov2;     // overlap saved by the previous frame
          in0[80 + 96 + lane] = f32(x0 * w_hi);
          in1[80 + 96 + lane] = f32(x1 * w_hi);
          in2[144 + 224 + lane] = f32(x2 * w_hi);
        }
      }
      ov0 = nov0; ov1 = nov1; ov2 = nov2;
      if (!emit) { __syncthreads(); continue; }
      // zero regions and the body of every band (everything before the tail) straight into the MDCT inputs
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
      __syncthreads();
      float *coef = S.u.m.a.c.coef;
      mdct_long_r4(in0, S.u.zp.z, coef, G4, T, RT, EARLY);
      __syncthreads();

      // ---------------- coefficients out + scale-factor indices (bitallocation.js:80-90) ----------------
      const int64_t unit = f * L.channels + ch;
      {
        float4 *dst = reinterpret_cast<float4 *>(L.coefs + (unit << 9));
        const float4 *src = reinterpret_cast<const float4 *>(coef);
        dst[lane] = src[lane];
        dst[64 + lane] = src[64 + lane];
      }
      sf_long(coef, S.sfi, SFL, T);
      if (!ALL_LONG && lane == 63) S.sfi[52] = 0;   // modes byte: this frame is all long
      __syncthreads();
      if (lane < 16) reinterpret_cast<uint32_t *>(L.side + unit * kSideBytes)[lane] = reinterpret_cast<const uint32_t *>(S.sfi)[lane];
      __syncthreads();
    } else {
      // ---------------- mdctStage with short blocks (encoder.js:170-349), fixed block modes ----------------
      const FrameModes M{O->modes[0], O->modes[1], O->modes[2]};
      float *coef = S.u.m.a.c.coef;
      if (emit) {
        mix_stage(band_, S.ovl, S.u.m.a.g.in, M, lane, RT);
        __syncthreads();
        mdct_mixed_r4(S.u.m.a.g.in, S.u.m.zz.z, coef, GM, M.m0 == 0 || M.m1 == 0 || M.m2 == 0, M.m2 == 0, T, RT);
      }
      // applyTailWindowing's overlap half (encoder.js:309-316): W[i] * last 32 raw samples of the band
      for (int i = lane; i < 96; i += 64) {
        const int b = i >> 5, k = i & 31;
        const int Sb = b == 2 ? 256 : 128, off = b == 0 ? 0 : (b == 1 ? 128 : 256);
        S.ovl[i] = f32(T->window[k] * (double)band_[off + Sb - 32 + k]);
      }
      __syncthreads();
      if (!emit) continue;

      // ---------------- coefficients out + scale-factor indices (bitallocation.js:80-90) ----------------
      const int64_t unit = f * L.channels + ch;
      {
        float4 *dst = reinterpret_cast<float4 *>(L.coefs + (unit << 9));
        const float4 *src = reinterpret_cast<const float4 *>(coef);
        dst[lane] = src[lane];
        dst[64 + lane] = src[64 + lane];
      }
      if (lane < 52) {
        const int start = M.mode_of_band(band_of_bfu(lane)) == 0 ? my_long : my_short;
        const int n = my_size;
        float mx = 0.0f;
        for (int j = 0; j < n; j++) mx = fmaxf(mx, fabsf(coef[start + j]));
        S.sfi[lane] = (uint8_t)(T->sf_fast ? scale_factor_index_fast(mx, T->sf_m1, T->sf_m2) : scale_factor_index(mx, T));
      } else {
        S.sfi[lane] = lane == 52 ? (uint8_t)((M.m0 & 3) | ((M.m1 & 3) << 2) | ((M.m2 & 3) << 4)) : 0;
      }
      __syncthreads();
      if (lane < 16) reinterpret_cast<uint32_t *>(L.side + unit * kSideBytes)[lane] = reinterpret_cast<const uint32_t *>(S.sfi)[lane];
      __syncthreads();
    }
  }
}

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

__device__ __forceinline__ int tslot(int pos) { return pos + (pos >> 3); }

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

__global__ __launch_bounds__(C1_WAVE, 3) void k_detect_features(C1EncodeLaunch L, float *bands_ws, double *feat_ws) {
  __shared__ DetectLds S;
  const int lane0 = threadIdx.x;
  int lane = lane0;
  const int ch = blockIdx.x % L.channels;
  const int64_t f0 = (int64_t)(blockIdx.x / L.channels) * kRunFramesLong;
  const float *__restrict__ pcm = L.pcm[ch];
  for (int i = lane; i < 46; i += 64) { S.d1[i] = 0.0; S.d2[i] = 0.0; }
  for (int i = lane; i < 296; i += 64) S.hbuf[i] = 0.0f;
  float pmag[4] = {0.0f, 0.0f, 0.0f, 0.0f};      // magnitudes of the previous frame at this lane's four bins
  // ---- lane-only geometry of the transient FFT: lanes 0..15 band 0 (128 points), 16..31 band 1, 32..63 band 2 (256)
  const int tband = lane0 < 16 ? 0 : (lane0 < 32 ? 1 : 2);
  const int tg = lane0 - (tband == 0 ? 0 : (tband == 1 ? 16 : 32));
  const int tS = tband == 2 ? 32 : 16;                                   // N/8: sample stride of round A, point stride of round C
  const int t_src = (tband == 0 ? 0 : (tband == 1 ? 128 : 256)) + bitrev(tg, tband == 2 ? 5 : 4);
  const int t_pbase = tband == 0 ? 0 : (tband == 1 ? 128 : 256);
  const int t_za = tslot(t_pbase + 8 * tg);
  const int t_zb = tslot(t_pbase + 64 * (tg >> 3) + (tg & 7));
  const int t_twb = (int)offsetof(C1DevTables, fft_tw) + 16 * (7 + (tg & 7));
  const int t_zc = tslot(t_pbase + tg), t_zc_stride = tS + tS / 8;
  const int t_twc = (int)offsetof(C1DevTables, fft_tw) + 16 * ((tband == 2 ? 127 : 63) + tg), t_twc_stride = 16 * tS;
  const int t_twd = (int)offsetof(C1DevTables, fft_tw) + 16 * (63 + (tg & 31));
  const int t_mag = (tband == 0 ? 0 : (tband == 1 ? 64 : 128)) + tg;   // mag index of the lane's first bin; next bins + tS
  const TablesRsrc RT = tables_rsrc(L.tables);
  __syncthreads();

  const int64_t f_end = (f0 + kRunFramesLong < L.frames) ? f0 + kRunFramesLong : L.frames;
  int64_t f_first = f0 - 2;                                  // frame -2 rebuilds the QMF delay lines, frame -1 the magnitudes
  if (f_first < -(int64_t)L.halo_frames) f_first = -(int64_t)L.halo_frames;
  if (f_first > f0) f_first = f0;
  float4 pre_a, pre_b;
  {
    const float4 *p4 = reinterpret_cast<const float4 *>(pcm + f_first * 512);
    pre_a = p4[lane0]; pre_b = p4[64 + lane0];
  }
  for (int64_t f = f_first; f < f_end; ++f) {
    // frame -1 of the whole batch is emitted too (slot row 0): frame 0 needs its features and its band tails
    const bool emit = (f >= f0) || (f0 == 0 && f == -1);
    const bool qmf_only = (f == f0 - 2);
    TablesPtr T = tables_for_this_frame(L.tables);
    lane = lane_for_this_frame(lane0);

    // ---------------- qmfAnalysisStage (encoder.js:57-96) ----------------
    {
      const float4 a = pre_a, b = pre_b;
      if (f + 1 < f_end) {
        const float4 *p4 = reinterpret_cast<const float4 *>(pcm + (f + 1) * 512);
        pre_a = p4[lane]; pre_b = p4[64 + lane];
      }
      double *w1 = S.u.q1.w1;
      if (lane < 46) w1[pidx<3>(lane)] = S.d1[lane];
      const int e0 = 46 + 4 * lane;
      *reinterpret_cast<double2 *>(&w1[pidx<3>(e0)]) = make_double2((double)a.x, (double)a.y);
      *reinterpret_cast<double2 *>(&w1[pidx<3>(e0 + 2)]) = make_double2((double)a.z, (double)a.w);
      *reinterpret_cast<double2 *>(&w1[pidx<3>(e0 + 256)]) = make_double2((double)b.x, (double)b.y);
      *reinterpret_cast<double2 *>(&w1[pidx<3>(e0 + 258)]) = make_double2((double)b.z, (double)b.w);
    }
    __syncthreads();
    {
      double ev[4], od[4];
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
    __syncthreads();
    {
      double ev[2], od[2];
      if (own_block()) qmf_analysis_core<2, 2>(S.u.q2.w2, lane, T, ev, od); else { for (int d = 0; d < 2; d++) { ev[d] = S.u.q2.w2[lane + d]; od[d] = 1.0; } }
      *reinterpret_cast<float2 *>(&S.band[2 * lane]) = make_float2(f32(ev[0] + od[0]), f32(ev[1] + od[1]));
      *reinterpret_cast<float2 *>(&S.band[128 + 2 * lane]) = make_float2(f32(ev[0] - od[0]), f32(ev[1] - od[1]));
      *reinterpret_cast<float4 *>(&S.band[256 + 4 * lane]) = *reinterpret_cast<const float4 *>(&S.hbuf[4 * lane]);
      if (lane < 46) S.d2[lane] = S.u.q2.w2[pidx<2>(256 + lane)];
    }
    __syncthreads();
    {
      float keep = 0.0f;
      if (lane < 39) keep = S.hbuf[256 + lane];
      __syncthreads();
      if (lane < 39) S.hbuf[lane] = keep;
    }
    if (qmf_only) { __syncthreads(); continue; }
    const int64_t slot = (f + 1) * L.channels + ch;
    if (emit) {
      float4 *dst = reinterpret_cast<float4 *>(bands_ws + (slot << 9));
      const float4 *src = reinterpret_cast<const float4 *>(S.band);
      dst[lane] = src[lane];
      dst[64 + lane] = src[64 + lane];
    }

    // ---------------- performFFT (transient.js:17-35) in radix-8 rounds ----------------
    float2 x[8];
    {
      const float *src = S.band + t_src;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int jr = ((j & 1) << 2) | (j & 2) | (j >> 2);          // bitrev3
        x[j] = make_float2(src[jr * tS], 0.0f);
      }
    }
    // twiddles of round B are requested before round A computes, those of round C before round B
    const double2 w8 = table_pair(RT, t_twb), w16a = table_pair(RT, t_twb + 128), w16b = table_pair(RT, t_twb + 256);
    const double2 w32a = table_pair(RT, t_twb + 384), w32b = table_pair(RT, t_twb + 512);
    const double2 w32c = table_pair(RT, t_twb + 640), w32d = table_pair(RT, t_twb + 768);
    tfft_round_a(x, T);
    float2 *z = S.u.t.z;
    {
      float4 *dst = reinterpret_cast<float4 *>(z + t_za);
#pragma unroll
      for (int j = 0; j < 4; j++) dst[j] = make_float4(x[2 * j].x, x[2 * j].y, x[2 * j + 1].x, x[2 * j + 1].y);
    }
    __syncthreads();
    {
      float2 *p = z + t_zb;                                    // stages 8, 16, 32 on the points p + 8j
#pragma unroll
      for (int j = 0; j < 8; j++) x[j] = p[9 * j];
      r2_butterfly(x[0], x[1], w8); r2_butterfly(x[2], x[3], w8); r2_butterfly(x[4], x[5], w8); r2_butterfly(x[6], x[7], w8);
      r2_butterfly(x[0], x[2], w16a); r2_butterfly(x[1], x[3], w16b); r2_butterfly(x[4], x[6], w16a); r2_butterfly(x[5], x[7], w16b);
      r2_butterfly(x[0], x[4], w32a); r2_butterfly(x[1], x[5], w32b); r2_butterfly(x[2], x[6], w32c); r2_butterfly(x[3], x[7], w32d);
#pragma unroll
      for (int j = 0; j < 8; j++) p[9 * j] = x[j];
    }
    const double2 wDa = table_pair(RT, t_twd), wDb = table_pair(RT, t_twd + 512);
    const double2 wC0 = table_pair(RT, t_twc), wC1 = table_pair(RT, t_twc + t_twc_stride);
    const double2 wC2 = table_pair(RT, t_twc + 2 * t_twc_stride), wC3 = table_pair(RT, t_twc + 3 * t_twc_stride);
    __syncthreads();
    float mg[4];
    {
      // points g + tS*t, t = 0..7.  Band 2 first runs stage 64 on them; then stage N/2 (64 for the 128-point
      // transforms, 128 for the 256-point one) pairs (t, t+4) and only its e-outputs, the bins g + tS*t, are needed
      const float2 *p = z + t_zc;
#pragma unroll
      for (int t = 0; t < 8; t++) x[t] = p[t * t_zc_stride];
      if (tband == 2) {
        r2_butterfly(x[0], x[2], wDa); r2_butterfly(x[1], x[3], wDb); r2_butterfly(x[4], x[6], wDa); r2_butterfly(x[5], x[7], wDb);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float2 e = r2_butterfly_e(x[i], x[i + 4], i == 0 ? wC0 : (i == 1 ? wC1 : (i == 2 ? wC2 : wC3)));
        const double r = e.x, im = e.y;
        mg[i] = f32(sqrt(r * r + im * im));
      }
    }
    __syncthreads();                                           // the per-bin terms reuse the memory of the points
    if (emit) {
      // ---------------- feature terms per bin, then the reference's sequential sums ----------------
      bool valid[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int g = t_mag + i * tS;
        const double cm = (double)mg[i], pm = (double)pmag[i];
        const double diff = cm - pm;
        valid[i] = cm > 1e-10;
        S.u.tt.term[0][g] = diff > 0 ? diff : 0.0;            // spectral flux terms (transient.js:96-106)
        S.u.tt.term[1][g] = cm * cm;                          // energy terms (exact product)
        S.u.tt.term[2][g] = valid[i] ? log(cm) : 0.0;         // flatness terms (transient.js:126-133)
        S.u.tt.term[3][g] = valid[i] ? cm : 0.0;
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
      __syncthreads();
      double *feat = feat_ws + slot * kFeatureDoubles;
      if (lane < 18) {
        // 18 lanes each own one running sum (3 bands x {flux, energy, log, linear, low, high}), index ascending
        const int b = lane / 6, kind = lane - 6 * b;
        const int n = b == 2 ? 128 : 64, g0 = b == 0 ? 0 : (b == 1 ? 64 : 128);
        const int which = kind == 0 ? 0 : (kind == 2 ? 2 : (kind == 3 ? 3 : 1));
        const int start = g0 + (kind == 5 ? n / 2 : 0);
        const int len = kind >= 4 ? n / 2 : n;
        const double2 *arr = reinterpret_cast<const double2 *>(S.u.tt.term[which] + start);
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
#pragma unroll
    for (int i = 0; i < 4; i++) pmag[i] = mg[i];
    __syncthreads();
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
    const double gm = exp(s_log / (double)nv), am = s_lin / (double)nv;
    r.flat = am > 1e-10 ? gm / am : 0.0;
  }
  const double tot = s_lo + s_hi;                        // calculateHighFrequencyRatio :149-164
  r.hf = tot > 0 ? s_hi / tot : 0.0;
  r.energy = s_e;
  return r;
}

// block modes of one sound unit from the feature sums of its frame and of the previous one (encoder.js:137-143)
__device__ __forceinline__ int detect_decide_unit(const double *__restrict__ feat_ws, int channels, int64_t unit, int halo_frames,
                                                       const C1DevTables *tables, const C1DevEncOpts *opts) {
  const int64_t f = unit / channels;
  const double *cur = feat_ws + (unit + channels) * kFeatureDoubles;
  const double *prev = cur - (int64_t)channels * kFeatureDoubles;
  const bool have_prev = (f - 1 >= -(int64_t)halo_frames);      // else the zero state of a fresh BufferPool
  const double log1p10 = tables->log1p10, threshold = opts->threshold;
  int mode_byte = 0;
  for (int b = 0; b < 3; b++) {
    const BandFeatures c = band_features(cur + 6 * b, reinterpret_cast<const int *>(cur + 18)[b]);
    double prev_flat = 0.0, prev_hf = 0.0, prev_e = 0.0;
    if (have_prev) {
      const BandFeatures p = band_features(prev + 6 * b, reinterpret_cast<const int *>(prev + 18)[b]);
      prev_flat = p.flat; prev_hf = p.hf; prev_e = p.energy;
    }
    const double ce = c.energy > 1e-10 ? c.energy : 1e-10;     // calculateEnergyChange :172-189
    const double pe = prev_e > 1e-10 ? prev_e : 1e-10;
    const double db = 10.0 * log10(ce / pe);
    const double e_change = db > 0 ? db : 0.0;
    const double flat_c = sqrt(fabs(c.flat - prev_flat));       // calculateTransientScore :197-226
    const double hf_c = log1p(fabs(c.hf - prev_hf) * 10.0) / log1p10;
    const double e_c = e_change / 30.0 < 1.0 ? e_change / 30.0 : 1.0;
    const double score = (c.flux + flat_c + hf_c + e_c) / 4.0;
    const int mode = (score > threshold) ? (b + 1 > 2 ? b + 1 : 2) : 0;   // encoder.js:143
    mode_byte |= mode << (2 * b);
  }
  return mode_byte;
}

__global__ __launch_bounds__(256) void k_detect_decide(const double *__restrict__ feat_ws, int channels, int64_t frames,
                                                        int halo_frames, const C1DevTables *tables, const C1DevEncOpts *opts,
                                                        uint8_t *__restrict__ modes, uint32_t *__restrict__ lists) {
  const int64_t unit = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t units = frames * channels;
  const bool live = unit < units;
  int mode_byte = 0;
  if (live) mode_byte = detect_decide_unit(feat_ws, channels, unit, halo_frames, tables, opts);
  if (live) modes[unit] = (uint8_t)mode_byte;
  // two work lists for the MDCT stage: all-long units and units with a short band (lists[0], lists[1] = counts,
  // then `units` entries each); one atomic per wave and list
  uint32_t *list_long = lists + 4, *list_mixed = lists + 4 + units;
  const bool is_long = live && mode_byte == 0, is_mixed = live && mode_byte != 0;
  const uint64_t ml = __ballot(is_long), mm = __ballot(is_mixed);
  const int lane = threadIdx.x & 63;
  const uint64_t below = (1ull << lane) - 1ull;
  uint32_t base_l = 0, base_m = 0;
  if (lane == 0) {
    if (ml) base_l = atomicAdd(&lists[0], (uint32_t)__popcll(ml));
    if (mm) base_m = atomicAdd(&lists[1], (uint32_t)__popcll(mm));
  }
  base_l = __shfl(base_l, 0); base_m = __shfl(base_m, 0);
  if (is_long) list_long[base_l + __popcll(ml & below)] = (uint32_t)unit;
  if (is_mixed) list_mixed[base_m + __popcll(mm & below)] = (uint32_t)unit;
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

// mdctStage + scale factors of one sound unit from the stored band samples; units are independent.  Two
// instantiations work through the two lists k_detect_decide wrote: LONG (all three bands long, the common case;
// lean enough for 4 waves per SIMD) and mixed (at least one short band).
template <bool LONG>
__global__ __launch_bounds__(C1_WAVE, 3) void k_mdct_bands(C1EncodeLaunch L, const float *__restrict__ bands_ws,
                                                                         const uint8_t *__restrict__ modes,
                                                                         const uint32_t *__restrict__ lists) {
  __shared__ MdctLds S;
  const int lane0 = threadIdx.x;
  int lane = lane0;
  const int64_t units = L.frames * L.channels;
  const uint32_t count = lists[LONG ? 0 : 1];
  const uint32_t *__restrict__ list = lists + 4 + (LONG ? 0 : units);
  const R4Geometry G4 = r4_geometry(lane0);          // LONG
  const SfLong SFL = sf_long_geometry(lane0);        // LONG
  const int my_size = lane0 < 52 ? kSpecs[lane0] : 0, my_long = lane0 < 52 ? kStartLong[lane0] : 0, my_short = lane0 < 52 ? kStartShort[lane0] : 0;
  const TablesRsrc RT = tables_rsrc(L.tables);
  // tails of the previous frame: lane < 24 loads four samples of band lane / 8
  const int tail_band = lane0 >> 3, tail_k = 4 * (lane0 & 7);
  const int tail_src = (tail_band == 0 ? 96 : (tail_band == 1 ? 224 : 480)) + tail_k;
  // window values the lane needs every unit: fixed per lane, read once
  const double wt0 = C1_TABLES(L.tables)->window[tail_k & 31], wt1 = C1_TABLES(L.tables)->window[(tail_k + 1) & 31];
  const double wt2 = C1_TABLES(L.tables)->window[(tail_k + 2) & 31], wt3 = C1_TABLES(L.tables)->window[(tail_k + 3) & 31];
  const double win_hi = C1_TABLES(L.tables)->window[31 - (lane0 & 31)];
  uint32_t i = blockIdx.x;
  if (i >= count) return;
  float4 pre_a, pre_b, pre_t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  int pre_mode = 0;
  auto fetch = [&](int64_t u) {
    const int64_t slot = u + L.channels;
    const float4 *p4 = reinterpret_cast<const float4 *>(bands_ws + (slot << 9));
    pre_a = p4[lane0]; pre_b = p4[64 + lane0];
    const bool have_prev = ((L.channels == 2 ? u >> 1 : u) - 1 >= -(int64_t)L.halo_frames);
    pre_t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (have_prev && lane0 < 24) pre_t = *reinterpret_cast<const float4 *>(bands_ws + ((slot - L.channels) << 9) + tail_src);
    if (!LONG) pre_mode = modes[u];
  };
  int64_t unit = list[i];
  int64_t unit_next = i + gridDim.x < count ? list[i + gridDim.x] : 0;
  fetch(unit);
  for (; i < count; i += gridDim.x) {
    TablesPtr T = tables_for_this_frame(L.tables);
    lane = lane_for_this_frame(lane0);
    const float4 a = pre_a, b = pre_b, t = pre_t;
    const int mode_byte = LONG ? 0 : __builtin_amdgcn_readfirstlane(pre_mode);
    const int64_t unit_now = unit;
    if (i + gridDim.x < count) {
      unit = unit_next;
      fetch(unit);
      if (i + 2 * gridDim.x < count) unit_next = list[i + 2 * gridDim.x];
    }
    reinterpret_cast<float4 *>(S.band)[lane] = a;
    reinterpret_cast<float4 *>(S.band)[64 + lane] = b;
    if (lane < 24) {
      // mdctOverlap of the previous frame (applyTailWindowing, encoder.js:309-316): W[k] * tail sample
      float4 o;
      o.x = f32(wt0 * (double)t.x); o.y = f32(wt1 * (double)t.y);
      o.z = f32(wt2 * (double)t.z); o.w = f32(wt3 * (double)t.w);
      reinterpret_cast<float4 *>(S.ovl)[lane] = o;
    }
    __syncthreads();
    const FrameModes M{mode_byte & 3, (mode_byte >> 2) & 3, (mode_byte >> 4) & 3};
    float *coef = S.a.c.coef;
    if constexpr (LONG) {
      // ---------------- long blocks (encoder.js:228-258) ----------------
      float *in0 = S.a.i.in0, *in1 = S.a.i.in1, *in2 = S.a.i.in2;
      const float *band_ = S.band;
      if (lane < 32) {
        const double w_hi = win_hi;
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
      __syncthreads();
      mdct_long_r4(in0, S.zz.z, coef, G4, T, RT, r4_early(G4, RT));
      __syncthreads();
    } else {
      const MixGeometry GM = mix_geometry(lane, M);
      mix_stage(S.band, S.ovl, S.a.g.in, M, lane, RT);
      __syncthreads();
      mdct_mixed_r4(S.a.g.in, S.zz.z, coef, GM, M.m0 == 0 || M.m1 == 0 || M.m2 == 0, M.m2 == 0, T, RT);
      __syncthreads();
    }
    // ---------------- coefficients out + scale-factor indices (bitallocation.js:80-90) ----------------
    {
      float4 *dst = reinterpret_cast<float4 *>(L.coefs + (unit_now << 9));
      const float4 *src = reinterpret_cast<const float4 *>(coef);
      dst[lane] = src[lane];
      dst[64 + lane] = src[64 + lane];
    }
    if constexpr (LONG) {
      sf_long(coef, S.sfi, SFL, T);
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
    __syncthreads();
    if (lane < 16) reinterpret_cast<uint32_t *>(L.side + unit_now * kSideBytes)[lane] = reinterpret_cast<const uint32_t *>(S.sfi)[lane];
    __syncthreads();
  }
}

// =====================================================================================================
// bit allocation : allocateBits (bitallocation.js:74-142) as three small kernels
// =====================================================================================================
// The reference runs the greedy heap (distributeBitsRDO, :203-281) for all 8 candidate BFU counts and
// keeps the one with the smallest total distortion (:116-129).  Each candidate is independent, so:
//   k_alloc_first   one lane per sound unit: the 52-BFU candidate.  It then bounds every other candidate
//                   from below without running it: a candidate that codes n BFUs pays at least the
//                   zero-bit distortion of BFUs n..51, summed in the reference's own order (floating-point
//                   addition is monotone, every term is >= 0, so the bound holds for the rounded sums too).
//                   A candidate whose bound already exceeds the 52-BFU total can never win the strict `<`
//                   comparison and is skipped; the others are appended to a work list.
//   k_alloc_rest    one lane per work-list entry (unit, candidate): the same heap run.
//   k_alloc_select  one lane per unit: smallest total, smallest count on ties (:116-129), or the
//                   fallback when no total is finite (:132-139).
//
// Heap entry (one 32-bit word, bit 31 clear):  rank(10) | size(5) | sfi(6) | wl(4) | bfu(6).
// `rank` orders the Float32 priorities biasedSF[sfi]*DISTORTION_DELTA_FACTORS[wl]/WORD_LENGTH_DELTA_BITS[wl]
// (bitallocation.js:226-231,267-269) exactly: equal priorities have equal rank, so the strict `>` of
// siftDown (:325-331) -- and with it the tie order -- is reproduced.  With kLow = the 21 payload bits,
// rank(a) > rank(b)  <=>  a > (b | kLow): one integer compare, no field extraction.  Rank 0 never occurs
// in a live entry, so zeroed slots and parked entries (below) act as sentinels: no bounds checks in the sift.
//
// Heaps live in LDS as heap[slot][lane] (64 dwords per slot: conflict-free, the two children of a node one
// ds_read2st64 apart).  An entry that leaves the heap is parked, rank cleared, in the slot the shrinking
// heap frees, so when the loop ends slots [0, initial size) hold every BFU with its final word length.
constexpr int kHeapSlotsPerLane = 52 + 2;   // + two zero sentinels behind the last slot
constexpr uint32_t kLow = 0x1FFFFFu;
constexpr int kCandBytes = kCandidateBytes;  // per unit: 8 totals (double) + 8 x 32-byte results

__device__ __forceinline__ uint32_t heap_entry(uint32_t rank, int size, int s, int wl, int b) {
  return (rank << 21) | ((uint32_t)size << 16) | ((uint32_t)s << 10) | ((uint32_t)wl << 6) | (uint32_t)b;
}

// siftDown (bitallocation.js:314-341) for every lane of the wave at once.  `el`,`er` are the already
// loaded children of `i`; lanes finish at different depths, the loop runs while any lane still moves.
__device__ __forceinline__ void heap_sift_down(uint32_t *hp, int sentinel, int i, uint32_t v, uint32_t el,
                                               uint32_t er, bool active) {
  const uint32_t vmax = v | kLow;
  for (;;) {
    const uint32_t m = max(el | kLow, vmax);
    const bool take_r = er > m;                    // pr > max(pl, pv)
    const bool take_l = !take_r && el > vmax;      // pl > pv
    const bool moved = active && (take_r || take_l);
    if (active) hp[i * 64] = moved ? (take_r ? er : el) : v;   // a lane that stops here drops v into place
    active = moved;
    if (__builtin_amdgcn_ballot_w64(active) == 0) break;
    i = moved ? 2 * i + 1 + (take_r ? 1 : 0) : i;
    const uint32_t *src = hp + min(2 * i + 1, sentinel) * 64;
    el = src[0];
    er = src[64];
  }
}

// One greedy heap per lane: candidate with `n` BFUs.  sf = the unit's 52 scale-factor indices (13 dwords).
// Returns the 52 final word-length indices (4 bits each) and the candidate's total distortion.
// During the spending loop the three top slots of the heap live in registers (r0 = root, r1/r2 = its
// children): most steps end at the root (equal priorities do not move, :325-331), so they touch no memory.
__device__ __forceinline__ void run_candidate(uint32_t *hp, int n, const uint32_t (&sf)[13], const C1DevEncOpts *O,
                                              bool live, uint64_t &res0, uint64_t &res1, uint64_t &res2,
                                              uint64_t &res3, double &total) {
  const __attribute__((address_space(4))) uint16_t *rank_t = (const __attribute__((address_space(4))) uint16_t *)O->rank;
  const __attribute__((address_space(4))) double *biased = (const __attribute__((address_space(4))) double *)O->biased;
  const bool affine = O->rank_affine != 0;
  const int ka = O->rank_a, kb = O->rank_b, kc = O->rank_c, koff = O->rank_off;
  auto rank_of = [&](int s, int wl) -> uint32_t {
    if (affine) return (uint32_t)(ka * s + (wl == 0 ? kc : -kb * wl - kb) + koff);
    return rank_t[s * 16 + (wl & 15)];
  };
  int remaining = 212 * 8 - 40 - 10 * n;                   // bitallocation.js:97-100
  int hs = 0;
  // distributeBitsRDO (:203-281): initial heap = BFUs below n with a non-zero scale factor
#pragma unroll
  for (int b = 0; b < 52; b++) {
    const int s = (sf[b >> 2] >> ((b & 3) * 8)) & 63;
    if (b < n && s != 0 && live) {
      hp[hs * 64] = heap_entry(rank_of(s, 0), kSpecs[b], s, 0, b);
      hs++;
    }
  }
  const int hs0 = hs;
  for (int k = hs; k < kHeapSlotsPerLane; k++) hp[k * 64] = 0u;   // sentinels
  for (int i = (hs >> 1) - 1; i >= 0; i--) {               // heapify (:238-241); per-lane trip counts
    const int l = 2 * i + 1;
    heap_sift_down(hp, 52, i, hp[i * 64], hp[l * 64], hp[(l + 1) * 64], true);
  }
  // greedy spending loop (:244-278): the root either takes its next priority or leaves the heap (does not
  // fit :251-258, or reached the last word length :271-277); then one sift.
  uint32_t r0 = hp[0], r1 = hp[64], r2 = hp[128];
  bool run = remaining > 0 && hs > 0;
  while (__builtin_amdgcn_ballot_w64(run) != 0) {
    const uint32_t top = r0;
    const int wl = (top >> 6) & 15, size = (top >> 16) & 31, s = (top >> 10) & 63;
    const int cost = size << (wl == 0 ? 1 : 0);            // WORD_LENGTH_DELTA_BITS = [2,1,1,...]
    const bool fits = cost <= remaining;
    const int nxt = wl + (fits ? 1 : 0);
    const bool leaves = run && (!fits || nxt >= 15);
    const uint32_t upd = (top & ~((0x3FFu << 21) | (15u << 6))) | ((uint32_t)nxt << 6);   // same BFU, new word length, rank 0
    uint32_t v = upd | (rank_of(s, nxt) << 21);
    if (__builtin_amdgcn_ballot_w64(leaves) != 0) {
      // the last element replaces the root; the leaver is parked, rank 0, in the slot that frees
      const int li = hs - 1;
      const uint32_t deep = hp[(li > 3 ? li : 3) * 64];
      const uint32_t last = li == 0 ? r0 : (li == 1 ? r1 : (li == 2 ? r2 : deep));
      if (leaves) {
        v = last;
        hs = li;
        if (li > 2) hp[li * 64] = upd;
        r0 = li == 0 ? upd : r0; r1 = li == 1 ? upd : r1; r2 = li == 2 ? upd : r2;
      }
    }
    if (run) remaining -= fits ? cost : 0;
    const bool sift = run && hs > 0;
    // level 0: root against r1, r2
    const uint32_t vmax = v | kLow;
    const bool tr0 = r2 > max(r1 | kLow, vmax);
    const bool tl0 = !tr0 && r1 > vmax;
    const bool mv0 = sift && (tr0 || tl0);
    if (sift) r0 = mv0 ? (tr0 ? r2 : r1) : v;
    if (__builtin_amdgcn_ballot_w64(mv0) != 0) {
      // level 1: the chosen child's children are slots 3,4 or 5,6
      const int i1 = tr0 ? 2 : 1;
      const uint32_t *src = hp + (2 * i1 + 1) * 64;
      const uint32_t el = src[0], er = src[64];
      const bool tr1 = er > max(el | kLow, vmax);
      const bool tl1 = !tr1 && el > vmax;
      const bool mv1 = mv0 && (tr1 || tl1);
      const uint32_t val = mv1 ? (tr1 ? er : el) : v;
      if (mv0) { if (tr0) r2 = val; else r1 = val; }
      if (__builtin_amdgcn_ballot_w64(mv1) != 0) {
        const int i2 = 2 * i1 + 1 + (tr1 ? 1 : 0);
        const uint32_t *s2 = hp + (2 * i2 + 1) * 64;     // i2 <= 6: children 7..14 always inside the lane's slots
        heap_sift_down(hp, 52, i2, v, s2[0], s2[64], mv1);
      }
    }
    run = run && remaining > 0 && hs > 0;
  }
  hp[0] = r0; hp[64] = r1; hp[128] = r2;
  // every BFU that ever entered the heap now sits in slots [0, hs0) with its final word length
  res0 = res1 = res2 = res3 = 0;
  for (int k = 0; k < hs0; k++) {
    const uint32_t e = hp[k * 64];
    const int b = e & 63;
    const uint64_t v = (uint64_t)((e >> 6) & 15) << ((b & 15) * 4);
    const int w = b >> 4;
    res0 |= w == 0 ? v : 0; res1 |= w == 1 ? v : 0; res2 |= w == 2 ? v : 0; res3 |= w == 3 ? v : 0;
  }
  // calculateTotalDistortion (:157-190): sequential double sum, index ascending
  total = 0.0;
#pragma unroll
  for (int b = 0; b < 52; b++) {
    const int s = (sf[b >> 2] >> ((b & 3) * 8)) & 63;
    const uint64_t word = b < 16 ? res0 : (b < 32 ? res1 : (b < 48 ? res2 : res3));
    const int wl = (int)((word >> ((b & 15) * 4)) & 15);
    const int size = kSpecs[b];
    if (b >= n || wl == 0) {
      // zeroBitDistortions[b] = Float32(biasedSF * 2 * size), 0 when sfi == 0 (:76,87-89)
      total += s != 0 ? (double)f32(biased[s] * 2.0 * (double)size) : 0.0;
    } else if (s != 0) {
      const double ip2 = __hiloint2double((1023 - wl_bits(wl)) << 20, 0);   // INV_POWER_OF_TWO[bits] = 2^-bits
      total += biased[s] * ip2 * (double)size;
    }
  }
}

__device__ __forceinline__ void load_sfi(const uint8_t *side, int64_t unit, uint32_t (&sf)[13]) {
  const uint4 *src = reinterpret_cast<const uint4 *>(side + unit * kSideBytes);
  const uint4 a = src[0], b = src[1], c = src[2];
  const uint32_t d = reinterpret_cast<const uint32_t *>(src)[12];
  sf[0] = a.x; sf[1] = a.y; sf[2] = a.z; sf[3] = a.w; sf[4] = b.x; sf[5] = b.y; sf[6] = b.z; sf[7] = b.w;
  sf[8] = c.x; sf[9] = c.y; sf[10] = c.z; sf[11] = c.w; sf[12] = d;
}

__device__ __forceinline__ void store_candidate(uint8_t *cand, int64_t unit, int c, double total, uint64_t r0,
                                                uint64_t r1, uint64_t r2, uint64_t r3) {
  uint8_t *base = cand + unit * kCandBytes;
  reinterpret_cast<double *>(base)[c] = total;
  uint64_t *dst = reinterpret_cast<uint64_t *>(base + 64 + c * 32);
  dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3;
}

__global__ __launch_bounds__(C1_WAVE, 3) void k_alloc_first(C1EncodeLaunch L) {
  __shared__ uint32_t heap[kHeapSlotsPerLane * 64];
  const C1DevEncOpts *O = L.opts;
  const __attribute__((address_space(4))) double *biased = (const __attribute__((address_space(4))) double *)O->biased;
  const int lane = threadIdx.x;
  const int64_t units_total = L.frames * L.channels;
  const int64_t unit = (int64_t)blockIdx.x * 64 + lane;
  const bool live = unit < units_total;
  uint32_t sf[13];
  load_sfi(L.side, live ? unit : 0, sf);
  uint64_t r0, r1, r2, r3;
  double total;
  run_candidate(heap + lane, 52, sf, O, live, r0, r1, r2, r3, total);
  // lower bounds of the other seven candidates: zero-bit distortion of the BFUs they do not code, each
  // summed from its first uncoded BFU upwards (one pass, seven running sums)
  uint32_t survivors = 0;
  {
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
#pragma unroll
    for (int b = 20; b < 52; b++) {
      const int s = (sf[b >> 2] >> ((b & 3) * 8)) & 63;
      const double z = s != 0 ? (double)f32(biased[s] * 2.0 * (double)kSpecs[b]) : 0.0;
      t0 += z;
      if (b >= 28) t1 += z;
      if (b >= 32) t2 += z;
      if (b >= 36) t3 += z;
      if (b >= 40) t4 += z;
      if (b >= 44) t5 += z;
      if (b >= 48) t6 += z;
    }
    // skip only when the bound is strictly above a finite 52-BFU total; NaN / Inf totals prune nothing
    const bool finite = total < __builtin_huge_val();
    survivors = (!(finite && t0 > total) ? 1u : 0u) | (!(finite && t1 > total) ? 2u : 0u) |
                (!(finite && t2 > total) ? 4u : 0u) | (!(finite && t3 > total) ? 8u : 0u) |
                (!(finite && t4 > total) ? 16u : 0u) | (!(finite && t5 > total) ? 32u : 0u) |
                (!(finite && t6 > total) ? 64u : 0u);
  }
  if (live) {
    uint8_t *base = L.cand + unit * kCandBytes;
    double *tot = reinterpret_cast<double *>(base);
#pragma unroll
    for (int c = 0; c < 7; c++) tot[c] = __builtin_huge_val();
    store_candidate(L.cand, unit, 7, total < __builtin_huge_val() ? total : __builtin_huge_val(), r0, r1, r2, r3);
  }
  // append the surviving (unit, candidate) pairs to the work list: one atomic per wave
  const int mine = live ? __popc(survivors) : 0;
  int scan = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(scan, d);
    if (lane >= d) scan += o;
  }
  const int wave_total = __shfl(scan, 63);
  uint32_t base_idx = 0;
  if (lane == 0 && wave_total > 0) base_idx = atomicAdd(L.work_count, (uint32_t)wave_total);
  base_idx = __shfl(base_idx, 0);
  if (mine > 0) {
    uint32_t at = base_idx + (uint32_t)(scan - mine);
    for (int c = 0; c < 7; c++)
      if ((survivors >> c) & 1u) L.work_list[at++] = ((uint32_t)(unit - (int64_t)0) << 3) | (uint32_t)c;
  }
}

__global__ __launch_bounds__(C1_WAVE, 3) void k_alloc_rest(C1EncodeLaunch L) {
  __shared__ uint32_t heap[kHeapSlotsPerLane * 64];
  const C1DevEncOpts *O = L.opts;
  const int lane = threadIdx.x;
  const uint32_t count = *L.work_count;
  for (uint32_t base = blockIdx.x * 64u; base < count; base += gridDim.x * 64u) {
    const uint32_t idx = base + lane;
    const bool live = idx < count;
    const uint32_t item = live ? L.work_list[idx] : 0u;
    const int64_t unit = item >> 3;
    const int c = item & 7;
    uint32_t sf[13];
    load_sfi(L.side, unit, sf);
    uint64_t r0, r1, r2, r3;
    double total;
    run_candidate(heap + lane, bfu_amount(c), sf, O, live, r0, r1, r2, r3, total);
    if (live) store_candidate(L.cand, unit, c, total < __builtin_huge_val() ? total : __builtin_huge_val(), r0, r1, r2, r3);
  }
}

__global__ __launch_bounds__(256) void k_alloc_select(C1EncodeLaunch L) {
  const int64_t unit = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (unit >= L.frames * L.channels) return;
  const uint8_t *base = L.cand + unit * kCandBytes;
  const double *tot = reinterpret_cast<const double *>(base);
  double best = __builtin_huge_val();
  int best_c = 8;
#pragma unroll
  for (int c = 0; c < 8; c++) {            // ascending count, strict `<`: the smallest count wins ties (:116-129)
    const double t = tot[c];
    if (t < best) { best = t; best_c = c; }
  }
  uint64_t *dst = reinterpret_cast<uint64_t *>(L.alloc + unit * kAllocBytes);
  if (best_c < 8) {
    const uint64_t *src = reinterpret_cast<const uint64_t *>(base + 64 + best_c * 32);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    dst[3] = src[3] | ((uint64_t)best_c << 60);
  } else {
    dst[0] = 0; dst[1] = 0; dst[2] = 0;
    dst[3] = 1ull << 59;                   // fallback (:132-139): 20 BFUs, all indices zero
  }
}

// =====================================================================================================
// k_pack : quantize (quantization.js:34-56) + serializeFrame (serialization.js:41-98)
// =====================================================================================================
// ECMAScript ToInt32 of a double (what `| 0` does): truncate, wrap modulo 2^32.
__device__ __forceinline__ int32_t to_int32(double x) {
  const double t = trunc(x);
  if (fabs(t) < 2147483648.0) return (int32_t)t;
  const uint64_t bits = (uint64_t)__double_as_longlong(t);
  const int e = (int)((bits >> 52) & 0x7ff);
  if (e == 0x7ff) return 0;                                  // NaN, +-Infinity -> 0
  const int sh = e - 1075;                                   // value = mant * 2^sh, sh >= -21 here
  const uint64_t mant = (bits & 0xfffffffffffffull) | (1ull << 52);
  uint32_t low;
  if (sh >= 32) low = 0u;
  else if (sh >= 0) low = (uint32_t)(mant << sh);
  else low = (uint32_t)(mant >> (-sh));
  return (int32_t)((bits >> 63) ? (0u - low) : low);
}

__device__ __forceinline__ void put_bits_be(uint32_t *words, int pos, uint32_t v, int nbits) {
  // MSB-first (bitstream.js:15-40); words are big-endian 32-bit groups, assembled with LDS atomics
  const int w = pos >> 5, o = pos & 31;
  if (o + nbits <= 32) atomicOr(&words[w], v << (32 - o - nbits));
  else {
    const int lo = o + nbits - 32;
    atomicOr(&words[w], v >> lo);
    atomicOr(&words[w + 1], v << (32 - lo));
  }
}

// wave-level fence: LDS operations of one wave execute in issue order, so lanes of the same wave only
// need the compiler not to reorder across this point (no s_barrier: waves of a block run independently)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

constexpr int kPackWaves = 4;            // independent waves per workgroup, one sound unit at a time each
constexpr int kPackBlocks = 2048;        // persistent grid: waves stride over the units

struct PackLds {
  uint32_t words[56];         // the unit as big-endian 32-bit groups
  uint32_t desc[52];          // per BFU: bits(5) | mantissa bit offset(11) << 5 | first coefficient(9) << 16
  double normd[52];           // per BFU: quantRange / SCALE_FACTORS[sfi], 0 when nothing is coded
};

// One wave per sound unit.  A lane owns 8 consecutive coefficient slots (BFU-major order == bitstream
// order), quantizes them (quantization.js:34-56) and appends the mantissas MSB-first to a 64-bit
// accumulator (serialization.js:79-91).  Completed 32-bit groups that lie wholly inside the lane's bit
// range are plain LDS stores; only the first and last, which neighbours share, are atomic ORs.
// The loop is software pipelined: the allocation/side records of the unit two steps ahead and the
// coefficients of the next unit are in flight while the current unit is packed.
struct PackHeader {   // what a lane needs of one unit's allocation and side records
  uint32_t al_wl;     // allocation dword holding this lane's word-length nibble        (al[lane >> 3])
  uint32_t al7;       // last allocation dword: amount index, fallback flag
  uint32_t sd_sf;     // side dword holding this lane's scale-factor index              (side[lane >> 2])
  uint32_t sd_q;      // side dword lane & 15 (four scale factors for the 24-bit field, modes in dword 13)
  uint32_t al_a, al_b;   // allocation dwords (lane - 1) & 7 and lane & 7 (word-length bytes, lanes 0..7)
};
__device__ __forceinline__ PackHeader pack_load_header(const C1EncodeLaunch &L, int64_t unit, int lane) {
  const uint32_t *al = reinterpret_cast<const uint32_t *>(L.alloc + unit * kAllocBytes);
  const uint32_t *side = reinterpret_cast<const uint32_t *>(L.side + unit * kSideBytes);
  PackHeader h;
  h.al_wl = al[lane >> 3];
  h.al7 = al[7];
  h.sd_sf = side[lane >> 2];
  h.sd_q = side[lane & 15];
  h.al_a = al[(lane + 7) & 7];
  h.al_b = al[lane & 7];
  return h;
}

// ALL_LONG: the caller knows every unit of the batch has modes [0,0,0] (fixed block modes): coefficient order ==
// slot order, no per-slot position tables
template <bool ALL_LONG>
__global__ __launch_bounds__(C1_WAVE * kPackWaves, ALL_LONG ? 5 : 4) void k_pack(C1EncodeLaunch L) {
  __shared__ PackLds lds[kPackWaves];
  __shared__ double norm_s[64 * 16];        // quantRange / SCALE_FACTORS[sfi] (quantization.js:42-44)
  TablesPtr T = C1_TABLES(L.tables);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  PackLds &S = lds[wave];
  for (int i = threadIdx.x; i < 64 * 16; i += C1_WAVE * kPackWaves) norm_s[i] = T->norm[i];
  __syncthreads();
  // which BFU / which coefficient each of this lane's 8 slots is, and where it sits for long / short blocks
  int slot_b[8], slot_j[8], at_long[8], at_short[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int p = 8 * lane + m;
    slot_b[m] = bfu_of_slot(p);
    slot_j[m] = p - kBfuFirst[slot_b[m]];
    at_long[m] = ALL_LONG ? 0 : kStartLong[slot_b[m]] + slot_j[m];
    at_short[m] = ALL_LONG ? 0 : kStartShort[slot_b[m]] + slot_j[m];
  }
  const int my_size = lane < 52 ? kSpecs[lane] : 0;
  const int my_long = lane < 52 ? kStartLong[lane] : 0, my_short = lane < 52 ? kStartShort[lane] : 0;
  const int64_t units_total = L.frames * L.channels;
  const int64_t stride = (int64_t)gridDim.x * kPackWaves;
  const int64_t u_first = (int64_t)blockIdx.x * kPackWaves + wave;
  auto load_coefs = [&](int64_t unit, uint32_t modes_dword, float (&x)[8]) {
    const float *coefs = L.coefs + (unit << 9);
    const int modes = (int)(modes_dword & 0xff);
    if (ALL_LONG || modes == 0) {   // all long: coefficient order == slot order
      const float4 a = reinterpret_cast<const float4 *>(coefs)[2 * lane], c = reinterpret_cast<const float4 *>(coefs)[2 * lane + 1];
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = c.x; x[5] = c.y; x[6] = c.z; x[7] = c.w;
    } else {
      const int m0 = modes & 3, m1 = (modes >> 2) & 3, m2 = (modes >> 4) & 3;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int mode = slot_b[m] >= 36 ? m2 : (slot_b[m] >= 20 ? m1 : m0);
        x[m] = coefs[mode == 0 ? at_long[m] : at_short[m]];
      }
    }
  };
  if (u_first >= units_total) return;
  PackHeader h0 = pack_load_header(L, u_first, lane);
  PackHeader h1 = pack_load_header(L, u_first + stride < units_total ? u_first + stride : u_first, lane);
  float x[8];
  load_coefs(u_first, __shfl(h0.sd_q, 13), x);
  for (int64_t unit = u_first; unit < units_total; unit += stride) {
    // ---- issue the loads of the units ahead ----
    const int64_t u1 = unit + stride, u2 = unit + 2 * stride;
    PackHeader h2 = pack_load_header(L, u2 < units_total ? u2 : unit, lane);
    float xn[8];
    load_coefs(u1 < units_total ? u1 : unit, __shfl(h1.sd_q, 13), xn);
    // ---- this unit ----
    const uint32_t a7 = h0.al7;
    const bool fallback = (a7 >> 27) & 1;
    const int amount = (int)(a7 >> 28) & 7;
    const int n = bfu_amount(amount);
    const int modes = ALL_LONG ? 0 : (int)(__shfl(h0.sd_q, 13) & 0xff);
    const int m0 = modes & 3, m1 = (modes >> 2) & 3, m2 = (modes >> 4) & 3;
    if (lane < 56) S.words[lane] = 0;
    int wl = 0, sf = 0;
    if (lane < 52) {
      wl = lane < n ? (int)((h0.al_wl >> ((lane & 7) * 4)) & 15) : 0;
      sf = fallback ? 0 : (int)((h0.sd_sf >> ((lane & 3) * 8)) & 63);
    }
    // bit offset of every BFU's mantissas: exclusive prefix sum of bits*size over the wave
    const int bits_b = wl_bits(wl);
    const int mybits = bits_b * my_size;
    int scan = mybits;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(scan, d);
      if (lane >= d) scan += o;
    }
    if (lane < 52) {
      const int mode = lane >= 36 ? m2 : (lane >= 20 ? m1 : m0);
      S.desc[lane] = (uint32_t)bits_b | ((uint32_t)(16 + 10 * n + scan - mybits) << 5) | ((uint32_t)(mode == 0 ? my_long : my_short) << 16);
      S.normd[lane] = (sf != 0 && bits_b != 0) ? norm_s[sf * 16 + wl] : 0.0;
    }
    wave_sync();
    // header (serialization.js:46-53) and word-length indices (:55-64): the 4-bit indices are already
    // packed two per byte in the allocation record, low nibble first; the unit wants the high nibble first
    if (lane < 8) {
      auto wl_be = [](uint32_t v) -> uint32_t { return __builtin_bswap32(((v & 0x0F0F0F0Fu) << 4) | ((v >> 4) & 0x0F0F0F0Fu)); };
      const uint32_t header = ((uint32_t)(2 - m0) << 14) | ((uint32_t)(2 - m1) << 12) | ((uint32_t)(3 - m2) << 10) | ((uint32_t)amount << 5);
      const uint32_t prev = lane == 0 ? (header & 0xffffu) : wl_be(h0.al_a);      // word-length bytes 4(lane-1)..
      const uint32_t cur = lane == 7 ? 0u : wl_be(h0.al_b);                       // last dword carries flags, no indices
      atomicOr(&S.words[lane], (prev << 16) | (cur >> 16));
    }
    // scale-factor indices (:66-77): four 6-bit fields = 24 bits per lane
    if (lane < (n >> 2)) {
      const uint32_t q = fallback ? 0u : h0.sd_q;
      const uint32_t t = ((q & 63u) << 18) | (((q >> 8) & 63u) << 12) | (((q >> 16) & 63u) << 6) | ((q >> 24) & 63u);
      put_bits_be(S.words, 16 + 4 * n + 24 * lane, t, 24);
    }
    // mantissas
    uint64_t acc = 0;
    int cnt = -1, wi = 0;
    bool first = true;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const uint32_t dsc = S.desc[slot_b[m]];
      const int bits = dsc & 31;
      if (cnt < 0 && bits != 0) {                               // the lane's first coded slot fixes its bit cursor
        const int pos = (int)((dsc >> 5) & 0x7ff) + slot_j[m] * bits;
        cnt = pos & 31;                                         // phantom zero bits in front: a neighbour's bits
        wi = pos >> 5;
      }
      const double xs = (double)x[m] * S.normd[slot_b[m]];
      const double v = xs + (xs >= 0 ? 0.5 : -0.5);            // round half away from zero ...
      int32_t y = (int32_t)v;                                  // ... then `| 0`: truncation; exact wrap below
      if (__builtin_expect(!(fabs(v) < 2147483648.0), 0)) y = to_int32(v);
      const int32_t range = (1 << (bits > 0 ? bits - 1 : 0)) - 1;
      y = y > range ? range : (y < -range ? -range : y);
      acc = (acc << bits) | ((uint32_t)y & ((1u << bits) - 1u));
      cnt += bits != 0 ? bits : 0;
      if (cnt >= 32) {                                          // a 32-bit group is complete
        cnt -= 32;
        const uint32_t w = (uint32_t)(acc >> cnt);
        acc &= (1ull << cnt) - 1ull;
        if (first) atomicOr(&S.words[wi], w); else S.words[wi] = w;
        first = false;
        wi++;
      }
    }
    if (cnt > 0) atomicOr(&S.words[wi], (uint32_t)(acc << (32 - cnt)));
    wave_sync();
    if (lane < 53) reinterpret_cast<uint32_t *>(L.units + unit * C1_UNIT_BYTES)[lane] = __builtin_bswap32(S.words[lane]);
    wave_sync();
    h0 = h1; h1 = h2;
#pragma unroll
    for (int m = 0; m < 8; m++) x[m] = xn[m];
  }
}

// =====================================================================================================
// k_decode : deserializeFrame + decode() closure (decoder.js:408-411)
// =====================================================================================================
struct alignas(16) DecodeLds {
  double d1[46];        // stage-1 synthesis delay (qmfDelays.lowBand)
  double d2[46];        // stage-2 synthesis delay (qmfDelays.midBand)
  float dhi[39];        // high-band delay
  float tail[48];       // last 16 IMDCT samples per band (imdctOverlap tails, decoder.js:227-230)
  uint32_t words[56];   // the unit as big-endian words
  uint32_t desc[52];    // per BFU: bits(5) | sfi(6) << 5 | mantissa bit offset << 11 (may exceed the unit for arbitrary bytes)
  double sf_tab[64];    // SCALE_FACTORS and RN(1/range): lane-varying lookups, kept in LDS (a global load per
  double inv_tab[16];   // coefficient would cost a cache round trip each)
  union alignas(16) {
    float coef[512];    // dequantized coefficients: dead once the IMDCT pre-twiddle has read them
    float band[512];    // reconstructed bands: born at the overlap-add
  } cb;
  union alignas(16) {
    struct { union alignas(16) { float2 z[320]; } zz; alignas(16) float mid[512]; } m;   // IMDCT: points (4 pad per 16), outputs
    struct { alignas(16) double w2[454]; } q2;                   // stage-2 synthesis work buffer (padded 2 per 4)
    struct { alignas(16) double w1[698]; } q1;                   // stage-1 synthesis work buffer (padded 2 per 8), after w2 is consumed
  } u;
};

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
__device__ __forceinline__ IMixGeometry imix_geometry(int lane, const FrameModes &M) {
  IMixGeometry G;
  const int band = lane < 16 ? 0 : (lane < 32 ? 1 : 2);
  const int g = lane - (band == 0 ? 0 : (band == 1 ? 16 : 32));
  const bool lng = M.mode_of_band(band) == 0;
  const int nfft = lng ? (band == 2 ? 128 : 64) : 16, q4 = nfft / 4, n2 = 2 * nfft;
  const int r = lng ? bitrev(g, band == 2 ? 5 : 4) : bitrev(g & 3, 2);
  const int blk = lng ? 0 : (g >> 2);
  const int obase = (band == 0 ? 0 : (band == 1 ? 128 : 256)) + 32 * blk;       // coefficients in, samples out
  const int tab_base = lng ? (band == 2 ? (int)offsetof(C1DevTables, mdct_inv512) : (int)offsetof(C1DevTables, mdct_inv256))
                           : (int)offsetof(C1DevTables, mdct_inv64);
  const int tw_base = (int)offsetof(C1DevTables, fft_tw);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int jp = ((j & 1) << 1) | (j >> 1);
    const int i = r + q4 * jp;                              // position 4g+j holds point bitrev(4g+j)
    const int j0 = 2 * i, j1 = n2 - 1 - 2 * i;
    G.ja[j] = obase + (band > 0 ? n2 - 1 - j0 : j0);
    G.jb[j] = obase + (band > 0 ? n2 - 1 - j1 : j1);
    G.pre_tab[j] = tab_base + 16 * i;
  }
  const int pbase = band == 0 ? 0 : (band == 1 ? 64 : 128);
  G.za = zslot(pbase + 4 * g);
  G.zb = zslot(pbase + 16 * (g >> 2) + (g & 3));
  G.twb = tw_base + 16 * (3 + (g & 3));
  G.zc = zslot(pbase + 64 * (g >> 4) + (g & 15));
  G.twc = tw_base + 16 * (15 + (g & 15));
  G.zd = zslot(128 + (g & 31));
  G.twd = tw_base + 16 * (63 + (g & 31));
  G.is_long = lng;
  G.band2 = band == 2;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i = lng ? (band == 2 ? g + (j == 1 ? 64 : (j == 2 ? 32 : (j == 3 ? 96 : 0))) : g + 16 * j) : (g & 3) + 4 * j;
    const int idx = (i < nfft / 2) ? 2 * i : (2 * (i - nfft / 2) + nfft);
    G.post_tab[j] = tab_base + 16 * i;
    G.ox[j] = obase + n2 - 1 - idx;
    G.oy[j] = obase + idx;
  }
  return G;
}

// coef: 512 dequantized coefficients; z: 320 slots; mid: 512 outputs.  any_long / band2_long are wave-uniform.
__device__ __forceinline__ void imdct_r4(const float *coef, float2 *z, float *mid, const IMixGeometry &G, bool any_long,
                                         bool band2_long, TablesPtr T, TablesRsrc R) {
  float2 x[4];
  {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const double r = -(double)coef[G.ja[j]], mm = -(double)coef[G.jb[j]];
      const double2 t = table_pair(R, G.pre_tab[j]);
      x[j] = make_float2(f32(mm * t.y + r * t.x), f32(mm * t.x - r * t.y));
    }
    const double2 w0 = make_double2(T->fft_tw[0][0], T->fft_tw[0][1]);
    const double2 w1 = make_double2(T->fft_tw[1][0], T->fft_tw[1][1]);
    const double2 w2 = make_double2(T->fft_tw[2][0], T->fft_tw[2][1]);
    if (__all(r2_unit_ok(x[0], x[1]) && r2_unit_ok(x[2], x[3]))) { r2_butterfly_unit(x[0], x[1]); r2_butterfly_unit(x[2], x[3]); }
    else { r2_butterfly(x[0], x[1], w0); r2_butterfly(x[2], x[3], w0); }
    if (__all(r2_unit_ok(x[0], x[2]))) r2_butterfly_unit(x[0], x[2]);
    else r2_butterfly(x[0], x[2], w1);
    r2_butterfly(x[1], x[3], w2);
    float4 *dst = reinterpret_cast<float4 *>(z + G.za);
    dst[0] = make_float4(x[0].x, x[0].y, x[1].x, x[1].y);
    dst[1] = make_float4(x[2].x, x[2].y, x[3].x, x[3].y);
  }
  __syncthreads();
  {
    float2 *p = z + G.zb;
    const double2 wa = table_pair(R, G.twb), wb = table_pair(R, G.twb + 64), wc = table_pair(R, G.twb + 128);
    x[0] = p[0]; x[1] = p[4]; x[2] = p[8]; x[3] = p[12];
    r2_butterfly(x[0], x[1], wa); r2_butterfly(x[2], x[3], wa);
    r2_butterfly(x[0], x[2], wb); r2_butterfly(x[1], x[3], wc);
    if (G.is_long) { p[0] = x[0]; p[4] = x[1]; p[8] = x[2]; p[12] = x[3]; }
  }
  if (any_long) {
    __syncthreads();
    if (G.is_long) {
      float2 *p = z + G.zc;
      const double2 wa = table_pair(R, G.twc), wb = table_pair(R, G.twc + 256), wc = table_pair(R, G.twc + 512);
      x[0] = p[0]; x[1] = p[20]; x[2] = p[40]; x[3] = p[60];
      r2_butterfly(x[0], x[1], wa); r2_butterfly(x[2], x[3], wa);
      r2_butterfly(x[0], x[2], wb); r2_butterfly(x[1], x[3], wc);
      if (G.band2) { p[0] = x[0]; p[20] = x[1]; p[40] = x[2]; p[60] = x[3]; }
    }
    if (band2_long) {
      __syncthreads();
      if (G.band2) {
        const float2 *p = z + G.zd;
        const double2 wa = table_pair(R, G.twd), wb = table_pair(R, G.twd + 512);
        x[0] = p[0]; x[1] = p[80]; x[2] = p[40]; x[3] = p[120];
        r2_butterfly(x[0], x[1], wa); r2_butterfly(x[2], x[3], wb);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const double2 t = table_pair(R, G.post_tab[j]);
    const double rr = x[j].x, ii = x[j].y;
    mid[G.ox[j]] = f32(rr * t.x + ii * t.y);                // mdct.js:177-208
    mid[G.oy[j]] = f32(rr * t.y - ii * t.x);
  }
}

__global__ __launch_bounds__(C1_WAVE, 3) void k_decode(C1DecodeLaunch L) {
  __shared__ DecodeLds S;
  const int lane0 = threadIdx.x;
  int lane = lane0;
  const int ch = blockIdx.x % L.channels;
  const int64_t f0 = (int64_t)(blockIdx.x / L.channels) * kRunFramesDecode;
  float *__restrict__ pcm = L.pcm[ch];

  for (int i = lane; i < 46; i += 64) { S.d1[i] = 0.0; S.d2[i] = 0.0; }
  for (int i = lane; i < 39; i += 64) S.dhi[i] = 0.0f;
  for (int i = lane; i < 48; i += 64) S.tail[i] = 0.0f;
  if (lane < 3) S.words[53 + lane] = 0u;
  S.sf_tab[lane] = C1_TABLES(L.tables)->scale_factors[lane];
  if (lane < 16) S.inv_tab[lane] = C1_TABLES(L.tables)->inv_range[lane];
  // lane-only geometry, computed once per wave
  uint32_t slot[8];                                  // BFU(6) | index inside the BFU(5) << 6 | short-block position(9) << 11
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int p = lane0 + 64 * m;                    // coefficient slot in BFU-major order (== long-block position)
    const int b = bfu_of_slot(p), j = p - kBfuFirst[b];
    slot[m] = (uint32_t)b | ((uint32_t)j << 6) | ((uint32_t)(kStartShort[b] + j) << 11);
  }
  const int my_size = lane0 < 52 ? kSpecs[lane0] : 0;
  const IMixGeometry IGL = imix_geometry(lane0, FrameModes{0, 0, 0});   // all-long frames
  const TablesRsrc RT = tables_rsrc(L.tables);
  __syncthreads();

  const int64_t f_end = (f0 + kRunFramesDecode < L.frames) ? f0 + kRunFramesDecode : L.frames;
  for (int64_t f = f0 - 1; f < f_end; ++f) {
    if (f < -(int64_t)L.halo_units) continue;
    const bool emit = f >= f0;
    const int64_t unit = f * L.channels + ch;
    TablesPtr T = tables_for_this_frame(L.tables);
    lane = lane_for_this_frame(lane0);

    // ---------------- deserializeFrame (serialization.js:111-176) ----------------
    if (lane < 53) S.words[lane] = __builtin_bswap32(reinterpret_cast<const uint32_t *>(L.units + unit * C1_UNIT_BYTES)[lane]);
    {
      const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      reinterpret_cast<float4 *>(S.cb.coef)[lane] = zero4;
      reinterpret_cast<float4 *>(S.cb.coef)[64 + lane] = zero4;
    }
    __syncthreads();
    const uint32_t header = S.words[0] >> 16;
    const int m0 = 2 - (int)((header >> 14) & 3), m1 = 2 - (int)((header >> 12) & 3), m2 = 3 - (int)((header >> 10) & 3);
    const int n = bfu_amount((header >> 5) & 7);
    int wl = 0, sfi = 0;
    if (lane < n) {
      wl = (int)get_bits_be(S.words, 16 + 4 * lane, 4);
      sfi = (int)get_bits_be(S.words, 16 + 4 * n + 6 * lane, 6);
    }
    const int mybits = wl_bits(wl) * my_size;
    int scan = mybits;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(scan, d);
      if (lane >= d) scan += o;
    }
    if (lane < 52) S.desc[lane] = (uint32_t)wl_bits(wl) | ((uint32_t)sfi << 5) | ((uint32_t)(16 + 10 * n + scan - mybits) << 11);
    __syncthreads();
    // ---------------- dequantizationStage (decoder.js:52-98) ----------------
    const bool all_long = (m0 | m1 | m2) == 0;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int sb = slot[m] & 63, sj = (slot[m] >> 6) & 31;
      const uint32_t dsc = S.desc[sb];
      const int bits = dsc & 31;
      if (bits == 0) continue;                                  // BFU not coded (or beyond nBfu: its word length reads 0)
      const int sf = (dsc >> 5) & 63;
      const uint32_t raw = get_bits_be(S.words, (int)(dsc >> 11) + sj * bits, bits);
      const int32_t q = raw >= (1u << (bits - 1)) ? (int32_t)raw - (1 << bits) : (int32_t)raw;     // bitstream.js:78-82
      const int32_t range = (1 << (bits - 1)) - 1;
      float v = 0.0f;                                                                               // quantization.js:65-78
      if (sf != 0) {
        const double a = (double)q * S.sf_tab[sf];
        if (T->dq_fast) {
          const double y = S.inv_tab[bits - 1], q0 = a * y;
          v = f32(__builtin_fma(__builtin_fma(-q0, (double)range, a), y, q0));                  // == a / range (checked on the host)
        } else {
          v = f32(a / (double)range);
        }
      }
      const int mode = sb >= 36 ? m2 : (sb >= 20 ? m1 : m0);
      S.cb.coef[mode == 0 ? lane + 64 * m : (int)(slot[m] >> 11)] = v;
    }
    __syncthreads();

    // ---------------- imdctStage (decoder.js:116-330) ----------------
    float *mid = S.u.m.mid;
    if (all_long) {
      imdct_r4(S.cb.coef, S.u.m.zz.z, mid, IGL, true, true, T, RT);
      __syncthreads();
      // overlap-add of the first 32 samples of every band (mdct.js:230-245 via decoder.js:203-232) ...
      if (lane < 32) {
        const bool lo = lane < 16;
        const int i = lo ? lane : 31 - lane;
        const double wa = T->window[i], wb = T->window[31 - i];       // w1 = W[i], w2 = W[31-i]
#pragma unroll
        for (int b = 0; b < 3; b++) {
          const int off = b == 0 ? 0 : (b == 1 ? 128 : 256);
          const double pv = S.tail[16 * b + i], cv = mid[off + 15 - i];
          S.cb.band[off + lane] = lo ? f32(pv * wb - cv * wa) : f32(pv * wa + cv * wb);
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
    const IMixGeometry IG = imix_geometry(lane, M);
    imdct_r4(S.cb.coef, S.u.m.zz.z, mid, IG, m0 == 0 || m1 == 0 || m2 == 0, m2 == 0, T, RT);
    __syncthreads();
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
          const double w1 = T->window[k], w2 = T->window[31 - k];
          v = f32((double)prev[k] * w2 - (double)curr[15 - k] * w1);
        } else {
          const int i = 31 - k;
          const double w1 = T->window[i], w2 = T->window[31 - i];
          v = f32((double)prev[i] * w1 + (double)curr[15 - i] * w2);
        }
      } else {
        v = mid[off + k - 16];                     // long block only: invBuf[16 .. S-16)
      }
      S.cb.band[g] = v;
    }
    }
    __syncthreads();
    if (lane < 48) {
      const int b = lane >> 4, k = lane & 15;
      const int off = b == 0 ? 0 : (b == 1 ? 128 : 256), Sb = b == 2 ? 256 : 128;
      S.tail[lane] = mid[off + Sb - 16 + k];
    }
    __syncthreads();

    // ---------------- qmfSynthesisStage (decoder.js:349-389) ----------------
    double *w2 = S.u.q2.w2, *w1 = S.u.q1.w1;
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
        const double l = S.cb.band[i], h = S.cb.band[128 + i];
        *reinterpret_cast<double2 *>(&w2[pidx<2>(46 + 2 * i)]) = make_double2((double)f32(0.5 * (l + h)), (double)f32(0.5 * (l - h)));
      }
      __syncthreads();
      if (lane < 39) S.dhi[lane] = keep;
    }
    {
      double s0[2], s1[2];
      qmf_synthesis_core<2, 2>(w2, lane, T, s0, s1);
      if (lane < 46) S.d2[lane] = w2[pidx<2>(256 + lane)];
      __syncthreads();                                    // w1 reuses the memory of w2 from here on
      if (lane < 46) w1[pidx<3>(lane)] = S.d1[lane];
      // stage 1 input: (stage-2 output, delayed high); stage-2 output pair of i: out[2i] = s1, out[2i+1] = s0
#pragma unroll
      for (int d = 0; d < 2; d++) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const int sidx = 4 * lane + 2 * d + t;          // sample index in the 256-sample low band
          const double l = (double)f32(t == 0 ? s1[d] : s0[d]);
          const double h = hi4[2 * d + t];
          *reinterpret_cast<double2 *>(&w1[pidx<3>(46 + 2 * sidx)]) = make_double2((double)f32(0.5 * (l + h)), (double)f32(0.5 * (l - h)));
        }
      }
    }
    __syncthreads();
    {
      double s0[4], s1[4];
      qmf_synthesis_core<4, 3>(w1, lane, T, s0, s1);
      if (lane < 46) S.d1[lane] = w1[pidx<3>(512 + lane)];
      if (emit) {
        float4 *dst = reinterpret_cast<float4 *>(pcm + f * 512 + 8 * lane);
        dst[0] = make_float4(f32(s1[0]), f32(s0[0]), f32(s1[1]), f32(s0[1]));
        dst[1] = make_float4(f32(s1[2]), f32(s0[2]), f32(s1[3]), f32(s0[3]));
      }
    }
    __syncthreads();
  }
}

// =====================================================================================================
// synthetic signals
// =====================================================================================================
__device__ __forceinline__ double xorshift_u(uint32_t &s) {
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  return ((double)s / 4294967296.0) * 2.0 - 1.0;
}
// one thread per frame; frame_states[f] = PRNG state before the frame's first sample
__global__ void k_generate_white(const uint32_t *frame_states, int64_t frames, float *pcm) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= frames) return;
  uint32_t s = frame_states[f];
  float4 *dst = reinterpret_cast<float4 *>(pcm + f * 512);
  for (int i = 0; i < 128; i++) {
    float4 v;
    v.x = f32(xorshift_u(s) * 0.5); v.y = f32(xorshift_u(s) * 0.5);
    v.z = f32(xorshift_u(s) * 0.5); v.w = f32(xorshift_u(s) * 0.5);
    dst[i] = v;
  }
}
// one thread per 512-frame segment: p = 0.98p + 0.05u, plus 0.8u' in the second half of frames 5 mod 8
__global__ void k_generate_pink(const uint32_t *segment_states, int64_t frames, float *pcm) {
  const int64_t seg = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (seg * 512 >= frames) return;
  uint32_t s = segment_states[seg];
  const int64_t fend = (seg + 1) * 512 < frames ? (seg + 1) * 512 : frames;
  double p = 0.0;
  for (int64_t f = seg * 512; f < fend; f++) {
    float *dst = pcm + f * 512;
    const bool burst = (f & 7) == 5;
    for (int i = 0; i < 512; i++) {
      const double u = xorshift_u(s);
      p = 0.98 * p + 0.05 * u;
      double v = p;
      if (burst && i >= 256) v += 0.8 * xorshift_u(s);
      dst[i] = f32(v);
    }
  }
}

// =====================================================================================================
// PCM format conversion either side of the path: pure streaming, HBM bound
// =====================================================================================================
// bin/cli.js:394-404: value / 2^(bits-1); the Float32Array store rounds (only 32-bit input can round)
template <int BITS, int CH>
__global__ void k_pcm_from_int(const uint8_t *__restrict__ src, int64_t n, float *__restrict__ out0, float *__restrict__ out1) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const uint8_t *p = src + (i * CH + c) * (BITS / 8);
      float v;
      if (BITS == 16) v = f32((double)(int16_t)(p[0] | (p[1] << 8)) / 32768.0);
      else if (BITS == 24) {
        int32_t s = p[0] | (p[1] << 8) | (p[2] << 16);
        if (s > 0x7fffff) s -= 0x1000000;
        v = f32((double)s / 8388608.0);
      } else v = f32((double)(int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)) / 2147483648.0);
      (c == 0 ? out0 : out1)[i] = v;
    }
  }
}
// four frames per thread: CH*BITS/8 dword loads, one float4 store per channel (needs 4-byte aligned input, 16-byte
// aligned outputs; the launcher falls back to the scalar kernel otherwise and for the tail)
template <int BITS, int CH>
__global__ void k_pcm_from_int_x4(const uint32_t *__restrict__ src, int64_t quads, float *__restrict__ out0, float *__restrict__ out1) {
  constexpr int kBps = BITS / 8, kWords = CH * kBps;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
    uint32_t w[kWords];
#pragma unroll
    for (int k = 0; k < kWords; k++) w[k] = __builtin_nontemporal_load(src + q * kWords + k);
    auto byte_at = [&](int b) -> uint32_t { return (w[b >> 2] >> ((b & 3) * 8)) & 0xffu; };
#pragma unroll
    for (int c = 0; c < CH; c++) {
      float v[4];
#pragma unroll
      for (int f = 0; f < 4; f++) {
        const int b = (f * CH + c) * kBps;
        if (BITS == 16) v[f] = f32((double)(int16_t)(uint16_t)(byte_at(b) | (byte_at(b + 1) << 8)) / 32768.0);
        else if (BITS == 24) {
          const int32_t s = (int32_t)((byte_at(b) | (byte_at(b + 1) << 8) | (byte_at(b + 2) << 16)) << 8) >> 8;
          v[f] = f32((double)s / 8388608.0);
        } else v[f] = f32((double)(int32_t)w[b >> 2] / 2147483648.0);
      }
      reinterpret_cast<float4 *>(c == 0 ? out0 : out1)[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}
// codec/io/processor.js:381-394: Math.max(-1, Math.min(1, x)); negative * 0x8000, else * 0x7fff; setInt16
__device__ __forceinline__ int16_t pcm_to_i16(float x) {
  double s = (double)x;
  s = s < 1.0 ? s : 1.0;                     // Math.min(1, x): NaN stays NaN
  s = s > -1.0 ? s : -1.0;                   // Math.max(-1, .)
  if (x != x) return 0;                      // NaN -> setInt16 stores 0
  const double v = s < 0 ? s * 32768.0 : s * 32767.0;
  return (int16_t)(int32_t)v;                // ToInt16 of an in-range value: truncation toward zero
}
template <int CH>
__global__ void k_pcm_to_int16(const float *__restrict__ in0, const float *__restrict__ in1, int64_t n, int16_t *__restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (CH == 1) dst[i] = pcm_to_i16(in0[i]);
    else {
      const uint32_t l = (uint16_t)pcm_to_i16(in0[i]), r = (uint16_t)pcm_to_i16(in1[i]);
      reinterpret_cast<uint32_t *>(dst)[i] = l | (r << 16);
    }
  }
}

}  // namespace

void c1k_launch_pcm_from_int(const void *src, int bits, int channels, int64_t n, float *const *pcm, hipStream_t stream) {
  const uint8_t *s = static_cast<const uint8_t *>(src);
  float *o0 = pcm[0], *o1 = channels > 1 ? pcm[1] : nullptr;
  const dim3 block(256);
  int64_t done = 0;
  const bool aligned = ((uintptr_t)src & 3) == 0 && ((uintptr_t)o0 & 15) == 0 && ((uintptr_t)o1 & 15) == 0;
  if (aligned && n >= 4) {
    const int64_t quads = n / 4;
    const dim3 grid((unsigned)std::min<int64_t>((quads + 255) / 256, 256 * 32));
    const uint32_t *w = static_cast<const uint32_t *>(src);
#define C1_LAUNCH_X4(B, C) hipLaunchKernelGGL((k_pcm_from_int_x4<B, C>), grid, block, 0, stream, w, quads, o0, o1)
    if (channels == 1) { if (bits == 16) C1_LAUNCH_X4(16, 1); else if (bits == 24) C1_LAUNCH_X4(24, 1); else C1_LAUNCH_X4(32, 1); }
    else { if (bits == 16) C1_LAUNCH_X4(16, 2); else if (bits == 24) C1_LAUNCH_X4(24, 2); else C1_LAUNCH_X4(32, 2); }
#undef C1_LAUNCH_X4
    done = quads * 4;
  }
  if (done == n) return;
  const int64_t rest = n - done;
  s += done * channels * (bits / 8);
  o0 += done;
  if (o1) o1 += done;
  const dim3 grid((unsigned)std::min<int64_t>((rest + 255) / 256, 256 * 32));
#define C1_LAUNCH_FROM(B, C) hipLaunchKernelGGL((k_pcm_from_int<B, C>), grid, block, 0, stream, s, rest, o0, o1)
  if (channels == 1) { if (bits == 16) C1_LAUNCH_FROM(16, 1); else if (bits == 24) C1_LAUNCH_FROM(24, 1); else C1_LAUNCH_FROM(32, 1); }
  else { if (bits == 16) C1_LAUNCH_FROM(16, 2); else if (bits == 24) C1_LAUNCH_FROM(24, 2); else C1_LAUNCH_FROM(32, 2); }
#undef C1_LAUNCH_FROM
}
void c1k_launch_pcm_to_int16(const float *const *pcm, int channels, int64_t n, int16_t *dst, hipStream_t stream) {
  const dim3 grid((unsigned)std::min<int64_t>((n + 255) / 256, 256 * 32)), block(256);
  if (channels == 1) hipLaunchKernelGGL((k_pcm_to_int16<1>), grid, block, 0, stream, pcm[0], (const float *)nullptr, n, dst);
  else hipLaunchKernelGGL((k_pcm_to_int16<2>), grid, block, 0, stream, pcm[0], pcm[1], n, dst);
}

// ---- launchers -----------------------------------------------------------------------------------------
void c1k_launch_analysis(const C1EncodeLaunch &L, bool detect, hipStream_t stream) {
  const int64_t runs = (L.frames + kRunFramesLong - 1) / kRunFramesLong;
  const dim3 grid((unsigned)(runs * L.channels)), block(C1_WAVE);
  (void)detect;
  hipLaunchKernelGGL((k_analysis_fast<false>), grid, block, 0, stream, L);
}
void c1k_launch_detect(const C1EncodeLaunch &L, float *bands_ws, double *feat_ws, uint8_t *modes_ws, uint32_t *lists_ws,
                       hipStream_t stream) {
  const int64_t runs = (L.frames + kRunFramesLong - 1) / kRunFramesLong, units = L.frames * L.channels;
  (void)hipMemsetAsync(lists_ws, 0, 4 * sizeof(uint32_t), stream);
  hipLaunchKernelGGL(k_detect_features, dim3((unsigned)(runs * L.channels)), dim3(C1_WAVE), 0, stream, L, bands_ws, feat_ws);
  hipLaunchKernelGGL(k_detect_decide, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, stream, feat_ws, L.channels, L.frames,
                     L.halo_frames, L.tables, L.opts, modes_ws, lists_ws);
  // both list kernels size their grids for the whole batch and stop at the device-side count
  const dim3 grid((unsigned)std::min<int64_t>(units, 256 * 48)), block(C1_WAVE);
  hipLaunchKernelGGL((k_mdct_bands<true>), grid, block, 0, stream, L, bands_ws, modes_ws, lists_ws);
  hipLaunchKernelGGL((k_mdct_bands<false>), grid, block, 0, stream, L, bands_ws, modes_ws, lists_ws);
}
void c1k_launch_analysis_long(const C1EncodeLaunch &L, hipStream_t stream) {
  const int64_t runs = (L.frames + kRunFramesLong - 1) / kRunFramesLong;
  hipLaunchKernelGGL((k_analysis_fast<true>), dim3((unsigned)(runs * L.channels)), dim3(C1_WAVE), 0, stream, L);
}
void c1k_launch_allocate(const C1EncodeLaunch &L, hipStream_t stream) {
  const int64_t units = L.frames * L.channels;
  (void)hipMemsetAsync(L.work_count, 0, sizeof(uint32_t), stream);
  hipLaunchKernelGGL(k_alloc_first, dim3((unsigned)((units + 63) / 64)), dim3(C1_WAVE), 0, stream, L);
  const int64_t rest_blocks = std::min<int64_t>((units * 7 + 63) / 64, 256 * 10);
  hipLaunchKernelGGL(k_alloc_rest, dim3((unsigned)rest_blocks), dim3(C1_WAVE), 0, stream, L);
  hipLaunchKernelGGL(k_alloc_select, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, stream, L);
}
void c1k_launch_pack(const C1EncodeLaunch &L, bool all_long, hipStream_t stream) {
  const dim3 grid((unsigned)std::min<int64_t>(kPackBlocks, (L.frames * L.channels + kPackWaves - 1) / kPackWaves)), block(C1_WAVE * kPackWaves);
  if (all_long) hipLaunchKernelGGL((k_pack<true>), grid, block, 0, stream, L);
  else hipLaunchKernelGGL((k_pack<false>), grid, block, 0, stream, L);
}
void c1k_launch_decode(const C1DecodeLaunch &L, hipStream_t stream) {
  const int64_t runs = (L.frames + kRunFramesDecode - 1) / kRunFramesDecode;
  hipLaunchKernelGGL(k_decode, dim3((unsigned)(runs * L.channels)), dim3(C1_WAVE), 0, stream, L);
}
void c1k_launch_generate_white(const uint32_t *frame_states, int64_t frames, float *pcm, hipStream_t stream) {
  hipLaunchKernelGGL(k_generate_white, dim3((unsigned)((frames + 63) / 64)), dim3(64), 0, stream, frame_states, frames, pcm);
}
void c1k_launch_generate_pink(const uint32_t *segment_states, int64_t frames, float *pcm, hipStream_t stream) {
  const int64_t segs = (frames + 511) / 512;
  hipLaunchKernelGGL(k_generate_pink, dim3((unsigned)((segs + 63) / 64)), dim3(64), 0, stream, segment_states, frames, pcm);
}
static_assert(sizeof(LongLds) <= 8192, "all-long analysis: 20 waves per CU need <= 8 KiB of LDS per wave");
