// c1_device.h -- device-side code shared by the kernel translation units of libcarta1_hip.so: the numeric model's
// helpers, format tables, the QMF cores, scale-factor indices and the radix-4 MDCT cores.  Everything here is
// __device__ __forceinline__ or lives in an unnamed namespace, so every .hip file gets its own copy.
//
// Numeric model (the reference is ECMAScript): every operation is an IEEE-754
// double operation, no a*b+c fusion, and every Float32Array store rounds to binary32.  This file
// is compiled with -ffp-contract=off; the ONLY fused operations are the QMF convolution terms,
// where both factors are binary32 values so the double product is exact and fma(a,b,acc) equals
// round(a*b)+acc bit for bit.  No MFMA: nothing here is a dense contraction.
//
// Kernel files (one wavefront == one 64-thread workgroup in the frame-walking kernels, so __syncthreads() is a
// wave-local fence):
//   c1_k_analysis.hip  k_analysis_fast<ALL_LONG>: fixed block modes; one wave walks 64 consecutive frames of one
//                      channel carrying QMF delay lines and MDCT overlap like the reference's BufferPool
//   c1_k_detect.hip    k_detect_features -> k_detect_decide -> k_mdct_bands<LONG>: transient detection
//   c1_k_allocate.hip  k_alloc_first / k_alloc_rest / k_alloc_select: the greedy RDO heaps, one lane per heap
//   c1_k_pack.hip      k_pack<ALL_LONG>: quantize + MSB-first packing, one wave per sound unit
//   c1_k_decode.hip    k_decode: unpack, dequantize, IMDCT + overlap-add, QMF synthesis
//   c1_k_formats.hip   synthetic input, WAV sample formats
#pragma once
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
// The frame-walking kernels run ONE wave per workgroup, and the LDS operations of one wave execute in issue order: all
// they need between a phase that writes LDS and the next that reads it is that the COMPILER keeps the order.  That is a
// fence at wavefront scope.  __syncthreads() is a fence at workgroup scope plus a barrier, and the workgroup-scope
// release makes the compiler wait for every outstanding global access first (s_waitcnt vmcnt(0)): the PCM prefetched
// for the next frame, the table values requested a round ahead and the coefficient stores of the frame just finished
// were all waited for at the next fence instead of when (if ever) their registers were needed.
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
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
typedef float v4f __attribute__((ext_vector_type(4)));    // for __builtin_nontemporal_store of 16 bytes

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

// inclusive prefix sum over the 64 lanes of a wave with data-parallel primitives (row shifts inside the rows of 16
// lanes, then the row totals broadcast from lanes 15 and 31): 6 DPP adds, no LDS crossbar round trips
__device__ __forceinline__ int wave_inclusive_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return x;
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
// mode_of_band: block mode of bands 0..2 (0 = long); BFU b starts at kStartLong[b] or kStartShort[b] accordingly
__device__ __forceinline__ SfLong sf_geometry(int lane, int m0, int m1, int m2) {
  SfLong g;
  g.wide = lane >= 44;
  g.b = g.wide ? 44 + ((lane - 44) >> 1) : lane;
  const int half = g.wide ? (lane & 1) : 0;
  g.cnt = lane < 60 ? (g.wide ? 10 : (int)kSpecs[lane < 44 ? lane : 0]) : 1;
  const int bb = lane < 60 ? g.b : 0;
  const int mode = bb >= 36 ? m2 : (bb >= 20 ? m1 : m0);
  g.src = (mode == 0 ? kStartLong[bb] : kStartShort[bb]) + 10 * half;
  g.store = lane < 60 && (!g.wide || half == 0);
  return g;
}
__device__ __forceinline__ SfLong sf_long_geometry(int lane) { return sf_geometry(lane, 0, 0, 0); }
__device__ __forceinline__ void sf_long(const float *coef, uint8_t *sfi_out, const SfLong &g, TablesPtr T) {
  const float *src = coef + g.src;
  float mx = 0.0f;
#pragma unroll
  for (int j = 0; j < 12; j++) mx = fmaxf(mx, fabsf(src[j < g.cnt ? j : g.cnt - 1]));
  // the neighbour lane's maximum: a DPP quad permutation [1,0,3,2], not an LDS-crossbar shuffle
  const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0xB1, 0xf, 0xf, false));
  mx = fmaxf(mx, g.wide ? other : 0.0f);
  const int sfi = T->sf_fast ? scale_factor_index_fast(mx, T->sf_m1, T->sf_m2) : scale_factor_index(mx, T);
  if (g.store) sfi_out[g.b] = (uint8_t)sfi;
}

// ---- the same scan of an all-long unit from three 16-byte reads (c1_k_spec.hip) ---------------------------------
// A lane's <= 12 coefficients (sf_long_geometry: BFU b = lane below 44, half a 20-coefficient BFU above) lie in the three
// aligned groups of four that start at group src >> 2.  Which of those twelve elements belong to the lane's BFU depends
// on the lane alone, so "element k is inside" is a set of lanes known at compile time: it is used as the EXEC mask of
// one v_max_f32.  Against twelve 4-byte reads at lane-varying strides (4- to 8-way bank conflicts, an address select
// each): 3 LDS instructions instead of 12, 12 vector instructions instead of ~36, 24 scalar moves.
constexpr int kSpecsC[52] = {8, 8, 8, 8, 4,  4,  4,  4,  8,  8,  8,  8,  6,  6,  6,  6,  6,  6,  6,  6,  6,  6,  6,  6,  7,  7,
                             7, 7, 9, 9,  9,  9,  10, 10, 10, 10, 12, 12, 12, 12, 12, 12, 12, 12, 20, 20, 20, 20, 20, 20, 20, 20};
constexpr uint64_t sf_lane_mask(int k) {
  uint64_t m = 0;
  for (int lane = 0; lane < 60; lane++) {
    const bool wide = lane >= 44;
    const int b = wide ? 44 + ((lane - 44) >> 1) : lane;
    int start = 0;
    for (int i = 0; i < b; i++) start += kSpecsC[i];          // BFU_START_LONG = prefix sums of the sizes (constants.js:38-44)
    const int src = start + (wide ? 10 * (lane & 1) : 0), cnt = wide ? 10 : kSpecsC[b];
    const int idx = 4 * (src >> 2) + k;
    if (idx >= src && idx < src + cnt) m |= 1ull << lane;
  }
  return m;
}
#define C1_SF_STEP(K) "s_mov_b32 exec_lo, %[l" #K "]\n\ts_mov_b32 exec_hi, %[h" #K "]\n\tv_max_f32_e64 %[a], %[a], |%[x" #K "]|\n\t"
#define C1_SF_MASK(K) [l##K] "n"((uint32_t)sf_lane_mask(K)), [h##K] "n"((uint32_t)(sf_lane_mask(K) >> 32))
__device__ __forceinline__ float sf_scan_long_groups(float4 q0, float4 q1, float4 q2) {
  float mx = 0.0f;
  uint64_t saved;
  asm volatile("s_mov_b64 %[sv], exec\n\t"
               C1_SF_STEP(0) C1_SF_STEP(1) C1_SF_STEP(2) C1_SF_STEP(3) C1_SF_STEP(4) C1_SF_STEP(5)
               C1_SF_STEP(6) C1_SF_STEP(7) C1_SF_STEP(8) C1_SF_STEP(9) C1_SF_STEP(10) C1_SF_STEP(11)
               "s_mov_b64 exec, %[sv]"
               : [a] "+v"(mx), [sv] "=&s"(saved)
               : [x0] "v"(q0.x), [x1] "v"(q0.y), [x2] "v"(q0.z), [x3] "v"(q0.w), [x4] "v"(q1.x), [x5] "v"(q1.y), [x6] "v"(q1.z),
                 [x7] "v"(q1.w), [x8] "v"(q2.x), [x9] "v"(q2.y), [x10] "v"(q2.z), [x11] "v"(q2.w),
                 C1_SF_MASK(0), C1_SF_MASK(1), C1_SF_MASK(2), C1_SF_MASK(3), C1_SF_MASK(4), C1_SF_MASK(5),
                 C1_SF_MASK(6), C1_SF_MASK(7), C1_SF_MASK(8), C1_SF_MASK(9), C1_SF_MASK(10), C1_SF_MASK(11));
  return mx;
}
static_assert(sf_lane_mask(0) != 0 && (sf_lane_mask(0) >> 60) == 0, "lanes 60..63 own no BFU");

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
// Where the exact long-block core (mdct_long_r4) reads its 16 lane-varying (cos, sin) pairs from: the tables in global memory
// through the buffer resource, or a workgroup's copy in LDS of the two pieces it touches -- mdct_fwd256 | mdct_fwd512 (3 072
// bytes) and the first 128 pairs of fft_tw (2 048 bytes), one behind the other.  A lane-varying global read is a cache round
// trip on the counter the unit's loads and stores share; a kernel with nothing to hide it behind (k_mdct_bands: three such
// round trips in the chain of every unit) reads the copy instead.  r4_geometry_lds() moves a geometry's byte offsets over.
struct BufTab {
  static constexpr int origin = 0;                         // byte offset of C1DevTables that offset 0 of this reader stands for
  TablesRsrc R;
  __device__ __forceinline__ double2 pair(int byte_offset) const { return table_pair(R, byte_offset); }
};
struct LdsTab {
  static constexpr int origin = (int)offsetof(C1DevTables, mdct_fwd256);   // (of the forward tables; the twiddles sit behind them)
  const char *base;
  __device__ __forceinline__ double2 pair(int byte_offset) const { return *reinterpret_cast<const double2 *>(base + byte_offset); }
};
constexpr int kLdsTabFwdBytes = (128 + 256) * (int)sizeof(double), kLdsTabTwBytes = 128 * 2 * (int)sizeof(double);
constexpr int kLdsTabBytes = kLdsTabFwdBytes + kLdsTabTwBytes;
static_assert(offsetof(C1DevTables, mdct_fwd512) == offsetof(C1DevTables, mdct_fwd256) + 128 * sizeof(double), "the two forward tables are copied as one piece");
// byte k of the copy <- byte of C1DevTables (16-byte granules)
__device__ __forceinline__ int lds_tab_source(int k) {
  return k < kLdsTabFwdBytes ? (int)offsetof(C1DevTables, mdct_fwd256) + k : (int)offsetof(C1DevTables, fft_tw) + (k - kLdsTabFwdBytes);
}

// =====================================================================================================
// binary32 helpers of the speculative kernels (c1_k_spec.hip, the speculative transient detector in c1_k_detect.hip)
// =====================================================================================================
template <int CTRL>
__device__ __forceinline__ float dpp_read(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
// every lane of a 16-lane row ends with the row's sum ((q0 + q1) + (q2 + q3), q = ((v0 + v1) + (v2 + v3)) of a quad)
__device__ __forceinline__ float row_allreduce(float x) {
  x += dpp_read<0xB1>(x);    // quad_perm [1,0,3,2]
  x += dpp_read<0x4E>(x);    // quad_perm [2,3,0,1]
  x += dpp_read<0x141>(x);   // row_half_mirror
  x += dpp_read<0x140>(x);   // row_mirror
  return x;
}
__device__ __forceinline__ float lane_value(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
__device__ __forceinline__ float wave_sum(float x) {
  x = row_allreduce(x);
  return (lane_value(x, 0) + lane_value(x, 16)) + (lane_value(x, 32) + lane_value(x, 48));
}

// ---- packed binary32 arithmetic -------------------------------------------------------------------------------------
// On CDNA a wave64 vector instruction occupies its SIMD for four cycles whatever its type; the binary32 peak needs the
// packed forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two independent IEEE operations per lane and instruction
// on an aligned register pair).  The kernel is written on 2-vectors so that every pair it operates on is one the data
// already forms: the (even, odd) sample pairs of a 16-byte LDS read, (re, im) of a complex point.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f V2(float x, float y) { v2f r; r.x = x; r.y = y; return r; }

// Multiplying by +-1 is exact, so "negate one half" and "swap and negate" ride on an FMA or a product with one of
// these constant pairs instead of costing sign-bit instructions of their own.
#define PMN V2(1.0f, -1.0f)
#define PNM V2(-1.0f, 1.0f)
// complex product (x.x + i x.y)(w.x + i w.y): two products rounded, then two fused
__device__ __forceinline__ v2f cmul32(v2f x, v2f w) {
  const v2f wr = w.yx * PNM;                             // (-w.y, w.x), exact
  const v2f t = x.yy * wr;                               // (-(x.y w.y), x.y w.x)
  return pk_fma(x.xx, w, t);
}

__device__ __forceinline__ v2f table_f2(TablesRsrc R, int byte_offset) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b64(R, byte_offset, 0, 0);
  v2f d;
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
template <class Tab>
__device__ __forceinline__ R4Early r4_early_t(const R4Geometry &G, const Tab R) {
  R4Early e;
  e.t0 = R.pair(G.pre_tab[0]); e.t1 = R.pair(G.pre_tab[1]);
  e.t2 = R.pair(G.pre_tab[2]); e.t3 = R.pair(G.pre_tab[3]);
  return e;
}
__device__ __forceinline__ R4Early r4_early(const R4Geometry &G, TablesRsrc R) { return r4_early_t(G, BufTab{R}); }
// the geometry's table offsets as offsets into the LDS copy (LdsTab)
__device__ __forceinline__ R4Geometry r4_geometry_lds(R4Geometry G) {
  constexpr int fwd = (int)offsetof(C1DevTables, mdct_fwd256), tw = (int)offsetof(C1DevTables, fft_tw) - kLdsTabFwdBytes;
#pragma unroll
  for (int j = 0; j < 4; j++) { G.pre_tab[j] -= fwd; G.post_tab[j] -= fwd; }
  G.twb -= tw; G.twc -= tw; G.twd -= tw;
  return G;
}
// in: 1024 floats (in0 | in1 | in2, zero padded long-block inputs); z: 320 slots; coef: 512 floats (may share
// memory with `in`: the inputs are dead once round A has read them)
// late: when not null, the lane's values of the END of the transform read back from LDS instead of being carried in registers
// through the caller's frame loop: late[64 j] = cx[j] | cy[j] << 9 | (post-twiddle pair index of point j) << 18 (j = 0..3; the
// pointer is the lane's; r4_late_word).  In k_analysis_fast they pushed a dozen values to scratch, and a scratch reload is a
// vector-memory load: it waits on the counter the PCM prefetch and the frame's stores share.
__device__ __forceinline__ uint32_t r4_late_word(const R4Geometry &G, int j) {
  const int tab_base = G.band2 ? (int)offsetof(C1DevTables, mdct_fwd512) : (int)offsetof(C1DevTables, mdct_fwd256);
  return (uint32_t)G.cx[j] | ((uint32_t)G.cy[j] << 9) | ((uint32_t)((G.post_tab[j] - tab_base) >> 4) << 18);
}
template <class Tab>
__device__ __forceinline__ void mdct_long_r4_t(const float *in, float2 *z, float *coef, const R4Geometry &G, TablesPtr T, const Tab R, const R4Early &E,
                                               const uint32_t *late = nullptr) {
  float2 x[4];
  // the lane-varying table values of the frame are requested up front: the loads are in flight while round A
  // reads its inputs, instead of one cache round trip in front of every round
  const double2 t0 = E.t0, t1 = E.t1, t2 = E.t2, t3 = E.t3;
  const double2 wBa = R.pair(G.twb), wBb = R.pair(G.twb + 64), wBc = R.pair(G.twb + 128);
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
  const double2 wCa = R.pair(G.twc), wCb = R.pair(G.twc + 256), wCc = R.pair(G.twc + 512);
  const double2 wDa = R.pair(G.twd), wDb = R.pair(G.twd + 512);
  wave_fence();
  // ---- round B: stages 4, 8 ----
  {
    float2 *p = z + G.zb;
    const double2 wa = wBa, wb = wBb, wc = wBc;
    x[0] = p[0]; x[1] = p[4]; x[2] = p[8]; x[3] = p[12];
    r2_butterfly(x[0], x[1], wa); r2_butterfly(x[2], x[3], wa);
    r2_butterfly(x[0], x[2], wb); r2_butterfly(x[1], x[3], wc);
    p[0] = x[0]; p[4] = x[1]; p[8] = x[2]; p[12] = x[3];
  }
  uint32_t lw[4] = {0u, 0u, 0u, 0u};
  int pt[4] = {G.post_tab[0], G.post_tab[1], G.post_tab[2], G.post_tab[3]};
  if (late) {
    const int tab_base = (G.band2 ? (int)offsetof(C1DevTables, mdct_fwd512) : (int)offsetof(C1DevTables, mdct_fwd256)) - Tab::origin;
#pragma unroll
    for (int j = 0; j < 4; j++) { lw[j] = late[64 * j]; pt[j] = tab_base + (int)(lw[j] >> 18) * 16; }
  }
  const double2 p0 = R.pair(pt[0]), p1 = R.pair(pt[1]);
  const double2 p2 = R.pair(pt[2]), p3 = R.pair(pt[3]);
  wave_fence();
  // ---- round C: stages 16, 32 ----
  {
    float2 *p = z + G.zc;
    const double2 wa = wCa, wb = wCb, wc = wCc;
    x[0] = p[0]; x[1] = p[20]; x[2] = p[40]; x[3] = p[60];
    r2_butterfly(x[0], x[1], wa); r2_butterfly(x[2], x[3], wa);
    r2_butterfly(x[0], x[2], wb); r2_butterfly(x[1], x[3], wc);
    if (G.band2) { p[0] = x[0]; p[20] = x[1]; p[40] = x[2]; p[60] = x[3]; }
  }
  wave_fence();
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
    const uint32_t cc = late ? lw[j] : ((uint32_t)G.cx[j] | ((uint32_t)G.cy[j] << 9));
    coef[cc & 511u] = f32(-rr * t.x - ii * t.y);
    coef[(cc >> 9) & 511u] = f32(-rr * t.y + ii * t.x);
  }
}
__device__ __forceinline__ void mdct_long_r4(const float *in, float2 *z, float *coef, const R4Geometry &G, TablesPtr T, TablesRsrc R, const R4Early &E,
                                             const uint32_t *late = nullptr) {
  mdct_long_r4_t(in, z, coef, G, T, BufTab{R}, E, late);
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
  wave_fence();
  {
    float2 *p = z + G.zb;
    const double2 wa = wBa, wb = wBb, wc = wBc;
    x[0] = p[0]; x[1] = p[4]; x[2] = p[8]; x[3] = p[12];
    r2_butterfly(x[0], x[1], wa); r2_butterfly(x[2], x[3], wa);
    r2_butterfly(x[0], x[2], wb); r2_butterfly(x[1], x[3], wc);
    if (G.is_long) { p[0] = x[0]; p[4] = x[1]; p[8] = x[2]; p[12] = x[3]; }
  }
  if (any_long) {
    wave_fence();
    if (G.is_long) {
      float2 *p = z + G.zc;
      const double2 wa = table_pair(R, G.twc), wb = table_pair(R, G.twc + 256), wc = table_pair(R, G.twc + 512);
      x[0] = p[0]; x[1] = p[20]; x[2] = p[40]; x[3] = p[60];
      r2_butterfly(x[0], x[1], wa); r2_butterfly(x[2], x[3], wa);
      r2_butterfly(x[0], x[2], wb); r2_butterfly(x[1], x[3], wc);
      if (G.band2) { p[0] = x[0]; p[20] = x[1]; p[40] = x[2]; p[60] = x[3]; }
    }
    if (band2_long) {
      wave_fence();
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

}  // namespace
