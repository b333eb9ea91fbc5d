// c1_k_spec.hip -- the SPECULATIVE binary32 analysis: QMF analysis -> long-block MDCT -> scale-factor indices for
// fixed block modes [0,0,0], computed in binary32 with a rigorous bound on how far every coefficient can be from
// the reference's (which is binary64 arithmetic rounded to binary32 at every Float32Array store, encoder.js:57-349).
//
// Why: the exact formulation (c1_k_analysis.hip) is bound by fp64-rate VALU issue (4 cycles per wave instruction,
// conversions included); the same transforms in binary32 with fused multiply-adds need a third of the issue cycles.
// The integer outputs of the encoder only depend on DECISIONS taken on the coefficients -- which scale-factor
// interval a BFU's maximum falls in (bitallocation.js:290-299) and which integer x*norm +- 0.5 truncates to
// (quantization.js:43-53).  A unit is accepted from this path only when every such decision is the same for every
// value within the bound; all other units are redone by the exact kernels (work list, c1_api.hip).  DESIGN.md 3b
// derives the bound; tests/model/spec_model.c restates this kernel on the CPU operation for operation and
// tests/test_spec_bound.py checks bound >= |binary32 - reference| on the CPU.
//
// The arithmetic here is NOT the reference's order of operations and does not need to be: any binary32 algorithm
// with a proven bound serves.  It is chosen for few roundings: each 24-tap QMF sum is two chains running from the
// small outer taps to the centre (the largest tap enters last), the FFT runs radix-4 rounds with three twiddle
// products per butterfly, pre- and post-twiddle are fused multiply-adds.
#include "c1_device.h"

namespace {

// ---- LDS of one wave ---------------------------------------------------------------------------------------------
// mem is reused inside a frame (one wave: LDS operations execute in issue order, so a region may be rewritten as
// soon as every read of its previous content has been issued):
//   R1 = mem[0, 840)      stage-1 work buffer (blocks of 8 samples padded to 12 floats: conflict-free 16-byte
//                         window reads at "lane base + immediate")  ->  in2 (512, from mem[1]: a lane's four high-band
//                         outputs then start a 16-byte group)  ->  FFT points z (320 float2)
// The zero padding of the long-block MDCT inputs is never materialised: the pre-twiddle knows which of its operands
// fall into it (a lane-constant predicate) and takes 0 instead of reading.
//   R2 = mem[840, 1416)   stage-2 work buffer (302)  ->  in0 | in1 (256 each)  ->  coefficients (512)
// With short blocks (SHORT) the MDCT inputs are staged per band as E[s] = W[s & 31] x[s] behind the previous frame's
// 32 overlap values, and H[s] = x[s] W[31 - (s & 31)]: block q of a band is E[32 (q-1) ..) | H[32 q ..) (encoder.js:269-307).
//   band 2: E at mem[0, 288), H at mem[288, 544);   band 0: E at R2 + 0 (160), H at R2 + 160 (128);   band 1: R2 + 288, R2 + 448
constexpr int kR2 = 840;
constexpr int kMemFloats = kR2 + 576;
constexpr int kIn2 = 1;             // in2[i] = mem[kIn2 + i]
struct alignas(16) SpecLds {
  alignas(16) float mem[kMemFloats];
  alignas(16) float d1[48];          // stage-1 delay line (46)
  alignas(16) float d2[48];          // stage-2 delay line (46)
  alignas(16) float pre2[76];        // what the next frame's band-2 MDCT input starts with: windowed overlap (32), then the 39 delayed samples;
                                     // logical entry k lives at pre2[k + 1]: the tail lanes' four consecutive entries then start a 16-byte group
  alignas(4) uint8_t sfi[64];
  // lane-only values of the END of a frame (where the coefficients go, the post-twiddle pair, the scale-factor scan), read
  // back once per frame instead of being carried in registers through the whole loop: the register allocator spilled four
  // such values to scratch, and a scratch reload waits on the same counter as the coefficient stores issued just before it
  uint32_t geo[3][64];
};
// The binary32 tables every frame reads with lane-varying indices -- WINDOW_SHORT, the MDCT (cos, sin) pairs, the radix-4
// rounds' twiddles: C1DevTables::win32 .. r2d, contiguous, 2 784 bytes -- are kept in LDS, ONE copy per workgroup of
// kSpecWaves waves (the waves share nothing else and never meet again after the copy).  Read through the cache they were
// vector-memory loads: waits for them are waits on the counter the frame's stores share (in order), and under the write
// traffic of this kernel a store takes long enough to reach memory that the first table wait of the NEXT frame still sat
// behind it (measured with junk tables in LDS: -7 %).  With the tables in LDS the only vector-memory wait of a frame is the
// delivery of the next frame's PCM, a whole frame behind the stores.
constexpr int kSpecWaves = 4;
constexpr int kSpecTabBase = (int)offsetof(C1DevTables, win32);
constexpr int kSpecTabFloats = ((int)offsetof(C1DevTables, r2d) + (int)sizeof(((C1DevTables *)nullptr)->r2d) - kSpecTabBase) / 4;
static_assert(offsetof(C1DevTables, pre32_64) > offsetof(C1DevTables, win32) && offsetof(C1DevTables, r2d) > offsetof(C1DevTables, r4c) &&
              offsetof(C1DevTables, norm32) == offsetof(C1DevTables, r2d) + sizeof(((C1DevTables *)nullptr)->r2d), "win32 .. r2d are one contiguous block");
static_assert(kSpecWaves * sizeof(SpecLds) + kSpecTabFloats * 4 + 27 * 16 <= 32768, "speculative analysis: 5 workgroups of 4 waves per CU");
constexpr int kE2 = 0, kH2 = 288, kE0 = kR2, kH0 = kR2 + 160, kE1 = kR2 + 288, kH1 = kR2 + 448;

__device__ __forceinline__ int w1_phys(int v) { return __mul24(12, v >> 3) + (v & 7); }   // 24-bit multiply: v_mul_lo_u32 issues at a quarter of the rate

// D consecutive outputs of the decimating QMF from the lane's window w[0 .. 46 + 2 D), held as the pairs
// W[k] = (w[2 k], w[2 k + 1]).  Output d uses w[2 d ..]:
//   even = sum_j E[j] w[2 d + 47 - 2 j],   odd = sum_m E[m] w[2 d + 2 m]        (QMF_ODD[j] = QMF_EVEN[23 - j])
// Each sum is chain A (taps 0..11 ascending) + chain B (taps 23..13 descending), then the centre tap 12: the small
// outer taps first, the largest last (that order keeps the rounding bound small).  Term j of the even
// sum and term 23 - j of the odd sum read the two halves of the same pair W[d + 23 - j], so one packed FMA with the
// tap pair (E[23 - j], E[j]) advances (odd chain, even chain) together:
//   P1 = (odd B-chain m = 23..13 , even A-chain j = 0..10)      P2 = (odd A-chain m = 0..10 , even B-chain j = 23..13)
// then the two terms with tap 11 (a packed FMA whose other half multiplies by 0: x + 0 * w is exact), P1 + P2, and
// the centre taps as two plain FMAs.
template <int D, int NP>
__device__ __forceinline__ void qmf_core_f32(const v2f (&W)[NP], TablesPtr T, float (&lo)[D], float (&hi)[D]) {
  v2f tp[26];                                            // wave-uniform: aligned SGPR pairs, straight into the packed FMAs
#pragma unroll
  for (int j = 0; j < 26; j++) tp[j] = *reinterpret_cast<const __attribute__((address_space(4))) v2f *>(T->tap_pair[j]);
  const float e12 = T->tap32[12];
#pragma unroll
  for (int d = 0; d < D; d++) {
    v2f p1 = W[d + 23] * tp[0];
#pragma unroll
    for (int j = 1; j <= 10; j++) p1 = pk_fma(W[d + 23 - j], tp[j], p1);
    v2f p2 = W[d] * tp[23];
#pragma unroll
    for (int j = 22; j >= 13; j--) p2 = pk_fma(W[d + 23 - j], tp[j], p2);
    p1 = pk_fma(W[d + 12], tp[24], p1);           // even j = 11: w[2 d + 25]
    p2 = pk_fma(W[d + 11], tp[25], p2);           // odd  m = 11: w[2 d + 22]
    const v2f sum = p1 + p2;                             // (odd B + odd A, even A + even B)
    const float ev = __builtin_fmaf(e12, W[d + 11].y, sum.y);   // w[2 d + 23]
    const float od = __builtin_fmaf(e12, W[d + 12].x, sum.x);   // w[2 d + 24]
    lo[d] = ev + od;
    hi[d] = ev - od;
  }
}

// lane-only geometry of the long-block core (same ownership as mdct_long_r4: lanes 0..15 band 0, 16..31 band 1,
// 32..63 band 2, four FFT points per lane), with the MDCT inputs at mem[kR2] (in0), mem[kR2 + 256] (in1), mem[0] (in2).
// Only a few base values stay in registers across the frame loop; the 40-odd addresses of a frame are one add or
// multiply-add away from them (the register file, not the VALU, limits how many waves this kernel keeps in flight).
struct SpecBase {
  int ia0, ic0;      // float index of operands a, c of the lane's first point (position 4g)
  int q2;            // 2 * (points per quarter): the four points of a lane are 2 q2 input samples apart
  int ib, id;        // float index of the one (b, d) operand pair of the lane that is not zero padding (if any)
  int pt0;           // byte offset of the pre-twiddle pair of the first point
  int za, zb, zc, zd;
  int g;
  int e0, e1;        // coefficient indices 2 i and n2 - 1 - 2 i of the lane's first final point (band offset included)
  int po0;           // byte offset of its post-twiddle pair
  bool band0, band2, use_lo, use_hi;
};
__device__ __forceinline__ SpecBase spec_base(int lane) {
  SpecBase B;
  const int band = lane < 16 ? 0 : (lane < 32 ? 1 : 2);
  const int g = lane - (band == 0 ? 0 : (band == 1 ? 16 : 32));
  const int n4 = band == 2 ? 128 : 64, q = n4 / 4;
  const int r = bitrev(g, band == 2 ? 5 : 4);
  const int in_base = band == 0 ? kR2 : (band == 1 ? kR2 + 256 : kIn2);
  const int tab_base = band == 2 ? (int)offsetof(C1DevTables, pre32_512) : (int)offsetof(C1DevTables, pre32_256);
  B.ia0 = in_base + 3 * n4 - 1 - 2 * r;
  B.ic0 = in_base + n4 + 2 * r;
  B.q2 = 2 * q;
  // Long-block inputs are zero outside [ws, ws + 32 + band length).  Of a lane's four points only position 0 (first
  // half of the pre-twiddle, mdct.js:76-89) and position 3 (second half, :91-105) have operands b, d at all inside
  // the 2 N/4 outer samples, and they are non-zero for r < 8 (position 0) or r >= n4/4 - 8 (position 3) only.
  B.use_lo = r < 8;
  B.use_hi = r >= q - 8;
  B.ib = B.use_lo ? in_base + 3 * n4 + 2 * r : in_base + 2 * r + 2 * q;
  B.id = B.use_lo ? in_base + n4 - 1 - 2 * r : in_base + 14 * q - 1 - 2 * r;
  B.pt0 = tab_base + 8 * r;
  const int pbase = band == 0 ? 0 : (band == 1 ? 64 : 128);
  B.za = zslot(pbase + 4 * g);
  B.zb = zslot(pbase + 16 * (g >> 2) + (g & 3));
  B.zc = zslot(pbase + 64 * (g >> 4) + (g & 15));
  B.zd = zslot(128 + (g & 31));
  B.g = g;
  const int cbase = band == 0 ? 0 : (band == 1 ? 128 : 256), n2 = 2 * n4;
  B.e0 = cbase + 2 * g;
  B.e1 = cbase + n2 - 1 - 2 * g;
  B.po0 = tab_base + 8 * g;
  B.band0 = band == 0;
  B.band2 = band == 2;
  return B;
}

// one radix-4 round over two reference stages: x1, x2, x3 times wa, wb, wa*wb, then the 4-point butterfly
__device__ __forceinline__ void radix4_round(v2f (&x)[4], v2f wa, v2f wb, v2f wab) {
  const v2f y1 = cmul32(x[1], wa), y2 = cmul32(x[2], wb), y3 = cmul32(x[3], wab);
  const v2f t0 = x[0] + y1, t1 = x[0] - y1, t2 = y2 + y3, t3 = y2 - y3;
  x[0] = t0 + t2;
  x[2] = t0 - t2;
  x[1] = pk_fma(t3.yx, PMN, t1);                         // t1 - i t3 = t1 + (t3.y, -t3.x)
  x[3] = pk_fma(t3.yx, PNM, t1);                         // t1 + i t3
}

// Material-local speculation.  How many decisions of this unit will the guards leave open?  The greedy
// allocation (bitallocation.js:203-281) levels biasedSF 2^-bits over the BFUs it codes, so with L the log2 of that
// level -- the root of sum size_b clamp(lb_b - L, 0, 16) = budget, approached from the left by a Newton step on a convex
// piecewise-linear function -- a coefficient of BFU b is quantized with norm ~ 2^(bits_b - 1) / SF_b and its truncation
// is doubtful with probability ~ 2 eps norm; a scale-factor index is open with probability ~ 2 eps / (0.206 SF_b).
// The sum P of those probabilities predicts the flagged fraction 1 - exp(-P) (tools/spec_predictor_sim.py: white noise
// 0.07, pink noise with bursts 0.1-0.3, harmonics over a noise floor 3.6, stationary partials 12-200).  It only steers
// which kernels compute a unit -- either way the unit ends up bit-identical to the reference's -- so binary32
// arithmetic and approximate transcendental instructions are good enough.  One lane per BFU.
__device__ __attribute__((noinline)) bool spec_should_defer(const uint8_t *sfi, int lane, float e0, float e1, float e2,
                                                  const C1DevEncOpts *O, float threshold) {
  const int s = lane < 52 ? (int)sfi[lane] : 0;
  const float size = (float)(lane < 4 ? 8 : lane < 8 ? 4 : lane < 12 ? 8 : lane < 24 ? 6 : lane < 28 ? 7 : lane < 32 ? 9 : lane < 36 ? 10 : lane < 44 ? 12 : 20);
  const bool act = s > 0;
  const float sf = (float)s;
  const float l = __builtin_fmaf(sf, 0.33333334f, -21.0f);            // log2 SCALE_FACTORS[s]
  const float lb = __builtin_fmaf(O->la_slope, sf, O->la_off);         // log2 of the biased table
  const float eb = lane >= 36 ? e2 : (lane >= 20 ? e1 : e0);
  const float sz = act ? size : 0.0f;
  constexpr float kBudget = 1136.0f;                                   // bits the 52-BFU candidate spends (1696 - 40 - 10 * 52)
  const float n_all = wave_sum(sz);
  if (!(n_all > 0.0f)) return false;                                   // nothing coded: nothing to doubt
  float Lw = (wave_sum(sz * lb) - kBudget) * __builtin_amdgcn_rcpf(n_all);
  {
    // one Newton step (the BFUs the first guess leaves without bits drop out); further steps move P by less than 1 %
    const float d = lb - Lw;
    const float bits = fminf(fmaxf(d, 0.0f), 16.0f);
    const float spend = wave_sum(sz * bits);
    const float n = wave_sum((d > 0.0f && d < 16.0f) ? sz : 0.0f);
    if (n > 0.0f) Lw += (spend - kBudget) * __builtin_amdgcn_rcpf(n);
  }
  const float bits = fminf(fmaxf(lb - Lw, 0.0f), 16.0f);
  const float inv_sf = __builtin_amdgcn_exp2f(-l);
  const float pm = (act && bits >= 1.0f) ? size * eb * inv_sf * __builtin_amdgcn_exp2f(bits) : 0.0f;
  const float ps = act ? fminf(1.0f, 9.7f * eb * inv_sf) : 0.0f;
  const float P = wave_sum(pm + ps);
  return !(P <= threshold);                                            // bounds that are not finite: exact kernels
}

// SHORT = false: fixed block modes [0,0,0].  SHORT = true: all three bands coded with short blocks (any non-zero fixed
// modes, e.g. [2,2,3]): sixteen 64-sample MDCTs = 16-point transforms per frame, rounds A and B only.
// Which (run, channel) a workgroup takes: a pseudo-random bijection of the block index.  Workgroups are handed to the
// compute units round robin, and material comes in stretches (BASELINE configs[3]: 512-frame segments = 16 consecutive
// (run, channel) pairs, four kinds with period 64): in index order every compute unit kept getting the same kind, and
// the ones that only saw the tonal segments -- whose runs leave after their first frame -- idled while the others
// carried the whole launch (measured: 3.1 ms instead of 2.6 for a quarter of tonal runs).  Bijection on [0, 2^bits):
// odd multiplications and xor-shifts; values >= n walk on along their cycle (bits = ceil(log2 n): fewer than two
// steps on average).  Wave-uniform, scalar unit.
__device__ __forceinline__ uint32_t spread_block(uint32_t b, uint32_t n, int bits) {
  if (bits < 4) return b;
  const uint32_t mask = (1u << bits) - 1u;
  const int sh = (bits >> 1) + 1;
  uint32_t x = b;
  do {
    x = (x * 0x9E3779B1u) & mask;
    x ^= x >> sh;
    x = (x * 0x85EBCA6Bu) & mask;
    x ^= x >> sh;
  } while (x >= n);
  return x;
}

template <bool SHORT>
__global__ __launch_bounds__(C1_WAVE * kSpecWaves, 5) void k_analysis_spec(C1EncodeLaunch L) {
  __shared__ SpecLds Sw[kSpecWaves];
  __shared__ alignas(16) float tab[kSpecTabFloats];
  __shared__ alignas(16) float tail_w[27][4];               // band-2 tail weights of lanes 46..54 (this frame's input) and 46..63 (next frame's)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  SpecLds &S = Sw[wave];
  const int lane0 = threadIdx.x & 63;
  int lane = lane0;
  {
    const float *src = reinterpret_cast<const float *>(reinterpret_cast<const char *>(L.tables) + kSpecTabBase);
    for (int i = threadIdx.x; i < kSpecTabFloats; i += C1_WAVE * kSpecWaves) tab[i] = src[i];
    if (threadIdx.x < 27 * 4) {
      const int idx = threadIdx.x >> 2, j = threadIdx.x & 3;
      const bool input = idx < 9;                              // W[31 - k] x: positions 224..255 of this frame's input
      const int k = 4 * (input ? idx : idx - 9) - 1 + j;      // k = position - 224 of element j of lane 46 + ...
      tail_w[idx][j] = (k >= 0 && k < 32) ? src[input ? 31 - k : k] : 1.0f;     // src[0 .. 31] = fl32(WINDOW_SHORT)
    }
  }
  __syncthreads();                                          // the only time the waves of a workgroup meet
  const float *win = tab;                                   // fl32(WINDOW_SHORT) = the first 32 floats
  auto table_f2 = [&](int /*unused resource*/, int byte_offset) -> v2f {
    return *reinterpret_cast<const v2f *>(reinterpret_cast<const char *>(tab) + (byte_offset - kSpecTabBase));
  };
  const uint32_t n_slots = (uint32_t)((L.frames + L.run_frames - 1) / L.run_frames) * (uint32_t)L.channels;
  // the bijection moves whole workgroups: the four waves of one take four consecutive (run, channel) pairs, i.e. the same
  // material, so a tonal stretch gives its workgroups back at once (with four unrelated pairs a workgroup kept its LDS and
  // its four wave slots until the last non-tonal run was through: a quarter of tonal runs left that share of the slots idle)
  const uint32_t group = L.spread_bits > 0 ? spread_block(blockIdx.x, gridDim.x, L.spread_bits) : blockIdx.x;
  const uint32_t slot = group * kSpecWaves + (uint32_t)wave;
  if (slot >= n_slots) return;
  const int ch = (int)(slot % (uint32_t)L.channels);
  const int64_t f0 = (int64_t)(slot / (uint32_t)L.channels) * L.run_frames;
  const float *__restrict__ pcm = L.pcm[ch];
  float *mem = S.mem;

  for (int i = lane; i < 48; i += 64) { S.d1[i] = 0.0f; S.d2[i] = 0.0f; }
  float ov0a = 0.0f, ov0b = 0.0f, ov1a = 0.0f, ov1b = 0.0f;   // lanes 48..63: windowed overlap of bands 0, 1 for the next frame
  for (int i = lane; i < 76; i += 64) S.pre2[i] = 0.0f;
  if (lane < 16) reinterpret_cast<uint32_t *>(S.sfi)[lane] = 0u;
  SpecBase B0 = spec_base(lane0);              // not const: passed through an opaque asm in place, once per frame (below)
  const SfLong SFL0 = SHORT ? sf_geometry(lane0, 2, 2, 3) : sf_long_geometry(lane0);
  if constexpr (!SHORT) {
    S.geo[0][lane0] = (uint32_t)(SFL0.src & ~3) | ((uint32_t)SFL0.b << 9) | (SFL0.wide ? 1u << 15 : 0u) | (SFL0.store ? 1u << 16 : 0u);
    S.geo[1][lane0] = (uint32_t)(B0.band0 ? B0.e0 : B0.e1) | ((uint32_t)(B0.band0 ? B0.e1 : B0.e0) << 16);
    S.geo[2][lane0] = (uint32_t)B0.po0;
  }
  if (SHORT && lane == 0) S.sfi[52] = (uint8_t)((L.opts->modes[0] & 3) | ((L.opts->modes[1] & 3) << 2) | ((L.opts->modes[2] & 3) << 4));
  // short blocks: lane = (band, block, r) with four points of one 16-point transform
  const int s_band = lane0 < 16 ? 0 : (lane0 < 32 ? 1 : 2);
  const int s_g = lane0 - (s_band == 0 ? 0 : (s_band == 1 ? 16 : 32));
  const int s_blk = s_g >> 2, s_r2 = 2 * bitrev(s_g & 3, 2);
  const int s_eb0 = (s_band == 0 ? kE0 : (s_band == 1 ? kE1 : kE2)) + 32 * s_blk;
  const int s_hb0 = (s_band == 0 ? kH0 : (s_band == 1 ? kH1 : kH2)) + 32 * s_blk;
  const int s_c0 = (s_band == 0 ? 0 : (s_band == 1 ? 128 : 256)) + 32 * s_blk + 2 * (s_g & 3);   // coefficient 2 i of the first final point
  constexpr int RT = 0;                         // table_f2(RT, offset): the tables are in LDS (above)
  float p_prev = 0.0f, q_prev = 0.0f;        // PCM / stage-1-low energies of the previous frame
  wave_fence();

  const int64_t f_end = (f0 + L.run_frames < L.frames) ? f0 + L.run_frames : L.frames;
  int64_t f_first = f0 - 1;                   // one frame of history rebuilds the state
  if (f_first < -(int64_t)L.halo_frames) f_first = -(int64_t)L.halo_frames;
  if (f_first > f0) f_first = f0;
  typedef float v4f __attribute__((ext_vector_type(4)));   // whole 16-byte register groups: the delivery asm below takes them as they are
  v4f pre_a, pre_b;
  {
    const v4f *p4 = reinterpret_cast<const v4f *>(pcm + f_first * 512);
    pre_a = p4[lane0]; pre_b = p4[64 + lane0];
    // delivered before the loop: a load still pending at the loop's entry makes the compiler wait at the top of the loop, every frame
    asm volatile("" : "+v"(pre_a), "+v"(pre_b));
  }
  uint32_t deferred_from = 0xffffffffu;       // first unit of the part of this run that goes to the exact kernels (none)
  unsigned long long open_bits = 0ull;        // frames of this run whose scale-factor guard stayed open (wave-uniform)
  for (int64_t f = f_first; f < f_end; ++f) {
    const bool emit = (f >= f0);
    TablesPtr T = tables_for_this_frame(L.tables);
    lane = lane_for_this_frame(lane0);
    if constexpr (!SHORT) {
      // the MDCT's base values pass through an opaque asm once per frame (see there) -- HERE, where every path through the
      // frame passes: on the emitting path alone the loop header saw two versions of them (the warm-up frame's untouched
      // ones) and the emitting path paid 23 register moves a frame to keep them apart
      asm volatile("" : "+v"(B0.ia0), "+v"(B0.ic0), "+v"(B0.q2), "+v"(B0.ib), "+v"(B0.id), "+v"(B0.pt0));
      asm volatile("" : "+v"(B0.za), "+v"(B0.zb), "+v"(B0.zc), "+v"(B0.zd), "+v"(B0.g));
    }

    // ---------------- stage-1 work buffer, PCM energy ----------------
    float P;
    {
      const v4f a = pre_a, b = pre_b;
      // sample 4 lane + j of the frame sits at work index 46 + 4 lane + j: shifted by two samples against the lanes'
      // 16-byte groups.  Each lane takes the last two samples of its left neighbour (DPP wave shift) and writes
      // whole groups; the delay line is written afterwards and covers the two slots lane 0 filled with junk.
      {
        const float pz = dpp_read<0x138>(a.z), pw = dpp_read<0x138>(a.w);          // wave_shr:1
        const float qz0 = dpp_read<0x138>(b.z), qw0 = dpp_read<0x138>(b.w);
        const float az63 = lane_value(a.z, 63), aw63 = lane_value(a.w, 63);
        const float qz = lane == 0 ? az63 : qz0, qw = lane == 0 ? aw63 : qw0;
        const int v = 44 + 4 * lane;
        *reinterpret_cast<float4 *>(&mem[w1_phys(v)]) = make_float4(pz, pw, a.x, a.y);
        *reinterpret_cast<float4 *>(&mem[w1_phys(v + 256)]) = make_float4(qz, qw, b.x, b.y);
        if (lane == 63) *reinterpret_cast<float2 *>(&mem[w1_phys(556)]) = make_float2(b.z, b.w);
        if (lane < 46) mem[w1_phys(lane)] = S.d1[lane];
      }
      v2f p2 = V2(a.x, a.y) * V2(a.x, a.y);
      p2 = pk_fma(V2(a.z, a.w), V2(a.z, a.w), p2);
      p2 = pk_fma(V2(b.x, b.y), V2(b.x, b.y), p2);
      p2 = pk_fma(V2(b.z, b.w), V2(b.z, b.w), p2);
      const float p = p2.x + p2.y;
      P = wave_sum(p);
    }
    wave_fence();
    // ---------------- first QMF stage ----------------
    float Q;
    {
      float lo[4], hi[4];
      if (own_block()) {
        v2f W[28];
        const float4 *src = reinterpret_cast<const float4 *>(mem + __mul24(12, lane));
#pragma unroll
        for (int k = 0; k < 14; k++) {
          const float4 t = src[3 * (k >> 1) + (k & 1)];      // floats 12 (k >> 1) + 4 (k & 1): blocks of 8 padded to 12
          W[2 * k] = V2(t.x, t.y); W[2 * k + 1] = V2(t.z, t.w);
        }
        qmf_core_f32<4>(W, T, lo, hi);
      } else { for (int d = 0; d < 4; d++) { lo[d] = mem[lane + d]; hi[d] = 1.0f; } }
      {
        // the next frame's PCM: requested once the window registers are free, used a frame later.  Unconditional (the
        // last frame of a run asks for itself again): under a condition the loaded values were copied into the
        // loop-carried registers right behind the load, i.e. waited for on the spot
        const v4f *p4 = reinterpret_cast<const v4f *>(pcm + ((f + 1 < f_end) ? f + 1 : f) * 512);
        pre_a = p4[lane]; pre_b = p4[64 + lane];
      }
      if (lane < 46) { S.d1[lane] = mem[w1_phys(512 + lane)]; mem[kR2 + lane] = S.d2[lane]; }
      *reinterpret_cast<float2 *>(&mem[kR2 + 46 + 4 * lane]) = make_float2(lo[0], lo[1]);
      *reinterpret_cast<float2 *>(&mem[kR2 + 48 + 4 * lane]) = make_float2(lo[2], lo[3]);
      v2f q2 = V2(lo[0], lo[1]) * V2(lo[0], lo[1]);
      q2 = pk_fma(V2(lo[2], lo[3]), V2(lo[2], lo[3]), q2);
      const float q = q2.x + q2.y;
      Q = wave_sum(q);
      // band 2 = the high band behind its 39-sample delay (encoder.js:84-90): what the previous frame left (overlap,
      // 39 samples), then this frame's outputs; the last 32 samples of the band are windowed (encoder.js:309-316)
      if constexpr (!SHORT) {
      if (emit) for (int i = lane; i < 71; i += 64) mem[kIn2 + 112 + i] = S.pre2[1 + i];
      if (lane <= 45) {                                     // positions 39 + 4 lane .. + 3 < 224: plain samples, one 16-byte group
        if (emit) *reinterpret_cast<float4 *>(&mem[kIn2 + 183 + 4 * lane]) = make_float4(hi[0], hi[1], hi[2], hi[3]);
      } else {
        // Lanes 46..63 hold positions 223..294: the band's last 32 samples (224..255) are windowed both ways -- W[31 - k] x
        // into this frame's input, W[k] x as the next frame's overlap (k = position - 224) -- and positions >= 256 are the
        // next frame's delayed samples.  Branch-free: a lane's four weights of either kind come from a table (tail_w: 1
        // outside the windowed stretch; the product with 1 is exact), its four results are one 16-byte store each.  (Per
        // element, with three-way branches, this cost 5 % of the kernel.)
        const float4 wl = *reinterpret_cast<const float4 *>(tail_w[9 + lane - 46]);
        const v2f l01 = V2(hi[0], hi[1]) * V2(wl.x, wl.y), l23 = V2(hi[2], hi[3]) * V2(wl.z, wl.w);
        *reinterpret_cast<float4 *>(&S.pre2[4 * (lane - 46)]) = make_float4(l01.x, l01.y, l23.x, l23.y);   // logical entries 4 (lane - 46) - 1 ..
        if (lane <= 54) {
          // positions < 256 go into this frame's input; lane 54's last three land in the input's zero padding, which is never read
          const float4 wh = *reinterpret_cast<const float4 *>(tail_w[lane - 46]);
          const v2f h01 = V2(hi[0], hi[1]) * V2(wh.x, wh.y), h23 = V2(hi[2], hi[3]) * V2(wh.z, wh.w);
          if (emit) *reinterpret_cast<float4 *>(&mem[kIn2 + 183 + 4 * lane]) = make_float4(h01.x, h01.y, h23.x, h23.y);
        }
      }
      } else {
        // short blocks: every sample enters twice, E = W[pos & 31] x (second half of block q-1's... first half of the
        // NEXT block's input) and H = x W[31 - (pos & 31)]; the frame starts with the overlap and the 39 delayed samples
        if (emit) {
          if (lane < 32) mem[kE2 + lane] = S.pre2[1 + lane];
          if (lane < 39) {
            const float x = S.pre2[33 + lane];
            mem[kE2 + 32 + lane] = win[lane & 31] * x;
            mem[kH2 + lane] = x * win[31 - (lane & 31)];
          }
        }
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const int pos = 39 + 4 * lane + d;
          const float x = hi[d];
          if (pos < 256) {
            const float e = win[pos & 31] * x;
            if (emit) { mem[kE2 + 32 + pos] = e; mem[kH2 + pos] = x * win[31 - (pos & 31)]; }
            if (pos >= 224) S.pre2[1 + pos - 224] = e;
          } else S.pre2[33 + pos - 256] = x;
        }
      }
    }
    wave_fence();
    // ---------------- second QMF stage ----------------
    {
      float lo[2], hi[2];
      if (own_block()) {
        v2f W[26];
        const float4 *src = reinterpret_cast<const float4 *>(mem + kR2 + 4 * lane);
#pragma unroll
        for (int k = 0; k < 13; k++) {
          const float4 t = src[k];
          W[2 * k] = V2(t.x, t.y); W[2 * k + 1] = V2(t.z, t.w);
        }
        qmf_core_f32<2>(W, T, lo, hi);
      } else { for (int d = 0; d < 2; d++) { lo[d] = mem[lane + d]; hi[d] = 1.0f; } }
      {
        // (a lane index of its own: from `lane` the compiler derives this 4-byte-stride address as the window's 16-byte-stride
        // one minus 12 lane -- a 64-bit multiply-add at a quarter of the issue rate)
        const int ld = lane_for_this_frame(lane0);
        if (ld < 46) S.d2[ld] = mem[kR2 + 256 + ld];
      }
      if constexpr (!SHORT) {
      if (lane < 48) {
        if (emit) {
          *reinterpret_cast<float2 *>(&mem[kR2 + 80 + 2 * lane]) = make_float2(lo[0], lo[1]);
          *reinterpret_cast<float2 *>(&mem[kR2 + 256 + 80 + 2 * lane]) = make_float2(hi[0], hi[1]);
        }
      } else {
        // the last 32 samples of bands 0, 1 (encoder.js:309-316): windowed into this frame's input, and, with the
        // mirrored window, kept in registers as the next frame's overlap, which these same lanes write then
        const int k = 2 * (lane - 48);
        const float wl0 = win[k], wl1 = win[k + 1], wh0 = win[31 - k], wh1 = win[30 - k];
        if (emit) {
          *reinterpret_cast<float2 *>(&mem[kR2 + 48 + k]) = make_float2(ov0a, ov0b);
          *reinterpret_cast<float2 *>(&mem[kR2 + 256 + 48 + k]) = make_float2(ov1a, ov1b);
          *reinterpret_cast<float2 *>(&mem[kR2 + 80 + 2 * lane]) = make_float2(lo[0] * wh0, lo[1] * wh1);
          *reinterpret_cast<float2 *>(&mem[kR2 + 256 + 80 + 2 * lane]) = make_float2(hi[0] * wh0, hi[1] * wh1);
        }
        ov0a = wl0 * lo[0]; ov0b = wl1 * lo[1];
        ov1a = wl0 * hi[0]; ov1b = wl1 * hi[1];
      }
      } else {
        const int k = (2 * lane) & 31;
        const float wl0 = win[k], wl1 = win[k + 1], wh0 = win[31 - k], wh1 = win[30 - k];
        const float e00 = wl0 * lo[0], e01 = wl1 * lo[1], e10 = wl0 * hi[0], e11 = wl1 * hi[1];
        if (emit) {
          if (lane >= 48) {                                 // the overlap the previous frame left: E[0, 32)
            *reinterpret_cast<float2 *>(&mem[kE0 + 2 * (lane - 48)]) = make_float2(ov0a, ov0b);
            *reinterpret_cast<float2 *>(&mem[kE1 + 2 * (lane - 48)]) = make_float2(ov1a, ov1b);
          }
          *reinterpret_cast<float2 *>(&mem[kE0 + 32 + 2 * lane]) = make_float2(e00, e01);
          *reinterpret_cast<float2 *>(&mem[kE1 + 32 + 2 * lane]) = make_float2(e10, e11);
          *reinterpret_cast<float2 *>(&mem[kH0 + 2 * lane]) = make_float2(lo[0] * wh0, lo[1] * wh1);
          *reinterpret_cast<float2 *>(&mem[kH1 + 2 * lane]) = make_float2(hi[0] * wh0, hi[1] * wh1);
        }
        if (lane >= 48) { ov0a = e00; ov0b = e01; ov1a = e10; ov1b = e11; }
      }
    }
    const float W = __builtin_amdgcn_sqrtf(P + p_prev), Lw = __builtin_amdgcn_sqrtf(Q + q_prev);
    p_prev = P; q_prev = Q;
    wave_fence();
    // The next frame's PCM, requested during the first QMF stage, is taken delivery of HERE: before this frame's stores are
    // issued -- loads and stores share one in-order counter on this part (vmcnt), so a wait for the PCM placed behind the
    // stores, where the compiler would put it (the loop's back edge), is a wait for the stores to reach memory as well -- and
    // at a point every path to the top of the loop passes: a load still pending on one path (the warm-up frame's `continue`)
    // makes the compiler wait at the top of the loop on all of them.
    if (!emit) {
      asm volatile("" : "+v"(pre_a), "+v"(pre_b));
      continue;
    }

    v2f x[4];
    float zrow;
    if constexpr (!SHORT) {
    // ---------------- long-block MDCT in binary32 ----------------
    // position 4g + j of the lane holds point k_j = r + q * bitrev2(j): j = 1 -> 2q, j = 2 -> q, j = 3 -> 3q
    // (the base values pass through an opaque asm once per frame: otherwise every address derived from them is
    // loop invariant, gets hoisted out of the frame loop and spilled)
    // (in place, on the loop-carried registers themselves: through a per-frame copy it cost 14 register moves a frame)
    const SpecBase &B = B0;
    {
      const int qb = 4 * B.q2;                               // bytes between the pre-twiddle pairs of points q apart
      const v2f t0 = table_f2(RT, B.pt0), t1 = table_f2(RT, B.pt0 + 2 * qb);
      const v2f t2 = table_f2(RT, B.pt0 + qb), t3 = table_f2(RT, B.pt0 + __mul24(3, qb));
      const float a0 = mem[B.ia0], c0 = mem[B.ic0];
      const float a1 = mem[B.ia0 - 2 * B.q2], c1 = mem[B.ic0 + 2 * B.q2];
      const float a2 = mem[B.ia0 - B.q2], c2 = mem[B.ic0 + B.q2];
      const int q3 = __mul24(3, B.q2);
      const float a3 = mem[B.ia0 - q3], c3 = mem[B.ic0 + q3];
      const float bb = mem[B.ib], dd = mem[B.id];             // the lane's one pair outside the zero padding (or unused)
      const float b0 = B.use_lo ? bb : 0.0f, d0 = B.use_lo ? dd : 0.0f;
      const float b3 = B.use_hi ? bb : 0.0f, d3 = B.use_hi ? dd : 0.0f;
      // (r, m) = (a + b, c - d) in the first half of the pre-twiddle (mdct.js:76-89), (a - b, c + d) in the second
      // (:91-105); for positions 1, 2 the operands b, d are the zero padding.  Then (r c + m s, m c - r s).
      const v2f rm0 = pk_fma(V2(b0, d0), PMN, V2(a0, c0)), rm3 = pk_fma(V2(b3, d3), PNM, V2(a3, c3));
      const v2f rm1 = V2(a1, c1), rm2 = V2(a2, c2);
      auto twiddle = [](v2f rm, v2f t) { const v2f u = rm.yx * (t.yy * PMN); return pk_fma(rm, t.xx, u); };   // u = (m s, -(r s))
      x[0] = twiddle(rm0, t0); x[1] = twiddle(rm1, t1); x[2] = twiddle(rm2, t2); x[3] = twiddle(rm3, t3);
      v2f en2 = x[0] * x[0];
#pragma unroll
      for (int j = 1; j < 4; j++) en2 = pk_fma(x[j], x[j], en2);
      zrow = row_allreduce(en2.x + en2.y);
      // stages 1, 2: twiddles 1 and -i, no products
      const v2f u0 = x[0] + x[1], u1 = x[0] - x[1], u2 = x[2] + x[3], u3 = x[2] - x[3];
      x[0] = u0 + u2;
      x[2] = u0 - u2;
      x[1] = pk_fma(u3.yx, PMN, u1);
      x[3] = pk_fma(u3.yx, PNM, u1);
    }
    v2f *z = reinterpret_cast<v2f *>(mem);
    {
      float4 *dst = reinterpret_cast<float4 *>(z + B.za);
      dst[0] = make_float4(x[0].x, x[0].y, x[1].x, x[1].y);
      dst[1] = make_float4(x[2].x, x[2].y, x[3].x, x[3].y);
    }
    const int twb = (int)offsetof(C1DevTables, r4b) + 24 * (B.g & 3);
    const v2f wBa = table_f2(RT, twb), wBb = table_f2(RT, twb + 8), wBc = table_f2(RT, twb + 16);
    wave_fence();
    {
      v2f *p = z + B.zb;
      x[0] = p[0]; x[1] = p[4]; x[2] = p[8]; x[3] = p[12];
      radix4_round(x, wBa, wBb, wBc);
      p[0] = x[0]; p[4] = x[1]; p[8] = x[2]; p[12] = x[3];
    }
    const int twc = (int)offsetof(C1DevTables, r4c) + 24 * (B.g & 15), twd = (int)offsetof(C1DevTables, r2d) + 8 * (B.g & 31);
    const v2f wCa = table_f2(RT, twc), wCb = table_f2(RT, twc + 8), wCc = table_f2(RT, twc + 16);
    const v2f wDa = table_f2(RT, twd), wDb = table_f2(RT, twd + 256);
    wave_fence();
    {
      v2f *p = z + B.zc;
      x[0] = p[0]; x[1] = p[20]; x[2] = p[40]; x[3] = p[60];
      radix4_round(x, wCa, wCb, wCc);
      if (B.band2) { p[0] = x[0]; p[20] = x[1]; p[40] = x[2]; p[60] = x[3]; }
    }
    // final points: bands 0/1 hold g + 16 j after round C; band 2 holds g, g + 64, g + 32, g + 96 after round D
    const int d1 = B.band2 ? 64 : 16, d2 = 32, d3 = B.band2 ? 96 : 48;
    const int po0 = (int)S.geo[2][lane];
    const uint32_t ew = S.geo[1][lane];                      // where -o.x and o.y of the first final point go (band 0: 2 i and n2 - 1 - 2 i; bands 1, 2 reversed)
    const v2f p0 = table_f2(RT, po0), p1 = table_f2(RT, po0 + 8 * d1);
    const v2f p2 = table_f2(RT, po0 + 8 * d2), p3 = table_f2(RT, po0 + 8 * d3);
    wave_fence();
    if (B.band2) {
      const v2f *p = z + B.zd;
      x[0] = p[0]; x[1] = p[80]; x[2] = p[40]; x[3] = p[120];
      const v2f y1 = cmul32(x[1], wDa), y3 = cmul32(x[3], wDb);
      const v2f e0 = x[0], e2 = x[2];
      x[0] = e0 + y1; x[1] = e0 - y1; x[2] = e2 + y3; x[3] = e2 - y3;
    }
    float *coefw = mem + kR2;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const v2f t = j == 0 ? p0 : (j == 1 ? p1 : (j == 2 ? p2 : p3));
      const int dj = j == 0 ? 0 : (j == 1 ? d1 : (j == 2 ? d2 : d3));
      const int step = B.band0 ? 2 * dj : -2 * dj;           // bands 1 and 2 are stored reversed (utils.js:42-48)
      // mdct.js:110-119: out[2 i] = -(re c + im s), out[n2 - 1 - 2 i] = im c - re s
      const v2f u = x[j].yx * (t.yy * PMN);                  // (im s, -(re s))
      const v2f o = pk_fma(x[j], t.xx, u);
      coefw[(int)(ew & 0xffffu) + step] = -o.x;
      coefw[(int)(ew >> 16) - step] = o.y;
    }
    wave_fence();

    }
    float *coef = mem + kR2;
    if constexpr (SHORT) {
    // ---------------- short-block MDCTs in binary32: 16 blocks of 64 samples, a 16-point transform each ----------------
    // input idx of a block: idx < 32 -> E[32 blk + idx] (the block before, windowed W[i]; the overlap for block 0),
    // else H[32 blk + idx - 32].  Point k = r + 4 bitrev2(j) at position j, i = 2 k (mdct.js:76-105 with N = 64).
    int eb = s_eb0, hb = s_hb0, r2 = s_r2;
    asm volatile("" : "+v"(eb), "+v"(hb), "+v"(r2));
    {
      const int pt = (int)offsetof(C1DevTables, pre32_64) + 4 * r2;
      const v2f t0 = table_f2(RT, pt), t1 = table_f2(RT, pt + 64), t2 = table_f2(RT, pt + 32), t3 = table_f2(RT, pt + 96);
      // position 0: i = r2 (first half), 2: i = r2 + 8 (first half), 1: i = r2 + 16 (second half), 3: i = r2 + 24 (second half)
      const float a0 = mem[hb + 15 - r2], b0 = mem[hb + 16 + r2], c0 = mem[eb + 16 + r2], d0 = mem[eb + 15 - r2];
      const float a2 = mem[hb + 7 - r2], b2 = mem[hb + 24 + r2], c2 = mem[eb + 24 + r2], d2 = mem[eb + 7 - r2];
      const float a1 = mem[eb + 31 - r2], b1 = mem[eb + r2], c1 = mem[hb + r2], d1 = mem[hb + 31 - r2];
      const float a3 = mem[eb + 23 - r2], b3 = mem[eb + 8 + r2], c3 = mem[hb + 8 + r2], d3 = mem[hb + 23 - r2];
      const v2f rm0 = pk_fma(V2(b0, d0), PMN, V2(a0, c0)), rm2 = pk_fma(V2(b2, d2), PMN, V2(a2, c2));   // (a + b, c - d)
      const v2f rm1 = pk_fma(V2(b1, d1), PNM, V2(a1, c1)), rm3 = pk_fma(V2(b3, d3), PNM, V2(a3, c3));   // (a - b, c + d)
      auto twiddle = [](v2f rm, v2f t) { const v2f u = rm.yx * (t.yy * PMN); return pk_fma(rm, t.xx, u); };
      x[0] = twiddle(rm0, t0); x[1] = twiddle(rm1, t1); x[2] = twiddle(rm2, t2); x[3] = twiddle(rm3, t3);
      v2f en2 = x[0] * x[0];
#pragma unroll
      for (int j = 1; j < 4; j++) en2 = pk_fma(x[j], x[j], en2);
      // energy of the block's 16 points (four lanes), then the largest block of the row
      float en = en2.x + en2.y;
      en += dpp_read<0xB1>(en);
      en += dpp_read<0x4E>(en);
      en = fmaxf(en, dpp_read<0x141>(en));
      zrow = fmaxf(en, dpp_read<0x140>(en));
      const v2f u0 = x[0] + x[1], u1 = x[0] - x[1], u2 = x[2] + x[3], u3 = x[2] - x[3];
      x[0] = u0 + u2;
      x[2] = u0 - u2;
      x[1] = pk_fma(u3.yx, PMN, u1);
      x[3] = pk_fma(u3.yx, PNM, u1);
    }
    {
      v2f *z = reinterpret_cast<v2f *>(mem);
      int za = B0.za, zb = B0.zb, g3 = s_g & 3;
      asm volatile("" : "+v"(za), "+v"(zb), "+v"(g3));
      const int twb = (int)offsetof(C1DevTables, r4b) + 24 * g3;
      const v2f wBa = table_f2(RT, twb), wBb = table_f2(RT, twb + 8), wBc = table_f2(RT, twb + 16);
      const int po = (int)offsetof(C1DevTables, pre32_64) + 8 * g3;
      const v2f p0 = table_f2(RT, po), p1 = table_f2(RT, po + 32), p2 = table_f2(RT, po + 64), p3 = table_f2(RT, po + 96);
      wave_fence();                                     // every lane has read its inputs: the points may overwrite them
      float4 *dst = reinterpret_cast<float4 *>(z + za);
      dst[0] = make_float4(x[0].x, x[0].y, x[1].x, x[1].y);
      dst[1] = make_float4(x[2].x, x[2].y, x[3].x, x[3].y);
      wave_fence();
      const v2f *p = z + zb;
      x[0] = p[0]; x[1] = p[4]; x[2] = p[8]; x[3] = p[12];
      radix4_round(x, wBa, wBb, wBc);
      wave_fence();
      float *coefw = mem + kR2;
      int c0i = s_c0;
      asm volatile("" : "+v"(c0i));
      const bool band0 = B0.band0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const v2f t = j == 0 ? p0 : (j == 1 ? p1 : (j == 2 ? p2 : p3));
        const int e0 = c0i + 8 * j, e1 = (c0i | 31) - ((c0i & 31) + 8 * j);     // 32 blk + 2 i  and  32 blk + 31 - 2 i
        const v2f u = x[j].yx * (t.yy * PMN);
        const v2f o = pk_fma(x[j], t.xx, u);
        coefw[band0 ? e0 : e1] = -o.x;
        coefw[band0 ? e1 : e0] = o.y;
      }
    }
    wave_fence();

    }
    // ---------------- the bound, coefficients out, scale-factor indices with their guard ----------------
    const int64_t unit = f * L.channels + ch;
    asm volatile("" : "+v"(pre_a), "+v"(pre_b));
    {
      // streaming stores (the packing kernel reads these 4 GB long after they have left the cache): -2 % of the kernel
      v4f *dst = reinterpret_cast<v4f *>(L.coefs + (unit << 9));
      const v4f *src = reinterpret_cast<const v4f *>(coef);
      const int lq = lane_for_this_frame(lane0);               // its own lane index, as for the delay line above: 16-byte stride here
      __builtin_nontemporal_store(src[lq], &dst[lq]);
      __builtin_nontemporal_store(src[64 + lq], &dst[64 + lq]);
    }
    // eps_b = cz_b Z_b + cw_b W + cl_b L + eabs  (DESIGN.md 3b); Z_b^2 = energy of the band's pre-twiddled points
    // (short blocks: of the block with the most energy; the coefficients then belong to 16-point transforms)
    const float Z0 = __builtin_amdgcn_sqrtf(lane_value(zrow, 0)), Z1 = __builtin_amdgcn_sqrtf(lane_value(zrow, 16));
    const float Z2 = __builtin_amdgcn_sqrtf(SHORT ? fmaxf(lane_value(zrow, 32), lane_value(zrow, 48)) : lane_value(zrow, 32) + lane_value(zrow, 48));
    const float eabs = T->spec_eabs;
    const float cz0 = SHORT ? T->spec_cz_short[0] : T->spec_cz[0], cz1 = SHORT ? T->spec_cz_short[1] : T->spec_cz[1], cz2 = SHORT ? T->spec_cz_short[2] : T->spec_cz[2];
    const float cw0 = SHORT ? T->spec_cw_short[0] : T->spec_cw[0], cw1 = SHORT ? T->spec_cw_short[1] : T->spec_cw[1], cw2 = SHORT ? T->spec_cw_short[2] : T->spec_cw[2];
    const float cl0 = SHORT ? T->spec_cl_short[0] : T->spec_cl[0], cl1 = SHORT ? T->spec_cl_short[1] : T->spec_cl[1], cl2 = SHORT ? T->spec_cl_short[2] : T->spec_cl[2];
    const float e0 = __builtin_fmaf(cz0, Z0, __builtin_fmaf(cw0, W, __builtin_fmaf(cl0, Lw, eabs)));
    const float e1 = __builtin_fmaf(cz1, Z1, __builtin_fmaf(cw1, W, __builtin_fmaf(cl1, Lw, eabs)));
    const float e2 = __builtin_fmaf(cz2, Z2, __builtin_fmaf(cw2, W, __builtin_fmaf(cl2, Lw, eabs)));
    bool unstable;
    {
      SfLong SFL = SFL0;
      float mx = 0.0f;
      if constexpr (SHORT) {
        asm volatile("" : "+v"(SFL.src), "+v"(SFL.cnt));
        const float *src = coef + SFL.src;
#pragma unroll
        for (int j = 0; j < 12; j++) mx = fmaxf(mx, fabsf(src[j < SFL.cnt ? j : SFL.cnt - 1]));
      } else {
        const uint32_t sw = S.geo[0][lane];
        SFL.b = (int)((sw >> 9) & 63u);
        SFL.wide = ((sw >> 15) & 1u) != 0u;
        SFL.store = ((sw >> 16) & 1u) != 0u;
        const float4 *grp = reinterpret_cast<const float4 *>(coef + (sw & 0x1ffu));
        mx = sf_scan_long_groups(grp[0], grp[1], grp[2]);
      }
      const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0xB1, 0xf, 0xf, false));
      mx = fmaxf(mx, SFL.wide ? other : 0.0f);
      const float e = SFL.b >= 36 ? e2 : (SFL.b >= 20 ? e1 : e0);
      // every value the reference's maximum can take lies in [mx - e, mx + e]; the index is monotone in the maximum,
      // so it is certain when both ends (widened by the rounding of this very subtraction / addition) agree
      const float lo = fmaxf((mx - e) * 0.99999976f, 0.0f), hi = (mx + e) * 1.00000024f;
      const int s_lo = scale_factor_index_fast(lo, T->sf_m1, T->sf_m2), s_hi = scale_factor_index_fast(hi, T->sf_m1, T->sf_m2);
      // lo <= mx <= hi and the index is monotone: when both ends agree, that is the index of mx too
      if (SFL.store) S.sfi[SFL.b] = (uint8_t)s_lo;
      unstable = lane < 60 && !(s_lo == s_hi && e < __builtin_huge_valf());
    }
    const bool any_unstable = __builtin_amdgcn_ballot_w64(unstable) != 0;
    wave_fence();
    // every kSpecCheckFrames frames: is this material worth speculating on?  If not, the rest of the run (this frame
    // included: nothing of it has been handed on yet but the coefficients, which the exact kernel overwrites) goes to
    // the exact kernels' run list
    if (L.defer_list != nullptr && (((int)(f - f0)) & (kSpecCheckFrames - 1)) == 0) {
      if (spec_should_defer(S.sfi, lane, e0, e1, e2, L.opts, L.spec_defer)) { deferred_from = (uint32_t)unit; break; }
    }
    if (any_unstable) open_bits |= 1ull << (int)((f - f0) & 63);        // (behind the hand-over test: a deferred frame is the exact kernels' already)
    if (lane < 16) reinterpret_cast<uint32_t *>(L.side + unit * kSideBytes)[lane] = reinterpret_cast<const uint32_t *>(S.sfi)[lane];
    if (lane == 0) *reinterpret_cast<float4 *>(L.eps + unit * kEpsFloats) = make_float4(e0, e1, e2, __int_as_float(any_unstable ? 1 : 0));
    wave_fence();
  }
  // one slot per run and channel, written by every workgroup (no list appends: tens of thousands of atomics on one counter
  // serialise at ~6-20 ns each, which cost 0.5 ms per 2 M units when a quarter of the runs were handed over)
  if (L.defer_list != nullptr && lane0 == 0) L.defer_list[slot] = deferred_from;
  if (L.open_masks != nullptr && lane0 == 0) L.open_masks[slot] = open_bits;
}

// the slots of the deferred runs -> a dense list for the exact kernels (one atomic per wave; the order of the list is
// free: every entry is processed on its own).  counts[0] = entries, counts[1] = units they cover.
__global__ __launch_bounds__(256) void k_defer_compact(const uint32_t *__restrict__ slots, uint32_t n_slots, uint32_t *__restrict__ list, uint32_t *__restrict__ counts,
                                                        int channels, int run_frames, uint32_t frames) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t u = i < n_slots ? slots[i] : 0xffffffffu;
  const bool on = u != 0xffffffffu;
  uint32_t covered = 0;
  if (on) {
    const uint32_t f = u / (uint32_t)channels;
    uint32_t end = (f / (uint32_t)run_frames + 1u) * (uint32_t)run_frames;
    if (end > frames) end = frames;
    covered = end - f;
  }
  const uint64_t mask = __builtin_amdgcn_ballot_w64(on);
  if (mask == 0) return;
  int cov = (int)covered;
  cov = wave_inclusive_scan(cov);
  const int lane = threadIdx.x & 63;
  uint32_t at = 0;
  if (lane == 63) {
    at = atomicAdd(counts, (uint32_t)__popcll(mask));
    atomicAdd(counts + 1, (uint32_t)cov);
  }
  at = (uint32_t)__builtin_amdgcn_readlane((int)at, 63);
  if (on) list[at + __popcll(mask & ((1ull << lane) - 1ull))] = u;
}

// the open-scale-factor masks of the runs -> a dense list of units (one atomic per wave; the order of the list is free)
__global__ __launch_bounds__(256) void k_open_compact(const unsigned long long *__restrict__ masks, uint32_t n_slots, uint32_t *__restrict__ list,
                                                       uint32_t *__restrict__ count, int channels, int run_frames) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  unsigned long long m = i < n_slots ? masks[i] : 0ull;
  const int mine = __popcll(m);
  if (__builtin_amdgcn_ballot_w64(mine != 0) == 0) return;
  const int scan = wave_inclusive_scan(mine);
  const int lane = threadIdx.x & 63;
  uint32_t at = 0;
  if (lane == 63) at = atomicAdd(count, (uint32_t)scan);
  at = (uint32_t)__builtin_amdgcn_readlane((int)at, 63) + (uint32_t)(scan - mine);
  const uint32_t run = i / (uint32_t)channels, ch = i % (uint32_t)channels;
  while (m != 0ull) {
    const int k = __ffsll((long long)m) - 1;
    m &= m - 1ull;
    list[at++] = (run * (uint32_t)run_frames + (uint32_t)k) * (uint32_t)channels + ch;
  }
}

// kind 0: a speculative call -- counts = the list head (redo, realloc, re-analysis, deferred runs, deferred units);
// kind 1: exact coefficients quantized in binary32 -- counts[0] = units packed again;  kind 2: the speculative detector --
// counts[0] = units rechecked.  Statistics only (c1_ctx_*_stats).
__global__ void k_spec_totals(unsigned long long *totals, unsigned long long units, const uint32_t *counts, int kind) {
  if (kind == 0) {
    const unsigned long long deferred = counts[4];
    totals[0] += units - deferred;
    totals[1] += counts[2] + counts[5];                     // re-analysed behind the packing pass + in front of the allocation (open scale factors)
    totals[2] += deferred;
    totals[3] += counts[0] - counts[2];
    totals[6] += deferred;
  } else if (kind == 1) {
    totals[2] += units;
    totals[3] += counts[0];
  } else {
    totals[4] += units;
    totals[5] += counts[0];
  }
}

}  // namespace

void c1k_launch_spec_totals(unsigned long long *totals, uint64_t units, const uint32_t *counts, int kind, hipStream_t stream) {
  hipLaunchKernelGGL(k_spec_totals, dim3(1), dim3(1), 0, stream, totals, (unsigned long long)units, counts, kind);
}
void c1k_launch_open_compact(const C1EncodeLaunch &L, uint32_t *list, uint32_t *count, hipStream_t stream) {
  const int run = c1k_pick_run(L.frames, L.channels, 0);
  const int64_t slots = (L.frames + run - 1) / run * L.channels;
  hipLaunchKernelGGL(k_open_compact, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, (const unsigned long long *)L.open_masks, (uint32_t)slots, list, count,
                     L.channels, run);
}
void c1k_launch_defer_compact(const C1EncodeLaunch &L, uint32_t *list, uint32_t *counts, hipStream_t stream) {
  const int run = c1k_pick_run(L.frames, L.channels, 0);
  const int64_t slots = (L.frames + run - 1) / run * L.channels;
  hipLaunchKernelGGL(k_defer_compact, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, (const uint32_t *)L.defer_list, (uint32_t)slots, list, counts,
                     L.channels, run, (uint32_t)L.frames);
}

void c1k_launch_analysis_spec(const C1EncodeLaunch &L0, bool all_short, hipStream_t stream) {
  static const int slots = c1k_wave_slots(k_analysis_spec<false>);
  C1EncodeLaunch L = L0;
  L.run_frames = c1k_pick_run(L.frames, L.channels, slots);
  const int64_t runs = (L.frames + L.run_frames - 1) / L.run_frames, n_slots = runs * L.channels;
  const dim3 grid((unsigned)((n_slots + kSpecWaves - 1) / kSpecWaves)), block(C1_WAVE * kSpecWaves);
  static const bool no_spread = getenv("C1_NO_SPREAD") != nullptr;     // experiments: runs in stream order
  L.spread_bits = 0;
  if (!no_spread) while ((1ll << L.spread_bits) < (int64_t)grid.x) L.spread_bits++;   // the bijection is over workgroups
  if (all_short) hipLaunchKernelGGL((k_analysis_spec<true>), grid, block, 0, stream, L);
  else hipLaunchKernelGGL((k_analysis_spec<false>), grid, block, 0, stream, L);
}
