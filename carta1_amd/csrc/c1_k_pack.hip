// c1_k_pack.hip -- quantize + MSB-first packing of the 212-byte sound unit (quantization.js:34-56, serialization.js:41-98)
#include "c1_device.h"

namespace {

// =====================================================================================================
// k_pack : quantize (quantization.js:34-56) + serializeFrame (serialization.js:41-98)
// =====================================================================================================
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
// persistent grid: waves stride over the units.  Exactly the workgroups the device holds at once (pack_resident_blocks): with
// more, the second round only partly fills the machine (2 048 workgroups at 1 280 resident: 1.085 ms per 2 M units; 2 560: 1.065;
// 1 280: 1.024)

constexpr int kRedoBatch = 32;           // one atomic on the list's counter per 32 listed units of a wave (every listed unit paying
                                         // its own serialises the whole kernel on that one address when most units are listed)
struct alignas(16) PackLds {
  uint32_t words[56];         // the unit as big-endian 32-bit groups
  uint32_t desc[52];          // per BFU: bits(5) | mantissa bit offset(11) << 5 | first coefficient(9) << 16
  double normd[52];           // per BFU: quantRange / SCALE_FACTORS[sfi], 0 when nothing is coded
  // speculative path, one 16-byte record per BFU (one ds_read_b128 per mantissa): bits | offset << 5, fl32(norm), the guard
  // band eps_band * norm (1 + 2^-20) + 2^-22, the quantizer's range 2^(bits-1) - 1 plus one half, as a binary32 number
  alignas(16) uint4 rec[52];
  uint32_t redo[kRedoBatch];  // speculative path: units to redo, appended to the global list a batch at a time
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
  float eps;             // speculative path: eps[lane & 3] of the unit (bands 0..2, flag word)
};
template <bool SPEC>
__device__ __forceinline__ PackHeader pack_load_header(const C1EncodeLaunch &L, int64_t unit, int lane) {
  const uint32_t *al = reinterpret_cast<const uint32_t *>(L.alloc + unit * kAllocBytes);
  const uint32_t *side = reinterpret_cast<const uint32_t *>(L.side + unit * kSideBytes);
  PackHeader h;
  h.al_wl = al[lane >> 3];
  // wave-uniform (amount index, fallback flag): a SCALAR load.  As a vector load it was converted to a scalar with
  // v_readfirstlane right behind the load -- a wait, at the top of every unit, for a load just issued and, the counter being
  // in order, for the previous unit's store before it (tools/isa_waits.py)
  h.al7 = *reinterpret_cast<const __attribute__((address_space(4))) uint32_t *>(reinterpret_cast<uintptr_t>(al + 7));
  h.sd_sf = side[lane >> 2];
  h.sd_q = side[lane & 15];
  h.al_a = al[(lane + 7) & 7];
  h.al_b = al[lane & 7];
  h.eps = SPEC ? L.eps[unit * kEpsFloats + (lane & 3)] : 0.0f;
  return h;
}

// ALL_LONG: the caller knows every unit of the batch has modes [0,0,0] (fixed block modes): coefficient order ==
// slot order, no per-slot position tables
// SPEC: the coefficients come from the speculative binary32 analysis and carry a per-band bound eps on their distance
// from the reference's (c1_k_spec.hip).  Then x * norm + 0.5 is formed in binary32 and a mantissa is accepted only
// when no value within the bound (plus the rounding of this very computation) truncates to another integer; a unit
// with any doubtful mantissa, or whose scale-factor indices were doubtful, goes to the redo list and is encoded
// again by the exact kernels.
template <bool ALL_LONG, bool SPEC>
__global__ __launch_bounds__(C1_WAVE * kPackWaves, ALL_LONG ? 5 : 4) void k_pack(C1EncodeLaunch L) {
  __shared__ PackLds lds[kPackWaves];
  __shared__ typename std::conditional<SPEC, float, double>::type norm_s[64 * 16];   // quantRange / SCALE_FACTORS[sfi] (quantization.js:42-44)
  TablesPtr T = C1_TABLES(L.tables);
  // the wave index is uniform across the wave; telling the compiler so moves the per-unit address arithmetic (unit
  // index, record and coefficient pointers, this wave's LDS block) from the vector to the scalar unit
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  PackLds &S = lds[wave];
  for (int i = threadIdx.x; i < 64 * 16; i += C1_WAVE * kPackWaves) {
    if constexpr (SPEC) norm_s[i] = T->norm32[i]; else norm_s[i] = T->norm[i];
  }
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
  // list mode (exact redo of the units the speculative pass could not certify): positions index L.unit_list
  const bool listed = L.unit_list != nullptr;
  const int64_t units_total = listed ? (int64_t)*L.unit_count : L.frames * L.channels;
  const int64_t stride = (int64_t)gridDim.x * kPackWaves;
  const int64_t u_first = (int64_t)blockIdx.x * kPackWaves + wave;
  auto unit_at = [&](int64_t pos) -> int64_t { return listed ? (int64_t)L.unit_list[pos] : pos; };
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
  int n_redo = 0;                          // wave-uniform: entries waiting in S.redo
  auto flush_redo = [&]() {
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(L.redo_count, (uint32_t)n_redo);
    at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
    const uint32_t entry = lane < n_redo ? S.redo[lane] : 0x40000000u;
    if (lane < n_redo) L.redo_list[at + lane] = entry & 0x3fffffffu;
    // bit 31: a scale-factor index of the unit was open, so its bits are allocated again as well (the allocation reads
    // nothing but the indices, bitallocation.js:74-142; the exact analysis of the others reproduces the ones it ran on)
    const uint64_t open_mask = __builtin_amdgcn_ballot_w64((entry >> 31) != 0u);
    if (open_mask != 0) {
      uint32_t at2 = 0;
      if (lane == 0) at2 = atomicAdd(L.realloc_count, (uint32_t)__popcll(open_mask));
      at2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)at2);
      if ((entry >> 31) != 0u) L.realloc_list[at2 + __popcll(open_mask & ((1ull << lane) - 1ull))] = entry & 0x3fffffffu;
    }
    // bit 30 clear: the coefficients are binary32 ones, the exact analysis has to rebuild them (a unit whose coefficients
    // are the exact kernels' already -- bounds of zero -- is only packed again)
    const uint64_t ana_mask = __builtin_amdgcn_ballot_w64((entry & 0x40000000u) == 0u);
    if (ana_mask != 0) {
      uint32_t at3 = 0;
      if (lane == 0) at3 = atomicAdd(L.reana_count, (uint32_t)__popcll(ana_mask));
      at3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)at3);
      if ((entry & 0x40000000u) == 0u) L.reana_list[at3 + __popcll(ana_mask & ((1ull << lane) - 1ull))] = entry & 0x3fffffffu;
    }
    n_redo = 0;
  };
  // Two register sets (header + coefficients) alternate between "the unit being packed" and "the unit in flight": the
  // loop body exists twice with the sets swapped, so nothing is copied from one iteration to the next (the
  // h0 = h1, h1 = h2, x = xn rotation of a three-deep pipeline was 22 of the kernel's 310 vector instructions per unit).
  // The block modes a mixed-mode batch needs to address the next unit's coefficients are fetched one unit earlier.
  // In list mode the index of a unit is itself a load: it is fetched two units ahead (a wave-uniform scalar), so that
  // the loads of the unit in flight never wait for it.
  auto modes_of_unit = [&](int64_t unit) -> uint32_t {
    if constexpr (ALL_LONG) return 0u;
    else return reinterpret_cast<const uint32_t *>(L.side + unit * kSideBytes)[13];
  };
  auto clamped = [&](int64_t pos) -> int64_t { return pos < units_total ? pos : units_total - 1; };
  int64_t unit_cur = unit_at(u_first), unit_nxt = unit_at(clamped(u_first + stride));
  PackHeader hA = pack_load_header<SPEC>(L, unit_cur, lane), hB;
  float xA[8], xB[8];
  load_coefs(unit_cur, modes_of_unit(unit_cur), xA);
  uint32_t modes_next = modes_of_unit(unit_nxt);
  auto step = [&](const PackHeader &h0, const float (&x)[8], PackHeader &hn, float (&xn)[8], int64_t pos) {
    const int64_t unit = unit_cur;
    // ---- issue the loads of the unit ahead, and the index of the one after it ----
    hn = pack_load_header<SPEC>(L, unit_nxt, lane);
    load_coefs(unit_nxt, modes_next, xn);
    const int64_t unit_nn = unit_at(clamped(pos + 2 * stride));
    modes_next = modes_of_unit(unit_nn);
    unit_cur = unit_nxt;
    unit_nxt = unit_nn;
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
    const int scan = wave_inclusive_scan(mybits);
    const float eb = SPEC ? __shfl(h0.eps, lane >= 36 ? 2 : (lane >= 20 ? 1 : 0)) : 0.0f;
    if (lane < 52) {
      const int mode = lane >= 36 ? m2 : (lane >= 20 ? m1 : m0);
      if constexpr (SPEC) {
        const float nf = (sf != 0 && bits_b != 0) ? norm_s[sf * 16 + wl] : 0.0f;
        const float g = eb * nf;
        S.rec[lane] = make_uint4((uint32_t)bits_b | ((uint32_t)(16 + 10 * n + scan - mybits) << 5), __float_as_uint(nf),
                                 __float_as_uint(__builtin_fmaf(g, 9.5367431640625e-07f, g) + 2.384185791015625e-07f),
                                 __float_as_uint((float)((1 << (bits_b > 0 ? bits_b - 1 : 0)) - 1) + 0.5f));
      } else {
        S.desc[lane] = (uint32_t)bits_b | ((uint32_t)(16 + 10 * n + scan - mybits) << 5) | ((uint32_t)(mode == 0 ? my_long : my_short) << 16);
        S.normd[lane] = (sf != 0 && bits_b != 0) ? norm_s[sf * 16 + wl] : 0.0;
      }
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
    bool doubtful = false;
    if constexpr (SPEC) {
      // Branch-free: a lane's eight slots are consecutive in the bit stream, so its cursor is the position of slot 0.
      // Two mantissas (<= 32 bits) are appended to a 64-bit accumulator that holds < 32 bits, then at most one
      // 32-bit group is complete; it is OR-ed into the unit (a zero when none is: every lane issues the same five
      // LDS operations, groups shared with a neighbour need the OR anyway).
      // Quantization: |x| * norm + 0.5 in one fused operation; the reference's value of it lies within
      // et = guard + 2^-22 a of a, so the truncation is certain when fract(a) is in (et, 1 - et).
      // The sign rides on the conversion (trunc(copysign(a, x)) = sign(x) trunc(a)), the clamp is one median of three.
      // Guard: the truncation is certain when et < fract(a) < 1 - et, i.e. |fract(a) - 1/2| + et < 1/2; the two
      // roundings of that sum are worth 2^-24 at most, the comparison below gives away 2^-23.
      const uint32_t d0 = S.rec[slot_b[0]].x;
      const int pos0 = (int)((d0 >> 5) & 0x7ff) + slot_j[0] * (int)(d0 & 31);
      int cnt = pos0 & 31, wi = pos0 >> 5;
      uint64_t acc = 0;
      uint32_t worst = 0u;                                       // largest |fract - 1/2| + et of the lane, as a bit pattern: for values
                                                                 // >= 0 the unsigned order is the numeric one, and infinities and NaNs
                                                                 // (coefficients that are not finite) come out on top instead of being dropped
#pragma unroll
      for (int p = 0; p < 4; p++) {
        float tt[2];
#pragma unroll
        for (int m = 2 * p; m < 2 * p + 2; m++) {
          const uint4 r = S.rec[slot_b[m]];
          const int bits = r.x & 31;
          const float a = __builtin_fmaf(fabsf(x[m]), __uint_as_float(r.y), 0.5f);
          const float d = __builtin_amdgcn_fractf(a);
          const float et = __builtin_fmaf(a, 2.384185791015625e-07f, __uint_as_float(r.z));
          tt[m - 2 * p] = fabsf(d - 0.5f) + et;
          // the clamp to +-range (quantization.js:49-53) before the conversion: a >= 1/2, and trunc(min(a, range + 1/2)) =
          // min(trunc(a), range) for every finite a (range + 1/2 is a binary32 number); the sign rides on the conversion.
          // (a NaN comes out as +-range instead of 0: its unit is on the redo list anyway, `worst` above)
          const int32_t qc = (int32_t)__builtin_copysignf(fminf(a, __uint_as_float(r.w)), x[m]);
          const uint32_t v = __builtin_amdgcn_ubfe((uint32_t)qc, 0u, (uint32_t)bits);
          acc = (acc << bits) | v;
          cnt += bits;
        }
        worst = max(worst, max(__float_as_uint(tt[0]), __float_as_uint(tt[1])));
        const bool full = cnt >= 32;
        const uint32_t w = full ? (uint32_t)(acc >> ((cnt - 32) & 31)) : 0u;
        if (w != 0u) atomicOr(&S.words[wi], w);                 // lanes with nothing to add stay out: atomics of many lanes on one word serialise
        cnt = full ? cnt - 32 : cnt;
        wi += full ? 1 : 0;
      }
      { const uint32_t w = (uint32_t)(acc << ((32 - cnt) & 63)); if (w != 0u) atomicOr(&S.words[wi], w); }   // the low cnt bits are the unwritten ones (cnt = 0: nothing)
      doubtful = !(worst < 0x3EFFFFFCu);                          // 0.49999988 = 1/2 - 2^-23
    } else {
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
      const int32_t range = (1 << (bits > 0 ? bits - 1 : 0)) - 1;
      int32_t y;
      {
        const double xs = (double)x[m] * S.normd[slot_b[m]];
        const double v = xs + (xs >= 0 ? 0.5 : -0.5);            // round half away from zero ...
        y = (int32_t)v;                                          // ... then `| 0`: truncation; exact wrap below
        if (__builtin_expect(!(fabs(v) < 2147483648.0), 0)) y = to_int32(v);
        y = y > range ? range : (y < -range ? -range : y);
      }
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
    }
    wave_sync();
    if (lane < 53) reinterpret_cast<uint32_t *>(L.units + unit * C1_UNIT_BYTES)[lane] = __builtin_bswap32(S.words[lane]);
    if constexpr (SPEC) {
      // flag word of the analysis (a scale-factor index was not certain) or any doubtful mantissa -> exact redo
      const uint32_t epsw = __float_as_uint(h0.eps);
      const bool sf_open = (__builtin_amdgcn_readlane((int)epsw, 3) & (int)kEpsFlagSfOpen) != 0;
      const bool redo = __builtin_amdgcn_ballot_w64(doubtful) != 0 || sf_open;
      if (redo) {
        // bounds of zero (not even the 2^-70 every speculative bound contains): the exact kernels' coefficients
        const bool exact = (__builtin_amdgcn_readlane((int)epsw, 0) | __builtin_amdgcn_readlane((int)epsw, 1) | __builtin_amdgcn_readlane((int)epsw, 2)) == 0;
        if (lane == 0) S.redo[n_redo] = (uint32_t)unit | (sf_open ? 0x80000000u : 0u) | (exact ? 0x40000000u : 0u);
        n_redo++;
      }
    }
    wave_sync();
    if constexpr (SPEC) { if (n_redo == kRedoBatch) flush_redo(); }
  };
  for (int64_t pos = u_first; pos < units_total; pos += 2 * stride) {
    step(hA, xA, hB, xB, pos);
    if (pos + stride < units_total) step(hB, xB, hA, xA, pos + stride);
  }
  if constexpr (SPEC) { if (n_redo > 0) flush_redo(); }
}

}  // namespace

template <class Kernel>
static int pack_resident_blocks(Kernel kernel) {
  int per_cu = 0, cus = 0, dev = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, C1_WAVE * kPackWaves, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  return per_cu * cus;
}
template <bool ALL_LONG, bool SPEC>
static void launch_pack(const C1EncodeLaunch &L, hipStream_t stream) {
  // the all-long speculative instantiation (5 waves per SIMD, the headline's): exactly the resident workgroups; the others
  // (4 per SIMD) measured no better that way than with the two rounds they had
  static const int resident = pack_resident_blocks(k_pack<ALL_LONG, SPEC>) * ((ALL_LONG && SPEC) ? 1 : 2);
  const dim3 grid((unsigned)std::min<int64_t>(resident, (L.frames * L.channels + kPackWaves - 1) / kPackWaves)), block(C1_WAVE * kPackWaves);
  hipLaunchKernelGGL((k_pack<ALL_LONG, SPEC>), grid, block, 0, stream, L);
}
void c1k_launch_pack(const C1EncodeLaunch &L, bool all_long, hipStream_t stream) {
  if (all_long) launch_pack<true, false>(L, stream); else launch_pack<false, false>(L, stream);
}
void c1k_launch_pack_spec(const C1EncodeLaunch &L, bool all_long, hipStream_t stream) {
  if (all_long) launch_pack<true, true>(L, stream); else launch_pack<false, true>(L, stream);
}
