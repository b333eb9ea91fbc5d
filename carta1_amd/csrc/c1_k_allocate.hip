// c1_k_allocate.hip -- allocateBits (bitallocation.js:74-142): the greedy RDO heaps, one lane per heap
#include "c1_device.h"

namespace {

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
//   k_alloc_bound   one lane per unit that still has candidates alive: a second, much sharper lower bound for each of
//                   them (Lagrangian relaxation of the greedy's knapsack, below), drops what that already excludes
//                   and sends the most promising candidate of the unit to the first round of heaps.
//   k_alloc_rest    one lane per work-list entry (unit, candidate): the same heap run.  First round: one entry per
//                   unit; the lane then sends the unit's other candidates whose bound does not exceed the best total
//                   known now (52 BFUs or this one) to the second round; the others can never win.
//   k_alloc_select  one lane per unit: smallest total, smallest count on ties (:116-129), or the
//                   fallback when no total is finite (:132-139).
// On white noise 97 % of the units stop after k_alloc_first; stationary tonal material keeps all eight candidates
// alive under the first bound (its upper BFUs are nearly silent, so dropping them costs nothing) and ran 8 heaps per
// unit; with the second bound it runs 2.
//
// Heap entry (one 32-bit word, bit 31 clear):  rank(10) | size(5) | sfi(6) | wl(4) | bfu(6).
// `rank` orders the Float32 priorities biasedSF[sfi]*DISTORTION_DELTA_FACTORS[wl]/WORD_LENGTH_DELTA_BITS[wl]
// (bitallocation.js:226-231,267-269) exactly: equal priorities have equal rank, so the strict `>` of
// siftDown (:325-331) -- and with it the tie order -- is reproduced.  With kLow = the 21 payload bits,
// rank(a) > rank(b)  <=>  a > (b | kLow): one integer compare, no field extraction.  Rank 0 never occurs
// in a live entry, so unused slots (kSentinel: rank 0) and parked entries (below) act as sentinels: no bounds checks in the sift.
//
// Heaps live in LDS as heap[slot][lane] (64 dwords per slot: conflict-free, the two children of a node one
// ds_read2st64 apart).  An entry that leaves the heap is parked, rank cleared, in the slot the shrinking
// heap frees, so when the loop ends slots [0, initial size) hold every BFU with its final word length.
constexpr int kHeapSlotsPerLane = 52 + 1;   // + one sentinel slot behind the last (13 568 bytes per wave: 12 waves per CU; 54 rows leave 11)
constexpr uint32_t kLow = 0x1FFFFFu;
constexpr uint32_t kSentinel = 52u;           // rank 0, word length 0, "BFU 52": the gather behind the heap run sends it to the spare slot
constexpr int kCandBytes = kCandidateBytes;  // per unit: 8 totals (double) + 8 x 32-byte results + 8 lower bounds
constexpr double kAlive = -1.0;              // total of a candidate that may still win and has not been evaluated (real totals are >= 0)

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
    // children 2i+1, 2i+2; a node without children reads (slot 51, sentinel) and its left value is masked
    const int l = 2 * i + 1;
    const uint32_t *src = hp + min(l, sentinel - 1) * 64;
    el = l < sentinel ? src[0] : 0u;
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
  // (an LDS copy of this table is slower: 512 more bytes per wave cost a wave per CU, measured 1.50 -> 1.63 ms)
  const __attribute__((address_space(4))) double *biased = (const __attribute__((address_space(4))) double *)O->biased;
  const bool affine = O->rank_affine != 0;
  const int ka = O->rank_a, kb = O->rank_b, kc = O->rank_c, koff = O->rank_off;
  // (24-bit multiplies: every factor is below 64 in magnitude, and the full 32-bit v_mul_lo_u32 issues at a quarter of the rate)
  const int nkb = -kb, kc_off = kc + koff;
  auto rank_of = [&](int s, int wl) -> uint32_t {
    if (affine) return (uint32_t)(wl == 0 ? __mul24(ka, s) + kc_off : __mul24(ka, s) + koff + __mul24(nkb, wl + 1));
    return rank_t[s * 16 + (wl & 15)];
  };
  int remaining = 212 * 8 - 40 - 10 * n;                   // bitallocation.js:97-100
  int hs = 0;
  // distributeBitsRDO (:203-281): initial heap = BFUs below n with a non-zero scale factor
  // branch-free: every BFU writes its entry at the cursor, only live ones advance it (the next one overwrites the
  // slot otherwise; what the last dead one leaves behind is cleared with the sentinels)
#pragma unroll
  for (int b = 0; b < 52; b++) {
    if (b < n) {
      const int s = (sf[b >> 2] >> ((b & 3) * 8)) & 63;
      hp[hs * 64] = heap_entry(rank_of(s, 0), kSpecs[b], s, 0, b);
      hs += (s != 0 && live) ? 1 : 0;
    }
  }
  for (int k = hs; k < kHeapSlotsPerLane; k++) hp[k * 64] = kSentinel;   // rank 0: below every live entry
  for (int i = (hs >> 1) - 1; i >= 0; i--) {               // heapify (:238-241); per-lane trip counts
    const int l = 2 * i + 1;
    heap_sift_down(hp, 52, i, hp[i * 64], hp[l * 64], hp[(l + 1) * 64], true);
  }
  // greedy spending loop (:244-278): the root either takes its next priority or leaves the heap (does not
  // fit :251-258, or reached the last word length :271-277); then one sift.
  uint32_t r0 = hp[0], r1 = hp[64], r2 = hp[128];
  // the smallest BFU has 4 coefficients, so with fewer than 4 bits left no entry can take its next step: the
  // remaining iterations of the reference only pop entries that do not fit and change nothing (:251-258)
  bool run = remaining >= 4 && hs > 0;
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
    {
      // level 1: the chosen child's children are slots 3,4 or 5,6.  (No "does any lane still move?" test in front of
      // levels 1 and 2: among 64 heaps one nearly always does, and the test costs more than the level it would skip.)
      const int i1 = tr0 ? 2 : 1;
      const uint32_t *src = hp + (2 * i1 + 1) * 64;
      const uint32_t el = src[0], er = src[64];
      // both pairs of grandchildren (slots 4 i1 + 3 .. 4 i1 + 6 <= 14: always inside the lane's slots) are requested with the
      // children: one LDS round trip decides two levels -- the loop is a chain of dependent round trips, not of instructions
      const uint32_t g0 = src[(2 * i1 + 2) * 64], g1 = src[(2 * i1 + 3) * 64], g2 = src[(2 * i1 + 4) * 64], g3 = src[(2 * i1 + 5) * 64];
      const bool tr1 = er > max(el | kLow, vmax);
      const bool tl1 = !tr1 && el > vmax;
      const bool mv1 = mv0 && (tr1 || tl1);
      const uint32_t val = mv1 ? (tr1 ? er : el) : v;
      if (mv0) { if (tr0) r2 = val; else r1 = val; }
      // levels 2..5 unrolled (a heap of 52 has six): node i2 in slots 3..6 against its children (the grandchildren above),
      // then ONE more round trip for the children (slots 15..30) and grandchildren (31..62, clamped to the sentinel slot,
      // whose rank 0 never wins) of the node in slots 7..14 it moves to.  Two round trips per step where the loop of
      // heap_sift_down made up to five (one per level and one more to find that the last level has no children); a
      // lane that stops early just stops writing.
      {
        const int i2 = 2 * i1 + 1 + (tr1 ? 1 : 0);
        const uint32_t c0 = tr1 ? g2 : g0, c1 = tr1 ? g3 : g1;
        const bool tr2 = c1 > max(c0 | kLow, vmax);
        const bool tl2 = !tr2 && c0 > vmax;
        const bool mv2 = mv1 && (tr2 || tl2);
        if (mv1) hp[i2 * 64] = mv2 ? (tr2 ? c1 : c0) : v;
        if (__builtin_amdgcn_ballot_w64(mv2) != 0) {
          const int i3 = 2 * i2 + 1 + (tr2 ? 1 : 0);                  // 7..14
          const uint32_t *s3 = hp + (2 * i3 + 1) * 64;                // children: slots 15..30
          const uint32_t d0 = s3[0], d1 = s3[64];
          const int q = 4 * i3 + 3;                                   // grandchildren: slots 31..62
          const uint32_t h0 = hp[min(q, 52) * 64], h1 = hp[min(q + 1, 52) * 64], h2 = hp[min(q + 2, 52) * 64], h3 = hp[min(q + 3, 52) * 64];
          const bool tr3 = d1 > max(d0 | kLow, vmax);
          const bool tl3 = !tr3 && d0 > vmax;
          const bool mv3 = mv2 && (tr3 || tl3);
          if (mv2) hp[i3 * 64] = mv3 ? (tr3 ? d1 : d0) : v;
          const int i4 = 2 * i3 + 1 + (tr3 ? 1 : 0);                  // 15..30
          const uint32_t e0 = tr3 ? h2 : h0, e1 = tr3 ? h3 : h1;
          const bool tr4 = e1 > max(e0 | kLow, vmax);
          const bool tl4 = !tr4 && e0 > vmax;
          const bool mv4 = mv3 && (tr4 || tl4);
          if (mv3) hp[i4 * 64] = mv4 ? (tr4 ? e1 : e0) : v;
          if (mv4) hp[(2 * i4 + 1 + (tr4 ? 1 : 0)) * 64] = v;          // slots 31..51: no children
        }
      }
    }
    run = run && remaining >= 4 && hs > 0;
  }
  hp[0] = r0; hp[64] = r1; hp[128] = r2;
  // Every BFU that ever entered the heap now sits in one of the first slots (as many as the heap started with) with its final
  // word length; the slots behind hold the sentinel, which names slot 52.  The word lengths are put in BFU order through the lane's own column of the heap: the
  // top byte of slot b takes BFU b's word length (a byte store: the rank bits up there are done with, and the low bits
  // that name an unread entry's BFU and word length stay as they are), then the 52 top bytes are read back in order and
  // packed with constant shifts.  6 vector instructions per BFU; selecting one of four 64-bit words by a lane-varying
  // index cost 25 (1 300 of the 8 300 a wave of k_alloc_first issued).
  {
    uint8_t *hb = reinterpret_cast<uint8_t *>(hp);
#pragma unroll
    for (int k = 0; k < 52; k++) hb[k * 256 + 3] = 0;
#pragma unroll
    for (int k0 = 0; k0 < 52; k0 += 8) {
      uint32_t e[8];
#pragma unroll
      for (int j = 0; j < 8; j++) if (k0 + j < 52) e[j] = hp[(k0 + j) * 64];
#pragma unroll
      for (int j = 0; j < 8; j++) if (k0 + j < 52) hb[(e[j] & 63u) * 256 + 3] = (uint8_t)((e[j] >> 6) & 15u);
    }
    uint32_t d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      d[i] = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) if (8 * i + j < 52) d[i] |= (hp[(8 * i + j) * 64] >> 24) << (4 * j);
    }
    res0 = (uint64_t)d[0] | ((uint64_t)d[1] << 32); res1 = (uint64_t)d[2] | ((uint64_t)d[3] << 32);
    res2 = (uint64_t)d[4] | ((uint64_t)d[5] << 32); res3 = (uint64_t)d[6] | ((uint64_t)d[7] << 32);
  }
  // calculateTotalDistortion (:157-190): sequential double sum, index ascending
  total = 0.0;
#pragma unroll
  for (int b = 0; b < 52; b++) {
    const int s = (sf[b >> 2] >> ((b & 3) * 8)) & 63;
    const uint64_t word = b < 16 ? res0 : (b < 32 ? res1 : (b < 48 ? res2 : res3));
    const int wl = (int)((word >> ((b & 15) * 4)) & 15);
    const int size = kSpecs[b];
    // the table value is loaded whatever s is (index 0 is a valid entry) and masked afterwards: a lane-varying table read is
    // a cache round trip, and under a condition each of the 52 is waited for where it is issued, one after the other;
    // unconditional, the compiler requests them in batches (tools/isa_waits.py)
    const double bs = biased[s];
    if (b >= n || wl == 0) {
      // zeroBitDistortions[b] = Float32(biasedSF * 2 * size), 0 when sfi == 0 (:76,87-89)
      total += s != 0 ? (double)f32(bs * 2.0 * (double)size) : 0.0;
    } else if (s != 0) {
      const double ip2 = __hiloint2double((1023 - wl_bits(wl)) << 20, 0);   // INV_POWER_OF_TWO[bits] = 2^-bits
      total += bs * ip2 * (double)size;
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

__device__ __forceinline__ bool getenv_no_tonal(const C1DevEncOpts *O) { return O->alloc_no_tonal != 0; }   // C1_ALLOC_NO_TONAL=1 (experiments)

__global__ __launch_bounds__(C1_WAVE, 3) void k_alloc_first(C1EncodeLaunch L) {
  __shared__ uint32_t heap[kHeapSlotsPerLane * 64];
  const C1DevEncOpts *O = L.opts;
  const __attribute__((address_space(4))) double *biased = (const __attribute__((address_space(4))) double *)O->biased;
  const int lane = threadIdx.x;
  // list mode (exact redo after the speculative pass): positions index L.unit_list
  const bool listed = L.unit_list != nullptr;
  const int64_t units_total = listed ? (int64_t)*L.unit_count : L.frames * L.channels;
  for (int64_t pos0 = (int64_t)blockIdx.x * 64; pos0 < units_total; pos0 += (int64_t)gridDim.x * 64) {
  const int64_t pos = pos0 + lane;
  const bool live = pos < units_total;
  const int64_t unit = listed ? (int64_t)L.unit_list[live ? pos : 0] : pos;
  uint32_t sf[13];
  load_sfi(L.side, live ? unit : 0, sf);
  // Which candidate to run first?  Where the upper BFUs carry next to nothing (tones, low-passed material) dropping
  // them is free, a smaller candidate wins, and the 52-BFU heap would be run for nothing: such units go to
  // k_alloc_bound with all eight candidates open.  The test is a guess (it only chooses the order of work, every
  // path ends in the same exact comparison): the continuous relaxation of the 52-BFU problem spends its 1 136 bits at
  // log2 lambda = (sum size_b log2 biasedSF_b - 1136) / N over the N coefficients of non-silent BFUs and then has a
  // total of about N lambda / ln 2 (measured: the 52-BFU total is 0.7-1.0 of that estimate).  Units whose 52-BFU
  // candidate wins have a zero-bit distortion t6 of the top four BFUs above 0.44 of the estimate (5 % quantile over
  // white, pink and mixed material), units where a smaller candidate wins mostly below 0.4 (tools/alloc_tonal_stats.py).
  bool tonal = false;
  {
    float n_coef = 0.0f, s_la = 0.0f;
#pragma unroll
    for (int b = 0; b < 52; b++) {
      const int s = (sf[b >> 2] >> ((b & 3) * 8)) & 63;
      const float w = s != 0 ? (float)kSpecs[b] : 0.0f;
      n_coef += w;
      s_la = __builtin_fmaf(w, (float)s, s_la);
    }
    // sum size la = la_slope * sum size s + la_off * N
    const float la_sum = __builtin_fmaf(O->la_slope, s_la, O->la_off * n_coef);
    const float t_est = 1.442695f * n_coef * __builtin_amdgcn_exp2f((la_sum - 1136.0f) / n_coef);
    float t6f = 0.0f;                                       // zero-bit distortion of the top four BFUs (a guess needs no more than binary32)
#pragma unroll
    for (int b = 48; b < 52; b++) {
      const int s = (sf[b >> 2] >> ((b & 3) * 8)) & 63;
      const float bs = (float)biased[s];
      t6f += s != 0 ? bs * (2.0f * (float)kSpecs[b]) : 0.0f;
    }
    const bool mine = n_coef > 0.0f && t6f < 0.35f * t_est && !getenv_no_tonal(O);
    // one decision per wave (the majority's): a heap loop runs as long as any lane of the wave needs it, so skipping
    // it pays only when every lane does
    tonal = 2 * __popcll(__builtin_amdgcn_ballot_w64(live && mine)) > __popcll(__builtin_amdgcn_ballot_w64(live));
  }
  uint64_t r0, r1, r2, r3;
  double total;
  run_candidate(heap + lane, 52, sf, O, live && !tonal, r0, r1, r2, r3, total);
  // lower bounds of the other seven candidates: zero-bit distortion of the BFUs they do not code, each
  // summed from its first uncoded BFU upwards (one pass, seven running sums)
  uint32_t survivors = 0;
  {
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
#pragma unroll
    for (int b = 20; b < 52; b++) {
      const int s = (sf[b >> 2] >> ((b & 3) * 8)) & 63;
      const double bs = biased[s];                                  // unconditional: see run_candidate
      const double z = s != 0 ? (double)f32(bs * 2.0 * (double)kSpecs[b]) : 0.0;
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
  if (tonal) survivors = 0xffu;                               // nothing evaluated, everything open (bit 7 = the 52-BFU candidate)
  const bool finite52 = total < __builtin_huge_val();
  if (live && survivors == 0) {
    // nothing else can win: the 52-BFU candidate is the allocation (or the fallback when its total is not finite,
    // bitallocation.js:132-139); no candidate record, no selection pass for this unit
    uint64_t *dst = reinterpret_cast<uint64_t *>(L.alloc + unit * kAllocBytes);
    if (finite52) { dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3 | (7ull << 60); }
    else { dst[0] = 0; dst[1] = 0; dst[2] = 0; dst[3] = 1ull << 59; }
  } else if (live) {
    // totals of the other candidates: +inf = cannot win, kAlive = still possible and not evaluated yet
    uint8_t *base = L.cand + unit * kCandBytes;
    double *tot = reinterpret_cast<double *>(base);
#pragma unroll
    for (int c = 0; c < 7; c++) tot[c] = ((survivors >> c) & 1u) ? kAlive : __builtin_huge_val();
    if (tonal) tot[7] = kAlive;
    else store_candidate(L.cand, unit, 7, finite52 ? total : __builtin_huge_val(), r0, r1, r2, r3);
  }
  // append the unit to the selection list: one atomic per wave
  const uint64_t sel_mask = __builtin_amdgcn_ballot_w64(live && survivors != 0);
  uint32_t sel_base = 0;
  if (lane == 0 && sel_mask != 0) sel_base = atomicAdd(L.work_count + 1, (uint32_t)__popcll(sel_mask));
  sel_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)sel_base);
  if (live && survivors != 0) L.sel_list[sel_base + (uint32_t)__popcll(sel_mask & ((1ull << lane) - 1ull))] = (uint32_t)unit;
  }
}

// =====================================================================================================
// The second bound.  For a candidate that codes n BFUs with B = 1656 - 10 n bits, ANY word lengths wl_b that fit the
// budget (sum size_b bits(wl_b) <= B) -- in particular the ones the greedy heap ends with -- satisfy, for every
// multiplier lambda >= 0,
//     sum_{b<n} D_b(wl_b)  >=  sum_{b<n} [D_b(wl_b) + lambda size_b bits(wl_b)] - lambda B
//                          >=  sum_{b<n} min_wl [D_b(wl) + lambda size_b bits(wl)] - lambda B,
// with D_b(0) = zeroBitDistortions[b] and D_b(wl) = biasedSF * 2^-bits * size the very terms calculateTotalDistortion adds
// (bitallocation.js:157-190).  Adding the zero-bit terms of the BFUs the candidate does not code gives a lower bound
// on its total for every lambda; it is sharpest near the multiplier at which the relaxed problem spends exactly B bits,
// found by bisection on log2 lambda (the relaxed spend is a decreasing step function of it).  The inner minimum is
// explicit: D + lambda size bits is convex in bits >= 2 (each further bit gains biasedSF 2^-(bits+1) per coefficient),
// so its minimum over 2..16 bits sits at clamp(floor(log2(biasedSF / lambda)), 2, 16); the zero-bit term competes
// separately because it is stored as a Float32.  Measured against the totals the heaps produce the bound is within
// 2-4 % (tools/alloc_bound_sim.py).
// Rounding: the reference's total is a sequential sum of at most 52 non-negative doubles, so it is at least
// (1 - 51 * 2^-53) times the exact sum; our P = sum of minima and M = lambda B carry a few roundings each.  The bound
// handed on is (P - M) - 2^-44 (P + M): three orders of magnitude more slack than all of that, and still 1e-13 of the
// totals it is compared with.  A candidate is dropped only when this bound is strictly above a total that was really
// computed, i.e. when it cannot even tie (:116-129 keeps the earlier candidate on ties).
// =====================================================================================================
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
constexpr int kSpecsHost[52] = {8, 8, 8, 8, 4, 4, 4, 4, 8, 8, 8, 8, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 7, 7, 7, 7, 9, 9, 9, 9, 10, 10, 10, 10,
                                12, 12, 12, 12, 12, 12, 12, 12, 20, 20, 20, 20, 20, 20, 20, 20};   // kSpecs as compile-time constants
__global__ __launch_bounds__(256) void k_alloc_bound(C1EncodeLaunch L) {
  __shared__ double biased_s[64];
  const C1DevEncOpts *O = L.opts;
  // index 0 = silent BFU: with a zero here every term of such a BFU below is zero by itself
  if (threadIdx.x < 64) biased_s[threadIdx.x] = threadIdx.x == 0 ? 0.0 : ((const __attribute__((address_space(4))) double *)O->biased)[threadIdx.x];
  __syncthreads();
  const float la_slope = O->la_slope, la_off = O->la_off;
  const int lane = threadIdx.x & 63;
  const uint32_t count = L.work_count[1];
  for (uint32_t pos0 = blockIdx.x * 256u; pos0 < count; pos0 += gridDim.x * 256u) {
    const uint32_t pos = pos0 + threadIdx.x;
    const bool live = pos < count;
    const int64_t unit = live ? (int64_t)L.sel_list[pos] : (int64_t)L.sel_list[0];
    uint32_t sf[13];
    load_sfi(L.side, unit, sf);
    uint8_t *base = L.cand + unit * kCandBytes;
    double *tot = reinterpret_cast<double *>(base);
    double *lb = reinterpret_cast<double *>(base + kCandLbOffset);
    const double t52 = tot[7];
    const double best = t52 == kAlive ? __builtin_huge_val() : t52;     // the 52-BFU candidate may be open too (k_alloc_first)
    // bracket of log2 lambda for the first candidate: from "every BFU at 16 bits" to "nothing coded"
    int smin = 64, smax = 0;
#pragma unroll
    for (int b = 0; b < 52; b++) {
      const int s = (sf[b >> 2] >> ((b & 3) * 8)) & 63;
      if (s != 0) { smin = s < smin ? s : smin; smax = s > smax ? s : smax; }
    }
    // The multiplier search needs no precision (every lambda gives a valid bound; a poor one only prunes less), so it
    // runs on packed halves: two BFUs per instruction, the spend accumulated by v_dot2_f32_f16.  The relaxed word
    // length floor(y) is replaced by its mean y - 1/2 (clamped to 2..16 bits, nothing below the point where two bits
    // stop paying); silent BFUs sit far below everything.
    h2 sp[26];
#pragma unroll
    for (int k = 0; k < 26; k++) {
      const int s0 = (sf[(2 * k) >> 2] >> (((2 * k) & 3) * 8)) & 63, s1 = (sf[(2 * k + 1) >> 2] >> (((2 * k + 1) & 3) * 8)) & 63;
      sp[k] = h2{s0 != 0 ? (_Float16)s0 : (_Float16)-1000.0f, s1 != 0 ? (_Float16)s1 : (_Float16)-1000.0f};
    }
    const h2 slope2 = h2{(_Float16)la_slope, (_Float16)la_slope};
    float x_prev = 0.0f;
    bool have_prev = false;
    int c_star = -1;
    double lb_star = __builtin_huge_val();
    for (int c = 0; c < 8; c++) {
      const bool mine = live && tot[c] == kAlive;
      if (__builtin_amdgcn_ballot_w64(mine) == 0) continue;
      const int n = bfu_amount(c);
      const int B = 212 * 8 - 40 - 10 * n;
      float lo = have_prev ? x_prev - 0.05f : __builtin_fmaf(la_slope, (float)smin, la_off) - 17.0f;
      float hi = have_prev ? x_prev + 3.15f : __builtin_fmaf(la_slope, (float)smax, la_off);
      const int iters = __builtin_amdgcn_ballot_w64(mine && !have_prev) != 0 ? 8 : 5;
      for (int it = 0; it < iters; it++) {
        const float x = 0.5f * (lo + hi);
        const _Float16 by = (_Float16)(la_off - x - 0.5f);
        const h2 base2 = h2{by, by};
        float used = 0.0f;
#pragma unroll
        for (int k = 0; k < 26; k++) {
          if (2 * k < n) {
            const h2 y = __builtin_elementwise_fma(sp[k], slope2, base2);                     // log2(biasedSF / lambda) - 1/2
            const h2 bits = __builtin_elementwise_min(__builtin_elementwise_max(y, h2{(_Float16)2.0f, (_Float16)2.0f}), h2{(_Float16)16.0f, (_Float16)16.0f});
            const h2 ramp = __builtin_elementwise_fma(y, h2{(_Float16)64.0f, (_Float16)64.0f}, h2{(_Float16)20.0f, (_Float16)20.0f});   // 0 below y = -0.31, 1 above -0.30
            const h2 on = __builtin_elementwise_min(__builtin_elementwise_max(ramp, h2{(_Float16)0.0f, (_Float16)0.0f}), h2{(_Float16)1.0f, (_Float16)1.0f});
            used = __builtin_amdgcn_fdot2(bits * on, h2{(_Float16)(float)kSpecsHost[2 * k], (_Float16)(float)kSpecsHost[2 * k + 1]}, used, false);
          }
        }
        if (used > (float)B) lo = x; else hi = x;
      }
      const float x = hi;
      const double lambda = (double)__builtin_amdgcn_exp2f(x);
      const double inv_lambda = 1.0 / lambda;
      double P = 0.0;
#pragma unroll
      for (int b = 0; b < 52; b++) {
        const int s = (sf[b >> 2] >> ((b & 3) * 8)) & 63;
        const double a = biased_s[s], size = (double)kSpecs[b];
        const double z = (double)f32(a * 2.0 * size);                 // zeroBitDistortions[b] (:87-89)
        double g = z;
        if (b < n) {
          const double q = a * inv_lambda;
          int e = ((__double2hiint(q) >> 20) & 0x7ff) - 1023;         // floor(log2 q) for normal q
          e = e < 2 ? 2 : (e > 16 ? 16 : e);
          const double ip2 = __hiloint2double((1023 - e) << 20, 0);
          const double h = a * ip2 * size + lambda * size * (double)e;   // (:183-187) + the price of e bits
          g = h < z ? h : z;
        }
        P += g;
      }
      const double M = lambda * (double)B;
      const double bound = (P - M) - 5.684341886080802e-14 * (P + M);
      if (mine) {
        if (bound > best) tot[c] = __builtin_huge_val();               // cannot win, cannot tie
        else {
          lb[c] = bound;
          if (bound < lb_star || c_star < 0) { lb_star = bound; c_star = c; }
        }
        x_prev = x; have_prev = true;
      }
    }
    // first round of heaps: the candidate with the smallest bound
    const bool has = live && c_star >= 0;
    const uint64_t m = __builtin_amdgcn_ballot_w64(has);
    uint32_t at = 0;
    if (lane == 0 && m != 0) at = atomicAdd(L.work_count, (uint32_t)__popcll(m));
    at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
    if (has) L.work_list[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)unit << 3) | (uint32_t)c_star;
  }
}

// FIRST_ROUND: every entry is the most promising candidate of its unit (one entry per unit).  With its total known,
// the unit's other live candidates are settled on the spot: a bound above the best total known now (52 BFUs or this
// one) can never win, the rest is appended to the second round's list.
template <bool FIRST_ROUND>
__global__ __launch_bounds__(C1_WAVE, 3) void k_alloc_rest(C1EncodeLaunch L, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_count,
                                                           uint32_t *__restrict__ list2) {
  __shared__ uint32_t heap[kHeapSlotsPerLane * 64];
  const C1DevEncOpts *O = L.opts;
  const int lane = threadIdx.x;
  const uint32_t count = *list_count;
  for (uint32_t base = blockIdx.x * 64u; base < count; base += gridDim.x * 64u) {
    const uint32_t idx = base + lane;
    const bool live = idx < count;
    const uint32_t item = live ? list[idx] : 0u;
    const int64_t unit = item >> 3;
    const int c = item & 7;
    uint32_t sf[13];
    load_sfi(L.side, unit, sf);
    uint64_t r0, r1, r2, r3;
    double total;
    run_candidate(heap + lane, bfu_amount(c), sf, O, live, r0, r1, r2, r3, total);
    total = total < __builtin_huge_val() ? total : __builtin_huge_val();
    if (live) store_candidate(L.cand, unit, c, total, r0, r1, r2, r3);
    if constexpr (FIRST_ROUND) {
      uint8_t *cbase = L.cand + unit * kCandBytes;
      double *tot = reinterpret_cast<double *>(cbase);
      const double *lb = reinterpret_cast<const double *>(cbase + kCandLbOffset);
      uint32_t keep = 0;
      if (live) {
        const double t52 = tot[7];
        const double best = (c == 7 || t52 == kAlive || total < t52) ? total : t52;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          if (k != c && tot[k] == kAlive) {
            if (lb[k] > best) tot[k] = __builtin_huge_val();
            else keep |= 1u << k;
          }
        }
      }
      const int mine = __popc(keep);
      const int scan = wave_inclusive_scan(mine);
      const int wave_total = __builtin_amdgcn_readlane(scan, 63);
      uint32_t at = 0;
      if (lane == 0 && wave_total > 0) at = atomicAdd(L.work_count + 2, (uint32_t)wave_total);
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at) + (uint32_t)(scan - mine);
      for (int k = 0; k < 8; k++)
        if ((keep >> k) & 1u) list2[at++] = ((uint32_t)unit << 3) | (uint32_t)k;
    }
  }
}

// the units on the selection list (those with more than one candidate alive): smallest total, smallest count on ties
__global__ __launch_bounds__(256) void k_alloc_select(C1EncodeLaunch L) {
  const int64_t count = (int64_t)L.work_count[1];
  for (int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x; pos < count; pos += (int64_t)gridDim.x * 256) {
  const int64_t unit = (int64_t)L.sel_list[pos];
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
}

// ---- test tap: every candidate's total next to its lower bound (c1_alloc_bounds_device) -------------------------------
__global__ __launch_bounds__(256) void k_alloc_tap_init(C1EncodeLaunch L, int64_t units) {
  const int64_t unit = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (unit == 0) { L.work_count[0] = 0; L.work_count[1] = (uint32_t)units; L.work_count[2] = (uint32_t)(units * 8); }
  if (unit >= units) return;
  double *tot = reinterpret_cast<double *>(L.cand + unit * kCandBytes);
  for (int c = 0; c < 7; c++) tot[c] = kAlive;            // everything alive, nothing known: the bound kernel drops nothing
  tot[7] = __builtin_huge_val();
  L.sel_list[unit] = (uint32_t)unit;
  for (int c = 0; c < 8; c++) L.work_list[units + unit * 8 + c] = ((uint32_t)unit << 3) | (uint32_t)c;
}
__global__ __launch_bounds__(256) void k_alloc_tap_gather(C1EncodeLaunch L, int64_t units, double *out) {
  const int64_t unit = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (unit >= units) return;
  const double *tot = reinterpret_cast<const double *>(L.cand + unit * kCandBytes);
  const double *lb = reinterpret_cast<const double *>(L.cand + unit * kCandBytes + kCandLbOffset);
  for (int c = 0; c < 8; c++) { out[unit * 16 + c] = tot[c]; out[unit * 16 + 8 + c] = c < 7 ? lb[c] : 0.0; }   // (slot 15 is overwritten below)
  // slot 15: what the production path (c1k_launch_allocate, run before the tap) chose: amount index, or -1 for the fallback
  const uint32_t a7 = reinterpret_cast<const uint32_t *>(L.alloc + unit * kAllocBytes)[7];
  out[unit * 16 + 15] = ((a7 >> 27) & 1u) ? -1.0 : (double)((a7 >> 28) & 7u);
}

}  // namespace

// out: units x 16 doubles: totals of the eight candidates (calculateTotalDistortion after distributeBitsRDO), then the
// seven lower bounds k_alloc_bound gives them, then the candidate the production path chose.  Needs a work list of
// 9 entries per unit.
void c1k_launch_alloc_tap(const C1EncodeLaunch &L, double *out, hipStream_t stream) {
  const int64_t units = L.frames * L.channels;
  c1k_launch_allocate(L, stream);                           // the pruned path, for slot 15
  const dim3 grid((unsigned)((units + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_alloc_tap_init, grid, block, 0, stream, L, units);
  hipLaunchKernelGGL(k_alloc_bound, dim3((unsigned)std::min<int64_t>((units + 255) / 256, 2048)), block, 0, stream, L);
  hipLaunchKernelGGL((k_alloc_rest<false>), dim3((unsigned)std::min<int64_t>((units * 8 + 63) / 64, 256 * 10)), dim3(C1_WAVE), 0, stream, L,
                     (const uint32_t *)(L.work_list + units), (const uint32_t *)(L.work_count + 2), (uint32_t *)nullptr);
  hipLaunchKernelGGL(k_alloc_tap_gather, grid, block, 0, stream, L, units, out);
}

namespace {
}  // namespace

void c1k_launch_allocate(const C1EncodeLaunch &L, hipStream_t stream) {
  const int64_t units = L.frames * L.channels;
  const bool listed = L.unit_list != nullptr;            // the list's length is only known on the device: bounded grids stride over it
  (void)hipMemsetAsync(L.work_count, 0, 3 * sizeof(uint32_t), stream);
  const int64_t first_blocks = listed ? std::min<int64_t>((units + 63) / 64, 256 * 12) : (units + 63) / 64;
  hipLaunchKernelGGL(k_alloc_first, dim3((unsigned)first_blocks), dim3(C1_WAVE), 0, stream, L);
  // every kernel below strides over a list whose length only the device knows
  const int64_t unit_blocks = std::min<int64_t>((units + 255) / 256, 2048);
  uint32_t *list2 = L.work_list + units;                   // first round: at most one entry per unit
  hipLaunchKernelGGL(k_alloc_bound, dim3((unsigned)unit_blocks), dim3(256), 0, stream, L);
  const int64_t rest1_blocks = std::min<int64_t>((units + 63) / 64, 256 * 10);
  hipLaunchKernelGGL((k_alloc_rest<true>), dim3((unsigned)rest1_blocks), dim3(C1_WAVE), 0, stream, L, (const uint32_t *)L.work_list, (const uint32_t *)L.work_count, list2);
  const int64_t rest2_blocks = std::min<int64_t>((units * 6 + 63) / 64, 256 * 10);
  hipLaunchKernelGGL((k_alloc_rest<false>), dim3((unsigned)rest2_blocks), dim3(C1_WAVE), 0, stream, L, (const uint32_t *)list2, (const uint32_t *)(L.work_count + 2), (uint32_t *)nullptr);
  const int64_t select_blocks = std::min<int64_t>((units + 255) / 256, 1024);      // strides over the selection list
  hipLaunchKernelGGL(k_alloc_select, dim3((unsigned)select_blocks), dim3(256), 0, stream, L);
}
