// c1_internal.h -- shared between the HIP kernels (c1_k_*.hip, c1_device.h) and the C-ABI host side
// (c1_api.hip).  Not part of the public boundary (that is include/carta1_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/carta1_hip.h"

// ---- device-resident tables (one copy per context, in HBM; hot entries live in L1/K$) -------
struct C1DevTables {
  double tap_e[24];          // (double)QMF_EVEN[j]   constants.js:94-107
  double tap_o[24];          // (double)QMF_ODD[j]  == tap_e[23-j]
  double window[32];         // WINDOW_SHORT
  double mdct_fwd64[32];
  double mdct_fwd256[128];
  double mdct_fwd512[256];
  double mdct_inv64[32];
  double mdct_inv256[128];
  double mdct_inv512[256];
  double fft_tw[256][2];     // twiddle of butterfly k in a stage of half-stride h: [h-1+k]; fft.js:44-64
  double scale_factors[64];
  double norm[64 * 16];      // quantRange(wl)/SCALE_FACTORS[sfi]   quantization.js:42-44
  double log1p10;
  // fraction bits (23) of floor_f32(2^(1/3)) and floor_f32(2^(2/3)) when every octave of scale_factors
  // shares them (always true for the reference's table); sf_fast = 0 selects the table compare
  uint32_t sf_m1, sf_m2;
  int32_t sf_fast;
  // dequantization (q * SF) / range as q0 = a*y, r = fma(-q0, range, a), fma(r, y, q0) with y = RN(1/range)
  // (Markstein): dq_fast = 1 when the host has checked it against the division for every (bits, sfi, q) of
  // the installed table (8.3 M cases), else the kernel divides
  int32_t dq_fast;
  // stronger: Float32((q * SF) / range) == Float32(q * RN(SF * y)) for every (bits, sfi, q) of the installed table (the
  // host checks all of them): one product per BFU, then a conversion, a multiply and a conversion per coefficient
  int32_t dq_step;
  double inv_range[16];      // RN(1 / (2^(wl) - 1)) by word-length index wl = 1..15
  // ---- binary32 tables of the speculative path (c1_k_spec.hip): roundings of the tables above ----
  float tap32[24];           // QMF_EVEN (binary32 in the reference already)
  float tap_pair[26][2];     // (E[23-j], E[j]) for j < 24: one packed FMA advances an (odd, even) chain pair; then (0, E[11]), (E[11], 0)
  float win32[32];           // fl32(WINDOW_SHORT)
  float pre32_64[16][2];     // fl32 of the MDCT (cos, sin) pairs
  float pre32_256[64][2];
  float pre32_512[128][2];
  float r4b[4][3][2];        // radix-4 round over stages 4, 8:  k -> wa = tw[3+k], wb = tw[7+k], fl32(wa*wb)
  float r4c[16][3][2];       // radix-4 round over stages 16, 32: k -> wa = tw[15+k], wb = tw[31+k], fl32(wa*wb)
  float r2d[64][2];          // stage 64: fl32(tw[63+k])
  float norm32[64 * 16];     // fl32(norm)
  float inv32_64[16][2], inv32_256[64][2], inv32_512[128][2];   // fl32 of the inverse MDCT tables (binary32 decode)
  float tw32[256][2];        // fl32(fft_tw)
  // error-bound coefficients per band (rounded up): eps_b = cz*Z_b + cw*W + cl*L + eabs
  float spec_cz[4], spec_cw[4], spec_cl[4];
  float spec_cz_short[4], spec_cw_short[4], spec_cl_short[4];   // the same for a band coded with short blocks (16-point transforms)
  float spec_eabs;
  // speculative transient detector (c1_detect_bound.h): Delta_b = det_ck[b] * ||band samples|| + det_eabs, rounded up
  float det_ck[4];
  float det_eabs;
  int32_t spec_ok;           // 0: the installed tables fail the structural checks the bound relies on -> exact path only
};

// ---- per-call encoder options in device form ---------------------------------------------------
struct C1DevEncOpts {
  double biased[64];         // pow(SCALE_FACTORS, bias) from the host
  double threshold;
  int32_t modes[3];          // fixed modes or -1
  int32_t pad_;
  uint16_t rank[64 * 16];    // rank (1..960) of the Float32 heap priority of (sfi, wl); 0 = unused
  // when the rank order is the order of an integer form A*sfi - B*wl (wl >= 1) / A*sfi + C (wl == 0),
  // as it is for the usual biases, the kernels compute it instead of reading the table
  int32_t rank_affine, rank_a, rank_b, rank_c, rank_off;
  // log2(biased[s]) ~ la_slope * s + la_off: only steers the multiplier search of the candidate bounds (k_alloc_bound);
  // the bounds themselves are formed from `biased`, so an inexact fit costs pruning power, never correctness
  float la_slope, la_off;
  int32_t alloc_no_tonal;     // experiments (C1_ALLOC_NO_TONAL=1): always run the 52-BFU candidate first
};

// ---- geometry ------------------------------------------------------------------------------------
constexpr int kRunFrames = 16;   // consecutive frames of one channel processed by one wave
constexpr int kRunFramesDecode = 64; // consecutive units of one channel decoded by one wave
constexpr int kRunFramesLong = 64;   // same, in the all-long-blocks fast path
constexpr int kSideBytes = 64;   // per unit: sfi[52], modes byte, pad
constexpr int kAllocBytes = 32;  // per unit: 52 wl nibbles, amount index, fallback flag
constexpr int kCandidateBytes = 8 * 8 + 8 * 32 + 8 * 8;  // per unit: 8 totals + 8 results + 8 lower bounds (bit allocation scratch)
constexpr int kCandLbOffset = 8 * 8 + 8 * 32;
constexpr int kEpsFloats = 4;    // per unit (speculative path): error bound of bands 0..2, flags
constexpr uint32_t kEpsFlagSfOpen = 1u;   // flag word: a scale-factor index is not certain within the bound
constexpr uint32_t kEpsFlagExact = 2u;    // flag word: the coefficients are the exact kernels' (bounds are zero)
constexpr int kSpecCheckFrames = 16;      // the speculative analysis re-assesses its material every so many frames of a run

struct C1EncodeLaunch {
  const float *pcm[C1_MAX_CHANNELS];
  int channels;
  int64_t frames;
  int halo_frames;
  const C1DevTables *tables;
  const C1DevEncOpts *opts;
  float *coefs;      // frames*channels*512   (workspace or caller tap)
  uint8_t *side;     // frames*channels*64
  uint8_t *alloc;    // frames*channels*32
  uint8_t *cand;     // frames*channels*kCandBytes: per-candidate totals and results
  uint32_t *work_list;   // frames*channels*7 entries (unit<<3 | candidate): first-round heaps at [0, units), second round behind them
  uint32_t *work_count;  // [0] first-round entries of work_list, [1] entries of sel_list, [2] second-round entries
  uint32_t *sel_list;    // units that kept more than the 52-BFU candidate alive: k_alloc_select picks among their results
  float *bands;      // optional tap (may be null)
  float *mags;       // optional tap of the transient detector's magnitude spectra, frames*channels*256 (64 | 64 | 128)
  float *mag_bounds; // optional tap (speculative detector only): frames*channels*3, Delta of the three bands
  uint8_t *units;    // frames*channels*212   (may be null for stage taps)
  // speculative binary32 path
  float *eps;            // frames*channels*kEpsFloats: per-band bound on |binary32 coefficient - reference coefficient|
  uint32_t *redo_list;   // units whose decisions are not certain within the bound (filled by the pack kernel)
  uint32_t *redo_count;
  uint32_t *realloc_list;   // the listed units whose scale-factor indices were not certain: their allocation is redone too
  uint32_t *realloc_count;
  uint32_t *reana_list;     // the listed units whose coefficients are binary32 ones: the exact analysis rebuilds them (units
  uint32_t *reana_count;    // of runs the exact kernels analysed in the first place are only packed again)
  // material-local speculation: the speculative analysis estimates, every kSpecCheckFrames frames of a
  // run, how many decisions of a unit the guards will leave open; past spec_defer it hands the rest of its run to the
  // exact kernels: defer_list[run * channels + channel] = first unit of the deferred part of that run, 0xffffffff = none
  uint32_t *defer_list;
  // per run and channel, bit k set when the scale-factor guard left frame k of the run open (open_masks[run * channels +
  // channel]; null: not collected).  Those units are re-analysed exactly BEFORE the allocation (c1_api.hip), so that the
  // allocation sees the reference's indices the first time and the redo pass has no allocation chain of its own.
  unsigned long long *open_masks;
  int list_zero_eps;        // list mode of the analysis kernels: also write bounds of zero (the unit's coefficients are exact now)
  float spec_defer;         // threshold on the predicted number of open decisions per unit; +inf: never defer
  // list mode: when unit_list is non-null the kernels process units unit_list[0 .. *unit_count) instead of all
  const uint32_t *unit_list;
  const uint32_t *unit_count;
  int list_runs;         // list mode of the analysis kernels: an entry is the first unit of a run that ends where the run of
                         // run_frames frames around it ends (the runs the speculative kernel handed over), not a single unit
  int run_frames;        // consecutive frames of one channel a wave walks (set by the launchers, c1k_pick_run)
  int spread_bits;       // speculative analysis: ceil(log2(workgroups)) for its block -> run permutation, 0 = stream order
};

struct C1DecodeLaunch {
  const uint8_t *units;
  int channels;
  int64_t frames;
  int halo_units;
  const C1DevTables *tables;
  float *pcm[C1_MAX_CHANNELS];
  int run_frames;        // consecutive units of one channel a wave decodes (set by the launcher)
};

// Run length of the frame-walking kernels: consecutive frames of one channel a wave walks, carrying the filter state
// (one extra warm-up frame per run).  Measured on MI355X (1 M stereo frames, A/B in one session): 64-frame runs beat
// runs sized to fill the wave slots exactly once (205 frames: every wave then finishes at the same moment, and the
// dispatcher has nothing left to balance) by 5 % (speculative analysis) to 8 % (decode); 32 and 128 are within noise
// of 64.  C1_RUN_FRAMES overrides it for experiments.
constexpr int kRunDefault = 64;
// Small batches (a streaming push, a frame closure) take shorter runs: 64 frames walked by one wave are 64 frames of
// latency (a 64-frame mono push: 0.41 ms, of which the walk is 0.3) while the rest of the machine idles; the batch is
// spread over up to 2 048 waves instead, and a run is never shorter than 4 frames (each pays one warm-up frame).
// A function of the batch alone: every kernel of a call, and the run lists they hand each other, use the same length.
inline int c1k_pick_run(int64_t frames, int channels, int slots) {
  static const int forced = getenv("C1_RUN_FRAMES") ? atoi(getenv("C1_RUN_FRAMES")) : 0;
  (void)slots;
  if (forced > 0) return forced;
  const int64_t units = frames * channels;
  if (units >= (int64_t)kRunDefault * 2048) return kRunDefault;
  const int run = (int)((units + 2047) / 2048);
  return run < 4 ? 4 : run;
}
// wave slots of the current device for a 64-thread kernel (occupancy x compute units), cached by the caller
template <class Kernel>
inline int c1k_wave_slots(Kernel kernel) {
  int per_cu = 0, cus = 0, dev = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, 0) != hipSuccess || per_cu <= 0) per_cu = 8;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  return per_cu * cus;
}

// launchers (one per c1_k_*.hip file); all asynchronous on `stream`
void c1k_launch_analysis(const C1EncodeLaunch &L, bool detect, hipStream_t stream);
void c1k_launch_analysis_long(const C1EncodeLaunch &L, hipStream_t stream);   // fixed modes [0,0,0]
// binary32 + error bound: fixed modes [0,0,0] (all_short = false) or all three bands short (true)
void c1k_launch_analysis_spec(const C1EncodeLaunch &L, bool all_short, hipStream_t stream);
// transient detection: features (runs) -> decisions (per unit) -> MDCT from the stored bands (per unit).
// bands_ws: (units + channels) * 512 floats, feat_ws: (units + channels) * kFeatureWsDoubles doubles, modes_ws: units bytes
constexpr int kFeatureWsDoubles = 20;
// lists_ws: 4 + 3 * units uint32 (three counts, then the all-long list, the mixed list, and the units the speculative
// detector left to the exact recheck).  speculative: binary32 detector with a score interval.
// score_tap (tests): units * 3 * 2 doubles {lo, hi} (speculative) or {score, score}.  L.coefs null: decisions only.
void c1k_launch_detect(const C1EncodeLaunch &L, float *bands_ws, double *feat_ws, uint8_t *modes_ws, uint32_t *lists_ws,
                       bool speculative, double *score_tap, hipStream_t stream);
// mdctStage + scale factors from stored band samples (k_mdct_bands): bands_ws (units + channels) * 512 floats with slot row 0 =
// frame -1, modes_ws one byte per unit, lists_ws = {count all-long, count mixed, -, -} then the two unit lists, `units` entries apart
void c1k_launch_mdct_bands(const C1EncodeLaunch &L, const float *bands_ws, const uint8_t *modes_ws, const uint32_t *lists_ws, hipStream_t stream);
// the single-stage functions the reference exports next to encode()/decode() (c1_k_stages.hip); device pointers
void c1k_launch_quantize_one(const C1DevTables *tables, const float *x, int n, int sfi, int bits, int32_t *out, hipStream_t stream);
void c1k_launch_dequantize_one(const C1DevTables *tables, const int32_t *q, int n, int sfi, int bits, float *out, hipStream_t stream);
void c1k_launch_fft_reference(float *real, float *imag, int n, const double *w, double *twiddle_scratch /* n doubles */, hipStream_t stream);
void c1k_launch_window_bands(const float *bands, const uint8_t *modes, int64_t units, const C1DevTables *tables, float *out, hipStream_t stream);
void c1k_launch_detect_spec_tap(const C1EncodeLaunch &L, float *bands_ws, double *feat_ws, hipStream_t stream);   // L.mags, L.mag_bounds
void c1k_launch_libm(int fn, const double *in, double *out, int64_t n, hipStream_t stream);
void c1k_launch_log2f_error(uint32_t first, uint64_t count, unsigned long long *out, hipStream_t stream);   // out: 2 x u64 on the device
void c1k_launch_allocate(const C1EncodeLaunch &L, hipStream_t stream);
void c1k_launch_alloc_tap(const C1EncodeLaunch &L, double *out, hipStream_t stream);   // test tap: totals and lower bounds of all candidates
void c1k_launch_pack(const C1EncodeLaunch &L, bool all_long, hipStream_t stream);   // all_long: every unit has modes [0,0,0]
void c1k_launch_pack_spec(const C1EncodeLaunch &L, bool all_long, hipStream_t stream);   // binary32 quantization with the guard band; fills the redo list
// running totals (c1_ctx::d_spec_totals) and their page-locked mirror; kind 0 speculative call (counts = list head), 1 binary32
// quantization of exact coefficients (counts[0] = repacked), 2 speculative detector (counts[0] = rechecked)
void c1k_launch_spec_totals(unsigned long long *totals, uint64_t units, const uint32_t *counts, int kind, hipStream_t stream);
// the speculative kernel's slot array (L.defer_list) -> dense list of deferred runs; counts[0] = entries, counts[1] = units covered
void c1k_launch_defer_compact(const C1EncodeLaunch &L, uint32_t *list, uint32_t *counts, hipStream_t stream);
// open_masks -> a dense list of units (counts[0] = entries), for the exact pre-pass of the units with an open scale factor
void c1k_launch_open_compact(const C1EncodeLaunch &L, uint32_t *list, uint32_t *count, hipStream_t stream);
void c1k_launch_decode(const C1DecodeLaunch &L, bool binary32, hipStream_t stream);   // binary32: opt-in, PCM within rounding noise of the reference
// kind_mask: bit k = fill the 512-frame segments with (segment & 3) == k (15 = all)
void c1k_launch_generate_white(const uint32_t *frame_states, int64_t frames, float *pcm, int kind_mask, double amp, hipStream_t stream);
void c1k_launch_generate_pink(const uint32_t *segment_states, int64_t frames, float *pcm, int kind_mask, hipStream_t stream);
void c1k_launch_generate_sines(int64_t frames, float *pcm, int kind_mask, uint32_t seed, hipStream_t stream);
void c1k_launch_pcm_from_int(const void *src, int bits, int channels, int64_t n, float *const *pcm, hipStream_t stream);
void c1k_launch_pcm_to_int16(const float *const *pcm, int channels, int64_t n, int16_t *dst, hipStream_t stream);
