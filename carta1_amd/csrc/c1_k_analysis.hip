// c1_k_analysis.hip -- QMF analysis -> MDCT -> scale factors for fixed block modes (encoder.js:57-349), one wave per run of frames
#include "c1_device.h"

namespace {

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
  float d1[46];                  // stage-1 QMF delay line (binary32 samples; widened when they enter the work buffer)
  float d2[46];                  // stage-2 QMF delay line
  uint32_t late[4][64];          // mdct_long_r4's end-of-transform values per lane (r4_late_word)
  uint32_t sfw[64];              // the lane's scale-factor scan: first coefficient | count << 9 | BFU << 13 | wide << 19 | store << 20
  double win[32];                // WINDOW_SHORT (the tail windowing's lane-varying lookups: a global load each, waited for on the spot)
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

template <bool ALL_LONG>
__global__ __launch_bounds__(C1_WAVE, ALL_LONG ? 4 : 3) void k_analysis_fast(C1EncodeLaunch L) {
  using Lds = typename std::conditional<ALL_LONG, LongLds, MixedLds>::type;
  __shared__ Lds S;
  float *band_;                                        // low128 | mid128 | high256 of the current frame, raw
  if constexpr (ALL_LONG) band_ = S.u.m.zz.band; else band_ = S.band;
  const C1DevEncOpts *O = L.opts;
  const int lane0 = threadIdx.x;
  int lane = lane0;
  // list mode (exact redo of the units the speculative pass could not certify): one item = one unit, rebuilt from
  // its own warm-up frame; otherwise one item = this workgroup's run of L.run_frames frames
  const bool listed = L.unit_list != nullptr;
  const int64_t n_items = listed ? (int64_t)*L.unit_count : (int64_t)gridDim.x;
  // lane-only geometry of the long-block MDCT core and the LDS copies of small tables: once per wave, not once per listed unit
  R4Geometry G4 = r4_geometry(lane0);
  const SfLong SFL = sf_long_geometry(lane0);
  if constexpr (ALL_LONG) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      S.late[j][lane0] = r4_late_word(G4, j);
      G4.cx[j] = G4.cy[j] = G4.post_tab[j] = 0;             // not carried through the loop
    }
    S.sfw[lane0] = (uint32_t)SFL.src | ((uint32_t)SFL.b << 13) | (SFL.wide ? 1u << 19 : 0u) | (SFL.store ? 1u << 20 : 0u);
    if (lane0 < 32) S.win[lane0] = C1_TABLES(L.tables)->window[lane0];
  }
  const SfLong SFM = sf_geometry(lane0, O->modes[0], O->modes[1], O->modes[2]);     // used when !ALL_LONG
  const MixGeometry GM = mix_geometry(lane0, FrameModes{O->modes[0], O->modes[1], O->modes[2]});   // used when !ALL_LONG
  const TablesRsrc RT = tables_rsrc(L.tables);
  for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
  const int64_t listed_unit = listed ? (int64_t)L.unit_list[item] : 0;
  const int ch = listed ? (int)(listed_unit % L.channels) : (int)(blockIdx.x % L.channels);
  const int64_t f0 = listed ? listed_unit / L.channels : (int64_t)(blockIdx.x / L.channels) * L.run_frames;
  // a listed run (material the speculative analysis handed over, c1_k_spec.hip) ends where the run around it ends
  const int64_t run_frames = listed ? (L.list_runs ? (f0 / L.run_frames + 1) * L.run_frames - f0 : 1) : L.run_frames;
  const float *__restrict__ pcm = L.pcm[ch];
  lane = lane0;

  for (int i = lane; i < 46; i += 64) { S.d1[i] = 0.0; S.d2[i] = 0.0; }
  for (int i = lane; i < 296; i += 64) S.hbuf[i] = 0.0f;
  if (lane < 12) S.sfi[52 + lane] = 0;   // modes byte (all long) and padding of the side record
  if constexpr (!ALL_LONG) {
    for (int i = lane; i < 96; i += 64) S.ovl[i] = 0.0f;
  }
  float ov0 = 0.0f, ov1 = 0.0f, ov2 = 0.0f;     // lanes 0..31: mdctOverlap of the three bands, carried in registers
  wave_fence();

  const int64_t f_end = (f0 + run_frames < L.frames) ? f0 + run_frames : L.frames;
  constexpr int kWarm = 1;                         // one frame of history rebuilds the state
  int64_t f_first = f0 - kWarm;
  if (f_first < -(int64_t)L.halo_frames) f_first = -(int64_t)L.halo_frames;   // before the stream start the zero state stays
  if (f_first > f0) f_first = f0;
  // the PCM of the next frame is fetched while the current one is processed (two 16-byte loads per lane)
  // The next frame's PCM is requested a frame ahead, unconditionally (the last frame of a run asks for itself again: under a
  // condition the loaded values are copied into the loop-carried registers right behind the load, i.e. waited for on the
  // spot), and taken delivery of at a point every path through the frame passes, BEFORE the frame's stores are issued: loads
  // and stores share one in-order counter on this part (vmcnt), so a wait for a load behind a store is a wait for the store
  // to reach memory as well (c1_k_spec.hip, tools/isa_waits.py).
  typedef float v4f __attribute__((ext_vector_type(4)));   // whole 16-byte register groups (as eight scalars the asm cost eight moves a frame)
  v4f pre_a, pre_b;
  auto deliver = [&]() {
    asm volatile("" : "+v"(pre_a), "+v"(pre_b));
  };
  {
    const v4f *p4 = reinterpret_cast<const v4f *>(pcm + f_first * 512);
    pre_a = p4[lane0]; pre_b = p4[64 + lane0];
    deliver();
  }
  for (int64_t f = f_first; f < f_end; ++f) {
    const bool emit = (f >= f0);
    TablesPtr T = tables_for_this_frame(L.tables);
    lane = lane_for_this_frame(lane0);

    // ---------------- qmfAnalysisStage (encoder.js:57-96) ----------------
    {
      const v4f a = pre_a, b = pre_b;
      {
        const v4f *p4 = reinterpret_cast<const v4f *>(pcm + ((f + 1 < f_end) ? f + 1 : f) * 512);
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
    wave_fence();
    {
      double ev[4], od[4];
      if constexpr (ALL_LONG) __builtin_amdgcn_s_setprio(3);
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
    wave_fence();
    R4Early EARLY;
    if constexpr (ALL_LONG) EARLY = r4_early(G4, RT);          // in flight during the second QMF stage
    {
      double ev[2], od[2];
      if (own_block()) qmf_analysis_core<2, 2>(S.u.q2.w2, lane, T, ev, od); else { for (int d = 0; d < 2; d++) { ev[d] = S.u.q2.w2[lane + d]; od[d] = 1.0; } }
      if constexpr (ALL_LONG) __builtin_amdgcn_s_setprio(1);
      *reinterpret_cast<float2 *>(&band_[2 * lane]) = make_float2(f32(ev[0] + od[0]), f32(ev[1] + od[1]));
      *reinterpret_cast<float2 *>(&band_[128 + 2 * lane]) = make_float2(f32(ev[0] - od[0]), f32(ev[1] - od[1]));
      *reinterpret_cast<float4 *>(&band_[256 + 4 * lane]) = *reinterpret_cast<const float4 *>(&S.hbuf[4 * lane]);
      if (lane < 46) S.d2[lane] = S.u.q2.w2[pidx<2>(256 + lane)];
    }
    wave_fence();
    {
      float keep = 0.0f;
      if (lane < 39) keep = S.hbuf[256 + lane];
      wave_fence();
      if (lane < 39) S.hbuf[lane] = keep;
    }
    deliver();
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
        const double w_lo = S.win[lane], w_hi = S.win[31 - lane];
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
      if (!emit) { wave_fence(); continue; }
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
      wave_fence();
      float *coef = S.u.m.a.c.coef;
      // Wave priorities: the QMF cores saturate VALU and LDS together (3), the MDCT rounds are chains of dependent
      // round trips that the other waves fill anyway (0), staging and output in between (1).  Measured: -4 %.
      __builtin_amdgcn_s_setprio(0);
      mdct_long_r4(in0, S.u.zp.z, coef, G4, T, RT, EARLY, &S.late[0][0] + lane);
      __builtin_amdgcn_s_setprio(1);
      wave_fence();

      // ---------------- scale-factor indices (bitallocation.js:80-90), then every store of the frame ----------------
      const int64_t unit = f * L.channels + ch;
      {
        // the scan of sf_long from three 16-byte reads (sf_scan_long_groups), its geometry read back from LDS
        const uint32_t sw = S.sfw[lane];
        const float4 *grp = reinterpret_cast<const float4 *>(coef + (sw & 0x1fcu));
        float mx = sf_scan_long_groups(grp[0], grp[1], grp[2]);
        const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0xB1, 0xf, 0xf, false));
        mx = fmaxf(mx, ((sw >> 19) & 1u) ? other : 0.0f);
        const int sfi = T->sf_fast ? scale_factor_index_fast(mx, T->sf_m1, T->sf_m2) : scale_factor_index(mx, T);
        if ((sw >> 20) & 1u) S.sfi[(sw >> 13) & 63u] = (uint8_t)sfi;
      }
      {
        float4 *dst = reinterpret_cast<float4 *>(L.coefs + (unit << 9));
        const float4 *src = reinterpret_cast<const float4 *>(coef);
        dst[lane] = src[lane];
        dst[64 + lane] = src[64 + lane];
      }
      wave_fence();
      if (lane < 16) reinterpret_cast<uint32_t *>(L.side + unit * kSideBytes)[lane] = reinterpret_cast<const uint32_t *>(S.sfi)[lane];
      if ((L.list_runs || L.list_zero_eps) && lane == 16) *reinterpret_cast<float4 *>(L.eps + unit * kEpsFloats) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // exact coefficients: bounds of zero
      wave_fence();
    } else {
      // ---------------- mdctStage with short blocks (encoder.js:170-349), fixed block modes ----------------
      const FrameModes M{O->modes[0], O->modes[1], O->modes[2]};
      float *coef = S.u.m.a.c.coef;
      if (emit) {
        mix_stage(band_, S.ovl, S.u.m.a.g.in, M, lane, RT);
        wave_fence();
        mdct_mixed_r4(S.u.m.a.g.in, S.u.m.zz.z, coef, GM, M.m0 == 0 || M.m1 == 0 || M.m2 == 0, M.m2 == 0, T, RT);
      }
      // applyTailWindowing's overlap half (encoder.js:309-316): W[i] * last 32 raw samples of the band
      for (int i = lane; i < 96; i += 64) {
        const int b = i >> 5, k = i & 31;
        const int Sb = b == 2 ? 256 : 128, off = b == 0 ? 0 : (b == 1 ? 128 : 256);
        S.ovl[i] = f32(T->window[k] * (double)band_[off + Sb - 32 + k]);
      }
      wave_fence();
      if (!emit) continue;

      // ---------------- coefficients out + scale-factor indices (bitallocation.js:80-90) ----------------
      const int64_t unit = f * L.channels + ch;
      {
        float4 *dst = reinterpret_cast<float4 *>(L.coefs + (unit << 9));
        const float4 *src = reinterpret_cast<const float4 *>(coef);
        dst[lane] = src[lane];
        dst[64 + lane] = src[64 + lane];
      }
      sf_long(coef, S.sfi, SFM, T);                       // same 12-read scheme, BFU starts of the fixed modes
      if (lane >= 60 && lane < 63) reinterpret_cast<uint32_t *>(S.sfi)[13 + (lane - 60)] = lane == 60 ? (uint32_t)((M.m0 & 3) | ((M.m1 & 3) << 2) | ((M.m2 & 3) << 4)) : 0u;
      wave_fence();
      if (lane < 16) reinterpret_cast<uint32_t *>(L.side + unit * kSideBytes)[lane] = reinterpret_cast<const uint32_t *>(S.sfi)[lane];
      if ((L.list_runs || L.list_zero_eps) && lane == 16) *reinterpret_cast<float4 *>(L.eps + unit * kEpsFloats) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      wave_fence();
    }
  }
  wave_fence();
  }
}

}  // namespace

void c1k_launch_analysis(const C1EncodeLaunch &L0, bool detect, hipStream_t stream) {
  static const int slots = c1k_wave_slots(k_analysis_fast<false>);
  C1EncodeLaunch L = L0;
  L.run_frames = c1k_pick_run(L.frames, L.channels, slots);
  const int64_t runs = (L.frames + L.run_frames - 1) / L.run_frames;
  const int64_t blocks = L.unit_list ? std::min<int64_t>(L.frames * L.channels, 256 * 12) : runs * L.channels;   // list mode: bounded grid
  const dim3 grid((unsigned)blocks), block(C1_WAVE);
  (void)detect;
  hipLaunchKernelGGL((k_analysis_fast<false>), grid, block, 0, stream, L);
}
void c1k_launch_analysis_long(const C1EncodeLaunch &L0, hipStream_t stream) {
  static const int slots = c1k_wave_slots(k_analysis_fast<true>);
  C1EncodeLaunch L = L0;
  L.run_frames = c1k_pick_run(L.frames, L.channels, slots);
  const int64_t runs = (L.frames + L.run_frames - 1) / L.run_frames;
  // list mode: the number of listed units is only known on the device; a bounded grid strides over the list
  const int64_t blocks = L.unit_list ? std::min<int64_t>(L.frames * L.channels, 256 * 16) : runs * L.channels;
  hipLaunchKernelGGL((k_analysis_fast<true>), dim3((unsigned)blocks), dim3(C1_WAVE), 0, stream, L);
}
static_assert(sizeof(LongLds) <= 10240, "all-long analysis: 16 waves per CU (4 per SIMD, 128 registers) need <= 10 KiB of LDS per wave");
