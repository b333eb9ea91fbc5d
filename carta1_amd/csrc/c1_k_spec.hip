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
//                         window reads at "lane base + immediate")  ->  in2 (512)  ->  FFT points z (320 float2)
//   R2 = mem[840, 1352)   stage-2 work buffer (302)  ->  in0 | in1 (256 each)  ->  coefficients (512)
constexpr int kR2 = 840;
constexpr int kMemFloats = kR2 + 512;
struct alignas(16) SpecLds {
  alignas(16) float mem[kMemFloats];
  alignas(16) float d1[48];          // stage-1 delay line (46)
  alignas(16) float d2[48];          // stage-2 delay line (46)
  alignas(16) float pre0[32];        // what the next frame's MDCT inputs start with: windowed overlap of bands 0, 1 ...
  alignas(16) float pre1[32];
  alignas(16) float pre2[72];        // ... and of band 2 (32), then the 39 delayed high-band samples
  alignas(16) float win[32];         // fl32(WINDOW_SHORT)
  alignas(4) uint8_t sfi[64];
};
static_assert(sizeof(SpecLds) <= 6656, "speculative analysis: 24 waves per CU");

__device__ __forceinline__ int w1_phys(int v) { return 12 * (v >> 3) + (v & 7); }

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

// D consecutive outputs of the decimating QMF from the lane's window w[0 .. 46 + 2 D): output d uses w[2 d ..]:
//   even = sum_j E[j] w[2 d + 47 - 2 j],  odd = sum_m E[m] w[2 d + 2 m]   (QMF_ODD[j] = QMF_EVEN[23 - j])
// each as chain A (taps 0..11 ascending) + chain B (taps 23..13 descending), then the centre tap 12.
template <int D, int N>
__device__ __forceinline__ void qmf_core_f32(const float (&w)[N], TablesPtr T, float (&lo)[D], float (&hi)[D]) {
  float tap[24];
#pragma unroll
  for (int j = 0; j < 24; j++) tap[j] = T->tap32[j];
#pragma unroll
  for (int d = 0; d < D; d++) {
    const int o = 2 * d;
    float a = tap[0] * w[o + 47];
#pragma unroll
    for (int j = 1; j <= 11; j++) a = __builtin_fmaf(tap[j], w[o + 47 - 2 * j], a);
    float b = tap[23] * w[o + 1];
#pragma unroll
    for (int j = 22; j >= 13; j--) b = __builtin_fmaf(tap[j], w[o + 47 - 2 * j], b);
    const float ev = __builtin_fmaf(tap[12], w[o + 23], a + b);
    float c = tap[0] * w[o];
#pragma unroll
    for (int m = 1; m <= 11; m++) c = __builtin_fmaf(tap[m], w[o + 2 * m], c);
    float e = tap[23] * w[o + 46];
#pragma unroll
    for (int m = 22; m >= 13; m--) e = __builtin_fmaf(tap[m], w[o + 2 * m], e);
    const float od = __builtin_fmaf(tap[12], w[o + 24], c + e);
    lo[d] = ev + od;
    hi[d] = ev - od;
  }
}

__device__ __forceinline__ float2 cmul32(float2 x, float2 w) {
  return make_float2(__builtin_fmaf(x.x, w.x, -(x.y * w.y)), __builtin_fmaf(x.x, w.y, x.y * w.x));
}
__device__ __forceinline__ float2 operator+(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 operator-(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

__device__ __forceinline__ float2 table_f2(TablesRsrc R, int byte_offset) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b64(R, byte_offset, 0, 0);
  float2 d;
  __builtin_memcpy(&d, &v, sizeof d);
  return d;
}

// lane-only geometry of the long-block core (same ownership as mdct_long_r4: lanes 0..15 band 0, 16..31 band 1,
// 32..63 band 2, four FFT points per lane), with the MDCT inputs at mem[kR2] (in0), mem[kR2 + 256] (in1), mem[0] (in2)
struct SpecGeometry {
  int ia[4], ic[4], ib0, id0, ib3, id3;
  int pre_tab[4];
  int za, zb, zc, zd;
  int twb, twc, twd;
  int cx[4], cy[4];
  int post_tab[4];
  bool band2;
};
__device__ __forceinline__ SpecGeometry spec_geometry(int lane) {
  SpecGeometry G;
  const int band = lane < 16 ? 0 : (lane < 32 ? 1 : 2);
  const int g = lane - (band == 0 ? 0 : (band == 1 ? 16 : 32));
  const int n4 = band == 2 ? 128 : 64, q = n4 / 4;
  const int r = bitrev(g, band == 2 ? 5 : 4);
  const int in_base = band == 0 ? kR2 : (band == 1 ? kR2 + 256 : 0);
  const int tab_base = band == 2 ? (int)offsetof(C1DevTables, pre32_512) : (int)offsetof(C1DevTables, pre32_256);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int jp = ((j & 1) << 1) | (j >> 1);
    const int k = r + q * jp, i = 2 * k;
    G.ia[j] = in_base + 3 * n4 - 1 - i;
    G.ic[j] = in_base + n4 + i;
    G.pre_tab[j] = tab_base + 8 * k;
  }
  G.ib0 = in_base + 3 * n4 + 2 * r;
  G.id0 = in_base + n4 - 1 - 2 * r;
  const int i3 = 2 * (r + 3 * q);
  G.ib3 = in_base + i3 - n4;
  G.id3 = in_base + 5 * n4 - 1 - i3;
  const int pbase = band == 0 ? 0 : (band == 1 ? 64 : 128);
  G.za = zslot(pbase + 4 * g);
  G.zb = zslot(pbase + 16 * (g >> 2) + (g & 3));
  G.twb = (int)offsetof(C1DevTables, r4b) + 24 * (g & 3);
  G.zc = zslot(pbase + 64 * (g >> 4) + (g & 15));
  G.twc = (int)offsetof(C1DevTables, r4c) + 24 * (g & 15);
  G.band2 = band == 2;
  G.zd = zslot(128 + (g & 31));
  G.twd = (int)offsetof(C1DevTables, r2d) + 8 * (g & 31);
  const int cbase = band == 0 ? 0 : (band == 1 ? 128 : 256), n2 = 2 * n4;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i = band == 2 ? g + (j == 1 ? 64 : (j == 2 ? 32 : (j == 3 ? 96 : 0))) : g + 16 * j;
    G.post_tab[j] = tab_base + 8 * i;
    const int e0 = cbase + 2 * i, e1 = cbase + n2 - 1 - 2 * i;
    G.cx[j] = band == 0 ? e0 : e1;
    G.cy[j] = band == 0 ? e1 : e0;
  }
  return G;
}

// one radix-4 round over two reference stages: x1, x2, x3 times wa, wb, wa*wb, then the 4-point butterfly
__device__ __forceinline__ void radix4_round(float2 (&x)[4], float2 wa, float2 wb, float2 wab) {
  const float2 y1 = cmul32(x[1], wa), y2 = cmul32(x[2], wb), y3 = cmul32(x[3], wab);
  const float2 t0 = x[0] + y1, t1 = x[0] - y1, t2 = y2 + y3, t3 = y2 - y3;
  x[0] = t0 + t2;
  x[2] = t0 - t2;
  x[1] = make_float2(t1.x + t3.y, t1.y - t3.x);
  x[3] = make_float2(t1.x - t3.y, t1.y + t3.x);
}

__global__ __launch_bounds__(C1_WAVE, 5) void k_analysis_spec(C1EncodeLaunch L) {
  __shared__ SpecLds S;
  const int lane0 = threadIdx.x;
  int lane = lane0;
  const int ch = blockIdx.x % L.channels;
  const int64_t f0 = (int64_t)(blockIdx.x / L.channels) * kRunFramesLong;
  const float *__restrict__ pcm = L.pcm[ch];
  float *mem = S.mem;

  for (int i = lane; i < 48; i += 64) { S.d1[i] = 0.0f; S.d2[i] = 0.0f; }
  if (lane < 32) { S.pre0[lane] = 0.0f; S.pre1[lane] = 0.0f; S.win[lane] = C1_TABLES(L.tables)->win32[lane]; }
  for (int i = lane; i < 72; i += 64) S.pre2[i] = 0.0f;
  if (lane < 16) reinterpret_cast<uint32_t *>(S.sfi)[lane] = 0u;
  const SpecGeometry G = spec_geometry(lane0);
  const SfLong SFL = sf_long_geometry(lane0);
  const TablesRsrc RT = tables_rsrc(L.tables);
  float p_prev = 0.0f, q_prev = 0.0f;        // PCM / stage-1-low energies of the previous frame
  __syncthreads();

  const int64_t f_end = (f0 + kRunFramesLong < L.frames) ? f0 + kRunFramesLong : L.frames;
  int64_t f_first = f0 - 1;                   // one frame of history rebuilds the state
  if (f_first < -(int64_t)L.halo_frames) f_first = -(int64_t)L.halo_frames;
  if (f_first > f0) f_first = f0;
  float4 pre_a, pre_b;
  {
    const float4 *p4 = reinterpret_cast<const float4 *>(pcm + f_first * 512);
    pre_a = p4[lane0]; pre_b = p4[64 + lane0];
  }
  for (int64_t f = f_first; f < f_end; ++f) {
    const bool emit = (f >= f0);
    TablesPtr T = tables_for_this_frame(L.tables);
    lane = lane_for_this_frame(lane0);

    // ---------------- stage-1 work buffer, PCM energy ----------------
    float P;
    {
      const float4 a = pre_a, b = pre_b;
      if (f + 1 < f_end) {
        const float4 *p4 = reinterpret_cast<const float4 *>(pcm + (f + 1) * 512);
        pre_a = p4[lane]; pre_b = p4[64 + lane];
      }
      if (lane < 46) mem[w1_phys(lane)] = S.d1[lane];
      const int v = 46 + 4 * lane;
      *reinterpret_cast<float2 *>(&mem[w1_phys(v)]) = make_float2(a.x, a.y);
      *reinterpret_cast<float2 *>(&mem[w1_phys(v + 2)]) = make_float2(a.z, a.w);
      *reinterpret_cast<float2 *>(&mem[w1_phys(v + 256)]) = make_float2(b.x, b.y);
      *reinterpret_cast<float2 *>(&mem[w1_phys(v + 258)]) = make_float2(b.z, b.w);
      float p = a.x * a.x;
      p = __builtin_fmaf(a.y, a.y, p); p = __builtin_fmaf(a.z, a.z, p); p = __builtin_fmaf(a.w, a.w, p);
      p = __builtin_fmaf(b.x, b.x, p); p = __builtin_fmaf(b.y, b.y, p); p = __builtin_fmaf(b.z, b.z, p); p = __builtin_fmaf(b.w, b.w, p);
      P = wave_sum(p);
    }
    __syncthreads();
    // ---------------- first QMF stage ----------------
    float Q;
    {
      float w[56];
      const float4 *src = reinterpret_cast<const float4 *>(mem + 12 * lane);
#pragma unroll
      for (int k = 0; k < 14; k++) {
        const float4 t = src[3 * (k >> 1) + (k & 1)];      // floats 12 (k >> 1) + 4 (k & 1)
        w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w;
      }
      float lo[4], hi[4];
      if (own_block()) qmf_core_f32<4>(w, T, lo, hi); else { for (int d = 0; d < 4; d++) { lo[d] = w[d]; hi[d] = 1.0f; } }
      if (lane < 46) { S.d1[lane] = mem[w1_phys(512 + lane)]; mem[kR2 + lane] = S.d2[lane]; }
      *reinterpret_cast<float2 *>(&mem[kR2 + 46 + 4 * lane]) = make_float2(lo[0], lo[1]);
      *reinterpret_cast<float2 *>(&mem[kR2 + 48 + 4 * lane]) = make_float2(lo[2], lo[3]);
      float q = lo[0] * lo[0];
      q = __builtin_fmaf(lo[1], lo[1], q); q = __builtin_fmaf(lo[2], lo[2], q); q = __builtin_fmaf(lo[3], lo[3], q);
      Q = wave_sum(q);
      // band 2 = the high band behind its 39-sample delay (encoder.js:84-90): what the previous frame left (overlap,
      // 39 samples), then this frame's outputs; the last 32 samples of the band are windowed (encoder.js:309-316)
      if (emit) {
        for (int i = lane; i < 71; i += 64) mem[112 + i] = S.pre2[i];
        const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (lane < 56) *reinterpret_cast<float4 *>(&mem[lane < 28 ? 4 * lane : 400 + 4 * (lane - 28)]) = zero4;   // [0,112), [400,512)
      }
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const int pos = 39 + 4 * lane + d;                  // position in band 2 of this frame
        const float x = hi[d];
        if (pos < 224) { if (emit) mem[144 + pos] = x; }
        else if (pos < 256) {
          const int k = pos - 224;
          S.pre2[k] = S.win[k] * x;
          if (emit) mem[144 + pos] = x * S.win[31 - k];
        } else S.pre2[32 + pos - 256] = x;
      }
    }
    __syncthreads();
    // ---------------- second QMF stage ----------------
    {
      float w[52];
      const float4 *src = reinterpret_cast<const float4 *>(mem + kR2 + 4 * lane);
#pragma unroll
      for (int k = 0; k < 13; k++) {
        const float4 t = src[k];
        w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w;
      }
      float lo[2], hi[2];
      if (own_block()) qmf_core_f32<2>(w, T, lo, hi); else { for (int d = 0; d < 2; d++) { lo[d] = w[d]; hi[d] = 1.0f; } }
      if (lane < 46) S.d2[lane] = mem[kR2 + 256 + lane];
      if (emit) {
        if (lane < 32) { mem[kR2 + 48 + lane] = S.pre0[lane]; mem[kR2 + 256 + 48 + lane] = S.pre1[lane]; }
        const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (lane < 48) {
          const int b = lane < 24 ? 0 : 256, q = lane < 24 ? lane : lane - 24;      // 24 float4 per band: [0,48) and [208,256)
          *reinterpret_cast<float4 *>(&mem[kR2 + b + (q < 12 ? 4 * q : 208 + 4 * (q - 12))]) = zero4;
        }
      }
      if (lane < 48) {
        if (emit) {
          *reinterpret_cast<float2 *>(&mem[kR2 + 80 + 2 * lane]) = make_float2(lo[0], lo[1]);
          *reinterpret_cast<float2 *>(&mem[kR2 + 256 + 80 + 2 * lane]) = make_float2(hi[0], hi[1]);
        }
      } else {
        const int k = 2 * (lane - 48);
        const float wl0 = S.win[k], wl1 = S.win[k + 1], wh0 = S.win[31 - k], wh1 = S.win[30 - k];
        *reinterpret_cast<float2 *>(&S.pre0[k]) = make_float2(wl0 * lo[0], wl1 * lo[1]);
        *reinterpret_cast<float2 *>(&S.pre1[k]) = make_float2(wl0 * hi[0], wl1 * hi[1]);
        if (emit) {
          *reinterpret_cast<float2 *>(&mem[kR2 + 80 + 2 * lane]) = make_float2(lo[0] * wh0, lo[1] * wh1);
          *reinterpret_cast<float2 *>(&mem[kR2 + 256 + 80 + 2 * lane]) = make_float2(hi[0] * wh0, hi[1] * wh1);
        }
      }
    }
    const float W = __builtin_amdgcn_sqrtf(P + p_prev), Lw = __builtin_amdgcn_sqrtf(Q + q_prev);
    p_prev = P; q_prev = Q;
    __syncthreads();
    if (!emit) continue;

    // ---------------- long-block MDCT in binary32 ----------------
    float2 x[4];
    float zrow;
    {
      const float2 t0 = table_f2(RT, G.pre_tab[0]), t1 = table_f2(RT, G.pre_tab[1]);
      const float2 t2 = table_f2(RT, G.pre_tab[2]), t3 = table_f2(RT, G.pre_tab[3]);
      const float a0 = mem[G.ia[0]], c0 = mem[G.ic[0]], b0 = mem[G.ib0], d0 = mem[G.id0];
      const float a1 = mem[G.ia[1]], c1 = mem[G.ic[1]];
      const float a2 = mem[G.ia[2]], c2 = mem[G.ic[2]];
      const float a3 = mem[G.ia[3]], c3 = mem[G.ic[3]], b3 = mem[G.ib3], d3 = mem[G.id3];
      const float r0 = a0 + b0, m0 = c0 - d0;               // mdct.js:76-89
      const float r3 = a3 - b3, m3 = c3 + d3;               // mdct.js:91-105; for positions 1, 2 the second operands are the zero padding
      x[0] = make_float2(__builtin_fmaf(r0, t0.x, m0 * t0.y), __builtin_fmaf(m0, t0.x, -(r0 * t0.y)));
      x[1] = make_float2(__builtin_fmaf(a1, t1.x, c1 * t1.y), __builtin_fmaf(c1, t1.x, -(a1 * t1.y)));
      x[2] = make_float2(__builtin_fmaf(a2, t2.x, c2 * t2.y), __builtin_fmaf(c2, t2.x, -(a2 * t2.y)));
      x[3] = make_float2(__builtin_fmaf(r3, t3.x, m3 * t3.y), __builtin_fmaf(m3, t3.x, -(r3 * t3.y)));
      float en = x[0].x * x[0].x;
      en = __builtin_fmaf(x[0].y, x[0].y, en);
#pragma unroll
      for (int j = 1; j < 4; j++) { en = __builtin_fmaf(x[j].x, x[j].x, en); en = __builtin_fmaf(x[j].y, x[j].y, en); }
      zrow = row_allreduce(en);
      // stages 1, 2: twiddles 1 and -i, no products
      const float2 u0 = x[0] + x[1], u1 = x[0] - x[1], u2 = x[2] + x[3], u3 = x[2] - x[3];
      x[0] = u0 + u2;
      x[2] = u0 - u2;
      x[1] = make_float2(u1.x + u3.y, u1.y - u3.x);
      x[3] = make_float2(u1.x - u3.y, u1.y + u3.x);
    }
    float2 *z = reinterpret_cast<float2 *>(mem);
    {
      float4 *dst = reinterpret_cast<float4 *>(z + G.za);
      dst[0] = make_float4(x[0].x, x[0].y, x[1].x, x[1].y);
      dst[1] = make_float4(x[2].x, x[2].y, x[3].x, x[3].y);
    }
    const float2 wBa = table_f2(RT, G.twb), wBb = table_f2(RT, G.twb + 8), wBc = table_f2(RT, G.twb + 16);
    __syncthreads();
    {
      float2 *p = z + G.zb;
      x[0] = p[0]; x[1] = p[4]; x[2] = p[8]; x[3] = p[12];
      radix4_round(x, wBa, wBb, wBc);
      p[0] = x[0]; p[4] = x[1]; p[8] = x[2]; p[12] = x[3];
    }
    const float2 wCa = table_f2(RT, G.twc), wCb = table_f2(RT, G.twc + 8), wCc = table_f2(RT, G.twc + 16);
    const float2 wDa = table_f2(RT, G.twd), wDb = table_f2(RT, G.twd + 256);
    __syncthreads();
    {
      float2 *p = z + G.zc;
      x[0] = p[0]; x[1] = p[20]; x[2] = p[40]; x[3] = p[60];
      radix4_round(x, wCa, wCb, wCc);
      if (G.band2) { p[0] = x[0]; p[20] = x[1]; p[40] = x[2]; p[60] = x[3]; }
    }
    const float2 p0 = table_f2(RT, G.post_tab[0]), p1 = table_f2(RT, G.post_tab[1]);
    const float2 p2 = table_f2(RT, G.post_tab[2]), p3 = table_f2(RT, G.post_tab[3]);
    __syncthreads();
    if (G.band2) {
      const float2 *p = z + G.zd;
      x[0] = p[0]; x[1] = p[80]; x[2] = p[40]; x[3] = p[120];
      const float2 y1 = cmul32(x[1], wDa), y3 = cmul32(x[3], wDb);
      const float2 e0 = x[0], e2 = x[2];
      x[0] = e0 + y1; x[1] = e0 - y1; x[2] = e2 + y3; x[3] = e2 - y3;
    }
    float *coef = mem + kR2;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float2 t = j == 0 ? p0 : (j == 1 ? p1 : (j == 2 ? p2 : p3));
      coef[G.cx[j]] = -__builtin_fmaf(x[j].x, t.x, x[j].y * t.y);            // mdct.js:110-119
      coef[G.cy[j]] = __builtin_fmaf(x[j].y, t.x, -(x[j].x * t.y));
    }
    __syncthreads();

    // ---------------- the bound, coefficients out, scale-factor indices with their guard ----------------
    const int64_t unit = f * L.channels + ch;
    {
      float4 *dst = reinterpret_cast<float4 *>(L.coefs + (unit << 9));
      const float4 *src = reinterpret_cast<const float4 *>(coef);
      dst[lane] = src[lane];
      dst[64 + lane] = src[64 + lane];
    }
    // eps_b = cz_b Z_b + cw_b W + cl_b L + eabs  (DESIGN.md 3b); Z_b^2 = energy of the band's pre-twiddled points
    const float Z0 = __builtin_amdgcn_sqrtf(lane_value(zrow, 0)), Z1 = __builtin_amdgcn_sqrtf(lane_value(zrow, 16));
    const float Z2 = __builtin_amdgcn_sqrtf(lane_value(zrow, 32) + lane_value(zrow, 48));
    const float eabs = T->spec_eabs;
    const float e0 = __builtin_fmaf(T->spec_cz[0], Z0, __builtin_fmaf(T->spec_cw[0], W, __builtin_fmaf(T->spec_cl[0], Lw, eabs)));
    const float e1 = __builtin_fmaf(T->spec_cz[1], Z1, __builtin_fmaf(T->spec_cw[1], W, __builtin_fmaf(T->spec_cl[1], Lw, eabs)));
    const float e2 = __builtin_fmaf(T->spec_cz[2], Z2, __builtin_fmaf(T->spec_cw[2], W, __builtin_fmaf(T->spec_cl[2], Lw, eabs)));
    bool unstable;
    {
      const float *src = coef + SFL.src;
      float mx = 0.0f;
#pragma unroll
      for (int j = 0; j < 12; j++) mx = fmaxf(mx, fabsf(src[j < SFL.cnt ? j : SFL.cnt - 1]));
      const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0xB1, 0xf, 0xf, false));
      mx = fmaxf(mx, SFL.wide ? other : 0.0f);
      const float e = SFL.b >= 36 ? e2 : (SFL.b >= 20 ? e1 : e0);
      // every value the reference's maximum can take lies in [mx - e, mx + e]; the index is monotone in the maximum,
      // so it is certain when both ends (widened by the rounding of this very subtraction / addition) agree
      const float lo = fmaxf((mx - e) * 0.99999976f, 0.0f), hi = (mx + e) * 1.00000024f;
      const int s_lo = scale_factor_index_fast(lo, T->sf_m1, T->sf_m2), s_hi = scale_factor_index_fast(hi, T->sf_m1, T->sf_m2);
      const int sfi = scale_factor_index_fast(mx, T->sf_m1, T->sf_m2);
      if (SFL.store) S.sfi[SFL.b] = (uint8_t)sfi;
      unstable = lane < 60 && !(s_lo == s_hi && e < __builtin_huge_valf());
    }
    const bool any_unstable = __builtin_amdgcn_ballot_w64(unstable) != 0;
    __syncthreads();
    if (lane < 16) reinterpret_cast<uint32_t *>(L.side + unit * kSideBytes)[lane] = reinterpret_cast<const uint32_t *>(S.sfi)[lane];
    if (lane == 0) *reinterpret_cast<float4 *>(L.eps + unit * kEpsFloats) = make_float4(e0, e1, e2, __int_as_float(any_unstable ? 1 : 0));
    __syncthreads();
  }
}

__global__ void k_spec_totals(unsigned long long *totals, unsigned long long units, const uint32_t *redo_count) {
  totals[0] += units;
  totals[1] += *redo_count;
}

}  // namespace

void c1k_launch_spec_totals(unsigned long long *totals, uint64_t units, const uint32_t *redo_count, hipStream_t stream) {
  hipLaunchKernelGGL(k_spec_totals, dim3(1), dim3(1), 0, stream, totals, (unsigned long long)units, redo_count);
}

void c1k_launch_analysis_spec(const C1EncodeLaunch &L, hipStream_t stream) {
  const int64_t runs = (L.frames + kRunFramesLong - 1) / kRunFramesLong;
  hipLaunchKernelGGL(k_analysis_spec, dim3((unsigned)(runs * L.channels)), dim3(C1_WAVE), 0, stream, L);
}
