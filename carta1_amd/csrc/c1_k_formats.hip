// c1_k_formats.hip -- synthetic input generators and the WAV sample-format kernels either side of the path
#include "c1_device.h"

namespace {

// =====================================================================================================
// synthetic signals
// =====================================================================================================
__device__ __forceinline__ double xorshift_u(uint32_t &s) {
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  return ((double)s / 4294967296.0) * 2.0 - 1.0;
}
// one thread per frame; frame_states[f] = PRNG state before the frame's first sample.
// kind_mask: bit k set = write the 512-frame segments with (segment & 3) == k (the mixed corpus interleaves generators)
__global__ void k_generate_white(const uint32_t *frame_states, int64_t frames, float *pcm, int kind_mask, double amp) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= frames || !((kind_mask >> ((f >> 9) & 3)) & 1)) return;
  uint32_t s = frame_states[f];
  float4 *dst = reinterpret_cast<float4 *>(pcm + f * 512);
  for (int i = 0; i < 128; i++) {
    float4 v;
    v.x = f32(xorshift_u(s) * amp); v.y = f32(xorshift_u(s) * amp);
    v.z = f32(xorshift_u(s) * amp); v.w = f32(xorshift_u(s) * amp);
    dst[i] = v;
  }
}
// stationary partials with a slow amplitude modulation: the input class where binary32 arithmetic flips the most
// decisions.  One thread per frame; the partials change from segment to segment.
__global__ void k_generate_sines(int64_t frames, float *pcm, int kind_mask, uint32_t seed) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= frames || !((kind_mask >> ((f >> 9) & 3)) & 1)) return;
  const uint32_t seg = (uint32_t)(f >> 9) * 2654435761u + seed;
  const double f0 = 55.0 * exp2((double)(seg % 61u) / 12.0), f1 = f0 * (2.0 + (double)((seg >> 8) % 5u)), f2 = 3000.0 + (double)((seg >> 16) % 9000u);
  const double w0 = 6.283185307179586 * f0 / 44100.0, w1 = 6.283185307179586 * f1 / 44100.0, w2 = 6.283185307179586 * f2 / 44100.0;
  const double wm = 6.283185307179586 * 0.7 / 44100.0;
  float *dst = pcm + f * 512;
  for (int i = 0; i < 512; i++) {
    const double t = (double)((f & 511) * 512 + i);
    const double v = (0.45 * sin(w0 * t) + 0.12 * sin(w1 * t + 1.0) + 0.02 * sin(w2 * t + 2.0)) * (1.0 + 0.3 * sin(wm * t));
    dst[i] = f32(v);
  }
}
// one thread per 512-frame segment: p = 0.98p + 0.05u, plus 0.8u' in the second half of frames 5 mod 8
__global__ void k_generate_pink(const uint32_t *segment_states, int64_t frames, float *pcm, int kind_mask) {
  const int64_t seg = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (seg * 512 >= frames || !((kind_mask >> (seg & 3)) & 1)) return;
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

void c1k_launch_generate_white(const uint32_t *frame_states, int64_t frames, float *pcm, int kind_mask, double amp, hipStream_t stream) {
  hipLaunchKernelGGL(k_generate_white, dim3((unsigned)((frames + 63) / 64)), dim3(64), 0, stream, frame_states, frames, pcm, kind_mask, amp);
}
void c1k_launch_generate_pink(const uint32_t *segment_states, int64_t frames, float *pcm, int kind_mask, hipStream_t stream) {
  const int64_t segs = (frames + 511) / 512;
  hipLaunchKernelGGL(k_generate_pink, dim3((unsigned)((segs + 63) / 64)), dim3(64), 0, stream, segment_states, frames, pcm, kind_mask);
}
void c1k_launch_generate_sines(int64_t frames, float *pcm, int kind_mask, uint32_t seed, hipStream_t stream) {
  hipLaunchKernelGGL(k_generate_sines, dim3((unsigned)((frames + 63) / 64)), dim3(64), 0, stream, frames, pcm, kind_mask, seed);
}
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
