// c1_k_stages.hip -- the single-stage functions the reference exports next to encode()/decode() (codec/index.js:30-35,42):
// quantize, dequantize (codec/coding/quantization.js:34-78) and FFT.fft (codec/transforms/fft.js:14-68) as small kernels in
// the reference's own arithmetic (binary64 operations, binary32 at every typed-array store, no fused multiply-add), plus the
// in-place windowing mdctStage leaves in its band arrays (encoder.js:244,292,309-316).  These are not the hot path (that
// quantizes inside k_pack and transforms inside the analysis kernels); they exist so that an application importing those
// names from the reference finds them here, computed on the device like everything else.
#include "c1_device.h"

namespace {

// quantization.js:34-56
__global__ void k_quantize_one(const C1DevTables *tables, const float *__restrict__ x, int n, int sfi, int bits, int32_t *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (bits == 0 || sfi == 0) { out[i] = 0; return; }
  const int32_t range = (int32_t)((1u << ((bits - 1) & 31)) - 1u);     // (1 << (bitsPerSample - 1)) - 1, shift count mod 32 as in ECMAScript
  const double norm = (double)range / tables->scale_factors[sfi & 63];
  const double v = (double)x[i] * norm;
  const int32_t y = to_int32(v + (v >= 0 ? 0.5 : -0.5));
  out[i] = y > range ? range : (y < -range ? -range : y);
}

// quantization.js:65-78
__global__ void k_dequantize_one(const C1DevTables *tables, const int32_t *__restrict__ q, int n, int sfi, int bits, float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (bits == 0 || sfi == 0) { out[i] = 0.0f; return; }
  const int32_t range = (int32_t)((1u << ((bits - 1) & 31)) - 1u);
  out[i] = f32(((double)q[i] * tables->scale_factors[sfi & 63]) / (double)range);
}

// fft.js:14-68, one workgroup.  The reference advances the twiddle of a stage by a complex multiplication per butterfly
// index k (:62-64), the same sequence in every start block: thread 0 runs that recurrence once per stage into `tw`, then
// the butterflies of the stage are independent.  real / imag are Float32Arrays: every store rounds to binary32.
__global__ __launch_bounds__(256) void k_fft_reference(float *real, float *imag, int n, const double *__restrict__ w, double *tw) {
  const int tid = threadIdx.x;
  int bits = 0;
  while ((1 << bits) < n) bits++;
  for (int i = tid; i < n; i += 256) {                       // bit reversal (:21-32)
    const int r = bits ? (int)(__brev((unsigned)i) >> (32 - bits)) : 0;
    if (r > i) {
      const float a = real[i], b = imag[i];
      real[i] = real[r]; imag[i] = imag[r];
      real[r] = a; imag[r] = b;
    }
  }
  __syncthreads();
  int stage = 0;
  for (int stride = 2; stride <= n; stride <<= 1, ++stage) {
    const int half = stride >> 1;
    if (tid == 0) {
      const double wr = w[2 * stage], wi = w[2 * stage + 1];
      double tr = 1.0, ti = 0.0;
      for (int k = 0; k < half; k++) {
        tw[2 * k] = tr; tw[2 * k + 1] = ti;
        const double nr = tr * wr - ti * wi;
        ti = tr * wi + ti * wr;
        tr = nr;
      }
    }
    __threadfence_block();
    __syncthreads();
    for (int b = tid; b < n / 2; b += 256) {
      const int k = b & (half - 1), start = (b / half) * stride;
      const int e = start + k, o = e + half;
      const double er = real[e], ei = imag[e], orr = real[o], oi = imag[o];
      const double tr = tw[2 * k], ti = tw[2 * k + 1];
      const double xr = orr * tr - oi * ti;
      const double xi = orr * ti + oi * tr;
      real[e] = f32(er + xr);
      imag[e] = f32(ei + xi);
      real[o] = f32(er - xr);
      imag[o] = f32(ei - xi);
    }
    __threadfence_block();
    __syncthreads();
  }
}

// what mdctStage leaves in the band arrays it was given (they are returned to the caller): a long band's last 32 samples
// times W[31 - i] (applyTailWindowing, encoder.js:309-316), every 32-sample block of a short band times W[31 - i]
// (transformShortBlocks, :279-304); Float32Array element *= double
__global__ void k_window_bands(const float *__restrict__ bands, const uint8_t *__restrict__ modes, int64_t units, const C1DevTables *tables, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= units * 512) return;
  const int64_t unit = i >> 9;
  const int s = (int)(i & 511);
  const int band = s < 128 ? 0 : (s < 256 ? 1 : 2);
  const int pos = s - (band == 0 ? 0 : (band == 1 ? 128 : 256)), len = band == 2 ? 256 : 128;
  const int mode = (modes[unit] >> (2 * band)) & 3;
  const float x = bands[i];
  const bool windowed = mode != 0 || pos >= len - 32;
  out[i] = windowed ? f32((double)x * tables->window[31 - (pos & 31)]) : x;
}

}  // namespace

void c1k_launch_quantize_one(const C1DevTables *tables, const float *x, int n, int sfi, int bits, int32_t *out, hipStream_t stream) {
  hipLaunchKernelGGL(k_quantize_one, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tables, x, n, sfi, bits, out);
}
void c1k_launch_dequantize_one(const C1DevTables *tables, const int32_t *q, int n, int sfi, int bits, float *out, hipStream_t stream) {
  hipLaunchKernelGGL(k_dequantize_one, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tables, q, n, sfi, bits, out);
}
void c1k_launch_fft_reference(float *real, float *imag, int n, const double *w, double *twiddle_scratch, hipStream_t stream) {
  hipLaunchKernelGGL(k_fft_reference, dim3(1), dim3(256), 0, stream, real, imag, n, w, twiddle_scratch);
}
void c1k_launch_window_bands(const float *bands, const uint8_t *modes, int64_t units, const C1DevTables *tables, float *out, hipStream_t stream) {
  hipLaunchKernelGGL(k_window_bands, dim3((unsigned)((units * 512 + 255) / 256)), dim3(256), 0, stream, bands, modes, units, tables, out);
}
