// Rate of v_mfma_f64_16x16x4_f64 on gfx950 and whether it overlaps with fp64 VALU work of other waves of the SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_f64_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

// mode 0: every wave issues MFMAs; mode 1: every wave issues VALU FMAs; mode 2: even waves MFMA, odd waves VALU
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double seed) {
  const int wave = threadIdx.x >> 6;
  double a = seed + threadIdx.x, b = seed * 0.5 + threadIdx.x;
  double4_t c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1}, c2 = {2, 2, 2, 2}, c3 = {3, 3, 3, 3};
  double v[8];
  for (int i = 0; i < 8; i++) v[i] = seed + i;
  const bool do_mfma = MODE == 0 || (MODE == 2 && (wave & 1) == 0);
  if (do_mfma) {
    for (int it = 0; it < iters; it++) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
  } else {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
    }
  }
  double acc = 0;
  for (int i = 0; i < 4; i++) acc += c0[i] + c1[i] + c2[i] + c3[i];
  for (int i = 0; i < 8; i++) acc += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
float run(double *out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 0, 0, out, 10, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 0, 0, out, iters, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double *out;
  hipMalloc(&out, 256 * 4 * 256 * sizeof(double));
  const int iters = 20000;
  // 4 workgroups of 4 waves per CU -> 4 waves per SIMD; per wave: iters*4 MFMAs or iters*64 VALU FMAs
  const float m0 = run<0>(out, iters), m1 = run<1>(out, iters), m2 = run<2>(out, iters);
  const double mfma_per_simd = 4.0 * iters * 4, valu_per_simd = 4.0 * iters * 64;
  printf("all waves MFMA : %.2f ms -> %.1f cycles per MFMA (1024 FMA) per SIMD at 2.4 GHz\n", m0, m0 * 1e-3 * 2.4e9 / mfma_per_simd);
  printf("all waves VALU : %.2f ms -> %.2f cycles per v_fma_f64 per SIMD\n", m1, m1 * 1e-3 * 2.4e9 / valu_per_simd);
  printf("half and half  : %.2f ms (sum of halves would be %.2f, max %.2f)\n", m2, (m0 + m1) / 2, (m0 > m1 ? m0 : m1) / 2);
  return 0;
}
