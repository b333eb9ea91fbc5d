// VALU issue cost of the instructions the ATRAC1 kernels are made of, on gfx950.
// Each kernel runs N iterations of 32 independent instructions of one kind per wave; 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP32(X) X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X

template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, int iters, double seed) {
  double a[8];
  float f[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; f[i] = (float)(seed + i); }
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
    } else if (KIND == 1) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
    } else if (KIND == 2) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
    } else if (KIND == 3) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
    } else if (KIND == 4) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
    } else if (KIND == 5) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
    } else if (KIND == 6) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
    } else if (KIND == 7) {
      double s = seed;
      asm volatile("" : "+s"(s));
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(a[(i + 1) & 7]));
    }
  }
  double acc = 0;
  for (int i = 0; i < 8; i++) acc += a[i] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int KIND>
double run(const char *name, double *out) {
  const int iters = 20000, blocks = 256 * 4;   // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 4 waves x iters x 32 instructions
  const double inst_per_simd = 4.0 * iters * 32;
  const double ns_per_inst = ms * 1e6 / inst_per_simd;
  printf("%-16s %8.3f ms  %.3f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms, ns_per_inst, ns_per_inst * 2.4);
  return ns_per_inst;
}

int main() {
  double *out;
  hipMalloc(&out, 256 * 4 * 256 * sizeof(double));
  run<0>("v_fma_f64", out);
  run<1>("v_add_f64", out);
  run<2>("v_mul_f64", out);
  run<3>("v_cvt_f32_f64", out);
  run<4>("v_cvt_f64_f32", out);
  run<5>("v_add_u32", out);
  run<6>("v_fma_f32", out);
  run<7>("v_fmac_f64 sgpr", out);
  return 0;
}
