// Does v_mfma_f64_16x16x4_f64 accumulate its four products like a chain of correctly rounded FMAs in k order?
// (Question behind a possible QMF-on-MFMA path; the reference's sums are sequential double additions of exact products.)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_sem mfma_f64_semantics.hip && ./mfma_sem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k(const double *A, const double *B, const double *C, double *D) {
  // A: 16x4 row-major [m][k]; B: 4x16 [k][n]; C/D: 16x16 [m][n]
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + (l / 16)];
  const double b = B[(l / 16) * 16 + (l % 16)];
  double4_t c;
  for (int i = 0; i < 4; i++) c[i] = C[(4 * (l / 16) + i) * 16 + (l % 16)];
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; i++) D[(4 * (l / 16) + i) * 16 + (l % 16)] = c[i];
}

static uint64_t s = 0x9E3779B97F4A7C15ull;
static double rnd(bool f32like) {
  s ^= s << 13; s ^= s >> 7; s ^= s << 17;
  double u = (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0;
  int e = (int)((s >> 3) % 40) - 20;
  double v = std::ldexp(u, e);
  return f32like ? (double)(float)v : v;
}

int main() {
  double *dA, *dB, *dC, *dD;
  hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 256 * 8); hipMalloc(&dD, 256 * 8);
  for (int mode = 0; mode < 2; mode++) {
    long n = 0, seq_fma = 0, seq_fma_rev = 0, seq_muladd = 0, tree = 0;
    for (int trial = 0; trial < 2000; trial++) {
      double A[64], B[64], C[256], D[256];
      for (int i = 0; i < 64; i++) { A[i] = rnd(mode == 0); B[i] = rnd(mode == 0); }
      for (int i = 0; i < 256; i++) C[i] = rnd(false);
      hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
      hipMemcpy(dC, C, sizeof C, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
      hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
      for (int m = 0; m < 16; m++) for (int c = 0; c < 16; c++) {
        double r1 = C[m * 16 + c], r2 = C[m * 16 + c], r3 = C[m * 16 + c];
        for (int kk = 0; kk < 4; kk++) { r1 = std::fma(A[m * 4 + kk], B[kk * 16 + c], r1); volatile double p = A[m * 4 + kk] * B[kk * 16 + c]; r3 = r3 + p; }
        for (int kk = 3; kk >= 0; kk--) r2 = std::fma(A[m * 4 + kk], B[kk * 16 + c], r2);
        volatile double p0 = A[m * 4] * B[c], p1 = A[m * 4 + 1] * B[16 + c], p2 = A[m * 4 + 2] * B[32 + c], p3 = A[m * 4 + 3] * B[48 + c];
        double r4 = ((p0 + p1) + (p2 + p3)) + C[m * 16 + c];
        const double d = D[m * 16 + c];
        n++; seq_fma += d == r1; seq_fma_rev += d == r2; seq_muladd += d == r3; tree += d == r4;
      }
    }
    printf("%s operands: %ld results; equal to FMA chain k=0..3: %ld, k=3..0: %ld, mul+add chain: %ld, tree: %ld\n",
           mode == 0 ? "binary32-valued" : "full binary64", n, seq_fma, seq_fma_rev, seq_muladd, tree);
  }
  return 0;
}
