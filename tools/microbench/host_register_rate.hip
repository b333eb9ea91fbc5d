// host_register_rate.hip -- what pinning pageable memory in place costs, and whether it can run beside a DMA copy.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hrr tools/microbench/host_register_rate.hip && /tmp/hrr
// (1) hipHostRegister + hipHostUnregister of blocks of 4 .. 256 MB of touched pageable memory: ms and GB/s;
// (2) a 1 GB pageable -> device copy: plain hipMemcpy, against the same bytes registered block by block (16 MB) just ahead of
//     an asynchronous copy of the block before, unregistered two blocks behind.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const size_t total = (size_t)1 << 30;
  char *host = (char *)aligned_alloc(4096, total);
  memset(host, 1, total);
  void *dev;
  CK(hipMalloc(&dev, total));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (size_t mb : {4, 16, 64, 256}) {
    const size_t bytes = mb << 20;
    double t0 = now();
    int reps = 0;
    for (size_t off = 0; off + bytes <= total && reps < 8; off += bytes, reps++) CK(hipHostRegister(host + off, bytes, hipHostRegisterPortable));
    double t1 = now();
    for (int r = 0; r < reps; r++) CK(hipHostUnregister(host + (size_t)r * bytes));
    double t2 = now();
    printf("register %4zu MB: %.3f ms (%.1f GB/s), unregister %.3f ms\n", mb, (t1 - t0) / reps * 1e3, bytes / ((t1 - t0) / reps) / 1e9, (t2 - t1) / reps * 1e3);
  }
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    CK(hipMemcpy(dev, host, total, hipMemcpyHostToDevice));
    printf("pageable hipMemcpy 1 GB: %.1f ms (%.1f GB/s)\n", (now() - t0) * 1e3, total / (now() - t0) / 1e9);
  }
  for (size_t mb : {8, 16, 32, 64}) {
    const size_t blk = mb << 20, nb = total / blk;
    double t0 = now();
    std::vector<hipEvent_t> ev(nb);
    for (size_t b = 0; b < nb; b++) {
      CK(hipHostRegister(host + b * blk, blk, hipHostRegisterPortable));
      CK(hipMemcpyAsync((char *)dev + b * blk, host + b * blk, blk, hipMemcpyHostToDevice, s));
      CK(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
      CK(hipEventRecord(ev[b], s));
      if (b >= 2) { CK(hipEventSynchronize(ev[b - 2])); CK(hipHostUnregister(host + (b - 2) * blk)); }
    }
    CK(hipStreamSynchronize(s));
    for (size_t b = nb >= 2 ? nb - 2 : 0; b < nb; b++) CK(hipHostUnregister(host + b * blk));
    double dt = now() - t0;
    printf("register-as-you-go, %2zu MB blocks: %.1f ms (%.1f GB/s)\n", mb, dt * 1e3, total / dt / 1e9);
    for (auto e : ev) hipEventDestroy(e);
  }
  CK(hipHostRegister(host, total, hipHostRegisterPortable));
  double t0 = now();
  CK(hipMemcpyAsync(dev, host, total, hipMemcpyHostToDevice, s));
  CK(hipStreamSynchronize(s));
  printf("registered (pinned) copy 1 GB: %.1f ms (%.1f GB/s)\n", (now() - t0) * 1e3, total / (now() - t0) / 1e9);
  return 0;
}
