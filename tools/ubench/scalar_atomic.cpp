// Probe: does gfx950 execute SMEM atomics (s_atomic_add ... glc: returns the old value in an SGPR, tracked by lgkmcnt, NOT vmcnt)?
// Every wave pops `per_wave` tickets from one counter; all tickets must be distinct and the counter must end at waves * per_wave.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* q, unsigned* out, int per_wave) {
  const int wave = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  for (int i = 0; i < per_wave; ++i) {
    unsigned v = 1u;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(q) : "memory");
    if ((threadIdx.x & 63) == 0) out[(size_t)wave * per_wave + i] = v;
  }
}
int main() {
  const int wgs = 1024, per = 64, waves = wgs * 4;
  unsigned *q, *out;
  hipMalloc(&q, 4); hipMemset(q, 0, 4);
  hipMalloc(&out, (size_t)waves * per * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, q, out, per);
  hipEventRecord(b, 0);
  if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned fin; hipMemcpy(&fin, q, 4, hipMemcpyDeviceToHost);
  std::vector<unsigned> h((size_t)waves * per);
  hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  bool ok = fin == (unsigned)(waves * per);
  for (size_t i = 0; i < h.size(); ++i) ok = ok && h[i] == i;
  std::printf("s_atomic_add: final %u (expect %d), tickets distinct and dense: %s; %.1f us for %d pops = %.1f pops/us on ONE word\n", fin, waves * per, ok ? "yes" : "NO", ms * 1e3,
              waves * per, waves * per / (ms * 1e3));
  return ok ? 0 : 1;
}
