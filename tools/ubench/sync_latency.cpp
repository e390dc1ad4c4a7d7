// sync_latency.cpp -- how a blocking call learns that its last kernel is done: hipStreamSynchronize (the runtime's completion signal) against the host
// polling a sequence number the kernel itself stores into mapped host memory behind its result (system-scope release).  Per variant: time from the launch call
// to the moment the host holds the result, for an (almost) empty kernel and for two dependent kernels (a blocking operator here is step kernel + finalize).
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/sync_latency.cpp -o /tmp/sync_latency && /tmp/sync_latency
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdint>

__global__ void k_work(float* p, int n) { if (threadIdx.x < n) p[threadIdx.x] += 1.f; }
__global__ void k_final(const float* p, float* result, uint32_t* flag, uint32_t seq) {
  if (threadIdx.x == 0) {
    result[0] = p[0];
    __atomic_store_n(flag, seq, __ATOMIC_RELEASE);   // system scope: orders the result store in front of the flag for the host
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  float* d; (void)hipMalloc(&d, 4096); (void)hipMemset(d, 0, 4096);
  float* res; uint32_t* flag;
  (void)hipHostMalloc((void**)&res, 64, hipHostMallocMapped | hipHostMallocCoherent);
  (void)hipHostMalloc((void**)&flag, 64, hipHostMallocMapped | hipHostMallocCoherent);
  float* dres; uint32_t* dflag;
  (void)hipHostGetDevicePointer((void**)&dres, res, 0); (void)hipHostGetDevicePointer((void**)&dflag, flag, 0);
  hipStream_t s; (void)hipStreamCreate(&s);
  *flag = 0;
  uint32_t seq = 0;
  for (int variant = 0; variant < 2; ++variant) {
    for (int two = 0; two < 2; ++two) {
      double tot = 0, mn = 1e30;
      const int N = 2000;
      for (int i = 0; i < N + 50; ++i) {
        ++seq;
        const double t0 = now_us();
        if (two) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, d, 16);
        hipLaunchKernelGGL(k_final, dim3(1), dim3(64), 0, s, d, dres, dflag, seq);
        if (variant == 0) (void)hipStreamSynchronize(s);
        else while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
        const double us = now_us() - t0;
        if (variant == 1) (void)hipStreamSynchronize(s);   // (outside the measurement: keep the queue short)
        if (i >= 50) { tot += us; mn = us < mn ? us : mn; }
      }
      std::printf("%-28s %s: mean %6.2f us  min %6.2f us\n", variant == 0 ? "hipStreamSynchronize" : "poll flag in mapped memory", two ? "two kernels" : "one kernel ", tot / N, mn);
    }
  }
  // the flag written by the command processor behind the kernels (hipStreamWriteValue32): needs no change to any kernel
  for (int two = 0; two < 2; ++two) {
    double tot = 0, mn = 1e30;
    const int N = 2000;
    for (int i = 0; i < N + 50; ++i) {
      ++seq;
      const double t0 = now_us();
      if (two) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, d, 16);
      hipLaunchKernelGGL(k_final, dim3(1), dim3(64), 0, s, d, dres, dflag + 8, seq);
      if (hipStreamWriteValue32(s, dflag, seq, 0) != hipSuccess) { std::printf("hipStreamWriteValue32 failed\n"); break; }
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
      const double us = now_us() - t0;
      if (i >= 50) { tot += us; mn = us < mn ? us : mn; }
    }
    (void)hipStreamSynchronize(s);
    std::printf("%-28s %s: mean %6.2f us  min %6.2f us\n", "poll, hipStreamWriteValue32", two ? "two kernels" : "one kernel ", tot / N, mn);
  }
  // the polling variant WITHOUT the trailing synchronize: consecutive blocking calls queue behind each other's end-of-kernel processing
  {
    double tot = 0, mn = 1e30;
    const int N = 2000;
    for (int i = 0; i < N + 50; ++i) {
      ++seq;
      const double t0 = now_us();
      hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, d, 16);
      hipLaunchKernelGGL(k_final, dim3(1), dim3(64), 0, s, d, dres, dflag, seq);
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
      const double us = now_us() - t0;
      if (i >= 50) { tot += us; mn = us < mn ? us : mn; }
    }
    (void)hipStreamSynchronize(s);
    std::printf("%-28s %s: mean %6.2f us  min %6.2f us\n", "poll, calls back to back", "two kernels", tot / N, mn);
  }
  return 0;
}
