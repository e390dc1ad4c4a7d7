// host_scatter.cpp -- what does it cost a finalize kernel to scatter a 4 KB result item, float by float, straight into pinned mapped HOST memory
// (k_sfm_finalize_b3: ~1000 four-byte stores from 6 workgroups) instead of into device memory, or instead of one coalesced 16-byte-per-lane copy?
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/host_scatter.cpp -o /tmp/host_scatter && /tmp/host_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

__global__ void k_scatter(float* out, const float* in) {   // 6 workgroups x 256 threads: thread t of workgroup b stores element (t * 37 + b * 173) % 1036
  const int e = (threadIdx.x * 37 + blockIdx.x * 173) % 1036;
  if (threadIdx.x < 173) out[e] = in[threadIdx.x] + 1.f;
}
__global__ void k_coalesced(float4* out, const float4* in) {   // one workgroup, 259 lanes x 16 bytes
  if (threadIdx.x < 259) out[threadIdx.x] = in[threadIdx.x];
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  float *dev_in, *dev_out, *host_out, *host_out_d;
  (void)hipMalloc(&dev_in, 8192); (void)hipMemset(dev_in, 0, 8192); (void)hipMalloc(&dev_out, 8192);
  (void)hipHostMalloc((void**)&host_out, 8192, hipHostMallocMapped | hipHostMallocCoherent);
  (void)hipHostGetDevicePointer((void**)&host_out_d, host_out, 0);
  hipStream_t s; (void)hipStreamCreate(&s);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* names[4] = { "scatter 1036 x 4 B -> device memory", "scatter 1036 x 4 B -> mapped host memory", "coalesced 259 x 16 B -> device memory", "coalesced 259 x 16 B -> mapped host memory" };
  for (int v = 0; v < 4; ++v) {
    float* out = (v & 1) ? host_out_d : dev_out;
    for (int i = 0; i < 20; ++i) { if (v < 2) hipLaunchKernelGGL(k_scatter, dim3(6), dim3(256), 0, s, out, dev_in); else hipLaunchKernelGGL(k_coalesced, dim3(1), dim3(320), 0, s, (float4*)out, (const float4*)dev_in); }
    (void)hipStreamSynchronize(s);
    // back-to-back kernels: the stream's throughput per kernel = launch floor + the kernel's own time
    const int N = 2000;
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < N; ++i) { if (v < 2) hipLaunchKernelGGL(k_scatter, dim3(6), dim3(256), 0, s, out, dev_in); else hipLaunchKernelGGL(k_coalesced, dim3(1), dim3(320), 0, s, (float4*)out, (const float4*)dev_in); }
    (void)hipEventRecord(e1, s);
    (void)hipStreamSynchronize(s);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::printf("%-44s %6.2f us per kernel (back to back)\n", names[v], ms * 1e3 / N);
  }
  return 0;
}
