// Read-only streaming ceiling of the box: every wave pulls 16 B/lane loads, UNROLL in flight, over a buffer far beyond the
// Infinity Cache.  Gives the attainable bound the step kernel's 5.9 TB/s should be priced against (the guide's 6.29 TB/s is a
// copy, i.e. read+write).   hipcc -O3 --offload-arch=gfx950 read_bw.cpp -o read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_read(const v4f* __restrict__ src, size_t n_vec, float* sink) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i + (UNROLL - 1) * stride < n_vec; i += UNROLL * stride) {
    v4f v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 123.456f) *sink = acc;
}

template <int UNROLL, bool NT>
double run(const v4f* src, size_t n_vec, float* sink, int grid) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 60; ++i) k_read<UNROLL, NT><<<grid, 256>>>(src, n_vec, sink);
  CK(hipEventRecord(a));
  const int reps = 40;
  for (int i = 0; i < reps; ++i) k_read<UNROLL, NT><<<grid, 256>>>(src, n_vec, sink);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return (double)n_vec * 16 * reps / (ms * 1e-3) / 1e12;
}

int main() {
  size_t bytes = 6ull << 30;
  v4f* src; float* sink;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(src, 0, bytes));
  size_t n = bytes / 16;
  for (int grid : {1024, 2048, 4096, 8192}) {
    printf("grid %5d  unroll4 %.3f  unroll8 %.3f  unroll8-nt %.3f  unroll16-nt %.3f TB/s\n", grid, run<4, false>(src, n, sink, grid),
           run<8, false>(src, n, sink, grid), run<8, true>(src, n, sink, grid), run<16, true>(src, n, sink, grid));
  }
  return 0;
}
