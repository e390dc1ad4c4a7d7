// Microbenchmark: what does ONE more instruction cost next to exact-fp32 MFMAs on a gfx950 SIMD?
//
// Every wave runs the same hand-placed stream (inline asm, program order = issue order):
//     repeat { v_mfma_f32_16x16x4_f32 (4 independent accumulators, round robin) ; K fillers of one kind }
// and brackets it with s_memtime.  Reported: core cycles per MFMA group as seen by a wave, for 1 / 2 / 4 waves per SIMD
// (256 / 512 / 1024-thread workgroups, one workgroup per CU forced by 100 KB of dynamic LDS), plus the SIMD-level figure
// cycles / waves (= the time one SIMD spends per MFMA group when all its waves run this stream).
//
// Filler kinds: 0 v_fma_f32 (8 independent chains), 1 v_mov_b32_dpp row_shr:8, 2 ds_read_b32, 3 v_mul_legacy_f32,
//               4 v_mfma_f32_4x4x1_16B_f32 (own accumulator), 5 s_mov (scalar), 6 mixed VALU: fma, dpp, mul_legacy round robin
// MODE 1 replaces the MFMA by nothing (fillers only) -> the fillers' solo issue rate.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/issue_cost.cpp -o gpurun_build/issue_cost
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ void filler(float (&f)[8], int j, float a, float b, f32x4& acc4, const float* lds_ptr, unsigned lds_off) {
  float& x = f[j & 7];
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
  else if (KIND == 1) asm volatile("v_mov_b32_dpp %0, %1 row_shr:8 row_mask:0xf bank_mask:0xc" : "+v"(x) : "v"(a));
  else if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(lds_off));
  else if (KIND == 3) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(x) : "v"(a));
  else if (KIND == 4) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc4) : "v"(a), "v"(b));
  else if (KIND == 5) asm volatile("s_mov_b32 s40, s41" ::: "s40");
  else {
    const int r = j % 3;
    if (r == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if (r == 1) asm volatile("v_mov_b32_dpp %0, %1 row_shr:8 row_mask:0xf bank_mask:0xc" : "+v"(x) : "v"(a));
    else asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(x) : "v"(a));
  }
}

template <int KIND, int K, int MODE>
__global__ __launch_bounds__(1024) void k_issue(int iters, unsigned long long* out, float seed) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  lds[threadIdx.x] = seed;
  __syncthreads();
  f32x4 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) acc[a] = f32x4{ seed, 0.f, 0.f, 0.f };
  f32x4 acc4 = f32x4{ 0.f, 0.f, 0.f, 0.f };
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = seed + j;
  float a = seed * 0.5f + lane * 1e-3f, b = seed * 0.25f;
  const unsigned lds_off = (unsigned)threadIdx.x * 4u;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {   // 8 MFMA groups per iteration
      if (MODE == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < K; ++j) filler<KIND>(f, u * K + j, a, b, acc4, lds, lds_off);
    }
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 7\n s_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = acc4[0];
#pragma unroll
  for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += f[j];
  if (s == 123.456f) out[0] = 1;   // keep everything alive
  if (lane == 0) out[1 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

// Two roles on one SIMD: waves 0-3 run MFMA only, waves 4-7 run fillers only (KIND), 512-thread workgroups.
template <int KIND>
__global__ __launch_bounds__(512) void k_roles(int iters, unsigned long long* out, float seed, int role_a, int role_b) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lds[threadIdx.x] = seed;
  __syncthreads();
  f32x4 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) acc[a] = f32x4{ seed, 0.f, 0.f, 0.f };
  f32x4 acc4 = f32x4{ 0.f, 0.f, 0.f, 0.f };
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = seed + j;
  float a = seed * 0.5f + lane * 1e-3f, b = seed * 0.25f;
  const unsigned lds_off = (unsigned)threadIdx.x * 4u;
  const int role = wave < 4 ? role_a : role_b;   // 0 idle, 1 MFMA, 2 fillers
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (role == 1) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
    }
  } else if (role == 2) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 48; ++u) filler<KIND>(f, u, a, b, acc4, lds, lds_off);
      if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_nop 7\n s_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = acc4[0];
#pragma unroll
  for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += f[j];
  if (s == 123.456f) out[0] = 1;
  if (lane == 0) out[1 + blockIdx.x * 16 + wave] = t1 - t0;
}

static unsigned long long* d_out;
static std::vector<unsigned long long> h_out(1 + 256 * 16);

static double median_cycles(int waves_per_wg, int first = 0, int last = -1) {
  if (last < 0) last = waves_per_wg;
  hipMemcpy(h_out.data(), d_out, h_out.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> v;
  for (int b = 0; b < 256; ++b)
    for (int w = first; w < last; ++w) v.push_back((double)h_out[1 + b * 16 + w]);
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

template <int KIND, int K, int MODE>
static void run_one(int iters, double* cyc /*[3]*/, double* us /*[3]*/) {
  const int wgs[3] = { 256, 512, 1024 };
  hipFuncSetAttribute((const void*)k_issue<KIND, K, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int c = 0; c < 3; ++c) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_issue<KIND, K, MODE>), dim3(256), dim3(wgs[c]), 100 * 1024, 0, iters, d_out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_issue<KIND, K, MODE>), dim3(256), dim3(wgs[c]), 100 * 1024, 0, iters, d_out, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    us[c] = ms * 1e3;
    cyc[c] = median_cycles(wgs[c] / 64) / ((double)iters * 8.0);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
}

template <int KIND, int MODE>
static void sweep(const char* name, int iters) {
  std::printf("# %s%s: cycles per group {1 MFMA + K fillers} per wave at 1 / 2 / 4 waves per SIMD | per SIMD (cycles / waves) | wall us\n", name,
              MODE ? " (fillers only, no MFMA)" : "");
  double c[3], u[3];
#define ROW(KK)                                                                                                         \
  run_one<KIND, KK, MODE>(iters, c, u);                                                                                 \
  std::printf("K=%d  wave: %7.1f %7.1f %7.1f | simd: %7.1f %7.1f %7.1f | us: %8.1f %8.1f %8.1f\n", KK, c[0], c[1], c[2], c[0], c[1] / 2, \
              c[2] / 4, u[0], u[1], u[2]);
  if (MODE == 0) { ROW(0) }
  ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(8)
#undef ROW
}

template <int KIND>
static void roles(const char* name, int iters) {
  hipFuncSetAttribute((const void*)k_roles<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const int combos[4][2] = { { 1, 0 }, { 0, 2 }, { 1, 2 }, { 2, 2 } };
  for (auto& cb : combos) {
    hipLaunchKernelGGL((k_roles<KIND>), dim3(256), dim3(512), 100 * 1024, 0, iters, d_out, 1.0f, cb[0], cb[1]);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_roles<KIND>), dim3(256), dim3(512), 100 * 1024, 0, iters, d_out, 1.0f, cb[0], cb[1]);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ca = median_cycles(8, 0, 4), cb2 = median_cycles(8, 4, 8);
    std::printf("roles %-12s A=%d B=%d  (1 = %d MFMA, 2 = %d fillers per wave): wave cycles A %10.0f  B %10.0f   wall %8.1f us\n", name, cb[0], cb[1],
                iters * 8, iters * 48, ca, cb2, ms * 1e3);
  }
}

int main() {
  hipMalloc(&d_out, h_out.size() * 8);
  hipMemset(d_out, 0, h_out.size() * 8);
  const int it = 4000;
  sweep<0, 0>("v_fma_f32", it);
  sweep<1, 0>("v_mov_b32_dpp", it);
  sweep<3, 0>("v_mul_legacy_f32", it);
  sweep<6, 0>("mixed VALU", it);
  sweep<2, 0>("ds_read_b32", it);
  sweep<4, 0>("v_mfma_f32_4x4x1", it);
  sweep<5, 0>("s_mov_b32", it);
  sweep<0, 1>("v_fma_f32", it);
  sweep<2, 1>("ds_read_b32", it);
  sweep<4, 1>("v_mfma_f32_4x4x1", it);
  roles<0>("v_fma_f32", it);
  roles<2>("ds_read_b32", it);
  return 0;
}
