// Microbenchmark: do MFMA and VALU work of two co-resident waves on one SIMD overlap on gfx950?
// 512-thread workgroups = 2 waves per SIMD.  Waves 0-3 run role A, waves 4-7 run role B (or idle).
// Roles: 0 idle, 1 fp32 MFMA 16x16x4 (6 independent accumulators), 2 bf16 MFMA 16x16x32, 3 VALU fp32 fma chain x8 independent.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int ROLE>
__device__ __forceinline__ float work(int iters, float seed) {
  if (ROLE == 1) {
    f32x4 acc[6];
    for (int a = 0; a < 6; ++a) acc[a] = f32x4{ seed, 0, 0, 0 };
    float x = seed, y = seed * 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int a = 0; a < 6; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0; for (int a = 0; a < 6; ++a) s += acc[a][0] + acc[a][3];
    return s;
  } else if (ROLE == 2) {
    f32x4 acc[6];
    for (int a = 0; a < 6; ++a) acc[a] = f32x4{ seed, 0, 0, 0 };
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(seed + e); y[e] = (__bf16)(seed - e); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int a = 0; a < 6; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0; for (int a = 0; a < 6; ++a) s += acc[a][0] + acc[a][3];
    return s;
  } else if (ROLE == 3) {
    float v[8];
    for (int a = 0; a < 8; ++a) v[a] = seed + a;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int a = 0; a < 8; ++a) v[a] = __builtin_fmaf(v[a], 1.0001f, 0.5f);   // 48 VALU per iteration
    }
    float s = 0; for (int a = 0; a < 8; ++a) s += v[a];
    return s;
  }
  else if (ROLE == 4 || ROLE == 5 || ROLE == 6 || ROLE == 7) {
    // same-wave interleave: per MFMA, F independent VALU fmas (ROLE 4: f32 MFMA + 6, 5: bf16 MFMA + 3, 6: f32 MFMA + 3, 7: bf16 + 6)
    constexpr int F = (ROLE == 4 || ROLE == 7) ? 6 : 3;
    f32x4 acc[6];
    for (int a = 0; a < 6; ++a) acc[a] = f32x4{ seed, 0, 0, 0 };
    float v[6];
    for (int a = 0; a < 6; ++a) v[a] = seed + a;
    float x = seed, y = seed * 0.5f;
    bf16x8 xb, yb;
    for (int e = 0; e < 8; ++e) { xb[e] = (__bf16)(seed + e); yb[e] = (__bf16)(seed - e); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        if (ROLE == 4 || ROLE == 6) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
        else acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, acc[a], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < F; ++f) v[f] = __builtin_fmaf(v[f], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float s = 0; for (int a = 0; a < 6; ++a) s += acc[a][0] + acc[a][3] + v[a];
    return s;
  }
  return 0.f;
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void k(int iters, float* out) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float r;
  if (wave < 4) r = work<RA>(iters, 1.0f + threadIdx.x * 1e-3f);
  else r = work<RB>(iters, 2.0f + threadIdx.x * 1e-3f);
  if (r == 123.456f) out[threadIdx.x] = r;
}

template <int RA, int RB>
float run(int iters, float* d) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, iters, d);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, iters, d);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f;
}

int main() {
  float* d; hipMalloc(&d, 4096);
  const int it = 20000;
  std::printf("per-wave work: role1/2 = %d x 6 MFMA, role3 = %d x 48 VALU fma; 256 WGs x 8 waves (2 waves/SIMD)\n", it, it);
  std::printf("f32mfma alone      %8.1f us\n", run<1, 0>(it, d));
  std::printf("bf16mfma alone     %8.1f us\n", run<2, 0>(it, d));
  std::printf("valu alone         %8.1f us\n", run<3, 0>(it, d));
  std::printf("f32mfma + valu     %8.1f us\n", run<1, 3>(it, d));
  std::printf("bf16mfma + valu    %8.1f us\n", run<2, 3>(it, d));
  std::printf("f32mfma + f32mfma  %8.1f us\n", run<1, 1>(it, d));
  std::printf("valu + valu        %8.1f us\n", run<3, 3>(it, d));
  std::printf("bf16mfma + bf16mfma%8.1f us\n", run<2, 2>(it, d));
  std::printf("same wave: f32mfma+6valu/mfma (1 wave/SIMD) %8.1f us  (valu part alone would be %d x 36)\n", run<4, 0>(it, d), it);
  std::printf("same wave: f32mfma+3valu/mfma              %8.1f us\n", run<6, 0>(it, d));
  std::printf("same wave: bf16mfma+3valu/mfma             %8.1f us\n", run<5, 0>(it, d));
  std::printf("same wave: bf16mfma+6valu/mfma             %8.1f us\n", run<7, 0>(it, d));
  std::printf("2 waves/SIMD both f32mfma+6valu            %8.1f us\n", run<4, 4>(it, d));
  std::printf("2 waves/SIMD both bf16mfma+3valu           %8.1f us\n", run<5, 5>(it, d));
  return 0;
}
