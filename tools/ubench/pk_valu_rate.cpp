// Microbenchmark: issue rate of packed fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) vs their scalar forms on
// gfx950, 1 and 2 waves per SIMD.  Decides whether a 2-pixels-per-lane phase A (packed math) could shorten the VALU time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int ROLE>
__device__ __forceinline__ float work(int iters, float seed) {
  if (ROLE == 0) {
    float v[8];
    for (int a = 0; a < 8; ++a) v[a] = seed + a;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int a = 0; a < 8; ++a) v[a] = __builtin_fmaf(v[a], 1.0001f, 0.5f);
    float s = 0; for (int a = 0; a < 8; ++a) s += v[a];
    return s;
  } else if (ROLE == 1) {
    f32x2 v[8];
    for (int a = 0; a < 8; ++a) v[a] = f32x2{ seed + a, seed - a };
    const f32x2 m{ 1.0001f, 0.9999f }, c{ 0.5f, 0.25f };
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int a = 0; a < 8; ++a) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[a]) : "v"(m), "v"(c));
    float s = 0; for (int a = 0; a < 8; ++a) s += v[a].x + v[a].y;
    return s;
  } else if (ROLE == 2) {
    f32x2 v[8];
    for (int a = 0; a < 8; ++a) v[a] = f32x2{ seed + a, seed - a };
    const f32x2 m{ 1.0001f, 0.9999f };
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int a = 0; a < 8; ++a) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[a]) : "v"(m));
    float s = 0; for (int a = 0; a < 8; ++a) s += v[a].x + v[a].y;
    return s;
  } else if (ROLE == 3) {
    f32x2 v[8];
    for (int a = 0; a < 8; ++a) v[a] = f32x2{ seed + a, seed - a };
    const f32x2 m{ 1.0001f, 0.9999f };
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int a = 0; a < 8; ++a) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[a]) : "v"(m));
    float s = 0; for (int a = 0; a < 8; ++a) s += v[a].x + v[a].y;
    return s;
  } else if (ROLE == 4) {   // scalar-form mul (e32)
    float v[8];
    for (int a = 0; a < 8; ++a) v[a] = seed + a;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int a = 0; a < 8; ++a) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[a]) : "v"(1.0001f));
    float s = 0; for (int a = 0; a < 8; ++a) s += v[a];
    return s;
  }
  return 0.f;
}

template <int R, int WAVES>
__global__ __launch_bounds__(WAVES * 256) void k(int iters, float* out) {
  const float r = work<R>(iters, 1.0f + threadIdx.x * 1e-3f);
  if (r == 123.456f) out[threadIdx.x] = r;
}

template <int R, int WAVES>
float run(int iters, float* d) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<R, WAVES>), dim3(256), dim3(WAVES * 256), 0, 0, iters, d);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  hipLaunchKernelGGL((k<R, WAVES>), dim3(256), dim3(WAVES * 256), 0, 0, iters, d);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f;
}

int main() {
  float* d; hipMalloc(&d, 8192);
  const int it = 20000;
  std::printf("per wave: %d x 48 instructions; 256 workgroups; columns: 1 wave/SIMD, 2 waves/SIMD, 4 waves/SIMD [us]\n", it);
  std::printf("v_fma_f32     %9.1f %9.1f %9.1f\n", run<0, 1>(it, d), run<0, 2>(it, d), run<0, 4>(it, d));
  std::printf("v_pk_fma_f32  %9.1f %9.1f %9.1f\n", run<1, 1>(it, d), run<1, 2>(it, d), run<1, 4>(it, d));
  std::printf("v_mul_f32     %9.1f %9.1f %9.1f\n", run<4, 1>(it, d), run<4, 2>(it, d), run<4, 4>(it, d));
  std::printf("v_pk_mul_f32  %9.1f %9.1f %9.1f\n", run<2, 1>(it, d), run<2, 2>(it, d), run<2, 4>(it, d));
  std::printf("v_pk_add_f32  %9.1f %9.1f %9.1f\n", run<3, 1>(it, d), run<3, 2>(it, d), run<3, 4>(it, d));
  return 0;
}
