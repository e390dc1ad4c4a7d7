// wave_fold_probe.cpp -- pins the semantics the pixel reductions' wave fold relies on (dfx_misc_kernels.hip wave_fold4):
// v_permlane32_swap / v_permlane16_swap (gfx950) and DPP row_shr with bound_ctrl.  Prints, for every register of the fold and every
// 16-lane row, which value's 64-lane total lane 15 of the row holds.
// build: hipcc -O3 --offload-arch=gfx950 -o wave_fold_probe wave_fold_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL>
__device__ __forceinline__ float row_shr_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// hipcc (ROCm 7.2, clang 22) folds `s[0] + s[1]` of a permlane swap's two results into `s[0] + s[0]`: keep the second result opaque
__device__ __forceinline__ float swap_sum(unsigned a, unsigned b) {
  float fb = __builtin_bit_cast(float, b);
  asm volatile("" : "+v"(fb));
  return __builtin_bit_cast(float, a) + fb;
}
template <int NQ>
__device__ __forceinline__ void wave_fold4(const float (&v)[4 * NQ], float (&out)[NQ]) {
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    float w[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const auto s = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[4 * j + 2 * h]), __builtin_bit_cast(unsigned, v[4 * j + 2 * h + 1]), false, false);
      w[h] = swap_sum(s[0], s[1]);
    }
    const auto s = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, w[0]), __builtin_bit_cast(unsigned, w[1]), false, false);
    float r = swap_sum(s[0], s[1]);
    r = row_shr_add<0x118>(r);
    r = row_shr_add<0x114>(r);
    r = row_shr_add<0x112>(r);
    r = row_shr_add<0x111>(r);
    out[j] = r;
  }
}

__global__ void k_probe(float* out, float* swap32, float* swap16) {
  const int lane = threadIdx.x;
  float v[8], f[2];
  for (int q = 0; q < 8; ++q) v[q] = (float)((q + 1) * 1000) + (float)lane;   // total of value q = 64000 (q + 1) + 2016
  wave_fold4<2>(v, f);
  out[lane] = f[0]; out[64 + lane] = f[1];
  // raw swaps of lane-tagged registers: a = lane, b = 100 + lane
  const auto s = __builtin_amdgcn_permlane32_swap((unsigned)lane, (unsigned)(100 + lane), false, false);
  swap32[lane] = (float)s[0]; swap32[64 + lane] = (float)s[1];
  const auto t = __builtin_amdgcn_permlane16_swap((unsigned)lane, (unsigned)(100 + lane), false, false);
  swap16[lane] = (float)t[0]; swap16[64 + lane] = (float)t[1];
}

int main() {
  float *d, *s32, *s16;
  hipMalloc(&d, 128 * 4); hipMalloc(&s32, 128 * 4); hipMalloc(&s16, 128 * 4);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, s32, s16);
  std::vector<float> h(128), a(128), b(128);
  hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost); hipMemcpy(a.data(), s32, 512, hipMemcpyDeviceToHost); hipMemcpy(b.data(), s16, 512, hipMemcpyDeviceToHost);
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 4; ++r) {
      const float t = h[64 * j + 16 * r + 15];
      printf("fold register %d row %d lane 15: %.0f -> value %.3f (expected an integer: (t - 2016) / 64000 - 1)\n", j, r, t, (t - 2016.f) / 64000.f - 1.f);
    }
  printf("permlane32_swap(a = lane, b = 100 + lane): new a, lanes 0 31 32 63: %.0f %.0f %.0f %.0f | new b: %.0f %.0f %.0f %.0f\n", a[0], a[31], a[32], a[63], a[64], a[95], a[96], a[127]);
  printf("permlane16_swap(a = lane, b = 100 + lane): new a, lanes 0 16 32 48: %.0f %.0f %.0f %.0f | new b: %.0f %.0f %.0f %.0f\n", b[0], b[16], b[32], b[48], b[64], b[80], b[96], b[112]);
  return 0;
}
