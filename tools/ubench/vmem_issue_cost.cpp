// vmem_issue_cost.cpp -- what one vector-memory wave-instruction costs a CU on gfx950, by kind: coalesced vs per-lane-addressed ("gather"),
// 4 / 8 / 16 bytes per lane, all lanes vs a few active, and the LDS (ds_read) equivalents.  All addresses stay inside a 32 KB window per
// workgroup (L1-resident), so the figure is the address / tag pipeline, not memory.  Found in round 4: the pixel reductions (SE3 step,
// EvaluateError) are bound by the NUMBER of per-lane-addressed loads, not by their bytes (profiles/r04_rowwalk_ablation.txt).
// build: hipcc -O3 --offload-arch=gfx950 -o vmem_issue_cost vmem_issue_cost.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// KIND: 0 coalesced b32, 1 coalesced b64, 2 coalesced b128, 3 gather b32, 4 gather b64, 5 gather b128,
//       6 gather b64 with 16 of 64 lanes active (one row), 7 gather b64 with every 4th lane active, 8 LDS gather b64, 9 LDS gather b128,
//       12 / 13 the same tap patterns out of LDS,
//       10 gather b64 overlapping neighbours (lane l reads bytes [4l, 4l + 8): the bilinear-tap pattern), 11 broadcast b32 (all lanes one address)
template <int KIND>
__global__ __launch_bounds__(256) void k_cost(const float* __restrict__ src, int iters, float* sink, const unsigned* __restrict__ perm) {
  __shared__ float lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = src[i];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = rsrc(src + (size_t)blockIdx.x * 8192, 32768);
  unsigned off;
  if (KIND <= 2) off = lane * (KIND == 0 ? 4 : KIND == 1 ? 8 : 16);
  else if (KIND == 10 || KIND == 12) off = lane * 4;
  else if (KIND == 13) off = lane * 8;
  else if (KIND == 11) off = 64;
  else off = (perm[threadIdx.x] % 1000u) * 16u;     // scattered over 16 KB, 16-byte aligned
  float acc = 0.f;
  const bool active = KIND == 6 ? lane < 16 : KIND == 7 ? (lane & 3) == 0 : true;
  if (active) {
    for (int it = 0; it < iters; ++it) {
      const unsigned o = off + (unsigned)(it & 7) * 1024u;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned a = o + u * 16u * ((KIND <= 2) ? 64u : 1u);
        if (KIND == 0 || KIND == 3 || KIND == 11) acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)a, 0, 0));
        else if (KIND == 1 || KIND == 4 || KIND == 6 || KIND == 7 || KIND == 10) { const f2 v = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)a, 0, 0)); acc += v.x + v.y; }
        else if (KIND == 2 || KIND == 5) { const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)a, 0, 0)); acc += v.x + v.w; }
        else if (KIND == 12) { const f2u v = *reinterpret_cast<const f2u*>(reinterpret_cast<const char*>(lds) + ((o + u * 16u) & 32767u)); acc += v.x + v.y; }
        else if (KIND == 13) { const f4u v = *reinterpret_cast<const f4u*>(reinterpret_cast<const char*>(lds) + ((o + u * 16u) & 32767u)); acc += v.x + v.w; }
        else if (KIND == 8) { const f2 v = *reinterpret_cast<const f2*>(reinterpret_cast<const char*>(lds) + (a & 32767u)); acc += v.x + v.y; }
        else { const f4 v = *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(lds) + (a & 32767u)); acc += v.x + v.w; }
      }
    }
  }
  if (acc == 123.456f) *sink = acc;
}

template <int KIND>
void run(const char* name, const float* src, float* sink, const unsigned* perm) {
  const int grid = 256 * 4, iters = 2000;   // 4 workgroups (16 waves) per CU
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_cost<KIND>, dim3(grid), dim3(256), 0, 0, src, iters, sink, perm);
  CK(hipEventRecord(a));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_cost<KIND>, dim3(grid), dim3(256), 0, 0, src, iters, sink, perm);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double insts_per_cu = (double)iters * 8 * 16;        // wave-instructions per CU and launch (16 waves per CU)
  const double ns = ms * 1e6 / reps / insts_per_cu;
  printf("%-58s %7.2f ns per wave-instruction per CU  (%5.1f cycles at 2.1 GHz)\n", name, ns, ns * 2.1);
}

int main() {
  float *src, *sink; unsigned* perm;
  const size_t n = (size_t)1024 * 8192;
  CK(hipMalloc(&src, n * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&perm, 256 * 4)); CK(hipMemset(src, 0, n * 4));
  unsigned h[256]; unsigned s = 12345u;
  for (int i = 0; i < 256; ++i) { s = s * 1664525u + 1013904223u; h[i] = s >> 8; }
  CK(hipMemcpy(perm, h, sizeof h, hipMemcpyHostToDevice));
  run<0>("coalesced  4 B / lane (256 B)", src, sink, perm);
  run<1>("coalesced  8 B / lane (512 B)", src, sink, perm);
  run<2>("coalesced 16 B / lane (1 KiB)", src, sink, perm);
  run<11>("broadcast  4 B (all lanes one address)", src, sink, perm);
  run<10>("overlapping 8 B at 4-byte lane stride (bilinear taps)", src, sink, perm);
  run<3>("gather     4 B / lane", src, sink, perm);
  run<4>("gather     8 B / lane", src, sink, perm);
  run<5>("gather    16 B / lane", src, sink, perm);
  run<6>("gather     8 B / lane, lanes 0-15 active", src, sink, perm);
  run<7>("gather     8 B / lane, every 4th lane active", src, sink, perm);
  run<8>("LDS gather 8 B / lane (ds_read_b64)", src, sink, perm);
  run<9>("LDS gather 16 B / lane (ds_read_b128)", src, sink, perm);
  run<12>("LDS overlapping 8 B at 4-byte lane stride (ds_read2_b32)", src, sink, perm);
  run<13>("LDS overlapping 16 B at 8-byte lane stride (ds_read2_b64)", src, sink, perm);
  return 0;
}
