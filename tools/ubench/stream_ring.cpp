// stream_ring.cpp -- can the memory floor of k_sfm_step's operand stream move?  
// The step kernel keeps 12 waves per CU (three workgroups of four), each with a 16-register-pair ring = 8 KB of the code-Jacobian stream in
// flight, refilled behind the consumer (nt).  With ALL arithmetic removed that pattern runs at 6.43 TB/s (profiles/r03_power_cap.txt); a
// read-only sweep at full occupancy reaches 6.7-6.9 (profiles/r02_ubench_read_bw.txt).  This program reproduces the kernel's occupancy and
// chunk walk with nothing but the stream and varies the transport:
//   V   the kernel's ring: 16 x 8 B per lane to VGPRs (512-byte wave loads), 8 KB per wave in flight
//   W   the same bytes as 8 x 16 B per lane (1 KiB wave loads)
//   X   twice the ring (16 x 16 B = 16 KB per wave: what 32 more registers would buy if the kernel had them)
//   L8  LDS-DMA (buffer_load_dwordx4 ... lds): 8 slots of 1 KiB per wave in LDS, consumer = one ds_read_b128 per slot, 8 KB per wave in flight
//   L10 10 slots (120 KB of LDS per CU: more than the step kernel could ever spare), L4 4 slots
//   H   hybrid = what 7 (2b) proposed: the VGPR ring V plus two LDS slots per wave (24 KB per CU, the LDS idle during the loop): every fifth KB
//       of the stream goes through LDS-DMA, 10 KB per wave in flight
// Every variant reads the same 128 x 640 x 480 x 128 B = 5.03 GB, once, with the non-temporal policy; a workgroup owns a contiguous range of
// the stream (as in the kernel), each of its four waves a contiguous quarter of it.
// build: hipcc -O3 --offload-arch=gfx950 -o stream_ring stream_ring.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int kChunk = 8192;       // bytes per wave and step (64 pixels x 32 floats)
constexpr int kWavesPerWG = 4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// ---- VGPR rings -------------------------------------------------------------------------------------------------------------------
// (no conditional inside the unrolled body: with one, hipcc renames the ring into a second register set and counts vmcnt down to 0 every round)
template <int VEC, int SLOTS, int AUX>   // VEC = dwords per lane and load (2 or 4); SLOTS loads in flight = one round of SLOTS * 256 * VEC bytes
__global__ __launch_bounds__(256, 3) void k_vgpr(const char* __restrict__ src, int chunks_per_wg, unsigned* sink, int total_kb_per_wave) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t wg_base = (size_t)blockIdx.x * chunks_per_wg * kChunk;
  const char* base = src + wg_base + (size_t)wave * total_kb_per_wave * 1024;
  typedef typename std::conditional<VEC == 2, u2, u4>::type V;
  constexpr int kRound = SLOTS * 256 * VEC;
  const int rounds = total_kb_per_wave * 1024 / kRound;
  V ring[SLOTS];
  unsigned acc = 0;
  const unsigned lane_off = lane * 4u * VEC;
  auto issue = [&](int r, int s) -> V {
    const __amdgpu_buffer_rsrc_t rs = rsrc(base + (size_t)r * kRound, kRound);
    if constexpr (VEC == 2) return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)lane_off, s * 256 * VEC, AUX));
    else return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_off, s * 256 * VEC, AUX));
  };
  auto eat = [&](const V& v) { if constexpr (VEC == 2) acc ^= v.x ^ v.y; else acc ^= v.x ^ v.y ^ v.z ^ v.w; };
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) ring[s] = issue(0, s);
  for (int r = 1; r < rounds; ++r) {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) { eat(ring[s]); ring[s] = issue(r, s); __builtin_amdgcn_sched_barrier(0); }   // refill right behind the consumer
  }
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) eat(ring[s]);
  if (acc == 0x12345u) *sink = acc;
}

// ---- LDS-DMA ring (+ optional VGPR ring beside it) ----------------------------------------------------------------------------------
// One slot = 1 KiB = one buffer_load_dwordx4 ... lds of the wave.  The consumer reads a slot back with one ds_read_b128 per lane.  Waits are
// written by hand (the compiler drains vmcnt to 0 at the first LDS access it can see behind an LDS-DMA): the ds_read is inline assembly.
template <int LSLOTS, int VSLOTS, int AUX>   // VSLOTS = 0: everything through LDS.  VSLOTS = 16: hybrid, per 10 KB: 8 KB as 16 x b64 to VGPRs + 2 KB to LDS
__global__ __launch_bounds__(256, 3) void k_lds(const char* __restrict__ src, int chunks_per_wg, unsigned* sink, int total_kb_per_wave) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* my = lds + wave * LSLOTS * 1024;
  // this wave's stream: a contiguous range of the workgroup's bytes (the DMA does not care about chunk order; bytes and in-flight depth are what is measured)
  const size_t wg_base = (size_t)blockIdx.x * chunks_per_wg * kChunk;
  const size_t wave_bytes = (size_t)total_kb_per_wave * 1024;
  const char* base = src + wg_base + (size_t)wave * wave_bytes;
  unsigned acc = 0;
  const unsigned lane16 = lane * 16u, lane8 = lane * 8u;
  auto dma = [&](int slot, size_t kb) {      // 1 KiB at byte offset kb * 1024 of the wave's range -> LDS slot
    const __amdgpu_buffer_rsrc_t r = rsrc(base + kb * 1024, 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(my + slot * 1024), 16, (int)lane16, 0, 0, AUX);
  };
  auto eat = [&](int slot) {
    u4 v;
    const unsigned a = (unsigned)(size_t)(my + slot * 1024) + lane16;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  };
  if constexpr (VSLOTS == 0) {
    const int n = total_kb_per_wave;          // KiB = DMA pieces of this wave
#pragma unroll
    for (int s = 0; s < LSLOTS; ++s) dma(s, s);
    int q = LSLOTS;
    for (; q + LSLOTS <= n; q += LSLOTS) {
#pragma unroll
      for (int s = 0; s < LSLOTS; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LSLOTS - 1) : "memory");   // the oldest piece has landed
        eat(s);
        dma(s, q + s);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < LSLOTS; ++s) eat(s);
    // (the ragged end, n % LSLOTS pieces, is not read: n is a multiple of LSLOTS in every launch below)
  } else {
    // hybrid: groups of 10 KiB: pieces 0..7 as 16 b64 wave loads into the VGPR ring, pieces 8, 9 through LDS slots 0, 1.  The VGPR loads are
    // inline assembly as well: behind an LDS-DMA hipcc's own count for a VGPR load ignores the DMA pieces queued in between (it waits
    // vmcnt(15) where 17 are younger), which would shorten the queue this variant is about.
    static_assert(VSLOTS == 16 && LSLOTS == 2, "hybrid shape");
    u2 ring[16];
    const int groups = total_kb_per_wave / 10;
    auto vload = [&](int g, int s, u2& dst) {
      const unsigned long long pa = (unsigned long long)(base + (size_t)g * 10240);
      u4 r;   // the buffer descriptor's four words, wave-uniform
      r.x = __builtin_amdgcn_readfirstlane((unsigned)pa); r.y = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu); r.z = 8192u; r.w = 0x00020000u;
      if constexpr (AUX == 2) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen nt" : "=v"(dst) : "v"(lane8), "s"(r), "s"(s * 512) : "memory");
      else asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(lane8), "s"(r), "s"(s * 512) : "memory");
    };
    auto landed = [&](u2& v) { asm volatile("s_waitcnt vmcnt(17)" : "+v"(v) :: "memory"); };   // 17 younger pieces may stay in flight
    // issue order per group: 16 VGPR loads then 2 DMA pieces -> in steady state 18 memory instructions (10 KB) per wave in flight
#pragma unroll
    for (int s = 0; s < 16; ++s) vload(0, s, ring[s]);
    dma(0, 8); dma(1, 9);
    for (int g = 1; g < groups; ++g) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        // ring[s] of group g-1 is the oldest outstanding: behind it 15 - s VGPR loads + 2 DMA pieces of g-1 and s VGPR loads of g = 17 younger
        landed(ring[s]);
        acc ^= ring[s].x ^ ring[s].y;
        vload(g, s, ring[s]);
      }
      asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); eat(0); dma(0, (size_t)g * 10 + 8);
      asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); eat(1); dma(1, (size_t)g * 10 + 9);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 16; ++s) { asm volatile("" : "+v"(ring[s])); acc ^= ring[s].x ^ ring[s].y; }
    eat(0); eat(1);
  }
  if (acc == 0x12345u) *sink = acc;
}

struct Result { double us, tbs; };
template <typename F> Result time_it(F launch, double bytes) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipDeviceSynchronize());
  double best = 1e30, sum = 0; const int reps = 12;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); sum += ms; if (ms < best) best = ms;
  }
  CK(hipGetLastError());
  return { sum / reps * 1e3, bytes / (sum / reps * 1e-3) / 1e12 };
}

int main(int argc, char** argv) {
  const int pairs = argc > 1 ? atoi(argv[1]) : 128;
  const size_t total = (size_t)pairs * 640 * 480 * 128;          // the code-Jacobian stream of the headline configuration
  char* src; unsigned* sink;
  CK(hipMalloc(&src, total + (1 << 20))); CK(hipMalloc(&sink, 4)); CK(hipMemset(src, 0, total + (1 << 20)));
  const size_t chunks = total / kChunk;                           // 614 400
  printf("stream %.3f GB, %zu chunks of 8 KB; three workgroups of four waves per CU (launch bounds 256, 3)\n", total / 1e9, chunks);
  for (int wgs_per_pair : {30, 60, 120}) {                        // 30 = the step kernel's shape for batches (40-chunk waves), 120 = short waves
    const int grid = pairs * wgs_per_pair;
    const int cpw = (int)(chunks / grid);                         // chunks per workgroup
    const int kb_per_wave = cpw * 8 / 4;                          // LDS variants: the workgroup's bytes split evenly over its four waves
    printf("-- %d workgroups (%d per pair), %d chunks per workgroup\n", grid, wgs_per_pair, cpw);
    auto line = [&](const char* name, Result r) { printf("   %-62s %8.1f us  %6.3f TB/s\n", name, r.us, r.tbs); };
    const double bytes = (double)grid * cpw * kChunk;
    line("V   VGPR ring 16 x b64 (8 KB / wave), nt   [the kernel's]", time_it([&] { hipLaunchKernelGGL((k_vgpr<2, 16, 2>), dim3(grid), dim3(256), 0, 0, src, cpw, sink, kb_per_wave); }, bytes));
    line("V0  the same, default cache policy", time_it([&] { hipLaunchKernelGGL((k_vgpr<2, 16, 0>), dim3(grid), dim3(256), 0, 0, src, cpw, sink, kb_per_wave); }, bytes));
    line("W   VGPR ring 8 x b128 (8 KB / wave), nt", time_it([&] { hipLaunchKernelGGL((k_vgpr<4, 8, 2>), dim3(grid), dim3(256), 0, 0, src, cpw, sink, kb_per_wave); }, bytes));
    line("X   VGPR ring 16 x b128 (16 KB / wave), nt", time_it([&] { hipLaunchKernelGGL((k_vgpr<4, 16, 2>), dim3(grid), dim3(256), 0, 0, src, cpw, sink, kb_per_wave); }, bytes));
    if (kb_per_wave % 8 == 0)
      line("L8  LDS-DMA 8 x 1 KiB slots / wave (96 KB LDS / CU), nt", time_it([&] { hipLaunchKernelGGL((k_lds<8, 0, 2>), dim3(grid), dim3(256), 4 * 8 * 1024, 0, src, cpw, sink, kb_per_wave); }, bytes));
    if (kb_per_wave % 10 == 0)
      line("L10 LDS-DMA 10 x 1 KiB slots / wave (120 KB LDS / CU), nt", time_it([&] { hipLaunchKernelGGL((k_lds<10, 0, 2>), dim3(grid), dim3(256), 4 * 10 * 1024, 0, src, cpw, sink, kb_per_wave); }, bytes));
    if (kb_per_wave % 4 == 0)
      line("L4  LDS-DMA 4 x 1 KiB slots / wave (48 KB LDS / CU), nt", time_it([&] { hipLaunchKernelGGL((k_lds<4, 0, 2>), dim3(grid), dim3(256), 4 * 4 * 1024, 0, src, cpw, sink, kb_per_wave); }, bytes));
    if (kb_per_wave % 10 == 0)
      line("H   hybrid: VGPR ring 16 x b64 + 2 LDS slots (10 KB / wave), nt", time_it([&] { hipLaunchKernelGGL((k_lds<2, 16, 2>), dim3(grid), dim3(256), 4 * 2 * 1024, 0, src, cpw, sink, kb_per_wave); }, bytes));
  }
  return 0;
}
