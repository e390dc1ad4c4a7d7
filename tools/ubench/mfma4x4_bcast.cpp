// Probe: CBSZ / ABID (A-matrix block broadcast) of v_mfma_f32_4x4x1_16B_f32 on gfx950.
// Hypothesis: within every aligned group of 2^CBSZ blocks, the A operand of block number ABID of the group is used by all
// blocks of the group; B is untouched.  D[blk][i][j] = A[bsrc(blk)][i] * B[blk][j], bsrc = (blk & ~(2^CBSZ - 1)) + ABID.
// Prints 1 per (cbsz, abid) when the hypothesis holds for every (lane, register).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void k(float* out) {
  const int l = threadIdx.x, blk = l >> 2, i = l & 3;
  const float a = 1.0f + i + 10.0f * blk;
  const float b = 100.0f * (1 + i) + 1000.0f * blk;
  f32x4 c = { 0, 0, 0, 0 };
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CBSZ, ABID, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

template <int CBSZ, int ABID>
static void run(float* d) {
  hipLaunchKernelGGL((k<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d);
  float h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int blk = l >> 2, j = l & 3;
      const int src = (blk & ~((1 << CBSZ) - 1)) + ABID;
      const float expect = (1.0f + r + 10.0f * src) * (100.0f * (1 + j) + 1000.0f * blk);   // reg = row i, lane = column j
      if (h[l * 4 + r] != expect) ok = 0;
    }
  std::printf("cbsz=%d abid=%d: hypothesis %s   lane0 %g %g  lane4 %g %g  lane8 %g lane12 %g\n", CBSZ, ABID, ok ? "HOLDS" : "FAILS", h[0], h[1], h[16], h[17],
              h[32], h[48]);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 4);
  run<0, 0>(d);
  run<1, 0>(d);
  run<1, 1>(d);
  run<2, 0>(d);
  run<2, 1>(d);
  run<2, 2>(d);
  run<2, 3>(d);
  run<3, 5>(d);
  run<4, 9>(d);
  return 0;
}
