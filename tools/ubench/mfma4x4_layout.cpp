// Probe: operand / result layout of v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products per instruction).
// A[l] = 100 * blk + 10 * (l & 3) + 1, B[l] = blk + 0.1 * (l & 3) + 1: prints which (blk, i, j) each (register, lane) of D holds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x, blk = l >> 2, i = l & 3;
  const float a = 1.0f + i + 10.0f * blk;        // row value of block blk
  const float b = 100.0f * (1 + i);             // column value
  f32x4 c = { 0, 0, 0, 0 };
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // expected: D[blk][i][j] = (1 + i + 10 blk) * 100 (1 + j)
  int ok_rowreg = 1, ok_colreg = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const int blk = l >> 2, q = l & 3;
    const float row_in_reg = (1.0f + r + 10.0f * blk) * 100.0f * (1 + q);   // reg = row i, lane = column j
    const float col_in_reg = (1.0f + q + 10.0f * blk) * 100.0f * (1 + r);   // reg = column j, lane = row i
    if (h[l * 4 + r] != row_in_reg) ok_rowreg = 0;
    if (h[l * 4 + r] != col_in_reg) ok_colreg = 0;
  }
  std::printf("layout: reg=row,lane=col: %d   reg=col,lane=row: %d\n", ok_rowreg, ok_colreg);
  for (int l = 0; l < 8; ++l) std::printf("lane %d: %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
