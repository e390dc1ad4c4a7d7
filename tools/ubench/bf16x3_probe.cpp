// Probe for an exact bf16 three-way split of the step kernel's fp32 rank-1 updates (one wave, milliseconds):
//   1. operand / result layout of v_mfma_f32_16x16x32_bf16 as assumed by the design sketch:
//        A: lane l, slot s (8 bf16 in 4 VGPRs)  = A[row = l & 15][k = 8 * (l >> 4) + s]
//        B: lane l, slot s                      = B[k = 8 * (l >> 4) + s][col = l & 15]
//        D: lane l, register r                  = D[row = 4 * (l >> 4) + r][col = l & 15]
//      checked with asymmetric small-integer matrices (exact in bf16) against the host product;
//   2. the split itself: x = h + m + l with h = RNE_bf16(x), m = RNE_bf16(x - h), l = RNE_bf16(x - h - m) through
//      v_cvt_pk_bf16_f32 -- is the sum EXACTLY x for every input (normal range)?
//   3. Z = sum_k z_k z_k^T over 32 random fp32 vectors of 16 entries from six bf16 MFMAs (hh, hm, mh, hl, lh, mm; fp32 accumulate)
//      against the double-precision sum and against a sequential fp32 fmaf chain (what v_mfma_f32_16x16x4_f32 computes).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/bf16x3_probe.cpp -o gpurun_build/bf16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {   // {bf16(lo) in bits 0-15, bf16(hi) in bits 16-31}, round to nearest even
  unsigned r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// in: A_bits[16][32], B_bits[32][16] as bf16 bit patterns (uint16); out: D[16][16]
__global__ void k_layout(const uint16_t* A, const uint16_t* B, float* D) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  bf16x8 a, b;
  for (int s = 0; s < 8; ++s) { a[s] = (short)A[i * 32 + 8 * g + s]; b[s] = (short)B[(8 * g + s) * 16 + i]; }
  f32x4 c = { 0, 0, 0, 0 };
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
}

// split n floats (n even): pieces out as float h, m, l
__global__ void k_split(const float* x, int n, float* h, float* m, float* lo) {
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (t + 1 >= n + 1) return;
  const float x0 = x[t], x1 = x[t + 1];
  const unsigned ph = cvt_pk(x0, x1);
  const float r0 = x0 - bf_lo(ph), r1 = x1 - bf_hi(ph);
  const unsigned pm = cvt_pk(r0, r1);
  const float s0 = r0 - bf_lo(pm), s1 = r1 - bf_hi(pm);
  const unsigned pl = cvt_pk(s0, s1);
  h[t] = bf_lo(ph); h[t + 1] = bf_hi(ph);
  m[t] = bf_lo(pm); m[t + 1] = bf_hi(pm);
  lo[t] = bf_lo(pl); lo[t + 1] = bf_hi(pl);
}

// the same split with the remainders through v_dot2c_f32_bf16: r = x + (h0, h1) . (-1, 0) -- expand + subtract in ONE instruction per value
// (the step kernel's form since round 3: 7 instead of 9 vector-ALU instructions per pair of values)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_v(float lo, float hi) { const f32x2_t v = { lo, hi }; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)); }
__device__ __forceinline__ float sub_lo(float x, unsigned p) { return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, p), __builtin_bit_cast(bf16x2_t, 0x8000bf80u), x, false); }
__device__ __forceinline__ float sub_hi(float x, unsigned p) { return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, p), __builtin_bit_cast(bf16x2_t, 0xbf808000u), x, false); }
__global__ void k_split_dot2(const float* x, int n, float* h, float* m, float* lo) {
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (t + 1 >= n + 1) return;
  const float x0 = x[t], x1 = x[t + 1];
  const unsigned ph = cvt_pk_v(x0, x1);
  const float r0 = sub_lo(x0, ph), r1 = sub_hi(x1, ph);
  const unsigned pm = cvt_pk_v(r0, r1);
  const float s0 = sub_lo(r0, pm), s1 = sub_hi(r1, pm);
  const unsigned pl = cvt_pk_v(s0, s1);
  h[t] = bf_lo(ph); h[t + 1] = bf_hi(ph);
  m[t] = bf_lo(pm); m[t + 1] = bf_hi(pm);
  lo[t] = bf_lo(pl); lo[t + 1] = bf_hi(pl);
}

// z[32][16] fp32 (pixel k, entry i) -> Z[16][16] by six bf16 MFMAs; lane (i, g) owns pixels 8g .. 8g+7 (any pixel -> (lane group, slot)
// assignment works as long as A and B use the same one: the sum over k is order-free)
__global__ void k_gram(const float* z, float* Z) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  union { bf16x8 v; unsigned u[4]; } H, M, L;
  for (int s = 0; s < 8; s += 2) {
    const float x0 = z[(8 * g + s) * 16 + i], x1 = z[(8 * g + s + 1) * 16 + i];
    const unsigned ph = cvt_pk(x0, x1);
    const float r0 = x0 - bf_lo(ph), r1 = x1 - bf_hi(ph);
    const unsigned pm = cvt_pk(r0, r1);
    const float s0 = r0 - bf_lo(pm), s1 = r1 - bf_hi(pm);
    H.u[s / 2] = ph; M.u[s / 2] = pm; L.u[s / 2] = cvt_pk(s0, s1);
  }
  f32x4 c = { 0, 0, 0, 0 };
  // smallest terms first
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(M.v, M.v, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H.v, L.v, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L.v, H.v, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H.v, M.v, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(M.v, H.v, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H.v, H.v, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) Z[(4 * g + r) * 16 + i] = c[r];
}

static uint16_t to_bf16_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

int main() {
  std::mt19937 rng(1234);
  // ---- 1. layout
  std::vector<uint16_t> A(16 * 32), B(32 * 16);
  std::vector<double> Ad(16 * 32), Bd(32 * 16);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) { const float v = (float)((i * 7 + k * 3) % 11 - 5); A[i * 32 + k] = to_bf16_bits(v); Ad[i * 32 + k] = v; }
  for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) { const float v = (float)((k * 5 + j * 13 + 1) % 9 - 4); B[k * 16 + j] = to_bf16_bits(v); Bd[k * 16 + j] = v; }
  uint16_t *dA, *dB; float* dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  float D[256]; hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
  int bad = 0, bad_t = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double s = 0; for (int k = 0; k < 32; ++k) s += Ad[i * 32 + k] * Bd[k * 16 + j];
    if (D[i * 16 + j] != (float)s) ++bad;
    if (D[j * 16 + i] != (float)s) ++bad_t;
  }
  std::printf("1. layout v_mfma_f32_16x16x32_bf16 (A row=l&15 k=8(l>>4)+s | B k=8(l>>4)+s col=l&15 | D row=4(l>>4)+r col=l&15): %s (%d wrong; transposed reading: %d wrong)\n",
              bad == 0 ? "AS ASSUMED" : "DIFFERENT", bad, bad_t);

  // ---- 2. split exactness
  const int n = 1 << 20;
  std::vector<float> x(n);
  std::uniform_real_distribution<float> mant(1.0f, 2.0f);
  std::uniform_int_distribution<int> ex(-40, 40), sg(0, 1);
  for (int t = 0; t < n; ++t) x[t] = std::ldexp(mant(rng), ex(rng)) * (sg(rng) ? 1.f : -1.f);
  x[0] = 0.f; x[1] = 1.f; x[2] = -1.f; x[3] = 1.0f + 1.1920929e-7f; x[4] = 0.99999994f; x[5] = 3.0f - 2.3841858e-7f; x[6] = 1.00390625f; x[7] = 1.99609375f;
  float *dx, *dh, *dm, *dl;
  hipMalloc(&dx, n * 4); hipMalloc(&dh, n * 4); hipMalloc(&dm, n * 4); hipMalloc(&dl, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_split, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, dh, dm, dl);
  std::vector<float> h(n), m(n), lo(n);
  hipMemcpy(h.data(), dh, n * 4, hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, n * 4, hipMemcpyDeviceToHost); hipMemcpy(lo.data(), dl, n * 4, hipMemcpyDeviceToHost);
  long inexact = 0, not_bf16 = 0; double worst = 0;
  for (int t = 0; t < n; ++t) {
    const double sum = (double)h[t] + (double)m[t] + (double)lo[t];
    if (sum != (double)x[t]) { ++inexact; const double e = std::fabs(sum - x[t]) / std::fabs(x[t]); if (e > worst) worst = e; }
    for (float p : { h[t], m[t], lo[t] }) { uint32_t u; std::memcpy(&u, &p, 4); if (u & 0xffffu) ++not_bf16; }
  }
  std::printf("2. three-way RNE split through v_cvt_pk_bf16_f32, %d floats in 2^-40 .. 2^41: %ld not reconstructed exactly (worst relative error %.3g), %ld pieces not bf16\n",
              n, inexact, worst, not_bf16);

  // ---- 2b. the same split with v_dot2c_f32_bf16 remainders: identical pieces?  (also on a range that reaches into the denormals)
  {
    hipLaunchKernelGGL(k_split_dot2, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, dh, dm, dl);
    std::vector<float> h2(n), m2(n), l2(n);
    hipMemcpy(h2.data(), dh, n * 4, hipMemcpyDeviceToHost); hipMemcpy(m2.data(), dm, n * 4, hipMemcpyDeviceToHost); hipMemcpy(l2.data(), dl, n * 4, hipMemcpyDeviceToHost);
    long differ = 0, inexact2 = 0;
    for (int t = 0; t < n; ++t) {
      if (std::memcmp(&h2[t], &h[t], 4) || std::memcmp(&m2[t], &m[t], 4) || std::memcmp(&l2[t], &lo[t], 4)) ++differ;
      if ((double)h2[t] + (double)m2[t] + (double)l2[t] != (double)x[t]) ++inexact2;
    }
    std::printf("2b. remainders through v_dot2c_f32_bf16: %ld of %d inputs give different pieces than the shift/mask/subtract form, %ld not reconstructed exactly\n", differ, n, inexact2);
    // pairs with one huge and one tiny member, and values near the bottom of the normal range
    std::vector<float> y(n);
    std::uniform_int_distribution<int> exw(-126, 100);
    for (int t = 0; t < n; ++t) y[t] = std::ldexp(mant(rng), exw(rng)) * (sg(rng) ? 1.f : -1.f);
    hipMemcpy(dx, y.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_split, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, dh, dm, dl);
    hipMemcpy(h.data(), dh, n * 4, hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, n * 4, hipMemcpyDeviceToHost); hipMemcpy(lo.data(), dl, n * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_split_dot2, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, dh, dm, dl);
    hipMemcpy(h2.data(), dh, n * 4, hipMemcpyDeviceToHost); hipMemcpy(m2.data(), dm, n * 4, hipMemcpyDeviceToHost); hipMemcpy(l2.data(), dl, n * 4, hipMemcpyDeviceToHost);
    long differ_w = 0, inexact_w = 0, inexact_w_small = 0; double worst_w = 0; float smallest_bad = 0.f;
    for (int t = 0; t < n; ++t) {
      if (std::memcmp(&h2[t], &h[t], 4) || std::memcmp(&m2[t], &m[t], 4) || std::memcmp(&l2[t], &lo[t], 4)) ++differ_w;
      const double sum = (double)h2[t] + (double)m2[t] + (double)l2[t];
      if (sum != (double)y[t]) {
        ++inexact_w;
        if (std::fabs(y[t]) < std::ldexp(1.0f, -100)) ++inexact_w_small;
        const double e = std::fabs(sum - y[t]) / std::fabs(y[t]); if (e > worst_w) { worst_w = e; smallest_bad = y[t]; }
      }
    }
    std::printf("    wide range 2^-126 .. 2^101: %ld different pieces, %ld not exact (%ld of them below 2^-100; worst relative error %.3g at %.3g)\n", differ_w, inexact_w,
                inexact_w_small, worst_w, (double)smallest_bad);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  }

  // ---- 3. Gram matrix of 32 x 16 fp32 values
  std::vector<float> z(32 * 16);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& v : z) v = nd(rng) * std::ldexp(1.0f, ex(rng) / 8);
  float *dz, *dZ; hipMalloc(&dz, z.size() * 4); hipMalloc(&dZ, 256 * 4);
  hipMemcpy(dz, z.data(), z.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_gram, dim3(1), dim3(64), 0, 0, dz, dZ);
  float Z[256]; hipMemcpy(Z, dZ, sizeof(Z), hipMemcpyDeviceToHost);
  double e_split = 0, e_chain = 0, scale = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double s = 0; float c = 0.f;
    for (int k = 0; k < 32; ++k) { s += (double)z[k * 16 + i] * z[k * 16 + j]; c = std::fmaf(z[k * 16 + i], z[k * 16 + j], c); }
    scale = std::fmax(scale, std::fabs(s));
    e_split = std::fmax(e_split, std::fabs(Z[i * 16 + j] - s));
    e_chain = std::fmax(e_chain, std::fabs((double)c - s));
  }
  std::printf("3. 16x16 Gram matrix over 32 fp32 vectors: max |error| / max |entry|  bf16x3 (six MFMAs) %.3g   sequential fp32 fmaf chain %.3g\n", e_split / scale,
              e_chain / scale);
  return 0;
}
