// dfx_sfm_step.hip -- SfmAligner::RunStep for gfx950 (replaces kernel_step_calculate + kernel_finalize_reduction,
// reference sources/cuda/cu_sfmaligner.cpp:40-70,149-185 and the DenseSfm item, common/algorithm/dense_sfm.h:133-201).
//
// Design:
//   The reference keeps 1035 fp32 accumulators per THREAD (4 KB, spilled) and launches 11x32 threads.
//   Here a 64-lane wave owns 64 consecutive pixels per step ("chunk") and the JtJ / Jtr / r^2 / inlier sums are
//   one rank-1-update GEMM  Z += z z^T  over pixels, executed on the matrix cores with exact-fp32 MFMA
//   (v_mfma_f32_16x16x4_f32: k = 4 pixels per instruction, bitwise an fmaf chain).
//   z (per pixel, "z-space") is laid out in 16-row blocks:
//     block P : [ w*J_pose0 (6), w*J_pose1 (6), w*r, s, inlier(1.0), 0 ]        s = w * dE/dprx
//     block Cb: [ s * jac[NCB*i + b] ]_{i=0..15}, b = 0..NCB-1                  (NCB = CS/16)
//   Only the upper block-triangle is accumulated: (P,P), (P,Cb), (Cb,Cb') b<=b'  -> 6 MFMAs per 4 pixels at CS=32.
//   (P,P)[12][12] = sum (w r)^2 = residual, (P,P)[14][14] = inliers, (P,*)[12][*] = Jtr: everything comes out of
//   the same accumulators.  Block P row 13 (s) only exists so that lanes can broadcast s; its products are ignored.
//
//   Phase A (lane = pixel): coalesced img0/dpt0 loads, warp, bilinear gathers of img1/grad1, Jacobian row,
//     Huber weight; the 16-float P row goes to LDS component-major (stride 66 -> conflict-free both ways).
//   Phase B (lane = (i = lane&15, k = lane>>4)): the code Jacobian is loaded from HBM directly in MFMA operand
//     layout -- lane (i,k) reads NCB consecutive floats of pixel 4g+k, i.e. every wave-load is one fully
//     contiguous 256*NCB-byte run of the [H][W*CS] stream (86 % of all bytes); loads are issued before phase A
//     so their latency hides under it.  Zero weights use v_mul_legacy (0 * NaN = 0) so that garbage in
//     the Jacobian of a masked pixel cannot poison the sums.
//   Epilogue: waves fold their accumulators through LDS in fixed order; each workgroup writes one z-space partial;
//     k_sfm_finalize sums the partials of a pair in double, in fixed order (bit-reproducible for a given launch
//     shape), and scatters into the reference's JTJJrReductionItem layout (reduction_items.h:77-143).
#include "dfx_device.hpp"
#include "dfx_kernels.hpp"

namespace dfx {

constexpr int kWaves = 4;                 // waves per workgroup
constexpr int kThreads = kWaves * 64;
constexpr int kUStride = 66;              // floats; 16 rows x 66: bank = (2*i + p) % 32 -> conflict-free
constexpr int kUFloats = 16 * kUStride;   // per wave

template <int NCB> struct JV;
template <> struct JV<1> { typedef float T; };
template <> struct JV<2> { typedef f32x2 T; };
template <> struct JV<4> { typedef f32x4 T; };

template <int NCB> __device__ __forceinline__ float jv_get(const typename JV<NCB>::T& v, int b);
template <> __device__ __forceinline__ float jv_get<1>(const float& v, int) { return v; }
template <> __device__ __forceinline__ float jv_get<2>(const f32x2& v, int b) { return b == 0 ? v.x : v.y; }
template <> __device__ __forceinline__ float jv_get<4>(const f32x4& v, int b) { return b == 0 ? v.x : (b == 1 ? v.y : (b == 2 ? v.z : v.w)); }

// MODE 0: SfmAligner::RunStep.  MODE 1: DepthAligner::RunStep (cu_depthaligner.cpp:32-72) -- same rank-1 GEMM with
// z = [0 (12), diff, s, 1, 0 | s * jac], s = -2 |diff| * dDepth/dPrx; `img0` carries the target depth and `dpt0`
// the current depth (already decoded by k_update_depth); every pixel is an inlier.
template <int NCB, int MODE>
__global__ __launch_bounds__(kThreads) void k_sfm_step(const SfmPairDev* __restrict__ pairs, const SfmParamsDev prm,
                                                       const int W, const int H, float* __restrict__ partials) {
  constexpr int CS = 16 * NCB;
  constexpr int NBLK = 1 + NCB;
  constexpr int NACC = NBLK * (NBLK + 1) / 2;
  constexpr int ZDIM = NACC * 256;
  constexpr int LDS_FLOATS = (kWaves * kUFloats > ZDIM) ? kWaves * kUFloats : ZDIM;
  typedef typename JV<NCB>::T jv_t;

  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const SfmPairDev& P = pairs[blockIdx.y];

  Geo g;
#pragma unroll
  for (int q = 0; q < 9; ++q) g.R[q] = P.R[q];
  g.t[0] = P.t[0]; g.t[1] = P.t[1]; g.t[2] = P.t[2];
  g.fx = P.fx; g.fy = P.fy; g.u0 = P.u0; g.v0 = P.v0; g.w = P.w; g.h = P.h;
  const ImgRef I0{ (const char*)P.img0, P.pitch_img0 }, I1{ (const char*)P.img1, P.pitch_img1 };
  const ImgRef D0{ (const char*)P.dpt0, P.pitch_dpt0 }, G1{ (const char*)P.grad1, P.pitch_grad1 };
  const char* jac_base = (const char*)P.jac;
  const uint32_t jac_pitch = P.pitch_jac;
  const float inv_a = 1.0f / prm.avg_dpt;

  float* U = lds + wave * kUFloats;
  U[15 * kUStride + lane] = 0.f;   // row 15 of block P is padding

  f32x4 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) acc[a] = f32x4{ 0.f, 0.f, 0.f, 0.f };

  const int npx = W * H;
  const int nchunks = (npx + 63) >> 6;
  const int li = lane & 15, lk = lane >> 4;

  for (int chunk = blockIdx.x * kWaves + wave; chunk < nchunks; chunk += gridDim.x * kWaves) {
    const int base = chunk << 6;
    const int y0 = base / W;
    const int x0 = base - y0 * W;

    // ---- issue the code-Jacobian loads in MFMA operand layout (independent of phase A)
    jv_t jv[16];
#pragma unroll
    for (int gq = 0; gq < 16; ++gq) {
      const int off = 4 * gq + lk;
      int x = x0 + off, y = y0;
      while (x >= W) { x -= W; ++y; }
      const bool inb = (base + off) < npx;
      const jv_t* src = reinterpret_cast<const jv_t*>(jac_base + (size_t)y * jac_pitch) + (x * 16 + li);
      jv_t v;
      if (inb) v = *src; else v = jv_t(0.f);
      jv[gq] = v;
    }

    // ---- phase A: lane = pixel
    {
      int x = x0 + lane, y = y0;
      while (x >= W) { x -= W; ++y; }
      const bool inb = (base + lane) < npx;
      float u16[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) u16[q] = 0.f;
      if (MODE == 1) {
        if (inb) {
          const float d = D0.row(y)[x];
          const float diff = I0.row(y)[x] - d;
          const float apd = prm.avg_dpt + d;
          u16[12] = diff;
          u16[13] = 2.0f * fabsf(diff) * (apd * apd) * inv_a;   // -2 |diff| * (-a / prx^2), prx = a / (a + d)
          u16[14] = 1.0f;
        }
      } else if (inb) {
        const float d = D0.row(y)[x];
        const float i0 = I0.row(y)[x];
        const Corr c = find_correspondence(g, x, y, d, prm.border, prm.min_dpt);
        if (c.valid) {
          const Taps tp = make_taps(c.u, c.v);
          float gx, gy;
          sample_grad(G1, tp, gx, gy);
          const float samp = sample_img(I1, tp);
          float gC[6], D00, D02, D11, D12;
          pose_row(g, c, d, gx, gy, gC, D00, D02, D11, D12);
          // J0 = gC * blkdiag(M, M);  J1 = gC * [[-M, -HM], [0, -M]]
          float J[12];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            J[j] = gC[0] * P.M[j] + gC[1] * P.M[3 + j] + gC[2] * P.M[6 + j];
            J[3 + j] = gC[3] * P.M[j] + gC[4] * P.M[3 + j] + gC[5] * P.M[6 + j];
            J[6 + j] = -J[j];
            J[9 + j] = -(gC[0] * P.HM[j] + gC[1] * P.HM[3 + j] + gC[2] * P.HM[6 + j]) - J[3 + j];
          }
          // d pix1 / d prx = D * (R ray) * (-a / prx^2),  prx = a / (a + d)   (warping.h:44-50,259-291)
          const float apd = prm.avg_dpt + d;
          const float dprx = -(apd * apd) * inv_a;
          const float pj0 = (D00 * c.rrx + D02 * c.rrz) * dprx;
          const float pj1 = (D11 * c.rry + D12 * c.rrz) * dprx;
          const float e = -(gx * pj0 + gy * pj1);
          const float r = i0 - samp;
          const float wgt = huber_weight(r, prm.huber_delta);   // * DenseSfm_UncertaintyWeight == 1 (dense_sfm.h:66)
#pragma unroll
          for (int j = 0; j < 12; ++j) u16[j] = wgt * J[j];
          u16[12] = wgt * r;
          u16[13] = wgt * e;
          u16[14] = 1.0f;
          if (P.valid0) reinterpret_cast<float*>((char*)P.valid0 + (size_t)y * P.pitch_valid0)[x] = 1.0f;   // dense_sfm.h:161
        }
      }
#pragma unroll
      for (int q = 0; q < 15; ++q) U[q * kUStride + lane] = u16[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- phase B: rank-4 updates on the matrix cores
#pragma unroll
    for (int gq = 0; gq < 16; ++gq) {
      const int pp = 4 * gq + lk;
      const float uP = U[li * kUStride + pp];
      const float s = U[13 * kUStride + pp];
      float sc[NCB];
#pragma unroll
      for (int b = 0; b < NCB; ++b) sc[b] = mul_zero_wins(s, jv_get<NCB>(jv[gq], b));
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uP, uP, acc[0], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < NCB; ++b) acc[1 + b] = __builtin_amdgcn_mfma_f32_16x16x4f32(uP, sc[b], acc[1 + b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < NCB; ++b)
#pragma unroll
        for (int b2 = b; b2 < NCB; ++b2) {
          const int a = 1 + NCB + (b * NCB - b * (b - 1) / 2) + (b2 - b);   // row-major upper block-triangle
          acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[b], sc[b2], acc[a], 0, 0, 0);
        }
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- epilogue: fold the waves' accumulators in fixed order (wave 0, 1, 2, 3), one partial per workgroup.
  // C/D layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
  __syncthreads();
  for (int wv = 0; wv < kWaves; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int idx = a * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15);
          if (wv == 0) lds[idx] = acc[a][r]; else lds[idx] += acc[a][r];
        }
    }
    __syncthreads();
  }
  float* out = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * ZDIM;
  for (int e = threadIdx.x; e < ZDIM; e += kThreads) out[e] = lds[e];
}

// ---- finalize: sum the workgroup partials of each pair (double, fixed order) and scatter into the item layout
template <int NCB, int NPOSE>
__global__ __launch_bounds__(1024) void k_sfm_finalize(const float* __restrict__ partials, const int bpp,
                                                       char* __restrict__ items, const size_t item_stride) {
  constexpr int CS = 16 * NCB;
  constexpr int NP = NPOSE + CS;
  constexpr int NBLK = 1 + NCB;
  constexpr int NACC = NBLK * (NBLK + 1) / 2;
  constexpr int ZDIM = NACC * 256;
  constexpr int NT = NP * (NP + 1) / 2;
  __shared__ double red[4][256];

  const int a = blockIdx.x, pair = blockIdx.y;
  const int col = threadIdx.x & 255, rg = threadIdx.x >> 8;
  const float* src = partials + (size_t)pair * bpp * ZDIM + a * 256 + col;
  double s = 0.0;
  for (int b = rg; b < bpp; b += 4) s += (double)src[(size_t)b * ZDIM];
  red[rg][col] = s;
  __syncthreads();
  if (rg != 0) return;
  s = ((red[0][col] + red[1][col]) + red[2][col]) + red[3][col];

  // block pair (bi <= bj) of accumulator a
  int bi = 0, bj = 0;
  {
    int q = a;
    for (bi = 0; bi < NBLK; ++bi) { const int n = NBLK - bi; if (q < n) { bj = bi + q; break; } q -= n; }
  }
  const int i = col >> 4, j = col & 15;
  float* item = reinterpret_cast<float*>(items + (size_t)pair * item_stride);
  // z index -> parameter index (or -1), 'r' = row 12 of block P
  const int n = (bi == 0) ? (i < NPOSE ? i : -1) : NPOSE + NCB * i + (bi - 1);
  const int m = (bj == 0) ? (j < NPOSE ? j : -1) : NPOSE + NCB * j + (bj - 1);
  const bool i_is_r = (bi == 0 && i == 12), j_is_r = (bj == 0 && j == 12);
  if (n >= 0 && m >= 0) {
    if (bi == bj && i > j) return;   // symmetric duplicate inside a diagonal block
    const int lo = n < m ? n : m, hi = n < m ? m : n;
    item[lo * NP - lo * (lo - 1) / 2 + (hi - lo)] = (float)s;
  } else if (i_is_r && m >= 0) {
    item[NT + m] = (float)s;                       // Jtr
  } else if (i_is_r && j_is_r) {
    item[NT + NP] = (float)s;                      // residual = sum (w r)^2
  } else if (bi == 0 && bj == 0 && i == 14 && j == 14) {
    const size_t off = (((size_t)(NT + NP + 1)) * 4 + 7) & ~(size_t)7;
    *reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(item) + off) = (unsigned long long)(s + 0.5);
  }
}

size_t sfm_step_partials_bytes(int cs, int npairs, int blocks_per_pair) {
  return (size_t)npairs * blocks_per_pair * sfm_zdim(cs / 16) * sizeof(float);
}

template <int NCB, int MODE>
static hipError_t launch_t(const SfmPairDev* pairs_dev, int npairs, int W, int H, const SfmParamsDev& prm, int bpp,
                           float* partials_dev, void* items_dev, size_t item_stride, hipStream_t stream,
                           hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr) {
  constexpr int NACC = (1 + NCB) * (2 + NCB) / 2;
  hipError_t e;
  if (ev_begin && (e = hipEventRecord(ev_begin, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL((k_sfm_step<NCB, MODE>), dim3(bpp, npairs), dim3(kThreads), 0, stream, pairs_dev, prm, W, H, partials_dev);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (ev_end && (e = hipEventRecord(ev_end, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL((k_sfm_finalize<NCB, MODE == 0 ? 12 : 0>), dim3(NACC, npairs), dim3(1024), 0, stream,
                     (const float*)partials_dev, bpp, (char*)items_dev, item_stride);
  return hipGetLastError();
}

hipError_t launch_sfm_step(int cs, const SfmPairDev* pairs_dev, int npairs, int W, int H, const SfmParamsDev& prm,
                           int blocks_per_pair, float* partials_dev, void* items_dev, size_t item_stride,
                           hipStream_t stream, hipEvent_t eb, hipEvent_t ee) {
  switch (cs) {
    case 16: return launch_t<1, 0>(pairs_dev, npairs, W, H, prm, blocks_per_pair, partials_dev, items_dev, item_stride, stream, eb, ee);
    case 32: return launch_t<2, 0>(pairs_dev, npairs, W, H, prm, blocks_per_pair, partials_dev, items_dev, item_stride, stream, eb, ee);
    case 64: return launch_t<4, 0>(pairs_dev, npairs, W, H, prm, blocks_per_pair, partials_dev, items_dev, item_stride, stream, eb, ee);
    default: return hipErrorInvalidValue;
  }
}

// DepthAligner::RunStep: `pair_dev` describes ONE pseudo-pair with img0 = target depth, dpt0 = current depth, jac.
hipError_t launch_depth_aligner_step(int cs, const SfmPairDev* pair_dev, int W, int H, float avg_dpt, int blocks,
                                     float* partials_dev, void* item_dev, hipStream_t stream) {
  SfmParamsDev prm{ 0.f, avg_dpt, 0.f, 0.f };
  switch (cs) {
    case 16: return launch_t<1, 1>(pair_dev, 1, W, H, prm, blocks, partials_dev, item_dev, 0, stream);
    case 32: return launch_t<2, 1>(pair_dev, 1, W, H, prm, blocks, partials_dev, item_dev, 0, stream);
    case 64: return launch_t<4, 1>(pair_dev, 1, W, H, prm, blocks, partials_dev, item_dev, 0, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace dfx
