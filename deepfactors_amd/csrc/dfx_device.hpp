// dfx_device.hpp -- device-side math shared by the gfx950 kernels.
//
// Restates (does not copy) the per-pixel math of the reference's L0 headers for CDNA4:
//   warping.h:204-291 (correspondence + Jacobians), pinhole_camera_impl.h:41-108,
//   m_estimators.h:50-56 (Huber), dense_sfm.h:133-201, lucas_kanade_se3.h:41-77.
// Everything here is wave64 code; nothing assumes a 32-wide warp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfx {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// 4-byte-aligned views for bilinear taps: (ix, ix+1) pairs start at arbitrary dword addresses.
// gfx950 global loads only need dword alignment for multi-dword accesses.
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4_u8 __attribute__((ext_vector_type(4), aligned(8)));

// Wave-uniform geometry of one (keyframe -> frame) pair; lives in kernarg/SGPRs.
struct Geo {
  float R[9];   // rotation of pose_10 = pose1^-1 * pose0, row-major
  float t[3];
  float fx, fy, u0, v0, w, h;
};

// Pointers arrive inside descriptor structs, so the compiler cannot infer their address space and would emit
// flat_load (which also ties up lgkmcnt).  Everything image-like is HBM: say so explicitly.
#define DFX_GLOBAL __attribute__((address_space(1)))
typedef const DFX_GLOBAL float* gfptr;
template <typename T>
__device__ __forceinline__ T gload(const void* p) { return *(const DFX_GLOBAL T*)(p); }
template <typename T>
__device__ __forceinline__ void gstore(void* p, const T& v) { *(DFX_GLOBAL T*)(p) = v; }

// Raw buffer resource (V#) over a wave-uniform base: loads past `bytes` return 0 and move no data, which replaces
// every tail / "no next chunk" branch on the streaming loads (cdna_hip_programming.md T8).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOobOffset = 0xF0000000u;   // far beyond any image; API enforces images < 0x70000000 bytes
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// AUX = cache policy bits of the instruction (0 default, 2 = nt: streamed once, do not keep in the L2 ahead of reusable lines)
template <int AUX = 0>
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned off, float*) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ f32x2 bload(__amdgpu_buffer_rsrc_t r, unsigned off, f32x2*) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4*) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, AUX));
}

struct ImgRef {
  const char* ptr;
  uint32_t pitch;   // bytes
  __device__ __forceinline__ const char* rowb(int y) const { return ptr + (size_t)y * pitch; }
  __device__ __forceinline__ float at(int x, int y) const { return gload<float>(rowb(y) + (size_t)x * 4); }
};

#ifndef DFX_DIV_SHARED
#define DFX_DIV_SHARED 1     // find_correspondence_ray<true>: the three divisions by q.z share one refined reciprocal (0: three compiler divisions everywhere)
#endif
// FindCorrespondence (warping.h:204-241): p = d * K^-1 (x,y,1); q = R p + t; pix1 = K q / q.z
struct Corr {
  float rrx, rry, rrz;   // R * ray            (ray = ReprojectDepthJacobian, pinhole_camera_impl.h:77-86)
  float vx, vy, vz;      // R * p              (p = ray * d; TransformJacobianPose uses this without +t)
  float qx, qy, iz;      // q.x, q.y, 1/q.z    (q = R p + t)
  float u, v;
  bool valid;
};

// The validity predicate decides which pixels enter the sums, so this function follows the reference's operation
// order exactly (Reproject -> se3 * pt -> Project -> PixelValid) with IEEE division and NO fma contraction: the
// inlier set is then bit-identical to a host evaluation of the same formulas (the oracle), even for degenerate
// poses (identity) where projected coordinates land exactly on the border.
// (rx, ry) = K^-1 (x, y, 1) is handed in: the step kernel reads it from a per-camera table (dfx_api.cpp ray_table, the same
// IEEE expression evaluated once per column / row on the host) instead of paying two divisions per pixel.
// SD: the shared-reciprocal form of the three divisions (below).  Same bits; it holds the reciprocal across the quotients, which costs the fp32
// chain of the step kernel its fourth wave per SIMD (126 -> 135 registers), so the caller chooses.
template <bool SD = false>
__device__ __forceinline__ Corr find_correspondence_ray(const Geo& g, float rx, float ry, float d, float border, float min_dpt) {
#pragma clang fp contract(off)
  Corr c;
  const float px = rx * d, py = ry * d, pz = d;
  c.vx = g.R[0] * px + g.R[1] * py + g.R[2] * pz;
  c.vy = g.R[3] * px + g.R[4] * py + g.R[5] * pz;
  c.vz = g.R[6] * px + g.R[7] * py + g.R[8] * pz;
  c.rrx = g.R[0] * rx + g.R[1] * ry + g.R[2];
  c.rry = g.R[3] * rx + g.R[4] * ry + g.R[5];
  c.rrz = g.R[6] * rx + g.R[7] * ry + g.R[8];
  c.qx = c.vx + g.t[0];
  c.qy = c.vy + g.t[1];
  const float qz = c.vz + g.t[2];
  if constexpr (SD && DFX_DIV_SHARED) {
  // The three IEEE quotients by q.z, with the refined reciprocal computed once: the arithmetic of the compiler's own division sequence
  // (v_rcp_f32, one Newton step, quotient, two residual corrections -- correctly rounded) minus its range scaling (v_div_scale /
  // v_div_fmas / v_div_fixup), which only acts on operands beyond 2^+-96: 18 instead of 33 vector-ALU instructions per pixel, the same
  // bits for every depth a camera produces.  (Non-finite or zero q.z: NaN here, a pixel without correspondence either way.)
    const float r0 = __builtin_amdgcn_rcpf(qz);
    const float e0 = __builtin_fmaf(-qz, r0, 1.0f);
    const float r = __builtin_fmaf(e0, r0, r0);
    auto quot = [&](float a) {
      const float q0 = a * r;
      const float q1 = __builtin_fmaf(__builtin_fmaf(-qz, q0, a), r, q0);
      return __builtin_fmaf(__builtin_fmaf(-qz, q1, a), r, q1);
    };
    c.iz = quot(1.0f);
    c.u = quot(g.fx * c.qx) + g.u0;
    c.v = quot(g.fy * c.qy) + g.v0;
  } else {
  c.iz = 1.0f / qz;
  c.u = g.fx * c.qx / qz + g.u0;
  c.v = g.fy * c.qy / qz + g.v0;
  }
  // PixelValid (pinhole_camera_impl.h:105-108) in float, exactly `x >= b && x < w - b`; NaN -> invalid
  c.valid = (qz > min_dpt) && (c.u >= border) && (c.u < g.w - border) && (c.v >= border) && (c.v < g.h - border);
  return c;
}
template <bool SD = false>
__device__ __forceinline__ Corr find_correspondence(const Geo& g, int x, int y, float d, float border, float min_dpt) {
#pragma clang fp contract(off)
  const float rx = ((float)x - g.u0) / g.fx;
  const float ry = ((float)y - g.v0) / g.fy;
  return find_correspondence_ray<SD>(g, rx, ry, d, border, min_dpt);
}

// VisionCore getBilinear convention: floor, lerp in x then y, lerp(a,b,t)=a+t(b-a)
struct Taps {
  int ix, iy;
  float ax, ay;
};
__device__ __forceinline__ Taps make_taps(float u, float v) {
  Taps t;
  const float fu = floorf(u), fv = floorf(v);
  t.ix = (int)fu; t.iy = (int)fv;
  t.ax = u - fu; t.ay = v - fv;
  return t;
}
__device__ __forceinline__ float lerp1(float a, float b, float t) {
#pragma clang fp contract(off)
  return a + t * (b - a);   // separate mul + add: sampled values are bit-identical to a host evaluation
}

__device__ __forceinline__ float sample_img(const ImgRef& I, const Taps& t) {
  const char* r0 = I.rowb(t.iy) + (size_t)t.ix * 4;
  const f32x2_u a = gload<f32x2_u>(r0);
  const f32x2_u b = gload<f32x2_u>(r0 + I.pitch);
  return lerp1(lerp1(a.x, a.y, t.ax), lerp1(b.x, b.y, t.ax), t.ay);
}
__device__ __forceinline__ void sample_grad(const ImgRef& G, const Taps& t, float& gx, float& gy) {
  const char* r0 = G.rowb(t.iy) + (size_t)t.ix * 8;
  const f32x4_u8 a = gload<f32x4_u8>(r0);
  const f32x4_u8 b = gload<f32x4_u8>(r0 + G.pitch);
  gx = lerp1(lerp1(a.x, a.z, t.ax), lerp1(b.x, b.z, t.ax), t.ay);
  gy = lerp1(lerp1(a.y, a.w, t.ax), lerp1(b.y, b.w, t.ax), t.ay);
}

// HuberWeight (m_estimators.h:50-56): sqrt-weight for both J and r.  The weight is a smooth factor, not a decision:
// v_sqrt_f32 * v_rcp_f32 (1 ulp each) instead of the IEEE sqrt + division sequences (~25 VALU ops per pixel).
__device__ __forceinline__ float huber_weight(float r, float delta) {
  const float aa = fabsf(r);
  const float wo = __builtin_amdgcn_sqrtf(delta * (2.0f * aa - delta)) * __builtin_amdgcn_rcpf(aa);
  return aa <= delta ? 1.0f : wo;
}

// -grad * ProjectPointJacobian(q) * [I | -hat(R p)]   (warping.h:156-164,247-257; Rp = rr * d, no translation)
__device__ __forceinline__ void pose_row(const Geo& g, const Corr& c, float d, float gx, float gy, float* gC /*6*/,
                                         float& D00, float& D02, float& D11, float& D12) {
  D00 = g.fx * c.iz;
  D11 = g.fy * c.iz;
  D02 = -(g.fx * c.qx) * c.iz * c.iz;
  D12 = -(g.fy * c.qy) * c.iz * c.iz;
  const float vx = c.vx, vy = c.vy, vz = c.vz;   // R p
  const float C03 = D02 * vy, C04 = D00 * vz - D02 * vx, C05 = -D00 * vy;
  const float C13 = -D11 * vz + D12 * vy, C14 = -D12 * vx, C15 = D11 * vx;
  gC[0] = -(gx * D00);
  gC[1] = -(gy * D11);
  gC[2] = -(gx * D02 + gy * D12);
  gC[3] = -(gx * C03 + gy * C13);
  gC[4] = -(gx * C04 + gy * C14);
  gC[5] = -(gx * C05 + gy * C15);
}

// v_mul_legacy_f32: DX9 multiply, 0 * anything (NaN, Inf) = 0.  Used to apply zero weights safely.
// Bound to the LLVM intrinsic (not inline asm): hipcc pads no hazards around an asm statement, and the product
// feeds an MFMA operand (VALU write -> MFMA read needs wait states the compiler must see).
extern "C" __device__ float dfx_llvm_fmul_legacy(float, float) __asm("llvm.amdgcn.fmul.legacy");
__device__ __forceinline__ float mul_zero_wins(float a, float b) { return dfx_llvm_fmul_legacy(a, b); }

// sum_{b = first, first+STEP, ... < n} src[b * stride] in double, in that fixed order, with DEPTH loads in flight (a plain loop
// pays one L2 round trip per element: 9.7 us for 1024 partial rows).
template <int STEP, int DEPTH = 8>
__device__ __forceinline__ double strided_sum_f64(const float* __restrict__ src, int first, int n, size_t stride) {
  double s = 0.0;
  int b = first;
  for (; b + (DEPTH - 1) * STEP < n; b += DEPTH * STEP) {
    float v[DEPTH];
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) v[q] = src[(size_t)(b + q * STEP) * stride];
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) s += (double)v[q];
  }
  if (b < n) {   // leftover rows: again all in flight (unconditional loads of existing rows, +0.0 for the rows past n)
    float v[DEPTH];
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) {
      const int r = b + q * STEP;
      const float t = src[(size_t)(r < n ? r : b) * stride];
      v[q] = r < n ? t : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) s += (double)v[q];
  }
  return s;
}

// Same sum, same order, for n <= STEP * MAXQ: every load of the thread is in flight at once (ONE L2 round trip instead of one per
// batch of 8 plus one per leftover row).  Rows past n contribute +0.0, which leaves the running double sum unchanged bit for bit.
template <int STEP, int MAXQ>
__device__ __forceinline__ double strided_sum_f64_wide(const float* __restrict__ src, int first, int n, size_t stride) {
  float v[MAXQ];
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    const int b = first + q * STEP;
    const float t = src[(size_t)(b < n ? b : 0) * stride];   // unconditional load of an existing row (a conditional one is a branch + wait each)
    v[q] = b < n ? t : 0.0f;
  }
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) s += (double)v[q];
  return s;
}

// Two strided sums whose loads are in flight TOGETHER (TWO = false: only a): the diagonal tiles of a pair's normal equations fold two blocks of every partial
// row, and a single pair's finalize kernel is nothing but these dependent round trips (0.55 us per batch of 16 rows: 4.5 us for the two sums one after the
// other, timestamps in profiles/r05_poll_result.txt).  Same order of additions per source as strided_sum_f64 (ascending rows, one chain): the same bits.
template <int STEP, int DEPTH, bool TWO>
__device__ __forceinline__ void strided_sum2_f64(const float* __restrict__ a, const float* __restrict__ b, int first, int n, size_t stride, double& sa_out, double& sb_out) {
  double sa = 0.0, sb = 0.0;
  for (int r0 = first; r0 < n; r0 += DEPTH * STEP) {
    float va[DEPTH], vb[DEPTH];
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) {
      const int r = r0 + q * STEP;
      const size_t off = (size_t)(r < n ? r : r0) * stride;   // unconditional loads of existing rows (a conditional one is a branch + a wait each), +0.0 past n
      const float ta = a[off];
      va[q] = r < n ? ta : 0.0f;
      if constexpr (TWO) { const float tb = b[off]; vb[q] = r < n ? tb : 0.0f; }
    }
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) { sa += (double)va[q]; if constexpr (TWO) sb += (double)vb[q]; }
  }
  sa_out = sa; sb_out = sb;
}

// 64-lane sum via shuffles (wave = 64 on gfx950)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

}  // namespace dfx
