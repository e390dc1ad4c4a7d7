// dfx_misc_kernels.hip -- the HBM/latency-bound siblings of the SfM step for gfx950:
//   SE3Aligner::RunStep / Warp   (reference sources/cuda/cu_se3aligner.cpp:37-176, lucas_kanade_se3.h:41-77)
//   SfmAligner::EvaluateError    (cu_sfmaligner.cpp:72-147, dense_sfm.h:79-119)
//   UpdateDepth, SobelGradients, GaussianBlurDown, SquaredError (cu_image_proc.cpp:57-277)
//   DepthAligner::RunStep        (cu_depthaligner.cpp:32-110)
// Warp / SquaredError: lane = pixel, grid-stride over 64-pixel chunks; SE3 step / EvaluateError: the row walk below.  All reductions: wave
// fold -> LDS across the 4 waves in fixed order -> one 32-float partial per workgroup -> k_finalize_rows sums the partials in double
// in fixed order.  Inlier counts travel as exact floats (< 2^24 per workgroup) and are summed in double.
#include "dfx_device.hpp"
#include "dfx_kernels.hpp"

namespace dfx {

constexpr int kT = 256;   // threads per workgroup (4 waves)

template <int N>
__device__ __forceinline__ void block_reduce_store(float (&v)[N], float* __restrict__ out_row) {
  static_assert(N <= kSimpleRow, "partial row too small");
  __shared__ float red[kT / 64][kSimpleRow];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < N; ++q) {
    const float s = wave_sum(v[q]);
    if (lane == 0) red[wave][q] = s;
  }
  __syncthreads();
  if (threadIdx.x < kSimpleRow) {
    float s = 0.f;
    if (threadIdx.x < N) s = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    out_row[threadIdx.x] = s;
  }
}

__device__ __forceinline__ Geo geo_from(const SimplePairDev& p) {
  Geo g;
#pragma unroll
  for (int q = 0; q < 9; ++q) g.R[q] = p.R[q];
  g.t[0] = p.t[0]; g.t[1] = p.t[1]; g.t[2] = p.t[2];
  g.fx = p.fx; g.fy = p.fy; g.u0 = p.u0; g.v0 = p.v0; g.w = p.w; g.h = p.h;
  return g;
}

// ---- the row walk shared by the pixel reductions (SE3 step, EvaluateError) --------------------------------------------------------------
// Round 4.  Rounds 1-3 walked pixels thread by thread (lane = pixel, grid stride) with the reference-order IEEE geometry for every pixel:
// 195 / 108 vector-ALU instructions per pixel and two dependent memory round trips per iteration; the batched kernels sat at 0.43 / 0.46
// of the HBM roofline.  Now:
//  * a wave owns a 64-pixel-wide column BAND and walks down a segment of its rows: x (and the ray K^-1 x) is loop-invariant per lane, the
//    row is wave-uniform, so every streaming load is `buffer_load  voffset = x * 4 (constant), soffset = y * pitch (scalar unit)` -- no
//    vector-ALU address arithmetic at all, 256 contiguous bytes per wave-load, adjacent waves of a workgroup take adjacent bands;
//  * a software pipeline DT rows deep (DFX_TAP_DIST_*): the depth / intensity of row y + 2 DT are loaded (read once: non-temporal), the
//    geometry of row y + DT is evaluated and its bilinear taps are issued, row y is consumed (the compiler counts the `vmcnt` waits: loads
//    return in order); the row's ray-table entry comes through the scalar cache;
//  * the taps: what bounds these kernels is the FORM of the tap loads, not bytes or arithmetic (tools/ubench/band_walk.cpp):
//    img1 as four dword loads (contiguous from lane to lane under a coherent warp), grad1 as two 16-byte loads; a tap coordinate within
//    2^-13 pixel of an integer is snapped to it so that floor() does not flip from lane to lane at (near-)integer warps;
//  * FAST geometry (FastGeo, dfx_kernels.hpp): fused multiply-adds, one v_rcp_f32, validity as a margin in homogeneous coordinates.  The
//    inlier set is still EXACTLY the reference's: a wave with a pixel whose margin is within the error bound E of zero re-evaluates those
//    pixels in the reference's operation order (find_correspondence_ray<true>) -- a wave-uniform branch taken for a ~1e-3-pixel band
//    along the view border.  The sums are within a few ulp per pixel of the reference-order arithmetic (the parity tests' tolerance is
//    1e-4 of each entry's own Cauchy-Schwarz scale sqrt(JtJ_ii JtJ_jj) against the fp64 oracle, tests/helpers.py);
//  * the Huber weight is folded into 1 / q.z before the pose row (J = [a | (R p) x a], a = -w grad D: warping.h:156-164 restated), the
//    inlier count rides the scalar unit (s_bcnt1 of the validity mask), and the 28 + 1 sums of a wave are folded with
//    v_permlane32_swap / v_permlane16_swap + four DPP row shifts (70 vector-ALU instructions instead of 29 64-lane shuffle ladders).
constexpr int kBand = 64;
#ifndef DFX_RW_UNROLL
#define DFX_RW_UNROLL 1   // rotations of the row states per loop iteration
#endif
// rows between the issue of a row's bilinear taps and their use (the depth of the software pipeline, see row_walk), per operator: the
// SE3 step carries 21 registers of row state per stage (5 waves per SIMD at depth 1; depth 2 costs a wave: 176 vs 165 us per 128 pairs),
// EvaluateError 8 (depth 2: 92 -> 87 us; depth 3 no better) -- profiles/r04_launch_shape.txt
#ifndef DFX_TAP_DIST_SE3
#define DFX_TAP_DIST_SE3 1
#endif
#ifndef DFX_TAP_DIST_ERR
#define DFX_TAP_DIST_ERR 2
#endif

__device__ __forceinline__ float rfl(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
// AUX = cache policy bits of the instruction (0 default, 2 = nt)
template <int AUX = 0>
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, AUX));
}
__device__ __forceinline__ f32x2 bload2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

struct RowIn { float d, i0, ry; };   // loads of one row of the band: depth, intensity (per lane), (y - v0) / fy (wave-uniform)
template <bool GRAD>
struct RowPix {                       // geometry of one row + its taps (possibly still in flight)
  float i0, ax, ay;
  float iz, U, V, vx, vy, vz;         // GRAD (SE3 step) only: 1 / q.z, u - u0, v - v0, R p
  unsigned vmask;                     // all ones: the pixel has a correspondence (and belongs to the wave's share)
  f32x2 ia, ib;                       // img1 rows iy, iy + 1: (x, x + 1)
  f32x4 ga, gb;                       // grad1 rows iy, iy + 1: (gx, gy)(x), (gx, gy)(x + 1)
};

// pose-dependent part of the ambiguity band on the device (the tracker's pose lives in device memory): E = e1 |d| + e2, derive_fast_geo
__device__ __forceinline__ void fast_band(const float (&R)[9], const float (&t)[3], const FastCam& c, float& e1, float& e2) {
  float rho = 0.f, tau = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    rho = fmaxf(rho, fabsf(R[3 * i]) * c.rxmax + fabsf(R[3 * i + 1]) * c.rymax + fabsf(R[3 * i + 2]));
    tau = fmaxf(tau, fabsf(t[i]));
  }
  e1 = rfl(c.gscale * rho * 1.00001f);
  e2 = rfl(c.gscale * tau * 1.00001f);
}

// Walks the calling wave's share of the image; `consume(const RowPix<GRAD>&)` is called once per row with exec = the row's inliers.
// Returns the wave's inlier count (wave-uniform).
template <bool GRAD, int DT, typename F>
__device__ __forceinline__ unsigned row_walk(const SimplePairDev& p, const float (&R)[9], const float (&t)[3], const float e1, const float e2,
                                             const int W, const int H, F&& consume) {
  const FastGeo& fg = p.fg;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned HB = (unsigned)H;
  const __amdgpu_buffer_rsrc_t rI0 = make_rsrc(p.img0, p.pitch_img0 * HB), rD0 = make_rsrc(p.dpt0, p.pitch_dpt0 * HB);
  const __amdgpu_buffer_rsrc_t rI1 = make_rsrc(p.img1, p.pitch_img1 * HB);
  const __amdgpu_buffer_rsrc_t rG1 = make_rsrc(GRAD ? p.grad1 : p.img1, (GRAD ? p.pitch_grad1 : p.pitch_img1) * HB);
  const __amdgpu_buffer_rsrc_t rRay = make_rsrc(p.ray_tab, (unsigned)(W + H) * 4u);
  // the row part of the ray table is wave-uniform: read through the scalar cache (s_load_dword), not the vector memory pipeline -- the
  // reductions are bound by the number of vector-memory instructions per row (tools/ubench/band_walk.cpp: a broadcast dword load per row
  // costs the EvaluateError skeleton 6 us of 82 per 128 pairs; the kernels: EvaluateError 114.3 -> 110.3 us, SE3 step 189.5 -> 185.7)
  typedef const float __attribute__((address_space(4)))* ConstF;
  const ConstF ray_rows = (ConstF)(unsigned long long)(p.ray_tab + W);
  // tap offsets are relative to pixel (icx, icy): fold it into a scalar
  const unsigned c1 = (unsigned)fg.icy * p.pitch_img1 + (unsigned)fg.icx * 4u;
  const unsigned cg = GRAD ? (unsigned)fg.icy * p.pitch_grad1 + (unsigned)fg.icx * 8u : 0u;
  unsigned four = 4u;
  asm volatile("" : "+s"(four));
  const float ixmax = (float)(W - 2 - fg.icx), iymax = (float)(H - 2 - fg.icy);   // last cell of the tap grid, relative to pixel (icx, icy)
  Geo g;   // the reference-order fall-back
#pragma unroll
  for (int q = 0; q < 9; ++q) g.R[q] = R[q];
  g.t[0] = t[0]; g.t[1] = t[1]; g.t[2] = t[2];
  g.fx = p.fx; g.fy = p.fy; g.u0 = p.u0; g.v0 = p.v0; g.w = p.w; g.h = p.h;

  // items = (band, row segment): all bands of a segment are adjacent items, so the waves of a workgroup read adjacent 256-byte runs
  const unsigned nb = ((unsigned)W + kBand - 1) / kBand;
  const unsigned waves = gridDim.x * (kT / 64);
  unsigned nseg = waves / nb;
  nseg = nseg > HB ? HB : nseg;
  nseg = nseg < 1 ? 1 : nseg;
  unsigned rps = (HB + nseg - 1) / nseg;                // rows per segment ...
  constexpr unsigned kUnr = (unsigned)((DT + 1) * DFX_RW_UNROLL);
  rps = (rps + 2 * DT + kUnr - 1) / kUnr * kUnr - 2 * DT;   // ... such that the warm-up + the rows are whole groups of the unrolled loop
  const unsigned nitems = nb * ((HB + rps - 1) / rps);
  unsigned inliers = 0;
  for (unsigned item = blockIdx.x * (kT / 64) + (unsigned)wave; item < nitems; item += waves) {
    const unsigned seg = item / nb, band = item - seg * nb;
    const int y0 = (int)(seg * rps), y1 = (int)(seg * rps + rps < HB ? seg * rps + rps : HB);
    const int x = (int)band * kBand + lane;
    const bool lane_ok = x < W;
    const unsigned voff = (unsigned)(lane_ok ? x : W - 1) * 4u;   // lanes past the last column repeat it (never consumed)
    const float rx = bload1(rRay, voff, 0);
    // loop-invariant per lane: the column part of (K) R ray, so that a row's (K) R ray is ONE fma per component (an fma reads one scalar
    // register: `R1 * ry + R2` with two of them costs a move)
    const float* M = GRAD ? R : fg.KR;
    const float cx = __builtin_fmaf(M[0], rx, M[2]), cy = __builtin_fmaf(M[3], rx, M[5]), cz = __builtin_fmaf(M[6], rx, M[8]);
    float e2v = e2;
    asm volatile("" : "+v"(e2v));   // keep it in a vector register: as a second scalar operand of the fma below it costs a move per row
    const unsigned lanemask = lane_ok ? ~0u : 0u;
    auto load_row = [&](int y) {
      const unsigned yc = (unsigned)(y < y1 ? y : y1 - 1);        // rows past the segment repeat its last row (masked)
      RowIn L;
      // depth and intensity are read ONCE: non-temporal, so that they do not push the img1 / grad1 rows the taps re-read (the next row of this
      // band, the neighbouring band) out of the L2 / Infinity Cache.  The operators are bound by the memory system (profiles/
      // r04_rowwalk_memory_bound.txt): EvaluateError 120 -> 110 us per 128 pairs, SE3 step -1 %; `nt` on the taps themselves costs 5 - 30 %.
      L.d = bload1<2>(rD0, voff, yc * p.pitch_dpt0);
      L.i0 = bload1<2>(rI0, voff, yc * p.pitch_img0);
      L.ry = ray_rows[yc];
      return L;
    };
    // geometry of one row + issue of its taps.  `rowmask` (scalar): all ones if the row belongs to the segment
    auto geom = [&](const RowIn& L, RowPix<GRAD>& S, const unsigned rowmask) {
      const float d = L.d, ry = L.ry;
      S.i0 = L.i0;
      float X, Y, Z;
      const float rrx = __builtin_fmaf(M[1], ry, cx), rry = __builtin_fmaf(M[4], ry, cy), rrz = __builtin_fmaf(M[7], ry, cz);
      if constexpr (GRAD) {
        S.vx = rrx * d; S.vy = rry * d; S.vz = rrz * d;
        Z = S.vz + t[2];
        X = __builtin_fmaf(p.fx, S.vx + t[0], fg.cu * Z);
        Y = __builtin_fmaf(p.fy, S.vy + t[1], fg.cv * Z);
      } else {
        X = __builtin_fmaf(rrx, d, fg.Kt[0]);
        Y = __builtin_fmaf(rry, d, fg.Kt[1]);
        Z = __builtin_fmaf(rrz, d, fg.Kt[2]);
      }
      float iz = __builtin_amdgcn_rcpf(Z);
      // |u_c| < hw and |v_c| < hh and q.z > 0  <=>  max(|X| - hw Z, |Y| - hh Z) < 0
      const float mu = __builtin_fmaf(-fg.hw, Z, fabsf(X)), mv = __builtin_fmaf(-fg.hh, Z, fabsf(Y));
      const float E = __builtin_fmaf(e1, fabsf(d), e2v);
      bool valid = (mu < -E) && (mv < -E);                 // NaN anywhere -> false
      const bool amb = fabsf(fmaxf(mu, mv)) < E;
      float tu = __builtin_fmaf(X, iz, fg.fcx), tv = __builtin_fmaf(Y, iz, fg.fcy);   // tap coordinates relative to pixel (icx, icy)
      float U = 0.f, V = 0.f;
      if constexpr (GRAD) { U = __builtin_fmaf(X, iz, fg.du); V = __builtin_fmaf(Y, iz, fg.dv); }
      if (amb) {   // rare (a ~1e-3-pixel band along the view border): the reference's operation order decides, for these lanes only
        const Corr c = find_correspondence_ray<true>(g, rx, ry, d, 1.0f, 0.0f);
        valid = c.valid;
        tu = c.u - (float)fg.icx;                          // and supplies the coordinates: its taps are in range where it says valid
        tv = c.v - (float)fg.icy;
        if constexpr (GRAD) { U = c.u - p.u0; V = c.v - p.v0; iz = c.iz; }
      }
      // validity travels to the consuming step as a lane value (a carried bool costs three scalar mask instructions per step and stage)
      const unsigned vm = (valid ? lanemask : 0u) & rowmask;
      S.vmask = vm;
      if constexpr (GRAD) { S.iz = iz; S.U = U; S.V = V; }
      // A tap coordinate less than 2^-13 pixel BELOW an integer is that integer (floor of the biased coordinate, weight clamped at 0): the fast
      // projection itself is only good to ~3e-5 pixel (E above), so the snapped position is as right as the computed one -- and at (near-)
      // integer warps, the exact identity first of all, it keeps floor() from flipping between x and x - 1 with the rounding noise from lane to
      // lane, which would cost the dword taps below their contiguity (SE3 step at the identity: 224 us per 128 pairs with the flips, 165 with
      // the snap = what real poses take; profiles/r04_tap_loads.txt).  Coordinates just ABOVE an integer need nothing: their floor is stable.
      // (never past the last cell: an inlier with u in [W - 1 - 2^-13, W - 1) keeps ix = W - 2 with a weight of almost 1 -- its right-hand tap must stay
      // inside the row, a zero weight does not neutralise a NaN in whatever lies behind it)
      const float fu = fminf(floorf(tu + 0x1p-13f), ixmax), fv = fminf(floorf(tv + 0x1p-13f), iymax);
      S.ax = fmaxf(tu - fu, 0.f); S.ay = fmaxf(tv - fv, 0.f);
      const int ix = (int)fu, iy = (int)fv;
      // a lane without correspondence reads offset 0: in range, never used (and no wave-load is ever entirely out of range)
      const unsigned o1 = (unsigned)(__mul24(iy, (int)p.pitch_img1) + ((ix << 2) + (int)c1)) & vm;
      // four dword loads, not two 8-byte ones: an 8-byte load at a 4-byte lane stride (every lane's pair overlaps its neighbour's) costs the
      // texture addresser 14 cycles per wave, a dword load whose lanes are (nearly) contiguous 4 (profiles/r04_ubench_vmem_issue_cost.txt);
      // the reductions' skeleton runs 11 % faster with them (profiles/r04_band_walk.txt), EvaluateError 108 -> 101 us, the SE3 step 181 ->
      // 172 us per 128 pairs (profiles/r04_tap_loads.txt; the snap above keeps them contiguous at the identity).
      S.ia.x = bload1(rI1, o1, 0);
      S.ia.y = bload1(rI1, o1, four);                       // (`four` is opaque: the compiler would fuse the pairs back into 8-byte loads)
      S.ib.x = bload1(rI1, o1, p.pitch_img1);
      S.ib.y = bload1(rI1, o1, p.pitch_img1 + four);
      if constexpr (GRAD) {
        const unsigned og = (unsigned)(__mul24(iy, (int)p.pitch_grad1) + ((ix << 3) + (int)cg)) & vm;
        S.ga = bload4(rG1, og, 0);
        S.gb = bload4(rG1, og, p.pitch_grad1);
      }
    };
    auto eat = [&](const RowPix<GRAD>& S) {
      const bool v = S.vmask != 0;
      inliers += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(v));
      if (v) consume(S);
    };
    // Software pipeline, DT rows deep: at step y the loads of row y + 2 DT are issued, the geometry of row y + DT is evaluated and its
    // taps are issued, row y is consumed.  The DT + 1 row states rotate through the unrolled steps, so nothing is copied.  The warm-up
    // runs INSIDE the loop and without branches: steps y0 - 2 DT .. y0 - 1 run every stage on zero-initialised / repeated rows whose
    // lanes are masked (`rowmask`), so the loop header sees the same queue of outstanding loads from the preheader and from the back
    // edge (the compiler's vmcnt counts stay exact; a separate prologue made the header wait for all but 4 loads) and a step holds no
    // scalar control flow but the loop itself and the rare border branch.
    constexpr int NS = DT + 1;
    RowIn L[NS];
    RowPix<GRAD> S[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      L[j].d = 0.f; L[j].i0 = 0.f; L[j].ry = 0.f;
      S[j].i0 = S[j].ax = S[j].ay = 0.f; S[j].vmask = 0u;
      S[j].ia = S[j].ib = f32x2{ 0.f, 0.f };
      if constexpr (GRAD) { S[j].iz = S[j].U = S[j].V = S[j].vx = S[j].vy = S[j].vz = 0.f; S[j].ga = S[j].gb = f32x4{ 0.f, 0.f, 0.f, 0.f }; }
    }
    // UNR steps per loop iteration and no exit inside: the steps past the segment's last row are masked like the warm-up's (the
    // segments are sized so that only a wave's last group of a short last segment has any), and the compiler's conservative vmcnt at the
    // loop header (it waits for every load older than the header's own) is paid once per UNR steps
    constexpr int UNR = NS * DFX_RW_UNROLL;
    const int groups = (y1 - y0 + 2 * DT + UNR - 1) / UNR;
    int y = y0 - 2 * DT;
    for (int grp = 0; grp < groups; ++grp) {
#pragma unroll
      for (int jj = 0; jj < UNR; ++jj) {
        const int j = jj % NS;
        L[(j + 2 * DT) % NS] = load_row(y + 2 * DT);
        geom(L[(j + DT) % NS], S[(j + DT) % NS], (y + DT >= y0 && y + DT < y1) ? ~0u : 0u);
        eat(S[j]);
        ++y;
      }
    }
  }
  return inliers;
}

__device__ __forceinline__ float lerpf(float a, float b, float t) { return __builtin_fmaf(t, b - a, a); }
template <bool GRAD>
__device__ __forceinline__ float pix_img(const RowPix<GRAD>& S) { return lerpf(lerpf(S.ia.x, S.ia.y, S.ax), lerpf(S.ib.x, S.ib.y, S.ax), S.ay); }

// v + (v shifted right by N lanes inside its 16-lane row); lanes without a source add 0
template <int CTRL>
__device__ __forceinline__ float row_shr_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// Sum over the 64 lanes of NV = 4 * NQ per-lane values: v_permlane32_swap folds the wave's halves of two values into one register
// (A, B -> [A.lo + A.hi | B.lo + B.hi]), v_permlane16_swap the 16-lane rows of two such registers ([A A B B], [C C D D] -> rows A C B D),
// four DPP row shifts finish inside the rows.  out[j], lane 16 r + 15, then holds the total of value 4 j + {0, 2, 1, 3}[r].
// hipcc (ROCm 7.2, clang 22) turns `s[0] + s[1]` of a permlane swap's two results into `s[0] + s[0]` (tools/ubench/wave_fold_probe.cpp,
// profiles/r04_wave_fold_probe.txt: found there, pinned there): the second result goes through an empty asm so that it stays a value of
// its own.  The asm holds no instruction, so no hazard can hide in it.
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
    r = row_shr_add<0x118>(r);   // row_shr:8
    r = row_shr_add<0x114>(r);   // row_shr:4
    r = row_shr_add<0x112>(r);   // row_shr:2
    r = row_shr_add<0x111>(r);   // row_shr:1
    out[j] = r;
  }
}

// the workgroup's partial row from per-wave rows in LDS: ((w0 + w1) + w2) + w3, fixed order
__device__ __forceinline__ void fold_waves_store(float (&red)[kT / 64][kSimpleRow], const int nvals, float* __restrict__ out_row) {
  __syncthreads();
  if (threadIdx.x < kSimpleRow) {
    float s = 0.f;
    if ((int)threadIdx.x < nvals) s = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    out_row[threadIdx.x] = s;
  }
}

// ---- SE3 step: 21 JtJ + 6 Jtr + r^2 + inliers = 29 floats per workgroup row ---------------------------------------------------------
// One SE3 Gauss-Newton step (lucas_kanade_se3.h:41-77) over this workgroup's share of the pair; shared by the blocking operator, the
// batched form and the device-resident tracker (R, t = the pose the step is evaluated at).
__device__ __forceinline__ void se3_step_body(const SimplePairDev& p, const float (&R)[9], const float (&t)[3], const float e1, const float e2,
                                              const int W, const int H, const float huber_delta, float* __restrict__ out_row) {
  __shared__ float red[kT / 64][kSimpleRow];
  float acc[28];
#pragma unroll
  for (int q = 0; q < 28; ++q) acc[q] = 0.f;
  const float fx = p.fx, fy = p.fy;
  const unsigned inl = row_walk<true, DFX_TAP_DIST_SE3>(p, R, t, e1, e2, W, H, [&](const RowPix<true>& S) {
    const float gx = lerpf(lerpf(S.ga.x, S.ga.z, S.ax), lerpf(S.gb.x, S.gb.z, S.ax), S.ay);
    const float gy = lerpf(lerpf(S.ga.y, S.ga.w, S.ax), lerpf(S.gb.y, S.gb.w, S.ax), S.ay);
    float r = S.i0 - pix_img(S);
    const float wgt = huber_weight(r, huber_delta);
    r *= wgt;
    // J = -w grad D [I | -hat(R p)] = [a | (R p) x a],  a = -w grad D,  D = ProjectPointJacobian(q)  (pinhole_camera_impl.h:91-97)
    const float wz = wgt * S.iz;
    float J[6];
    J[0] = -(gx * fx) * wz;
    J[1] = -(gy * fy) * wz;
    J[2] = __builtin_fmaf(gx, S.U, gy * S.V) * wz;
    J[3] = S.vy * J[2] - S.vz * J[1];
    J[4] = S.vz * J[0] - S.vx * J[2];
    J[5] = S.vx * J[1] - S.vy * J[0];
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = a; b < 6; ++b) { acc[k] = __builtin_fmaf(J[a], J[b], acc[k]); ++k; }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] = __builtin_fmaf(J[a], r, acc[21 + a]);
    acc[27] = __builtin_fmaf(r, r, acc[27]);
  });
  float f[7];
  wave_fold4<7>(acc, f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((lane & 15) == 15) {
    const int r = lane >> 4;
    const int slot = ((r & 1) << 1) | (r >> 1);   // rows hold values {0, 2, 1, 3} of their group of four
#pragma unroll
    for (int j = 0; j < 7; ++j) red[wave][4 * j + slot] = f[j];
  }
  if (lane == 0) red[wave][28] = (float)inl;      // exact: < 2^24 pixels per wave
  fold_waves_store(red, 29, out_row);
}

__global__ __launch_bounds__(kT) void k_se3_step(const SimplePairDev p, const int W, const int H, const float huber_delta,
                                                 float* __restrict__ partials) {
  se3_step_body(p, p.R, p.t, p.fg.e1, p.fg.e2, W, H, huber_delta, partials + (size_t)blockIdx.x * kSimpleRow);
}

// ---- device-resident tracker (CameraTracker::TrackFrame, reference core/system/camera_tracker.cpp:42-71) -------------
// The pose lives in device memory between iterations, and an iteration is ONE launch: every workgroup of k_se3_step_dev first folds the partial rows of the
// PREVIOUS evaluation (double, fixed order), solves the 6x6 normal equations by LDL^T in double and applies the reference's update (t += dt, R = exp(dw) R;
// lucas_kanade_se3.h:85-95) -- all workgroups compute the same bits from the same rows -- and then evaluates its rows of the image at the new pose.  Workgroup 0
// also writes the new state out (states are an array over iterations: nobody overwrites what another workgroup still reads, and the partial rows alternate
// between two buffers).  k_track_final applies the last evaluation.  Rounds 2-4 ran a separate update kernel behind every step kernel: two dependent launches of
// 3.5-5.6 us each per iteration (profiles/r05_tracker.txt); a grid-wide hand-over inside one kernel was measured slower in round 3.  A whole coarse-to-fine
// schedule is enqueued without host syncs.
struct TrackState {      // device-resident
  double R[9], t[3];     // pose_ck, kept in double across iterations
  float Rf[9], tf[3];    // fp32 copy consumed by the step kernel
  float last_residual; float last_inliers; int solver_failures; int iterations_done;
};
struct TrackLds { double red[32][kSimpleRow]; double sum[kSimpleRow]; TrackState st; };

__device__ __forceinline__ double bcast(const double v, const int src_lane) {   // wave-uniform copy of lane `src_lane`'s value
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src_lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src_lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// One WAVE: normal equations in `sum` (the fp64 fold of the partial rows) + the state they were evaluated at -> the next state.
// reference precision: the item is fp32 (JTJJrReductionItem<float,6>) before the solve (camera_tracker.cpp:59); LDL^T (no pivoting), forward / diagonal /
// backward substitution in double; update = -x ; t += update[0:3] ; R = exp(update[3:6]) * R   (lucas_kanade_se3.h:85-95).
// A wave issues one instruction every four cycles whether or not it depends on the last one, so what this costs is its instruction COUNT, and every workgroup of
// a tracker iteration waits for it: one thread walking the 6x6 system was ~700 double-precision instructions (2 us of a 9 us iteration).  Here lane i holds row i
// of the matrix (rows below the pivot are eliminated by all lanes at once, pivot rows travel by v_readlane), the substitutions run on wave-uniform values, and lane
// j holds column j of the rotation for the retraction: ~250 instructions.  Everything is unrolled: the small matrices are REGISTERS (as dynamically indexed private
// arrays they landed in scratch memory; as LDS arrays every dependent access paid an LDS round trip).  The 6 divisions by the pivots are refined reciprocals
// (v_rcp_f64 + three Newton steps: relative error < 2^-52), the exponential's sin / cos factors their Taylor series in theta^2 for the small rotations a tracker
// update has (|theta| < 1.5: twelve terms, < 1e-16; the library functions beyond).
__device__ __forceinline__ void track_solve_wave(const double* sum, const TrackState* __restrict__ in, TrackState& out) {
  const int lane = threadIdx.x & 63;
  const int i = lane < 6 ? lane : 5, jc = lane < 3 ? lane : 2;
  const double Ro0 = in->R[jc], Ro1 = in->R[3 + jc], Ro2 = in->R[6 + jc], tin = in->t[jc];   // column jc of R, t[jc]
  double row[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int a = i < q ? i : q, b = i < q ? q : i;   // upper triangle, row-major: (a, b) -> a (13 - a) / 2 + b - a
    row[q] = (double)(float)sum[(a * (13 - a)) / 2 + (b - a)];
  }
  double acc = (double)(float)sum[21 + i];
  bool ok = sum[28] > 0.0;
  auto rcp = [](double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return r;
  };
  double D[6], Di[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    // lanes i > j: L[i][j] = (A[i][j] - sum_q L[i][q] L[j][q] D[q]) / d_j ; lane j itself computes d_j by the same expression
    double v = row[j];
#pragma unroll
    for (int q = 0; q < j; ++q) v = __builtin_fma(-row[q], bcast(row[q], j) * D[q], v);
    const double d = bcast(v, j);
    ok = ok && (fabs(d) > 0.0);
    D[j] = d;
    Di[j] = rcp(d);
    row[j] = v * Di[j];
  }
  if (!ok) {   // the pose stays (the update of a failed solve is not applied)
    if (lane == 0) {
      out.last_residual = (float)sum[27]; out.last_inliers = (float)sum[28];
      out.iterations_done = in->iterations_done + 1; out.solver_failures = in->solver_failures + 1;
#pragma unroll
      for (int q = 0; q < 9; ++q) { out.R[q] = in->R[q]; out.Rf[q] = in->Rf[q]; }
#pragma unroll
      for (int q = 0; q < 3; ++q) { out.t[q] = in->t[q]; out.tf[q] = in->tf[q]; }
    }
    return;
  }
  double z[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {   // forward: y_q is final in lane q once the lanes above it have been subtracted
    const double y = bcast(acc, q);
    acc = __builtin_fma(-row[q], y, acc);
    z[q] = y * Di[q];
  }
#pragma unroll
  for (int q = 5; q >= 1; --q)    // backward, on uniform values: x_q = z[q] is final, L[q][i] lives in lane q
#pragma unroll
    for (int r = 0; r < q; ++r) z[r] = __builtin_fma(-bcast(row[r], q), z[q], z[r]);
  const double w0 = -z[3], w1 = -z[4], w2 = -z[5];
  const double th2 = w0 * w0 + w1 * w1 + w2 * w2;
  double Ac, Bc;   // sin(theta) / theta, (1 - cos(theta)) / theta^2
  if (th2 < 2.25) {
    // sum_k (-1)^k th2^k / (2k+1)!  and  sum_k (-1)^k th2^k / (2k+2)!, Horner from k = 11
    double sa = 1.0 / 25852016738884976640000.0, sb = 1.0 / 620448401733239439360000.0;   // 1 / 23!, 1 / 24!
    const double ia[11] = { 1.0 / 51090942171709440000.0, 1.0 / 121645100408832000.0, 1.0 / 355687428096000.0, 1.0 / 1307674368000.0, 1.0 / 6227020800.0, 1.0 / 39916800.0,
                            1.0 / 362880.0, 1.0 / 5040.0, 1.0 / 120.0, 1.0 / 6.0, 1.0 };   // 1 / 21!, 19!, ... 1!
    const double ib[11] = { 1.0 / 1124000727777607680000.0, 1.0 / 2432902008176640000.0, 1.0 / 6402373705728000.0, 1.0 / 20922789888000.0, 1.0 / 87178291200.0, 1.0 / 479001600.0,
                            1.0 / 3628800.0, 1.0 / 40320.0, 1.0 / 720.0, 1.0 / 24.0, 1.0 / 2.0 };   // 1 / 22!, 20!, ... 2!
#pragma unroll
    for (int k = 0; k < 11; ++k) { sa = __builtin_fma(-th2, sa, ia[k]); sb = __builtin_fma(-th2, sb, ib[k]); }
    Ac = sa; Bc = sb;
  } else {
    const double th = sqrt(th2);
    Ac = sin(th) / th; Bc = (1.0 - cos(th)) / th2;
  }
  // exp(w) = I + Ac K + Bc K^2 with K = [w]x, K^2 = w w^T - theta^2 I (its diagonal written without the cancellation)
  const double p01 = Bc * w0 * w1, p02 = Bc * w0 * w2, p12 = Bc * w1 * w2;
  const double E00 = 1.0 - Bc * (w1 * w1 + w2 * w2), E01 = p01 - Ac * w2, E02 = p02 + Ac * w1;
  const double E10 = p01 + Ac * w2, E11 = 1.0 - Bc * (w0 * w0 + w2 * w2), E12 = p12 - Ac * w0;
  const double E20 = p02 - Ac * w1, E21 = p12 + Ac * w0, E22 = 1.0 - Bc * (w0 * w0 + w1 * w1);
  const double Rn0 = E00 * Ro0 + E01 * Ro1 + E02 * Ro2, Rn1 = E10 * Ro0 + E11 * Ro1 + E12 * Ro2, Rn2 = E20 * Ro0 + E21 * Ro1 + E22 * Ro2;
  const double tn = tin - (lane == 0 ? z[0] : lane == 1 ? z[1] : z[2]);
  if (lane < 3) {
    out.R[lane] = Rn0; out.R[3 + lane] = Rn1; out.R[6 + lane] = Rn2;
    out.Rf[lane] = (float)Rn0; out.Rf[3 + lane] = (float)Rn1; out.Rf[6 + lane] = (float)Rn2;
    out.t[lane] = tn; out.tf[lane] = (float)tn;
  }
  if (lane == 0) {
    out.last_residual = (float)sum[27]; out.last_inliers = (float)sum[28];
    out.iterations_done = in->iterations_done + 1; out.solver_failures = in->solver_failures;
  }
}

// All kT threads of a workgroup: fold `nblocks` partial rows (thread = one float4 column of every 32nd row, up to 16 loads in flight; rows ascending within a
// thread, row groups ascending in the second stage: the same bits in every workgroup and for every grid size), wave 0 solves into l.st.
__device__ __forceinline__ void track_fold_and_solve(const float* __restrict__ partials, const int nblocks, const TrackState* __restrict__ in, TrackLds& l) {
  static_assert(kT == 256 && kSimpleRow == 32, "8 float4 columns x 32 row groups");
  const int c = threadIdx.x & 7, rg = threadIdx.x >> 3;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int b0 = rg; b0 < nblocks; b0 += 32 * 16) {
    f32x4 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int b = b0 + 32 * q;
      v[q] = b < nblocks ? *reinterpret_cast<const f32x4*>(partials + (size_t)b * kSimpleRow + c * 4) : f32x4{ 0.f, 0.f, 0.f, 0.f };
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) { s0 += (double)v[q][0]; s1 += (double)v[q][1]; s2 += (double)v[q][2]; s3 += (double)v[q][3]; }
  }
  l.red[rg][c * 4 + 0] = s0; l.red[rg][c * 4 + 1] = s1; l.red[rg][c * 4 + 2] = s2; l.red[rg][c * 4 + 3] = s3;
  __syncthreads();
  if (threadIdx.x < kSimpleRow) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += l.red[q][threadIdx.x];
    l.sum[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x < 64) track_solve_wave(l.sum, in, l.st);
  __syncthreads();
}

// blockIdx.y = candidate: Relocalize / the loop-closure geometry checks track ONE live frame against N keyframes
// (deepfactors.cpp:713-743, loop_detector.cpp:146-167); descriptors, states and partials are arrays over candidates.
// nblocks_prev == 0: the first evaluation of a frame, at states_in as the host wrote it.
// BYVAL (one tracker, the camera-rate case): the level's descriptor and -- for the first evaluation -- the initial state travel in the kernel arguments, so a
// frame has no host-to-device copy in front of its first kernel (a 4 us blit kernel and a launch boundary); workgroup 0 of the first evaluation writes the
// initial state to states_out, where the second evaluation's update reads it.
template <bool BYVAL>
__global__ __launch_bounds__(kT) void k_se3_step_dev(const SimplePairDev* __restrict__ descs, const SimplePairDev one, const TrackState st0,
                                                     const TrackState* __restrict__ states_in, TrackState* __restrict__ states_out,
                                                     const float* __restrict__ partials_prev, const int nblocks_prev, const int W, const int H,
                                                     const float huber_delta, float* __restrict__ partials_all) {
  const SimplePairDev& p = BYVAL ? one : descs[blockIdx.y];
  float* partials = partials_all + (size_t)blockIdx.y * gridDim.x * kSimpleRow;
  __shared__ TrackLds l;
  float R[9], t[3], e1, e2;
  if (nblocks_prev > 0) {
    track_fold_and_solve(partials_prev + (size_t)blockIdx.y * nblocks_prev * kSimpleRow, nblocks_prev, states_in + blockIdx.y, l);
    if (blockIdx.x == 0 && threadIdx.x < sizeof(TrackState) / 4) reinterpret_cast<uint32_t*>(states_out + blockIdx.y)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&l.st)[threadIdx.x];
#pragma unroll
    for (int q = 0; q < 9; ++q) R[q] = rfl(l.st.Rf[q]);   // wave-uniform: scalar registers, as the loads of the other branch
#pragma unroll
    for (int q = 0; q < 3; ++q) t[q] = rfl(l.st.tf[q]);
  } else {
    const TrackState* st = BYVAL ? &st0 : states_in + blockIdx.y;
#pragma unroll
    for (int q = 0; q < 9; ++q) R[q] = st->Rf[q];
    t[0] = st->tf[0]; t[1] = st->tf[1]; t[2] = st->tf[2];
    if (BYVAL && blockIdx.x == 0 && threadIdx.x == 0) states_out[blockIdx.y] = st0;
  }
  fast_band(R, t, p.fc, e1, e2);   // the descriptor's band belongs to the pose it was filled with, not to the state's
  se3_step_body(p, R, t, e1, e2, W, H, huber_delta, partials + (size_t)blockIdx.x * kSimpleRow);
}

// the update behind the last evaluation of a frame
// (states_out: the caller's mapped host buffer -- the result needs no copy engine behind this kernel)
__global__ __launch_bounds__(kT) void k_track_final(const float* __restrict__ partials_prev, const int nblocks_prev, const TrackState* __restrict__ states_in,
                                                    TrackState* __restrict__ states_out, const DoneFlag done) {
  __shared__ TrackLds l;
  track_fold_and_solve(partials_prev + (size_t)blockIdx.x * nblocks_prev * kSimpleRow, nblocks_prev, states_in + blockIdx.x, l);
  static_assert(sizeof(TrackState) / 4 <= 64, "the state is stored by wave 0");
  if (threadIdx.x < sizeof(TrackState) / 4) reinterpret_cast<uint32_t*>(states_out + blockIdx.x)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&l.st)[threadIdx.x];
  if (threadIdx.x < 64) signal_done_wave(done);
}

size_t track_state_bytes() { return sizeof(TrackState); }

hipError_t launch_track_iteration(const SimplePairDev* descs_dev, int n, const void* states_in, void* states_out, const float* partials_prev, int blocks_prev,
                                  int W, int H, float huber_delta, int blocks, float* partials_dev, hipStream_t stream, const SimplePairDev* one_host,
                                  const void* state0_host) {
  if (one_host && n == 1) {
    TrackState st0{};
    if (state0_host) st0 = *static_cast<const TrackState*>(state0_host);
    hipLaunchKernelGGL(k_se3_step_dev<true>, dim3(blocks, 1), dim3(kT), 0, stream, (const SimplePairDev*)nullptr, *one_host, st0, (const TrackState*)states_in,
                       (TrackState*)states_out, partials_prev, blocks_prev, W, H, huber_delta, partials_dev);
  } else {
    hipLaunchKernelGGL(k_se3_step_dev<false>, dim3(blocks, n), dim3(kT), 0, stream, descs_dev, SimplePairDev{}, TrackState{}, (const TrackState*)states_in,
                       (TrackState*)states_out, partials_prev, blocks_prev, W, H, huber_delta, partials_dev);
  }
  return hipGetLastError();
}
hipError_t launch_track_final(int n, const void* states_in, void* states_out, const float* partials_prev, int blocks_prev, hipStream_t stream, const DoneFlag& done) {
  hipLaunchKernelGGL(k_track_final, dim3(n), dim3(kT), 0, stream, partials_prev, blocks_prev, (const TrackState*)states_in, (TrackState*)states_out, n == 1 ? done : DoneFlag{});
  return hipGetLastError();
}

void track_state_init(void* host_state, const double* R, const double* t) {
  TrackState* s = (TrackState*)host_state;
  for (int i = 0; i < 9; ++i) { s->R[i] = R[i]; s->Rf[i] = (float)R[i]; }
  for (int i = 0; i < 3; ++i) { s->t[i] = t[i]; s->tf[i] = (float)t[i]; }
  s->last_residual = 0.f; s->last_inliers = 0.f; s->solver_failures = 0; s->iterations_done = 0;
}
void track_state_read(const void* host_state, double* R, double* t, float* residual, float* inliers, int* failures, int* iters) {
  const TrackState* s = (const TrackState*)host_state;
  for (int i = 0; i < 9; ++i) R[i] = s->R[i];
  for (int i = 0; i < 3; ++i) t[i] = s->t[i];
  *residual = s->last_residual; *inliers = s->last_inliers; *failures = s->solver_failures; *iters = s->iterations_done;
}

// ---- SparseGeometricFactor::linearize (reference core/gtsam/sparse_geometric_factor.cpp:147-275) ---------------------------
// Per sampled point: decode depth at the point in kf0, warp, decode kf1's depth at the nearest-neighbour pixel of the projection, residual
// err = dpt1 - (R p + t).z, Huber weight, one row [err_J_pose0 (6) | err_J_pose1 (6) | err_J_cde0 (CS) | err_J_cde1 (CS) | err] (zero row if
// no correspondence).  Round 5: ALL factors of a round in one launch (blockIdx.y = factor; the reference linearises every factor of the
// graph inside one ISAM2::update, built per keyframe pair at core/mapping/mapper.cpp:308-311), CS / 4 lanes per point -- both code dot
// products and the two code blocks of the row move as float4 per lane, 16 CS contiguous bytes per point (the layout of update_depth_body)
// -- instead of one lane per point walking 2 CS dwords twice (65 us per blocking factor, 7.8 ms per 120-factor round).  The geometry is
// evaluated redundantly by the lanes of a point (wave-uniform control flow inside a point's lane group).
struct alignas(16) SparseGeoDev {   // (float4 loads of the codes: 16-byte slots)
  float R[9], t[3], M[9], HM[9];
  float fx, fy, u0, v0, w, h;
  float code0[64], code1[64];
  const float* prx0; const float* jac0; const float* prx1; const float* jac1; const float* dgrad1;
  const int* pts;        // device: (x, y) pairs
  float* rows;           // device: [npts][12 + 2 CS + 1]
  uint32_t pitch_prx0, pitch_jac0, pitch_prx1, pitch_jac1, pitch_dgrad1;
  int npts;
  int W, H;              // image size: points are clamped into it (device-resident point lists cannot be checked by the host)
  float huber_delta, avg_dpt;
};
static_assert(sizeof(SparseGeoDev) % 16 == 0 && offsetof(SparseGeoDev, code0) % 16 == 0 && offsetof(SparseGeoDev, code1) % 16 == 0, "float4 loads of the codes");
typedef float f32x4_u4 __attribute__((ext_vector_type(4), aligned(4)));   // rows are 12 + 2 CS + 1 floats apart: dword-aligned vectors

template <int CS>
__global__ __launch_bounds__(kT) void k_sparse_geometric_batch(const SparseGeoDev* __restrict__ descs) {
  constexpr int NC = 12 + 2 * CS + 1;
  constexpr int LPP = CS / 4;        // lanes per point
  constexpr int PPW = 64 / LPP;      // points per wave step
  const SparseGeoDev& P = descs[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane % LPP, grp = lane / LPP;
  const int npts = P.npts;
  Geo g;
#pragma unroll
  for (int k = 0; k < 9; ++k) g.R[k] = P.R[k];
  g.t[0] = P.t[0]; g.t[1] = P.t[1]; g.t[2] = P.t[2];
  g.fx = P.fx; g.fy = P.fy; g.u0 = P.u0; g.v0 = P.v0; g.w = P.w; g.h = P.h;
  const f32x4 c0 = *reinterpret_cast<const f32x4*>(P.code0 + 4 * q), c1 = *reinterpret_cast<const f32x4*>(P.code1 + 4 * q);
  const float a = P.avg_dpt;
  auto dot4 = [](const f32x4& j, const f32x4& c) { return __builtin_fmaf(j.w, c.w, __builtin_fmaf(j.z, c.z, __builtin_fmaf(j.y, c.y, j.x * c.x))); };
  auto lanes_sum = [](float d) {
#pragma unroll
    for (int m = 1; m < LPP; m <<= 1) d += __shfl_xor(d, m, 64);   // butterfly: every lane of the point ends with the same bits
    return d;
  };
  for (int base = (blockIdx.x * (kT / 64) + wave) * PPW; base < npts; base += gridDim.x * (kT / 64) * PPW) {
    const int i = base + grp;
    const bool have = i < npts;
    const int ic = have ? i : npts - 1;
    const int x = min(max(P.pts[2 * ic], 0), P.W - 1), y = min(max(P.pts[2 * ic + 1], 0), P.H - 1);
    const f32x4 j0 = gload<f32x4>((const char*)P.jac0 + (size_t)y * P.pitch_jac0 + ((size_t)x * CS + 4 * q) * 4);
    const float dot0 = lanes_sum(dot4(j0, c0));
    const float d0 = a / (gload<float>((const char*)P.prx0 + (size_t)y * P.pitch_prx0 + (size_t)x * 4) + dot0) - a;
    const Corr c = find_correspondence(g, x, y, d0, 1.0f, 0.0f);
    // lanes of a point agree on c.valid (same inputs, same arithmetic); a point without correspondence taps pixel (0, 0) and writes zeros
    const int nx = c.valid ? (int)c.u : 0, ny = c.valid ? (int)c.v : 0;   // cast<int>: truncation (sparse_geometric_factor.cpp:207)
    const f32x4 j1 = gload<f32x4>((const char*)P.jac1 + (size_t)ny * P.pitch_jac1 + ((size_t)nx * CS + 4 * q) * 4);
    const float dot1 = lanes_sum(dot4(j1, c1));
    const float d1 = a / (gload<float>((const char*)P.prx1 + (size_t)ny * P.pitch_prx1 + (size_t)nx * 4) + dot1) - a;
    const float err = d1 - (c.vz + g.t[2]);
    const f32x2 dg = gload<f32x2>((const char*)P.dgrad1 + (size_t)ny * P.pitch_dgrad1 + (size_t)nx * 8);
    // gC = -(dgrad . C) with C = D [I | -hat(R p)]; third row of [I | -hat(R p)] is (0, 0, 1, v.y, -v.x, 0)
    float gC[6], D00, D02, D11, D12;
    pose_row(g, c, d0, dg.x, dg.y, gC, D00, D02, D11, D12);
    const float a10[6] = { gC[0], gC[1], 1.0f + gC[2], c.vy + gC[3], -c.vx + gC[4], gC[5] };
    float e[12];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      e[j] = a10[0] * P.M[j] + a10[1] * P.M[3 + j] + a10[2] * P.M[6 + j];
      e[3 + j] = a10[3] * P.M[j] + a10[4] * P.M[3 + j] + a10[5] * P.M[6 + j];
      e[6 + j] = -e[j];
      e[9 + j] = -(a10[0] * P.HM[j] + a10[1] * P.HM[3 + j] + a10[2] * P.HM[6 + j]) - e[3 + j];
    }
    const float apd0 = a + d0, apd1 = a + d1;
    const float dprx0 = -(apd0 * apd0) / a;
    const float pj0 = D00 * c.rrx + D02 * c.rrz, pj1 = D11 * c.rry + D12 * c.rrz;
    const float sc0 = (c.rrz - (dg.x * pj0 + dg.y * pj1)) * dprx0;
    const float sc1 = (apd1 * apd1) / a;   // -DepthJacobianPrx(dpt1)
    const float wgt = c.valid ? huber_weight(err, P.huber_delta) : 0.f;
    if (!have) continue;
    float* row = P.rows + (size_t)i * NC;
    const f32x4 z4 = f32x4{ 0.f, 0.f, 0.f, 0.f };
    // pose blocks: three float4 by the first three lanes of the point; the residual by its last lane (LPP >= 4)
    if (q < 3) {
      f32x4 v = q == 0 ? f32x4{ e[0], e[1], e[2], e[3] } : (q == 1 ? f32x4{ e[4], e[5], e[6], e[7] } : f32x4{ e[8], e[9], e[10], e[11] });
      v = v * wgt;
      gstore<f32x4_u4>(row + 4 * q, c.valid ? v : z4);
    }
    if (q == LPP - 1) gstore<float>(row + 12 + 2 * CS, c.valid ? err * wgt : 0.f);
    const f32x4 r0 = (sc0 * j0) * wgt, r1 = (sc1 * j1) * wgt;
    gstore<f32x4_u4>(row + 12 + 4 * q, c.valid ? r0 : z4);
    gstore<f32x4_u4>(row + 12 + CS + 4 * q, c.valid ? r1 : z4);
  }
}

size_t sparse_geo_desc_bytes() { return sizeof(SparseGeoDev); }
void sparse_geo_fill(void* desc, const float* R, const float* t, const float* M, const float* HM, const float* cam6, const float* code0, const float* code1, int cs,
                     const float* prx0, uint32_t pp0, const float* jac0, uint32_t pj0, const float* prx1, uint32_t pp1, const float* jac1, uint32_t pj1,
                     const float* dgrad1, uint32_t pg1, const int* pts_dev, int npts, int W, int H, float* rows_dev, float huber_delta, float avg_dpt) {
  SparseGeoDev* d = (SparseGeoDev*)desc;
  for (int i = 0; i < 9; ++i) { d->R[i] = R[i]; d->M[i] = M[i]; d->HM[i] = HM[i]; }
  for (int i = 0; i < 3; ++i) d->t[i] = t[i];
  d->fx = cam6[0]; d->fy = cam6[1]; d->u0 = cam6[2]; d->v0 = cam6[3]; d->w = cam6[4]; d->h = cam6[5];
  for (int i = 0; i < 64; ++i) { d->code0[i] = i < cs ? code0[i] : 0.f; d->code1[i] = i < cs ? code1[i] : 0.f; }
  d->prx0 = prx0; d->jac0 = jac0; d->prx1 = prx1; d->jac1 = jac1; d->dgrad1 = dgrad1;
  d->pts = pts_dev; d->rows = rows_dev; d->npts = npts; d->W = W; d->H = H;
  d->pitch_prx0 = pp0; d->pitch_jac0 = pj0; d->pitch_prx1 = pp1; d->pitch_jac1 = pj1; d->pitch_dgrad1 = pg1;
  d->huber_delta = huber_delta; d->avg_dpt = avg_dpt;
}
// ---- the factors' normal equations on the device: G = [A | b]^T [A | b] per factor (upper triangle, row-major) ---------------------------------------------
// gtsam eliminates a JacobianFactor by forming exactly this product on the host; the rows of a 1024-factor graph are 157 MB per round (3 ms into pinned
// memory, 32 into pageable), their Gram blocks 12 MB.  One workgroup per factor: tiles of 32 rows are staged in LDS (the rows are contiguous in memory: a flat,
// coalesced copy), thread t owns the entries t, t + 256, ... of the triangle and adds row after row in ascending order (fp32 fma; deterministic).  Rows of
// invalid points are zero rows (k_sparse_geometric_batch writes them so) and add nothing.
template <int CS>
__global__ __launch_bounds__(256) void k_rows_gram(const SparseGeoDev* __restrict__ descs, float* __restrict__ gram_all) {
  constexpr int NC = 12 + 2 * CS + 1, NE = NC * (NC + 1) / 2, EPT = (NE + 255) / 256, TR = 32, LD = NC + 1;
  __shared__ float tile[TR * LD];
  const SparseGeoDev& d = descs[blockIdx.x];
  const float* __restrict__ rows = d.rows;
  const int n = d.npts;
  int oi[EPT], oj[EPT];   // LDS offsets of the entry's two columns
  float acc[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    int e = (int)threadIdx.x + 256 * k, i = 0;
    if (e >= NE) e = NE - 1;                       // (padding entries recompute the last one; never stored)
    while (e >= NC - i) { e -= NC - i; ++i; }      // row-major upper triangle: row i holds NC - i entries
    oi[k] = i; oj[k] = i + e; acc[k] = 0.f;
  }
  for (int r0 = 0; r0 < n; r0 += TR) {
    const int nr = n - r0 < TR ? n - r0 : TR;
    for (int idx = threadIdx.x; idx < TR * NC; idx += 256) {
      const int r = idx / NC, c = idx - r * NC;
      tile[r * LD + c] = r < nr ? rows[(size_t)r0 * NC + idx] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < TR; ++r) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) acc[k] = __builtin_fmaf(tile[r * LD + oi[k]], tile[r * LD + oj[k]], acc[k]);
    }
    __syncthreads();
  }
  float* out = gram_all + (size_t)blockIdx.x * NE;
#pragma unroll
  for (int k = 0; k < EPT; ++k) { const int e = (int)threadIdx.x + 256 * k; if (e < NE) out[e] = acc[k]; }
}

hipError_t launch_rows_gram(int cs, const void* descs_dev, int n_factors, float* gram_dev, hipStream_t stream) {
  const SparseGeoDev* d = (const SparseGeoDev*)descs_dev;
  switch (cs) {
    case 16: hipLaunchKernelGGL(k_rows_gram<16>, dim3(n_factors), dim3(256), 0, stream, d, gram_dev); break;
    case 32: hipLaunchKernelGGL(k_rows_gram<32>, dim3(n_factors), dim3(256), 0, stream, d, gram_dev); break;
    case 64: hipLaunchKernelGGL(k_rows_gram<64>, dim3(n_factors), dim3(256), 0, stream, d, gram_dev); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_sparse_geometric_batch(int cs, const void* descs_dev, int n_factors, int max_points, hipStream_t stream) {
  const int ppb = (kT / 64) * (64 / (cs / 4));        // points per workgroup step
  int bx = (max_points + ppb - 1) / ppb;
  const int cap = (16 * 256 + n_factors - 1) / n_factors;   // ~16 workgroups per CU over the whole round: a factor's workgroups loop beyond that
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  const dim3 grid(bx, n_factors);
  const SparseGeoDev* d = (const SparseGeoDev*)descs_dev;
  switch (cs) {
    case 16: hipLaunchKernelGGL(k_sparse_geometric_batch<16>, grid, dim3(kT), 0, stream, d); break;
    case 32: hipLaunchKernelGGL(k_sparse_geometric_batch<32>, grid, dim3(kT), 0, stream, d); break;
    case 64: hipLaunchKernelGGL(k_sparse_geometric_batch<64>, grid, dim3(kT), 0, stream, d); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---- SfM error: sum (w r)^2, inliers (dense_sfm.h:79-119: default border 1, min_dpt 0) -------------------------------------------------
__device__ __forceinline__ void sfm_error_body(const SimplePairDev& p, const int W, const int H, const float huber_delta, float* __restrict__ out_row) {
  __shared__ float red[kT / 64][kSimpleRow];
  float acc = 0.f;
  const unsigned inl = row_walk<false, DFX_TAP_DIST_ERR>(p, p.R, p.t, p.fg.e1, p.fg.e2, W, H, [&](const RowPix<false>& S) {
    float r = S.i0 - pix_img(S);
    r *= huber_weight(r, huber_delta);
    acc = __builtin_fmaf(r, r, acc);
  });
  const float s = wave_sum(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = s; red[wave][1] = (float)inl; }
  fold_waves_store(red, 2, out_row);
}

__global__ __launch_bounds__(kT) void k_sfm_error(const SimplePairDev p, const int W, const int H, const float huber_delta,
                                                  float* __restrict__ partials) {
  sfm_error_body(p, W, H, huber_delta, partials + (size_t)blockIdx.x * kSimpleRow);
}

// ---- batched forms (blockIdx.y = pair): PhotometricFactor::error over a factor set evaluates one pair per blocking call in the reference
// (core/gtsam/photometric_factor.cpp:61-81,197-216); a relocalisation / loop-closure check steps one live frame against many keyframes.
// Same per-pair arithmetic and reduction order as the single-pair kernels launched with the same number of workgroups.
__global__ __launch_bounds__(kT) void k_sfm_error_batch(const SimplePairDev* __restrict__ descs, const int W, const int H, const float huber_delta,
                                                        float* __restrict__ partials_all) {
  sfm_error_body(descs[blockIdx.y], W, H, huber_delta, partials_all + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kSimpleRow);
}

__global__ __launch_bounds__(kT) void k_se3_step_batch(const SimplePairDev* __restrict__ descs, const int W, const int H, const float huber_delta,
                                                       float* __restrict__ partials_all) {
  const SimplePairDev& p = descs[blockIdx.y];
  se3_step_body(p, p.R, p.t, p.fg.e1, p.fg.e2, W, H, huber_delta, partials_all + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kSimpleRow);
}

// ---- Warp: render img1 into frame 0, SIGNED residual sum (cu_se3aligner.cpp:106) ----------------------------
__global__ __launch_bounds__(kT) void k_se3_warp(const SimplePairDev p, const int W, const int H, float* __restrict__ partials) {
  const Geo g = geo_from(p);
  const ImgRef I0{ (const char*)p.img0, p.pitch_img0 }, I1{ (const char*)p.img1, p.pitch_img1 };
  const ImgRef D0{ (const char*)p.dpt0, p.pitch_dpt0 };
  float acc[2] = { 0.f, 0.f };
  const int npx = W * H;
  for (int i = blockIdx.x * kT + threadIdx.x; i < npx; i += gridDim.x * kT) {
    const int y = i / W, x = i - y * W;
    const float d = D0.at(x, y);
    const float i0 = I0.at(x, y);
    const Corr c = find_correspondence(g, x, y, d, 1.0f, 0.0f);   // `depth <= 0 -> skip`, PixelValid(pix1, 1)
    float outv = 0.f;
    if (c.valid) {
      const Taps tp = make_taps(c.u, c.v);
      outv = sample_img(I1, tp);
      acc[0] += i0 - outv;
      acc[1] += 1.0f;
    }
    gstore<float>((char*)p.img2 + (size_t)y * p.pitch_img2 + (size_t)x * 4, outv);
  }
  block_reduce_store<2>(acc, partials + (size_t)blockIdx.x * kSimpleRow);
}

// ---- SquaredError ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void k_squared_error(const float* __restrict__ a, const uint32_t pa, const float* __restrict__ b,
                                                      const uint32_t pb, const int W, const int H, float* __restrict__ partials) {
  float acc[1] = { 0.f };
  const int npx = W * H;
  for (int i = blockIdx.x * kT + threadIdx.x; i < npx; i += gridDim.x * kT) {
    const int y = i / W, x = i - y * W;
    const float d = reinterpret_cast<const float*>((const char*)a + (size_t)y * pa)[x] -
                    reinterpret_cast<const float*>((const char*)b + (size_t)y * pb)[x];
    acc[0] += d * d;
  }
  block_reduce_store<1>(acc, partials + (size_t)blockIdx.x * kSimpleRow);
}

// ---- finalize: out[e] = sum_b partials[b][e] in double, fixed order; layout-specific scatter -------------------
enum FinalKind { kFinalItem6 = 0, kFinalCorr = 1, kFinalScalar = 2 };

// MAXQ: rows per row group the launch can hold (nblocks <= 32 MAXQ).  Rows past nblocks add +0.0, so every MAXQ gives the same bits; the
// batched reductions run 24 - 60 workgroups per pair and take MAXQ = 2 (2 loads per thread instead of 32: 4.7 -> ~3 us per 128 pairs).
template <int MAXQ>
__global__ __launch_bounds__(1024) void k_finalize_rows(const float* __restrict__ partials_all, const int nblocks, const int kind,
                                                        char* __restrict__ out_all, const size_t out_stride, const DoneFlag done) {
  // blockIdx.x = pair of a batched launch (0 for the single-pair operators)
  const float* partials = partials_all + (size_t)blockIdx.x * nblocks * kSimpleRow;
  char* out = out_all + (size_t)blockIdx.x * out_stride;
  __shared__ double red[32][kSimpleRow];
  const int e = threadIdx.x & 31, rg = threadIdx.x >> 5;   // 32 row groups
  static_assert(kMaxSimpleBlocks <= 32 * 32, "one load per row group and thread");
  double s = strided_sum_f64_wide<32, MAXQ>(partials + e, rg, nblocks, kSimpleRow);
  red[rg][e] = s;
  __syncthreads();
  if (rg != 0) return;
  s = 0.0;
  for (int q = 0; q < 32; ++q) s += red[q][e];
  if (kind == kFinalItem6) {
    // JTJJrReductionItem<float,6>: 21 + 6 + 1 floats, then u64 inliers at byte 112
    if (e < 28) reinterpret_cast<float*>(out)[e] = (float)s;
    else if (e == 28) *reinterpret_cast<unsigned long long*>(out + 112) = (unsigned long long)(s + 0.5);
  } else if (kind == kFinalCorr) {
    if (e == 0) reinterpret_cast<float*>(out)[0] = (float)s;
    else if (e == 1) *reinterpret_cast<unsigned long long*>(out + 8) = (unsigned long long)(s + 0.5);
  } else {
    if (e == 0) reinterpret_cast<float*>(out)[0] = (float)s;
  }
  signal_done_wave(done);   // (single-pair launches: the result was stored by lanes 0..31 of this wave)
}

// ---- UpdateDepth: dpt = a / (prx0 + jac . code) - a  (the code-Jacobian decoder, GEMV, 8 + 4 CS bytes / pixel) ---
// LPP = CS/4 lanes share one pixel (float4 each -> every wave-load is 1 KiB contiguous); 64 pixels per wave step so
// that the prx read and the depth write are single coalesced 256-byte accesses.
template <int CS>
__device__ __forceinline__ void update_depth_body(const float* __restrict__ code, const float* __restrict__ prx,
                                                  const uint32_t pitch_prx, const float* __restrict__ jac,
                                                  const uint32_t pitch_jac, const float avg_dpt, float* __restrict__ out,
                                                  const uint32_t pitch_out, const int W, const int H) {
  constexpr int LPP = CS / 4;        // lanes per pixel
  constexpr int PPL = 64 / LPP;      // pixels per wave-load
  constexpr int NSUB = 64 / PPL;     // wave-loads per 64-pixel chunk (== LPP)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane % LPP, grp = lane / LPP;
  const f32x4 c4 = *reinterpret_cast<const f32x4*>(code + 4 * q);
  const int npx = W * H;
  const int nchunks = (npx + 63) >> 6;
  // Fast path (every image of the reference's pyramids): rows are whole chunks (W % 64 == 0), so a chunk lies in ONE image row and its
  // 64 * CS Jacobian floats are one contiguous run -- no per-load row arithmetic, no bounds branches, and the stream is read with the
  // non-temporal policy (read once; keeps the L2 for the 8 B/px of prx / depth traffic next to it).
  const bool rows_are_chunks = (W & 63) == 0;
  for (int chunk = blockIdx.x * (kT / 64) + wave; chunk < nchunks; chunk += gridDim.x * (kT / 64)) {
    const int base = chunk << 6;
    const int y0 = base / W, x0 = base - y0 * W;
    // sub-step s: lane group `grp` handles pixel base + grp*NSUB + s  -> after NSUB steps lane l owns pixel base + l
    f32x4 v[NSUB];
    if (rows_are_chunks) {
      const f32x4* row = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>((const char*)jac + (size_t)y0 * pitch_jac) + (size_t)x0 * CS) + q;
#pragma unroll
      for (int s = 0; s < NSUB; ++s) v[s] = __builtin_nontemporal_load(row + (grp * NSUB + s) * LPP);
    } else {
#pragma unroll
      for (int s = 0; s < NSUB; ++s) {
        const int off = grp * NSUB + s;
        int x = x0 + off, y = y0;
        while (x >= W) { x -= W; ++y; }
        f32x4 t = f32x4{ 0.f, 0.f, 0.f, 0.f };
        if (base + off < npx) t = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>((const char*)jac + (size_t)y * pitch_jac) + (size_t)x * CS + 4 * q);
        v[s] = t;
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      float d = v[s].x * c4.x + v[s].y * c4.y + v[s].z * c4.z + v[s].w * c4.w;
#pragma unroll
      for (int m = 1; m < LPP; m <<= 1) d += __shfl_xor(d, m, 64);
      if (q == s) mine = d;   // lane l = grp*LPP + q keeps pixel grp*NSUB + q = l (NSUB == LPP)
    }
    int x = x0 + lane, y = y0;
    if (!rows_are_chunks) while (x >= W) { x -= W; ++y; }
    if (rows_are_chunks || base + lane < npx) {
      const float p0 = reinterpret_cast<const float*>((const char*)prx + (size_t)y * pitch_prx)[x];
      const float pr = p0 + mine;
      reinterpret_cast<float*>((char*)out + (size_t)y * pitch_out)[x] = avg_dpt / pr - avg_dpt;   // (default policy: the step reads it next; nt here, or on prx, costs 2 - 4 %)
    }
  }
}

template <int CS>
__global__ __launch_bounds__(kT) void k_update_depth(const float* __restrict__ code, const float* __restrict__ prx,
                                                     const uint32_t pitch_prx, const float* __restrict__ jac,
                                                     const uint32_t pitch_jac, const float avg_dpt, float* __restrict__ out,
                                                     const uint32_t pitch_out, const int W, const int H) {
  update_depth_body<CS>(code, prx, pitch_prx, jac, pitch_jac, avg_dpt, out, pitch_out, W, H);
}

// n decode jobs of one image size in ONE launch (grid.y = job): Mapper::UpdateMap re-decodes every changed keyframe per level
// (core/mapping/mapper.cpp:860-888), dfx_sfm_linearize_batch decodes every distinct keyframe of a batch of pairs.
template <int CS>
__global__ __launch_bounds__(kT) void k_update_depth_batch(const DepthJobDev* __restrict__ jobs, const float avg_dpt, const int W, const int H) {
  const DepthJobDev& j = jobs[blockIdx.y];
  update_depth_body<CS>(j.code, j.prx, j.pitch_prx, j.jac, j.pitch_jac, avg_dpt, j.out, j.pitch_out, W, H);
}

// ---- Sobel / 8 with clamped borders (cu_image_proc.cpp:57-92) ---------------------------------------------------
__global__ __launch_bounds__(kT) void k_sobel(const float* __restrict__ img, const uint32_t pitch, float* __restrict__ grad,
                                              const uint32_t gpitch, const int W, const int H) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const int xm = max(x - 1, 0), xp = min(x + 1, W - 1), ym = max(y - 1, 0), yp = min(y + 1, H - 1);
  const float* r0 = reinterpret_cast<const float*>((const char*)img + (size_t)ym * pitch);
  const float* r1 = reinterpret_cast<const float*>((const char*)img + (size_t)y * pitch);
  const float* r2 = reinterpret_cast<const float*>((const char*)img + (size_t)yp * pitch);
  const float a = r0[xm], b = r0[x], c = r0[xp], d = r1[xm], f = r1[xp], g = r2[xm], h = r2[x], i = r2[xp];
  // same tap order as the reference loop (py outer, px inner), zero taps skipped
  float sx = 0.f, sy = 0.f;
  sx += a * -1.f; sy += a * -1.f;
  sy += b * -2.f;
  sx += c * 1.f;  sy += c * -1.f;
  sx += d * -2.f;
  sx += f * 2.f;
  sx += g * -1.f; sy += g * 1.f;
  sy += h * 2.f;
  sx += i * 1.f;  sy += i * 1.f;
  f32x2 o = { sx / 8.f, sy / 8.f };
  *reinterpret_cast<f32x2*>(reinterpret_cast<float*>((char*)grad + (size_t)y * gpitch) + 2 * x) = o;
}

// ---- 5x5 binomial blur + decimate (cu_image_proc.cpp:134-164) ----------------------------------------------------
__global__ __launch_bounds__(kT) void k_blur_down(const float* __restrict__ in, const uint32_t pitch, const int W, const int H,
                                                  float* __restrict__ out, const uint32_t opitch, const int OW, const int OH) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= OW || y >= OH) return;
  const float B[5] = { 1.f, 4.f, 6.f, 4.f, 1.f };
  float sum = 0.f, wall = 0.f;
#pragma unroll
  for (int py = 0; py < 5; ++py) {
    const int ny = min(max(2 * y + py - 2, 0), H - 1);
    const float* r = reinterpret_cast<const float*>((const char*)in + (size_t)ny * pitch);
#pragma unroll
    for (int px = 0; px < 5; ++px) {
      const int nx = min(max(2 * x + px - 2, 0), W - 1);
      const float k = B[px] * B[py];
      sum += r[nx] * k;
      wall += k;
    }
  }
  reinterpret_cast<float*>((char*)out + (size_t)y * opitch)[x] = sum / wall;
}

// ---- one pyramid level of n frames in ONE launch: Sobel gradient of level i AND the blur-down to level i + 1 from the same LDS tile ----------
// Frame::FillPyramids (core/mapping/frame.h:80-94) / UploadLiveFrame (core/deepfactors.cpp:616-630) build L images + L gradients per frame
// at camera rate with 2 L - 1 blocking single-image launches (cu_image_proc.cpp:94-112,166-186).  Here a level of ALL frames of a batch is one
// launch (grid.z = frame) that reads the level ONCE: a workgroup stages a (kPyrTW + 8) x (kPyrTH + 4) window of the image in LDS (float4 row
// loads on the interior; clamped dword loads only in tiles that touch the image border), its waves write the Sobel / 8 gradient row by row
// (lane = column) and every thread the 5 x 5 binomial blur-down of two pixels of the next level.  Level 0 of 640x480: 4 B read + 8 + 1 B
// written per pixel.  Same tap order and arithmetic as k_sobel / k_blur_down below (the per-level operators): the same bits.
constexpr int kPyrTW = 64, kPyrTH = 32;                 // input pixels per workgroup tile
constexpr int kPyrLW = kPyrTW + 8, kPyrLH = kPyrTH + 4; // staged window: columns x0 - 4 .. x0 + 67 (16-byte aligned), rows y0 - 2 .. y0 + 33
#ifndef DFX_PYR_NT
#define DFX_PYR_NT 0    // 1: non-temporal stores of the outputs (A/B)
#endif

template <typename T>
__device__ __forceinline__ void pyr_store(void* p, const T& v) {
  if (DFX_PYR_NT) __builtin_nontemporal_store(v, (DFX_GLOBAL T*)p); else gstore<T>(p, v);
}
// The FIRST launch of a build reads its descriptors out of the pinned staging slot (zero-copy) and its workgroup 0 leaves a copy of ALL levels' descriptors
// in device memory for the launches behind it: no host-to-device copy command in front of the build (a 4.4 us blit kernel + 6.3 us of boundary per 64-frame
// build in the round-6 trace, profiles/r06_pyramid.txt), and only the ~10^3 workgroups of one launch start with a read across the host link.
// The same workgroup first stores the build's sequence number into a host-visible word: a kernel of build q running means everything in front of it on the stream
// has completed, so the staging slots of the builds up to q - 1 are free again -- the host learns it without an event (a command-processor packet of ~3.5 us
// behind every build, profiles/r06_pyramid.txt).
__device__ __forceinline__ void pyr_mirror_descs(const PyrLevelDev* __restrict__ src, PyrLevelDev* __restrict__ mirror, const int count, const bool first_wg, const PyrStart st) {
  static_assert(sizeof(PyrLevelDev) % 8 == 0, "descriptors are copied as 8-byte words");
  if (!first_wg) return;
  if (st.word && threadIdx.x == 0) __hip_atomic_store(st.word, st.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (!mirror) return;
  const int words = count * (int)(sizeof(PyrLevelDev) / 8);
  const unsigned long long* s = reinterpret_cast<const unsigned long long*>(src);
  unsigned long long* d = reinterpret_cast<unsigned long long*>(mirror);
  constexpr int NB = 8;   // every load of a batch is in flight before the first store: the source is across the host link (~2 us per round trip)
  for (int base = 0; base < words; base += NB * (int)blockDim.x) {
    unsigned long long v[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) { const int i = base + j * (int)blockDim.x + (int)threadIdx.x; v[j] = i < words ? s[i] : 0ull; }
#pragma unroll
    for (int j = 0; j < NB; ++j) { const int i = base + j * (int)blockDim.x + (int)threadIdx.x; if (i < words) d[i] = v[j]; }
  }
}
__global__ __launch_bounds__(kT) void k_pyr_level(const PyrLevelDev* __restrict__ descs, PyrLevelDev* __restrict__ mirror, const int mirror_count, const PyrStart st) {
  pyr_mirror_descs(descs, mirror, mirror_count, (blockIdx.x | blockIdx.y | blockIdx.z) == 0, st);
  const PyrLevelDev& P = descs[blockIdx.z];
  const int W = P.W, H = P.H, OW = P.OW, OH = P.OH;
  const int x0 = blockIdx.x * kPyrTW, y0 = blockIdx.y * kPyrTH;
  // the window twice: as it is (the Sobel taps: lane = column, unit stride) and split into its even and odd columns (the blur taps of neighbouring
  // output pixels are TWO columns apart: in the plain window half of the LDS banks would serve them, 2-4 lanes each)
  __shared__ float tile[kPyrLH][kPyrLW];
  __shared__ float tev[kPyrLH][kPyrLW / 2 + 4], tod[kPyrLH][kPyrLW / 2 + 4];
  const bool interior = x0 >= 4 && x0 + kPyrTW + 4 <= W && y0 >= 2 && y0 + kPyrTH + 2 <= H && ((P.pitch_in | (uint32_t)(uintptr_t)P.in) & 15) == 0;
  const bool blur = P.out != nullptr;
  if (interior) {
    for (int e = threadIdx.x; e < kPyrLH * (kPyrLW / 4); e += kT) {
      const int r = e / (kPyrLW / 4), c4 = e - r * (kPyrLW / 4);
      const f32x4 v = gload<f32x4>((const char*)P.in + (size_t)(y0 - 2 + r) * P.pitch_in + (size_t)(x0 - 4 + 4 * c4) * 4);
      *reinterpret_cast<f32x4*>(&tile[r][4 * c4]) = v;
      if (blur) {
        *reinterpret_cast<f32x2*>(&tev[r][2 * c4]) = f32x2{ v.x, v.z };
        *reinterpret_cast<f32x2*>(&tod[r][2 * c4]) = f32x2{ v.y, v.w };
      }
    }
  } else {
    for (int e = threadIdx.x; e < kPyrLH * kPyrLW; e += kT) {
      const int r = e / kPyrLW, c = e - r * kPyrLW;
      const int y = min(max(y0 - 2 + r, 0), H - 1), x = min(max(x0 - 4 + c, 0), W - 1);   // getWithClampedRange (cu_image_proc.cpp:66-70,141-146)
      const float v = gload<float>((const char*)P.in + (size_t)y * P.pitch_in + (size_t)x * 4);
      tile[r][c] = v;
      if (c & 1) tod[r][c >> 1] = v; else tev[r][c >> 1] = v;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // ---- Sobel / 8 (cu_image_proc.cpp:57-92): lane = column, a wave walks 8 rows with the 3 x 3 window sliding through registers (three LDS reads per
  // pixel, conflict-free; one 512-byte store per wave and row)
  if (P.grad && x0 + lane < W) {
    const int lx = lane + 4;
    int ly = wave * (kPyrTH / 4) + 2;                       // window row of the wave's first pixel row
    float a = tile[ly - 1][lx - 1], b = tile[ly - 1][lx], c = tile[ly - 1][lx + 1];
    float d = tile[ly][lx - 1], m = tile[ly][lx], f = tile[ly][lx + 1];
#pragma unroll
    for (int k = 0; k < kPyrTH / 4; ++k, ++ly) {
      const float g = tile[ly + 1][lx - 1], h = tile[ly + 1][lx], i = tile[ly + 1][lx + 1];
      float sx = 0.f, sy = 0.f;   // the tap order of k_sobel (the reference loop, zero taps skipped)
      sx += a * -1.f; sy += a * -1.f;
      sy += b * -2.f;
      sx += c * 1.f;  sy += c * -1.f;
      sx += d * -2.f;
      sx += f * 2.f;
      sx += g * -1.f; sy += g * 1.f;
      sy += h * 2.f;
      sx += i * 1.f;  sy += i * 1.f;
      const int y = y0 + ly - 2;
      if (y < H) pyr_store<f32x2>((char*)P.grad + (size_t)y * P.pitch_grad + (size_t)(x0 + lane) * 8, f32x2{ sx / 8.f, sy / 8.f });
      a = d; b = m; c = f; d = g; m = h; f = i;
    }
  }
  // ---- 5 x 5 binomial blur + decimate (cu_image_proc.cpp:134-164): thread = two VERTICALLY adjacent pixels of the next level (they share three of their
  // five window rows: 35 LDS reads for two pixels instead of 50), lane = output column (unit stride in the even / odd column arrays).  Window column of input
  // x0 - 4 + c is c; output column ox reads inputs 2 ox - 2 .. 2 ox + 2 = window columns 2 ox + 2 .. 2 ox + 6: even ones at tev[.][ox + 1 .. ox + 3], odd
  // ones at tod[.][ox + 1 .. ox + 2].  Each pixel is summed in the order of k_blur_down (py outer, px inner): the same bits.
  if (blur) {
    const float B[5] = { 1.f, 4.f, 6.f, 4.f, 1.f };
    const int ox = threadIdx.x & 31, oyp = threadIdx.x >> 5;      // 32 columns x 8 row pairs
    const int X = (x0 >> 1) + ox;
    float w[7][5];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const int wr = 4 * oyp + r;                                  // window row of input row 2 (2 oyp) - 2 + r
      w[r][0] = tev[wr][ox + 1]; w[r][1] = tod[wr][ox + 1]; w[r][2] = tev[wr][ox + 2]; w[r][3] = tod[wr][ox + 2]; w[r][4] = tev[wr][ox + 3];
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int Y = (y0 >> 1) + 2 * oyp + half;
      if (X < OW && Y < OH) {
        float sum = 0.f, wall = 0.f;
#pragma unroll
        for (int py = 0; py < 5; ++py) {
#pragma unroll
          for (int px = 0; px < 5; ++px) {
            const float k = B[px] * B[py];
            sum += w[2 * half + py][px] * k;
            wall += k;
          }
        }
        pyr_store<float>((char*)P.out + (size_t)Y * P.pitch_out + (size_t)X * 4, sum / wall);
      }
    }
  }
}

// ---- the same level, ROW-STREAMING (round 6): no LDS, no barrier, no per-pixel address arithmetic ---------------------------------------------------
// A WAVE owns a strip of 128 columns (two pixels per lane: one 8-byte load, one 16-byte gradient store per lane and row -> 512-byte reads and
// 1-KB write runs per wave and row) and a segment of R rows, and walks down the rows with everything it needs in registers:
//   * the horizontal neighbours of a lane's two pixels (columns c0 - 2, c0 - 1, c1 + 1) come from the adjacent lanes by DPP wavefront shifts
//     (v_mov_b32_dpp wave_shr:1 / wave_shl:1); lanes 0 and 63 get theirs from ONE extra 8-byte load per row in which only those two lanes carry
//     an in-range offset; the clamped columns of getWithClampedRange (cu_image_proc.cpp:66-70,141-146) are selects;
//   * Sobel of row y needs rows y - 1, y, y + 1: a three-row window rotates through registers;
//   * the 5 x 5 blur-down is accumulated AS THE ROWS ARRIVE, in the tap order of k_blur_down (py outer, px inner): input row v carries the taps
//     py = v - 2 Y + 2 of the output rows Y it touches -- three accumulators in flight on even rows (py = 4, 2, 0), two on odd ones (py = 3, 1) -- so
//     every output pixel is the same chain of 25 fused multiply-adds as in the per-level operator: the same bits;
//   * addressing is a buffer resource of ONE ROW (loads past the row end return 0 and stores are dropped: no column predicate), re-based from row to
//     row on the SCALAR unit (s_add_u32 / s_addc_u32 on the descriptor's base; a scalar row OFFSET would not do: gfx950 range-checks vector + scalar
//     offset against the row length): no vector-ALU address arithmetic at all;
//   * the next two rows are requested before the current two are worked on.
// Work items are (frame, row segment, strip); a workgroup is four consecutive strips / segments.  Requires an even width and 8 / 16-byte aligned
// image / gradient rows (the host checks; anything else takes k_pyr_level above).
__device__ __forceinline__ float dpp_wave_shr1(float old, float src) {   // lane i <- src of lane i - 1; lane 0 keeps `old`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_wave_shl1(float old, float src) {   // lane i <- src of lane i + 1; lane 63 keeps `old`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x130, 0xf, 0xf, false));
}
struct PyrWin { float L1, A, B, R1; };   // columns c0 - 1, c0, c1, c1 + 1 of one row
__device__ __forceinline__ f32x2 pyr_sobel(const float a, const float b, const float c, const float d, const float f, const float g, const float h, const float i) {
  float sx = 0.f, sy = 0.f;   // the tap order of k_sobel (the reference loop, zero taps skipped)
  sx += a * -1.f; sy += a * -1.f;
  sy += b * -2.f;
  sx += c * 1.f;  sy += c * -1.f;
  sx += d * -2.f;
  sx += f * 2.f;
  sx += g * -1.f; sy += g * 1.f;
  sy += h * 2.f;
  sx += i * 1.f;  sy += i * 1.f;
  return f32x2{ sx / 8.f, sy / 8.f };
}
// one row of taps.  EXPLICIT fused multiply-adds: k_blur_down's `sum += r[nx] * k` contracts to v_fma_f32 (each product has one use); here a pixel
// feeds up to three accumulators, equal products (k = 6 for py = 0 and py = 4, ...) would be shared as ONE rounded v_mul_f32 + two adds: one ulp off
template <int PY>
__device__ __forceinline__ void pyr_blur_taps(float& sum, const float L2, const PyrWin& r) {
  const float B[5] = { 1.f, 4.f, 6.f, 4.f, 1.f };
  sum = __builtin_fmaf(L2, B[0] * B[PY], sum);
  sum = __builtin_fmaf(r.L1, B[1] * B[PY], sum);
  sum = __builtin_fmaf(r.A, B[2] * B[PY], sum);
  sum = __builtin_fmaf(r.B, B[3] * B[PY], sum);
  sum = __builtin_fmaf(r.R1, B[4] * B[PY], sum);
}
constexpr int kPyrStrip = 128;
#ifndef DFX_PYR_LD_AUX
#define DFX_PYR_LD_AUX 0   // cache policy of k_pyr_rows' image loads (bit 0 = sc0, bit 1 = nt)
#endif
#ifndef DFX_PYR_ST_AUX
#define DFX_PYR_ST_AUX 0   // ... of its next-level image stores (re-read by the next launch of the build)
#endif
#ifndef DFX_PYR_GRAD_AUX
#define DFX_PYR_GRAD_AUX 2 // ... and of its gradient stores (nt: written once, not read by the build)
#endif
template <int NP>
__device__ __forceinline__ void pyr_rows_body(const PyrLevelDev& P, const int nstrips, const int nsegs, const int R);
template <int NP>
__global__ __launch_bounds__(512) void k_pyr_rows(const PyrLevelDev* __restrict__ descs, const int nstrips, const int nsegs, const int R,
                                                  PyrLevelDev* __restrict__ mirror, const int mirror_count, const PyrStart st) {
  // (see k_pyr_level.  Measured and dropped: an extra workgroup column for the bookkeeping -- 64 more workgroups shift the whole grid's placement, level 0 45.8 -> 47.5 us;
  // the level's descriptors by value in the kernel arguments for <= 64 frames -- no change: what level 0 gained over the 43.9 us it took behind the copy command is the
  // tail of the previous build's write-backs, which the copy used to absorb.)
  pyr_mirror_descs(descs, mirror, mirror_count, (blockIdx.x | blockIdx.y) == 0, st);   // (the short last row segment instead of workgroup 0: no difference)
  pyr_rows_body<NP>(descs[blockIdx.y], nstrips, nsegs, R);
}
template <int NP>
__device__ __forceinline__ void pyr_rows_body(const PyrLevelDev& P, const int nstrips, const int nsegs, const int R) {
  const int lane = threadIdx.x & 63;
  // a workgroup = the `wpg` strips side by side of one row segment (wpg = blockDim.x / 64 divides into nstrips groups): its waves walk down the same rows, so
  // the workgroup writes whole image rows (5 KB at 640 pixels) rather than 1-KB pieces of four different places
  const int wpg = (int)(blockDim.x >> 6), gps = (nstrips + wpg - 1) / wpg;   // strip groups per segment
  const int seg = (int)blockIdx.x / gps, strip = ((int)blockIdx.x - seg * gps) * wpg + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (strip >= nstrips || seg >= nsegs) return;
  const int W = P.W, H = P.H, OH = P.OH;
  const int xs = strip * kPyrStrip, ys = seg * R;
  const int c0 = xs + 2 * lane;
  const bool has_left = xs >= 2, blur = P.out != nullptr, grad = P.grad != nullptr;
  // one ROW per resource
  auto row_rsrc = [](const void* base, int row, uint32_t pitch, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(base) + (size_t)row * pitch), 0, bytes, 0x00020000);
  };
  const int voff = c0 * 4;                                                                  // own two pixels (past the row end: zeros, no access)
  const int eoff = lane == 0 ? (has_left ? (xs - 2) * 4 : -1) : (lane == 63 ? (xs + kPyrStrip) * 4 : -1);   // lanes 0 / 63: the neighbouring strip's pixels
  const bool right_in = c0 + 2 < W;                                                         // column c1 + 1 exists (else it clamps to c1)
  auto fetch = [&](int v, f32x2& own, f32x2& edge) {
    const __amdgpu_buffer_rsrc_t rin = row_rsrc(P.in, min(max(v, 0), H - 1), P.pitch_in, W * 4);
    own = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rin, voff, 0, DFX_PYR_LD_AUX));
    edge = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rin, eoff, 0, DFX_PYR_LD_AUX));
  };
  auto store_grad = [&](int y, const f32x2 g0, const f32x2 g1) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{ g0.x, g0.y, g1.x, g1.y }), row_rsrc(P.grad, y, P.pitch_grad, W * 8), c0 * 8, 0, DFX_PYR_GRAD_AUX);
  };
  auto window = [&](const f32x2 own, const f32x2 edge, float& L2, PyrWin& w) {
    w.A = own.x; w.B = own.y;
    w.L1 = dpp_wave_shr1(has_left ? edge.y : own.x, own.y);
    L2 = dpp_wave_shr1(has_left ? edge.x : own.x, own.x);
    const float r = dpp_wave_shl1(edge.x, own.x);
    w.R1 = right_in ? r : own.y;
  };
  const int y_end = min(ys + R, H);                 // Sobel rows [ys, y_end)
  const int Y0 = ys >> 1, Y1 = min((ys + R) >> 1, OH);   // blur rows [Y0, Y1)
  const int v_last = min(ys + R, H + 1);            // last virtual row anything of this segment reads
  PyrWin rm{}, r0{};
  float accA = 0.f, accB = 0.f, accC = 0.f;
  auto do_pair = [&](const int v, const f32x2 own0, const f32x2 edge0, const f32x2 own1, const f32x2 edge1) {   // rows v (even) and v + 1
    {
      float L2; PyrWin cur;
      window(own0, edge0, L2, cur);
      const int y = v - 1;
      if (grad && y >= ys && y < y_end) {
        const f32x2 g0 = pyr_sobel(rm.L1, rm.A, rm.B, r0.L1, r0.B, cur.L1, cur.A, cur.B);
        const f32x2 g1 = pyr_sobel(rm.A, rm.B, rm.R1, r0.A, r0.R1, cur.A, cur.B, cur.R1);
        store_grad(y, g0, g1);
      }
      if (blur) {
        pyr_blur_taps<4>(accA, L2, cur);            // output row v / 2 - 1 is complete
        const int Y = (v >> 1) - 1;
        if (Y >= Y0 && Y < Y1) {
          float wall = 0.f;
          const float B[5] = { 1.f, 4.f, 6.f, 4.f, 1.f };
#pragma unroll
          for (int py = 0; py < 5; ++py)
#pragma unroll
            for (int px = 0; px < 5; ++px) wall += B[px] * B[py];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, accA / wall), row_rsrc(P.out, Y, P.pitch_out, P.OW * 4), (xs / 2 + lane) * 4, 0, DFX_PYR_ST_AUX);
        }
        pyr_blur_taps<2>(accB, L2, cur);
        accC = 0.f;
        pyr_blur_taps<0>(accC, L2, cur);
      }
      rm = r0; r0 = cur;
    }
    if (v + 1 <= v_last) {
      float L2; PyrWin cur;
      window(own1, edge1, L2, cur);
      const int y = v;
      if (grad && y >= ys && y < y_end) {
        const f32x2 g0 = pyr_sobel(rm.L1, rm.A, rm.B, r0.L1, r0.B, cur.L1, cur.A, cur.B);
        const f32x2 g1 = pyr_sobel(rm.A, rm.B, rm.R1, r0.A, r0.R1, cur.A, cur.B, cur.R1);
        store_grad(y, g0, g1);
      }
      if (blur) {
        pyr_blur_taps<3>(accB, L2, cur);
        pyr_blur_taps<1>(accC, L2, cur);
      }
      rm = r0; r0 = cur;
    }
    accA = accB; accB = accC;
  };
  // NP row pairs per iteration, and the NP pairs of the NEXT iteration requested before the current ones are worked on: 2 NP rows (1 KB each incl. the
  // edge load) in flight per wave -- the loop is bound by memory latency x requests in flight, not by its ~60 vector-ALU instructions per row
  f32x2 own[2 * NP], edge[2 * NP];
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) fetch(ys - 2 + j, own[j], edge[j]);
  for (int v = ys - 2; v <= v_last; v += 2 * NP) {
    f32x2 nown[2 * NP], nedge[2 * NP];
#pragma unroll
    for (int j = 0; j < 2 * NP; ++j) fetch(v + 2 * NP + j, nown[j], nedge[j]);   // (clamped rows past the end are re-reads of the last row: harmless)
#pragma unroll
    for (int q = 0; q < NP; ++q)
      if (v + 2 * q <= v_last) do_pair(v + 2 * q, own[2 * q], edge[2 * q], own[2 * q + 1], edge[2 * q + 1]);
#pragma unroll
    for (int j = 0; j < 2 * NP; ++j) { own[j] = nown[j]; edge[j] = nedge[j]; }
  }
}

// ---- the SMALL levels of a build in ONE launch (round 6): a few workgroups per frame keep their bands of the levels in LDS ----------------------------------------
// From the first level k0 >= 1 from which everything below is small (160 x 120 and 80 x 60 of a 640 x 480 input), the chain of launches -- each a few microseconds
// of latency-bound work plus a launch boundary -- becomes one.  A frame is cut into `nb` horizontal BANDS; a workgroup owns one band of every level (band rows of
// level l = twice those of level l + 1) and holds, per level, the rows it needs in LDS: its own rows +- 1 for the Sobel window, and whatever the blur-down into the
// rows it holds of the NEXT level reads -- so the halo rows of a level are RECOMPUTED by the neighbouring bands from the level above (identical arithmetic on identical
// inputs: identical bits), and nothing is exchanged between workgroups.  Per level: Sobel / 8 of the own rows from LDS to global memory; blur-down of the held rows of the
// next level into LDS, the own ones also to global memory; barrier.  Taps, order and arithmetic are k_sobel's / k_blur_down's (explicit fmaf chain, py outer, px
// inner): the same bits.  Any width, pitch and alignment.
constexpr int kPyrTailThreads = 512;
constexpr int kPyrTailMaxLevels = 4;   // instantiated for 2, 3, 4 levels (the row tables must unroll into registers: with a run-time level count they live in scratch memory)
struct PyrTailRows { int own_lo[kPyrTailMaxLevels], own_hi[kPyrTailMaxLevels], hold_lo[kPyrTailMaxLevels], hold_hi[kPyrTailMaxLevels]; };
// rows of levels k0 .. L-1 (index l - k0) that band `band` of `nb` owns and holds; Hs[l - k0] = height of level l; rows_per = band height at the LAST level
template <int NL>
__host__ __device__ inline void pyr_tail_rows(const int* Hs, int band, int nb, int rows_per, PyrTailRows& r) {
  constexpr int nl = NL, last = nl - 1;
  r.own_lo[last] = band * rows_per;
  r.own_hi[last] = (band == nb - 1) ? Hs[last] : (band + 1) * rows_per;
#pragma unroll
  for (int l = last - 1; l >= 0; --l) {
    r.own_lo[l] = 2 * r.own_lo[l + 1];
    r.own_hi[l] = (band == nb - 1) ? Hs[l] : 2 * r.own_hi[l + 1];
  }
#pragma unroll
  for (int l = last; l >= 0; --l) {
    int lo = r.own_lo[l] - 1, hi = r.own_hi[l] + 1;                       // the Sobel window of the own rows
    if (l < last) {                                                       // the 5 x 5 windows of the rows held of the next level
      const int blo = 2 * r.hold_lo[l + 1] - 2, bhi = 2 * (r.hold_hi[l + 1] - 1) + 3;
      lo = blo < lo ? blo : lo; hi = bhi > hi ? bhi : hi;
    }
    r.hold_lo[l] = lo < 0 ? 0 : lo;
    r.hold_hi[l] = hi > Hs[l] ? Hs[l] : hi;
  }
}
template <int NL>
__global__ __launch_bounds__(1024) void k_pyr_tail(const PyrLevelDev* __restrict__ descs, const int n, const int k0, const int nb, const int rows_per) {
  extern __shared__ float pyr_lds[];
  constexpr int nl = NL;
  const int f = blockIdx.y, band = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  int Hs[NL];
#pragma unroll
  for (int l = 0; l < nl; ++l) Hs[l] = descs[(size_t)(k0 + l) * n + f].H;
  PyrTailRows R;
  pyr_tail_rows<NL>(Hs, band, nb, rows_per, R);
  float* cur = pyr_lds;
  // p -> (row, column) of a W-wide block without an integer division: (p + 0.5) / W is at least 0.5 / W away from an integer, float rounding is 1e-7 of it
  auto split = [](int p, float invW, int W, int& r, int& x) { r = (int)(((float)p + 0.5f) * invW); x = p - r * W; };
  {
    const PyrLevelDev& P = descs[(size_t)k0 * n + f];
    const int W = P.W, lo = R.hold_lo[0], npx = (R.hold_hi[0] - lo) * W;
    const float invW = 1.0f / (float)W;
    constexpr int NB = 8;     // every load of a batch is issued before the first LDS store (a plain loop is a chain of dependent global round trips)
    for (int base = 0; base < npx; base += NB * nthr) {
      float v[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int p = base + j * nthr + tid;
        int r, x;
        split(p < npx ? p : 0, invW, W, r, x);
        v[j] = gload<float>((const char*)P.in + (size_t)(lo + r) * P.pitch_in + (size_t)x * 4);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int p = base + j * nthr + tid;
        if (p < npx) cur[p] = v[j];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int l = 0; l < nl; ++l) {
    const PyrLevelDev& P = descs[(size_t)(k0 + l) * n + f];
    const int W = P.W, H = P.H, lo = R.hold_lo[l];
    const float invW = 1.0f / (float)W;
    float* nxt = cur + (R.hold_hi[l] - lo) * W;
    if (P.grad) {
      const int y0 = R.own_lo[l], npx = (R.own_hi[l] - y0) * W;
      for (int p = tid; p < npx; p += nthr) {
        int r, x;
        split(p, invW, W, r, x);
        const int y = y0 + r;
        const int xm = max(x - 1, 0), xp = min(x + 1, W - 1);
        const float* r0 = cur + (max(y - 1, 0) - lo) * W;
        const float* r1 = cur + (y - lo) * W;
        const float* r2 = cur + (min(y + 1, H - 1) - lo) * W;
        const f32x2 g = pyr_sobel(r0[xm], r0[x], r0[xp], r1[xm], r1[xp], r2[xm], r2[x], r2[xp]);
        __builtin_nontemporal_store(g, (DFX_GLOBAL f32x2*)(void*)((char*)P.grad + (size_t)y * P.pitch_grad + (size_t)x * 8));   // written once, not read by the build (see k_pyr_rows)
      }
    }
    if (P.out && l + 1 < nl) {
      const int OW = P.OW, Y0 = R.hold_lo[l + 1], npx = (R.hold_hi[l + 1] - Y0) * OW;
      const int oY0 = R.own_lo[l + 1], oY1 = R.own_hi[l + 1];
      const float invOW = 1.0f / (float)OW;
      const float B[5] = { 1.f, 4.f, 6.f, 4.f, 1.f };
      for (int p = tid; p < npx; p += nthr) {
        int r, X;
        split(p, invOW, OW, r, X);
        const int Y = Y0 + r;
        float sum = 0.f, wall = 0.f;
#pragma unroll
        for (int py = 0; py < 5; ++py) {
          const float* row = cur + (min(max(2 * Y + py - 2, 0), H - 1) - lo) * W;
#pragma unroll
          for (int px = 0; px < 5; ++px) {
            const float k = B[px] * B[py];
            sum = __builtin_fmaf(row[min(max(2 * X + px - 2, 0), W - 1)], k, sum);
            wall += k;
          }
        }
        const float v = sum / wall;
        if (Y >= oY0 && Y < oY1) gstore<float>((char*)P.out + (size_t)Y * P.pitch_out + (size_t)X * 4, v);
        nxt[p] = v;
      }
    }
    __syncthreads();
    cur = nxt;
  }
}
// bands per frame and the LDS a workgroup needs; 0 bands = the levels do not qualify
int pyr_tail_plan(const int* Hs, const int* Ws, int nl, int n, int* rows_per, size_t* lds_bytes) {
  if (nl < 2 || nl > kPyrTailMaxLevels) return 0;
  auto rows = [&](int b, int nbb, int rp, PyrTailRows& r) { if (nl == 2) pyr_tail_rows<2>(Hs, b, nbb, rp, r); else if (nl == 3) pyr_tail_rows<3>(Hs, b, nbb, rp, r); else pyr_tail_rows<4>(Hs, b, nbb, rp, r); };
  int nb = 512 / (n > 0 ? n : 1);                     // ~two workgroups per CU when there are that many frames; up to 8 bands for a single frame
  if (nb > 8) nb = 8;
  if (nb < 1) nb = 1;
  const int Hl = Hs[nl - 1];
  if (nb > Hl / 2) nb = Hl / 2 > 0 ? Hl / 2 : 1;      // at least two rows of the last level per band
  const int rp = (Hl + nb - 1) / nb;
  nb = (Hl + rp - 1) / rp;                            // no empty band
  size_t worst = 0;
  for (int b = 0; b < nb; ++b) {
    PyrTailRows r;
    rows(b, nb, rp, r);
    size_t px = 0;
    for (int l = 0; l < nl; ++l) px += (size_t)(r.hold_hi[l] - r.hold_lo[l]) * Ws[l];
    worst = px > worst ? px : worst;
  }
  *rows_per = rp;
  *lds_bytes = 4 * worst;
  return worst * 4 <= kPyrTailMaxLds ? nb : 0;
}
hipError_t launch_pyr_tail(const PyrLevelDev* descs_dev, int n, int k0, int L, int nb, int rows_per, size_t lds_bytes, hipStream_t stream) {
  static bool attr_set = false;   // (one process-wide function attribute: the largest request this kernel may get)
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyr_tail<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPyrTailMaxLds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyr_tail<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPyrTailMaxLds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyr_tail<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPyrTailMaxLds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  // 2 - 16 bands x 256 - 1024 threads all measure 9 - 10 us for the 64-frame tail (profiles/r06_pyramid.txt): ~4 us of launch and a chain of short phases, not the shape
  const dim3 grid(nb, n), block(kPyrTailThreads);
  switch (L - k0) {
    case 2: hipLaunchKernelGGL(k_pyr_tail<2>, grid, block, lds_bytes, stream, descs_dev, n, k0, nb, rows_per); break;
    case 3: hipLaunchKernelGGL(k_pyr_tail<3>, grid, block, lds_bytes, stream, descs_dev, n, k0, nb, rows_per); break;
    case 4: hipLaunchKernelGGL(k_pyr_tail<4>, grid, block, lds_bytes, stream, descs_dev, n, k0, nb, rows_per); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// Rows per segment of the row-streaming launch (`wpg` waves per workgroup, `gps` workgroups side by side per segment, n frames, `cus` compute units).
// Every workgroup of such a launch lives about as long as the launch, so what matters first is that the compute units hold EQUAL numbers of them: level 0 of a
// 64-frame 640x480 build with 13 segments per frame (832 workgroups = 3.25 per CU: the old "~4096 waves" rule) takes 46.2 us, with 12 segments (3 per CU) 41.1,
// with 8 (2 per CU, 10 waves) 40.4, with 16 (4 per CU) 42.2, with 4 (1 per CU, 5 waves) 41.5; level 1 (three-wave workgroups) with 20 segments (5 per CU) 12.6, with
// 12 (3 per CU, 9 waves) 12.4, with 24 (6 per CU) 13.1, with 8 (2 per CU, 6 waves) 13.7, with 10 (2.5 per CU) 13.5 -- same-box A/B, profiles/r06_pyramid.txt.
// Rule: among the even segment heights whose busiest CU holds 8-16 waves, the one with the best product of (CU balance: workgroups / (workgroups of the busiest CU x
// CUs)) x (segment balance: rows / (segments x segment height) -- a short last segment idles its share) x (1 - 1.5 / R: the 4 halo rows a segment re-reads, at the weight
// the sweeps give them); exact tilings (64 frames: 8 segments of 60 rows, 2 workgroups per CU) score highest, frame counts that do not divide the CUs still get within a few
// per cent of equal shares (60 frames: 480 workgroups on 256 CUs instead of 840).  Below a score of 0.8 -- few frames: fewer workgroups than CUs, or only tiny segments
// balance -- ~4096 waves per launch as before.  At least 4 rows per segment (a segment re-reads 3 rows of its neighbours).
int pyr_rows_per_segment(int H, int nstrips, int wpg, int gps, int n, int cus) {
  if (cus > 0) {
    int best = 0;
    double best_score = 0.8;
    for (int R = 4; R <= ((H + 1) & ~1); R += 2) {
      const int nsegs = (H + R - 1) / R;
      const long long wgs = (long long)n * gps * nsegs;
      if (wgs < cus) break;                                   // (fewer workgroups than CUs from here on)
      const long long busiest = (wgs + cus - 1) / cus;
      const long long waves = busiest * wpg;
      if (waves < 8 || waves > 16) continue;
      const double score = ((double)wgs / (double)(busiest * cus)) * ((double)H / ((double)nsegs * R)) * (1.0 - 1.5 / R);
      if (score > best_score) { best_score = score; best = R; }
    }
    if (best) return best;
  }
  const long long total = (long long)H * nstrips * n;
  int R = (int)((total + 4095) / 4096);
  R = (R + 1) & ~1;
  if (R < 4) R = 4;
  if (R > 64) R = 64;
  return R;
}

void pyr_rows_shape(int W, int H, int n, int cus, int* wpg_out, int* gps_out, int* R_out) {
  const int nstrips = (W + kPyrStrip - 1) / kPyrStrip;
  // a workgroup = the strips side by side of one row segment (at most 8 waves): whole image rows per workgroup (51.6 against 55.2 us with four-wave groups)
  int wpg = nstrips;
  if (wpg > 8) wpg = (nstrips + ((nstrips + 7) / 8) - 1) / ((nstrips + 7) / 8);
  const int gps = (nstrips + wpg - 1) / wpg;
  *wpg_out = wpg; *gps_out = gps;
  *R_out = pyr_rows_per_segment(H, nstrips, wpg, gps, n, cus);
}

hipError_t launch_pyr_level(const PyrLevelDev* descs_dev, int n, int W, int H, hipStream_t stream, bool rows_ok, PyrLevelDev* mirror, int mirror_count, PyrStart st, int cus) {
  if (rows_ok && (W & 1) == 0) {
    const int nstrips = (W + kPyrStrip - 1) / kPyrStrip;
    int wpg, gps, R;
    pyr_rows_shape(W, H, n, cus, &wpg, &gps, &R);
    const int nsegs = (H + R - 1) / R;
    // (rows in flight per wave -- the template argument -- 2 / 4 / 8: no gain at any launch shape, profiles/r06_pyramid.txt)
    hipLaunchKernelGGL(k_pyr_rows<1>, dim3(gps * nsegs, n), dim3(64 * wpg), 0, stream, descs_dev, nstrips, nsegs, R, mirror, mirror_count, st);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_pyr_level, dim3((W + kPyrTW - 1) / kPyrTW, (H + kPyrTH - 1) / kPyrTH, n), dim3(kT), 0, stream, descs_dev, mirror, mirror_count, st);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------
static void launch_finalize_rows(int n, int blocks, int kind, const float* partials_dev, void* out_dev, size_t stride, hipStream_t stream,
                                 const DoneFlag& done_in = DoneFlag{}) {
  const DoneFlag done = n == 1 ? done_in : DoneFlag{};   // one workgroup per pair: only a single-pair launch has ONE last writer
  if (blocks <= 64) hipLaunchKernelGGL(k_finalize_rows<2>, dim3(n), dim3(1024), 0, stream, partials_dev, blocks, kind, (char*)out_dev, stride, done);
  else if (blocks <= 256) hipLaunchKernelGGL(k_finalize_rows<8>, dim3(n), dim3(1024), 0, stream, partials_dev, blocks, kind, (char*)out_dev, stride, done);
  else hipLaunchKernelGGL(k_finalize_rows<32>, dim3(n), dim3(1024), 0, stream, partials_dev, blocks, kind, (char*)out_dev, stride, done);
}

hipError_t launch_se3_step(const SimplePairDev& p, int W, int H, float huber_delta, int blocks, float* partials_dev,
                           void* item_dev, hipStream_t stream, const DoneFlag& done) {
  hipLaunchKernelGGL(k_se3_step, dim3(blocks), dim3(kT), 0, stream, p, W, H, huber_delta, partials_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  launch_finalize_rows(1, blocks, (int)kFinalItem6, (const float*)partials_dev, item_dev, (size_t)0, stream, done);
  return hipGetLastError();
}

hipError_t launch_sfm_error(const SimplePairDev& p, int W, int H, float huber_delta, int blocks, float* partials_dev,
                            void* corr_item_dev, hipStream_t stream, const DoneFlag& done) {
  hipLaunchKernelGGL(k_sfm_error, dim3(blocks), dim3(kT), 0, stream, p, W, H, huber_delta, partials_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  launch_finalize_rows(1, blocks, (int)kFinalCorr, (const float*)partials_dev, corr_item_dev, (size_t)0, stream, done);
  return hipGetLastError();
}

hipError_t launch_sfm_error_batch(const SimplePairDev* descs_dev, int n, int W, int H, float huber_delta, int blocks, float* partials_dev,
                                  void* corr_items_dev, hipStream_t stream, hipEvent_t ev_begin, hipEvent_t ev_end) {
  if (ev_begin) (void)hipEventRecord(ev_begin, stream);
  hipLaunchKernelGGL(k_sfm_error_batch, dim3(blocks, n), dim3(kT), 0, stream, descs_dev, W, H, huber_delta, partials_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (ev_end) (void)hipEventRecord(ev_end, stream);
  launch_finalize_rows(n, blocks, (int)kFinalCorr, (const float*)partials_dev, corr_items_dev, (size_t)16, stream);
  return hipGetLastError();
}

hipError_t launch_se3_step_batch(const SimplePairDev* descs_dev, int n, int W, int H, float huber_delta, int blocks, float* partials_dev,
                                 void* items_dev, hipStream_t stream, hipEvent_t ev_begin, hipEvent_t ev_end) {
  if (ev_begin) (void)hipEventRecord(ev_begin, stream);
  hipLaunchKernelGGL(k_se3_step_batch, dim3(blocks, n), dim3(kT), 0, stream, descs_dev, W, H, huber_delta, partials_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (ev_end) (void)hipEventRecord(ev_end, stream);
  launch_finalize_rows(n, blocks, (int)kFinalItem6, (const float*)partials_dev, items_dev, (size_t)120, stream);
  return hipGetLastError();
}

hipError_t launch_se3_warp(const SimplePairDev& p, int W, int H, int blocks, float* partials_dev, void* corr_item_dev,
                           hipStream_t stream, const DoneFlag& done) {
  hipLaunchKernelGGL(k_se3_warp, dim3(blocks), dim3(kT), 0, stream, p, W, H, partials_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  launch_finalize_rows(1, blocks, (int)kFinalCorr, (const float*)partials_dev, corr_item_dev, (size_t)0, stream, done);
  return hipGetLastError();
}

hipError_t launch_squared_error(const float* a, uint32_t pitch_a, const float* b, uint32_t pitch_b, int W, int H, int blocks,
                                float* partials_dev, float* out_dev, hipStream_t stream, const DoneFlag& done) {
  hipLaunchKernelGGL(k_squared_error, dim3(blocks), dim3(kT), 0, stream, a, pitch_a, b, pitch_b, W, H, partials_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  launch_finalize_rows(1, blocks, (int)kFinalScalar, (const float*)partials_dev, out_dev, (size_t)0, stream, done);
  return hipGetLastError();
}

hipError_t launch_update_depth(int cs, const float* code_dev, const float* prx_orig, uint32_t pitch_prx, const float* jac,
                               uint32_t pitch_jac, float avg_dpt, float* dpt_out, uint32_t pitch_out, int W, int H,
                               hipStream_t stream) {
  const int nchunks = (W * H + 63) / 64;
  int blocks = (nchunks + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  switch (cs) {
    case 16: hipLaunchKernelGGL(k_update_depth<16>, dim3(blocks), dim3(kT), 0, stream, code_dev, prx_orig, pitch_prx, jac, pitch_jac, avg_dpt, dpt_out, pitch_out, W, H); break;
    case 32: hipLaunchKernelGGL(k_update_depth<32>, dim3(blocks), dim3(kT), 0, stream, code_dev, prx_orig, pitch_prx, jac, pitch_jac, avg_dpt, dpt_out, pitch_out, W, H); break;
    case 64: hipLaunchKernelGGL(k_update_depth<64>, dim3(blocks), dim3(kT), 0, stream, code_dev, prx_orig, pitch_prx, jac, pitch_jac, avg_dpt, dpt_out, pitch_out, W, H); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_update_depth_batch(int cs, const DepthJobDev* jobs_dev, int njobs, float avg_dpt, int W, int H, hipStream_t stream) {
  const int nchunks = (W * H + 63) / 64;
  int blocks = (nchunks + 3) / 4;
  // 32768 workgroups over the whole batch = two or three 64-pixel chunks per wave.  Swept on MI355X for 64 keyframes of 640x480 (2.7 GB; tools/decoder_bench.py, two
  // interleaved rounds): 4096 -> 469-472 us, 8192 (rounds 2-5) -> 452-456, 16384 -> 443-444, 32768 -> 433-436 = 0.77 of 8 TB/s, 65536 -> 433-434, 131072 -> 436-438.  A streaming form
  // (contiguous chunk range per wave, next chunk prefetched) measured 500 us: the 90 registers it needs cost three of the eight waves per SIMD
  const int cap = (32768 + njobs - 1) / njobs;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const dim3 grid(blocks, njobs);
  switch (cs) {
    case 16: hipLaunchKernelGGL(k_update_depth_batch<16>, grid, dim3(kT), 0, stream, jobs_dev, avg_dpt, W, H); break;
    case 32: hipLaunchKernelGGL(k_update_depth_batch<32>, grid, dim3(kT), 0, stream, jobs_dev, avg_dpt, W, H); break;
    case 64: hipLaunchKernelGGL(k_update_depth_batch<64>, grid, dim3(kT), 0, stream, jobs_dev, avg_dpt, W, H); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_sobel(const float* img, uint32_t pitch, float* grad, uint32_t gpitch, int W, int H, hipStream_t stream) {
  hipLaunchKernelGGL(k_sobel, dim3((W + 63) / 64, (H + 3) / 4), dim3(kT), 0, stream, img, pitch, grad, gpitch, W, H);
  return hipGetLastError();
}

hipError_t launch_blur_down(const float* in, uint32_t pitch, int W, int H, float* out, uint32_t opitch, int OW, int OH,
                            hipStream_t stream) {
  hipLaunchKernelGGL(k_blur_down, dim3((OW + 63) / 64, (OH + 3) / 4), dim3(kT), 0, stream, in, pitch, W, H, out, opitch, OW, OH);
  return hipGetLastError();
}

}  // namespace dfx
