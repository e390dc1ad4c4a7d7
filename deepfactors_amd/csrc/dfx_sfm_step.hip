// dfx_sfm_step.hip -- SfmAligner::RunStep for gfx950 (replaces kernel_step_calculate + kernel_finalize_reduction,
// reference sources/cuda/cu_sfmaligner.cpp:40-70,149-185 and the DenseSfm item, common/algorithm/dense_sfm.h:133-201).
//
// Design:
//   The reference keeps 1035 fp32 accumulators per THREAD (4 KB, spilled) and launches 11x32 threads.
//   Here a 64-lane wave owns 64 consecutive pixels per step ("chunk") and the JtJ / Jtr / r^2 / inlier sums are
//   one rank-1-update GEMM  Z += z z^T  over pixels, executed on the matrix cores with exact-fp32 MFMA
//   (v_mfma_f32_16x16x4_f32: k = 4 pixels per instruction, bitwise an fmaf chain).
//   z (per pixel, "z-space") = [ P | C_0 .. C_{NCB-1} ]:
//     P  : [ w*gC (6), w*r, 0 ]                                                  (8 rows; s = w * dE/dprx rides in LDS row 13)
//     C_b: [ s * jac[NCB*i + b] ]_{i=0..15}, b = 0..NCB-1                        (NCB = CS/16)
//   gC is the 1x6 Jacobian of the residual w.r.t. the RELATIVE pose (warping.h:156-164,247-257).  The chain rule onto
//   (pose0, pose1), J = gC * [blkdiag(M,M) | [[-M,-HM],[0,-M]]] (warping.h:119-134), is a per-pair constant 6 -> 12 map
//   T, so it is applied to the reduced sums by k_sfm_finalize (T G T^T, T X, T g in double) instead of to every pixel.
//   Matrix-core time and vector-ALU time ADD on gfx950, so MFMA slots are not wasted on the symmetric halves of the
//   diagonal blocks: the 16x16 products are PACKED (A and B are arbitrary 16-row selections of z):
//     X(b,b') : C_b x C_b'  (b < b')                       all 256 sums needed
//     Pm(b)   : b even: [ P (8) ; C_b rows 8..15 ] x C_b    pose-code, Jtr(code) and the rows 8..15 of C_b x C_b
//               b odd : [ C_b rows 0..7 ; P (8) ] x C_b     same with the rows 0..7 of C_b x C_b
//     Dd(q)   : M = [ C_2q rows 0..7 ; C_2q+1 rows 8..15 ]: the six 4x4 tiles of the two 8x8 diagonal blocks the Pm products
//               leave open, on TWO v_mfma_f32_4x4x1_16B_f32 (8 cycles each, A = B = M): the operand layout of a 16x16x4 MFMA
//               (lane = 16 k + row) is also 16 blocks (pixel k, row group rg) of the 4x4x1 form, and its A-broadcast modifier
//               (CBSZ = 1, ABID = 0: both blocks of a pair use the A rows of the first) yields M0 M0^T, M0 M1^T, M2 M2^T,
//               M2 M3^T in one instruction; the plain form adds M1 M1^T and M3 M3^T.  16 cycles instead of a 32-cycle 16x16x4
//               MFMA of which 72 of 256 sums were needed (tools/ubench/mfma4x4_bcast.cpp pins the modifier's semantics).
//   Matrix-pipe cycles per 4 pixels at CS = 32: 3 x 32 + 16 = 112 (was 128; 192 for the plain upper block-triangle); the 55
//   needed 4x4 tiles of the 40 x 40 z-space cost 110 at the pipe's rate.
//   The 29 needed sums of P x P (6x6 upper triangle, 6 Jtr, r^2, inliers; P row 7 = inlier flag) run on
//   v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 outer products per instruction = the three upper 4x4 tiles of P P^T for
//   five pixels, 13 instructions (8 cycles each) per chunk, operands straight from the LDS P rows -- 4 accumulator
//   registers instead of 29 per-lane sums (which cost an occupancy step) and no cross-lane reduction at the end.
//   Mixed operands are lane selects (one v_cndmask_b32 each, loop-invariant mask): the P rows are read from LDS by lanes
//   0..7 AND 8..15 of every 16-lane row (address (lane & 7): a broadcast, no bank conflict), so either half can carry them.
//   fp32 MFMAs execute on the vector ALU's lanes (tools/ubench/issue_cost.cpp: a co-resident wave's VALU instructions
//   get no issue slot while fp32 MFMAs run; every VALU instruction adds ~2.2 cycles, a DPP one ~4.4, to the 32 of a
//   16x16x4), so SIMD time per chunk = 32 x #MFMA16 + 8 x #MFMA4 + 2.2 x #VALU: both terms are kept minimal.
//
//   Phase A (lane = pixel): coalesced img0/dpt0 loads, warp, bilinear gathers of img1/grad1, Jacobian row,
//     Huber weight; the P row goes to LDS component-major (row stride 68 floats).
//   Phase B (lane = (i = lane&15, k = lane>>4)): the code Jacobian is loaded from HBM directly in MFMA operand
//     layout -- lane (i,k) reads NCB consecutive floats of pixel 4g+k, i.e. every wave-load is one fully
//     contiguous 256*NCB-byte run of the [H][W*CS] stream (86 % of all bytes).  The 16 operand registers are a
//     ring: as soon as the MFMAs of group g have consumed jv[g] it is refilled with the same group of the wave's
//     NEXT chunk, so every wave keeps ~8 KB of the stream in flight through both phases (no VGPR double buffer).
//     Zero weights use v_mul_legacy (0 * NaN = 0) so that garbage in the Jacobian of a masked pixel cannot
//     poison the sums.
//   Epilogue: waves fold their accumulators through LDS in fixed order; each workgroup writes one z-space partial;
//     k_sfm_finalize sums the partials of a pair in double, in fixed order (bit-reproducible for a given launch
//     shape), and scatters into the reference's JTJJrReductionItem layout (reduction_items.h:77-143).
//   The DEFAULT evaluation mode since round 3 (DFX_MFMA_BF16X3, template flag B3; the chain above is DFX_MFMA_F32_CHAIN): the fp32 MFMA runs at the
//     vector ALU's rate, the bf16 MFMA at 16x that.  Every fp32 entry of z is split EXACTLY into three bf16 pieces (x = h + m + l, round-to-nearest-
//     even through v_cvt_pk_bf16_f32, remainders through v_dot2c_f32_bf16: 7 VALU instructions per pair of values) and z z^T is summed as
//     hh + hm + mh + hl + lh + mm on v_mfma_f32_16x16x32_bf16 (fp32 accumulate; the dropped terms ml + lm + ll are <= 2^-23 of a product in the worst case -- |m| <= 2^-8 |x|, |l| <= 2^-16 |x| -- the order of an fp32 multiply's own rounding, 2^-24).  16x16 tiles (P,P),
//     (P,C_b), (C_b,C_b'), b <= b', with the P block stacked as [P_h ; P_m] and four-product diagonal tiles: 48 MFMAs per chunk at CS = 32.  The
//     operand ring, phase A, the pipeline, both schedules and the epilogue are shared.  Reduction tail: k_sfm_tail_b3 (a workgroup per pair, the
//     keyframe graph's assembly folded in) for batches, k_sfm_finalize_b3 (a workgroup per tile) for single pairs and pairs with > 512 KB of partials.
//     Measured on MI355X: CS = 32 at 640x480 956 us per 128 pairs (chain 1037), CS = 64 at 1280x960 883-922 us per 16 pairs (1148).
//     On this part the kernel runs at the package's power cap (shader clock 1.65-1.8 GHz), where its vector-ALU + matrix issue time is 95 % of
//     the kernel time and the memory system delivers what it can sustain: both terms above still count one to one.
#include "dfx_device.hpp"
#include "dfx_kernels.hpp"

namespace dfx {

// ---- build-time switches: launch shape (A/B-tested on MI355X) and diagnosis-only instrumentation ----
#ifndef DFX_MIN_WAVES
#define DFX_MIN_WAVES 3      // __launch_bounds__ waves per SIMD the register allocator must allow
#endif
#ifndef DFX_MIN_WAVES_CS64
#define DFX_MIN_WAVES_CS64 2 // same for NCB = 4 (15 accumulators + a 64-register ring: 168 VGPRs spill a little, 256 do not)
#endif
#ifndef DFX_EXTRA_LDS
#define DFX_EXTRA_LDS 0      // diagnosis only: dynamic LDS bytes added to the step launch to cap the workgroups per CU
#endif
#ifndef DFX_BANDED
#define DFX_BANDED 1         // chunk -> wave map: 1 = vertical bands (taps of consecutive chunks share image rows), 0 = interleaved
#endif
#ifndef DFX_TRACE
#define DFX_TRACE 0          // 1: per-wave s_memtime sums of phase A / phase B in the junk row 15 of the (P,P) partial
#endif
#ifndef DFX_ABLATE
#define DFX_ABLATE 0         // diagnosis only (wrong results; the bf16 split honours bits 1, 2, 4, 1024 = matrix instructions replaced by vector-ALU ones, 2048 = phase B reduced to the operand stream): 1 = no MFMAs, 2 = no phase-A math/gathers, 4 = no ring loads,
                             // 16 = no P x P fmas, 32 = no ray-table / valid0 reads, 64 = no operand shuffles, 128 = no tap gathers
#endif

#ifndef DFX_B3_DIAG4_MAX_NCB
#define DFX_B3_DIAG4_MAX_NCB 4   // DFX_MFMA_BF16X3: up to this many code blocks the diagonal tiles (C_b,C_b) take four products instead of six: S = hh + mm
#endif                           // and N = hm + hl in two accumulators, Z = S + N + N^T in the finalize kernel.  MI355X (profiles/r03_ab_variants.txt): CS = 32,
                                 // 128 pairs 1039.5 -> 1004.0 us; CS = 64, 1280x960, 16 pairs 950.9 -> 929.9 us (232 -> 247 VGPRs, still two waves).  0 disables
                                 // it.  Two round-2 ideas measured in round 3 and removed: the P block split in phase A and handed over as packed bf16
                                 // through LDS (+2.4 %), 128 registers for a fourth wave (spills: +37 %).
#ifndef DFX_STEP_TAPS_DWORD
#define DFX_STEP_TAPS_DWORD 0   // A/B: the img1 taps of phase A as four dword loads instead of two 8-byte loads (the row-walk reductions gain 5-7 % from it;
                                // this kernel nothing: 957.4 / 962.9 against 975.7 / 960.3 us in interleaved bench runs, profiles/r04_tap_loads.txt)
#endif
#ifndef DFX_RING_AUX
#define DFX_RING_AUX 2       // cache policy of the code-Jacobian stream loads: 2 = nt (read once: do not displace the img1 / grad1 rows
#endif                       // the bilinear taps of the next chunk row re-use from the L2); -1.6 % kernel time, -3 % read requests with the collapse below
#ifndef DFX_STREAM_AUX
#define DFX_STREAM_AUX 1     // cache policy of the coalesced img0 / dpt0 loads (read once per launch): 1 = sc0, they bypass the CU's L1 and leave
                             // it to the bilinear taps; -1 % (1055-1059 vs 1067 us, 4 interleaved runs each); nt (2) and sc0+nt (3) are no better
#endif
// Measured in round 3 and removed (profiles/r03_ab_occupancy.txt): a fourth wave per SIMD for the bf16 split (147 -> 128 registers by way of a
// half-size operand ring and a two-part epilogue fold that brings the workgroup's LDS from 37 to 20 KB): the 16-20 registers that still
// spill put scratch loads into the in-order vmcnt queue of the pipelined loop: 1367-1415 us against 995-1007 us.  The half-size ring alone
// (+0.6 %) and the two-part fold alone (+-0) at three waves change nothing.
#ifndef DFX_TAIL_KERNEL
#ifndef DFX_FIN_TRACE
#define DFX_FIN_TRACE 0   // debug builds (tools/variants.sh fintrace:"-DDFX_FIN_TRACE=1"): timestamps of the single-pair finalize kernel through printf
#endif
#define DFX_TAIL_KERNEL 1    // batched bf16-split launches: 1 = k_sfm_tail_b3 (a workgroup per pair, graph assembly folded in), 0 = k_sfm_finalize_b3 (a workgroup
#endif                       // per tile of a pair) and a separate assembly kernel -- the A/B switch of DESIGN.md 3.7
#ifndef DFX_TAIL_MAX_KB
#define DFX_TAIL_MAX_KB 512  // ... while a pair's partials are at most this many KB: the tail kernel reads a pair with ONE workgroup, the per-tile kernel with one per tile.
#endif                       // 128 pairs x 30 partials x 9 KB = 270 KB per pair: 16 us against 15 + 6.5 us in two kernels; 16 pairs of 1280x960 at CS = 64 (160 x 20 KB
                             // = 3.2 MB per pair, 16 workgroups in all): 90 us against 45 us (profiles/r03_ab_launch_shape.txt, call 24) -- those take the per-tile kernel
#ifndef DFX_DYN_ROT
#define DFX_DYN_ROT 4        // dynamic schedule: member row m of the teams serves the pairs rotated by DFX_DYN_ROT * m (0: a pair's team sits on one XCD)
#endif
#ifndef DFX_WAVES
#define DFX_WAVES 4
#endif
constexpr int kWaves = DFX_WAVES;         // waves per workgroup
constexpr int kThreads = kWaves * 64;
#ifndef DFX_USTRIDE
#define DFX_USTRIDE 68       // floats between the P rows of a wave in LDS: bank = (4 * row + pixel) % 32 keeps the 16x16 operand reads (8 rows x 4
#endif                       // pixels) and the 4x4 tile reads (8 rows x 3 pixels per half wave) apart.  66 had twice the bank conflicts (32 M vs
                             // 16 M per 128-pair launch), 65 five times -- and all of 65 .. 80 measure the same kernel time within 0.5 %
                             // (an A/B of round 2, script in git history): the LDS is 22 % busy and never the limiter.
constexpr int kUStride = DFX_USTRIDE;
constexpr int kUFloats = 16 * kUStride;   // per wave

template <int NCB> struct JV;
template <> struct JV<1> { typedef float T; };
template <> struct JV<2> { typedef f32x2 T; };
template <> struct JV<4> { typedef f32x4 T; };

template <int NCB> __device__ __forceinline__ float jv_get(const typename JV<NCB>::T& v, int b);
template <> __device__ __forceinline__ float jv_get<1>(const float& v, int) { return v; }
template <> __device__ __forceinline__ float jv_get<2>(const f32x2& v, int b) { return b == 0 ? v.x : v.y; }
template <> __device__ __forceinline__ float jv_get<4>(const f32x4& v, int b) { return b == 0 ? v.x : (b == 1 ? v.y : (b == 2 ? v.z : v.w)); }

// Byte offset of one code-Jacobian operand vector of the ring inside the [H][W*CS] image: pixel (pbase + 4*gq + lk),
// codes NCB*li .. NCB*li+NCB-1.  JDENSE (rows back to back) makes the stream linear in the pixel index; the pitched
// variant pays one integer division per vector.  Pixels past the image exist only in the last chunk of an image whose
// pixel count is not a multiple of 64; that case always takes the pitched variant (launch_t), which clamps them to pixel 0
// (their weight is 0): no wave-load of the pipelined loop ever has all its lanes out of range (see the note in k_sfm_step).
template <int NCB, bool JDENSE>
__device__ __forceinline__ unsigned jv_offset(unsigned pbase, int gq, int li, int lk, unsigned W, unsigned npx, unsigned jac_pitch) {
  constexpr unsigned VB = 4u * NCB;   // bytes per operand vector
  const unsigned p = pbase + 4u * gq + lk;
  if (JDENSE) return (p * 16u + li) * VB;
  const unsigned y = p / W, x = p - y * W;
  return p < npx ? y * jac_pitch + (x * 16u + li) * VB : (unsigned)li * VB;
}

// ---- DFX_MFMA_BF16X3 helpers -----------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));   // operand type of __builtin_amdgcn_mfma_f32_16x16x32_bf16 (8 bf16 = 4 VGPRs)
constexpr int b3_tiles(int ncb) { return 1 + ncb + ncb * (ncb + 1) / 2; }
constexpr bool b3_diag4(int ncb) { return ncb <= DFX_B3_DIAG4_MAX_NCB; }                     // two accumulators per diagonal (C_b,C_b) tile
constexpr int b3_blocks(int ncb) { return b3_tiles(ncb) + 1 + (b3_diag4(ncb) ? ncb : 0); }   // 256-float blocks of a partial: the tiles, then the N parts of (P,P) [, (C_b,C_b)]
// {RNE_bf16(lo) in bits 0..15, RNE_bf16(hi) in bits 16..31}: one v_cvt_pk_bf16_f32.  Written as a vector conversion, NOT as inline assembly:
// the compiler's hazard recognizer does not look inside an asm statement, so the wait states gfx950 needs between a vector-ALU write of
// a register and a matrix instruction reading it were missing whenever the scheduler put the two next to each other -- the MFMA then
// read the register's PREVIOUS content.  With the l pieces (2^-16 of a value) that is an error of ~1e-5 of single entries that varies
// from run to run with the waves' interleaving: found in round 3 as run-to-run differences of the (C_0,C_1) tile once the four-product
// diagonals had changed the instruction order (tools/diag_nan_batch.py, profiles/r03_inline_asm_hazard.txt).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  const f32x2 v = { lo, hi };
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// Exact three-way split of two fp32 values into packed bf16 pieces (tools/ubench/bf16x3_probe.cpp: 2^20 inputs reconstructed exactly
// on the hardware).  The subtractions are exact (Sterbenz-like: each remainder fits the fp32 mantissa).
// The remainders x - h through v_dot2c_f32_bf16: x + (p_lo, p_hi) . (-1, 0) expands the packed piece and subtracts it in ONE instruction per
// value (exact: the products are exact and the sum is representable) -- 7 instead of 9 vector-ALU instructions per pair of values, and the
// split is 40 % of the bf16 kernel's vector-ALU work (tools/ubench/bf16x3_probe.cpp section 2b: the same pieces for 2^20 inputs).
// The selector is (-1, -0) / (-0, -1), NOT (-1, 0): the compiler encodes the packed constant 0x0000bf80 as the inline constant "-1.0", which
// the instruction reads as the fp32 pattern 0xbf800000 = (0, -1) -- the wrong half (found by the probe: every second value wrong).  A
// pattern that is no inline constant travels as a 32-bit literal; the -0 * piece term only ever adds a signed zero.
#ifndef DFX_SPLIT_DOT2
#define DFX_SPLIT_DOT2 1
#endif
__device__ __forceinline__ float sub_bf16_lo(float x, unsigned p) {
#if DFX_SPLIT_DOT2
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, p), __builtin_bit_cast(bf16x2_t, 0x8000bf80u), x, false);
#else
  return x - __uint_as_float(p << 16);
#endif
}
__device__ __forceinline__ float sub_bf16_hi(float x, unsigned p) {
#if DFX_SPLIT_DOT2
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, p), __builtin_bit_cast(bf16x2_t, 0xbf808000u), x, false);
#else
  return x - __uint_as_float(p & 0xffff0000u);
#endif
}
__device__ __forceinline__ void split3_bf16(float x0, float x1, unsigned& ph, unsigned& pm, unsigned& pl) {
  ph = cvt_pk_bf16(x0, x1);
  const float r0 = sub_bf16_lo(x0, ph), r1 = sub_bf16_hi(x1, ph);
  pm = cvt_pk_bf16(r0, r1);
  const float s0 = sub_bf16_lo(r0, pm), s1 = sub_bf16_hi(r1, pm);
  pl = cvt_pk_bf16(s0, s1);
}
__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
#if DFX_ABLATE & 1
  return c;   // diagnosis: no matrix instructions (and, their operands being dead, no split either)
#elif DFX_ABLATE & 1024
  f32x4 r = c; r[0] += __uint_as_float((a[0] ^ b[1]) & 0x3fffffffu) + __uint_as_float((a[2] ^ b[3]) & 0x3fffffffu); return r;   // diagnosis: the split stays, the matrix instruction becomes 5 vector-ALU ones
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

// MODE 0: SfmAligner::RunStep.  MODE 1: DepthAligner::RunStep (cu_depthaligner.cpp:32-72) -- same rank-1 GEMM with
// z = [0 (6), diff, 0 | s * jac], s = -2 |diff| * dDepth/dPrx; `img0` carries the target depth and `dpt0`
// the current depth (already decoded by k_update_depth); every pixel is an inlier.

// lanes of the banks in BANKS (4-lane groups of every 16-lane row) <- src shifted inside its row (CTRL: 0x110 + n = row_shr:n,
// 0x100 + n = row_shl:n); the other lanes keep `old`.  One v_mov_b32_dpp.
template <int CTRL, int BANKS>
__device__ __forceinline__ float dpp_merge(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xF, BANKS, false));
}

// BYVAL: a single pair travels in the kernel arguments (`one`) instead of a descriptor array in device memory -- the reference's call
// pattern is one pair per blocking call, and the staged host-to-device copy of a 200-byte descriptor cost more than the launch.
//
// DYN (opt-in, dfx_set_schedule(ctx, DFX_SCHEDULE_DYNAMIC); W % 64 == 0, dense Jacobian rows): the waves are independent workers.  The
// grid is sized to be resident at once; wave g is member g / #pairs of one pair's team and pops ITEMS (one 64-pixel column x
// `rows_per_item` image rows, walked downwards) from the pair's queue with a scalar atomic until the queue is empty; its accumulators
// live in registers across all its items and are written ONCE, straight to global memory (no cross-wave fold, no barrier anywhere).
// Why: the four waves of a SIMD progress at very different rates (oldest-first arbitration: the same 150 chunks take one wave
// 520 us and another 1080 us), so any static split leaves slots idle -- a static single round measured 30 % idle slot-time, the
// multi-round launch 19 % (dispatch gaps, prologues, the barrier in front of the fold).  With the queue every slot streams until
// the pair's work is gone (3.5 - 7 % idle).  What it buys is small, though -- HBM and the issue pipe are the limiters, not the slots:
// -1.5 % kernel time on most boxes, +4.5 % on some, which is why the static launch is the default.  The
// price: which items a wave sums is decided at run time, so results are reproducible to fp32 re-association (1e-7 relative), not
// bit for bit.  Teams mix the dispatch ages (members g, g + #pairs, ...) and are rotated across the XCDs.
// VSH: every valid0 map of the launch is library-owned and carries a shadow (1 bit per pixel "known to hold 1.0"): 8 bytes are read per
// chunk instead of the map's 256.
template <int NCB, int MODE, bool JDENSE, bool TABLDS, bool BYVAL, bool DYN, bool B3, bool VSH>
__global__ __launch_bounds__(kThreads, NCB == 4 ? DFX_MIN_WAVES_CS64 : DFX_MIN_WAVES) void k_sfm_step(const SfmPairDev* __restrict__ pairs, const SfmPairDev one, const SfmParamsDev prm,
                                                       const int Wk, const int Hk, float* __restrict__ partials, const DynDev dyn,
                                                       const unsigned* __restrict__ blkmap) {
  static_assert(!DYN || (MODE == 0 && JDENSE && TABLDS && !BYVAL), "the dynamic schedule exists for the batched dense SfM step");
  static_assert(!VSH || MODE == 0, "valid0 maps exist for the SfM step only");
  constexpr int NX = NCB * (NCB - 1) / 2, ND = (NCB + 1) / 2;   // X(b,b'), Dd(q); Pm(b): NCB
  constexpr int NACC = NX + NCB;                                 // 16x16x4 accumulators
  constexpr int NT3 = b3_tiles(NCB);                             // B3: plain 16x16 tiles (P,P), (P,C_b), (C_b,C_b') b <= b'
  constexpr int NB3 = b3_blocks(NCB);                            // B3: accumulators = blocks of the partial
  constexpr int ZDIM = B3 ? NB3 * 256 : (1 + NACC + 2 * ND) * 256;   // block 0: the 29 P x P sums; then X, Pm, (Dd broadcast, Dd plain) per q
  constexpr int SLOT = (kUFloats > ZDIM) ? kUFloats : ZDIM;   // per wave: its P rows in the loop, its accumulators in the epilogue (same place)
  constexpr int LDS_FLOATS = kWaves * (DYN ? kUFloats : SLOT);   // DYN: no epilogue fold, only the P rows
  typedef typename JV<NCB>::T jv_t;
  // "No next chunk" is handled by re-reading the wave's current chunk (L2-hot, results never consumed), NOT by
  // out-of-range offsets: a wave-load whose lanes are all out of range completes without touching memory and may
  // retire ahead of older real loads, which breaks the counted vmcnt waits (observed as run-to-run differences).

  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  extern __shared__ float ray_lds[];   // TABLDS: the per-camera ray table, W + H + kRayTabSlack floats (dynamic LDS)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // DYN: wave -> (pair, team member); surplus waves (team * #pairs < waves of the grid) retire at once
  const int dyn_gid = DYN ? (int)blockIdx.x * kWaves + wave : 0;
  // member row m = gid / #pairs serves the pairs rotated by 4 m: workgroup b runs on XCD b mod 8, and without the rotation all
  // members of a pair (b = const mod 32 at 128 pairs) would sit on ONE XCD -- nothing would balance the XCDs against each other
  const int dyn_member = DYN ? dyn_gid / dyn.npairs : 0;
  const int dyn_pair = DYN ? (dyn_gid - dyn_member * dyn.npairs + DFX_DYN_ROT * dyn_member) % dyn.npairs : 0;
  if (DYN && dyn_member >= dyn.team) return;
  // Pairs of ONE image size: grid = (workgroups per pair, pairs).  Pairs of several sizes (pyramid levels in one launch): a 1-D grid and
  // `blkmap` (workgroup -> pair << 16 | block of that pair), the large pairs first so that the small levels fill the tail of the launch;
  // the pair's size, its number of workgroups and the index of its first partial come from its descriptor.
  const bool ragged = !BYVAL && !DYN && blkmap != nullptr;   // wave-uniform
  const unsigned bm = ragged ? blkmap[blockIdx.x] : 0u;
  const SfmPairDev& P = BYVAL ? one : pairs[DYN ? dyn_pair : (ragged ? (int)(bm >> 16) : (int)blockIdx.y)];
  const int W = ragged ? (int)P.w_px : Wk, H = ragged ? (int)P.h_px : Hk;
  const int blk_in_pair = ragged ? (int)(bm & 0xffffu) : (int)blockIdx.x;
  const int blks_of_pair = ragged ? (int)P.nblk : (int)gridDim.x;

  Geo g;
#pragma unroll
  for (int q = 0; q < 9; ++q) g.R[q] = P.R[q];
  g.t[0] = P.t[0]; g.t[1] = P.t[1]; g.t[2] = P.t[2];
  g.fx = P.fx; g.fy = P.fy; g.u0 = P.u0; g.v0 = P.v0; g.w = P.w; g.h = P.h;
  const uint32_t jac_pitch = P.pitch_jac;
  const uint32_t pitch_d0 = P.pitch_dpt0, pitch_i0 = P.pitch_img0, pitch_i1 = P.pitch_img1, pitch_g1 = P.pitch_grad1;
  const __amdgpu_buffer_rsrc_t i1_rs = make_rsrc(P.img1, (unsigned)H * pitch_i1);
  const __amdgpu_buffer_rsrc_t g1_rs = make_rsrc(P.grad1, (unsigned)H * pitch_g1);
  float* const valid0 = P.valid0;
  const uint32_t pitch_v0 = P.pitch_valid0;
  const __amdgpu_buffer_rsrc_t jac_rs = make_rsrc(P.jac, (unsigned)H * jac_pitch);
  // Dense Jacobian stream: the ring's addresses are (wave-uniform chunk base) + (per-lane constant) + (compile-time group
  // offset).  Re-basing the buffer resource per chunk keeps the first term on the scalar unit and the last in the
  // instruction's immediate, so the 16 refills cost no VALU op (a VGPR offset per load cost 19 v_or/v_add per chunk);
  // num_records shrinks with the base, so reads past the image still return 0.
  const char* const jac_ptr = reinterpret_cast<const char*>(P.jac);
  const unsigned jac_bytes = (unsigned)H * jac_pitch;
  auto ring_rsrc = [&](unsigned pbase) {
    const unsigned b = pbase * (64u * NCB);
    return make_rsrc(jac_ptr + b, jac_bytes - b);
  };
  const __amdgpu_buffer_rsrc_t d0_rs = make_rsrc(P.dpt0, (unsigned)H * pitch_d0);
  const __amdgpu_buffer_rsrc_t i0_rs = make_rsrc(P.img0, (unsigned)H * pitch_i0);
  const float inv_a = 1.0f / prm.avg_dpt;
  const char* const ray_tab = reinterpret_cast<const char*>(P.ray_tab);
  const char* const vld_base = valid0 ? reinterpret_cast<const char*>(valid0) : reinterpret_cast<const char*>(P.dpt0);
  const uint32_t vld_pitch = valid0 ? pitch_v0 : pitch_d0;
  // valid0 images owned by the library (dfx_img_alloc) carry a 1-bit-per-pixel SHADOW: bit p of the word array = "pixel p (linear index
  // y * W + x) is known to hold 1.0".  The wave then reads 8 bytes per 64-pixel chunk instead of 256 (the map is a write-only output of
  // the path, dense_sfm.h:161, all ones from the keyframe build on, mapper.cpp:937: the 4 B/px read was 2.7 % of the kernel's traffic
  // spent on learning that nothing has to be written).  A clear bit is always safe (the pixel is written again).  The bits are
  // maintained by the finalize kernel: see the stamp at the end of this kernel and rebuild_valid0_shadow.
  unsigned long long* const vshadow = (VSH && valid0) ? P.valid0_shadow : nullptr;
  // read as (uniform base) + (32-bit lane offset): the chunk's word, the lane's half of it.  A pair without a valid0 map reads the first
  // bytes of its depth image instead (the load count stays static, the result is never used).  No buffer resource: the kernel is short
  // of scalar registers (a spilled SGPR costs a VGPR lane, and the chain kernel sits at the 128-VGPR step of four waves per SIMD).
  const char* const vsh_base = vshadow ? reinterpret_cast<const char*>(vshadow) : reinterpret_cast<const char*>(P.dpt0);
  // (lane-derived constants of the shadow path are recomputed per chunk from an opaque copy of `lane` -- 4 VALU instructions -- instead of
  // living in two more loop-invariant registers: the chain kernel sits at the 128-register step of four waves per SIMD)
  auto opaque_lane = [&]() { unsigned l = (unsigned)lane; asm volatile("" : "+v"(l)); return l; };

  const int ntab = W + H + kRayTabSlack;
  float* const ray_w = DYN ? ray_lds + wave * ntab : ray_lds;   // DYN: the four waves of a workgroup serve four pairs (possibly four cameras)
  if (MODE == 0 && TABLDS) {
    // Stage the ray table in LDS: two global loads per chunk for it cost 9 % of the kernel (the L1 / address unit of a CU is
    // the busiest shared resource: 28 vector-memory instructions per chunk), two LDS reads cost nothing measurable.
    if (DYN) { for (int e = lane; e < ntab; e += 64) ray_w[e] = gload<float>(ray_tab + (unsigned)e * 4u); }
    else { for (int e = threadIdx.x; e < ntab; e += kThreads) ray_lds[e] = gload<float>(ray_tab + (unsigned)e * 4u); }
  }
  if (MODE == 0 && TABLDS && !DYN) __syncthreads();
  float* U = lds + wave * (DYN ? kUFloats : SLOT);
  if (lane < 16) U[(lane >> 1) * kUStride + 64 + (lane & 1)] = 0.f;   // padding columns 64, 65 of P rows 0..7 (read as zeros by the 4x4 tiles)
  f32x4 acc3[B3 ? NB3 : 1];                // B3: one accumulator per tile (+ the N parts of the diagonal tiles with DFX_B3_DIAG4)
#pragma unroll
  for (int a = 0; a < (B3 ? NB3 : 1); ++a) acc3[a] = f32x4{ 0.f, 0.f, 0.f, 0.f };

  f32x4 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) acc[a] = f32x4{ 0.f, 0.f, 0.f, 0.f };
  f32x4 accd[2 * ND];                           // Dd(q): [2q] = A-broadcast form, [2q + 1] = plain form; block (lane >> 2) = (pixel k, row group)
#pragma unroll
  for (int a = 0; a < 2 * ND; ++a) accd[a] = f32x4{ 0.f, 0.f, 0.f, 0.f };
  f32x4 accpp = f32x4{ 0.f, 0.f, 0.f, 0.f };   // P x P tiles: block b = lane >> 2 holds tile (b % 3) summed over its pixels
  // 4x4x1 operands: lane (b, i) reads P row 4*tr + i (A) / 4*tc + i (B) of pixel 5*t + b / 3; tiles (tr, tc) = (0,0), (0,1), (1,1);
  // block 15 and pixel 64 read the zero padding column of the LDS rows
  const int ppb = lane >> 2, ppi = lane & 3, pptyp = ppb % 3;
  const int ppoffA = ppb < 15 ? (4 * (pptyp == 2 ? 1 : 0) + ppi) * kUStride + ppb / 3 : 64;
  const int ppoffB = ppb < 15 ? (4 * (pptyp != 0 ? 1 : 0) + ppi) * kUStride + ppb / 3 : 64;

  const int npx = W * H;
  const int nchunks = (npx + 63) >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const bool lo8 = li < 8;              // lanes 0..7 of every 16-lane row
  const int uP_row = (li & 7) * kUStride;   // both halves of a row read P rows 0..7
  const unsigned ring_lane_off = (unsigned)(lk * 16 + li) * (4u * NCB);
  // `lo` is the lane offset made opaque once per chunk (ring_opaque_off): the group offsets then fold into the
  // instructions' 12-bit immediates (+ one v_add per 4 KB step) instead of LICM hoisting sixteen offset VGPRs.
  // `real` = false ("no next chunk"): the refill's results are never consumed, but it must stay a load that touches memory (a
  // wave-load with every lane out of range may retire ahead of older loads and break the counted waits): all lanes then read
  // the group's first vector of the CURRENT chunk -- one cache line per instruction instead of a second 256*NCB*16-byte
  // pass over a chunk the stream has long pushed out of the L2 (that pass was 1/15 of all ring traffic).
  auto ring_opaque_off = [&](bool real) { unsigned lo = real ? ring_lane_off : 0u; asm volatile("" : "+v"(lo)); return lo; };
  auto ring_load = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned lo, unsigned pbase, int gq, bool real) -> jv_t {
    if (JDENSE) return bload<DFX_RING_AUX>(rs, lo + (unsigned)gq * (256u * NCB), (jv_t*)nullptr);
    return bload<DFX_RING_AUX>(jac_rs, jv_offset<NCB, false>(pbase, gq, real ? li : 0, real ? lk : 0, W, npx, jac_pitch), (jv_t*)nullptr);
  };
  // Chunk -> wave map.  Banded (default): a wave walks DOWN the image -- chunk, chunk + one image row, ... -- so the
  // img1 / grad1 rows its bilinear taps share with the chunk below are re-read by the same CU (L2 hit) instead of by a
  // wave on another XCD (a second HBM fetch).  The Jacobian stream is still one contiguous 256*NCB*16-byte run per chunk.
  // Grids with fewer waves than chunks per row fall back to the interleaved map (wave w: w, w + #waves, ...).
  const int total_waves = blks_of_pair * kWaves;
  const int wid = blk_in_pair * kWaves + wave;
  const int vs = (W + 32) >> 6;   // chunk stride of one image row (exact when W % 64 == 0)
  int chunk, cstride, cend;
  // DYN: item t of a pair = column t % vs, image rows [t / vs * R, + R): chunks row * vs + column, one image row apart
  int gen_next = -1, gen_end = 0;   // generator: next chunk of the item being handed out, its end (wave-uniform)
  bool gen_dry = false;
  auto dyn_pop = [&]() -> unsigned {   // SMEM atomic: returns through lgkmcnt, so the counted vmcnt waits of the pipeline never see it
    unsigned t = 1u;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(dyn.qhead + dyn_pair) : "memory");
    return t;
  };
  // chunk id -> image row (chunks per row = vs <= 64, ids < 2^20: the multiply-high by dyn.vs_magic = 2^32 / vs + 1 is exact)
  auto dyn_row = [&](int c) -> int { return (int)__builtin_amdgcn_readfirstlane((int)__umulhi((unsigned)c, dyn.vs_magic)); };
  auto dyn_gen = [&]() -> int {        // next chunk id of this wave's sequence, -1 once the pair's queue is empty
    if (gen_next >= 0 && gen_next < gen_end) { const int c = gen_next; gen_next += vs; return c; }
    if (!gen_dry) {
      const unsigned t = dyn_pop();
      if (t < (unsigned)dyn.items_per_pair) {
        const int band = dyn_row((int)t);          // item t: column t % vs, row band t / vs
        const int row0 = band * dyn.rows_per_item;
        const int c = row0 * vs + ((int)t - band * vs);
        gen_end = min((row0 + dyn.rows_per_item) * vs, nchunks);
        gen_next = c + vs;
        return c;
      }
      gen_dry = true;
    }
    gen_next = -1;
    return -1;
  };
  int q0 = -1, q1 = -1, q2 = -1;   // DYN: the chunk being processed and the two the pipeline looks ahead to
  if (DYN) {
    q0 = dyn_gen(); q1 = dyn_gen(); q2 = dyn_gen();
    chunk = q0 < 0 ? nchunks : q0; cstride = vs; cend = nchunks;
  } else if (DFX_BANDED && vs >= 1 && total_waves >= vs) {
    const int crows = (nchunks + vs - 1) / vs;
    const int nbands = total_waves / vs;
    const int per = (crows + nbands - 1) / nbands;   // chunk rows per band = chunks per wave
    const int b = wid / vs, j = wid - b * vs;
    chunk = b * per * vs + j;
    cstride = vs;
    cend = (b < nbands) ? min(nchunks, (b * per + per) * vs) : 0;
    if (b >= nbands) chunk = nchunks;   // surplus waves idle
  } else {
    chunk = wid; cstride = total_waves; cend = nchunks;
  }

  // Per-lane pipeline registers.  Loads complete in order, so anything a phase waits for must be ISSUED before the
  // younger streaming loads it does not need:
  //   iteration c:  A2(c)   consume the gathers of chunk c (issued one iteration ago), write the P rows to LDS
  //                 A1(c+1) geometry of chunk c+1 from its prefetched depth, issue its img1 / grad1 gathers
  //                 prefetch depth / intensity of chunk c+2
  //                 B(c)    MFMAs; ring refilled with the Jacobian of chunk c+1
  // Every memory latency therefore hides under a full MFMA phase; nothing younger sits in front of a wait.
  struct Pix {            // one chunk's per-lane state between its A1 and A2
    int x, y;
    float d, i0;
    float rx, ry;         // K^-1 (x, y, 1) from the per-camera table
    unsigned vl;          // bits of the current valid0(x, y), or the lane's half of the chunk's shadow word: pixels that already hold 1.0 are not written again
    f32x2 ia, ib;         // img1 taps (row iy, row iy+1)
    f32x4 ga, gb;         // grad1 taps
    Corr c;               // correspondence of A1 (kept: cheaper than re-deriving it, 5 IEEE divisions)
    float ax, ay;         // bilinear fractions
    bool ok;
  };
  // (x, y) of a lane advance by a fixed pixel stride per iteration: no per-chunk integer division
  const unsigned pstride = (unsigned)cstride << 6;
  const int sdy = pstride / (unsigned)W, sdx = pstride - (unsigned)sdy * W;
  auto advance_xy = [&](int& x, int& y) {
    x += sdx; y += sdy;
    const bool wrap = x >= W;
    x -= wrap ? W : 0; y += wrap ? 1 : 0;
  };
  auto prefetch_di0 = [&](unsigned pbase, Pix& q) {   // coalesced depth / intensity of the chunk at pixel pbase; q.x, q.y set by caller
    const unsigned p = pbase + lane;
    const bool inb = p < (unsigned)npx;
    unsigned od = q.y * pitch_d0 + q.x * 4u, oi = q.y * pitch_i0 + q.x * 4u;
    od = inb ? od : 0u; oi = inb ? oi : 0u;   // in range on purpose (weight 0), see the note on all-lanes-out-of-range loads
    q.d = bload<DFX_STREAM_AUX>(d0_rs, od, (float*)nullptr);
    q.i0 = bload<DFX_STREAM_AUX>(i0_rs, oi, (float*)nullptr);
    if (DFX_ABLATE & 32) { q.rx = 0.01f * (float)q.x; q.ry = 0.01f * (float)q.y; q.vl = 0x3f800000u; }
    else if (MODE == 0) {   // table rows are padded (kRayTabSlack), so lanes past the last pixel stay inside the allocation
#if DFX_ABLATE & 512
      q.rx = ((float)q.x - g.u0) * (1.0f / g.fx); q.ry = ((float)q.y - g.v0) * (1.0f / g.fy);
#else
      if (TABLDS) { q.rx = ray_w[q.x]; q.ry = ray_w[W + q.y]; }
      else { q.rx = gload<float>(ray_tab + (unsigned)q.x * 4u); q.ry = gload<float>(ray_tab + (unsigned)(W + q.y) * 4u); }
#endif
      // valid0 is all ones from BuildKeyframe on (mapper.cpp:937) and only ever set: reading it (4 B/px, coalesced) and
      // skipping pixels that already hold 1.0 makes the steady state write-free; HBM writes cost about twice their bytes.
      // Always issued (keeps the load count static): without a valid0 image the read goes to the depth image instead.
#if DFX_ABLATE & 256
      q.vl = 0x3f800000u;
#else
      // one 4-byte load either way (the load count stays static): the lane's half of the chunk's shadow word, or its valid0 pixel
      if (VSH) q.vl = gload<unsigned>(vsh_base + (((opaque_lane() >> 3) & 4u) + (vshadow ? (pbase >> 6) * 8u : 0u)));
      else q.vl = gload<unsigned>(vld_base + (inb ? (unsigned)q.y * vld_pitch + (unsigned)q.x * 4u : 0u));
#endif
    }
  };
  // A1: warp the pixel and issue its 4 bilinear tap loads.  Branch-free on purpose: a conditional load would make the
  // number of loads younger than the ring path-dependent and force the compiler's vmcnt to the conservative minimum
  // (i.e. wait for these very gathers at the start of phase B).  Lanes without a correspondence read offset 0 of the image
  // (their weight is 0 through v_mul_legacy): an out-of-range offset would be cheaper per lane, but a chunk whose 64 pixels
  // ALL lack a correspondence (image regions that leave the view) would then issue wave-loads that touch no memory, and
  // those may retire ahead of the older ring loads the counted waits of phase B rely on.
#if DFX_STEP_TAPS_DWORD
  unsigned tap_four = 4u;
  asm volatile("" : "+s"(tap_four));   // opaque: the compiler would fuse the dword pairs back into 8-byte loads
#endif
  auto issue_gathers = [&](unsigned pbase, Pix& q) {
    if (MODE == 0) {
      const Corr c = find_correspondence_ray<B3>(g, q.rx, q.ry, q.d, prm.border, prm.min_dpt);
      const Taps tp = make_taps(c.u, c.v);
      const bool ok = c.valid && (pbase + lane < (unsigned)npx);
      q.c = c; q.ax = tp.ax; q.ay = tp.ay; q.ok = ok;
      unsigned oi = (unsigned)tp.iy * pitch_i1 + (unsigned)tp.ix * 4u, oi2 = oi + pitch_i1;
      unsigned og = (unsigned)tp.iy * pitch_g1 + (unsigned)tp.ix * 8u, og2 = og + pitch_g1;
      oi = ok ? oi : 0u; oi2 = ok ? oi2 : 0u;
      og = ok ? og : 0u; og2 = ok ? og2 : 0u;
#if DFX_ABLATE & 128
      q.ia = f32x2{ c.u, c.v }; q.ib = f32x2{ c.qx, c.qy }; q.ga = f32x4{ c.vx, c.vy, c.vz, c.iz }; q.gb = q.ga;
      (void)oi; (void)oi2; (void)og; (void)og2;
#else
#if DFX_STEP_TAPS_DWORD
      q.ia.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(i1_rs, (int)oi, 0, 0));
      q.ia.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(i1_rs, (int)oi, (int)tap_four, 0));
      q.ib.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(i1_rs, (int)oi2, 0, 0));
      q.ib.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(i1_rs, (int)oi2, (int)tap_four, 0));
#else
      q.ia = bload(i1_rs, oi, (f32x2*)nullptr);
      q.ib = bload(i1_rs, oi2, (f32x2*)nullptr);
#endif
      q.ga = bload(g1_rs, og, (f32x4*)nullptr);
      q.gb = bload(g1_rs, og2, (f32x4*)nullptr);
#endif
    } else {
      q.ia = f32x2{ 0.f, 0.f }; q.ib = q.ia;
      q.ga = f32x4{ 0.f, 0.f, 0.f, 0.f }; q.gb = q.ga;
      q.ok = false; q.ax = q.ay = 0.f;
    }
  };

  // ---- prologue: ring + state of the first chunk, depth of the second
  jv_t jv[16];
  Pix cur, nxt;
  {
    const unsigned base = (chunk < cend) ? (unsigned)chunk << 6 : 0u;   // idle wave: harmless loads of chunk 0
    if (DYN) { const int c = q0 < 0 ? 0 : q0, r = dyn_row(c); cur.x = (c - r * vs) * 64 + lane; cur.y = r; }
    else {
      const unsigned p = base + lane;   // the only integer division: first chunk of the wave
      cur.y = p / (unsigned)W;
      cur.x = p - cur.y * W;
    }
    prefetch_di0(base, cur);
    const bool has1 = DYN ? (q0 >= 0 && q1 >= 0) : (chunk + cstride < cend);
    const unsigned nb0 = has1 ? (unsigned)(DYN ? q1 : chunk + cstride) << 6 : base;
    nxt.x = cur.x; nxt.y = cur.y;
    if (has1) { if (DYN) { const int r = dyn_row(q1); nxt.x = (q1 - r * vs) * 64 + lane; nxt.y = r; } else advance_xy(nxt.x, nxt.y); }
    prefetch_di0(nb0, nxt);
    issue_gathers(base, cur);
    // Keep the issue order of the loop body (gathers, depth prefetch, THEN ring): the waitcnt bookkeeping at the loop
    // header merges this path with the back edge, and a younger gather here would cost a full drain every iteration.
    __builtin_amdgcn_sched_barrier(0);
    const __amdgpu_buffer_rsrc_t rs0 = ring_rsrc(base);
#pragma unroll
    for (int gq = 0; gq < 16; ++gq) jv[gq] = ring_load(rs0, ring_lane_off, base, gq, true);
    __builtin_amdgcn_sched_barrier(0);
  }

  // valid0 is only ever SET (dense_sfm.h:161).  A store inside the pipelined loop costs 9.6 % of the kernel (it sits in the
  // in-order vmcnt queue in front of every counted wait and needs an exec-masked branch), so every lane collects its
  // chunks' validity bits in a register and the wave writes them in one burst per 32 chunks / at the end.
  unsigned vmask = 0;
  bool wrote_valid = false;         // wave-uniform: this wave stored 1.0 somewhere -- the map's shadow is then rebuilt by the finalize kernel
  int vk = 0;                       // wave-uniform: chunks recorded in vmask
  int vx0 = cur.x, vy0 = cur.y;     // pixel of bit 0
  auto flush_valid = [&]() {
    // wave-uniform early-out: in the steady state (the map already holds 1.0 wherever a pixel is an inlier) no bit is set and the
    // burst loop -- ~10 VALU instructions per recorded chunk even when nothing is stored -- is skipped altogether
    if (MODE == 0 && valid0 && __builtin_amdgcn_ballot_w64(vmask != 0u) != 0ull) {
      wrote_valid = true;
      int fx = vx0, fy = vy0;
      for (int k = 0; k < vk; ++k) {
        if ((vmask >> k) & 1u) gstore<float>((char*)valid0 + (size_t)fy * pitch_v0 + (size_t)fx * 4, 1.0f);
        if (DYN) fy += 1; else advance_xy(fx, fy);   // DYN: the recorded chunks are one item's, one image row apart
      }
    }
    vmask = 0; vk = 0;
  };

#if DFX_TRACE
  unsigned long long trA = 0, trB = 0, trN = 0;
  const unsigned long long trStart = __builtin_amdgcn_s_memtime();
  const unsigned long long trRealStart = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz
#endif
  for (; chunk < cend; chunk = DYN ? (q0 < 0 ? nchunks : q0) : chunk + cstride) {
#if DFX_TRACE
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    const int base = chunk << 6;
    const bool has2 = DYN ? (q1 >= 0 && q2 >= 0) : (chunk + 2 * cstride < cend);                             // wave-uniform
    const bool has1 = DYN ? (q1 >= 0) : (chunk + cstride < cend);                                            // wave-uniform
    const unsigned nbase = has1 ? (unsigned)(DYN ? q1 : chunk + cstride) << 6 : (unsigned)base;             // else: re-read this chunk
    const unsigned nnbase = has2 ? (unsigned)(DYN ? q2 : chunk + 2 * cstride) << 6 : nbase;

    // ---- A2(c): lane = pixel; taps of this chunk were issued one iteration ago
    {
      const bool inb = (base + lane) < npx;
      float u16[16];
      if (MODE == 1 || (DFX_ABLATE & 2)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) u16[q] = 0.f;
      }
      if (MODE == 1) {
        if (inb) {
          const float d = cur.d;
          const float diff = cur.i0 - d;
          const float apd = prm.avg_dpt + d;
          u16[6] = diff;
          u16[13] = 2.0f * fabsf(diff) * (apd * apd) * inv_a;   // -2 |diff| * (-a / prx^2), prx = a / (a + d)
          u16[14] = 1.0f;
        }
      } else if (DFX_ABLATE & 2) {
        u16[0] = cur.d; u16[6] = cur.i0; u16[13] = 1.0f; u16[14] = 1.0f;
      } else {
        // Branch-free: every lane computes its row; lanes without a correspondence (or past the image) get weight 0
        // through v_mul_legacy (0 * NaN = 0), so no zero-init / masked overwrite and no exec juggling.
        const float d = cur.d;
        const float i0 = cur.i0;
        const Corr& c = cur.c;
        const bool ok = cur.ok && inb;
        const float tax = cur.ax, tay = cur.ay;
        const float gx = lerp1(lerp1(cur.ga.x, cur.ga.z, tax), lerp1(cur.gb.x, cur.gb.z, tax), tay);
        const float gy = lerp1(lerp1(cur.ga.y, cur.ga.w, tax), lerp1(cur.gb.y, cur.gb.w, tax), tay);
        const float samp = lerp1(lerp1(cur.ia.x, cur.ia.y, tax), lerp1(cur.ib.x, cur.ib.y, tax), tay);
        float gC[6], D00, D02, D11, D12;
        pose_row(g, c, d, gx, gy, gC, D00, D02, D11, D12);
        // d pix1 / d prx = D * (R ray) * (-a / prx^2),  prx = a / (a + d)   (warping.h:44-50,259-291)
        const float apd = prm.avg_dpt + d;
        const float dprx = -(apd * apd) * inv_a;
        const float pj0 = (D00 * c.rrx + D02 * c.rrz) * dprx;
        const float pj1 = (D11 * c.rry + D12 * c.rrz) * dprx;
        const float e = -(gx * pj0 + gy * pj1);
        const float r = i0 - samp;
        const float wgt = ok ? huber_weight(r, prm.huber_delta) : 0.0f;   // * DenseSfm_UncertaintyWeight == 1 (dense_sfm.h:66)
#pragma unroll
        for (int j = 0; j < 6; ++j) u16[j] = mul_zero_wins(wgt, gC[j]);   // relative-pose basis; (pose0, pose1) in k_sfm_finalize
        u16[6] = mul_zero_wins(wgt, r);
        u16[13] = mul_zero_wins(wgt, e);
        u16[14] = ok ? 1.0f : 0.0f;
        const bool is_one = VSH ? ((cur.vl >> (opaque_lane() & 31u)) & 1u) != 0u : cur.vl == 0x3f800000u;   // 0x3f800000 is the only pattern equal to 1.0f
        vmask |= ((ok && !is_one) ? 1u : 0u) << vk;   // valid0(x, y) = 1 (dense_sfm.h:161), written in bursts: flush_valid
      }
      {
#pragma unroll
      for (int q = 0; q < 7; ++q) U[q * kUStride + lane] = u16[q];
      U[7 * kUStride + lane] = u16[14];    // inlier flag: its square sums to the inlier count
      U[13 * kUStride + lane] = u16[13];
      }
      if (++vk == 32 || (DYN && (int)nbase != base + (vs << 6))) {   // DYN: the item ends with this chunk -- its bits form one burst
        flush_valid();
        vx0 = nxt.x; vy0 = nxt.y;   // bit 0 of the next burst = this lane's pixel of chunk c+1 (what nxt holds until A1 below)
      }
    }
    // ---- A1(c+1) and the depth prefetch of c+2: issued BEFORE the ring refills of phase B
    // (Tried: A1 + prefetch inside phase B, after MFMA group 1 / 4 / 8, so that their VALU chains fill the issue gaps of the MFMA
    // stream of the same wave: 139 VGPRs -> 3 waves per SIMD, +2.7 % kernel time.  The overlap comes from the co-resident waves.)
    cur.x = nxt.x; cur.y = nxt.y; cur.d = nxt.d; cur.i0 = nxt.i0; cur.rx = nxt.rx; cur.ry = nxt.ry; cur.vl = nxt.vl;
#if !(DFX_ABLATE & 2)
    issue_gathers(nbase, cur);
#endif
    if (has2) { if (DYN) { const int r = dyn_row(q2); nxt.x = (q2 - r * vs) * 64 + lane; nxt.y = r; } else advance_xy(nxt.x, nxt.y); }   // otherwise nxt keeps pointing at the last real chunk
    prefetch_di0(nnbase, nxt);
    __builtin_amdgcn_wave_barrier();   // LDS hand-over inside one wave: program order is enough for the hardware

#if DFX_TRACE
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long tr1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    // ---- phase B: rank-4 updates on the matrix cores; operand ring refilled behind the consumer
    const __amdgpu_buffer_rsrc_t nrs = ring_rsrc(nbase);
    const unsigned rlo = ring_opaque_off(has1);
    // (Tried in round 3: not streaming the Jacobian of a next chunk none of whose 64 pixels has a correspondence -- 2 - 3 % of the chunks of
    // a typical pair -- by pointing its refills at an L2-resident zero page.  The extra scalar state cost 8 VGPRs, i.e. the fourth wave per
    // SIMD of the fp32 chain: 1064 -> 1077 us; +-0 for the bf16 split.  Removed.)
    if constexpr (B3) {
      // ---- phase B, exact bf16 split (DFX_MFMA_BF16X3): every fp32 entry of z is split into three bf16 pieces, x = h + m + l EXACTLY
      // (h = RNE_bf16(x), m = RNE_bf16(x - h), l = x - h - m: 8 + 8 + 8 significant bits and a sign each), and z z^T is summed as
      // hh + hm + mh + hl + lh + mm on v_mfma_f32_16x16x32_bf16 (fp32 accumulate; of the dropped terms ml and lm are each <= 2^-24 of the
      // product (|m| <= 2^-8 |x|, |l| <= 2^-16 |x|) and ll <= 2^-32: together <= 2^-23 in the worst case, ~2^-26 on average -- the order of
      // the rounding of an fp32 multiply, 2^-24; include/dfx.h says the same).  A bf16 MFMA covers 32 pixels: half a chunk; lane (li, lk), slot j
      // of a half h holds pixel 4 (8 h + j) + lk -- the ring registers jv[8 h + j] as they are.  A and B use the same
      // pixel -> (lane group, slot) assignment, which is all the sum over k needs.  z blocks: 0 = P (rows 0..7; 8..15 zero), 1 + b = C_b.
      // P stacking: the P block has 8 rows, a tile 16.  Both halves of a 16-lane row read P row (li & 7) and split it; lanes 0..7 keep the
      // h piece and lanes 8..15 the m piece (operand Pa = [P_h ; P_m]), and Pl0 = [P_l ; 0].  One MFMA then yields two products:
      //   (P,C_b):  Pa x C_h = [hh ; mh],  Pa x C_m = [hm ; mm],  Pa x C_l = [hl ; (ml)],  Pl0 x C_h = [lh ; 0]      4 instead of 6 MFMAs
      //   (P,P)  :  Pa x Pa  = [hh hm ; mh mm]   and   Pl0 x Pa = [lh (lm) ; 0 0] in a second accumulator             2 instead of 6
      // (terms in parentheses are of the dropped 2^-24 class: harmless).  The finalize kernel adds the row halves / quadrants and
      // symmetrises lh (P P^T = sum of p p^T: hl = lh^T).  CS = 32 with the four-product diagonal tiles: 48 MFMAs per chunk (72 plain).
#if DFX_ABLATE & 2048
      // diagnosis: the operand stream alone -- every ring vector is consumed by ONE xor (no scaling, no split, no matrix instruction) and refilled
      {
        unsigned sink = 0;
#pragma unroll
        for (int gq = 0; gq < 16; ++gq) {
#pragma unroll
          for (int b = 0; b < NCB; ++b) sink ^= __float_as_uint(jv_get<NCB>(jv[gq], b));
          jv[gq] = ring_load(nrs, rlo, nbase, gq, has1);
        }
        acc3[0][0] += __uint_as_float(sink & 0x007fffffu);
      }
#else
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4 oh[1 + NCB], om[1 + NCB], ol[1 + NCB];   // packed bf16 pairs (slots 2 jp, 2 jp + 1) of the three pieces of the code blocks (index 0 unused)
        u32x4 pa, pl0;
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          float x[1 + NCB][2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int gq = 8 * h + 2 * jp + e;
            const int pp = 4 * gq + lk;
            const float s = U[13 * kUStride + pp];
            x[0][e] = U[uP_row + pp];
#pragma unroll
            for (int b = 0; b < NCB; ++b) x[1 + b][e] = mul_zero_wins(s, jv_get<NCB>(jv[gq], b));
#if !(DFX_ABLATE & 4)
            jv[gq] = ring_load(nrs, rlo, nbase, gq, has1);
#endif
          }
          {
            unsigned ph, pm, pl;
            split3_bf16(x[0][0], x[0][1], ph, pm, pl);
            pa[jp] = lo8 ? ph : pm;
            pl0[jp] = lo8 ? pl : 0u;
          }
#pragma unroll
          for (int k = 1; k < 1 + NCB; ++k) {
            unsigned ph, pm, pl;
            split3_bf16(x[k][0], x[k][1], ph, pm, pl);
            oh[k][jp] = ph; om[k][jp] = pm; ol[k][jp] = pl;
          }
        }
        // blocks of the partial: 0 = (P,P); 1 + b = (P,C_b); then (C_b,C_b') for b <= b' in row-major order; NT3 = N part of (P,P); with the
        // four-product diagonals NT3 + 1 + b = N part of (C_b,C_b).  Issue order: consecutive MFMAs write different accumulators; small terms first.
        auto pc = [&](const u32x4& PA, const u32x4 (&Bv)[1 + NCB]) {
#pragma unroll
          for (int b = 0; b < NCB; ++b) acc3[1 + b] = mfma_bf16(PA, Bv[1 + b], acc3[1 + b]);
        };
        auto cc_off = [&](const u32x4 (&A)[1 + NCB], const u32x4 (&Bv)[1 + NCB]) {
          int t = 1 + NCB;
#pragma unroll
          for (int b = 0; b < NCB; ++b)
#pragma unroll
            for (int b2 = b; b2 < NCB; ++b2, ++t) if (b2 != b) acc3[t] = mfma_bf16(A[1 + b], Bv[1 + b2], acc3[t]);
        };
        auto cc_diag = [&](const u32x4 (&A)[1 + NCB], const u32x4 (&Bv)[1 + NCB], bool npart) {
          int t = 1 + NCB;
#pragma unroll
          for (int b = 0; b < NCB; ++b) {
            const int td = npart ? NT3 + 1 + b : t;
            acc3[td] = mfma_bf16(A[1 + b], Bv[1 + b], acc3[td]);
            t += NCB - b;
          }
        };
        if constexpr (b3_diag4(NCB)) {   // diagonal tiles: S = mm + hh in the tile, N = hl + hm in its N block
          pc(pl0, oh); cc_off(om, om); cc_diag(om, om, false);
          pc(pa, ol); cc_off(oh, ol); cc_diag(oh, ol, true); cc_off(ol, oh);
          pc(pa, om); cc_off(oh, om); cc_diag(oh, om, true); cc_off(om, oh);
          acc3[NT3] = mfma_bf16(pl0, pa, acc3[NT3]);
          pc(pa, oh); cc_off(oh, oh); cc_diag(oh, oh, false);
          acc3[0] = mfma_bf16(pa, pa, acc3[0]);
        } else {
          pc(pl0, oh); cc_off(om, om); cc_diag(om, om, false);
          pc(pa, ol); cc_off(oh, ol); cc_diag(oh, ol, false); cc_off(ol, oh); cc_diag(ol, oh, false);
          pc(pa, om); cc_off(oh, om); cc_diag(oh, om, false); cc_off(om, oh); cc_diag(om, oh, false);
          acc3[NT3] = mfma_bf16(pl0, pa, acc3[NT3]);
          pc(pa, oh); cc_off(oh, oh); cc_diag(oh, oh, false);
          acc3[0] = mfma_bf16(pa, pa, acc3[0]);
        }
      }
#endif
    } else {
#if !(DFX_ABLATE & 16)
#pragma unroll
    for (int t = 0; t < 13; ++t)   // P x P: five pixels per instruction
      accpp = __builtin_amdgcn_mfma_f32_4x4x1f32(U[ppoffA + 5 * t], U[ppoffB + 5 * t], accpp, 0, 0, 0);
#endif
#pragma unroll
    for (int gq = 0; gq < 16; ++gq) {
      const int pp = 4 * gq + lk;
      const float uP = U[uP_row + pp];
      const float s = U[13 * kUStride + pp];
      float sc[NCB];
#pragma unroll
      for (int b = 0; b < NCB; ++b) sc[b] = mul_zero_wins(s, jv_get<NCB>(jv[gq], b));
#if !(DFX_ABLATE & 4)
      jv[gq] = ring_load(nrs, rlo, nbase, gq, has1);
#endif
#if DFX_ABLATE & 1
      acc[0][0] += uP;
#pragma unroll
      for (int b = 0; b < NCB; ++b) acc[b % NACC][1] += sc[b];
      continue;
#endif
      int a = 0;
#pragma unroll
      for (int b = 0; b < NCB; ++b)        // X(b,b'): C_b x C_b'
#pragma unroll
        for (int b2 = b + 1; b2 < NCB; ++b2, ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[b], sc[b2], acc[a], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < NCB; ++b, ++a) {   // Pm(b): even b: rows 0..7 = P, 8..15 = C_b rows 8..15; odd b: rows 0..7 = C_b rows 0..7, 8..15 = P
        const float mixed = (DFX_ABLATE & 64) ? uP : ((b & 1) ? (lo8 ? sc[b] : uP) : (lo8 ? uP : sc[b]));
        acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(mixed, sc[b], acc[a], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < ND; ++q) {         // Dd(q): M = [C_2q rows 0..7 ; C_2q+1 rows 8..15] (odd NCB, last q: C_2q twice)
        const int b0 = 2 * q, b1 = (2 * q + 1 < NCB) ? 2 * q + 1 : 2 * q;
        const float M = ((DFX_ABLATE & 64) || b1 == b0) ? sc[b0] : (lo8 ? sc[b0] : sc[b1]);
        accd[2 * q] = __builtin_amdgcn_mfma_f32_4x4x1f32(M, M, accd[2 * q], 1, 0, 0);           // M0 M0^T, M0 M1^T, M2 M2^T, M2 M3^T per pixel
        accd[2 * q + 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(M, M, accd[2 * q + 1], 0, 0, 0);   // (M0 M0^T), M1 M1^T, (M2 M2^T), M3 M3^T
      }
    }
    }   // !B3
#if DFX_TRACE
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long tr2 = __builtin_amdgcn_s_memtime();
    trA += tr1 - tr0; trB += tr2 - tr1; trN += 1;
    __builtin_amdgcn_sched_barrier(0);
#endif
    __builtin_amdgcn_wave_barrier();
    if (DYN) { q0 = q1; q1 = q2; q2 = dyn_gen(); }
  }

  flush_valid();
  // shadow protocol: a wave that changed the map stamps the launch id into the word in front of the shadow's bit array; the finalize kernel of
  // THIS launch then rebuilds the bits from the map itself (rebuild_valid0_shadow).  Steady state: no wave writes, nothing is rebuilt.
  if (VSH && vshadow && wrote_valid && lane == 0)
    gstore<unsigned>(reinterpret_cast<unsigned*>(vshadow - 1), prm.launch_id);

  // ---- epilogue: one z-space partial per workgroup = ((wave 0 + wave 1) + wave 2) + wave 3, element by element.
  // Every wave parks its accumulators in its own LDS region, ONE barrier, then all threads sum and store (the earlier
  // wave-after-wave read-modify-write needed four barriers with three waves idle each time; same sums, same order).
  // C/D layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
  // (Tried: no barrier at all -- every wave parks its set in its own slot, draws a ticket from an LDS counter, and the last one to
  // arrive folds while the others retire at once.  Waves idle ~10 % of their lifetime at this barrier, yet the kernel time did not
  // move (1059 vs 1058 us at 128 pairs): the kernel is bound by HBM traffic, not by resident waves; 3 instead of 4 workgroups per
  // CU cost 1.3 %.  Also without effect, 3 interleaved runs each at 128 pairs (1071-1089 us all): the valid0 read through a buffer
  // resource with the sc0 policy of the other streams; the four waves of a workgroup serving ONE pair in the dynamic schedule.)
  if (DYN) {   // one partial per wave = per team member, same z-space layout; nothing to fold
    float* mine = partials + ((size_t)dyn_pair * dyn.team + dyn_member) * ZDIM;
    if constexpr (B3) {
#pragma unroll
      for (int a = 0; a < NB3; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[a * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc3[a][r];
    } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[ppb * 16 + r * 4 + ppi] = accpp[r];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(1 + a) * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[a][r];
#pragma unroll
    for (int a = 0; a < 2 * ND; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(1 + NACC + a) * 256 + ppb * 16 + r * 4 + ppi] = accd[a][r];
    }
#if DFX_TRACE
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {   // trace builds only: {phase A cycles, phase B cycles, loop start (100 MHz ticks mod 2^24), loop lifetime in ticks}, chunks
      mine[15 * 16 + 0] = (float)trA;
      mine[15 * 16 + 1] = (float)trB;
      mine[15 * 16 + 2] = (float)(trRealStart & 0xFFFFFFull);
      mine[15 * 16 + 3] = (float)(__builtin_amdgcn_s_memrealtime() - trRealStart);
      mine[11 * 16 + 2] = (float)trN;
    }
#endif
    return;
  }
  {
    float* mine = lds + wave * SLOT;
    if constexpr (B3) {   // every tile in the C/D layout of a 16x16 MFMA: [row = 4 (lane >> 4) + r][col = lane & 15]
#pragma unroll
      for (int a = 0; a < NB3; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[a * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc3[a][r];
    } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[ppb * 16 + r * 4 + ppi] = accpp[r];   // block 0: [4x4 block b][row r][column = lane & 3]
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(1 + a) * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[a][r];
#pragma unroll
    for (int a = 0; a < 2 * ND; ++a)   // 4x4x1 form: [block = 4 k + rg][row r of the A group][column = lane & 3 of the B group]
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(1 + NACC + a) * 256 + ppb * 16 + r * 4 + ppi] = accd[a][r];
    }
  }
  float* out = partials + (ragged ? (size_t)P.blk0 + blk_in_pair : (size_t)blockIdx.y * gridDim.x + blockIdx.x) * ZDIM;
  __syncthreads();
  for (int e = threadIdx.x; e < ZDIM; e += kThreads) {
    float v = lds[e];
#pragma unroll
    for (int wv = 1; wv < kWaves; ++wv) v += lds[wv * SLOT + e];
    out[e] = v;   // (non-temporal stores here do not shorten the step -> finalize boundary: 39.6 vs 39.5 us between step time and kernel time, round 3)
  }
#if DFX_TRACE
  __syncthreads();
  if (lane == 0) {   // trace builds only (their block 11 of P x P is overwritten, results are not valid): rows 15 and 11 of block 0 carry per wave {phase A cycles, phase B cycles, start mod 2^24, lifetime}, {HW_ID, XCC_ID}
    const unsigned long long trEnd = __builtin_amdgcn_s_memtime();
    out[15 * 16 + wave * 4 + 0] = (float)trA;
    out[15 * 16 + wave * 4 + 1] = (float)trB;
    out[15 * 16 + wave * 4 + 2] = (float)(trStart & 0xFFFFFFull);
    out[15 * 16 + wave * 4 + 3] = (float)(trEnd - trStart);
    out[11 * 16 + wave * 4 + 0] = (float)__builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (15 << 11));
    out[11 * 16 + wave * 4 + 1] = (float)__builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (3 << 11));
    out[11 * 16 + wave * 4 + 2] = (float)trN;
    out[11 * 16 + wave * 4 + 3] = (float)(__builtin_amdgcn_s_memrealtime() - trRealStart);
  }
#endif
}

// ---- valid0 shadow maintenance (finalize kernels): when a wave of THIS launch stored 1.0 into the pair's valid0 map (stamp == launch id),
// the shadow bits are recomputed from the map itself: bit p = (valid0[p] == 1.0f).  The pair's finalize workgroups split the words; a map
// shared by several pairs of the launch is rebuilt by each of them with the same values.  Runs once per change of the inlier set
// (first step on a fresh map, newly exposed pixels after a pose update) -- never in the steady state.
__device__ __forceinline__ unsigned valid0_shadow_stamp(const SfmPairDev& P, unsigned launch_id) {   // != launch_id: nothing to rebuild
  unsigned long long* const sh = P.valid0_shadow;
  const bool have = sh && P.valid0;
  // an unconditional load (of the descriptor itself when there is no shadow): a load inside a branch is waited for at the end of the branch,
  // which put one memory round trip in front of everything else the finalize kernels do
  const unsigned v = gload<unsigned>(have ? reinterpret_cast<const unsigned*>(sh - 1) : reinterpret_cast<const unsigned*>(&P.w_px));
  return have ? v : launch_id + 1u;
}
__device__ __forceinline__ void rebuild_valid0_shadow(const SfmPairDev& P, int W, int H, unsigned launch_id, int part, int nparts, unsigned stamp) {
  unsigned long long* const sh = P.valid0_shadow;
  if (stamp != launch_id) return;
  const unsigned npx = (unsigned)W * (unsigned)H, nwords = (npx + 63u) >> 6;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  const unsigned step = (unsigned)(nparts * nwv);
  // four words of a wave in flight at a time (the tail kernel rebuilds a pair's map with ONE workgroup: 4800 words at 640x480)
  for (unsigned wd = (unsigned)(part * nwv + wv); wd < nwords; wd += 4u * step) {
    unsigned val[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned p = (wd + (unsigned)q * step) * 64u + (unsigned)lane;
      const unsigned pc = p < npx ? p : 0u;
      const unsigned y = pc / (unsigned)W, x = pc - y * (unsigned)W;
      val[q] = gload<unsigned>(reinterpret_cast<const char*>(P.valid0) + (size_t)y * P.pitch_valid0 + (size_t)x * 4u);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned w2 = wd + (unsigned)q * step;
      const unsigned long long bits = __builtin_amdgcn_ballot_w64(w2 * 64u + (unsigned)lane < npx && val[q] == 0x3f800000u);
      if (lane == 0 && w2 < nwords) sh[w2] = bits;
    }
  }
}

// ---- finalize: sum the workgroup partials of each pair (double, fixed order), map the relative-pose basis onto
// (pose0, pose1) and scatter the packed z-space blocks into the item layout.
// grid = (1 + NACC + 2 ND, npairs), 1024 threads: thread = (element of one 256-float block, 1 of 4 partial groups); groups
// stride over the pair's `bpp` partials (1 KB coalesced reads, 8 loads in flight) and are folded in fixed order.
template <int NCB, int NPOSE, bool BYVAL>
__device__ __forceinline__ void sfm_finalize_body(const float* __restrict__ partials, const int bpp, const SfmPairDev* __restrict__ pairs, const SfmPairDev& one,
                                                  char* __restrict__ items, const size_t item_stride, unsigned* __restrict__ qhead,
                                                  const int Wk, const int Hk, const unsigned launch_id, const int ragged) {
  constexpr int CS = 16 * NCB;
  constexpr int NP = NPOSE + CS;
  constexpr int NX = NCB * (NCB - 1) / 2, ND = (NCB + 1) / 2;
  constexpr int NACC = NX + NCB;
  constexpr int ZDIM = (1 + NACC + 2 * ND) * 256;
  constexpr int NT = NP * (NP + 1) / 2;
  __shared__ double red[4][256];
  __shared__ double T[12][6];   // d(pose0, pose1) <- d(relative pose): J = gC * T^T

  const int blk = blockIdx.x, pair = blockIdx.y;
  const int el = threadIdx.x & 255, rg = threadIdx.x >> 8;
  // pairs of several image sizes in one launch (`ragged`): size, number of partials and first partial come from the pair's descriptor
  const SfmPairDev& PD = BYVAL ? one : pairs[pair];
  const int W = ragged ? (int)PD.w_px : Wk, H = ragged ? (int)PD.h_px : Hk;
  const int nparts = ragged ? (int)PD.nblk : bpp;
  const size_t part0 = ragged ? (size_t)PD.blk0 : (size_t)pair * bpp;
  if (NPOSE == 12) rebuild_valid0_shadow(PD, W, H, launch_id, blk, (int)gridDim.x, valid0_shadow_stamp(PD, launch_id));
  if (qhead && blk == 0 && threadIdx.x == 0) qhead[pair] = 0u;   // dynamic schedule: the pair's item queue is rewound for the next launch
  const float* src = partials + part0 * ZDIM + blk * 256 + el;
  red[rg][el] = strided_sum_f64<4, 16>(src, rg, nparts, ZDIM);   // a single pair has 240 partials: 60 rows per group = 4 round trips
  if (NPOSE == 12 && threadIdx.x < 72) {
    const int n = threadIdx.x / 6, i = threadIdx.x - n * 6;
    const float* M = BYVAL ? one.M : pairs[pair].M;
    const float* HM = BYVAL ? one.HM : pairs[pair].HM;
    const int j = n % 3, grp = n / 3;   // grp 0: pose0 trs, 1: pose0 rot, 2: pose1 trs, 3: pose1 rot
    double v = 0.0;
    if (grp == 0) v = i < 3 ? (double)M[3 * i + j] : 0.0;
    else if (grp == 1) v = i >= 3 ? (double)M[3 * (i - 3) + j] : 0.0;
    else if (grp == 2) v = i < 3 ? -(double)M[3 * i + j] : 0.0;
    else v = i < 3 ? -(double)HM[3 * i + j] : -(double)M[3 * (i - 3) + j];
    T[n][i] = v;
  }
  __syncthreads();
  if (rg == 0) red[0][el] = ((red[0][el] + red[1][el]) + red[2][el]) + red[3][el];
  __syncthreads();
  const double* S = red[0];   // the block's 256 sums: S[row * 16 + col] for the MFMA blocks, 16 4x4 P x P blocks for block 0

  float* item = reinterpret_cast<float*>(items + (size_t)pair * item_stride);
  auto tri = [](int lo, int hi) { return lo * NP - lo * (lo - 1) / 2 + (hi - lo); };
  auto put = [&](int lo, int hi, float v) { item[tri(lo, hi)] = v; };   // item entry (lo <= hi) -> packed upper triangle
  auto put_code = [&](int ca, int cb, float v) {   // code-code entry by code indices
    const int n = NPOSE + ca, m = NPOSE + cb;
    put(n < m ? n : m, n < m ? m : n, v);
  };
  auto put_g = [&](int n, float v) { item[NT + n] = v; };
  // block 0 = 16 4x4 blocks [b][i][j] of the P x P tiles: block b (< 15) holds tile b % 3 ((0,0), (0,1), (1,1) of the 8x8
  // matrix P P^T) summed over its pixels; P = (gC_0..5, w r, inlier flag)
  __shared__ double G8[8][8];   // block 0 only: P P^T (upper triangle), each entry folded over the five 4x4 blocks that hold it
  if (blk == 0) {
    if (threadIdx.x < 64) {
      const int p = threadIdx.x >> 3, q = threadIdx.x & 7;
      if (p <= q) {
        const int typ = q < 4 ? 0 : (p < 4 ? 1 : 2);
        double v = 0.0;
#pragma unroll
        for (int b = typ; b < 15; b += 3) v += S[b * 16 + (p & 3) * 4 + (q & 3)];
        G8[p][q] = v;
      }
    }
    __syncthreads();
  }
  auto pp = [&](int p, int q) { return G8[p][q]; };   // sum over pixels of P_p * P_q, p <= q < 8
  const int t = threadIdx.x;
  if (blk == 0) {
    if (NPOSE == 12) {
      if (t < 144) {                       // pose-pose: T G T^T
        const int n = t / 12, m = t - n * 12;
        if (n <= m) {
          double v = 0.0;
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j) r += (i <= j ? pp(i, j) : pp(j, i)) * T[m][j];
            v += T[n][i] * r;
          }
          put(n, m, (float)v);
        }
      } else if (t < 156) {                // Jtr (pose): T * (gC . wr)
        const int n = t - 144;
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) v += T[n][i] * pp(i, 6);
        put_g(n, (float)v);
      }
    }
    if (t == 160) item[NT + NP] = (float)pp(6, 6);          // residual = sum (w r)^2
    if (t == 161) {
      const size_t off = (((size_t)(NT + NP + 1)) * 4 + 7) & ~(size_t)7;
      *reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(item) + off) = (unsigned long long)(pp(7, 7) + 0.5);
    }
    return;
  }
  const int a = blk - 1;
  if (a < NX) {
    // ---- X(b,b'): S[i][j] = C_b[i] * C_b'[j], b < b'
    int b = 0, b2 = 0;
    { int q = a; for (b = 0; b < NCB; ++b) { const int n = NCB - 1 - b; if (q < n) { b2 = b + 1 + q; break; } q -= n; } }
    if (t < 256) put_code(NCB * (t >> 4) + b, NCB * (t & 15) + b2, (float)S[t]);
  } else if (a < NX + NCB) {
    // ---- Pm(b): columns = C_b; even b: rows 0..5 = gC, 6 = w r, 7 unused, rows 8..15 = C_b rows 8..15
    //                            odd b : rows 0..7 = C_b rows 0..7, rows 8..13 = gC, 14 = w r, 15 unused
    const int b = a - NX;
    const int pofs = (b & 1) ? 8 : 0, cofs = 8 - pofs;
    if (NPOSE == 12 && t < 192) {          // pose-code: T * S[P rows 0..5][j]
      const int n = t >> 4, j = t & 15;
      double v = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) v += T[n][i] * S[(pofs + i) * 16 + j];
      put(n, NPOSE + NCB * j + b, (float)v);
    } else if (t >= 192 && t < 208) {      // Jtr (code)
      const int j = t - 192;
      put_g(NPOSE + NCB * j + b, (float)S[(pofs + 6) * 16 + j]);
    } else if (t >= 256 && t < 384) {      // C_b[r] * C_b[c], r in the 8 rows this block carries: upper part, plus (even b) the rows 8..15 x rows 0..7 rectangle
      const int r = cofs + ((t - 256) >> 4), c = t & 15;
      if (c >= r || c < cofs) put_code(NCB * r + b, NCB * c + b, (float)S[r * 16 + c]);
    }
  } else {
    // ---- Dd(q) on 4x4x1: 16 blocks (pixel k, row group rg) of [row i of the A group][column j of the B group]; M = [C_b0 rows 0..7 ;
    // C_b1 rows 8..15] in groups M0..M3.  form 0 (A broadcast inside block pairs): M0 M0^T, M0 M1^T, M2 M2^T, M2 M3^T; form 1: Mrg Mrg^T.
    const int d = a - NACC, q = d >> 1, form = d & 1;
    const int b0 = 2 * q, b1 = 2 * q + 1;   // b1 == NCB (odd NCB, last q): the upper half repeats C_b0 rows 8..15, which Pm(b0) already covers
    if (t < 64) {
      const int rg = t >> 4, i = (t >> 2) & 3, j = t & 3;
      const double v = ((S[rg * 16 + (t & 15)] + S[(4 + rg) * 16 + (t & 15)]) + S[(8 + rg) * 16 + (t & 15)]) + S[(12 + rg) * 16 + (t & 15)];   // the group's 4 pixels
      const int bb = rg < 2 ? b0 : b1;
      if (bb < NCB) {
        const int base = rg < 2 ? 0 : 8;
        if (form == 0) {
          if ((rg & 1) == 0) { if (i <= j) put_code(NCB * (base + i) + bb, NCB * (base + j) + bb, (float)v); }
          else put_code(NCB * (base + i) + bb, NCB * (base + 4 + j) + bb, (float)v);
        } else if ((rg & 1) == 1) {
          if (i <= j) put_code(NCB * (base + 4 + i) + bb, NCB * (base + 4 + j) + bb, (float)v);
        }
      }
    }
  }
}

// ---- finalize of the DFX_MFMA_BF16X3 step: the partials are plain 16x16 tiles S[row][col] (tile 0: P x P; 1 + b: P x C_b; then
// C_b x C_b', b <= b', row-major), entry i of C_b = code NCB * i + b.  Same reduction (double, fixed order), same T map, same item.
// Two kernels share the pieces below: k_sfm_finalize_b3 (one workgroup per tile of a pair: single pairs, DepthAligner) and
// k_sfm_tail_b3 (one workgroup per pair + the graph assembly: every batched launch).

// Block of the partial that holds the N part of tile `blk` (NT3 + d), or -1: d = 0 (P,P): N = P_l x [P_h P_m]; d = 1 + b: (C_b,C_b) with the
// four-product diagonals: N = hm + hl
template <int NCB>
__device__ __forceinline__ int b3_dtile(int blk) {
  if (blk == 0) return 0;
  if (b3_diag4(NCB) && blk > NCB) { int q = blk - 1 - NCB; for (int b = 0; b < NCB; ++b) { if (q == 0) return 1 + b; q -= NCB - b; if (q < 0) break; } }
  return -1;
}

// d(pose0, pose1) <- d(relative pose): J = gC * T^T; entry (n, i) for n < 12, i < 6
__device__ __forceinline__ double b3_T_entry(const float* M, const float* HM, int n, int i) {
  const int j = n % 3, grp = n / 3;   // grp 0: pose0 trs, 1: pose0 rot, 2: pose1 trs, 3: pose1 rot
  if (grp == 0) return i < 3 ? (double)M[3 * i + j] : 0.0;
  if (grp == 1) return i >= 3 ? (double)M[3 * (i - 3) + j] : 0.0;
  if (grp == 2) return i < 3 ? -(double)M[3 * i + j] : 0.0;
  return i < 3 ? -(double)HM[3 * i + j] : -(double)M[3 * (i - 3) + j];
}

// Undo the packing of the step kernel for element el = (r, cc) of tile `blk` (same order for every launch: deterministic): S = the tile's
// 256 sums, Sn = the sums of its N block (dtile >= 0).  `keep` = false: the element stays as summed.
template <int NCB>
__device__ __forceinline__ double b3_unpack(int blk, int dtile, int el, const double* S, const double* Sn, bool& keep) {
  const int r = el >> 4, cc = el & 15;
  double v = 0.0;
  keep = true;
  if (blk == 0) {            // (P,P): quadrants [hh hm ; mh mm] of P stacked as [P_h ; P_m], plus lh + lh^T from the N block's top-left quadrant
    if (r < 8 && cc < 8) {
      v = ((S[r * 16 + cc] + S[r * 16 + 8 + cc]) + S[(8 + r) * 16 + cc]) + S[(8 + r) * 16 + 8 + cc];
      v += Sn[r * 16 + cc] + Sn[cc * 16 + r];
    }
  } else if (blk <= NCB) {   // (P,C_b): rows 0..7 = (h + l) pieces, rows 8..15 = m pieces of the same P rows
    if (r < 8) v = S[r * 16 + cc] + S[(8 + r) * 16 + cc];
  } else if (dtile >= 0) {   // (C_b,C_b) with four products: Z = S + N + N^T
    v = S[el] + Sn[el] + Sn[cc * 16 + r];
  } else {
    keep = false;
  }
  return v;
}

// Scatter the unpacked tile `blk` (S[row * 16 + col]) into the item; t = 0 .. 255
// COH: the item is read by OTHER workgroups of the same kernel (graph assembly in k_sfm_tail_b3): device-scope stores (write-through,
// visible to device-scope loads from any XCD once the store has completed) instead of a cache writeback + invalidate per workgroup
template <bool COH, typename V>
__device__ __forceinline__ void coh_store(V* p, V v) {
  if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <int NCB, int NPOSE, bool COH = false>
__device__ __forceinline__ void b3_scatter(int blk, int t, const double* S, const double (*T)[6], float* item) {
  constexpr int CS = 16 * NCB;
  constexpr int NP = NPOSE + CS;
  constexpr int NT = NP * (NP + 1) / 2;
  auto tri = [](int lo, int hi) { return lo * NP - lo * (lo - 1) / 2 + (hi - lo); };
  auto put = [&](int lo, int hi, float v) { coh_store<COH>(&item[tri(lo, hi)], v); };
  auto put_code = [&](int ca, int cb, float v) {
    const int n = NPOSE + ca, m = NPOSE + cb;
    put(n < m ? n : m, n < m ? m : n, v);
  };
  auto put_g = [&](int n, float v) { coh_store<COH>(&item[NT + n], v); };
  if (blk == 0) {
    // P = (gC_0..5, w r, inlier flag); the tile holds both triangles of P P^T, the upper one is used
    auto pp = [&](int p, int q) { return S[p * 16 + q]; };
    if (NPOSE == 12) {
      if (t < 144) {                       // pose-pose: T G T^T
        const int n = t / 12, m = t - n * 12;
        if (n <= m) {
          double v = 0.0;
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j) r += (i <= j ? pp(i, j) : pp(j, i)) * T[m][j];
            v += T[n][i] * r;
          }
          put(n, m, (float)v);
        }
      } else if (t < 156) {                // Jtr (pose): T * (gC . wr)
        const int n = t - 144;
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) v += T[n][i] * pp(i, 6);
        put_g(n, (float)v);
      }
    }
    if (t == 160) coh_store<COH>(&item[NT + NP], (float)pp(6, 6));          // residual = sum (w r)^2
    if (t == 161) {
      const size_t off = (((size_t)(NT + NP + 1)) * 4 + 7) & ~(size_t)7;
      coh_store<COH>(reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(item) + off), (unsigned long long)(pp(7, 7) + 0.5));
    }
    return;
  }
  if (blk <= NCB) {
    // ---- (P, C_b): rows 0..5 = gC, row 6 = w r; column j = code NCB * j + b
    const int b = blk - 1;
    if (NPOSE == 12 && t < 192) {          // pose-code: T * S[P rows 0..5][j]
      const int n = t >> 4, j = t & 15;
      double v = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) v += T[n][i] * S[i * 16 + j];
      put(n, NPOSE + NCB * j + b, (float)v);
    } else if (t >= 192 && t < 208) {      // Jtr (code)
      const int j = t - 192;
      put_g(NPOSE + NCB * j + b, (float)S[6 * 16 + j]);
    }
    return;
  }
  // ---- (C_b, C_b'), b <= b': S[i][j] = C_b[i] * C_b'[j]; the diagonal tiles carry both triangles, i <= j is used
  int b = 0, b2 = 0;
  { int q = blk - 1 - NCB; for (b = 0; b < NCB; ++b) { const int n = NCB - b; if (q < n) { b2 = b + q; break; } q -= n; } }
  if (t < 256) {
    const int i = t >> 4, j = t & 15;
    if (b != b2 || i <= j) put_code(NCB * i + b, NCB * j + b2, (float)S[t]);
  }
}

template <int NCB, int NPOSE, bool BYVAL>
__global__ __launch_bounds__(1024) void k_sfm_finalize(const float* __restrict__ partials, const int bpp, const SfmPairDev* __restrict__ pairs, const SfmPairDev one,
                                                       char* __restrict__ items, const size_t item_stride, unsigned* __restrict__ qhead,
                                                       const int Wk, const int Hk, const unsigned launch_id, const int ragged, const DoneFlag done) {
  sfm_finalize_body<NCB, NPOSE, BYVAL>(partials, bpp, pairs, one, items, item_stride, qhead, Wk, Hk, launch_id, ragged);
  if (BYVAL) signal_done_grid(done, gridDim.x * gridDim.y);   // a blocking single-pair call: the host polls the flag behind the item (dfx_kernels.hpp)
}

template <int NCB, int NPOSE, bool BYVAL>
__global__ __launch_bounds__(1024) void k_sfm_finalize_b3(const float* __restrict__ partials, const int bpp, const SfmPairDev* __restrict__ pairs, const SfmPairDev one,
                                                          char* __restrict__ items, const size_t item_stride, unsigned* __restrict__ qhead,
                                                          const int Wk, const int Hk, const unsigned launch_id, const int ragged, const DoneFlag done) {
  constexpr int NT3 = b3_tiles(NCB);
  constexpr int ZDIM = b3_blocks(NCB) * 256;
  __shared__ double red[4][256];
  __shared__ double redn[4][256];   // the N part of (P,P) [and of the diagonal tiles (C_b,C_b) with the four-product diagonals]
  __shared__ double T[12][6];

  const int blk = blockIdx.x, pair = blockIdx.y;
  const int el = threadIdx.x & 255, rg = threadIdx.x >> 8;
  const SfmPairDev& PD = BYVAL ? one : pairs[pair];
  const int W = ragged ? (int)PD.w_px : Wk, H = ragged ? (int)PD.h_px : Hk;
  const int nparts = ragged ? (int)PD.nblk : bpp;
  const size_t part0 = ragged ? (size_t)PD.blk0 : (size_t)pair * bpp;
  if (NPOSE == 12) rebuild_valid0_shadow(PD, W, H, launch_id, blk, (int)gridDim.x, valid0_shadow_stamp(PD, launch_id));
  if (qhead && blk == 0 && threadIdx.x == 0) qhead[pair] = 0u;
  const float* src = partials + part0 * ZDIM + blk * 256 + el;
  red[rg][el] = strided_sum_f64<4, 16>(src, rg, nparts, ZDIM);
  const int dtile = b3_dtile<NCB>(blk);
  if (dtile >= 0) redn[rg][el] = strided_sum_f64<4, 16>(partials + part0 * ZDIM + (NT3 + dtile) * 256 + el, rg, nparts, ZDIM);
  if (NPOSE == 12 && threadIdx.x < 72) {
    const int n = threadIdx.x / 6, i = threadIdx.x - n * 6;
    T[n][i] = b3_T_entry(BYVAL ? one.M : pairs[pair].M, BYVAL ? one.HM : pairs[pair].HM, n, i);
  }
  __syncthreads();
  if (rg == 0) red[0][el] = ((red[0][el] + red[1][el]) + red[2][el]) + red[3][el];
  if (dtile >= 0 && rg == 1) redn[0][el] = ((redn[0][el] + redn[1][el]) + redn[2][el]) + redn[3][el];
  __syncthreads();
  {
    bool keep = false;
    double v = 0.0;
    if (rg == 0) v = b3_unpack<NCB>(blk, dtile, el, red[0], redn[0], keep);
    __syncthreads();
    if (keep) red[0][el] = v;
    __syncthreads();
  }
  float* item = reinterpret_cast<float*>(items + (size_t)pair * item_stride);
  b3_scatter<NCB, NPOSE>(blk, (int)threadIdx.x, red[0], T, item);
  if (BYVAL) signal_done_grid(done, gridDim.x * gridDim.y);
}

// ---- a SINGLE pair's finalize (the reference's per-factor pattern: one blocking RunStep per linearisation) on more than b3_tiles compute units --------------
// k_sfm_finalize_b3 gives a pair one workgroup per tile: 6 at CS = 32, each folding 240-480 KB of partial rows through ONE compute unit's 64 B / clock --
// 10 us behind a 14 us step kernel (rocprofv3 kernel trace of tests/cpp/gn_round_bench's serial pattern).  Here a tile has FOUR workgroups of 256 threads:
// workgroup (tile, g) sums the rows g, g + 4, ... (the row group `rg` of k_sfm_finalize_b3: the same chain of additions), parks its 256 (+ 256) doubles,
// and counts itself in; the workgroup that arrives last folds the four groups ((g0 + g1) + g2) + g3 -- the same bits as the one-workgroup form --, unpacks
// and scatters.  Arrival protocol of k_sfm_tail_b3 (stores complete, barrier, one release fence + one relaxed read-modify-write, acquire in the last arriver).
template <int NCB, int NPOSE>
__global__ __launch_bounds__(256) void k_sfm_finalize_b3_split(const float* __restrict__ partials, const int nparts, const SfmPairDev one, char* __restrict__ item_bytes,
                                                               const int W, const int H, const unsigned launch_id, double* __restrict__ scratch,
                                                               unsigned* __restrict__ tile_cnt, const DoneFlag done) {
  constexpr int NT3 = b3_tiles(NCB);
  constexpr int ZDIM = b3_blocks(NCB) * 256;
  __shared__ double red[256], redn[256];
  __shared__ double T[12][6];
  __shared__ int last_s;
  const int blk = blockIdx.x, g = blockIdx.y, el = threadIdx.x;
#if DFX_FIN_TRACE
  const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (NPOSE == 12) rebuild_valid0_shadow(one, W, H, launch_id, blk * 4 + g, (int)gridDim.x * 4, valid0_shadow_stamp(one, launch_id));
#if DFX_FIN_TRACE
  const unsigned long long tr1 = __builtin_amdgcn_s_memrealtime();
#endif
  const int dtile = b3_dtile<NCB>(blk);
  double* mine = scratch + ((size_t)blk * 4 + g) * 512;
  {
    double sa, sb;
    if (dtile >= 0) strided_sum2_f64<4, 32, true>(partials + blk * 256 + el, partials + (NT3 + dtile) * 256 + el, g, nparts, ZDIM, sa, sb);
    else strided_sum2_f64<4, 32, false>(partials + blk * 256 + el, nullptr, g, nparts, ZDIM, sa, sb);
    mine[el] = sa;
    if (dtile >= 0) mine[256 + el] = sb;
  }
  __builtin_amdgcn_s_waitcnt(0);   // this wave's stores have completed
  __syncthreads();
#if DFX_FIN_TRACE
  const unsigned long long tr2 = __builtin_amdgcn_s_memrealtime();
#endif
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const bool last = __hip_atomic_fetch_add(&tile_cnt[blk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 3u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(&tile_cnt[blk], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // rewound for the next call
    }
    last_s = last ? 1 : 0;
  }
  __syncthreads();
#if DFX_FIN_TRACE
  const unsigned long long tr3 = __builtin_amdgcn_s_memrealtime();
  if (!last_s) { if (threadIdx.x == 0) printf("fin blk %d g %d  shadow %llu fold %llu arrive %llu (x10 ns) start %llu\n", blk, g, tr1 - tr0, tr2 - tr1, tr3 - tr2, tr0 % 1000000ull); return; }
#endif
  if (!last_s) return;
  {
    const double* t0 = scratch + (size_t)blk * 4 * 512;
    auto ld = [](const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };   // written by other workgroups of this kernel
    red[el] = ((ld(t0 + el) + ld(t0 + 512 + el)) + ld(t0 + 1024 + el)) + ld(t0 + 1536 + el);
    if (dtile >= 0) redn[el] = ((ld(t0 + 256 + el) + ld(t0 + 768 + el)) + ld(t0 + 1280 + el)) + ld(t0 + 1792 + el);
  }
  if (NPOSE == 12 && threadIdx.x < 72) {
    const int n = threadIdx.x / 6, i = threadIdx.x - n * 6;
    T[n][i] = b3_T_entry(one.M, one.HM, n, i);
  }
  __syncthreads();
  {
    bool keep = false;
    const double v = b3_unpack<NCB>(blk, dtile, el, red, redn, keep);
    __syncthreads();
    if (keep) red[el] = v;
    __syncthreads();
  }
  b3_scatter<NCB, NPOSE>(blk, el, red, T, reinterpret_cast<float*>(item_bytes));
#if DFX_FIN_TRACE
  const unsigned long long tr4 = __builtin_amdgcn_s_memrealtime();
#endif
  signal_done_grid(done, gridDim.x);   // the last arrivers of the tiles count themselves in; the last of THEM tells the host
#if DFX_FIN_TRACE
  if (threadIdx.x == 0) printf("fin blk %d g %d LAST shadow %llu fold %llu arrive %llu gather+scatter %llu signal %llu (x10 ns) start %llu\n", blk, g, tr1 - tr0, tr2 - tr1, tr3 - tr2, tr4 - tr3,
                               __builtin_amdgcn_s_memrealtime() - tr4, tr0 % 1000000ull);
#endif
}

// ---- the reduction tail of a batched bf16-split launch in ONE kernel: workgroup p sums ALL blocks of pair p's partials (same order as
// k_sfm_finalize_b3: four interleaved groups of partials, folded ((g0 + g1) + g2) + g3 in double), writes the item, and -- when the
// launch assembles a keyframe graph (dfx_sfm_step_batch_assemble_async) -- the workgroup that completes a node (the last of the
// node's incident local pairs to arrive, counted in `node_cnt`) gathers that node's diagonal block and gradient exactly as
// k_graph_assemble does (ascending pair index, double): no second kernel, no launch boundary between finalize and assembly.
// grid = pairs [+ nodes: workgroups that zero the blocks of nodes / pairs this rank holds no item of], 1024 threads.
// Round 3, 128 pairs: k_sfm_finalize_b3 (768 workgroups, two rounds) 15.3 us + boundary + k_graph_assemble 6.5 us -> see DESIGN.md 3.7.

// `tbeg` / `nt`: the workgroup's threads tbeg .. tbeg + nt - 1 do the node (a pair that completes BOTH its nodes gives each half of the workgroup one of them: the two
// chains of dependent reads -- graph tables, then the incident pairs' items -- run side by side instead of one behind the other)
template <int CS>
__device__ __forceinline__ void tail_assemble_node(const TailGraphDev& tg, const char* __restrict__ items, size_t item_stride, int n, int tbeg, int nt) {
  constexpr int NP = 12 + CS, D = 6 + CS, NT = NP * (NP + 1) / 2;
  const GraphDev& G = tg.G;
  float* const Hd = tg.sys;
  float* const gv = tg.sys + (size_t)G.n_nodes * D * D + (size_t)G.n_pairs * D * 6;
  auto item_of = [&](int p) -> const float* {
    const int l = p - tg.first_pair;
    return (l >= 0 && l < tg.n_local) ? reinterpret_cast<const float*>(items + (size_t)l * item_stride) : nullptr;
  };
  auto tri = [](int a, int b) { const int lo = a < b ? a : b, hi = a < b ? b : a; return lo * NP - lo * (lo - 1) / 2 + (hi - lo); };
  // items of other workgroups: device-scope loads (they were stored device-scope and had completed before the arrival was counted)
  auto ld = [](const float* p) { return (double)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  const int k0 = G.kf_begin[n], k1 = G.kf_begin[n + 1];
  const int f0 = G.fr_begin[n], f1 = G.fr_begin[n + 1];
  if ((int)threadIdx.x < tbeg || (int)threadIdx.x >= tbeg + nt) return;
  for (int e = (int)threadIdx.x - tbeg; e < D * D + D; e += nt) {
    double acc = 0.0;
    if (e < D * D) {
      const int r = e / D, c = e - r * D;
      const int ia = r < 6 ? r : r + 6, ib = c < 6 ? c : c + 6;
      const int t0 = tri(ia, ib);
      for (int q = k0; q < k1; ++q) { const float* it = item_of(G.kf_pairs[q]); if (it) acc += ld(it + t0); }
      if (r < 6 && c < 6) {
        const int t1 = tri(6 + r, 6 + c);
        for (int q = f0; q < f1; ++q) { const float* it = item_of(G.fr_pairs[q]); if (it) acc += ld(it + t1); }
      }
      Hd[(size_t)n * D * D + e] = (float)acc;
    } else {
      const int r = e - D * D;
      const int ia = r < 6 ? r : r + 6;
      for (int q = k0; q < k1; ++q) { const float* it = item_of(G.kf_pairs[q]); if (it) acc += ld(it + NT + ia); }
      if (r < 6) for (int q = f0; q < f1; ++q) { const float* it = item_of(G.fr_pairs[q]); if (it) acc += ld(it + NT + 6 + r); }
      gv[(size_t)n * D + r] = (float)acc;
    }
  }
}

// number of node n's incident pairs this rank holds an item of (wave 0 of the workgroup; the result is valid in every lane of wave 0)
__device__ __forceinline__ int tail_local_degree(const TailGraphDev& tg, int n) {
  const GraphDev& G = tg.G;
  const int lane = threadIdx.x & 63;
  const int k0 = G.kf_begin[n], k1 = G.kf_begin[n + 1], f0 = G.fr_begin[n], f1 = G.fr_begin[n + 1];
  int cnt = 0;
  for (int q = k0 + lane; q < k1; q += 64) { const int l = G.kf_pairs[q] - tg.first_pair; cnt += (l >= 0 && l < tg.n_local) ? 1 : 0; }
  for (int q = f0 + lane; q < f1; q += 64) { const int l = G.fr_pairs[q] - tg.first_pair; cnt += (l >= 0 && l < tg.n_local) ? 1 : 0; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  return cnt;
}

template <int NCB, bool ASM, bool ORD = true>   // ASM: the launch assembles a graph (tg.sys != null); ORD: release / acquire on the arrival counter (below)
__global__ __launch_bounds__(1024) void k_sfm_tail_b3(const float* __restrict__ partials, const int bpp, const SfmPairDev* __restrict__ pairs, const int npairs,
                                                      char* __restrict__ items, const size_t item_stride, unsigned* __restrict__ qhead,
                                                      const int Wk, const int Hk, const unsigned launch_id, const int ragged, const TailGraphDev tg) {
  constexpr int CS = 16 * NCB, D = 6 + CS, NP = 12 + CS;
  constexpr int NT3 = b3_tiles(NCB), NB3 = b3_blocks(NCB);
  constexpr int ZDIM = NB3 * 256;
#ifndef DFX_TAIL_ROWS
#define DFX_TAIL_ROWS 0
#endif
  constexpr int ROWS = DFX_TAIL_ROWS ? DFX_TAIL_ROWS : (NB3 <= 10 ? 8 : 2);   // partial rows per batch: ROWS * NB3 loads in flight per thread (CS = 32: a pair's 30 partials in ONE batch per group of rows, 101 registers)
  constexpr int MINE = (NT3 + 3) / 4;                         // tiles per thread group in the unpack / scatter stage
  __shared__ double S[NB3][256];
  __shared__ double T[12][6];
  __shared__ int todo[2];
  const int el = threadIdx.x & 255, rg = threadIdx.x >> 8;

  if (ASM && (int)blockIdx.x >= npairs) {
    // node workgroups (launched only when this rank does not hold every pair, or the graph has isolated nodes): what no local pair
    // writes is zero -- the diagonal block and gradient of a node without local pairs, the off-diagonal block of every remote pair
    const int n = (int)blockIdx.x - npairs;
    if (threadIdx.x < 64) { const int d = tail_local_degree(tg, n); if (threadIdx.x == 0) todo[0] = d; }
    __syncthreads();
    float* const Hd = tg.sys;
    float* const Ho = tg.sys + (size_t)tg.G.n_nodes * D * D;
    float* const gv = Ho + (size_t)tg.G.n_pairs * D * 6;
    if (todo[0] == 0) {
      for (int e = threadIdx.x; e < D * D; e += 1024) Hd[(size_t)n * D * D + e] = 0.f;
      if (threadIdx.x < D) gv[(size_t)n * D + threadIdx.x] = 0.f;
    }
    for (int q = tg.G.kf_begin[n]; q < tg.G.kf_begin[n + 1]; ++q) {
      const int p = tg.G.kf_pairs[q], l = p - tg.first_pair;
      if ((l < 0 || l >= tg.n_local) && threadIdx.x < D * 6) Ho[(size_t)p * D * 6 + threadIdx.x] = 0.f;
    }
    return;
  }

  const int pair = blockIdx.x;
  const SfmPairDev& PD = pairs[pair];
  const int W = ragged ? (int)PD.w_px : Wk, H = ragged ? (int)PD.h_px : Hk;
  const int nparts = ragged ? (int)PD.nblk : bpp;
  const size_t part0 = ragged ? (size_t)PD.blk0 : (size_t)pair * bpp;
  const unsigned stamp = valid0_shadow_stamp(PD, launch_id);   // read now, used at the very end
  if (qhead && threadIdx.x == 0) qhead[pair] = 0u;
  // ---- sums of this thread's group of partials, all blocks at once
  double s[NB3];
#pragma unroll
  for (int a = 0; a < NB3; ++a) s[a] = 0.0;
  const float* src = partials + part0 * ZDIM + el;
  for (int b = rg; b < nparts; b += 4 * ROWS) {
    float v[ROWS][NB3];
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
      const int r = b + 4 * q;
      const float* row = src + (size_t)(r < nparts ? r : b) * ZDIM;   // unconditional loads of an existing row; +0.0 for the rows past the end
#pragma unroll
      for (int a = 0; a < NB3; ++a) { const float t = row[a * 256]; v[q][a] = r < nparts ? t : 0.0f; }
    }
#pragma unroll
    for (int q = 0; q < ROWS; ++q)
#pragma unroll
      for (int a = 0; a < NB3; ++a) s[a] += (double)v[q][a];
  }
  if (threadIdx.x < 72) {   // (behind the partial loads: its descriptor reads would otherwise be waited for first)
    const int n = threadIdx.x / 6, i = threadIdx.x - n * 6;
    T[n][i] = b3_T_entry(PD.M, PD.HM, n, i);
  }
#ifdef DFX_TAIL_STOP   // phase profile builds only (results are not valid): the kernel ends behind phase DFX_TAIL_STOP
  if (DFX_TAIL_STOP == 1) { double acc = 0.0; for (int a = 0; a < NB3; ++a) acc += s[a]; if (acc == 1.2345e300) items[0] = 1; return; }
#endif
  // ---- ((g0 + g1) + g2) + g3 through one LDS copy of the blocks
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    if (rg == k) {
#pragma unroll
      for (int a = 0; a < NB3; ++a) S[a][el] = s[a];
    }
    __syncthreads();
    if (rg == 0) {
#pragma unroll
      for (int a = 0; a < NB3; ++a) s[a] += S[a][el];
    }
    __syncthreads();
  }
  if (rg == 0) {
#pragma unroll
    for (int a = 0; a < NB3; ++a) S[a][el] = s[a];
  }
  __syncthreads();
  // ---- unpack: thread group rg owns the tiles rg, rg + 4, ...
  {
    double u[MINE];
    bool keep[MINE];
#pragma unroll
    for (int j = 0; j < MINE; ++j) {
      const int blk = rg + 4 * j;
      keep[j] = false; u[j] = 0.0;
      if (blk < NT3) { const int dt = b3_dtile<NCB>(blk); u[j] = b3_unpack<NCB>(blk, dt, el, S[blk], S[NT3 + (dt >= 0 ? dt : 0)], keep[j]); }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MINE; ++j) { const int blk = rg + 4 * j; if (blk < NT3 && keep[j]) S[blk][el] = u[j]; }
    __syncthreads();
  }
#ifdef DFX_TAIL_STOP
  if (DFX_TAIL_STOP == 2) { if (S[0][el] == 1.2345e300) items[0] = 1; return; }
#endif
  float* item = reinterpret_cast<float*>(items + (size_t)pair * item_stride);
#pragma unroll
  for (int j = 0; j < MINE; ++j) { const int blk = rg + 4 * j; if (blk < NT3) b3_scatter<NCB, 12, ASM>(blk, el, S[blk], T, item); }
#ifdef DFX_TAIL_STOP
  if (DFX_TAIL_STOP == 3) return;
#endif

  // ---- graph assembly by the last pair to arrive at each of its two nodes.
  // Hand-over protocol (round 5: ordered by the memory model, not by cache behaviour).  Writer: the item goes out in device-scope stores
  // (write-through), every wave waits for its stores to complete (s_waitcnt 0), the workgroup barrier collects the waves, and ONE lane counts
  // the arrivals behind ONE RELEASE fence at device scope -- the barrier makes the other waves' stores happen-before it, the fence makes them
  // visible at device scope (one L2 write-back per workgroup; rounds 3-4 measured a __threadfence per WAVE, 2048 per launch: 80 us, and
  // therefore ran the counter relaxed).  Reader: the workgroup whose arrival completes a node issues one ACQUIRE fence at device scope behind
  // its read-modify-write (it read the last link of the chain of arrivals, each behind its writer's release fence), the barrier hands the
  // order to the rest of the workgroup, and the items are read with device-scope loads.  ORD = false keeps the relaxed counter of rounds 3-4 for A/B
  // ran: same bits.
  if (ASM) {
    const int gp = tg.first_pair + pair;
    // (the nodes' local degrees -- three dependent reads of the constant graph tables -- in front of the wait for the stores: the two latencies overlap)
    int need[2] = { 0, 0 }, node[2] = { 0, 0 };
    if (threadIdx.x < 64) {
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        node[side] = tg.pair_nodes[2 * gp + side];
        need[side] = tail_local_degree(tg, node[side]);
      }
    }
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt = lgkmcnt = expcnt = 0: this wave's stores have completed
    __syncthreads();
    if (threadIdx.x >= 64) {   // off-diagonal block of this pair: single writer, from the item this workgroup has just written (device-scope loads of device-scope stores:
      // complete since the barrier above); it waits for nobody's arrival, so the waves that do not count arrivals copy it WHILE wave 0 does
      float* const Ho = tg.sys + (size_t)tg.G.n_nodes * D * D;
      auto tri = [](int a, int b) { const int lo = a < b ? a : b, hi = a < b ? b : a; return lo * NP - lo * (lo - 1) / 2 + (hi - lo); };
      for (int e = (int)threadIdx.x - 64; e < D * 6; e += 1024 - 64) {
        const int r = e / 6, c = e - r * 6;
        const int ia = r < 6 ? r : r + 6;
        Ho[(size_t)gp * D * 6 + e] = __hip_atomic_load(&item[tri(ia, 6 + c)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      if (lane == 0) {
        // ONE release fence in front of the two arrivals (fence-atomic synchronisation: the relaxed read-modify-writes behind it publish what the
        // barrier collected), ONE acquire fence behind them if either completed its node
        if (ORD) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        bool any = false;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
          const unsigned got = __hip_atomic_fetch_add(&tg.node_cnt[node[side]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
          const bool last = got == (unsigned)need[side];
          any = any || last;
          todo[side] = last ? node[side] : -1;
        }
        if (ORD && any) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
        for (int side = 0; side < 2; ++side)   // rewound for the next launch: nobody else counts on a completed node any more
          if (todo[side] >= 0) __hip_atomic_store(&tg.node_cnt[todo[side]], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
#ifdef DFX_TAIL_STOP
    if (DFX_TAIL_STOP == 4) return;
#endif
    if (todo[0] >= 0 && todo[1] >= 0) {
      tail_assemble_node<CS>(tg, items, item_stride, todo[0], 0, 512);
      tail_assemble_node<CS>(tg, items, item_stride, todo[1], 512, 512);
    } else if (todo[0] >= 0) tail_assemble_node<CS>(tg, items, item_stride, todo[0], 0, 1024);
    else if (todo[1] >= 0) tail_assemble_node<CS>(tg, items, item_stride, todo[1], 0, 1024);
  }
  // ---- the pair's valid0 shadow, when a wave of this launch changed the map (never in the steady state)
  rebuild_valid0_shadow(PD, W, H, launch_id, 0, 1, stamp);
}

size_t sfm_step_partials_bytes(int cs, int npairs, int blocks_per_pair) {
  // one size for both evaluation modes: the fp32 chain's packed z-space or the bf16 split's tiles (more blocks only with DFX_B3_DIAG4)
  const int zc = sfm_zdim(cs / 16), zb = b3_blocks(cs / 16) * 256;
  return (size_t)npairs * blocks_per_pair * (zc > zb ? zc : zb) * sizeof(float);
}

template <int NCB, int MODE>
static hipError_t launch_t(const SfmPairDev* pairs_dev, int npairs, int W, int H, const SfmParamsDev& prm, int bpp,
                           float* partials_dev, void* items_dev, size_t item_stride, hipStream_t stream, bool jac_dense, int prec,
                           hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr, const SfmPairDev* one_host = nullptr,
                           const DynDev* dyn = nullptr, int dyn_grid = 0, bool vsh = false, hipStream_t fin_stream = nullptr, hipEvent_t ev_mid = nullptr,
                           const unsigned* blkmap = nullptr, int total_blocks = 0, const TailGraphDev* tail_graph = nullptr, int node_wgs = 0,
                           bool* assembled = nullptr, DoneFlag* done_io = nullptr, double* split_scratch = nullptr, unsigned* split_cnt = nullptr) {
  if (assembled) *assembled = false;
  // only the by-value (single pair) finalize kernels signal; a launch that takes another tail clears the caller's flag, and the caller waits for the stream
  const DoneFlag done = (done_io && one_host && !dyn) ? *done_io : DoneFlag{};
  if (done_io && !done.flag) done_io->flag = nullptr;
  const TailGraphDev tg = tail_graph ? *tail_graph : TailGraphDev{};
  constexpr int NACC = NCB * (NCB - 1) / 2 + NCB + 2 * ((NCB + 1) / 2);   // 256-float blocks after block 0 (16x16x4 and 4x4x1 accumulators)
  hipError_t e;
  if (ev_begin && (e = hipEventRecord(ev_begin, stream)) != hipSuccess) return e;
  const bool ragged_l = blkmap != nullptr;
  const dim3 grid = ragged_l ? dim3(total_blocks) : dim3(bpp, npairs);
  const dim3 block(kThreads);
  const int ragged = ragged_l ? 1 : 0;
  const bool b3 = prec == 1;   // DFX_MFMA_BF16X3 (include/dfx.h): exact three-way bf16 split on v_mfma_f32_16x16x32_bf16; 0: the fp32 chain
  if (!ragged_l && ((long long)W * H) % 64 != 0) jac_dense = false;   // ragged last chunk: the per-vector addressing clamps pixels past the image (pairs of several sizes: checked by the caller)
  // the ray table rides in dynamic LDS when it fits beside the static arrays (64 KB per workgroup); MODE 1 has no table
  constexpr int kZMax = (1 + NACC) * 256 > b3_blocks(NCB) * 256 ? (1 + NACC) * 256 : b3_blocks(NCB) * 256;   // either evaluation mode
  constexpr size_t kStaticLds = sizeof(float) * (size_t)kWaves * ((kUFloats > kZMax) ? kUFloats : kZMax);
  const size_t tab_bytes = sizeof(float) * ((size_t)W + H + kRayTabSlack);
  const bool tab_lds = MODE == 0 && kStaticLds + tab_bytes + DFX_EXTRA_LDS <= 64 * 1024;
  const size_t dyn_lds = DFX_EXTRA_LDS + (tab_lds ? tab_bytes : 0);
  const bool byval = one_host != nullptr && npairs == 1;
  // one workgroup per pair (k_sfm_tail_b3) or one per tile of a pair (k_sfm_finalize_b3)?  bpp = the partials of the largest pair
  const long long tail_kb = (long long)((dyn && dyn->qhead) ? dyn->team : bpp) * b3_blocks(NCB);
  const bool use_tail = DFX_TAIL_KERNEL && MODE == 0 && tail_kb <= DFX_TAIL_MAX_KB;
  const SfmPairDev one = byval ? *one_host : SfmPairDev{};
  const DynDev nodyn{ nullptr, 0, 0, 0, 0, 0u };
  // deferred tail: the finalize kernel runs on `fin_stream`, behind an event recorded after the step kernel
  hipStream_t fstream = stream;
  auto to_fin_stream = [&]() -> hipError_t {
    if (!fin_stream || fin_stream == stream || !ev_mid) return hipSuccess;
    hipError_t e2 = hipEventRecord(ev_mid, stream);
    if (e2 == hipSuccess) e2 = hipStreamWaitEvent(fin_stream, ev_mid, 0);
    fstream = fin_stream;
    return e2;
  };
  if (dyn && dyn->qhead) {
    // dynamic schedule: a resident grid of wave-workers (see k_sfm_step); partials = [pair][team member]
    if constexpr (MODE == 0) {
      const size_t dlds = sizeof(float) * (size_t)kWaves * ((size_t)W + H + kRayTabSlack);
#define DFX_LAUNCH_DYN(B3_, VSH_) hipLaunchKernelGGL((k_sfm_step<NCB, 0, true, true, false, true, B3_, VSH_>), dim3(dyn_grid), block, dlds, stream, pairs_dev, one, prm, W, H, partials_dev, *dyn, (const unsigned*)nullptr)
      if (b3) { if (vsh) DFX_LAUNCH_DYN(true, true); else DFX_LAUNCH_DYN(true, false); }
      else { if (vsh) DFX_LAUNCH_DYN(false, true); else DFX_LAUNCH_DYN(false, false); }
#undef DFX_LAUNCH_DYN
      e = hipGetLastError();
      if (e != hipSuccess) return e;
      if (ev_end && (e = hipEventRecord(ev_end, stream)) != hipSuccess) return e;
      if ((e = to_fin_stream()) != hipSuccess) return e;
      if (b3 && use_tail) {
        if (tg.sys) hipLaunchKernelGGL((k_sfm_tail_b3<NCB, true, true>), dim3(npairs + node_wgs), dim3(1024), 0, fstream,
                                       (const float*)partials_dev, dyn->team, pairs_dev, npairs, (char*)items_dev, item_stride, dyn->qhead, W, H, prm.launch_id, 0, tg);
        else hipLaunchKernelGGL((k_sfm_tail_b3<NCB, false>), dim3(npairs), dim3(1024), 0, fstream,
                                (const float*)partials_dev, dyn->team, pairs_dev, npairs, (char*)items_dev, item_stride, dyn->qhead, W, H, prm.launch_id, 0, tg);
        if (assembled) *assembled = tg.sys != nullptr;
      } else if (b3) {
        hipLaunchKernelGGL((k_sfm_finalize_b3<NCB, 12, false>), dim3(b3_tiles(NCB), npairs), dim3(1024), 0, fstream,
                           (const float*)partials_dev, dyn->team, pairs_dev, one, (char*)items_dev, item_stride, dyn->qhead, W, H, prm.launch_id, 0, DoneFlag{});
      } else hipLaunchKernelGGL((k_sfm_finalize<NCB, 12, false>), dim3(1 + NACC, npairs), dim3(1024), 0, fstream,
                              (const float*)partials_dev, dyn->team, pairs_dev, one, (char*)items_dev, item_stride, dyn->qhead, W, H, prm.launch_id, 0, DoneFlag{});
      return hipGetLastError();
    }
  }
#define DFX_LAUNCH_STEP__(M_, JD_, TL_, B3_, VSH_)                                                                                                  \
  do {                                                                                                                                             \
    if (byval) hipLaunchKernelGGL((k_sfm_step<NCB, M_, JD_, TL_, true, false, B3_, VSH_>), grid, block, dyn_lds, stream, (const SfmPairDev*)nullptr, one, prm, W, H, partials_dev, nodyn, (const unsigned*)nullptr); \
    else hipLaunchKernelGGL((k_sfm_step<NCB, M_, JD_, TL_, false, false, B3_, VSH_>), grid, block, dyn_lds, stream, pairs_dev, one, prm, W, H, partials_dev, nodyn, blkmap);              \
  } while (0)
#define DFX_LAUNCH_STEP_(M_, JD_, TL_, B3_) do { if (M_ == 0 && vsh) DFX_LAUNCH_STEP__(M_, JD_, TL_, B3_, (M_ == 0)); else DFX_LAUNCH_STEP__(M_, JD_, TL_, B3_, false); } while (0)
#define DFX_LAUNCH_STEP(M_, JD_, TL_) do { if (b3) DFX_LAUNCH_STEP_(M_, JD_, TL_, true); else DFX_LAUNCH_STEP_(M_, JD_, TL_, false); } while (0)
  if (MODE == 0 && tab_lds) {
    if (jac_dense) DFX_LAUNCH_STEP(0, true, true); else DFX_LAUNCH_STEP(0, false, true);
  } else {
    if (jac_dense) DFX_LAUNCH_STEP(MODE, true, false); else DFX_LAUNCH_STEP(MODE, false, false);
  }
#undef DFX_LAUNCH_STEP
#undef DFX_LAUNCH_STEP_
#undef DFX_LAUNCH_STEP__
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (ev_end && (e = hipEventRecord(ev_end, stream)) != hipSuccess) return e;
  if ((e = to_fin_stream()) != hipSuccess) return e;
  constexpr int NPOSE = MODE == 0 ? 12 : 0;
  const SfmPairDev* fpairs = byval ? (const SfmPairDev*)nullptr : pairs_dev;
  if (b3) {
    if (byval && split_scratch && !ragged) hipLaunchKernelGGL((k_sfm_finalize_b3_split<NCB, NPOSE>), dim3(b3_tiles(NCB), 4), dim3(256), 0, fstream,
                                  (const float*)partials_dev, bpp, one, (char*)items_dev, W, H, prm.launch_id, split_scratch, split_cnt, done);
    else if (byval) hipLaunchKernelGGL((k_sfm_finalize_b3<NCB, NPOSE, true>), dim3(b3_tiles(NCB), npairs), dim3(1024), 0, fstream,
                                  (const float*)partials_dev, bpp, fpairs, one, (char*)items_dev, item_stride, (unsigned*)nullptr, W, H, prm.launch_id, ragged, done);
    else if (MODE == 0 && use_tail) {   // batched launches: one workgroup per pair, the graph assembly folded in
      if (tg.sys) hipLaunchKernelGGL((k_sfm_tail_b3<NCB, true, true>), dim3(npairs + node_wgs), dim3(1024), 0, fstream,
                                     (const float*)partials_dev, bpp, fpairs, npairs, (char*)items_dev, item_stride, (unsigned*)nullptr, W, H, prm.launch_id, ragged, tg);
      else hipLaunchKernelGGL((k_sfm_tail_b3<NCB, false>), dim3(npairs), dim3(1024), 0, fstream,
                              (const float*)partials_dev, bpp, fpairs, npairs, (char*)items_dev, item_stride, (unsigned*)nullptr, W, H, prm.launch_id, ragged, tg);
      if (assembled) *assembled = tg.sys != nullptr;
    } else hipLaunchKernelGGL((k_sfm_finalize_b3<NCB, NPOSE, false>), dim3(b3_tiles(NCB), npairs), dim3(1024), 0, fstream,
                              (const float*)partials_dev, bpp, fpairs, one, (char*)items_dev, item_stride, (unsigned*)nullptr, W, H, prm.launch_id, ragged, DoneFlag{});
  } else {
    if (byval) hipLaunchKernelGGL((k_sfm_finalize<NCB, NPOSE, true>), dim3(1 + NACC, npairs), dim3(1024), 0, fstream,
                                  (const float*)partials_dev, bpp, fpairs, one, (char*)items_dev, item_stride, (unsigned*)nullptr, W, H, prm.launch_id, ragged, done);
    else hipLaunchKernelGGL((k_sfm_finalize<NCB, NPOSE, false>), dim3(1 + NACC, npairs), dim3(1024), 0, fstream,
                            (const float*)partials_dev, bpp, fpairs, one, (char*)items_dev, item_stride, (unsigned*)nullptr, W, H, prm.launch_id, ragged, DoneFlag{});
  }
  return hipGetLastError();
}

hipError_t launch_sfm_step(int cs, const SfmPairDev* pairs_dev, int npairs, int W, int H, const SfmParamsDev& prm,
                           int blocks_per_pair, float* partials_dev, void* items_dev, size_t item_stride,
                           hipStream_t stream, bool jac_dense, int prec, hipEvent_t eb, hipEvent_t ee, const SfmPairDev* one_host,
                           const DynDev* dyn, int dyn_grid, bool vsh, hipStream_t fin_stream, hipEvent_t ev_mid, const unsigned* blkmap_dev, int total_blocks,
                           const TailGraphDev* tail_graph, int node_wgs, bool* assembled, DoneFlag* done, double* split_scratch, unsigned* split_cnt) {
  switch (cs) {
    case 16: return launch_t<1, 0>(pairs_dev, npairs, W, H, prm, blocks_per_pair, partials_dev, items_dev, item_stride, stream, jac_dense, prec, eb, ee, one_host, dyn, dyn_grid, vsh, fin_stream, ev_mid, blkmap_dev, total_blocks, tail_graph, node_wgs, assembled, done, split_scratch, split_cnt);
    case 32: return launch_t<2, 0>(pairs_dev, npairs, W, H, prm, blocks_per_pair, partials_dev, items_dev, item_stride, stream, jac_dense, prec, eb, ee, one_host, dyn, dyn_grid, vsh, fin_stream, ev_mid, blkmap_dev, total_blocks, tail_graph, node_wgs, assembled, done, split_scratch, split_cnt);
    case 64: return launch_t<4, 0>(pairs_dev, npairs, W, H, prm, blocks_per_pair, partials_dev, items_dev, item_stride, stream, jac_dense, prec, eb, ee, one_host, dyn, dyn_grid, vsh, fin_stream, ev_mid, blkmap_dev, total_blocks, tail_graph, node_wgs, assembled, done, split_scratch, split_cnt);
    default: return hipErrorInvalidValue;
  }
}

// DepthAligner::RunStep: `pair_host` describes ONE pseudo-pair with img0 = target depth, dpt0 = current depth, jac (passed by value).
hipError_t launch_depth_aligner_step(int cs, const SfmPairDev* pair_host, int W, int H, float avg_dpt, int blocks,
                                     float* partials_dev, void* item_dev, hipStream_t stream, bool jac_dense, int prec, DoneFlag* done, double* split_scratch, unsigned* split_cnt) {
  SfmParamsDev prm{ 0.f, avg_dpt, 0.f, 0.f, 0u };
  switch (cs) {
    case 16: return launch_t<1, 1>(nullptr, 1, W, H, prm, blocks, partials_dev, item_dev, 0, stream, jac_dense, prec, nullptr, nullptr, pair_host, nullptr, 0, false, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, done, split_scratch, split_cnt);
    case 32: return launch_t<2, 1>(nullptr, 1, W, H, prm, blocks, partials_dev, item_dev, 0, stream, jac_dense, prec, nullptr, nullptr, pair_host, nullptr, 0, false, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, done, split_scratch, split_cnt);
    case 64: return launch_t<4, 1>(nullptr, 1, W, H, prm, blocks, partials_dev, item_dev, 0, stream, jac_dense, prec, nullptr, nullptr, pair_host, nullptr, 0, false, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, done, split_scratch, split_cnt);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace dfx
