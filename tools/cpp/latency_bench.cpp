// latency_bench.cpp -- blocking-call latency of the drop-in C++ interface (include/dfx_shim.hpp), measured from C++ as a
// reference call site sees it: PhotometricFactor::RunAlignmentStep calls SfmAligner::RunStep once per linearisation
// (photometric_factor.cpp:267-274), CameraTracker::TrackFrame calls SE3Aligner::RunStep once per iteration
// (camera_tracker.cpp:52-58).  Prints one line per operator: mean / min microseconds over N blocking calls at 640x480.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/dfx_shim.hpp"

using dfx::pod::Grad2;
using dfx::pod::Image2DView;
using dfx::pod::PinholeCamera;
using dfx::pod::SE3f;

// Per operator: the two ways a blocking call can wait (DFX_WAIT_POLL, the default, and DFX_WAIT_STREAM = hipStreamSynchronize), INTERLEAVED in blocks of n / 4
// calls so that both see the same box in the same state; mean / min microseconds over n calls each.
template <typename F>
static void timeit(const char* name, int n, F&& f) {
  auto ctx = dfx::Context::Default();
  double tot[2] = { 0, 0 }, mn[2] = { 1e30, 1e30 };
  for (int block = 0; block < 8; ++block) {
    const int mode = block & 1;   // 0: poll, 1: stream
    ctx->SetResultWait(mode == 0 ? DFX_WAIT_POLL : DFX_WAIT_STREAM);
    for (int i = 0; i < 5; ++i) f();
    for (int i = 0; i < n / 4; ++i) {
      const auto t0 = std::chrono::steady_clock::now();
      f();
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      tot[mode] += us; mn[mode] = us < mn[mode] ? us : mn[mode];
    }
  }
  ctx->SetResultWait(DFX_WAIT_POLL);
  const int m = (n / 4) * 4;
  std::printf("%-44s mean %7.1f us   min %7.1f us   (%d blocking calls; waiting for the stream instead: mean %7.1f  min %7.1f)\n", name, tot[0] / m, mn[0], m, tot[1] / m, mn[1]);
}

int main() {
  const int W = 640, H = 480, CS = 32;
  const PinholeCamera cam{ 554.256f, 579.411f, 320.f, 240.f, (float)W, (float)H };
  std::vector<float> img((size_t)W * H), dpt((size_t)W * H), jac((size_t)W * H * CS);
  for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
    img[(size_t)y * W + x] = 0.5f + 0.3f * std::sin(0.05f * x) * std::cos(0.04f * y);
    dpt[(size_t)y * W + x] = 2.5f + 0.2f * std::sin(0.01f * x + 0.02f * y);
  }
  for (size_t i = 0; i < jac.size(); ++i) jac[i] = 1e-3f * (float)((i * 2654435761u) % 1000) / 1000.f;
  float *d_img0, *d_img1, *d_dpt, *d_jac, *d_valid, *d_out;
  Grad2* d_grad;
  if (hipMalloc(&d_img0, img.size() * 4) != hipSuccess) { std::printf("no HIP device\n"); return 3; }
  (void)hipMalloc(&d_img1, img.size() * 4); (void)hipMalloc(&d_dpt, img.size() * 4); (void)hipMalloc(&d_valid, img.size() * 4);
  (void)hipMalloc(&d_out, img.size() * 4); (void)hipMalloc(&d_jac, jac.size() * 4); (void)hipMalloc(&d_grad, img.size() * 8);
  (void)hipMemcpy(d_img0, img.data(), img.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_img1, img.data(), img.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_dpt, dpt.data(), img.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_jac, jac.data(), jac.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(d_valid, 0, img.size() * 4);
  const Image2DView<float> v0{ d_img0, (size_t)W * 4, (size_t)W, (size_t)H }, v1{ d_img1, (size_t)W * 4, (size_t)W, (size_t)H };
  const Image2DView<float> vd{ d_dpt, (size_t)W * 4, (size_t)W, (size_t)H };
  Image2DView<float> vv{ d_valid, (size_t)W * 4, (size_t)W, (size_t)H };
  Image2DView<float> vo{ d_out, (size_t)W * 4, (size_t)W, (size_t)H };
  const Image2DView<float> vj{ d_jac, (size_t)W * CS * 4, (size_t)W * CS, (size_t)H };
  Image2DView<Grad2> vg{ d_grad, (size_t)W * 8, (size_t)W, (size_t)H };
  try {
    df::SobelGradients(v1, vg);
    SE3f p0, p1;
    p1.t[0] = 0.01f;
    df::SE3Aligner<float> se3;
    timeit("SE3Aligner::RunStep 640x480", 300, [&] { (void)se3.RunStep(p1, cam, v0, v1, vd, vg); });
    timeit("SE3Aligner::Warp 640x480", 300, [&] { (void)se3.Warp(p1, cam, v0, v1, vd, vo); });
    df::SfmAligner<float, CS> sfm;
    float code[CS] = { 0 };
    timeit("SfmAligner<32>::RunStep 640x480", 300, [&] { (void)sfm.RunStep(p0, p1, code, cam, v0, v1, vd, vd, vv, vj, vg); });
    if (std::getenv("DFX_LATENCY_SWEEP")) {   // workgroups of the single pair's step kernel (0 = the library's choice)
      for (int blocks : { 0, 160, 200, 240, 300, 400, 480, 600, 800 }) {
        sfm.SetStepThreadsBlocks(256, blocks);
        char name[96];
        std::snprintf(name, sizeof name, "SfmAligner<32>::RunStep, step_blocks %d", blocks);
        timeit(name, 200, [&] { (void)sfm.RunStep(p0, p1, code, cam, v0, v1, vd, vd, vv, vj, vg); });
      }
      sfm.SetStepThreadsBlocks(256, 0);
    }
    {
      std::vector<dfx_sfm_pair> batch(16, df::SfmAligner<float, CS>::MakePair(p0, p1, cam, v0, v1, vd, vv, vj, vg));
      const auto items = sfm.RunStepBatch(batch);
      const auto one = sfm.RunStep(p0, p1, code, cam, v0, v1, vd, vd, vv, vj, vg);
      if (items.size() != 16 || items[7].inliers != one.inliers) { std::printf("RunStepBatch mismatch\n"); return 1; }
      timeit("SfmAligner<32>::RunStepBatch, 16 x the same pair", 100, [&] { (void)sfm.RunStepBatch(batch); });
    }
    timeit("SfmAligner<32>::EvaluateError 640x480", 300, [&] { (void)sfm.EvaluateError(p0, p1, cam, v0, v1, vd, vd, vg); });
    timeit("UpdateDepth<32> 640x480", 300, [&] { df::UpdateDepth<float, CS>(code, vd, vj, 2.0f, vo); });
    timeit("SobelGradients 640x480", 300, [&] { df::SobelGradients(v1, vg); });
    {
      // CameraTracker::TrackFrame as ONE call (dfx_track_frame: 3 levels, 10 / 5 / 5 iterations from the coarsest -- BASELINE.json configs[0]'s schedule): the
      // live image is the keyframe image moved by two pixels, levels 1-2 from dfx_build_pyramid (depth: the level-0 depth's top-left quarter, a timing stand-in)
      dfx_ctx* c = dfx::Context::Default()->get();
      std::vector<float> moved((size_t)W * H);
      for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) moved[(size_t)y * W + x] = img[(size_t)y * W + (x + 2 < W ? x + 2 : W - 1)];
      (void)hipMemcpy(d_img1, moved.data(), moved.size() * 4, hipMemcpyHostToDevice);
      dfx_pyramid py0{}, py1{};
      py0.levels = py1.levels = 3;
      dfx_track_level lv[3];
      for (int l = 0; l < 3; ++l) {
        const uint32_t w = W >> l, h = H >> l;
        float *a = d_img0, *b = d_img1; Grad2* g = d_grad;
        if (l) { (void)hipMalloc(&a, (size_t)w * h * 4); (void)hipMalloc(&b, (size_t)w * h * 4); }
        if (l) (void)hipMalloc(&g, (size_t)w * h * 8);
        py0.img[l] = dfx_img{ a, (size_t)w * 4, w, h }; py1.img[l] = dfx_img{ b, (size_t)w * 4, w, h };
        py0.grad[l] = dfx_img{ nullptr, 0, 0, 0 }; py1.grad[l] = dfx_img{ g, (size_t)w * 8, w, h };
        const float sc = 1.f / (float)(1 << l);
        lv[l].cam = dfx_cam{ cam.fx() * sc, cam.fy() * sc, (cam.u0() + 0.5f) * sc - 0.5f, (cam.v0() + 0.5f) * sc - 0.5f, (float)w, (float)h };
        lv[l].img0 = py0.img[l]; lv[l].img1 = py1.img[l]; lv[l].grad1 = py1.grad[l];
        lv[l].dpt0 = dfx_img{ d_dpt, (size_t)W * 4, w, h };
        lv[l].iterations = l == 2 ? 10 : 5;
      }
      if (dfx_build_pyramid(c, &py0) != DFX_OK || dfx_build_pyramid(c, &py1) != DFX_OK) { std::printf("dfx_build_pyramid: %s\n", dfx_last_error()); return 1; }
      const dfx_se3 init{ { 0, 0, 0, 1 }, { 0, 0, 0 } };
      dfx_track_result res{};
      if (dfx_track_frame(c, &init, lv, 3, 0.1f, &res) != DFX_OK) { std::printf("dfx_track_frame: %s\n", dfx_last_error()); return 1; }
      std::printf("dfx_track_frame: %d iterations, %d failed solves, inliers %.3f, t = (%.4f %.4f %.4f)\n", res.iterations, res.solver_failures, res.inliers_frac, res.pose_ck.t[0],
                  res.pose_ck.t[1], res.pose_ck.t[2]);
      timeit("dfx_track_frame 640x480, 3 levels, 20 iterations", 300, [&] { (void)dfx_track_frame(c, &init, lv, 3, 0.1f, &res); });
    }
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 1;
  }
  return 0;
}
