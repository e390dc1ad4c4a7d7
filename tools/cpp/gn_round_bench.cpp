// gn_round_bench.cpp -- one relinearisation round of a keyframe window from C++ (include/dfx_host.hpp), the three ways a mapper can ask for it:
//   serial cold    PhotometricFactor::linearize factor by factor, every call a blocking UpdateDepth + RunStep  (the reference's pattern under iSAM2:
//                  core/gtsam/photometric_factor.cpp:86-181, 225-293 -- what a pure header-swap build delivers)
//   batched        dfx::LinearizeAll: ONE decoder launch over the distinct keyframes + ONE batched step per pyramid level, every factor's cache seeded
//   serial warmed  LinearizeAll first, then the same serial linearize() calls (they hit their caches: iSAM2's call pattern, no launches)
// each followed by the slicing into G11..G33 / g1..g3 / f (Hessian()).  BASELINE configs[2]: 16 keyframes of 640x480, code size 32, all 120 pairs i < j.
// Prints one line per variant: median milliseconds per round and per factor.  Built by tests/cpp/Makefile, run by bench.py when present.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <memory>
#include <vector>

#include "../../include/dfx_host.hpp"

constexpr int CS = 32;

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
  const int K = argc > 1 ? std::atoi(argv[1]) : 16, REPS = argc > 2 ? std::atoi(argv[2]) : 7;
  try {
    const std::size_t W = 640, H = 480;
    auto ctx = dfx::Context::Default();
    std::vector<std::shared_ptr<dfx::Keyframe<CS>>> kfs;
    std::vector<float> img(W * H), prx(W * H), sd(W * H, 0.0f), jac(W * H * CS);
    for (int k = 0; k < K; ++k) {
      auto kf = std::make_shared<dfx::Keyframe<CS>>(1, W, H, ctx);
      kf->id = (std::size_t)k;
      for (std::size_t y = 0; y < H; ++y) for (std::size_t x = 0; x < W; ++x) {
        const double u = (double)x + 1.5 * k, v = (double)y - 0.8 * k;
        img[y * W + x] = (float)(0.5 + 0.2 * std::sin(0.081 * u + 0.047 * v) + 0.15 * std::sin(0.033 * u - 0.112 * v + 1.0) + 0.1 * std::sin(0.15 * u + 0.09 * v + 2.0));
        prx[y * W + x] = (float)(2.0 / (2.0 + 2.5 + 0.3 * ((double)x / W - 0.5) - 0.3 * ((double)y / H - 0.5) + 0.05 * k));
      }
      for (std::size_t i = 0; i < jac.size(); ++i) jac[i] = 0.004f * (float)std::sin(0.37 * (double)(i % 9973) + k);
      kf->FillPyramids(img.data(), 1);
      kf->SetDecoderOutputs(0, prx.data(), sd.data(), jac.data());
      for (int c = 0; c < CS; ++c) kf->code[(std::size_t)c] = 0.2f * (float)std::sin(0.9 * c + k);
      kf->UpdateDepthMaps(2.0f, false);
      kfs.push_back(kf);
    }
    const dfx_cam cam{ 554.256f, 579.411f, 320.f, 240.f, (float)W, (float)H };
    std::vector<dfx_se3> pose((std::size_t)K);
    for (int k = 0; k < K; ++k) { const float a = 0.002f * k; pose[(std::size_t)k] = dfx_se3{ { 0, std::sin(a / 2), 0, std::cos(a / 2) }, { 0.004f * k, -0.002f * k, 0.001f * k } }; }
    df::SfmAligner<float, CS> aligner;
    std::vector<std::unique_ptr<dfx::PhotometricFactor<CS>>> own;
    std::vector<dfx::PhotometricFactor<CS>*> factors;
    std::vector<dfx::FactorValues<CS>> values;
    for (int i = 0; i < K; ++i) for (int j = i + 1; j < K; ++j) {
      own.emplace_back(new dfx::PhotometricFactor<CS>(cam, kfs[(std::size_t)i], kfs[(std::size_t)j], 0));
      factors.push_back(own.back().get());
      values.push_back(dfx::FactorValues<CS>{ pose[(std::size_t)i], pose[(std::size_t)j], kfs[(std::size_t)i]->code });
    }
    const int n = (int)factors.size();
    double sink = 0;
    auto moved = [&](int rep) {   // every round sees every pose moved (well above the 1e-6 relinearisation threshold): nothing is cached from the round before
      auto v = values;
      for (auto& q : v) { q.pose0.t[0] += 1e-4f * (rep + 1); q.pose1.t[0] += 1e-4f * (rep + 1); }
      return v;
    };
    std::vector<double> cold, batched, warmed;
    for (int rep = 0; rep < REPS + 2; ++rep) {
      auto v = moved(3 * rep);
      double t0 = now_ms();
      for (int k = 0; k < n; ++k) { factors[(std::size_t)k]->GetJacobiansIfNeeded(aligner, v[(std::size_t)k].pose0, v[(std::size_t)k].pose1, v[(std::size_t)k].code0); sink += factors[(std::size_t)k]->Hessian().f; }
      const double c = now_ms() - t0;
      v = moved(3 * rep + 1);
      t0 = now_ms();
      const int done = dfx::LinearizeAll(aligner, factors, v);
      const double b = now_ms() - t0;
      for (int k = 0; k < n; ++k) sink += factors[(std::size_t)k]->Hessian().f;
      v = moved(3 * rep + 2);
      t0 = now_ms();
      dfx::LinearizeAll(aligner, factors, v);
      for (int k = 0; k < n; ++k) { factors[(std::size_t)k]->GetJacobiansIfNeeded(aligner, v[(std::size_t)k].pose0, v[(std::size_t)k].pose1, v[(std::size_t)k].code0); sink += factors[(std::size_t)k]->Hessian().f; }
      const double w = now_ms() - t0;
      if (done != n) { std::printf("LinearizeAll relinearised %d of %d factors\n", done, n); return 1; }
      if (rep >= 2) { cold.push_back(c); batched.push_back(b); warmed.push_back(w); }
    }
    std::printf("gn_round_bench keyframes %d factors %d (640x480, cs %d), median of %d rounds\n", K, n, CS, REPS);
    std::printf("serial_cold_ms %.3f  per_factor_us %.1f\n", median(cold), median(cold) / n * 1e3);
    std::printf("batched_ms %.3f  per_factor_us %.1f\n", median(batched), median(batched) / n * 1e3);
    std::printf("serial_warmed_ms %.3f  per_factor_us %.1f\n", median(warmed), median(warmed) / n * 1e3);
    if (!std::isfinite(sink)) std::printf("(sink %g)\n", sink);
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 1;
  }
  return 0;
}
