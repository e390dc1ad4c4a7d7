// gn_round_bench.cpp -- one relinearisation round of a keyframe window from C++ (include/dfx_host.hpp), the three ways a mapper can ask for it:
//   serial cold    PhotometricFactor::linearize factor by factor, every call a blocking UpdateDepth + RunStep  (the reference's pattern under iSAM2:
//                  core/gtsam/photometric_factor.cpp:86-181, 225-293 -- what a pure header-swap build delivers)
//   batched        dfx::LinearizeAll: ONE decoder launch over the distinct keyframes + ONE batched step per pyramid level, every factor's cache seeded
//   serial warmed  LinearizeAll first, then the same serial linearize() calls (they hit their caches: iSAM2's call pattern, no launches)
// each followed by the slicing into G11..G33 / g1..g3 / f (Hessian()).  BASELINE configs[2]: 16 keyframes of 640x480, code size 32, all 120 pairs i < j.
//   geometric      the sparse geometric half of the round (SparseGeometricFactor::linearize, geo_npoints = 500 per factor): ONE launch over all factors with the
//                  rows left on the device / fetched with one copy (dfx::SparseGeometricLinearizeAll), and factor by factor (one blocking call each)
// usage: gn_round_bench [keyframes = 16] [reps = 7] [neighbours = 0]   neighbours = 0: all pairs i < j (configs[2]: 16 -> 120 factors); neighbours = n: every
//        keyframe linked to its n nearest by index (configs[3]: 64 keyframes, 16 neighbours -> 1024 factors)
// Prints one line per variant: median milliseconds per round and per factor.  Built by tests/cpp/Makefile, run by bench.py when present.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <utility>
#include <vector>

#include "../../include/dfx_host.hpp"

constexpr int CS = 32;

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
  const int K = argc > 1 ? std::atoi(argv[1]) : 16, REPS = argc > 2 ? std::atoi(argv[2]) : 7, NB = argc > 3 ? std::atoi(argv[3]) : 0;
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
      if (k == 0) for (std::size_t i = 0; i < jac.size(); ++i) jac[i] = 0.004f * (float)std::sin(0.37 * (double)(i % 9973));
      else for (std::size_t i = 0; i < jac.size(); i += 7) jac[i] += 1e-5f;   // (distinct buffers are what the timing needs; the content only has to be smooth and non-zero)
      kf->FillPyramids(img.data(), 1);
      kf->SetDecoderOutputs(0, prx.data(), sd.data(), jac.data());
      for (int c = 0; c < CS; ++c) kf->code[(std::size_t)c] = 0.2f * (float)std::sin(0.9 * c + k);
      kf->UpdateDepthMaps(2.0f, true);
      kfs.push_back(kf);
    }
    const dfx_cam cam{ 554.256f, 579.411f, 320.f, 240.f, (float)W, (float)H };
    std::vector<dfx_se3> pose((std::size_t)K);
    for (int k = 0; k < K; ++k) { const float a = 0.002f * k; pose[(std::size_t)k] = dfx_se3{ { 0, std::sin(a / 2), 0, std::cos(a / 2) }, { 0.004f * k, -0.002f * k, 0.001f * k } }; }
    df::SfmAligner<float, CS> aligner;
    std::vector<std::unique_ptr<dfx::PhotometricFactor<CS>>> own;
    std::vector<dfx::PhotometricFactor<CS>*> factors;
    std::vector<dfx::FactorValues<CS>> values;
    std::vector<std::pair<int, int>> links;
    if (NB <= 0) { for (int i = 0; i < K; ++i) for (int j = i + 1; j < K; ++j) links.emplace_back(i, j); }
    else {
      for (int i = 0; i < K; ++i) {   // the NB nearest keyframes by index, on both sides (deepfactors_amd.dist.PairGraph.window)
        std::vector<int> cand;
        for (int j = 0; j < K; ++j) if (j != i) cand.push_back(j);
        std::sort(cand.begin(), cand.end(), [&](int a, int b) { return std::abs(a - i) != std::abs(b - i) ? std::abs(a - i) < std::abs(b - i) : a < b; });
        cand.resize((std::size_t)std::min<int>(NB, (int)cand.size()));
        std::sort(cand.begin(), cand.end());
        for (int j : cand) links.emplace_back(i, j);
      }
    }
    std::vector<std::unique_ptr<dfx::SparseGeometricFactor<CS>>> gown;
    std::vector<dfx::SparseGeometricFactor<CS>*> gfac;
    std::vector<dfx::GeoValues<CS>> gval;
    unsigned rs = 2463534242u;
    auto rnd = [&](unsigned m) { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return (int32_t)(rs % m); };
    constexpr int kGeoPoints = 500;   // data/flags/common.flags: geo_npoints
    for (const auto& ij : links) {
      const int i = ij.first, j = ij.second;
      own.emplace_back(new dfx::PhotometricFactor<CS>(cam, kfs[(std::size_t)i], kfs[(std::size_t)j], 0));
      factors.push_back(own.back().get());
      values.push_back(dfx::FactorValues<CS>{ pose[(std::size_t)i], pose[(std::size_t)j], kfs[(std::size_t)i]->code });
      std::vector<std::array<int32_t, 2>> pts((std::size_t)kGeoPoints);
      for (auto& q : pts) q = { rnd((unsigned)W), rnd((unsigned)H) };
      gown.emplace_back(new dfx::SparseGeometricFactor<CS>(cam, pts, kfs[(std::size_t)i], kfs[(std::size_t)j], 0.1f));
      gfac.push_back(gown.back().get());
      gval.push_back(dfx::GeoValues<CS>{ pose[(std::size_t)i], pose[(std::size_t)j], kfs[(std::size_t)i]->code, kfs[(std::size_t)j]->code });
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
    // the geometric half
    std::vector<double> g_dev, g_host, g_pinned, g_serial, g_gram;
    dfx::PinnedBuffer<float> rows_pinned((std::size_t)n * kGeoPoints * dfx::SparseGeometricFactor<CS>::kCols, ctx);
    dfx::DeviceImage<float> rows_dev((std::size_t)n * kGeoPoints * dfx::SparseGeometricFactor<CS>::kCols / 64 + 1, 64, ctx);
    for (int rep = 0; rep < REPS + 2; ++rep) {
      double t0 = now_ms();
      (void)dfx::SparseGeometricLinearizeAll<CS>(gfac, gval, rows_dev.ptr());
      dfx::check(dfx_sync(ctx->get()));
      const double a = now_ms() - t0;
      t0 = now_ms();
      const std::vector<float> rows = dfx::SparseGeometricLinearizeAll<CS>(gfac, gval);
      const double b = now_ms() - t0;
      sink += rows[rows.size() / 2];
      t0 = now_ms();
      const std::size_t nrows = dfx::SparseGeometricLinearizeAll<CS>(gfac, gval, rows_pinned);
      const double bp = now_ms() - t0;
      t0 = now_ms();
      const std::vector<float> gram = dfx::SparseGeometricGramAll<CS>(gfac, gval);
      const double bg = now_ms() - t0;
      sink += gram[gram.size() / 2];
      if (std::memcmp(rows.data(), rows_pinned.data(), nrows * dfx::SparseGeometricFactor<CS>::kCols * sizeof(float)) != 0) { std::printf("pinned rows differ from the pageable copy\n"); return 1; }
      double c = 0;
      if (rep < 4) {   // (the per-factor pattern is slow: a few rounds suffice)
        t0 = now_ms();
        for (int k = 0; k < n; ++k) sink += gfac[(std::size_t)k]->Linearize(gval[(std::size_t)k].pose0, gval[(std::size_t)k].pose1, gval[(std::size_t)k].code0, gval[(std::size_t)k].code1)[5];
        c = now_ms() - t0;
      }
      if (rep >= 2) { g_dev.push_back(a); g_host.push_back(b); g_pinned.push_back(bp); g_gram.push_back(bg); if (rep < 4) g_serial.push_back(c); }
    }
    std::printf("gn_round_bench keyframes %d factors %d (640x480, cs %d), median of %d rounds\n", K, n, CS, REPS);
    std::printf("geometric_batched_rows_on_device_ms %.3f  per_factor_us %.1f\n", median(g_dev), median(g_dev) / n * 1e3);
    std::printf("geometric_batched_rows_to_host_ms %.3f  per_factor_us %.1f\n", median(g_host), median(g_host) / n * 1e3);
    std::printf("geometric_batched_rows_to_pinned_host_ms %.3f  per_factor_us %.1f\n", median(g_pinned), median(g_pinned) / n * 1e3);
    std::printf("geometric_batched_gram_to_host_ms %.3f  per_factor_us %.1f\n", median(g_gram), median(g_gram) / n * 1e3);
    std::printf("geometric_serial_ms %.3f  per_factor_us %.1f\n", median(g_serial), median(g_serial) / n * 1e3);
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
