// host_test.cpp -- include/dfx_host.hpp on a GPU: the device-resident keyframe store (Frame / Keyframe<CS>: FillPyramids, decoder
// hand-over, UpdateDepthMaps) and the batching seam (LinearizeAll over the PhotometricFactors of a small window, two pyramid
// levels) against per-factor evaluation through the same C ABI; the relinearisation cache of GetJacobiansIfNeeded
// (photometric_factor.cpp:296-327); the HessianFactor blocks (:105-161); error() (:60-81).  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/dfx_host.hpp"

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s at %s:%d\n", #c, __FILE__, __LINE__); return 1; } } while (0)

constexpr int CS = 32;

static double tex(int k, double u, double v) {
  return 0.5 + 0.2 * std::sin(0.081 * u + 0.047 * v + 0.3 * k) + 0.15 * std::sin(0.033 * u - 0.112 * v + 1.0 + k) + 0.1 * std::sin(0.15 * u + 0.09 * v + 2.0);
}

int main() {
  try {
    const std::size_t W = 320, H = 240, L = 2;
    const int K = 4;
    auto ctx = dfx::Context::Default();
    std::vector<std::shared_ptr<dfx::Keyframe<CS>>> kfs;
    std::vector<std::vector<float>> img_host(K);
    for (int k = 0; k < K; ++k) {
      auto kf = std::make_shared<dfx::Keyframe<CS>>(L, W, H, ctx);
      kf->id = k;
      img_host[k].resize(W * H);
      for (std::size_t y = 0; y < H; ++y) for (std::size_t x = 0; x < W; ++x) img_host[k][y * W + x] = (float)tex(k, (double)x + 1.5 * k, (double)y - 0.8 * k);
      kf->FillPyramids(img_host[k].data(), L);
      for (std::size_t l = 0; l < L; ++l) {   // decoder outputs per level: proximity of a tilted plane, zero log-uncertainty, smooth code Jacobian
        const std::size_t w = W >> l, h = H >> l;
        std::vector<float> prx(w * h), sd(w * h, 0.0f), jac(w * h * CS);
        for (std::size_t y = 0; y < h; ++y) for (std::size_t x = 0; x < w; ++x) {
          const double d = 2.5 + 0.3 * ((double)x / w - 0.5) - 0.3 * ((double)y / h - 0.5) + 0.05 * k;
          prx[y * w + x] = (float)(2.0 / (2.0 + d));
          for (int c = 0; c < CS; ++c) jac[(y * w + x) * CS + c] = 0.004f * (float)(std::sin(0.02 * (c + 1) * x * (1 << l) / 4.0 + 0.7 * c + k) * std::cos(0.015 * (c + 2) * y * (1 << l) / 4.0 - 0.3 * c));
        }
        kf->SetDecoderOutputs(l, prx.data(), sd.data(), jac.data());
      }
      for (int c = 0; c < CS; ++c) kf->code[c] = 0.2f * (float)std::sin(0.9 * c + k);
      kf->UpdateDepthMaps(2.0f, true);
      kfs.push_back(kf);
    }
    // FillPyramids: level 1 is the 5x5 binomial blur-down of level 0 (cu_image_proc.cpp:134-164) -- spot-check one interior pixel on the host
    {
      const std::vector<float> l1 = kfs[0]->pyr_img[1].Download();
      const int B[5] = { 1, 4, 6, 4, 1 };
      const std::size_t x = 40, y = 30;
      double s = 0;
      for (int py = 0; py < 5; ++py) for (int px = 0; px < 5; ++px) s += B[px] * B[py] * (double)img_host[0][(2 * y + py - 2) * W + (2 * x + px - 2)];
      REQUIRE(std::fabs(l1[y * (W >> 1) + x] - s / 256.0) < 1e-6);
      const std::vector<float> vld = kfs[0]->pyr_vld[0].Download();
      REQUIRE(vld[0] == 1.0f && vld[W * H - 1] == 1.0f);                       // mapper.cpp:937
      const std::vector<float> dpt = kfs[1]->pyr_dpt[1].Download();
      REQUIRE(dpt[10] > 1.5f && dpt[10] < 4.0f);
      const std::vector<dfx::Grad2f> dg = kfs[1]->dpt_grad.Download();
      REQUIRE(std::isfinite(dg[W * 10 + 10].gx));
    }
    {   // FillPyramids is ONE enqueue (dfx_build_pyramid_batch_async): the same bytes as the per-level operators, for one frame and for a batch
      std::vector<dfx::Frame*> fr;
      std::vector<std::unique_ptr<dfx::Frame>> own_fr;
      for (int k = 0; k < K; ++k) { own_fr.emplace_back(new dfx::Frame(L, W, H, ctx)); own_fr.back()->pyr_img[0].Upload(img_host[k].data()); fr.push_back(own_fr.back().get()); }
      dfx::FillPyramidsBatch(fr, L);
      for (int k = 0; k < K; ++k) {
        dfx::Frame ref(L, W, H, ctx);
        ref.pyr_img[0].Upload(img_host[k].data());
        for (std::size_t l = 0; l < L; ++l) {
          if (l > 0) df::GaussianBlurDown(ref.pyr_img[l - 1], ref.pyr_img[l], ctx);
          df::SobelGradients(ref.pyr_img[l], ref.pyr_grad[l], ctx);
          const std::vector<float> a = fr[k]->pyr_img[l].Download(), b = ref.pyr_img[l].Download(), c = kfs[k]->pyr_img[l].Download();
          REQUIRE(a == b && c == b);
          const std::vector<dfx::Grad2f> ga = fr[k]->pyr_grad[l].Download(), gb = ref.pyr_grad[l].Download(), gc = kfs[k]->pyr_grad[l].Download();
          REQUIRE(std::memcmp(ga.data(), gb.data(), ga.size() * sizeof(dfx::Grad2f)) == 0 && std::memcmp(gc.data(), gb.data(), gc.size() * sizeof(dfx::Grad2f)) == 0);
        }
      }
    }
    // cameras per level (camera_pyramid.h:41-46)
    std::vector<dfx_cam> cams;
    for (std::size_t l = 0; l < L; ++l) { const float s = 1.0f / (1 << l); cams.push_back(dfx_cam{ 277.128f * s, 289.706f * s, 160.f * s, 120.f * s, (float)(W >> l), (float)(H >> l) }); }
    // poses on a small arc
    std::vector<dfx_se3> pose(K);
    for (int k = 0; k < K; ++k) { const float a = 0.004f * k; pose[k] = dfx_se3{ { 0, std::sin(a / 2), 0, std::cos(a / 2) }, { 0.01f * k, -0.004f * k, 0.002f * k } }; }

    df::SfmAligner<float, CS> aligner;
    // the factors of the window: every ordered pair i != j, both levels (mapper.cpp:308-311 links both directions)
    std::vector<std::unique_ptr<dfx::PhotometricFactor<CS>>> own, own_single;
    std::vector<dfx::PhotometricFactor<CS>*> factors;
    std::vector<dfx::FactorValues<CS>> values;
    for (std::size_t l = 0; l < L; ++l)
      for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) if (i != j) {
        own.emplace_back(new dfx::PhotometricFactor<CS>(cams[l], kfs[i], kfs[j], (int)l));
        own_single.emplace_back(new dfx::PhotometricFactor<CS>(cams[l], kfs[i], kfs[j], (int)l));
        factors.push_back(own.back().get());
        values.push_back(dfx::FactorValues<CS>{ pose[i], pose[j], kfs[i]->code });
      }
    const int n = (int)factors.size();
    REQUIRE(n == 24);
    REQUIRE(dfx::LinearizeAll(aligner, factors, values) == n);
    // == per-factor evaluation (a batch of 12 and a single pair take different launch shapes: same sums up to fp32 reassociation)
    for (int k = 0; k < n; ++k) {
      const auto& one = own_single[k]->GetJacobiansIfNeeded(aligner, values[k].pose0, values[k].pose1, values[k].code0);
      const auto& bat = factors[k]->system();
      REQUIRE(one.inliers == bat.inliers && one.inliers > 0);
      double scale = 0, err = 0;
      for (std::size_t e = 0; e < one.JtJ.coeff().size(); ++e) { scale = std::fmax(scale, std::fabs(one.JtJ.coeff()[e])); err = std::fmax(err, std::fabs(one.JtJ.coeff()[e] - bat.JtJ.coeff()[e])); }
      REQUIRE(err <= 3e-6 * scale);
      REQUIRE(std::fabs(one.residual - bat.residual) <= 1e-5f * one.residual);
      REQUIRE(factors[k]->linearizations() == 1);
    }
    // cache: nothing moved -> nothing is launched; node 2's pose moved by 1e-3 -> exactly the factors touching that node relinearise
    REQUIRE(dfx::LinearizeAll(aligner, factors, values) == 0);
    auto moved = values;
    int expect = 0;
    {
      int k = 0;
      for (std::size_t l = 0; l < L; ++l)
        for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) if (i != j) {
          if (i == 2) { moved[k].pose0.t[0] += 1e-3f; ++expect; }
          else if (j == 2) { moved[k].pose1.t[0] += 1e-3f; ++expect; }
          ++k;
        }
    }
    REQUIRE(dfx::LinearizeAll(aligner, factors, moved) == expect && expect == 12);
    REQUIRE(factors[0]->linearizations() == 1);   // factor 0 -> 1 at level 0 did not involve node 2
    // a move below the threshold does not relinearise (1e-6 in the tangent space)
    auto tiny = moved;
    tiny[0].pose0.t[1] += 1e-8f;
    REQUIRE(dfx::LinearizeAll(aligner, factors, tiny) == 0);

    // HessianFactor blocks (photometric_factor.cpp:105-161) and the rescaled f (:275-282)
    {
      const auto& sys = factors[3]->system();
      const dfx::HessianBlocks<CS> Hb = factors[3]->Hessian();
      REQUIRE(Hb.G11[1 * 6 + 4] == (double)sys.JtJ(1, 4) && Hb.G12[2 * 6 + 3] == (double)sys.JtJ(2, 9) && Hb.G13[5 * CS + 31] == (double)sys.JtJ(5, 43));
      REQUIRE(Hb.G22[0] == (double)sys.JtJ(6, 6) && Hb.G23[3 * CS + 7] == (double)sys.JtJ(9, 19) && Hb.G33[31 * CS + 2] == (double)sys.JtJ(14, 43));
      REQUIRE(Hb.g1[0] == -(double)sys.Jtr[0] && Hb.g2[5] == -(double)sys.Jtr[11] && Hb.g3[31] == -(double)sys.Jtr[43] && Hb.f == (double)sys.residual);
      REQUIRE(Hb.f > 0 && std::isfinite(Hb.f));
      // f is the residual rescaled to the full image: residual / inliers * w * h
      dfx_sfm_pair p = factors[3]->MakePair(moved[3].pose0, moved[3].pose1);
      const dfx_sfm_params prm = aligner.Params();
      std::vector<unsigned char> raw(dfx_item_size(12 + CS));
      dfx::check(dfx_sfm_step(aligner.ContextHandle(), CS, &p.pose0, &p.pose1, &p.cam, &prm, &p.img0, &p.img1, &p.dpt0, nullptr, &p.valid0, &p.prx0_jac, &p.grad1, raw.data()));
      const float want = dfx_item_residual(raw.data(), 12 + CS) / dfx_item_inliers(raw.data(), 12 + CS) * p.cam.w * p.cam.h;
      REQUIRE(std::fabs(Hb.f - want) <= 2e-5 * want);
      const double e = factors[3]->error(aligner, moved[3].pose0, moved[3].pose1, moved[3].code0);
      REQUIRE(e > 0 && std::isfinite(e));
      // error() over the whole factor set in one decode + one EvaluateError launch per level == factor by factor (same per-pair arithmetic;
      // the number of workgroups per pair, hence the fp32 summation order, may differ between the batched and the single launch)
      std::vector<double> per;
      const double total = dfx::ErrorAll<CS>(aligner, factors, moved, &per);
      REQUIRE((int)per.size() == n && std::fabs(per[3] - e) <= 1e-5 * e && std::isfinite(total));
      double sum = 0;
      for (int k = 0; k < n; ++k) {
        const double ek = factors[k]->error(aligner, moved[k].pose0, moved[k].pose1, moved[k].code0);
        REQUIRE(std::fabs(per[k] - ek) <= 1e-5 * ek);
        sum += per[k];
      }
      REQUIRE(sum == total);
    }
    {   // SparseGeometricFactor (sparse_geometric_factor.cpp:147-275): every factor of the window in ONE launch == factor by factor, bit for bit
      std::vector<std::unique_ptr<dfx::SparseGeometricFactor<CS>>> gown;
      std::vector<dfx::SparseGeometricFactor<CS>*> gf;
      std::vector<dfx::GeoValues<CS>> gv;
      unsigned rs = 12345u;
      auto rnd = [&](unsigned m) { rs = rs * 1664525u + 1013904223u; return (int)((rs >> 8) % m); };
      for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) if (i != j) {
        std::vector<std::array<int32_t, 2>> pts((std::size_t)(37 + 11 * i + j));   // ragged point counts
        for (auto& q : pts) q = { rnd((unsigned)W), rnd((unsigned)H) };
        gown.emplace_back(new dfx::SparseGeometricFactor<CS>(cams[0], pts, kfs[i], kfs[j], 0.1f));
        gf.push_back(gown.back().get());
        gv.push_back(dfx::GeoValues<CS>{ pose[i], pose[j], kfs[i]->code, kfs[j]->code });
      }
      const std::vector<float> all = dfx::SparseGeometricLinearizeAll<CS>(gf, gv);
      std::size_t off = 0, nonzero = 0;
      for (std::size_t k = 0; k < gf.size(); ++k) {
        const std::vector<float> one = gf[k]->Linearize(gv[k].pose0, gv[k].pose1, gv[k].code0, gv[k].code1);
        REQUIRE(std::memcmp(one.data(), all.data() + off, one.size() * sizeof(float)) == 0);
        for (float v : one) nonzero += v != 0.0f;
        off += one.size();
      }
      REQUIRE(off == all.size() && nonzero > all.size() / 2);
      // rows left on the device (enqueue only) == the fetched ones
      dfx::DeviceImage<float> rows_dev(all.size(), 1, ctx);
      (void)dfx::SparseGeometricLinearizeAll<CS>(gf, gv, rows_dev.ptr());
      REQUIRE(rows_dev.Download() == all);
      // the round's normal equations formed on the device == [A | b]^T [A | b] of the fetched rows (double on the host), factor by factor
      const std::vector<float> gram = dfx::SparseGeometricGramAll<CS>(gf, gv);
      constexpr int NC = dfx::SparseGeometricFactor<CS>::kCols, NE = NC * (NC + 1) / 2;
      REQUIRE(gram.size() == gf.size() * (std::size_t)NE);
      off = 0;
      for (std::size_t k = 0; k < gf.size(); ++k) {
        const std::size_t np = (std::size_t)gf[k]->n_points();
        std::vector<double> diag((std::size_t)NC, 0.0);
        for (std::size_t r = 0; r < np; ++r) for (int i = 0; i < NC; ++i) { const double a = all[off + r * NC + i]; diag[(std::size_t)i] += a * a; }
        int e = 0;
        for (int i = 0; i < NC; ++i)
          for (int j = i; j < NC; ++j, ++e) {
            double want = 0.0;
            for (std::size_t r = 0; r < np; ++r) want += (double)all[off + r * NC + i] * (double)all[off + r * NC + j];
            const double scale = std::sqrt(diag[(std::size_t)i] * diag[(std::size_t)j]) + 1e-30;
            REQUIRE(std::fabs((double)gram[k * NE + (std::size_t)e] - want) <= 5e-5 * scale);
          }
        off += np * NC;
      }
    }
    {   // keyframe replication over the multi-GPU C ABI (real RCCL, a world of one): the root's content stays, the call orders on the stream
      unsigned char id[DFX_COMM_ID_BYTES];
      dfx::check(dfx_comm_get_unique_id(id));
      dfx_comm* comm = nullptr;
      dfx::check(dfx_comm_create(ctx->get(), id, 0, 1, &comm));
      const std::vector<float> before = kfs[1]->pyr_jac[0].Download();
      const std::array<float, CS> code_before = kfs[1]->code;
      dfx::KeyframeBroadcast<CS>(*kfs[1], comm, 0);
      REQUIRE(kfs[1]->pyr_jac[0].Download() == before && kfs[1]->code == code_before && kfs[1]->id == 1);
      dfx_comm_destroy(comm);
    }
    std::printf("host_test OK (%d factors, %d relinearised after moving one node)\n", n, expect);
    std::fflush(stdout);   // the verdict reaches the pipe before the destructors and the GPU runtime's teardown run
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 1;
  }
  return 0;
}
