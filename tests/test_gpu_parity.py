"""HIP path (through the C ABI of libdfx.so) vs the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

from helpers import assert_blocks_below, assert_item_close

pytestmark = pytest.mark.gpu


def _pair(dfx, w, h, cs, seed, **kw):
    from deepfactors_amd import synth
    p = synth.make_pair(w, h, cs, seed=seed, device="cpu", **kw)
    return p, synth.to_numpy(p), synth.to_device(p, "cuda")


@pytest.mark.parametrize("w,h,cs", [(160, 120, 32), (320, 240, 32), (100, 77, 32), (64, 48, 16), (96, 64, 64), (640, 480, 32), (128, 96, 64),
                                    (101, 67, 16)])
def test_sfm_step_matches_oracle(dfx, oracle, w, h, cs):
    """Every entry of the 44x44 (28x28, 76x76) system, Jtr, residual and inliers against the fp64-accumulating oracle: covers
    each block of the packed z-space (X, Pm, Dd, the vector-ALU P x P sums) for the three code sizes."""
    p, n, g = _pair(dfx, w, h, cs, seed=0xDF02 + w)
    # perturb the pose a little so the gradient is not ~0
    pose1 = n["pose1"].copy(); pose1[4] += 0.01
    al = dfx.SfmAligner(code_size=cs)
    valid_gpu = torch.zeros_like(g["img0"])
    got = al.RunStep(n["pose0"], pose1, n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], g["std0"], valid_gpu,
                     g["prx_jac"], g["grad1"])
    valid_ref = np.zeros_like(n["img0"])
    ref = oracle.sfm_step(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"],
                          valid0=valid_ref, accum_f64=True)
    assert ref.inliers > 0.5 * w * h
    assert_item_close(got, ref, w, h, what=f"sfm_step {w}x{h} cs={cs}")
    # valid0 is written 1.0 exactly where the oracle does (up to boundary flips)
    assert int((valid_gpu.cpu().numpy() != valid_ref).sum()) <= max(1, int(1e-5 * w * h))


def test_sfm_step_reference_test_poses(dfx, oracle):
    """Poses of ut_sfmaligner.cpp:254-268 (pose1 = inverse(exp(0.1,0.1,0), t=(-.5,-.5,0)), huber 0.5), scaled by 0.1."""
    from deepfactors_amd import synth
    w, h, cs = 256, 192, 32
    p, n, g = _pair(dfx, w, h, cs, seed=7)
    R = synth.so3_exp(np.array([0.1, 0.1, 0.0]) * 0.1)
    t = np.array([-0.5, -0.5, 0.0]) * 0.1
    pose1 = synth.pose_qt(R.T, -R.T @ t)
    params = dfx.SfmAlignerParams(dfx.DenseSfmParams(huber_delta=0.5))
    al = dfx.SfmAligner(params, code_size=cs)
    got = al.RunStep(n["pose0"], pose1, n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, None, g["prx_jac"], g["grad1"])
    ref = oracle.sfm_step(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], huber_delta=0.5)
    assert_item_close(got, ref, w, h)
    # the reference's own GPU-vs-CPU bar (ut_sfmaligner.cpp:320-326): inliers equal, |dJtJ| <= 1e-1 abs
    assert abs(got.inliers - ref.inliers) <= 1
    assert np.abs(got.toDenseMatrix() - ref.dense()).max() <= 1e-1 or np.abs(ref.JtJ).max() > 1e3


def test_fp32_chain_is_tight_and_unknown_modes_are_rejected(dfx, oracle):
    """The fp32 MFMA chain against the fp64-accumulated oracle: every entry of every block within 1e-5 of its own Cauchy-Schwarz
    scale sqrt(JtJ_ii JtJ_jj) (the stated tolerance is 1e-4).  An unknown evaluation mode must be refused loudly, not silently mapped to something else (the exact bf16 split,
    DFX_MFMA_BF16X3, has its own tests: tests/test_gpu_bf16x3.py)."""
    from deepfactors_amd import _lib
    w, h, cs = 320, 240, 32
    p, n, g = _pair(dfx, w, h, cs, seed=77)
    ref = oracle.sfm_step(n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"])
    ctx = dfx.Context(0)
    ctx.set_mfma_mode(_lib.DFX_MFMA_F32_CHAIN)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    got = al.RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, None, g["prx_jac"], g["grad1"])
    assert got.inliers == ref.inliers
    assert_blocks_below(got, ref, 1e-5, what="fp32 chain 320x240 cs=32")
    with pytest.raises(dfx.DfxError):
        ctx.set_mfma_mode(7)


def test_sfm_step_batch(dfx, oracle):
    w, h, cs = 128, 96, 32
    al = dfx.SfmAligner(code_size=cs)
    host, dev = [], []
    for k in range(5):
        p, n, g = _pair(dfx, w, h, cs, seed=100 + k, motion_scale=0.5 + 0.3 * k)
        host.append(n); dev.append(g)
    arr = al.make_pairs([dict(pose0=n["pose0"], pose1=n["pose1"], cam=n["cam"], img0=g["img0"], img1=g["img1"], dpt0=g["dpt0"],
                              prx0_jac=g["prx_jac"], grad1=g["grad1"]) for n, g in zip(host, dev)])
    items = al.RunStepBatch(arr)
    for k, (n, it) in enumerate(zip(host, items)):
        ref = oracle.sfm_step(n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"])
        assert_item_close(it, ref, w, h, what=f"batch item {k}")
    # async variant into device memory gives the same bytes
    out = torch.zeros(len(arr) * dfx.item_size(12 + cs), dtype=torch.uint8, device="cuda")
    al.RunStepBatchAsync(arr, out)
    al.ctx.sync()
    items2 = al.items_from_bytes(out.cpu().numpy(), cs)
    for a, b in zip(items, items2):
        assert np.array_equal(a.JtJ, b.JtJ) and np.array_equal(a.Jtr, b.Jtr) and a.inliers == b.inliers


def test_sfm_step_batch_mixed_cameras(dfx, oracle):
    """Every pair of a batch may carry its own intrinsics: the per-camera ray tables (dfx_api.cpp ray_table) are looked up
    per pair, and a context survives more distinct cameras than its table cache holds (256)."""
    w, h, cs = 96, 64, 16
    al = dfx.SfmAligner(code_size=cs)
    host, dev = [], []
    for k in range(3):
        p, n, g = _pair(dfx, w, h, cs, seed=400 + k)
        n["cam"] = n["cam"].copy()
        n["cam"][0] *= 1.0 + 0.07 * k; n["cam"][1] *= 1.0 - 0.05 * k; n["cam"][2] += 1.5 * k; n["cam"][3] -= 0.75 * k
        host.append(n); dev.append(g)
    arr = al.make_pairs([dict(pose0=n["pose0"], pose1=n["pose1"], cam=n["cam"], img0=g["img0"], img1=g["img1"], dpt0=g["dpt0"],
                              prx0_jac=g["prx_jac"], grad1=g["grad1"]) for n, g in zip(host, dev)])
    for k, (n, it) in enumerate(zip(host, al.RunStepBatch(arr))):
        ref = oracle.sfm_step(n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"])
        assert_item_close(it, ref, w, h, what=f"mixed-camera item {k}")
    n, g = host[0], dev[0]
    first = None
    for k in range(300):   # 300 distinct cameras through one context: the cache is recycled, results stay right
        cam = n["cam"].copy(); cam[2] += 1e-3 * k
        it = al.RunStep(n["pose0"], n["pose1"], None, cam, g["img0"], g["img1"], g["dpt0"], None, None, g["prx_jac"], g["grad1"])
        if k == 0:
            first = it
        assert it.inliers > 0
    again = al.RunStep(n["pose0"], n["pose1"], None, n["cam"], g["img0"], g["img1"], g["dpt0"], None, None, g["prx_jac"], g["grad1"])
    assert np.array_equal(first.raw, again.raw)


def test_native_graph_assembly_matches_torch(dfx):
    """dfx_graph_assemble_async == the torch formulation (deepfactors_amd/dist.py) on a general keyframe graph: nodes with many
    incident pairs in both roles (keyframe of some pairs, frame of others, both directions of one link), a shard of the pair
    list, dirty target buffers, the three code sizes."""
    from deepfactors_amd.dist import NormalEquations, PairGraph
    w, h = 96, 64
    for cs in (32, 16, 64):
        al = dfx.SfmAligner(code_size=cs)
        graph = PairGraph(5, [(0, 1), (0, 2), (1, 0), (1, 2), (2, 4), (3, 2), (0, 4), (4, 3), (2, 1)])
        n = graph.n_pairs
        dev = [_pair(dfx, w, h, cs, seed=300 + k)[1:] for k in range(n)]
        arr = al.make_pairs([dict(pose0=nn["pose0"], pose1=nn["pose1"], cam=nn["cam"], img0=g["img0"], img1=g["img1"], dpt0=g["dpt0"],
                                  prx0_jac=g["prx_jac"], grad1=g["grad1"]) for nn, g in dev])
        isz = dfx.item_size(12 + cs)
        items = torch.zeros(n * isz, dtype=torch.uint8, device="cuda")
        al.RunStepBatchAsync(arr, items)
        al.ctx.sync()
        for first, cnt in ((0, n), (2, 5), (7, 2)):
            a, b = NormalEquations(graph, cs, "cuda"), NormalEquations(graph, cs, "cuda")
            b.buf.fill_(123.0)   # e.g. the root's copy after a reduce: every entry must be overwritten
            a.assemble(items[first * isz:(first + cnt) * isz], first, cnt, isz)
            for _ in range(2):
                b.assemble_native(al.ctx, items[first * isz:(first + cnt) * isz], first, cnt)
            al.ctx.sync()
            assert torch.equal(a.buf, b.buf), (cs, first, cnt, float((a.buf - b.buf).abs().max()))
        M = a.dense()
        assert torch.allclose(M, M.T) and float(M.abs().max()) > 0
        # step + assembly in one call: same bytes
        c = NormalEquations(graph, cs, "cuda")
        items2 = torch.zeros_like(items)
        al.RunStepBatchAssembleAsync(arr, items2, c, 0)
        al.ctx.sync()
        full = NormalEquations(graph, cs, "cuda")
        full.assemble(items, 0, n, isz)
        assert torch.equal(items, items2) and torch.equal(full.buf, c.buf)
        # the dense system equals the sum of the pairs' 44x44 systems placed by hand (PhotometricFactor::linearize's slicing)
        D, NP = 6 + cs, 12 + cs
        ref = np.zeros((graph.n_nodes * D, graph.n_nodes * D))
        gref = np.zeros(graph.n_nodes * D)
        for p, it in enumerate(al.items_from_bytes(items.cpu().numpy(), cs)):
            ka, fb = [int(v) for v in graph.pairs[p]]
            idx = np.array([ka * D + i for i in range(6)] + [fb * D + i for i in range(6)] + [ka * D + 6 + i for i in range(cs)])
            ref[np.ix_(idx, idx)] += it.toDenseMatrix().astype(np.float64)
            gref[idx] += it.Jtr.astype(np.float64)
        assert np.abs(full.dense().numpy() - ref).max() <= 1e-6 * np.abs(ref).max()
        assert np.abs(full.g.cpu().double().numpy().reshape(-1) - gref).max() <= 1e-6 * np.abs(gref).max()


def test_sfm_step_deterministic(dfx):
    p, n, g = _pair(dfx, 320, 240, 32, seed=3)
    al = dfx.SfmAligner(code_size=32)
    a = al.RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, None, g["prx_jac"], g["grad1"])
    for _ in range(3):
        b = al.RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, None, g["prx_jac"], g["grad1"])
        assert np.array_equal(a.raw, b.raw)
    # valid0: set where valid, never cleared, already-set pixels left alone (the kernel reads before it writes): a marker
    # value survives on invalid pixels, 1.0 everywhere else, and the sums do not depend on the buffer's content
    vld = torch.full_like(g["img0"], 7.0)
    c = al.RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, vld, g["prx_jac"], g["grad1"])
    assert np.array_equal(a.raw, c.raw)
    v = vld.cpu().numpy()
    assert int((v == 1.0).sum()) == a.inliers and int((v == 7.0).sum()) == v.size - a.inliers
    d = al.RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, vld, g["prx_jac"], g["grad1"])
    assert np.array_equal(a.raw, d.raw) and np.array_equal(v, vld.cpu().numpy())


def test_sfm_step_masked_nan_jacobian(dfx, oracle):
    """Garbage (NaN) in the code Jacobian / depth of pixels that are not inliers must not reach the sums."""
    w, h, cs = 128, 96, 32
    p, n, g = _pair(dfx, w, h, cs, seed=11)
    n["dpt0"][10:20, 30:50] = np.nan
    jac = n["prx_jac"].reshape(h, w, cs)
    jac[10:20, 30:50, :] = np.nan
    g["dpt0"] = torch.from_numpy(n["dpt0"]).cuda()
    g["prx_jac"] = torch.from_numpy(n["prx_jac"]).cuda()
    al = dfx.SfmAligner(code_size=cs)
    got = al.RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, None, g["prx_jac"], g["grad1"])
    ref = oracle.sfm_step(n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"])
    assert np.isfinite(got.JtJ).all() and np.isfinite(got.Jtr).all()
    assert_item_close(got, ref, w, h)


def test_sfm_step_no_overlap(dfx):
    """inliers == 0 is the in-band 'no overlap' signal (photometric_factor.cpp:279-282)."""
    p, n, g = _pair(dfx, 128, 96, 32, seed=5)
    pose1 = n["pose1"].copy(); pose1[4:] = [100.0, 0.0, 0.0]
    al = dfx.SfmAligner(code_size=32)
    got = al.RunStep(n["pose0"], pose1, n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, None, g["prx_jac"], g["grad1"])
    assert got.inliers == 0 and got.residual == 0.0 and not got.JtJ.any()


@pytest.mark.parametrize("w,h", [(160, 120), (320, 240), (101, 67), (640, 480)])
def test_se3_step_and_error_and_warp(dfx, oracle, w, h):
    from deepfactors_amd import synth
    p, n, g = _pair(dfx, w, h, 16, seed=21 + w, with_decoder=False)
    se3 = synth.IDENTITY.copy()
    al = dfx.SE3Aligner()
    al.SetHuberDelta(0.1)
    got = al.RunStep(se3, n["cam"], g["img0"], g["img1"], g["dpt0"], g["grad1"])
    ref = oracle.se3_step(se3, n["cam"], n["img0"], n["img1"], n["dpt0"], n["grad1"], 0.1)
    assert_item_close(got, ref, w, h, what="se3_step")
    # error
    sal = dfx.SfmAligner(code_size=32)
    e = sal.EvaluateError(n["pose0"], n["pose1"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, g["grad1"])
    er, en = oracle.sfm_error(n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], n["dpt0"], 0.1)
    assert abs(e.inliers - en) <= 1 and abs(e.residual - er) <= 1e-4 * max(er, 1e-3) + 1e-6
    # warp
    img2 = torch.full_like(g["img0"], -1.0)
    wi = al.Warp(n["pose10_true"], n["cam"], g["img0"], g["img1"], g["dpt0"], img2)
    img2_ref, wr, wn = oracle.se3_warp(n["pose10_true"], n["cam"], n["img0"], n["img1"], n["dpt0"])
    assert abs(wi.inliers - wn) <= 1
    diff = np.abs(img2.cpu().numpy() - img2_ref)
    assert (diff > 1e-5).sum() <= 2   # boundary flips only
    assert abs(wi.residual - wr) <= 1e-3 * max(abs(wr), 1.0) + 1e-3


@pytest.mark.parametrize("w,h,cs", [(160, 120, 32), (101, 67, 32), (64, 48, 16), (96, 64, 64), (640, 480, 32)])
def test_update_depth(dfx, oracle, w, h, cs):
    p, n, g = _pair(dfx, w, h, cs, seed=31 + w)
    out = torch.empty_like(g["img0"])
    dfx.UpdateDepth(n["code"], g["prx_orig"], g["prx_jac"], 2.0, out)
    ref = oracle.update_depth(n["code"], n["prx_orig"], n["prx_jac"], 2.0)
    got = out.cpu().numpy()
    # dpt = a/prx - a: fp32 summation-order differences in the dot product scale by a/prx^2
    assert np.abs(got - ref).max() <= 2e-6 * float(((2.0 + ref) ** 2 / 2.0).max())


@pytest.mark.parametrize("w,h", [(160, 120), (101, 67), (640, 480)])
def test_image_proc(dfx, oracle, w, h):
    p, n, g = _pair(dfx, w, h, 16, seed=41 + w, with_decoder=False)
    grad = torch.empty((h, w, 2), dtype=torch.float32, device="cuda")
    dfx.SobelGradients(g["img0"], grad)
    assert np.array_equal(grad.cpu().numpy(), oracle.sobel(n["img0"]))       # exact: taps are x1, x2, /8
    out = torch.empty((h // 2, w // 2), dtype=torch.float32, device="cuda")
    dfx.GaussianBlurDown(g["img0"], out)
    assert np.abs(out.cpu().numpy() - oracle.blur_down(n["img0"])).max() <= 1e-6
    se = dfx.SquaredError(g["img0"], g["img1"])
    sr = oracle.squared_error(n["img0"], n["img1"])
    assert abs(se - sr) <= 1e-5 * sr


@pytest.mark.parametrize("cs", [16, 32])
def test_depth_aligner(dfx, oracle, cs):
    w, h = 128, 96
    p, n, g = _pair(dfx, w, h, cs, seed=51)
    tgt = n["dpt0"] * 1.05 + 0.02
    code = n["code"] * 0.5
    al = dfx.DepthAligner(code_size=cs)
    got = al.RunStep(code, torch.from_numpy(tgt).cuda(), g["prx_orig"], g["prx_jac"], 2.0)
    ref = oracle.depth_aligner_step(code, tgt, n["prx_orig"], n["prx_jac"], 2.0)
    assert got.inliers == ref.inliers == w * h
    assert_item_close(got, ref, w, h, what="depth_aligner")


def test_pitched_rows(dfx, oracle):
    """VisionCore device images are pitched; the ABI carries pitch_bytes."""
    w, h, cs = 100, 60, 32
    p, n, g = _pair(dfx, w, h, cs, seed=61)

    def pitched(t, pad):
        shape = list(t.shape); shape[1] += pad
        big = torch.full(shape, float("nan"), dtype=t.dtype, device=t.device)
        big[:, : t.shape[1]] = t
        return big[:, : t.shape[1]]

    al = dfx.SfmAligner(code_size=cs)
    got = al.RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], pitched(g["img0"], 28), pitched(g["img1"], 12),
                     pitched(g["dpt0"], 4), None, None, pitched(g["prx_jac"], 64), pitched(g["grad1"], 6), )
    ref = oracle.sfm_step(n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"])
    assert_item_close(got, ref, w, h)


def test_errors_are_loud(dfx):
    p, n, g = _pair(dfx, 64, 48, 32, seed=71)
    al = dfx.SfmAligner(code_size=32)
    with pytest.raises(dfx.DfxError):   # CPU tensor: there is no CPU path
        al.RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], g["img0"].cpu(), g["img1"], g["dpt0"], None, None, g["prx_jac"], g["grad1"])
    with pytest.raises(dfx.DfxError):   # size mismatch
        al.RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], g["img0"], g["img1"][:-1], g["dpt0"], None, None, g["prx_jac"], g["grad1"])
    with pytest.raises(dfx.DfxError):   # unsupported code size
        dfx.SfmAligner(code_size=48).RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, None,
                                             torch.zeros((48, 64 * 48), device="cuda"), g["grad1"])


def test_very_wide_image_uses_the_global_ray_table(dfx, oracle):
    """W + H too large for the LDS copy of the ray table (64 KB per workgroup): the kernel variant that reads it from global
    memory must give the same answer."""
    w, h, cs = 12288, 2, 16
    p, n, g = _pair(dfx, w, h, cs, seed=91)
    al = dfx.SfmAligner(code_size=cs)
    got = al.RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, None, g["prx_jac"], g["grad1"])
    ref = oracle.sfm_step(n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"])
    assert_item_close(got, ref, w, h, what="12288x2")


@pytest.mark.parametrize("w,h", [(40, 30), (17, 9), (64, 1), (8, 8)])
def test_tiny_and_narrow_images(dfx, oracle, w, h):
    """Images narrower than one 64-pixel chunk (a chunk wraps several rows) and smaller than one chunk."""
    from deepfactors_amd import synth
    cs = 32
    p, n, g = _pair(dfx, max(w, 16), max(h, 16), cs, seed=90 + w)
    # crop the synthetic pair to (h, w); keep the camera of the crop consistent (principal point inside)
    def crop(a, c=1):
        return np.ascontiguousarray(a[:h, : w * c] if a.ndim == 2 else a[:h, :w])
    n2 = dict(img0=crop(n["img0"]), img1=crop(n["img1"]), dpt0=crop(n["dpt0"]), grad1=crop(n["grad1"]),
              prx_jac=np.ascontiguousarray(n["prx_jac"].reshape(n["img0"].shape[0], -1, cs)[:h, :w].reshape(h, w * cs)),
              prx_orig=crop(n["prx_orig"]))
    cam = np.array([w * 0.9, h * 1.2, w / 2, h / 2, w, h], np.float32)
    g2 = {k: torch.from_numpy(v).cuda() for k, v in n2.items()}
    pose0, pose1 = synth.IDENTITY.copy(), synth.IDENTITY.copy()
    pose1[4:] = [0.002, -0.001, 0.003]
    al = dfx.SfmAligner(code_size=cs)
    got = al.RunStep(pose0, pose1, n["code"], cam, g2["img0"], g2["img1"], g2["dpt0"], None, None, g2["prx_jac"], g2["grad1"])
    ref = oracle.sfm_step(pose0, pose1, cam, n2["img0"], n2["img1"], n2["dpt0"], n2["prx_jac"], n2["grad1"])
    assert got.inliers == ref.inliers
    if ref.inliers:
        assert_item_close(got, ref, w, h, what=f"tiny {w}x{h}")
    se3 = dfx.SE3Aligner()
    s_got = se3.RunStep(pose1, cam, g2["img0"], g2["img1"], g2["dpt0"], g2["grad1"])
    s_ref = oracle.se3_step(pose1, cam, n2["img0"], n2["img1"], n2["dpt0"], n2["grad1"], 0.1)
    assert s_got.inliers == s_ref.inliers
    # EvaluateError walks rows two deep (row_walk DT = 2): images with fewer rows than its warm-up, and the batched forms on them
    e_got = al.EvaluateError(pose0, pose1, cam, g2["img0"], g2["img1"], g2["dpt0"], None, g2["grad1"])
    e_res, e_inl = oracle.sfm_error(pose0, pose1, cam, n2["img0"], n2["img1"], n2["dpt0"], 0.1)
    assert e_got.inliers == e_inl
    assert abs(e_got.residual - e_res) <= 1e-4 * abs(e_res) + 1e-6
    pl = [dict(pose0=pose0, pose1=pose1, cam=cam, img0=g2["img0"], img1=g2["img1"], dpt0=g2["dpt0"], prx0_jac=g2["prx_jac"], grad1=g2["grad1"])] * 3
    for e in al.EvaluateErrorBatch(al.make_pairs(pl)):
        assert e.inliers == e_got.inliers and e.residual == e_got.residual
    sl = [dict(se3=pose1, cam=cam, img0=g2["img0"], img1=g2["img1"], dpt0=g2["dpt0"], grad1=g2["grad1"])] * 3
    for st in se3.RunStepBatch(se3.make_pairs(sl)):
        assert st.inliers == s_got.inliers
    out = torch.empty_like(g2["img0"])
    dfx.UpdateDepth(n["code"], g2["prx_orig"], g2["prx_jac"], 2.0, out)
    d_ref = oracle.update_depth(n["code"], n2["prx_orig"], n2["prx_jac"], 2.0)
    assert np.abs(out.cpu().numpy() - d_ref).max() <= 2e-6 * float(((2.0 + d_ref) ** 2 / 2.0).max())


def test_batched_error_and_se3_step_match_the_single_pair_operators_and_the_oracle(dfx, oracle):
    """dfx_sfm_error_batch / dfx_se3_step_batch: n pairs in one launch -- against the oracle (border-1 quirk of EvaluateError kept,
    dense_sfm.h:91) and against the blocking single-pair operators."""
    from deepfactors_amd import synth
    w, h, cs, n = 192, 144, 32, 7
    ctx = dfx.Context(0)
    al, se3 = dfx.SfmAligner(code_size=cs, ctx=ctx), dfx.SE3Aligner(ctx=ctx)
    plist, slist, keep = [], [], []
    for k in range(n):
        p = synth.make_pair(w, h, cs, seed=900 + k, device="cpu", motion_scale=0.4 + 0.2 * k)
        nn, g = synth.to_numpy(p), synth.to_device(p, "cuda")
        keep.append((nn, g))
        plist.append(dict(pose0=nn["pose0"], pose1=nn["pose1"], cam=nn["cam"], img0=g["img0"], img1=g["img1"], dpt0=g["dpt0"],
                          prx0_jac=g["prx_jac"], grad1=g["grad1"]))
        slist.append(dict(se3=nn["pose10_true"] if k % 2 else synth.IDENTITY, cam=nn["cam"], img0=g["img0"], img1=g["img1"], dpt0=g["dpt0"], grad1=g["grad1"]))
    errs = al.EvaluateErrorBatch(al.make_pairs(plist))
    steps = se3.RunStepBatch(se3.make_pairs(slist))
    for k, (nn, g) in enumerate(keep):
        ref_res, ref_inl = oracle.sfm_error(nn["pose0"], nn["pose1"], nn["cam"], nn["img0"], nn["img1"], nn["dpt0"], 0.1)
        assert errs[k].inliers == ref_inl
        assert abs(errs[k].residual - ref_res) <= 1e-4 * abs(ref_res) + 1e-6
        one = al.EvaluateError(nn["pose0"], nn["pose1"], nn["cam"], g["img0"], g["img1"], g["dpt0"], None, g["grad1"])
        assert one.inliers == errs[k].inliers and abs(one.residual - errs[k].residual) <= 2e-6 * abs(one.residual)
        sref = oracle.se3_step(slist[k]["se3"], nn["cam"], nn["img0"], nn["img1"], nn["dpt0"], nn["grad1"], 0.1)
        assert_item_close(steps[k], sref, w, h, what=f"se3 batch pair {k}")
    # device-side outputs of the async forms equal the blocking forms bit for bit
    ed = torch.zeros(16 * n, dtype=torch.uint8, device="cuda")
    sd = torch.zeros(dfx.item_size(6) * n, dtype=torch.uint8, device="cuda")
    al.EvaluateErrorBatch(al.make_pairs(plist), ed)
    se3.RunStepBatch(se3.make_pairs(slist), sd)
    ctx.sync()
    e = ed.cpu().numpy()
    for k in range(n):
        assert np.frombuffer(e[16 * k:16 * k + 4].tobytes(), np.float32)[0] == np.float32(errs[k].residual)
        assert int(np.frombuffer(e[16 * k + 8:16 * k + 16].tobytes(), np.uint64)[0]) == errs[k].inliers
    assert np.array_equal(sd.cpu().numpy(), np.concatenate([s.raw for s in steps]))
    # an identity pair gives the same item whatever the poses of the other pairs of its launch are
    ident = [dict(s, se3=synth.IDENTITY) for s in slist]
    steps_id = se3.RunStepBatch(se3.make_pairs(ident))
    for k in range(0, n, 2):
        assert np.array_equal(steps_id[k].raw, steps[k].raw), k
    # the measurement hook brackets the reduction kernel of the batched forms too
    ctx.set_profiling(True)
    ctx.profile_read()
    al.EvaluateErrorBatch(al.make_pairs(plist), ed)
    se3.RunStepBatch(se3.make_pairs(slist), sd)
    nl, ms = ctx.profile_read()
    ctx.set_profiling(False)
    assert nl == 2 and 0.0 < ms < 50.0


def test_snapped_taps_never_touch_what_lies_behind_a_row(dfx, oracle):
    """The row walk snaps a tap coordinate that is less than 2^-13 pixel below an integer onto it.  An inlier whose u lies that
    close under W - 1 must keep its taps at columns W - 2, W - 1: a right-hand tap at column W would read the row's padding (or the next row) --
    with a weight of zero, which does not neutralise a NaN.  Constructed case: constant depth 1, a pure x translation of (1 - 6e-5) / fx, so
    column W - 2 of every row maps to u = W - 1 - 6e-5; img1 / grad1 pitched with NaN behind every row and NaN rows behind the image."""
    from deepfactors_amd import synth
    w, h = 192, 144
    p = synth.make_pair(w, h, 16, seed=77, device="cpu", with_decoder=False)
    n = synth.to_numpy(p)
    cam = n["cam"]
    n["dpt0"] = np.ones((h, w), np.float32)
    for shift_px, axis in ((1.0 - 6e-5, 0), (1.0 - 6e-5, 1)):
        pose = synth.IDENTITY.copy()
        pose[4 + axis] = np.float32(shift_px / float(cam[axis]))        # u = x + fx tx / d   (v = y + fy ty / d)

        def nan_padded(a, pad_cols, pad_rows):
            t = torch.from_numpy(a).cuda()
            big = torch.full((a.shape[0] + pad_rows, a.shape[1] + pad_cols) + tuple(a.shape[2:]), float("nan"), dtype=t.dtype, device="cuda")
            big[: a.shape[0], : a.shape[1]] = t
            return big[: a.shape[0], : a.shape[1]]
        i0, d0 = torch.from_numpy(n["img0"]).cuda(), torch.from_numpy(n["dpt0"]).cuda()
        i1, g1 = nan_padded(n["img1"], 8, 3), nan_padded(n["grad1"], 8, 3)
        se3, al = dfx.SE3Aligner(), dfx.SfmAligner(code_size=16)
        s_got = se3.RunStep(pose, cam, i0, i1, d0, g1)
        s_ref = oracle.se3_step(pose, cam, n["img0"], n["img1"], n["dpt0"], n["grad1"], 0.1)
        assert np.all(np.isfinite(s_got.JtJ)) and np.isfinite(s_got.residual)
        assert s_got.inliers == s_ref.inliers
        assert_item_close(s_got, s_ref, w, h, what=f"snap at the border, axis {axis}")
        e_got = al.EvaluateError(synth.IDENTITY, pose, cam, i0, i1, d0, None, g1)
        e_res, e_inl = oracle.sfm_error(synth.IDENTITY, pose, cam, n["img0"], n["img1"], n["dpt0"], 0.1)
        assert np.isfinite(e_got.residual) and e_got.inliers == e_inl
        assert abs(e_got.residual - e_res) <= 1e-4 * abs(e_res) + 1e-6
