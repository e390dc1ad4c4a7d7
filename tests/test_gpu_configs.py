"""BASELINE.json configurations against the CPU oracle at their real sizes, through the C ABI:
  configs[4]  1280x960, 64-code (the occupancy-2 kernel variant, 12 MFMAs + 4 tile products per group) and 640x480x64;
  configs[1]  640x480x32 with PITCHED Jacobian rows (per-vector addressing variant of the step kernel);
  configs[2]  a 16-keyframe window: 120 pairs per pyramid level in ONE launch, all pairs of a keyframe sharing its
              prx_jac / dpt0 / valid0 buffers (concurrent 1.0-writes into one valid0 image), levels 0..2;
  the global-ray-table kernel variant at a realistic height; image regions that leave the view (whole chunks without a
  correspondence) with bit-determinism over many runs.
Reference: tests/ut_sfmaligner.cpp:235-327 (GPU vs host evaluation of the same inputs: inliers equal, |dJtJ| <= 1e-1); the
tolerance here is tests/helpers.py: every entry within 1e-4 of its own Cauchy-Schwarz scale sqrt(JtJ_ii JtJ_jj) against the fp64-accumulating oracle."""
import numpy as np
import pytest
import torch

from helpers import assert_item_close

pytestmark = pytest.mark.gpu


def _pair_dev(w, h, cs, seed, **kw):
    """Synthetic pair generated on the GPU (seconds at 1280x960x64), plus its host copy for the oracle."""
    from deepfactors_amd import synth
    g = synth.make_pair(w, h, cs, seed=seed, device="cuda", **kw)
    return synth.to_numpy(g), g


@pytest.mark.parametrize("w,h,cs", [(1280, 960, 64), (640, 480, 64), (1280, 960, 32)])
def test_sfm_step_matches_oracle_at_full_size(dfx, oracle, w, h, cs):
    n, g = _pair_dev(w, h, cs, seed=0xDF05)
    pose1 = n["pose1"].copy(); pose1[4] += 0.01
    al = dfx.SfmAligner(code_size=cs)
    valid_gpu = torch.zeros_like(g["img0"])
    got = al.RunStep(n["pose0"], pose1, n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], g["std0"], valid_gpu, g["prx_jac"], g["grad1"])
    valid_ref = np.zeros_like(n["img0"])
    ref = oracle.sfm_step(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], valid0=valid_ref, accum_f64=True)
    assert ref.inliers > 0.8 * w * h
    assert_item_close(got, ref, w, h, what=f"sfm_step {w}x{h} cs={cs}")
    assert int((valid_gpu.cpu().numpy() != valid_ref).sum()) <= max(1, int(1e-5 * w * h))
    # EvaluateError and UpdateDepth at the same size
    e_got = al.EvaluateError(n["pose0"], pose1, n["cam"], g["img0"], g["img1"], g["dpt0"], None, g["grad1"])
    e_res, e_inl = oracle.sfm_error(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], 0.1)
    assert abs(int(e_got.inliers) - e_inl) <= max(1, int(1e-5 * w * h))
    assert abs(e_got.residual - e_res) <= 1e-4 * e_res
    out = torch.empty_like(g["img0"])
    dfx.UpdateDepth(n["code"], g["prx_orig"], g["prx_jac"], 2.0, out)
    d_ref = oracle.update_depth(n["code"], n["prx_orig"], n["prx_jac"], 2.0)
    assert np.abs(out.cpu().numpy() - d_ref).max() <= 2e-6 * float(((2.0 + d_ref) ** 2 / 2.0).max())


def _pitched(t, pad):
    shape = list(t.shape); shape[1] += pad
    big = torch.full(shape, float("nan"), dtype=t.dtype, device=t.device)
    big[:, : t.shape[1]] = t
    return big[:, : t.shape[1]]


@pytest.mark.parametrize("w,h,cs,jpad", [(640, 480, 32, 64), (640, 480, 32, 0), (320, 240, 64, 4 * 64)])
def test_pitched_jacobian_at_full_size(dfx, oracle, w, h, cs, jpad):
    """jpad > 0 selects the per-vector addressing variant of the step kernel (JDENSE = false); every other image is pitched too."""
    n, g = _pair_dev(w, h, cs, seed=0xDF06)
    al = dfx.SfmAligner(code_size=cs)
    valid_gpu = _pitched(torch.zeros_like(g["img0"]), 20)
    valid_gpu.zero_()
    got = al.RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], _pitched(g["img0"], 28), _pitched(g["img1"], 12), _pitched(g["dpt0"], 4), None,
                     valid_gpu, _pitched(g["prx_jac"], jpad) if jpad else g["prx_jac"], _pitched(g["grad1"], 6))
    valid_ref = np.zeros_like(n["img0"])
    ref = oracle.sfm_step(n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], valid0=valid_ref)
    assert_item_close(got, ref, w, h, what=f"pitched {w}x{h} cs={cs} jpad={jpad}")
    assert int((valid_gpu.cpu().numpy() != valid_ref).sum()) <= max(1, int(1e-5 * w * h))


def test_global_ray_table_variant_at_a_realistic_height(dfx, oracle):
    """W + H + 80 floats beyond what fits beside the workgroup's static LDS (64 KB): TABLDS = false.  64 rows, 655 k pixels."""
    w, h, cs = 10240, 64, 16
    n, g = _pair_dev(w, h, cs, seed=0xDF07)
    al = dfx.SfmAligner(code_size=cs)
    got = al.RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, None, g["prx_jac"], g["grad1"])
    ref = oracle.sfm_step(n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"])
    assert ref.inliers > 0.5 * w * h
    assert_item_close(got, ref, w, h, what="10240x64 global ray table")


def _rand_pose(rng, synth, trs, rot):
    R = synth.so3_exp(rng.normal(0, rot, 3))
    return synth.pose_qt(R, rng.normal(0, trs, 3))


def test_keyframe_window_120_pairs_share_keyframe_buffers(dfx, oracle):
    """BASELINE configs[2]: 16 keyframes, every pair i < j (120) per pyramid level in one launch.  All pairs out of keyframe i
    read ITS prx_jac / dpt0 and write ITS valid0 (15 concurrent writers of 1.0 at most); level-1/-2 cameras are halved
    (camera_pyramid.h:41-46).  Every item of levels 1 and 2 and every fourth item of level 0 is compared with the oracle;
    the valid0 images must equal the union of the oracle's masks."""
    from deepfactors_amd import synth
    K, cs = 16, 32
    rng = np.random.default_rng(0xDF03)
    poses = [_rand_pose(rng, synth, 0.01, 0.006) for _ in range(K)]
    al = dfx.SfmAligner(code_size=cs)
    for lvl, (w, h) in enumerate([(640, 480), (320, 240), (160, 120)]):
        kfs = [_pair_dev(w, h, cs, seed=0x1600 + k) for k in range(K)]   # same seeds at every level: the same scenes, halved cameras
        valid = [torch.zeros_like(g["img0"]) for _, g in kfs]
        plist, idx = [], []
        for i in range(K):
            for j in range(i + 1, K):
                ni, gi = kfs[i]; nj, gj = kfs[j]
                plist.append(dict(pose0=poses[i], pose1=poses[j], cam=ni["cam"], img0=gi["img0"], img1=gj["img0"], dpt0=gi["dpt0"], valid0=valid[i],
                                  prx0_jac=gi["prx_jac"]))
                idx.append((i, j))
        # gradient of the target image: Sobel / 8 of img0 of keyframe j (the Frame::FillPyramids product, frame.h:84-90)
        grads = []
        for _, g in kfs:
            gr = torch.empty((h, w, 2), dtype=torch.float32, device="cuda")
            dfx.SobelGradients(g["img0"], gr)
            grads.append(gr)
        for d, (i, j) in zip(plist, idx):
            d["grad1"] = grads[j]
        assert len(plist) == 120
        items = al.RunStepBatch(al.make_pairs(plist))
        vref = [np.zeros((h, w), np.float32) for _ in range(K)]
        step = 4 if lvl == 0 else 1
        grads_n = [g.cpu().numpy() for g in grads]
        for q, (i, j) in enumerate(idx):
            if q % step:
                continue
            ni, nj = kfs[i][0], kfs[j][0]
            ref = oracle.sfm_step(poses[i], poses[j], ni["cam"], ni["img0"], nj["img0"], ni["dpt0"], ni["prx_jac"], grads_n[j], valid0=vref[i])
            assert ref.inliers > 0.5 * w * h
            assert_item_close(items[q], ref, w, h, what=f"level {lvl} pair {i}->{j}")
        if step == 1:
            for k in range(K - 1):
                assert int((valid[k].cpu().numpy() != vref[k]).sum()) <= max(1, int(1e-5 * w * h) * (K - 1 - k)), f"valid0 of keyframe {k} level {lvl}"
        else:   # every pixel some compared pair marked must be marked; nothing is marked outside the image interior rule
            for k in range(K - 1):
                v = valid[k].cpu().numpy()
                assert np.all(v[vref[k] == 1.0] == 1.0) and set(np.unique(v)) <= {0.0, 1.0}
        assert float(valid[K - 1].abs().max()) == 0.0   # the last keyframe is nobody's keyframe 0


def test_regions_out_of_view_are_deterministic_and_match_the_oracle(dfx, oracle):
    """Large motion + a band of NaN depth: several whole 64-pixel chunks (and whole chunk rows) have no correspondence at all
    while their neighbours do.  Their tap loads must not disturb the counted waits of the software pipeline: results are
    bit-identical over 40 runs, alone and inside a batch next to benign pairs, and equal the oracle's."""
    from deepfactors_amd import synth
    w, h, cs = 640, 480, 32
    n, g = _pair_dev(w, h, cs, seed=0xDF08)
    R = synth.so3_exp(np.array([0.0, 0.35, 0.02]))           # 20 degrees about y: a third of the columns leaves the view
    pose1 = synth.pose_qt(R.T, -R.T @ np.array([0.4, 0.05, 0.0]))
    dpt = g["dpt0"].clone()
    dpt[100:140, :] = float("nan")                            # 40 full rows = 400 chunks without a correspondence
    dpt[300:330, 200:520] = float("nan")
    dpt_n = dpt.cpu().numpy()
    from deepfactors_amd import _lib
    ctx = dfx.Context()
    ctx.set_schedule(_lib.DFX_SCHEDULE_STATIC)   # bit-reproducibility is the static schedule's contract (the dynamic one: test below)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    ref = oracle.sfm_step(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], dpt_n, n["prx_jac"], n["grad1"])
    assert 0.2 * w * h < ref.inliers < 0.7 * w * h
    first = al.RunStep(n["pose0"], pose1, n["code"], n["cam"], g["img0"], g["img1"], dpt, None, None, g["prx_jac"], g["grad1"])
    assert_item_close(first, ref, w, h, what="out-of-view regions")
    for _ in range(40):
        again = al.RunStep(n["pose0"], pose1, n["code"], n["cam"], g["img0"], g["img1"], dpt, None, None, g["prx_jac"], g["grad1"])
        assert np.array_equal(first.raw, again.raw)
    n2, g2 = _pair_dev(w, h, cs, seed=0xDF09)
    bad = dict(pose0=n["pose0"], pose1=pose1, cam=n["cam"], img0=g["img0"], img1=g["img1"], dpt0=dpt, prx0_jac=g["prx_jac"], grad1=g["grad1"])
    good = dict(pose0=n2["pose0"], pose1=n2["pose1"], cam=n2["cam"], img0=g2["img0"], img1=g2["img1"], dpt0=g2["dpt0"], prx0_jac=g2["prx_jac"], grad1=g2["grad1"])
    arr = al.make_pairs([good, bad, good, bad, bad, good, bad, good] * 4)
    ref_good = oracle.sfm_step(n2["pose0"], n2["pose1"], n2["cam"], n2["img0"], n2["img1"], n2["dpt0"], n2["prx_jac"], n2["grad1"])
    base = None
    for r in range(10):
        items = al.RunStepBatch(arr)
        raw = np.stack([it.raw for it in items])
        if base is None:
            base = raw
            assert_item_close(items[1], ref, w, h, what="batched out-of-view pair")
            assert_item_close(items[0], ref_good, w, h, what="batched benign pair")
        assert np.array_equal(raw, base), f"run {r} differs"


def test_linearize_batch_is_update_depth_then_step(dfx, oracle):
    """dfx_sfm_linearize_batch = PhotometricFactor::RunAlignmentStep over a batch (photometric_factor.cpp:225-293): UpdateDepthMaps once
    per distinct keyframe, then RunStep.  Against the oracle's update_depth o sfm_step, for pairs that share keyframes (decoded
    once) and pairs that do not; the keyframes' depth maps end up decoded; conflicting codes for one depth map are refused."""
    from deepfactors_amd import synth
    w, h, cs, K = 160, 120, 32, 3
    rng = np.random.default_rng(8)
    kfs = [_pair_dev(w, h, cs, seed=0x2200 + k) for k in range(K)]
    codes = [rng.normal(0, 0.3, cs).astype(np.float32) for _ in range(K)]
    poses = [_rand_pose(rng, synth, 0.01, 0.006) for _ in range(K)]
    dpt = [torch.full_like(g["img0"], float("nan")) for _, g in kfs]      # stale depth maps: must be overwritten before they are read
    idx = [(0, 1), (0, 2), (1, 2), (2, 0), (1, 0)]
    al = dfx.SfmAligner(code_size=cs)
    arr = al.make_pairs([dict(pose0=poses[i], pose1=poses[j], cam=kfs[i][0]["cam"], img0=kfs[i][1]["img0"], img1=kfs[j][1]["img0"], dpt0=dpt[i],
                              prx0_jac=kfs[i][1]["prx_jac"], grad1=kfs[j][1]["grad1"]) for i, j in idx])
    items = al.LinearizeBatch(arr, [kfs[i][1]["prx_orig"] for i, _ in idx], [codes[i] for i, _ in idx])
    for q, (i, j) in enumerate(idx):
        ni, nj = kfs[i][0], kfs[j][0]
        d_ref = oracle.update_depth(codes[i], ni["prx_orig"], ni["prx_jac"], 2.0)
        assert np.abs(dpt[i].cpu().numpy() - d_ref).max() <= 2e-6 * float(((2.0 + d_ref) ** 2 / 2.0).max())
        ref = oracle.sfm_step(poses[i], poses[j], ni["cam"], ni["img0"], nj["img0"], dpt[i].cpu().numpy(), ni["prx_jac"], nj["grad1"])
        assert ref.inliers > 0.5 * w * h
        assert_item_close(items[q], ref, w, h, what=f"linearize pair {i}->{j}")
    # async form into device memory gives the same bytes
    out = torch.zeros(len(idx) * dfx.item_size(12 + cs), dtype=torch.uint8, device="cuda")
    al.LinearizeBatch(arr, [kfs[i][1]["prx_orig"] for i, _ in idx], [codes[i] for i, _ in idx], out)
    al.ctx.sync()
    for a, b in zip(items, al.items_from_bytes(out.cpu().numpy(), cs)):
        assert np.array_equal(a.raw, b.raw)
    with pytest.raises(dfx.DfxError):   # two pairs of keyframe 0 with different codes
        al.LinearizeBatch(arr, [kfs[i][1]["prx_orig"] for i, _ in idx], [codes[i] if q != 1 else codes[1] for q, (i, _) in enumerate(idx)])
    # batched decoder alone == the single-image operator, bit for bit
    outs_b = [torch.empty_like(g["img0"]) for _, g in kfs]
    dfx.UpdateDepthBatch(codes, [g["prx_orig"] for _, g in kfs], [g["prx_jac"] for _, g in kfs], 2.0, outs_b)
    for k, (_, g) in enumerate(kfs):
        one = torch.empty_like(g["img0"])
        dfx.UpdateDepth(codes[k], g["prx_orig"], g["prx_jac"], 2.0, one)
        assert torch.equal(one, outs_b[k])


@pytest.mark.parametrize("w,h,n,mode", [(640, 480, 20, "dynamic"), (128, 96, 130, "dynamic")])
def test_dynamic_schedule_matches_the_oracle_and_the_static_schedule(dfx, oracle, w, h, n, mode):
    """Resident wave-workers popping items from per-pair queues (opt-in, DFX_SCHEDULE_DYNAMIC; teams of 204 and of 31 waves): every item
    still equals the oracle at the stated tolerance and the static schedule to fp32
    re-association; valid0 images are written alike; repeated launches (the queues are rewound by the finalize kernel) stay
    correct; pairs with regions out of view and mixed cameras included."""
    from deepfactors_amd import _lib, synth
    cs = 32
    rng = np.random.default_rng(12)
    host, dev = [], []
    for k in range(6):
        nk, gk = _pair_dev(w, h, cs, seed=0x3300 + k, motion_scale=0.5 + 0.2 * k)
        if k == 3:   # another camera for one keyframe
            nk["cam"] = nk["cam"].copy(); nk["cam"][0] *= 1.05; nk["cam"][2] += 2.0
        if k == 4:   # a band without correspondences
            gk["dpt0"][h * 5 // 12:h * 13 // 24, :] = float("nan"); nk["dpt0"] = gk["dpt0"].cpu().numpy()
        host.append(nk); dev.append(gk)
    idx = [(int(rng.integers(0, 6)), int(rng.integers(0, 6))) for _ in range(n)]
    plist, valid_dyn, valid_sta = [], [torch.zeros_like(g["img0"]) for g in dev], [torch.zeros_like(g["img0"]) for g in dev]
    def pairs(valid):
        out = []
        for (i, j) in idx:
            pose1 = host[j]["pose1"].copy(); pose1[4] += 0.004 * j
            out.append(dict(pose0=host[i]["pose0"], pose1=pose1, cam=host[i]["cam"], img0=dev[i]["img0"], img1=dev[j]["img1"], dpt0=dev[i]["dpt0"], valid0=valid[i],
                            prx0_jac=dev[i]["prx_jac"], grad1=dev[j]["grad1"]))
        return out
    ctx_d, ctx_s = dfx.Context(), dfx.Context()
    ctx_s.set_schedule(_lib.DFX_SCHEDULE_STATIC)
    ctx_d.set_schedule(_lib.DFX_SCHEDULE_DYNAMIC)
    al_d, al_s = dfx.SfmAligner(code_size=cs, ctx=ctx_d), dfx.SfmAligner(code_size=cs, ctx=ctx_s)
    arr_d, arr_s = al_d.make_pairs(pairs(valid_dyn)), al_s.make_pairs(pairs(valid_sta))
    it_s = al_s.RunStepBatch(arr_s)
    for rep in range(3):
        it_d = al_d.RunStepBatch(arr_d)
        assert ctx_d.last_schedule_dynamic() and not ctx_s.last_schedule_dynamic()
        for q, (i, j) in enumerate(idx):
            assert it_d[q].inliers == it_s[q].inliers
            sc = float(np.abs(it_s[q].JtJ).max())
            assert np.abs(it_d[q].JtJ.astype(np.float64) - it_s[q].JtJ).max() <= 3e-6 * sc, (rep, q)
            assert abs(it_d[q].residual - it_s[q].residual) <= 1e-5 * it_s[q].residual
    for q in range(0, n, 3 if n < 64 else 17):
        i, j = idx[q]
        pose1 = host[j]["pose1"].copy(); pose1[4] += 0.004 * j
        ref = oracle.sfm_step(host[i]["pose0"], pose1, host[i]["cam"], host[i]["img0"], host[j]["img1"], host[i]["dpt0"], host[i]["prx_jac"], host[j]["grad1"])
        assert_item_close(it_d[q], ref, w, h, what=f"dynamic schedule pair {q}")
    for a, b in zip(valid_dyn, valid_sta):
        assert torch.equal(a, b)
    assert any(float(v.max()) == 1.0 for v in valid_dyn)
    # a batch of ONE pair never takes the dynamic schedule (its descriptor travels in the kernel arguments, which the wave-worker grid
    # does not read): the forced context runs the static partition and returns the static bits
    one_d, one_s = al_d.RunStepBatch(al_d.make_pairs(pairs(valid_dyn)[:1])), al_s.RunStepBatch(al_s.make_pairs(pairs(valid_sta)[:1]))
    assert not ctx_d.last_schedule_dynamic()
    assert np.array_equal(one_d[0].raw, one_s[0].raw)


PYR3 = [(640, 480), (320, 240), (160, 120)]
PYR4_CFG4 = [(1280, 960), (640, 480), (320, 240), (160, 120)]   # BASELINE configs[4]: "1280x960 input, 4-level pyramid, 64-dim code"
PYR_SMALL = [(160, 120), (80, 60), (40, 30)]


@pytest.mark.parametrize("cs,sizes,sets,mode", [(32, PYR3, 4, "auto"), (32, PYR3, 4, "f32"),
                                                (64, PYR4_CFG4, 2, "auto"), (64, PYR4_CFG4, 2, "f32"),
                                                (64, PYR_SMALL, 5, "auto"), (16, PYR_SMALL, 5, "auto")])
def test_pyramid_levels_in_one_launch(dfx, oracle, cs, sizes, sets, mode):
    """A batch whose pairs differ in image size: the pyramid levels of several factor sets in ONE launch (1-D grid, workgroups per pair in
    proportion to the pixel count, large pairs first).  Every item against the fp64 oracle, against the same pairs launched level by level,
    bit-reproducible, valid0 maps (library-owned, shadowed) written per level; then the same through dfx_sfm_linearize_batch (one decoder
    launch per image size).  CS = 32: levels 640x480 ... 160x120 of four factor sets (the one-workgroup-per-pair tail kernel).  CS = 64,
    1280x960 ... 160x120: BASELINE configs[4] AS STATED -- four levels (decoder_network.cpp:258-259; deepfactors_options.h:43,83) -- whose
    largest pair has 3.2 MB of partials, so the launch takes the per-tile finalize kernel in its mixed-size form (DFX_TAIL_MAX_KB,
    dfx_sfm_step.hip launch_t); CS = 64 / 16 on small levels: the tail kernel's mixed-size form at the other code sizes."""
    from deepfactors_amd import _lib, synth
    ctx = dfx.Context(0)
    ctx.set_mfma_mode(_lib.DFX_MFMA_AUTO if mode == "auto" else _lib.DFX_MFMA_F32_CHAIN)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    nlv = len(sizes)
    plist, meta = [], []
    for k in range(sets):
        for lv, (w, h) in enumerate(sizes):
            g = synth.make_pair(w, h, cs, seed=0x9A0 + k, device="cuda", motion_scale=0.5 + 0.15 * k)
            n = synth.to_numpy(g)
            vld = ctx.alloc_image(w, h)
            plist.append(dict(pose0=n["pose0"], pose1=n["pose1"], cam=n["cam"], img0=g["img0"], img1=g["img1"], dpt0=g["dpt0"], prx0_jac=g["prx_jac"],
                              grad1=g["grad1"], valid0=vld))
            meta.append((w, h, n, g, vld))
    # interleaved order on purpose: the library sorts the workgroups by size itself
    items = al.RunStepBatch(al.make_pairs(plist))
    again = al.RunStepBatch(al.make_pairs(plist))
    for q, (it, (w, h, n, g, vld)) in enumerate(zip(items, meta)):
        vref = np.zeros_like(n["img0"])
        ref = oracle.sfm_step(n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], valid0=vref, accum_f64=True)
        assert_item_close(it, ref, w, h, what=f"mixed batch pair {q} ({w}x{h})")
        assert np.array_equal(it.raw, again[q].raw), "a mixed batch is bit-reproducible like any static launch"
        v = vld.download()
        assert int((v != vref).sum()) <= max(1, int(1e-5 * w * h))
        assert np.array_equal(vld.valid0_shadow(), v == 1.0)
    assert ctx.last_mfma_mode() == (_lib.DFX_MFMA_F32_CHAIN if mode == "f32" else _lib.DFX_MFMA_BF16X3)
    # level by level (one image size per launch): same sums up to fp32 re-association (the number of workgroups per pair differs)
    for lv in range(nlv):
        sel = [q for q in range(len(plist)) if q % nlv == lv]
        sub = al.RunStepBatch(al.make_pairs([plist[q] for q in sel]))
        for q, it in zip(sel, sub):
            assert it.inliers == items[q].inliers
            assert np.abs(it.JtJ.astype(np.float64) - items[q].JtJ).max() <= 3e-6 * np.abs(items[q].JtJ).max()
    # the decoder in front: dpt0 of every pair is re-decoded from (prx_orig, prx_jac, code), one launch per size, then the mixed step
    prx = [m[3]["prx_orig"] for m in meta]
    codes = np.stack([m[2]["code"] for m in meta])
    lin = al.LinearizeBatch(al.make_pairs(plist), prx, codes)
    for q, it in enumerate(lin):
        assert abs(int(it.inliers) - int(items[q].inliers)) <= (0 if meta[q][0] <= 640 else 2)   # (a depth that differs in the last bit can move a border pixel)
        assert np.abs(it.JtJ.astype(np.float64) - items[q].JtJ).max() <= 2e-5 * np.abs(items[q].JtJ).max()   # dpt0 is decoded in fp32 here, in fp64 by the generator
