"""Device-resident CameraTracker::TrackFrame (dfx_track_frame) vs the host-driven loop of the reference
(camera_tracker.cpp:42-71) evaluated with the oracle, and the reference's own ImageAlignmentTest criterion."""
import os

import numpy as np
import pytest
import torch
from scipy import ndimage

pytestmark = pytest.mark.gpu


def _pyramid_np(oracle, img0, img1, dpt0, levels):
    out = [dict(img0=img0, img1=img1, dpt0=dpt0, grad1=oracle.sobel(img1))]
    for _ in range(1, levels):
        q = out[-1]
        i0, i1 = oracle.blur_down(q["img0"]), oracle.blur_down(q["img1"])
        out.append(dict(img0=i0, img1=i1, dpt0=np.ascontiguousarray(q["dpt0"][::2, ::2][: i0.shape[0], : i0.shape[1]]), grad1=oracle.sobel(i1)))
    return out


def test_tracker_matches_host_loop(dfx, oracle):
    from deepfactors_amd import synth
    n = synth.to_numpy(synth.make_pair(320, 240, 16, seed=21, with_decoder=False))
    cams = synth.camera_pyramid(n["cam"], 3)
    lv = _pyramid_np(oracle, n["img0"], n["img1"], n["dpt0"], 3)
    iters = (10, 5, 5)   # level 0 .. 2 (flags tracking_iters=5,5,10 are coarse-to-fine)
    # host loop, reference style: RunStep -> ldlt solve -> retract
    qt = synth.IDENTITY.copy()
    for level in (2, 1, 0):
        for _ in range(iters[level]):
            r = oracle.se3_step(qt, cams[level], lv[level]["img0"], lv[level]["img1"], lv[level]["dpt0"], lv[level]["grad1"], 0.1)
            qt = oracle.se3_solve_update(r.JtJ, r.Jtr, qt)
    g = [{k: torch.from_numpy(v).cuda() for k, v in l.items()} for l in lv]
    tr = dfx.CameraTracker(cams, dfx.TrackerConfig(3, iters, 0.1))
    tr.SetKeyframe([l["img0"] for l in g], [l["dpt0"] for l in g])
    pose = tr.TrackFrame([l["img1"] for l in g], [l["grad1"] for l in g])
    assert tr.last_result_.iterations == 20 and tr.last_result_.solver_failures == 0
    assert np.linalg.norm(pose[4:] - qt[4:]) < 1e-4 and np.linalg.norm(pose[:4] - qt[:4]) < 1e-4
    gt = n["pose10_true"]
    assert np.linalg.norm(pose[4:] - gt[4:]) < 2e-3 and np.linalg.norm(pose[:4] - gt[:4]) < 1e-3
    assert tr.GetInliers() > 0.9 and tr.GetError() < 1e-4
    # a second call continues from the converged estimate (pose_ck_ persists, camera_tracker.cpp:59-63)
    pose2 = tr.TrackFrame([l["img1"] for l in g], [l["grad1"] for l in g])
    assert np.linalg.norm(pose2 - pose) < 1e-4


def test_tracker_reference_fixture_kat(dfx, oracle):
    """ut_se3aligner.cpp:173-211 through the device-resident loop: 40 iterations from identity, residual/inliers <= 1e-3."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "se3_fixture_1047_1052.npz"))
    img0 = ndimage.uniform_filter(d["img0"].astype(np.float32) / np.float32(255), 25, mode="mirror")
    img1 = ndimage.uniform_filter(d["img1"].astype(np.float32) / np.float32(255), 25, mode="mirror")
    dpt0 = d["dpt0_mm"].astype(np.float32) / np.float32(1000)
    cam = np.array([np.float32(160 / 0.5773502691896257), np.float32(120 / 0.41421356237309503), 160, 120, 320, 240], np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    grad = torch.empty((240, 320, 2), dtype=torch.float32, device="cuda")
    dfx.SobelGradients(t(img1), grad)
    tr = dfx.CameraTracker([cam], dfx.TrackerConfig(1, (40,), 0.1))
    tr.SetKeyframe([t(img0)], [t(dpt0)])
    pose = tr.TrackFrame([t(img1)], [grad])
    assert tr.GetError() <= 1e-3 and tr.GetInliers() > 0.9
    # same answer as the oracle-driven loop
    qt = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    g1 = oracle.sobel(img1)
    for _ in range(40):
        r = oracle.se3_step(qt, cam, img0, img1, dpt0, g1, 0.1)
        qt = oracle.se3_solve_update(r.JtJ, r.Jtr, qt)
    assert np.linalg.norm(pose - qt) < 2e-4


def test_tracker_no_overlap_is_reported(dfx):
    from deepfactors_amd import synth
    p = synth.make_pair(128, 96, 16, seed=22, device="cuda", with_decoder=False)
    tr = dfx.CameraTracker([p["cam"]], dfx.TrackerConfig(1, (3,), 0.1))
    tr.SetKeyframe([p["img0"]], [p["dpt0"]])
    tr.SetPoseEstimate(np.array([0, 0, 0, 1, 100.0, 0, 0], np.float32))
    tr.TrackFrame([p["img1"]], [p["grad1"]])
    assert tr.GetInliers() == 0.0 and tr.GetError() == float("inf") and tr.last_result_.solver_failures == 3


def test_relocalize_batch_equals_sequential(dfx, oracle):
    """DeepFactors::Relocalize (deepfactors.cpp:713-743): the live frame tracked against N keyframes in ONE schedule of
    launches gives, per keyframe, the bytes the one-after-the-other loop gives, and picks the keyframe it belongs to."""
    from deepfactors_amd import synth
    import time
    cfg = dfx.TrackerConfig(3, (10, 5, 5), 0.1)
    pairs = [synth.to_numpy(synth.make_pair(320, 240, 16, seed=30 + k, with_decoder=False)) for k in range(5)]
    cams = synth.camera_pyramid(pairs[0]["cam"], 3)
    # live frame = img1 of pair 2; keyframes = img0/dpt0 of all five pairs (only keyframe 2 sees the same scene)
    live = _pyramid_np(oracle, pairs[2]["img0"], pairs[2]["img1"], pairs[2]["dpt0"], 3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    live_img, live_grad = [t(l["img1"]) for l in live], [t(l["grad1"]) for l in live]
    kfs = []
    for p in pairs:
        lv = _pyramid_np(oracle, p["img0"], p["img1"], p["dpt0"], 3)
        kfs.append(([t(l["img0"]) for l in lv], [t(l["dpt0"]) for l in lv]))
    tr = dfx.CameraTracker(cams, cfg)
    seq = []
    for kimg, kdpt in kfs:   # the reference's loop
        tr.SetKeyframe(kimg, kdpt)
        tr.Reset()
        pose = tr.TrackFrame(live_img, live_grad)
        seq.append((pose.copy(), tr.GetError(), tr.GetInliers()))
    res = tr.TrackFrameBatch(kfs, live_img, live_grad)
    for k, r in enumerate(res):
        pose = np.array(list(r.pose_ck.q) + list(r.pose_ck.t), np.float32)
        assert np.array_equal(pose, seq[k][0]), k
        assert (r.error == seq[k][1] or (np.isinf(r.error) and np.isinf(seq[k][1]))) and r.inliers_frac == np.float32(seq[k][2])
    best, pose = tr.Relocalize(kfs, live_img, live_grad)
    assert best == 2 and best == int(np.argmin([s[1] for s in seq]))
    assert np.array_equal(pose, seq[2][0]) and tr.GetError() == seq[2][1]
    gt = pairs[2]["pose10_true"]
    assert np.linalg.norm(pose[4:] - gt[4:]) < 2e-3 and np.linalg.norm(pose[:4] - gt[:4]) < 1e-3
    # mismatched schedules are rejected before any work
    with pytest.raises(dfx.DfxError):
        bad = [(kfs[0][0], kfs[0][1]), ([kfs[1][0][0][:100].contiguous()] + kfs[1][0][1:], kfs[1][1])]
        tr.TrackFrameBatch(bad, live_img, live_grad)
