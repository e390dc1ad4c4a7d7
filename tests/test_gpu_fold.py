"""The reduction operators (SE3Aligner::RunStep / Warp, SfmAligner::EvaluateError, SquaredError, their batched forms, the device-resident
tracker) as ONE launch each: the last workgroup of a pair to arrive folds the pair's partial rows (block_reduce_fold,
dfx_misc_kernels.hip).  Same bits as the two-kernel form (DFX_FOLD=0: k_finalize_rows / k_track_update as a second launch), call after
call (the arrival counters rewind), at sizes with one and with a thousand workgroups per pair."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _contexts(dfx):
    """(folded context, two-kernel context): DFX_FOLD is read when a context is created"""
    old = os.environ.pop("DFX_FOLD", None)
    try:
        folded = dfx.Context(0)
        os.environ["DFX_FOLD"] = "0"
        plain = dfx.Context(0)
    finally:
        if old is None:
            os.environ.pop("DFX_FOLD", None)
        else:
            os.environ["DFX_FOLD"] = old
    return folded, plain


@pytest.mark.parametrize("w,h", [(640, 480), (96, 64), (21, 13), (1280, 960)])
def test_single_pair_operators_fold_equals_two_kernels(dfx, w, h):
    from deepfactors_amd import synth
    cs = 16
    p = synth.make_pair(w, h, cs, seed=0x6100 + w, device="cuda")
    res = []
    for ctx in _contexts(dfx):
        se3, al = dfx.SE3Aligner(ctx=ctx), dfx.SfmAligner(code_size=cs, ctx=ctx)
        out = []
        for rep in range(4):   # the counters are back at zero for every call
            it = se3.RunStep(synth.IDENTITY, p["cam"], p["img0"], p["img1"], p["dpt0"], p["grad1"])
            img2 = torch.zeros_like(p["img0"])
            wr = se3.Warp(synth.IDENTITY, p["cam"], p["img0"], p["img1"], p["dpt0"], img2)
            er = al.EvaluateError(p["pose0"], p["pose1"], p["cam"], p["img0"], p["img1"], p["dpt0"], None, p["grad1"])
            sq = dfx.SquaredError(p["img0"], p["img1"], ctx=ctx)
            out.append((it.raw.copy(), np.asarray([wr.residual, wr.inliers], np.float64), np.asarray([er.residual, er.inliers], np.float64), float(sq)))
        res.append(out)
    folded, plain = res
    assert folded[0][1][1] > 0 and np.isfinite(folded[0][3])
    for rep in range(4):
        for a, b in zip(folded[rep], plain[0]):
            assert np.array_equal(np.asarray(a), np.asarray(b)), (w, h, rep)


def test_batched_operators_and_tracker_fold_equals_two_kernels(dfx, oracle):
    from deepfactors_amd import synth
    from test_gpu_tracker import _pyramid_np
    cs, w, h, P = 16, 320, 240, 9
    prs = [synth.make_pair(w, h, cs, seed=0x6200 + k, device="cuda", motion_scale=0.3 + 0.1 * k) for k in range(P)]
    n = synth.to_numpy(synth.make_pair(320, 240, 16, seed=21, with_decoder=False))
    cams = synth.camera_pyramid(n["cam"], 3)
    lv = _pyramid_np(oracle, n["img0"], n["img1"], n["dpt0"], 3)
    g = [{k: torch.from_numpy(v).cuda() for k, v in l.items()} for l in lv]
    res = []
    for ctx in _contexts(dfx):
        se3, al = dfx.SE3Aligner(ctx=ctx), dfx.SfmAligner(code_size=cs, ctx=ctx)
        sarr = se3.make_pairs([dict(se3=synth.IDENTITY, cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], grad1=p["grad1"]) for p in prs])
        earr = al.make_pairs([dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], prx0_jac=p["prx_jac"],
                                   grad1=p["grad1"]) for p in prs])
        sitems = torch.zeros(P * dfx.item_size(6), dtype=torch.uint8, device="cuda")
        eitems = torch.zeros(P * 16, dtype=torch.uint8, device="cuda")
        snaps = []
        for rep in range(3):
            se3.RunStepBatch(sarr, sitems)
            al.EvaluateErrorBatch(earr, eitems)
            ctx.sync()
            snaps.append((sitems.cpu().numpy().copy(), eitems.cpu().numpy().copy()))
        tr = dfx.CameraTracker(cams, dfx.TrackerConfig(3, (10, 5, 5), 0.1), ctx=ctx)
        tr.SetKeyframe([l["img0"] for l in g], [l["dpt0"] for l in g])
        pose = tr.TrackFrame([l["img1"] for l in g], [l["grad1"] for l in g]).copy()
        stats = (tr.last_result_.iterations, tr.last_result_.solver_failures, tr.GetInliers(), tr.GetError())
        # N trackers per launch (relocalisation): the same frame against three keyframes
        kfs = [([l["img0"] for l in g], [l["dpt0"] for l in g])] * 3
        rb = tr.TrackFrameBatch(kfs, [l["img1"] for l in g], [l["grad1"] for l in g])
        batch = [(list(r.pose_ck.q) + list(r.pose_ck.t), r.error, r.inliers_frac) for r in rb]
        res.append((snaps, pose, stats, batch))
    (s_f, pose_f, stats_f, batch_f), (s_p, pose_p, stats_p, batch_p) = res
    for rep in range(3):
        assert np.array_equal(s_f[rep][0], s_p[0][0]) and np.array_equal(s_f[rep][1], s_p[0][1]), rep
    assert np.any(s_f[0][0] != 0) and np.any(s_f[0][1] != 0)
    assert np.array_equal(pose_f, pose_p) and stats_f == stats_p and stats_f[0] == 20 and stats_f[1] == 0
    assert batch_f == batch_p
