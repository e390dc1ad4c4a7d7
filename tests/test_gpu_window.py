"""A keyframe window end to end: F frames in a chain, F - 1 photometric pairs, unknowns (pose_1..pose_{F-1}, code_0..code_{F-2});
every Gauss-Newton step = UpdateDepth per keyframe + ONE batched RunStep + the assembly of the keyframe graph's block-sparse
normal equations (dfx_graph_assemble_async) + a dense solve of the system on the host.  Validates the block placement
and signs of the assembled system (PhotometricFactor::linearize's G11..G33 / g1..g3, photometric_factor.cpp:105-161) by the
only test that matters for a solver: it converges to the generating poses and codes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _R(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_keyframe_window_gauss_newton(dfx):
    from deepfactors_amd import synth
    from deepfactors_amd.dist import NormalEquations, PairGraph
    w, h, cs, F = 160, 120, 16, 4
    D = 6 + cs
    prs = [synth.make_pair(w, h, cs, seed=60 + k, device="cuda", motion_scale=0.6 + 0.2 * k) for k in range(F - 1)]
    # world poses of the chain: T_0 = I, T_{k+1} = T_k * T10_k^-1   (pose_10 = pose1^-1 * pose0 must equal the pair's true motion)
    Rw, tw = [np.eye(3)], [np.zeros(3)]
    for p in prs:
        R10, t10 = _R(p["pose10_true"][:4]), p["pose10_true"][4:].astype(np.float64)
        Rw.append(Rw[-1] @ R10.T)
        tw.append(tw[-1] - Rw[-1] @ t10)
    truth_pose = [synth.pose_qt(R, t) for R, t in zip(Rw, tw)]
    truth_code = [p["code"] for p in prs]
    # start: perturbed poses (pose_0 is the gauge), half-size codes
    rng = np.random.default_rng(5)
    pose = [truth_pose[0].copy()] + [synth.pose_qt(synth.so3_exp(rng.normal(0, 2e-3, 3)) @ Rw[k], tw[k] + rng.normal(0, 5e-3, 3)) for k in range(1, F)]
    code = [(0.5 * c).astype(np.float32) for c in truth_code]

    al = dfx.SfmAligner(code_size=cs)
    neq = NormalEquations(PairGraph.chain(F - 1), cs, "cuda")
    items = torch.zeros((F - 1) * dfx.item_size(12 + cs), dtype=torch.uint8, device="cuda")
    dpt = [torch.empty_like(p["img0"]) for p in prs]
    hist = []
    for it in range(12):
        for k, p in enumerate(prs):
            dfx.UpdateDepth(code[k], p["prx_orig"], p["prx_jac"], 2.0, dpt[k], al.ctx)
        arr = al.make_pairs([dict(pose0=pose[k], pose1=pose[k + 1], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=dpt[k],
                                  prx0_jac=p["prx_jac"], grad1=p["grad1"]) for k, p in enumerate(prs)])
        al.RunStepBatchAssembleAsync(arr, items, neq, 0)
        al.ctx.sync()
        its = al.items_from_bytes(items.cpu().numpy(), cs)
        hist.append(sum(i.residual for i in its))
        H = neq.dense().numpy()
        g = neq.g.detach().cpu().double().numpy().reshape(-1)
        # unknowns: everything except pose_0 (gauge) and code_{F-1} (no factor touches it)
        keep = np.ones(F * D, bool)
        keep[0:6] = False
        keep[(F - 1) * D + 6:] = False
        A = H[np.ix_(keep, keep)]
        A = A + 1e-4 * np.diag(np.diag(A))
        d = np.zeros(F * D)
        d[keep] = -np.linalg.solve(A, g[keep])
        for k in range(1, F):
            dk = d[k * D:k * D + 6]
            pose[k] = synth.pose_qt(synth.so3_exp(dk[3:]) @ _R(pose[k][:4]), pose[k][4:].astype(np.float64) + dk[:3])
        for k in range(F - 1):
            code[k] = (code[k] + d[k * D + 6:(k + 1) * D]).astype(np.float32)
    assert hist[-1] < 0.05 * hist[0] and abs(hist[-1] - hist[-2]) < 1e-2 * hist[-1], hist
    for k in range(1, F):
        assert np.abs(pose[k][4:] - truth_pose[k][4:]).max() < 3e-3, (k, pose[k], truth_pose[k])
        assert np.abs(_R(pose[k][:4]) - Rw[k]).max() < 2e-3
    for k in range(F - 1):
        assert np.abs(code[k] - truth_code[k]).max() < 0.25 * np.abs(0.5 * truth_code[k]).max()
