"""The pixel reductions (SE3 step, EvaluateError, the device-resident tracker) fetch the right-hand column of a lane's 2 x 2 bilinear taps from the
NEXT lane's own tap loads where the two lanes tap neighbouring cells ("neighbour exchange", dfx_misc_kernels.hip row_walk<.., NX>), and load it
themselves elsewhere.  Same taps, same arithmetic: every result must equal the direct tap loads (DFX_RW_NX=0) BIT FOR BIT -- at the identity,
at real poses, under rotations about the optical axis and scale changes (floor(u) / floor(v) jump between neighbouring lanes many times per
row), on widths that are not a multiple of the 64-pixel band, and on pitched images.  The switch is read once per process, so each setting runs
in its own interpreter.  (Parity with the oracle is asserted by tests/test_gpu_parity.py etc., which run on the default = the exchange.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, %r)
import deepfactors_amd as dfx
from deepfactors_amd import synth
ctx = dfx.Context(0)
al, se3 = dfx.SfmAligner(code_size=16, ctx=ctx), dfx.SE3Aligner(ctx=ctx)
hsh = hashlib.sha256()
def pitched(t, pad):
    if t.dim() == 2:
        buf = torch.full((t.shape[0], t.shape[1] + pad), float("nan"), dtype=t.dtype, device=t.device)
        buf[:, : t.shape[1]] = t
        return buf[:, : t.shape[1]]
    buf = torch.full((t.shape[0], t.shape[1] + pad, t.shape[2]), float("nan"), dtype=t.dtype, device=t.device)
    buf[:, : t.shape[1]] = t
    return buf[:, : t.shape[1]]
rng = np.random.default_rng(77)
for (w, h) in ((640, 480), (200, 150), (129, 97), (64, 48), (70, 9)):
    plist, slist, keep = [], [], []
    for k in range(6):
        p = synth.make_pair(w, h, 16, seed=9100 + k, device="cuda", motion_scale=0.4 + 0.3 * k)
        if k %% 2:
            for name in ("img0", "img1", "dpt0", "grad1"):
                p[name] = pitched(p[name], 3 + k)
        keep.append(p)
        # poses: the exact identity, the generating pose, and perturbed poses with a roll about the optical axis (the tap row changes along a
        # band) and a forward / backward motion (the tap column advances by more / less than one cell per lane)
        R = synth.so3_exp(np.array([0.01 * k, -0.008 * k, 0.02 * (k - 2)]))
        pose = synth.pose_qt(R, np.array([0.03 * (k - 2), 0.01 * k, 0.12 * (k - 3)]))
        for q in (synth.IDENTITY, p["pose10_true"], pose):
            slist.append(dict(se3=q, cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], grad1=p["grad1"]))
        plist.append(dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], prx0_jac=p["prx_jac"], grad1=p["grad1"]))
        plist.append(dict(pose0=synth.IDENTITY, pose1=pose, cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], prx0_jac=p["prx_jac"], grad1=p["grad1"]))
    items = se3.RunStepBatch(se3.make_pairs(slist))
    errs = al.EvaluateErrorBatch(al.make_pairs(plist))
    inl = sum(it.inliers for it in items)
    assert inl > 0.2 * w * h * len(items) or w < 100, (w, h, inl)
    for it in items:
        hsh.update(it.raw.tobytes())
    for e in errs:
        hsh.update(np.array([e.residual], np.float32).tobytes() + np.array([e.inliers], np.uint64).tobytes())
    # the blocking single-pair operators (1024 workgroups: other segment shapes)
    for q in slist[:4]:
        hsh.update(se3.RunStep(q["se3"], q["cam"], q["img0"], q["img1"], q["dpt0"], q["grad1"]).raw.tobytes())
    for q in plist[:3]:
        e = al.EvaluateError(q["pose0"], q["pose1"], q["cam"], q["img0"], q["img1"], q["dpt0"], None, None)
        hsh.update(np.array([e.residual], np.float32).tobytes() + np.array([e.inliers], np.uint64).tobytes())
# the device-resident tracker
p = synth.make_pair(320, 240, 16, seed=0xDF01, device="cuda", with_decoder=False)
trk = dfx.CameraTracker([p["cam"]], dfx.TrackerConfig(1, (12,), 0.1), ctx)
trk.SetKeyframe([p["img0"]], [p["dpt0"]])
pose = trk.TrackFrame([p["img1"]], [p["grad1"]])
hsh.update(np.asarray(pose, np.float32).tobytes())
print("DIGEST", hsh.hexdigest())
''' % ROOT


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST ")]
    assert lines, r.stdout[-500:]
    return lines[-1].split()[1]


def test_neighbour_exchange_equals_the_direct_tap_loads_bit_for_bit():
    assert _run({"DFX_RW_NX": "1"}) == _run({"DFX_RW_NX": "0"})
