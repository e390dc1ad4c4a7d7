"""Calls the library REFUSES although the reference (or an earlier round of this ABI) would have run them -- each refusal must be loud, with a
message that names the problem -- and the guarantee that nothing in the environment steers a context.  Listed for integrators in INTEGRATION.md 3a."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_camera_of_another_size_is_refused(dfx):
    """fill_simple (dfx_api.cpp): SE3 step / Warp / their batch forms / the tracker decide validity in the CAMERA's coordinates and address taps in the
    image's, so a camera whose w x h is not the image's (a zeroed cam.w / cam.h, a full-resolution camera on a coarser pyramid level) is an error, not
    silently out-of-bounds taps.  The reference's SE3Aligner does not check."""
    from deepfactors_amd import synth
    p = synth.make_pair(128, 96, 16, seed=5, device="cuda", with_decoder=False)
    se3 = dfx.SE3Aligner()
    good = se3.RunStep(synth.IDENTITY, p["cam"], p["img0"], p["img1"], p["dpt0"], p["grad1"])
    assert good.inliers > 0
    for cam in (np.array([*p["cam"][:4], 0, 0], np.float32), np.array([*p["cam"][:4], 256, 192], np.float32)):
        with pytest.raises(dfx.DfxError, match="camera is .* images are 128x96"):
            se3.RunStep(synth.IDENTITY, cam, p["img0"], p["img1"], p["dpt0"], p["grad1"])
        with pytest.raises(dfx.DfxError, match="camera is"):
            se3.Warp(synth.IDENTITY, cam, p["img0"], p["img1"], p["dpt0"], torch.empty_like(p["img0"]))
        with pytest.raises(dfx.DfxError, match="camera is"):
            se3.RunStepBatch(se3.make_pairs([dict(se3=synth.IDENTITY, cam=cam, img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], grad1=p["grad1"])]))
        with pytest.raises(dfx.DfxError, match="camera is"):
            tr = dfx.CameraTracker([cam], dfx.TrackerConfig(1, (2,), 0.1))
            tr.SetKeyframe([p["img0"]], [p["dpt0"]])
            tr.TrackFrame([p["img1"]], [p["grad1"]])
    again = se3.RunStep(synth.IDENTITY, p["cam"], p["img0"], p["img1"], p["dpt0"], p["grad1"])   # the context is usable after a refusal
    assert np.array_equal(again.raw, good.raw)


def test_unknown_options_and_code_sizes_are_refused(dfx):
    from deepfactors_amd import _lib
    ctx = dfx.Context(0)
    ctx.configure(_lib.DFX_OPT_STEP_DESC_ZEROCOPY, 1)
    ctx.configure(_lib.DFX_OPT_STEP_DESC_ZEROCOPY, 0)
    with pytest.raises(dfx.DfxError, match="unknown option"):
        ctx.configure(99, 1)
    with pytest.raises(dfx.DfxError, match="takes 0 or 1"):
        ctx.configure(_lib.DFX_OPT_SIMPLE_DESC_ZEROCOPY, 7)
    with pytest.raises(dfx.DfxError):
        ctx.set_mfma_mode(7)
    with pytest.raises(dfx.DfxError):
        ctx.set_result_wait(5)
    from deepfactors_amd import synth
    p = synth.make_pair(64, 48, 16, seed=6, device="cuda")
    al = dfx.SfmAligner(code_size=24, ctx=ctx)     # the kernels are instantiated for CS 16 / 32 / 64 (the reference: one compile-time DF_CODE_SIZE)
    with pytest.raises(dfx.DfxError, match="unsupported code size 24"):
        al.RunStep(p["pose0"], p["pose1"], None, p["cam"], p["img0"], p["img1"], p["dpt0"], None, None, p["prx_jac"], p["grad1"])


def test_no_environment_variable_steers_a_context():
    """Rounds 3-5 read DFX_MFMA / DFX_SCHEDULE / DFX_POLL_RESULT / DFX_*_DESC_ZEROCOPY and tuning aids from the environment -- a stray variable could
    change result bits of a drop-in build.  They are gone: with every one of them set to something (even garbage, which used to FAIL dfx_ctx_create)
    a fresh process computes the same bytes, in the library's default mode."""
    worker = r'''
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
import deepfactors_amd as dfx
from deepfactors_amd import synth, _lib
ctx = dfx.Context(0)
p = synth.make_pair(192, 144, 32, seed=77, device="cuda")
al = dfx.SfmAligner(code_size=32, ctx=ctx)
it = al.RunStep(p["pose0"], p["pose1"], None, p["cam"], p["img0"], p["img1"], p["dpt0"], None, None, p["prx_jac"], p["grad1"])
print("DIGEST", hashlib.sha256(np.asarray(it.raw).tobytes()).hexdigest(), ctx.last_mfma_mode())
''' % ROOT
    stray = dict(DFX_MFMA="garbage", DFX_SCHEDULE="dynamic", DFX_POLL_RESULT="0", DFX_SIMPLE_DESC_ZEROCOPY="0", DFX_STEP_DESC_ZEROCOPY="1", DFX_CPW_MAX="7",
                 DFX_DYN_ROWS="3", DFX_BATCH_WGS_PER_CU="5", DFX_TRACK_BLOCKS="64", DFX_TAIL_ORDERED="0")
    outs = []
    clean = {k: v for k, v in os.environ.items() if k not in stray}
    for env in (clean, dict(clean, **stray)):
        r = subprocess.run([sys.executable, "-c", worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST ")][-1])
    assert outs[0] == outs[1], outs
