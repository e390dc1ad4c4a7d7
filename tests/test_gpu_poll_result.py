"""Blocking single-result calls return as soon as the host reads the call's sequence number, which the call's last kernel stores BEHIND the result in
mapped host memory (deepfactors_amd/csrc/dfx_kernels.hpp, DoneFlag) -- not when the stream reports idle.  A result read before it landed would be the
previous call's: alternate two different inputs a few hundred times per operator and compare every result with that input's (bit-exact: the kernels are
deterministic).  Covers the single-workgroup finalize kernel (SE3 step, EvaluateError, Warp, SquaredError), the multi-workgroup one with its arrival
counter (SfmAligner::RunStep at CS 16 / 32 / 64 in both evaluation modes, DepthAligner) and the tracker's final kernel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPS = 300


def _pairs(cs, w=320, h=240):
    from deepfactors_amd import synth
    return [synth.make_pair(w, h, cs, seed=0x7100 + k, device="cuda") for k in range(2)]


def _alternate(call, reps=REPS):
    """call(k) -> bytes for input k in {0, 1}; every repetition must reproduce the first answer for that input."""
    want = [call(0), call(1)]
    assert want[0] != want[1]
    for r in range(reps):
        k = (r * 7 + r // 3) & 1
        assert call(k) == want[k], f"repetition {r}: the result of input {k} differs from its first evaluation"


def test_se3_step_error_warp_squared_error(dfx):
    from deepfactors_amd import synth
    prs = _pairs(16)
    se3, sfm = dfx.SE3Aligner(), dfx.SfmAligner(code_size=16)
    out = torch.empty_like(prs[0]["img0"])
    _alternate(lambda k: se3.RunStep(prs[k]["pose10_true"], prs[k]["cam"], prs[k]["img0"], prs[k]["img1"], prs[k]["dpt0"], prs[k]["grad1"]).raw.tobytes())
    _alternate(lambda k: repr(sfm.EvaluateError(prs[k]["pose0"], prs[k]["pose1"], prs[k]["cam"], prs[k]["img0"], prs[k]["img1"], prs[k]["dpt0"], None, None).__dict__))
    _alternate(lambda k: repr(se3.Warp(synth.IDENTITY, prs[k]["cam"], prs[k]["img0"], prs[k]["img1"], prs[k]["dpt0"], out).__dict__))
    _alternate(lambda k: np.float32(dfx.SquaredError(prs[k]["img0"], prs[1 - k]["dpt0"])).tobytes())


@pytest.mark.parametrize("cs", [16, 32, 64])
@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_sfm_step_single_pair(dfx, cs, mode):
    from deepfactors_amd import _lib
    prs = _pairs(cs)
    ctx = dfx.Context(0)
    ctx.set_mfma_mode(_lib.DFX_MFMA_BF16X3 if mode == "bf16x3" else _lib.DFX_MFMA_F32_CHAIN)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    _alternate(lambda k: al.RunStep(prs[k]["pose0"], prs[k]["pose1"], None, prs[k]["cam"], prs[k]["img0"], prs[k]["img1"], prs[k]["dpt0"], None, prs[k]["valid0"],
                                    prs[k]["prx_jac"], prs[k]["grad1"]).raw.tobytes(), reps=150)


def test_depth_aligner_and_tracker(dfx):
    from deepfactors_amd import synth
    prs = _pairs(32)
    da = dfx.DepthAligner(code_size=32)
    _alternate(lambda k: da.RunStep(prs[k]["code"], prs[1 - k]["dpt0"], prs[k]["prx_orig"], prs[k]["prx_jac"], 2.0).raw.tobytes(), reps=150)
    trk = [dfx.CameraTracker([p["cam"]], dfx.TrackerConfig(1, (6,), 0.1)) for p in prs]
    for t, p in zip(trk, prs):
        t.SetKeyframe([p["img0"]], [p["dpt0"]])

    def frame(k):
        trk[k].Reset()
        return trk[k].TrackFrame([prs[k]["img1"]], [prs[k]["grad1"]]).tobytes() + np.float32(trk[k].GetError()).tobytes()
    _alternate(frame, reps=150)


def test_both_waits_give_the_same_bytes_and_bad_modes_are_refused(dfx):
    from deepfactors_amd import _lib
    prs = _pairs(32)
    ctx = dfx.Context(0)
    al, se3 = dfx.SfmAligner(code_size=32, ctx=ctx), dfx.SE3Aligner(ctx=ctx)
    out = {}
    for mode in (_lib.DFX_WAIT_POLL, _lib.DFX_WAIT_STREAM, _lib.DFX_WAIT_POLL):
        ctx.set_result_wait(mode)
        p = prs[0]
        got = (al.RunStep(p["pose0"], p["pose1"], None, p["cam"], p["img0"], p["img1"], p["dpt0"], None, p["valid0"], p["prx_jac"], p["grad1"]).raw.tobytes(),
               se3.RunStep(p["pose10_true"], p["cam"], p["img0"], p["img1"], p["dpt0"], p["grad1"]).raw.tobytes())
        assert out.setdefault("first", got) == got
    with pytest.raises(dfx.DfxError):
        ctx.set_result_wait(7)
